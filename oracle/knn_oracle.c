/*
 * oracle/knn_oracle.c — CPU restatement of the k-NN graph builder on the DynEdge hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under graphnet_amd/ may import, link or call this file;
 * it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the arithmetic lives in third-party wheels that are absent from
 * /root/reference and from this image (torch-cluster >= 1.6, torch-geometric >= 2.3;
 * reference pins: setup.py:52-59).  The reference's own tests hold no golden vector for
 * knn_graph, so this restates the PUBLISHED algorithm of
 * torch_cluster's brute-force kernel and anchors on the reference call sites:
 *
 *   src/graphnet/models/graphs/edges/edges.py:72-80      KNNEdges._construct_edges
 *       graph.edge_index = knn_graph(graph.x[:, columns], k, graph.batch)
 *   src/graphnet/models/components/layers.py:63-67       DynEdgeConv.forward (re-kNN)
 *       edge_index = knn_graph(x=x[:, features_subset], k=nb_neighbors, batch=batch)
 *
 * Published semantics restated (torch_geometric.nn.knn_graph, loop=False,
 * flow="source_to_target"; torch_cluster.knn brute force):
 *   for each query i, inside its own event only:
 *     best[0..k] (k+1 slots, INCLUDING the query itself) initialised to (1e10, -1);
 *     candidates j are scanned in ascending index order;
 *     d2 = sum_d (x[j,d]-x[i,d])^2 accumulated left to right in fp32, no FMA contraction;
 *     j is inserted before the first slot whose distance is STRICTLY greater than d2
 *       -> total order (d2, j);
 *   edges (j -> i) for every filled slot with j != i     (degree k, or k+1 when the query
 *     itself is not among its k+1 best because > k other points tie at d2 == 0).
 *   edge_index[0] = j (source / neighbour), edge_index[1] = i (target / centre),
 *   grouped by i ascending, ascending (d2, j) inside a group.
 *
 * mode 0 ("compat", default)  = the above.
 * mode 1 ("strict")           = self excluded by index, k slots, degree = min(k, n_i - 1).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <stdint.h>
#include <stddef.h>

#define GN_ORACLE_KMAX 64

/*
 * x      : [N, ld] fp32 row-major; distance uses columns cols[0..D-1]
 * ptr    : [B+1] event offsets (int64), ptr[0] = 0, ptr[B] = N
 * nbr    : [N, k+1] int32 out, -1 padded; row i holds the neighbours of centre i in order
 * deg    : [N] int32 out, number of valid entries in row i
 * returns total number of edges, or -1 on bad arguments
 */
int64_t gn_oracle_knn_graph(const float *x, int64_t ld, const int32_t *cols, int32_t D,
                            const int64_t *ptr, int32_t B, int32_t k, int32_t mode,
                            int32_t *nbr, int32_t *deg)
{
    if (k < 1 || k + 1 > GN_ORACLE_KMAX || D < 1) return -1;
    const int32_t kk = (mode == 0) ? k + 1 : k;
    int64_t total = 0;
    for (int32_t b = 0; b < B; ++b) {
        const int64_t lo = ptr[b], hi = ptr[b + 1];
        for (int64_t i = lo; i < hi; ++i) {
            float best_d[GN_ORACLE_KMAX];
            int64_t best_j[GN_ORACLE_KMAX];
            for (int32_t e = 0; e < kk; ++e) { best_d[e] = 1e10f; best_j[e] = -1; }
            for (int64_t j = lo; j < hi; ++j) {
                if (mode != 0 && j == i) continue;
                float d2 = 0.0f;
                for (int32_t d = 0; d < D; ++d) {
                    const float diff = x[j * ld + cols[d]] - x[i * ld + cols[d]];
                    const float sq = diff * diff;
                    d2 = d2 + sq;
                }
                for (int32_t e1 = 0; e1 < kk; ++e1) {
                    if (best_d[e1] > d2) {
                        for (int32_t e2 = kk - 1; e2 > e1; --e2) {
                            best_d[e2] = best_d[e2 - 1];
                            best_j[e2] = best_j[e2 - 1];
                        }
                        best_d[e1] = d2;
                        best_j[e1] = j;
                        break;
                    }
                }
            }
            int32_t c = 0;
            for (int32_t e = 0; e < kk; ++e) {
                if (best_j[e] < 0 || best_j[e] == i) continue;
                nbr[i * (int64_t)(k + 1) + c] = (int32_t)best_j[e];
                ++c;
            }
            deg[i] = c;
            total += c;
            for (; c < k + 1; ++c) nbr[i * (int64_t)(k + 1) + c] = -1;
        }
    }
    return total;
}

/*
 * Minkowski-metric k-NN (the only k-NN known-answer test the reference holds:
 * tests/models/test_minkowski.py:104-160, following
 * src/graphnet/models/graphs/edges/minkowski.py:12-81).  Used solely to pin the
 * selection/ordering code of this oracle against those golden edge lists.
 *   dist[i][j] = sum_{d<3} (x_i - x_j)^2 - (c (t_i - t_j))^2   (space_coords 0..2, time 3)
 *   negative (time-like) entries are mapped to -time_like_weight * dist before ranking;
 *   per centre: the k smallest non-self entries, ascending (the edge list the test expects).
 * out_src/out_dst: [N*k] int64 (source = neighbour, target = centre).
 */
int64_t gn_oracle_minkowski_knn(const float *x, int64_t ld, int32_t n, int32_t k, float c,
                                float time_like_weight,
                                int64_t *out_src, int64_t *out_dst, float *out_dist)
{
    int64_t e = 0;
    for (int32_t i = 0; i < n; ++i) {
        float best_d[GN_ORACLE_KMAX];
        int32_t best_j[GN_ORACLE_KMAX];
        for (int32_t s = 0; s < k; ++s) { best_d[s] = 3.0e38f; best_j[s] = -1; }
        for (int32_t j = 0; j < n; ++j) {
            float sp = 0.0f;
            for (int32_t d = 0; d < 3; ++d) {
                const float diff = x[i * ld + d] - x[j * ld + d];
                sp = sp + diff * diff;
            }
            const float dt = x[i * ld + 3] - x[j * ld + 3];
            const float tc = dt * c;
            float m = sp - tc * tc;
            if (out_dist) out_dist[(int64_t)i * n + j] = m;
            if (j == i) continue;
            if (m < 0.0f) m = m * (-time_like_weight);   /* minkowski.py:88 */
            for (int32_t e1 = 0; e1 < k; ++e1) {
                if (best_d[e1] > m) {
                    for (int32_t e2 = k - 1; e2 > e1; --e2) {
                        best_d[e2] = best_d[e2 - 1];
                        best_j[e2] = best_j[e2 - 1];
                    }
                    best_d[e1] = m;
                    best_j[e1] = j;
                    break;
                }
            }
        }
        for (int32_t s = 0; s < k; ++s) {
            if (best_j[s] < 0) continue;
            out_src[e] = best_j[s];
            out_dst[e] = i;
            ++e;
        }
    }
    return e;
}
