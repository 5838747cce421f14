/* oracle/knn_selftest.c — sanitizer harness for the C oracle (TEST INFRASTRUCTURE ONLY).
 * Built with -fsanitize=address,undefined by `make -C oracle selftest` and run by tests/test_oracle_pins.py:
 * ragged events (empty, 1, 2, k, k+1, many pulses, duplicated positions), every supported k, both modes; checks
 * in-range indices, no self loops, event locality, degree bookkeeping and (d2, j) order.  Exit code 0 = clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t gn_oracle_knn_graph(const float *x, int64_t ld, const int32_t *cols, int32_t D, const int64_t *ptr, int32_t B,
                            int32_t k, int32_t mode, int32_t *nbr, int32_t *deg);

static float d2(const float *x, int64_t ld, int64_t i, int64_t j) {
    float s = 0.0f;
    for (int d = 0; d < 3; ++d) { const float t = x[j * ld + d] - x[i * ld + d]; s = s + t * t; }
    return s;
}

int main(void) {
    const int sizes[] = {0, 1, 2, 5, 8, 9, 10, 33, 0, 257, 64, 3};
    const int B = (int)(sizeof(sizes) / sizeof(sizes[0]));
    int64_t ptr[16];
    ptr[0] = 0;
    for (int b = 0; b < B; ++b) ptr[b + 1] = ptr[b] + sizes[b];
    const int64_t N = ptr[B], ld = 5;
    float *x = (float *)malloc(sizeof(float) * (size_t)(N * ld));
    uint32_t s = 12345u;
    for (int64_t i = 0; i < N * ld; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) / 16777216.0f; }
    for (int64_t i = ptr[9]; i < ptr[9] + 40; ++i)          /* 40 pulses on one position: more than k ties at d2 = 0 */
        for (int d = 0; d < 3; ++d) x[i * ld + d] = x[ptr[9] * ld + d];
    const int32_t cols[3] = {0, 1, 2};
    int bad = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int k = 1; k <= 16; k += (k < 8 ? 1 : 8)) {
            int32_t *nbr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N * (k + 1)));
            int32_t *deg = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
            const int64_t E = gn_oracle_knn_graph(x, ld, cols, 3, ptr, B, k, mode, nbr, deg);
            int64_t total = 0;
            for (int b = 0; b < B; ++b)
                for (int64_t i = ptr[b]; i < ptr[b + 1]; ++i) {
                    total += deg[i];
                    const int n = sizes[b];
                    const int cap = mode == 0 ? k + 1 : k;
                    if (deg[i] > cap || deg[i] > n - 1 || (deg[i] < k && deg[i] != n - 1)) { ++bad; }
                    for (int e = 0; e < k + 1; ++e) {
                        const int32_t j = nbr[i * (k + 1) + e];
                        if (e >= deg[i]) { if (j != -1) ++bad; continue; }
                        if (j < ptr[b] || j >= ptr[b + 1] || j == i) ++bad;
                        if (e > 0) {
                            const int32_t jp = nbr[i * (k + 1) + e - 1];
                            const float a = d2(x, ld, i, jp), c = d2(x, ld, i, j);
                            if (c < a || (c == a && j < jp)) ++bad;
                        }
                    }
                }
            if (total != E) ++bad;
            free(nbr); free(deg);
        }
    if (gn_oracle_knn_graph(x, ld, cols, 3, ptr, B, 0, 0, NULL, NULL) != -1) ++bad;     /* bad k is refused */
    free(x);
    printf("knn_selftest: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
