// graphnet_amd/csrc/attn.hip — ragged (ptr-driven) multi-head self attention for DynTrans
// (reference models/components/layers.py:166-197: to_dense_batch -> TransformerEncoder with key-padding mask ->
// x[mask]).  No padded [B, Lmax, d] tensor exists here: a query tile of 64 pulses (event-aligned, the k-NN tile
// plan of graph.hip) scans the keys of ITS OWN event only, so the work is sum_e n_e^2 instead of B * Lmax^2.
//
// First version: exact-fp32 flash formulation on the vector ALU (one lane = one query / one key, K and V
// tiles broadcast out of LDS, online softmax in the exp2 domain).  Forward keeps lse2 = log2(sum_j 2^(s2_ij)) per
// (query, head); backward is the usual two passes (queries own dQ, keys own dK / dV), no atomics.
#include "common.hpp"
#include "launchers.hpp"

namespace gn {

constexpr int ATT_TILE = 64;
constexpr float LOG2E = 1.4426950408889634f;

// event of query tile `tile`: largest e with tile_ptr[e] <= tile  (wave-uniform)
__device__ __forceinline__ int attn_event_of_tile(const int* __restrict__ tile_ptr, int B, int tile) {
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_ptr[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// stage rows [r0, r0+64) (clamped to r_last) x DH floats of column block `col0` into LDS [64][DH]
template <int DH>
__device__ __forceinline__ void attn_stage(float* __restrict__ dst, const float* __restrict__ src, long long ld,
                                           int col0, int r0, int r_last, int lane) {
    constexpr int V4 = DH / 4;                       // float4 per row
#pragma unroll
    for (int it = 0; it < V4; ++it) {
        const int idx = it * 64 + lane;              // 0 .. 64*V4-1
        const int r = idx / V4, c4 = idx % V4;
        const int row = min(r0 + r, r_last);
        const float4 v = *reinterpret_cast<const float4*>(src + (long long)row * ld + col0 + c4 * 4);
        *reinterpret_cast<float4*>(dst + r * DH + c4 * 4) = v;
    }
}

template <int DH>
__global__ __launch_bounds__(64) void attn_fwd_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale2, float* __restrict__ out, long long ldo, float* __restrict__ lse2) {
    __shared__ float Ks[ATT_TILE * DH];
    __shared__ float Vs[ATT_TILE * DH];
    const int tile = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int q0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE;
    const int E = H * DH;
    const int qi = q0 + lane;
    const bool valid = qi < kend;
    const int qrow = valid ? qi : kend - 1;
    float q[DH], o[DH];
#pragma unroll
    for (int d4 = 0; d4 < DH / 4; ++d4) {
        const float4 v = *reinterpret_cast<const float4*>(qkv + (long long)qrow * ld + head * DH + d4 * 4);
        q[4 * d4] = v.x * scale2; q[4 * d4 + 1] = v.y * scale2; q[4 * d4 + 2] = v.z * scale2; q[4 * d4 + 3] = v.w * scale2;
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    for (int kt = kbeg; kt < kend; kt += ATT_TILE) {
        __syncthreads();
        attn_stage<DH>(Ks, qkv, ld, E + head * DH, kt, kend - 1, lane);
        attn_stage<DH>(Vs, qkv, ld, 2 * E + head * DH, kt, kend - 1, lane);
        __syncthreads();
        const int nk = min(ATT_TILE, kend - kt);
        for (int c = 0; c < nk; c += 8) {
            float s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float* kr = Ks + (c + t) * DH;
                float acc = 0.0f;
#pragma unroll
                for (int d4 = 0; d4 < DH / 4; ++d4) {
                    const float4 kk = *reinterpret_cast<const float4*>(kr + 4 * d4);
                    acc = fmaf(q[4 * d4], kk.x, acc); acc = fmaf(q[4 * d4 + 1], kk.y, acc);
                    acc = fmaf(q[4 * d4 + 2], kk.z, acc); acc = fmaf(q[4 * d4 + 3], kk.w, acc);
                }
                s[t] = (c + t < nk) ? acc : -INFINITY;
            }
            float mc = s[0];
#pragma unroll
            for (int t = 1; t < 8; ++t) mc = fmaxf(mc, s[t]);
            const float mn = fmaxf(m, mc);
            const float corr = exp2f(m - mn);
            l *= corr;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] *= corr;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float p = exp2f(s[t] - mn);
                l += p;
                const float* vr = Vs + (c + t) * DH;
#pragma unroll
                for (int d4 = 0; d4 < DH / 4; ++d4) {
                    const float4 vv = *reinterpret_cast<const float4*>(vr + 4 * d4);
                    o[4 * d4] = fmaf(p, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vv.y, o[4 * d4 + 1]);
                    o[4 * d4 + 2] = fmaf(p, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vv.w, o[4 * d4 + 3]);
                }
            }
            m = mn;
        }
    }
    if (valid) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4)
            *reinterpret_cast<float4*>(out + (long long)qi * ldo + head * DH + d4 * 4) =
                make_float4(o[4 * d4] * inv, o[4 * d4 + 1] * inv, o[4 * d4 + 2] * inv, o[4 * d4 + 3] * inv);
        lse2[(long long)qi * H + head] = m + log2f(l);
    }
}

// dQ pass: lane = query.  Also writes delta[i, h] = sum_d dO[i,d] O[i,d] for the dK/dV pass.
template <int DH>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale, const float* __restrict__ out, long long ldo, const float* __restrict__ dout, long long lddo,
    const float* __restrict__ lse2, float* __restrict__ delta, float* __restrict__ dqkv, long long lddq) {
    __shared__ float Ks[ATT_TILE * DH];
    __shared__ float Vs[ATT_TILE * DH];
    const int tile = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int q0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE;
    const int E = H * DH;
    const int qi = q0 + lane;
    const bool valid = qi < kend;
    const int qrow = valid ? qi : kend - 1;
    const float scale2 = scale * LOG2E;
    float q[DH], go[DH], dq[DH];
    float dl = 0.0f;
#pragma unroll
    for (int d4 = 0; d4 < DH / 4; ++d4) {
        const float4 v = *reinterpret_cast<const float4*>(qkv + (long long)qrow * ld + head * DH + d4 * 4);
        q[4 * d4] = v.x * scale2; q[4 * d4 + 1] = v.y * scale2; q[4 * d4 + 2] = v.z * scale2; q[4 * d4 + 3] = v.w * scale2;
        const float4 g = *reinterpret_cast<const float4*>(dout + (long long)qrow * lddo + head * DH + d4 * 4);
        go[4 * d4] = g.x; go[4 * d4 + 1] = g.y; go[4 * d4 + 2] = g.z; go[4 * d4 + 3] = g.w;
        const float4 ov = *reinterpret_cast<const float4*>(out + (long long)qrow * ldo + head * DH + d4 * 4);
        dl = fmaf(g.x, ov.x, dl); dl = fmaf(g.y, ov.y, dl); dl = fmaf(g.z, ov.z, dl); dl = fmaf(g.w, ov.w, dl);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.0f;
    const float ls = lse2[(long long)qrow * H + head];
    for (int kt = kbeg; kt < kend; kt += ATT_TILE) {
        __syncthreads();
        attn_stage<DH>(Ks, qkv, ld, E + head * DH, kt, kend - 1, lane);
        attn_stage<DH>(Vs, qkv, ld, 2 * E + head * DH, kt, kend - 1, lane);
        __syncthreads();
        const int nk = min(ATT_TILE, kend - kt);
        for (int c = 0; c < nk; c += 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (c + t < nk) {
                    const float* kr = Ks + (c + t) * DH;
                    const float* vr = Vs + (c + t) * DH;
                    float s = 0.0f, dp = 0.0f;
#pragma unroll
                    for (int d4 = 0; d4 < DH / 4; ++d4) {
                        const float4 kk = *reinterpret_cast<const float4*>(kr + 4 * d4);
                        const float4 vv = *reinterpret_cast<const float4*>(vr + 4 * d4);
                        s = fmaf(q[4 * d4], kk.x, s); s = fmaf(q[4 * d4 + 1], kk.y, s);
                        s = fmaf(q[4 * d4 + 2], kk.z, s); s = fmaf(q[4 * d4 + 3], kk.w, s);
                        dp = fmaf(go[4 * d4], vv.x, dp); dp = fmaf(go[4 * d4 + 1], vv.y, dp);
                        dp = fmaf(go[4 * d4 + 2], vv.z, dp); dp = fmaf(go[4 * d4 + 3], vv.w, dp);
                    }
                    const float ds = exp2f(s - ls) * (dp - dl);
#pragma unroll
                    for (int d4 = 0; d4 < DH / 4; ++d4) {
                        const float4 kk = *reinterpret_cast<const float4*>(kr + 4 * d4);
                        dq[4 * d4] = fmaf(ds, kk.x, dq[4 * d4]); dq[4 * d4 + 1] = fmaf(ds, kk.y, dq[4 * d4 + 1]);
                        dq[4 * d4 + 2] = fmaf(ds, kk.z, dq[4 * d4 + 2]); dq[4 * d4 + 3] = fmaf(ds, kk.w, dq[4 * d4 + 3]);
                    }
                }
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4)
            *reinterpret_cast<float4*>(dqkv + (long long)qi * lddq + head * DH + d4 * 4) =
                make_float4(dq[4 * d4] * scale, dq[4 * d4 + 1] * scale, dq[4 * d4 + 2] * scale, dq[4 * d4 + 3] * scale);
        delta[(long long)qi * H + head] = dl;
    }
}

// dK / dV pass: lane = key; queries of the event stream through LDS.
template <int DH>
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale, const float* __restrict__ dout, long long lddo, const float* __restrict__ lse2,
    const float* __restrict__ delta, float* __restrict__ dqkv, long long lddq) {
    __shared__ float Qs[ATT_TILE * DH];
    __shared__ float Gs[ATT_TILE * DH];
    __shared__ float Ls[ATT_TILE];
    __shared__ float Ds[ATT_TILE];
    const int tile = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int k0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE;
    const int E = H * DH;
    const int kj = k0 + lane;
    const bool valid = kj < kend;
    const int krow = valid ? kj : kend - 1;
    const float scale2 = scale * LOG2E;
    float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
    for (int d4 = 0; d4 < DH / 4; ++d4) {
        const float4 a = *reinterpret_cast<const float4*>(qkv + (long long)krow * ld + E + head * DH + d4 * 4);
        k[4 * d4] = a.x * scale2; k[4 * d4 + 1] = a.y * scale2; k[4 * d4 + 2] = a.z * scale2; k[4 * d4 + 3] = a.w * scale2;
        const float4 b = *reinterpret_cast<const float4*>(qkv + (long long)krow * ld + 2 * E + head * DH + d4 * 4);
        v[4 * d4] = b.x; v[4 * d4 + 1] = b.y; v[4 * d4 + 2] = b.z; v[4 * d4 + 3] = b.w;
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) { dk[d] = 0.0f; dv[d] = 0.0f; }
    for (int qt = kbeg; qt < kend; qt += ATT_TILE) {
        __syncthreads();
        attn_stage<DH>(Qs, qkv, ld, head * DH, qt, kend - 1, lane);
        attn_stage<DH>(Gs, dout, lddo, head * DH, qt, kend - 1, lane);
        {
            const int row = min(qt + lane, kend - 1);
            Ls[lane] = lse2[(long long)row * H + head];
            Ds[lane] = delta[(long long)row * H + head];
        }
        __syncthreads();
        const int nq = min(ATT_TILE, kend - qt);
        for (int c = 0; c < nq; c += 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (c + t < nq) {
                    const float* qr = Qs + (c + t) * DH;
                    const float* gr = Gs + (c + t) * DH;
                    float s = 0.0f, dp = 0.0f;
#pragma unroll
                    for (int d4 = 0; d4 < DH / 4; ++d4) {
                        const float4 qq = *reinterpret_cast<const float4*>(qr + 4 * d4);
                        const float4 gg = *reinterpret_cast<const float4*>(gr + 4 * d4);
                        s = fmaf(k[4 * d4], qq.x, s); s = fmaf(k[4 * d4 + 1], qq.y, s);
                        s = fmaf(k[4 * d4 + 2], qq.z, s); s = fmaf(k[4 * d4 + 3], qq.w, s);
                        dp = fmaf(v[4 * d4], gg.x, dp); dp = fmaf(v[4 * d4 + 1], gg.y, dp);
                        dp = fmaf(v[4 * d4 + 2], gg.z, dp); dp = fmaf(v[4 * d4 + 3], gg.w, dp);
                    }
                    const float p = exp2f(s - Ls[c + t]);
                    const float ds = p * (dp - Ds[c + t]);
#pragma unroll
                    for (int d4 = 0; d4 < DH / 4; ++d4) {
                        const float4 qq = *reinterpret_cast<const float4*>(qr + 4 * d4);
                        const float4 gg = *reinterpret_cast<const float4*>(gr + 4 * d4);
                        dv[4 * d4] = fmaf(p, gg.x, dv[4 * d4]); dv[4 * d4 + 1] = fmaf(p, gg.y, dv[4 * d4 + 1]);
                        dv[4 * d4 + 2] = fmaf(p, gg.z, dv[4 * d4 + 2]); dv[4 * d4 + 3] = fmaf(p, gg.w, dv[4 * d4 + 3]);
                        dk[4 * d4] = fmaf(ds, qq.x, dk[4 * d4]); dk[4 * d4 + 1] = fmaf(ds, qq.y, dk[4 * d4 + 1]);
                        dk[4 * d4 + 2] = fmaf(ds, qq.z, dk[4 * d4 + 2]); dk[4 * d4 + 3] = fmaf(ds, qq.w, dk[4 * d4 + 3]);
                    }
                }
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4) {
            *reinterpret_cast<float4*>(dqkv + (long long)kj * lddq + E + head * DH + d4 * 4) =
                make_float4(dk[4 * d4] * scale, dk[4 * d4 + 1] * scale, dk[4 * d4 + 2] * scale, dk[4 * d4 + 3] * scale);
            *reinterpret_cast<float4*>(dqkv + (long long)kj * lddq + 2 * E + head * DH + d4 * 4) =
                make_float4(dv[4 * d4], dv[4 * d4 + 1], dv[4 * d4 + 2], dv[4 * d4 + 3]);
        }
    }
}

static bool attn_shape_ok(int DH, long long ld, long long ldo) {
    return (DH == 8 || DH == 16 || DH == 32 || DH == 64) && ld % 4 == 0 && ldo % 4 == 0;
}

hipError_t launch_attn_fwd(const float* qkv, long long ld, int H, int DH, const int* ptr, const int* tile_ptr, int B,
                           int N, float* out, long long ldo, float* lse2, hipStream_t st) {
    if (N == 0 || B == 0) return hipSuccess;
    if (!attn_shape_ok(DH, ld, ldo)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(N / ATT_TILE + B), (unsigned)H), block(ATT_TILE);
    const float scale2 = LOG2E / sqrtf((float)DH);
#define GN_ATT(D) hipLaunchKernelGGL((attn_fwd_kernel<D>), grid, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale2, out, ldo, lse2)
    switch (DH) { case 8: GN_ATT(8); break; case 16: GN_ATT(16); break; case 32: GN_ATT(32); break; default: GN_ATT(64); }
#undef GN_ATT
    return hipGetLastError();
}

hipError_t launch_attn_bwd(const float* qkv, long long ld, int H, int DH, const int* ptr, const int* tile_ptr, int B,
                           int N, const float* out, long long ldo, const float* dout, long long lddo,
                           const float* lse2, float* delta, float* dqkv, long long lddq, hipStream_t st) {
    if (N == 0 || B == 0) return hipSuccess;
    if (!attn_shape_ok(DH, ld, ldo) || lddo % 4 || lddq % 4) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(N / ATT_TILE + B), (unsigned)H), block(ATT_TILE);
    const float scale = 1.0f / sqrtf((float)DH);
#define GN_ATT(D)                                                                                                  \
    {                                                                                                              \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D>), grid, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale, out, ldo, \
                           dout, lddo, lse2, delta, dqkv, lddq);                                                   \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D>), grid, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale, dout, \
                           lddo, lse2, delta, dqkv, lddq);                                                         \
    }
    switch (DH) { case 8: GN_ATT(8) break; case 16: GN_ATT(16) break; case 32: GN_ATT(32) break; default: GN_ATT(64) }
#undef GN_ATT
    return hipGetLastError();
}

}  // namespace gn
