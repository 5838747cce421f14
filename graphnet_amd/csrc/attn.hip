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


// ================================================================================ bf16 MFMA kernels (mode 1)
// One wave = 32 queries (forward, dQ pass) or 32 keys (dK/dV pass), head width DH = 32*NB.  Every product is
// formed TRANSPOSED so that the quantity a softmax row needs is a per-lane scalar:
//   S^T[key][query] = K Q^T   -> accumulator column = lane&31 = query; a lane's 16 registers (+ the other half
//                                wave's 16) are the keys of ITS query: max / sum / rescale are in-lane,
//   O^T[d][query]  += V^T P^T -> P^T leaves the S^T accumulators already in B-operand layout (registers 8t..8t+7
//                                of lane-half h are keys 16t+4h+{0..3, 8..11}); V^T fragments with the same key
//                                order come out of a row-major LDS tile through ds_read_b64_tr_b16.
// No fragment ever crosses lanes except one xor-32 exchange of the running maximum per block.
typedef short s16x4_a __attribute__((ext_vector_type(4)));
typedef short s16x8_a __attribute__((ext_vector_type(8)));
typedef float f32x8_a __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_a lds_s16x4_a;
__host__ __device__ constexpr int attn_tr_pitch(int row_bytes) { return row_bytes + ((64 - row_bytes % 256) + 256) % 256; }

__device__ __forceinline__ bf16x8 attn_load8(const float* __restrict__ p, float mul) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    const f32x8_a v = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, b.x * mul, b.y * mul, b.z * mul, b.w * mul};
    return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ bf16x8 attn_pack8(const f32x16& v, int t) {
    const f32x8_a w = {v[8 * t], v[8 * t + 1], v[8 * t + 2], v[8 * t + 3], v[8 * t + 4], v[8 * t + 5], v[8 * t + 6], v[8 * t + 7]};
    return __builtin_convertvector(w, bf16x8);
}
// A-operand fragment (M = 32 tile columns starting at col0, k = the 16 tile rows {16t + 4h + 0..3, 16t + 4h + 8..11})
__device__ __forceinline__ bf16x8 attn_tr_frag(const unsigned char* tile, int pitch, int t, int col0, int lane) {
    const int g4 = lane >> 4, li = lane & 15;
    const unsigned char* p = tile + (16 * t + 4 * (g4 >> 1) + (li >> 2)) * pitch + (col0 + 16 * (g4 & 1) + 4 * (li & 3)) * 2;
    const s16x4_a a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(p));
    const s16x4_a a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(p + 8 * pitch));
    const s16x8_a av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, av);
}

template <int NB>
__global__ __launch_bounds__(64) void attn_fwd_mfma_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale2, float* __restrict__ out, long long ldo, float* __restrict__ lse2) {
    constexpr int DH = 32 * NB, KS = DH / 16, VP = attn_tr_pitch(DH * 2);
    __shared__ __attribute__((aligned(16))) unsigned char Vs[32 * VP];
    const int tile = blockIdx.x >> 1, half = blockIdx.x & 1, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int q0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE + 32 * half;
    if (q0 >= kend) return;
    const int E = H * DH, c = lane & 31, h = lane >> 5;
    const int qi = q0 + c;
    const int qrow = min(qi, kend - 1);
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = attn_load8(qkv + (long long)qrow * ld + head * DH + 16 * s + 8 * h, scale2);
    f32x16 o[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) zero_acc(o[nb]);
    float m = -INFINITY, l = 0.0f;
    for (int kt = kbeg; kt < kend; kt += 32) {
        const int krow = min(kt + c, kend - 1);
        const float* kp = qkv + (long long)krow * ld + E + head * DH;
        bf16x8 kf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = attn_load8(kp + 16 * s + 8 * h, 1.0f);
        bf16x8 vv[DH / 16];
#pragma unroll
        for (int j = 0; j < DH / 16; ++j) vv[j] = attn_load8(kp + E + h * (DH / 2) + 8 * j, 1.0f);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DH / 16; ++j)
            *reinterpret_cast<bf16x8*>(&Vs[c * VP + (h * (DH / 2) + 8 * j) * 2]) = vv[j];
        __syncthreads();
        f32x16 s;
        zero_acc(s);
#pragma unroll
        for (int t = 0; t < KS; ++t) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qf[t], s, 0, 0, 0);
        if (kt + 32 > kend) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt + acc_row(r, h) >= kend) s[r] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float corr = exp2f(m - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mn); ps += s[r]; }
        l = l * corr + ps;
        m = mn;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nb][r] *= corr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = attn_pack8(s, t);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag(Vs, VP, t, nb * 32, lane), pf, o[nb], 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32);
    if (qi < kend) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(out + (long long)qi * ldo + head * DH + nb * 32 + 8 * g + 4 * h) =
                    make_float4(o[nb][4 * g] * inv, o[nb][4 * g + 1] * inv, o[nb][4 * g + 2] * inv, o[nb][4 * g + 3] * inv);
        if (h == 0) lse2[(long long)qi * H + head] = m + log2f(l);
    }
}

template <int NB>
__global__ __launch_bounds__(64) void attn_bwd_dq_mfma_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale, const float* __restrict__ out, long long ldo, const float* __restrict__ dout, long long lddo,
    const float* __restrict__ lse2, float* __restrict__ delta, float* __restrict__ dqkv, long long lddq) {
    constexpr int DH = 32 * NB, KS = DH / 16, VP = attn_tr_pitch(DH * 2);
    __shared__ __attribute__((aligned(16))) unsigned char Ks[32 * VP];
    const int tile = blockIdx.x >> 1, half = blockIdx.x & 1, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int q0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE + 32 * half;
    if (q0 >= kend) return;
    const int E = H * DH, c = lane & 31, h = lane >> 5;
    const int qi = q0 + c;
    const int qrow = min(qi, kend - 1);
    const float scale2 = scale * LOG2E;
    bf16x8 qf[KS], gf[KS];
    float dl = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = attn_load8(qkv + (long long)qrow * ld + head * DH + 16 * s + 8 * h, scale2);
        const float* gp = dout + (long long)qrow * lddo + head * DH + 16 * s + 8 * h;
        const float* op = out + (long long)qrow * ldo + head * DH + 16 * s + 8 * h;
        gf[s] = attn_load8(gp, 1.0f);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl = fmaf(gp[j], op[j], dl);
    }
    dl += __shfl_xor(dl, 32);
    const float ls = lse2[(long long)qrow * H + head];
    f32x16 dq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) zero_acc(dq[nb]);
    for (int kt = kbeg; kt < kend; kt += 32) {
        const int krow = min(kt + c, kend - 1);
        const float* kp = qkv + (long long)krow * ld + E + head * DH;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { kf[s] = attn_load8(kp + 16 * s + 8 * h, 1.0f); vf[s] = attn_load8(kp + E + 16 * s + 8 * h, 1.0f); }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) *reinterpret_cast<bf16x8*>(&Ks[c * VP + (16 * s + 8 * h) * 2]) = kf[s];
        __syncthreads();
        f32x16 s, dp;
        zero_acc(s); zero_acc(dp);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qf[t], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t], gf[t], dp, 0, 0, 0);
        }
        const bool tail = kt + 32 > kend;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = exp2f(s[r] - ls);
            if (tail && kt + acc_row(r, h) >= kend) p = 0.0f;
            s[r] = p * (dp[r] - dl);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 df = attn_pack8(s, t);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                dq[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag(Ks, VP, t, nb * 32, lane), df, dq[nb], 0, 0, 0);
        }
    }
    if (qi < kend) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dqkv + (long long)qi * lddq + head * DH + nb * 32 + 8 * g + 4 * h) =
                    make_float4(dq[nb][4 * g] * scale, dq[nb][4 * g + 1] * scale, dq[nb][4 * g + 2] * scale, dq[nb][4 * g + 3] * scale);
        if (h == 0) delta[(long long)qi * H + head] = dl;
    }
}

template <int NB>
__global__ __launch_bounds__(64) void attn_bwd_dkv_mfma_kernel(
    const float* __restrict__ qkv, long long ld, int H, const int* __restrict__ ptr, const int* __restrict__ tile_ptr,
    int B, float scale, const float* __restrict__ dout, long long lddo, const float* __restrict__ lse2,
    const float* __restrict__ delta, float* __restrict__ dqkv, long long lddq) {
    constexpr int DH = 32 * NB, KS = DH / 16, VP = attn_tr_pitch(DH * 2);
    __shared__ __attribute__((aligned(16))) unsigned char Qs[32 * VP];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[32 * VP];
    __shared__ __attribute__((aligned(16))) float Ls[32];
    __shared__ __attribute__((aligned(16))) float Ds[32];
    const int tile = blockIdx.x >> 1, half = blockIdx.x & 1, head = blockIdx.y, lane = threadIdx.x;
    if (tile >= tile_ptr[B]) return;
    const int e = attn_event_of_tile(tile_ptr, B, tile);
    const int kbeg = ptr[e], kend = ptr[e + 1];
    const int k0 = kbeg + (tile - tile_ptr[e]) * ATT_TILE + 32 * half;
    if (k0 >= kend) return;
    const int E = H * DH, c = lane & 31, h = lane >> 5;
    const int kj = k0 + c;
    const int krow = min(kj, kend - 1);
    const float scale2 = scale * LOG2E;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = attn_load8(qkv + (long long)krow * ld + E + head * DH + 16 * s + 8 * h, scale2);
        vf[s] = attn_load8(qkv + (long long)krow * ld + 2 * E + head * DH + 16 * s + 8 * h, 1.0f);
    }
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { zero_acc(dk[nb]); zero_acc(dv[nb]); }
    for (int qt = kbeg; qt < kend; qt += 32) {
        const int qrow = min(qt + c, kend - 1);
        bf16x8 qa[KS], ga[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qa[s] = attn_load8(qkv + (long long)qrow * ld + head * DH + 16 * s + 8 * h, 1.0f);
            ga[s] = attn_load8(dout + (long long)qrow * lddo + head * DH + 16 * s + 8 * h, 1.0f);
        }
        const float lv = lse2[(long long)qrow * H + head], dvv = delta[(long long)qrow * H + head];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8*>(&Qs[c * VP + (16 * s + 8 * h) * 2]) = qa[s];
            *reinterpret_cast<bf16x8*>(&Gs[c * VP + (16 * s + 8 * h) * 2]) = ga[s];
        }
        if (h == 0) { Ls[c] = lv; Ds[c] = dvv; }
        __syncthreads();
        f32x16 s, dp;
        zero_acc(s); zero_acc(dp);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[t], kf[t], s, 0, 0, 0);       // rows = queries, col = key
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[t], vf[t], dp, 0, 0, 0);
        }
        const bool tail = qt + 32 > kend;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 L4 = *reinterpret_cast<const float4*>(&Ls[8 * g + 4 * h]);
            const float4 D4 = *reinterpret_cast<const float4*>(&Ds[8 * g + 4 * h]);
            const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float p = exp2f(s[r] - Lr[j]);
                if (tail && qt + 8 * g + 4 * h + j >= kend) p = 0.0f;
                s[r] = p;
                dp[r] = p * (dp[r] - Dr[j]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = attn_pack8(s, t), df = attn_pack8(dp, t);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag(Gs, VP, t, nb * 32, lane), pf, dv[nb], 0, 0, 0);
                dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag(Qs, VP, t, nb * 32, lane), df, dk[nb], 0, 0, 0);
            }
        }
    }
    if (kj < kend) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long o = (long long)kj * lddq + head * DH + nb * 32 + 8 * g + 4 * h;
                *reinterpret_cast<float4*>(dqkv + o + E) =
                    make_float4(dk[nb][4 * g] * scale, dk[nb][4 * g + 1] * scale, dk[nb][4 * g + 2] * scale, dk[nb][4 * g + 3] * scale);
                *reinterpret_cast<float4*>(dqkv + o + 2 * E) =
                    make_float4(dv[nb][4 * g], dv[nb][4 * g + 1], dv[nb][4 * g + 2], dv[nb][4 * g + 3]);
            }
    }
}

static bool attn_shape_ok(int DH, long long ld, long long ldo) {
    return (DH == 8 || DH == 16 || DH == 32 || DH == 64) && ld % 4 == 0 && ldo % 4 == 0;
}

hipError_t launch_attn_fwd(int mode, const float* qkv, long long ld, int H, int DH, const int* ptr, const int* tile_ptr,
                           int B, int N, float* out, long long ldo, float* lse2, hipStream_t st) {
    if (N == 0 || B == 0) return hipSuccess;
    if (!attn_shape_ok(DH, ld, ldo)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(N / ATT_TILE + B), (unsigned)H), block(ATT_TILE);
    const float scale2 = LOG2E / sqrtf((float)DH);
    if (mode == 1 && (DH == 32 || DH == 64)) {            // bf16 operands on the matrix core
        const dim3 grid2(2 * grid.x, (unsigned)H);
        if (DH == 32) hipLaunchKernelGGL((attn_fwd_mfma_kernel<1>), grid2, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale2, out, ldo, lse2);
        else hipLaunchKernelGGL((attn_fwd_mfma_kernel<2>), grid2, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale2, out, ldo, lse2);
        return hipGetLastError();
    }
#define GN_ATT(D) hipLaunchKernelGGL((attn_fwd_kernel<D>), grid, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale2, out, ldo, lse2)
    switch (DH) { case 8: GN_ATT(8); break; case 16: GN_ATT(16); break; case 32: GN_ATT(32); break; default: GN_ATT(64); }
#undef GN_ATT
    return hipGetLastError();
}

hipError_t launch_attn_bwd(int mode, const float* qkv, long long ld, int H, int DH, const int* ptr, const int* tile_ptr,
                           int B, int N, const float* out, long long ldo, const float* dout, long long lddo,
                           const float* lse2, float* delta, float* dqkv, long long lddq, hipStream_t st) {
    if (N == 0 || B == 0) return hipSuccess;
    if (!attn_shape_ok(DH, ld, ldo) || lddo % 4 || lddq % 4) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(N / ATT_TILE + B), (unsigned)H), block(ATT_TILE);
    const float scale = 1.0f / sqrtf((float)DH);
    if (mode == 1 && (DH == 32 || DH == 64)) {
        const dim3 grid2(2 * grid.x, (unsigned)H);
#define GN_ATTM(NB_)                                                                                                \
    {                                                                                                               \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<NB_>), grid2, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale, out, \
                           ldo, dout, lddo, lse2, delta, dqkv, lddq);                                               \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<NB_>), grid2, block, 0, st, qkv, ld, H, ptr, tile_ptr, B, scale, dout, \
                           lddo, lse2, delta, dqkv, lddq);                                                          \
    }
        if (DH == 32) GN_ATTM(1) else GN_ATTM(2)
#undef GN_ATTM
        return hipGetLastError();
    }
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
