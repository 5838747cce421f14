// graphnet_amd/csrc/edgeconv_v2.hip — persistent, weights-stationary fused EdgeConv kernels (bf16).
//
// MI355X-specific structure (512-entry unified VGPR file, 160 KB LDS, 256 CUs):
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per CU, persistent over 64-row edge tiles;
//   * each wave OWNS 32 output columns and keeps its whole W2 slice (K x 32) in registers as MFMA
//     B fragments for the life of the kernel -> W2 is read from HBM/L2 once per CU, no LDS
//     traffic and no per-tile staging for the weight operand;
//   * the A operand (h = relu(P[i]+Q[j]), gathered per edge) is built cooperatively into a
//     double-buffered LDS tile [64 rows][K] (16-byte row pad -> conflict-free ds_read_b128);
//     the gather for tile t+1 is in flight while the MFMAs of tile t run (issue-early /
//     write-late), one barrier per tile;
//   * epilogue: +b2, relu, relu bits (ballot), segmented slot sum in registers + one
//     cross-half shuffle, coalesced 128-byte row stores.
//
// Shapes: K = H1p in {128, 352} (KSTEPS = K/16 in {8, 22}), H2 <= 256 and H2 % 32 == 0, S = 8.
// Anything else (and the rare overflow rows, and f32 mode) runs the generic kernels of
// edgeconv.hip.
#include "common.hpp"

namespace gn {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// relu(p + q) on 8 packed bf16: unpack with shift/and, f32 adds, one v_cvt_pk_bf16_f32 per pair and
// the relu as a packed signed-16-bit max with 0 (bf16 is sign-magnitude: negative <=> int16 < 0).
// Rows without a neighbour need no zeroing: GEMM rows are independent and the epilogue masks them.
__device__ __forceinline__ u32x4 relu_sum_bf16x8(u32x4 p, u32x4 q) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float plo = __builtin_bit_cast(float, p[w] << 16), phi = __builtin_bit_cast(float, p[w] & 0xffff0000u);
        const float qlo = __builtin_bit_cast(float, q[w] << 16), qhi = __builtin_bit_cast(float, q[w] & 0xffff0000u);
        bf16x2 s;
        s[0] = (__bf16)(plo + qlo);
        s[1] = (__bf16)(phi + qhi);
        const s16x2 m = __builtin_elementwise_max(__builtin_bit_cast(s16x2, s), (s16x2){0, 0});
        o[w] = __builtin_bit_cast(unsigned int, m);
    }
    return o;
}

constexpr int V2_ROWS = 64;     // edge rows per tile
constexpr int V2_THREADS = 512;

template <int KSTEPS>
__global__ __launch_bounds__(V2_THREADS, 2) void edge_fwd_v2_kernel(
    EdgeGraph g, const __bf16* __restrict__ PQ, const __bf16* __restrict__ W2p, const float* __restrict__ b2,
    int H2, float* __restrict__ out, long long ldo, unsigned int* __restrict__ maskbits, int ntiles)
{
    constexpr int S = 8;
    constexpr int K = KSTEPS * 16;                 // = H1p
    constexpr int ROWB = K * 2 + 16;               // LDS row pitch (bytes)
    constexpr int CHUNKS = K / 8;                  // 16-byte chunks per row
    constexpr int CPT = (CHUNKS + 7) / 8;          // chunks per thread (8 threads per row)
    __shared__ __attribute__((aligned(16))) unsigned char As[2][V2_ROWS * ROWB];
    __shared__ int s_jc[2][V2_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H2w = (H2 + 31) >> 5;
    const long long ldpq = 2LL * K;
    const long long main_rows = (long long)g.N * S;
    const bool wave_on = wave * 32 < H2;

    // ---- stationary W2 slice: B fragments for all k-steps (row n = wave*32 + r of W2p)
    bf16x8 w2[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) w2[s][e] = (__bf16)0.0f;
    if (wave_on) {
        const __bf16* wrow = W2p + (long long)(wave * 32 + r) * K + h * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) w2[s] = *reinterpret_cast<const bf16x8*>(wrow + s * 16);
    }
    const int col = wave * 32 + r;
    const float bias = (col < H2) ? b2[col] : 0.0f;

    // ---- gather mapping: 8 threads per row, thread handles chunks (tid&7) + 8*i.
    // Everything here is branch-free: out-of-range rows/chunks are clamped to valid addresses
    // (chunk duplicates write identical bytes) and invalid rows are zeroed by a select.
    const int grow = tid >> 3, gc0 = tid & 7;
    u32x4 preg[CPT], qreg[CPT];
    const int kslots = g.K;

    // row info is fetched in two halves so that the load is issued FIRST in an iteration and
    // its result is consumed LAST (vmcnt is in-order: an early use would drain the whole gather)
#define GN_V2_INFO_ISSUE(tile_, ic_, raw_, ok_)                                                       \
    {                                                                                                 \
        long long row = (long long)(tile_) * V2_ROWS + grow;                                          \
        const bool inr = ((tile_) < ntiles) && (row < main_rows);                                     \
        row = inr ? row : 0;                                                                          \
        const int ii = (int)(row >> 3), sl = (int)(row & 7);                                          \
        const int slc = sl < kslots ? sl : 0;                                                         \
        (raw_) = g.nbr[(long long)ii * kslots + slc];                                                 \
        (ic_) = ii;                                                                                   \
        (ok_) = inr && sl < kslots;                                                                   \
    }
#define GN_V2_INFO(tile_, ic_, jc_)                                                                   \
    {                                                                                                 \
        int raw__; bool ok__;                                                                         \
        GN_V2_INFO_ISSUE(tile_, ic_, raw__, ok__);                                                    \
        (jc_) = ok__ ? raw__ : -1;                                                                    \
    }
#define GN_V2_GATHER(ic_, jc_)                                                                        \
    {                                                                                                 \
        const int js = (jc_) < 0 ? 0 : (jc_);                                                         \
        const __bf16* pp = PQ + (long long)(ic_) * ldpq;                                              \
        const __bf16* qq = PQ + (long long)js * ldpq + K;                                             \
        _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                             \
            const int c = gc0 + 8 * i;                                                                \
            const int cc = c < CHUNKS ? c : CHUNKS - 1;                                               \
            preg[i] = *reinterpret_cast<const u32x4*>(pp + cc * 8);                                   \
            qreg[i] = *reinterpret_cast<const u32x4*>(qq + cc * 8);                                   \
        }                                                                                             \
    }
#define GN_V2_WRITE(buf, jc_)                                                                         \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                             \
            const int c = gc0 + 8 * i;                                                                \
            const int cc = c < CHUNKS ? c : CHUNKS - 1;                                               \
            *reinterpret_cast<u32x4*>(&As[buf][grow * ROWB + cc * 16]) = relu_sum_bf16x8(preg[i], qreg[i]); \
        }                                                                                             \
        if (gc0 == 0) s_jc[buf][grow] = (jc_);                                                        \
    }

    // contiguous tile range per workgroup: an event's tiles stay on one CU, so the Q rows it
    // gathers (each reused by ~K centres) are L1/L2 hits instead of 8 XCDs fetching them each
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tile_end = min(ntiles, tile + per);
    int ic_n, jc_n;                                  // row info of the NEXT tile (one tile ahead)
    // prologue: stage the first tile, fetch the row info of the second
    {
        int ic, jc;
        GN_V2_INFO(tile, ic, jc);
        GN_V2_GATHER(ic, jc);
        GN_V2_INFO(tile + 1, ic_n, jc_n);
        GN_V2_WRITE(0, jc);
    }
    __syncthreads();

    int buf = 0;
    for (; tile < tile_end; ++tile, buf ^= 1) {
        // (a) issue the row-info load of tile t+2, then the gather of tile t+1 (whose row info is
        //     already in registers); all of it stays in flight under the MFMAs
        int ic_nn, raw_nn;
        bool ok_nn;
        GN_V2_INFO_ISSUE(tile + 2, ic_nn, raw_nn, ok_nn);
        GN_V2_GATHER(ic_n, jc_n);

        // (b) MFMAs of this tile: 2 row blocks x KSTEPS
        f32x16 acc0, acc1;
        zero_acc(acc0); zero_acc(acc1);
        if (wave_on) {
            const unsigned char* a0 = &As[buf][r * ROWB + h * 16];
            const unsigned char* a1 = a0 + 32 * ROWB;
            // fragment reads run two k-steps ahead of the MFMAs that consume them
            bf16x8 f0 = *reinterpret_cast<const bf16x8*>(a0);
            bf16x8 f1 = *reinterpret_cast<const bf16x8*>(a1);
            bf16x8 n0 = *reinterpret_cast<const bf16x8*>(a0 + 32);
            bf16x8 n1 = *reinterpret_cast<const bf16x8*>(a1 + 32);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                bf16x8 m0 = n0, m1 = n1;
                if (s + 2 < KSTEPS) {
                    m0 = *reinterpret_cast<const bf16x8*>(a0 + (s + 2) * 32);
                    m1 = *reinterpret_cast<const bf16x8*>(a1 + (s + 2) * 32);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, w2[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, w2[s], acc1, 0, 0, 0);
                f0 = n0; f1 = n1; n0 = m0; n1 = m1;
            }
        }

        // (c) epilogue
        if (wave_on) {
            const long long row0 = (long long)tile * V2_ROWS;
            const int rL = (r & 3) + 4 * (r >> 3), hL = (r >> 2) & 1;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f32x16& acc = rb ? acc1 : acc0;
                float v[16];
                unsigned int word = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int rl = rb * 32 + acc_row(q, h);
                    v[q] = (s_jc[buf][rl] >= 0) ? fmaxf(acc[q] + bias, 0.0f) : 0.0f;
                    const unsigned long long bal = __ballot(v[q] > 0.0f);
                    if (q == rL) word = hL ? (unsigned int)(bal >> 32) : (unsigned int)bal;
                }
                if (lane < 32) {
                    const long long rowg = row0 + rb * 32 + r;
                    if (rowg < main_rows) maskbits[rowg * H2w + wave] = word;
                }
                const long long c0 = (row0 + rb * 32) / S;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float sum = v[4 * c] + v[4 * c + 1] + v[4 * c + 2] + v[4 * c + 3];
                    sum += __shfl_xor(sum, 32);
                    if ((c >> 1) == h && c0 + c < g.N) out[(c0 + c) * ldo + col] = sum;
                }
            }
        }

        // (d) finish staging the next tile into the other buffer
        GN_V2_WRITE(buf ^ 1, jc_n);
        ic_n = ic_nn; jc_n = ok_nn ? raw_nn : -1;
        __syncthreads();
    }
#undef GN_V2_INFO
#undef GN_V2_INFO_ISSUE
#undef GN_V2_GATHER
#undef GN_V2_WRITE
}

// returns hipErrorNotSupported when the shape is outside the v2 envelope (caller falls back)
hipError_t launch_edge_fwd_v2(const EdgeGraph& g, const void* PQ, int H1p, const void* W2p, const float* b2, int H2,
                              float* out, long long ldo, unsigned int* maskbits, int num_cus, hipStream_t st) {
    if (g.K > 8 || H2 > 256 || (H2 & 31) || (H1p != 128 && H1p != 352)) return hipErrorNotSupported;
    if (g.N == 0) return hipSuccess;
    const long long rows = (long long)g.N * 8;
    const int ntiles = (int)((rows + V2_ROWS - 1) / V2_ROWS);
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    if (H1p == 128)
        hipLaunchKernelGGL((edge_fwd_v2_kernel<8>), dim3(grid), dim3(V2_THREADS), 0, st, g, (const __bf16*)PQ,
                           (const __bf16*)W2p, b2, H2, out, ldo, maskbits, ntiles);
    else
        hipLaunchKernelGGL((edge_fwd_v2_kernel<22>), dim3(grid), dim3(V2_THREADS), 0, st, g, (const __bf16*)PQ,
                           (const __bf16*)W2p, b2, H2, out, ldo, maskbits, ntiles);
    return hipGetLastError();
}

}  // namespace gn
