// tiled_kernels.hip -- LDS-tiled sweeps for the throughput regime (large batches, rows of 100+ nonzeros).
//
// Why: the generic sweeps gather one 64-byte source row per nonzero straight from L2.  On MI355X that is
// bound by the L1->L2 request path (profiles/r01_v2_spmm_pmc.txt: TA 93 % busy, one request per nonzero,
// HBM traffic == algorithmic bytes), ~0.6 ms per 64 M nonzeros whatever the lane mapping.  Here the source
// rows, the nonzeros and the row accumulators all live in LDS:
//
//   storage   the nonzeros of an orientation are re-blocked once (LPBatch.enable_tiled): row tiles of
//             R = 512 rows  x  column blocks of CB = 1024 source nodes.  Inside a (tile, block) the rows are
//             SORTED BY THEIR NUMBER OF ENTRIES in that block (descending) and their entries {col_local, value}
//             stored contiguously in that order; ptr2 gives the start of every sorted position, perm the row it
//             belongs to.  Same 8 bytes per nonzero as CSR (+ 6 bytes per (row, block)), streamed once, coalesced.
//   workgroup 1024 threads = one row tile = one CU (150 KB of LDS).  For each column block the tile touches:
//               LDS  <-  64 KB of H (CB x 64 B)  +  the block's entry segment (<= 48 KB per window)  +  offsets/perm,
//             all prefetched into REGISTERS while the previous block is computed (issue early, write late).
//             Then every QUAD of lanes takes one sorted position: the 16 quads of a wave therefore walk rows of
//             (nearly) EQUAL length in lockstep -- the first version (quad = fixed rows, Poisson lengths) ran the
//             SIMD and the LDS at 59 % efficiency.  A quad loads the row's accumulator from LDS, walks the row
//             (entry: ds_read_b64, source row piece: ds_read_b128, 4 FMAs per lane), stores it back.  Waves take
//             bundles in snake order (w and 31 - w) so that every wave gets the same amount of work.
//   traffic   per tile: the entry stream once from HBM + (blocks touched) x 64 KB of H from L2 as wide coalesced
//             loads: one L2 request per 128 B instead of one per nonzero; the output tile is written once.
#include <cstdlib>

#include "device_utils.h"
#include "internal.h"

namespace mllp {

constexpr int T_THREADS = 1024;            // 16 waves
constexpr int T_WAVES = T_THREADS / 64;
constexpr int T_R = 512;                   // rows per tile: 32 bundles of 16 sorted positions
constexpr int T_BUNDLES = T_R / 16;
constexpr int T_CB = 1024;                 // source nodes per column block (64 KB of fp32 x 16)
constexpr int T_EPT = 6;                   // entries prefetched per thread
constexpr int T_ECAP = T_EPT * T_THREADS;  // 6144 entries (48 KB) of a (tile, block) segment per window
constexpr int T_XPT = (T_CB * 4) / T_THREADS;   // float4 of H per thread
constexpr int T_MAXB = 255;                // column blocks one row tile may touch (checked at attach)

// sum over the 4 lanes of a quad (every lane gets the total)
__device__ __forceinline__ float quad_sum4(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
// exp(x), x <= ~0, 1-2 ulp (see sweep_kernels.hip::exp_acc)
__device__ __forceinline__ float exp_acc_t(float x) {
    const float L2E_HI = 1.44269502163e+00f, L2E_LO = 1.92596299112e-08f, LN2 = 0.693147180560f;
    x = fmaxf(x, -150.0f);
    const float t = x * L2E_HI;
    float r = fmaf(x, L2E_HI, -t);
    r = fmaf(x, L2E_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * LN2, e);
}

struct TiledDev {
    const int* __restrict__ tile_blk;   // [n_tiles + 1] first (tile,block) index of each tile
    const int* __restrict__ blk_id;     // [n_tb] global column-block id
    const int* __restrict__ ptr2;       // [n_tb * T_R + 1] entry offsets per (tile-block, sorted position)
    const int* __restrict__ perm;       // [n_tb * T_R] row (inside the tile) of each sorted position
    const int2* __restrict__ ent;       // [nnz] {col_local, value bits}
    int n_tiles, n_dst, n_src;
};

// ABL: ablation mask for timing-only builds (results wrong when != 0): 1 = no H loads, 2 = no entry loads,
// 4 = no walk, 8 = no LDS writes of H / entries
template <int ABL>
__global__ __launch_bounds__(T_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))   // 128 VGPRs: no spills
void spmm_tiled_kernel(TiledDev t, const float* __restrict__ X,
                                                               float* __restrict__ Y) {
    __shared__ float4 Xs[T_CB * 4];     // 64 KB  staged column block of H
    __shared__ int2 Es[T_ECAP];         // 48 KB  entry segment (window)
    __shared__ float4 Ya[T_R * 4];      // 32 KB  accumulators of the row tile
    __shared__ int Ps[T_R + 16];        // offsets of the sorted positions inside the segment
    __shared__ int Pm[T_R];             // row of each sorted position
    __shared__ int Sg[T_MAXB + 1];      // segment start of every block of this tile (no dependent global loads later)
    __shared__ int Bk[T_MAXB];          // column-block id of every block of this tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = lane >> 2, part = lane & 3;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];

    for (int i = tid; i < T_R * 4; i += T_THREADS) Ya[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid <= tb1 - tb0) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * T_R];
    if (tid < tb1 - tb0) Bk[tid] = t.blk_id[tb0 + tid];
    __syncthreads();

    // ---- register prefetch of a (tile, block): H piece, entries of the first window, offsets, perm.
    //      TWO register sets (A, B): the loads of block b+2 are issued while block b is computed.  Loads are
    //      never predicated (addresses are clamped instead) so that the compiler can use counted vmcnt waits,
    //      and half of the waves burst their loads before the walk, the other half after it.
    //      (Macros, not lambdas over a struct reference: that form was demoted to scratch memory.)
#define T_LDX(S, K)                                                                                         \
    px##S##K = (ABL & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : src_[min(tid + K * T_THREADS, c4_ - 1)];
#define T_LDE(S, K)                                                                                         \
    pe##S##K = (ABL & 2) ? make_int2(0, 0) : t.ent[seg0##S + min(tid + K * T_THREADS, max(len##S - 1, 0))];
#define T_PREFETCH(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbx_ = (TB);                                                                              \
        const int c0_ = Bk[tbx_ - tb0] * T_CB;                                                              \
        const int c4_ = min(T_CB, t.n_src - c0_) * 4;                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(X + (size_t)c0_ * 16);                        \
        T_LDX(S, 0) T_LDX(S, 1) T_LDX(S, 2) T_LDX(S, 3)                                                     \
        seg0##S = Sg[tbx_ - tb0];                                                                           \
        len##S = Sg[tbx_ - tb0 + 1] - seg0##S;                                                              \
        pp##S = t.ptr2[(size_t)tbx_ * T_R + min(tid, T_R - 1)] - seg0##S;                                   \
        pm##S = t.perm[(size_t)tbx_ * T_R + min(tid, T_R - 1)];                                             \
        T_LDE(S, 0) T_LDE(S, 1) T_LDE(S, 2) T_LDE(S, 3) T_LDE(S, 4) T_LDE(S, 5)                             \
    }

    // one block: registers -> LDS, refill the register set with block tb + 2, walk
#define T_DO_BLOCK(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbc_ = (TB);                                                                              \
        __syncthreads(); /* every wave is done with the previous block's LDS image */                      \
        Xs[tid] = px##S##0; Xs[tid + T_THREADS] = px##S##1;                                                 \
        Xs[tid + 2 * T_THREADS] = px##S##2; Xs[tid + 3 * T_THREADS] = px##S##3;                             \
        Es[tid] = pe##S##0; Es[tid + T_THREADS] = pe##S##1; Es[tid + 2 * T_THREADS] = pe##S##2;             \
        Es[tid + 3 * T_THREADS] = pe##S##3; Es[tid + 4 * T_THREADS] = pe##S##4;                             \
        Es[tid + 5 * T_THREADS] = pe##S##5;                                                                 \
        if (tid < T_R) {                                                                                    \
            Ps[tid] = pp##S;                                                                                \
            Pm[tid] = pm##S;                                                                                \
        }                                                                                                   \
        if (tid == 0) Ps[T_R] = len##S;                                                                     \
        const int cur_seg0 = seg0##S, cur_len = len##S;                                                     \
        __syncthreads();                                                                                    \
        const int tb_next = min(tbc_ + 2, tb1 - 1); /* the tail refetches the last block */                \
        if (early) T_PREFETCH(S, tb_next)                                                                   \
        walk(cur_seg0, cur_len);                                                                            \
        if (!early) T_PREFETCH(S, tb_next)                                                                  \
    }

    auto walk = [&](int cur_seg0, int cur_len) {
        for (int w0 = 0; w0 < ((ABL & 4) ? 0 : cur_len); w0 += T_ECAP) {
            if (w0 > 0) {                      // rare: segment longer than one window -> restage synchronously
                __syncthreads();
                for (int i = tid; i < min(T_ECAP, cur_len - w0); i += T_THREADS) Es[i] = t.ent[cur_seg0 + w0 + i];
                __syncthreads();
            }
            const int w1 = w0 + T_ECAP;
#pragma unroll
            for (int pass = 0; pass < T_BUNDLES / T_WAVES; ++pass) {
                // snake order over the length-sorted bundles: equal work per wave
                const int bundle = (pass & 1) ? (pass + 1) * T_WAVES - 1 - wave : pass * T_WAVES + wave;
                const int k = bundle * 16 + quad;
                const int s = max(Ps[k], w0), e = min(Ps[k + 1], w1);
                if (s < e) {
                    const int rl = Pm[k];
                    float4 a = Ya[rl * 4 + part];
                    int p = s - w0;
                    const int pe_ = e - w0;
                    for (; p + 1 < pe_; p += 2) {
                        const int2 e0 = Es[p], e1 = Es[p + 1];
                        const float4 x0 = Xs[e0.x * 4 + part], x1 = Xs[e1.x * 4 + part];
                        fma4(__int_as_float(e0.y), x0, a);
                        fma4(__int_as_float(e1.y), x1, a);
                    }
                    if (p < pe_) {
                        const int2 e0 = Es[p];
                        fma4(__int_as_float(e0.y), Xs[e0.x * 4 + part], a);
                    }
                    Ya[rl * 4 + part] = a;
                }
            }
        }
    };
    static_assert(T_XPT == 4 && T_EPT == 6, "the prefetch macros are written for 4 + 6 registers per set");
    float4 pxA0, pxA1, pxA2, pxA3, pxB0, pxB1, pxB2, pxB3;
    int2 peA0, peA1, peA2, peA3, peA4, peA5, peB0, peB1, peB2, peB3, peB4, peB5;
    int ppA = 0, pmA = 0, seg0A = 0, lenA = 0, ppB = 0, pmB = 0, seg0B = 0, lenB = 0;
    const bool early = wave < T_WAVES / 2;   // half of the waves burst their loads before the walk, half after
    if (tb0 < tb1) {
        T_PREFETCH(A, tb0)
        T_PREFETCH(B, min(tb0 + 1, tb1 - 1))
    }
    for (int tb = tb0; tb < tb1; tb += 2) {
        T_DO_BLOCK(A, tb)
        if (tb + 1 < tb1) T_DO_BLOCK(B, tb + 1)
    }
#undef T_LDX
#undef T_LDE
#undef T_PREFETCH
#undef T_DO_BLOCK
    __syncthreads();
    const int row0 = tile * T_R;
    const int n4 = min(T_R, t.n_dst - row0) * 4;
    float4* dst = reinterpret_cast<float4*>(Y + (size_t)row0 * 16);
    for (int i = tid; i < n4; i += T_THREADS) dst[i] = Ya[i];
}

// =================================================================================================
// Attention forward (16 source channels) in the tiled form: the reference's TransformerConv hot loop.
// Geometry: 512-row tiles x 512-column blocks (32 KB of H): besides the block image the tile keeps per row
// the folded query q' (64 B) and the online-softmax state {rowmax, L, u, t | Z[16]} (80 B) in LDS.
// A quad takes one length-sorted row of the block, loads that state, walks the row
// (logit = <q', X_j> + a t by 4 FMAs + a 2-step DPP quad sum; rescale only when some row's max moves),
// stores it back.  After the last block: o = relu(Wv Z + S bv + u we + Ws x + bs) with 16 lanes per row.
// =================================================================================================
constexpr int F_R = 512;
constexpr int F_BUNDLES = F_R / 16;
constexpr int F_CB = 512;
constexpr int F_ECAP = 3 * T_THREADS;      // 3072 entries (24 KB) per window

struct FwdTiledArgs {
    const float* __restrict__ X;        // [n_src, 16]
    const float* __restrict__ xd;       // [n_dst, 16]
    const float* __restrict__ qp;       // [n_dst, 16]
    const float* __restrict__ tq;       // [n_dst]
    ConvParams p;
    float* __restrict__ h;              // [n_dst, 16]
    float* __restrict__ Z;              // [n_dst, 16]
    float* __restrict__ aux;            // [n_dst, 4]
};

__global__ __launch_bounds__(T_THREADS) void fwd16_tiled_kernel(TiledDev t, FwdTiledArgs a) {
    __shared__ float4 Xs[F_CB * 4];     // 32 KB  staged column block of H
    __shared__ int2 Es[F_ECAP];         // 24 KB  entry segment (window)
    __shared__ float4 St[F_R * 5];      // 40 KB  per row: {rowmax, L, u, t}, Z[16]
    __shared__ float4 Qp[F_R * 4];      // 32 KB  per row: q'
    __shared__ int Ps[F_R + 16];
    __shared__ int Pm[F_R];
    __shared__ int Sg[T_MAXB + 1];
    __shared__ int Bk[T_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = lane >> 2, part = lane & 3;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];
    const int row0 = tile * F_R;
    const int n_rows = min(F_R, t.n_dst - row0);

    // per-row inputs and the initial state
    for (int i = tid; i < F_R * 4; i += T_THREADS)
        Qp[i] = (i >> 2) < n_rows ? reinterpret_cast<const float4*>(a.qp + (size_t)row0 * 16)[i]
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = tid; r < F_R; r += T_THREADS) {
        St[r * 5] = make_float4(NEG_BIG, 0.f, 0.f, r < n_rows ? a.tq[row0 + r] : 0.f);
        St[r * 5 + 1] = St[r * 5 + 2] = St[r * 5 + 3] = St[r * 5 + 4] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid <= tb1 - tb0) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * F_R];
    if (tid < tb1 - tb0) Bk[tid] = t.blk_id[tb0 + tid];
    __syncthreads();

#define F_PREFETCH(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbx_ = (TB);                                                                              \
        const int c0_ = Bk[tbx_ - tb0] * F_CB;                                                              \
        const int c4_ = min(F_CB, t.n_src - c0_) * 4;                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(a.X + (size_t)c0_ * 16);                      \
        px##S##0 = src_[min(tid, c4_ - 1)];                                                                 \
        px##S##1 = src_[min(tid + T_THREADS, c4_ - 1)];                                                     \
        seg0##S = Sg[tbx_ - tb0];                                                                           \
        len##S = Sg[tbx_ - tb0 + 1] - seg0##S;                                                              \
        pp##S = t.ptr2[(size_t)tbx_ * F_R + min(tid, F_R - 1)] - seg0##S;                                   \
        pm##S = t.perm[(size_t)tbx_ * F_R + min(tid, F_R - 1)];                                             \
        pe##S##0 = t.ent[seg0##S + min(tid, max(len##S - 1, 0))];                                           \
        pe##S##1 = t.ent[seg0##S + min(tid + T_THREADS, max(len##S - 1, 0))];                               \
        pe##S##2 = t.ent[seg0##S + min(tid + 2 * T_THREADS, max(len##S - 1, 0))];                           \
    }
#define F_DO_BLOCK(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbc_ = (TB);                                                                              \
        __syncthreads();                                                                                    \
        Xs[tid] = px##S##0; Xs[tid + T_THREADS] = px##S##1;                                                 \
        Es[tid] = pe##S##0; Es[tid + T_THREADS] = pe##S##1; Es[tid + 2 * T_THREADS] = pe##S##2;             \
        if (tid < F_R) {                                                                                    \
            Ps[tid] = pp##S;                                                                                \
            Pm[tid] = pm##S;                                                                                \
        }                                                                                                   \
        if (tid == 0) Ps[F_R] = len##S;                                                                     \
        const int cur_seg0 = seg0##S, cur_len = len##S;                                                     \
        __syncthreads();                                                                                    \
        const int tb_next = min(tbc_ + 2, tb1 - 1);                                                         \
        if (early) F_PREFETCH(S, tb_next)                                                                   \
        walk(cur_seg0, cur_len);                                                                            \
        if (!early) F_PREFETCH(S, tb_next)                                                                  \
    }

    auto walk = [&](int cur_seg0, int cur_len) {
        for (int w0 = 0; w0 < cur_len; w0 += F_ECAP) {
            if (w0 > 0) {   // rare: segment longer than one window
                __syncthreads();
                for (int i = tid; i < min(F_ECAP, cur_len - w0); i += T_THREADS) Es[i] = t.ent[cur_seg0 + w0 + i];
                __syncthreads();
            }
            const int w1 = w0 + F_ECAP;
#pragma unroll
            for (int pass = 0; pass < F_BUNDLES / T_WAVES; ++pass) {
                const int bundle = (pass & 1) ? (pass + 1) * T_WAVES - 1 - wave : pass * T_WAVES + wave;
                const int k = bundle * 16 + quad;
                const int s = max(Ps[k], w0), e = min(Ps[k + 1], w1);
                if (s < e) {
                    const int rl = Pm[k];
                    const float4 hd = St[rl * 5];
                    const float4 q = Qp[rl * 4 + part];
                    float4 z = St[rl * 5 + 1 + part];
                    float m = hd.x, L = hd.y, u = hd.z;
                    const float tq = hd.w;
                    int p = s - w0;
                    const int pe_ = e - w0;
                    for (; p + 1 < pe_; p += 2) {      // two entries per pass: their LDS reads and dot products are independent
                        const int2 e0 = Es[p], e1 = Es[p + 1];
                        const float4 x0 = Xs[e0.x * 4 + part], x1 = Xs[e1.x * 4 + part];
                        const float a0 = __int_as_float(e0.y), a1 = __int_as_float(e1.y);
                        const float d0 = fmaf(a0, tq, quad_sum4(dot4(q, x0)));
                        const float d1 = fmaf(a1, tq, quad_sum4(dot4(q, x1)));
                        const float dm = fmaxf(d0, d1);
                        if (__any(dm > m)) {           // some row of this wave moves its max: rescale those rows
                            const float mn = fmaxf(m, dm);
                            const float sc = exp_acc_t(m - mn);
                            L *= sc; u *= sc;
                            z.x *= sc; z.y *= sc; z.z *= sc; z.w *= sc;
                            m = mn;
                        }
                        const float p0 = exp_acc_t(d0 - m), p1 = exp_acc_t(d1 - m);
                        L += p0 + p1;
                        u = fmaf(p0, a0, fmaf(p1, a1, u));
                        fma4(p0, x0, z);
                        fma4(p1, x1, z);
                    }
                    if (p < pe_) {
                        const int2 en = Es[p];
                        const float4 x = Xs[en.x * 4 + part];
                        const float av = __int_as_float(en.y);
                        const float d = fmaf(av, tq, quad_sum4(dot4(q, x)));
                        if (__any(d > m)) {
                            const float mn = fmaxf(m, d);
                            const float sc = exp_acc_t(m - mn);
                            L *= sc; u *= sc;
                            z.x *= sc; z.y *= sc; z.z *= sc; z.w *= sc;
                            m = mn;
                        }
                        const float pr = exp_acc_t(d - m);
                        L += pr;
                        u = fmaf(pr, av, u);
                        fma4(pr, x, z);
                    }
                    St[rl * 5 + 1 + part] = z;
                    if (part == 0) St[rl * 5] = make_float4(m, L, u, tq);
                }
            }
        }
    };

    float4 pxA0, pxA1, pxB0, pxB1;
    int2 peA0, peA1, peA2, peB0, peB1, peB2;
    int ppA = 0, pmA = 0, seg0A = 0, lenA = 0, ppB = 0, pmB = 0, seg0B = 0, lenB = 0;
    const bool early = wave < T_WAVES / 2;
    if (tb0 < tb1) {
        F_PREFETCH(A, tb0)
        F_PREFETCH(B, min(tb0 + 1, tb1 - 1))
    }
    for (int tb = tb0; tb < tb1; tb += 2) {
        F_DO_BLOCK(A, tb)
        if (tb + 1 < tb1) F_DO_BLOCK(B, tb + 1)
    }
#undef F_PREFETCH
#undef F_DO_BLOCK
    __syncthreads();

    // ---- epilogue: 16 lanes per row (lane gl = output channel), 64 rows per pass ----
    const int gl = tid & 15;
    float wv[16], ws[16];
    load_row16(a.p.Wv + gl * 16, wv);
    load_row16(a.p.Ws + gl * 16, ws);
    const float b_s = a.p.bs[gl], b_v = a.p.bv[gl], w_e = a.p.we[gl];
    for (int r = tid >> 4; r < n_rows; r += T_THREADS / 16) {
        const float4 hd = St[r * 5];
        const float rinv = 1.0f / (hd.y + 1e-16f);
        const float S = hd.y * rinv, un = hd.z * rinv;
        float zn[16], xr[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 zq = St[r * 5 + 1 + q];
            zn[4 * q] = zq.x * rinv; zn[4 * q + 1] = zq.y * rinv; zn[4 * q + 2] = zq.z * rinv; zn[4 * q + 3] = zq.w * rinv;
        }
        const size_t row = (size_t)row0 + r;
        load_row16(a.xd + row * 16, xr);
        float o = b_s;
        o = fmaf(S, b_v, o);
        o = fmaf(un, w_e, o);
        o = dot16(wv, zn, o);
        o = dot16(ws, xr, o);
        a.h[row * 16 + gl] = fmaxf(o, 0.0f);
        a.Z[row * 16 + gl] = select16(zn, gl);
        if (gl == 0) reinterpret_cast<float4*>(a.aux)[row] = make_float4(un, hd.y > 0.0f ? hd.x : 0.0f, rinv, S);
    }
}

int launch_fwd16_tiled(const Tiled& tl, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                       const float* x_src, const float* x_dst, float* h_out, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    TiledDev d;
    d.tile_blk = tl.tile_blk; d.blk_id = tl.blk_id; d.ptr2 = tl.ptr2; d.perm = tl.perm;
    d.ent = reinterpret_cast<const int2*>(tl.ent);
    d.n_tiles = tl.n_tiles; d.n_dst = n_dst; d.n_src = n_src;
    FwdTiledArgs a;
    a.X = x_src; a.xd = x_dst; a.qp = w.qp; a.tq = w.t; a.p = conv_params_at(conv_params, 16);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    hipLaunchKernelGGL(fwd16_tiled_kernel, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "fwd16_tiled");
}

// timing-only ablation switch (MLLP_TILED_ABLATION environment variable; 0 in production)
static int g_tiled_ablation = [] {
    const char* e = getenv("MLLP_TILED_ABLATION");
    return e ? atoi(e) : 0;
}();

int launch_spmm_tiled(const Tiled& tl, int n_dst, int n_src, const float* H, float* Y, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    TiledDev d;
    d.tile_blk = tl.tile_blk; d.blk_id = tl.blk_id; d.ptr2 = tl.ptr2; d.perm = tl.perm;
    d.ent = reinterpret_cast<const int2*>(tl.ent);
    d.n_tiles = tl.n_tiles; d.n_dst = n_dst; d.n_src = n_src;
    switch (g_tiled_ablation) {
        case 0: hipLaunchKernelGGL(spmm_tiled_kernel<0>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 1: hipLaunchKernelGGL(spmm_tiled_kernel<1>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 2: hipLaunchKernelGGL(spmm_tiled_kernel<2>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 3: hipLaunchKernelGGL(spmm_tiled_kernel<3>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 4: hipLaunchKernelGGL(spmm_tiled_kernel<4>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 7: hipLaunchKernelGGL(spmm_tiled_kernel<7>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 12: hipLaunchKernelGGL(spmm_tiled_kernel<12>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        default: return fail(MLLP_EINVAL, "unknown ablation mask");
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "spmm_tiled");
}

int tiled_max_blocks_per_tile() { return T_MAXB; }

int tiled_geometry(int variant, int* rows_per_tile, int* cols_per_block, int* bundle_capacity) {
    if (variant == 0) {          // plain SpMM
        *rows_per_tile = T_R; *cols_per_block = T_CB; *bundle_capacity = T_ECAP;
    } else {                     // attention sweeps
        *rows_per_tile = F_R; *cols_per_block = F_CB; *bundle_capacity = F_ECAP;
    }
    return MLLP_OK;
}

}  // namespace mllp
