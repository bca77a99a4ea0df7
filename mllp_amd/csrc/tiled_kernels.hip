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
//             prefetched into REGISTERS while the previous block is computed (issue early, write late).
//             Then every QUAD of lanes takes one sorted position: the 16 quads of a wave therefore walk rows of
//             (nearly) EQUAL length in lockstep -- a first version (quad = fixed rows, Poisson lengths) ran the
//             SIMD and the LDS at 59 % efficiency.  A quad loads the row's accumulator from LDS, walks the row
//             (entry: ds_read_b64, source row piece: ds_read_b128, 4 FMAs per lane), stores it back.
//   traffic   per tile: the entry stream once from HBM + (blocks touched) x 64 KB of H from L2 as wide coalesced
//             loads: one L2 request per 128 B instead of one per nonzero; the output tile is written once.
#include <cstdlib>

#include "device_utils.h"
#include "internal.h"
#include "scalar_ops.h"

namespace mllp {

constexpr int T_THREADS = 1024;            // 16 waves
constexpr int T_WAVES = T_THREADS / 64;
constexpr int T_R = 512;                   // rows per tile: 32 bundles of 16 sorted positions
constexpr int T_BUNDLES = T_R / 16;
constexpr int T_CB = 1024;                 // source nodes per column block (64 KB of fp32 x 16)
constexpr int T_ECAP = 6 * T_THREADS;      // 6144 entries (48 KB) of a (tile, block) segment per window
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
    const int2* __restrict__ ent;       // [nnz] {byte offset of the source row inside the staged block = col_local * 64, value bits}
    int n_tiles, n_dst, n_src;
};

// =================================================================================================
// Wave-specialised SpMM.  Cycle counters of a first version in which all 16 waves loaded and walked
// (profiles/README.md) showed every wave blocked 21 % of its time while ISSUING its prefetch (the CU's vector-memory path takes
// 64 B/clk and all 16 waves burst 110 KB per block at once), 30 % in barriers / waiting for loads, and a walk
// that costs 15 VALU instructions per nonzero.  Here, same geometry (512 rows x 1024 columns):
//   waves 12..15  ENTRY LOADERS: entries, offsets and perm stream from HBM into two register sets (block b+1
//             landed, block b+2 in flight); between barrier A (walkers done) and barrier B (image ready) they
//             write the landed set into LDS and then refill it with block b+3.
//   waves 8..11   H LOADERS: the 64 KB of H of the next block (L2 hits: the tiles of an instance share them), one
//             register set, one block of lookahead.  Back-pressure of the memory path stalls only loader waves.
//   waves 0..7    WALKERS: two pairs of length-sorted bundles each (snake: w and 15 - w), two rows per quad at a
//             time.  A quad reads FOUR entries of a row with one 8-byte read per lane and shares them with
//             quad-perm DPP; per nonzero: one add with a DPP source (the stored entry holds the byte offset of the
//             H row), one DPP broadcast of the value, two packed FMAs = 6.5 VALU instructions.  The walk is bound
//             by VALU issue: 12 walker waves (768 x 768 geometry, walkers staging H themselves) walked a block no
//             faster than 8.
//   entries of a (row, block) run are ordered round-robin over (column mod 4), starting at the slot of the quad
//             inside its ds_read_b128 lane group (LPBatch.enable_tiled): four quads served in one LDS cycle then
//             hit four different quarters of the 256-byte bank row.
// A segment longer than one window is walked as several (block, window) items, each staged like a block; the
// item count is padded to even so that the entry loaders' two-set loop is straight-line code.
// =================================================================================================
constexpr int W_NW = 8;                     // walker waves
constexpr int W_LT = 256;                   // threads per loader role (4 waves load H, 4 waves load entries)
constexpr int W_XPT = (T_CB * 4) / W_LT;    // 16 float4 of H per H-loader thread
constexpr int W_EPT = T_ECAP / W_LT;        // 24 entries per entry-loader thread
static_assert(W_NW * 64 + 2 * W_LT == T_THREADS && T_R == 2 * W_LT && T_BUNDLES == 4 * W_NW, "role split");

template <int K>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// acc += v * x on four channels as two packed FMAs (v_pk_fma_f32)
__device__ __forceinline__ void pk_fma4(float v, const float4& x, f32x2& lo, f32x2& hi) {
    const f32x2 vv = {v, v};
    lo = __builtin_elementwise_fma(vv, f32x2{x.x, x.y}, lo);
    hi = __builtin_elementwise_fma(vv, f32x2{x.z, x.w}, hi);
}
// 16 bytes of the staged H image at a byte offset
__device__ __forceinline__ float4 lds_row(const float4* img, int byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(img) + byte_off);
}

// the same from a bf16 image (32-byte rows: `byte_off` is the fp32 offset, halved here): 4 channels of a lane, widened
// exactly (a bf16 is the upper half of an fp32)
__device__ __forceinline__ float4 lds_row_bf16(const float4* img, int byte_off) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(img) + (byte_off >> 1));
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
template <bool BF>
__device__ __forceinline__ float4 lds_hrow(const float4* img, int byte_off) {
    return BF ? lds_row_bf16(img, byte_off) : lds_row(img, byte_off);
}

// ABL: 4 = no walk (timing only), 16 = timing build: instead of the result Y receives cycle counters (s_memtime):
//   row 2*tile      walkers, summed over waves: [0] wait at A, [1] stage (A..B), [4] walk, [5] total; loaders at [8..]:
//                   [8] wait at B, [9] vmcnt wait + ds_write, [10] wait at A, [11] load issue, [13] total
//   row 2*tile + 1  per walker wave: walk cycles [0..7], wait at A [8..15]
// BF: H is bf16 [n_src, 16] (opt-in `mllp_spmm_csr_bf16`: half the H bytes from L2, half the staging writes and row
// reads in LDS, fp32 accumulation; the products differ from the fp32 path by the rounding of H to 8 mantissa bits)
template <int ABL, bool BF = false>
__global__ __launch_bounds__(T_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void spmm_tiled_ws_kernel(TiledDev t, const float* __restrict__ X, float* __restrict__ Y) {
    __shared__ float4 Xs[T_CB * 4];     // 64 KB  staged column block of H
    __shared__ int2 Es[T_ECAP];         // 48 KB  entry window
    __shared__ float4 Ya[T_R * 4];      // 32 KB  accumulators of the row tile
    __shared__ int2 PP[T_R + 8];        // per sorted position {offset inside the segment, row}; [T_R].x = segment length
    __shared__ int Sg[T_MAXB + 1];      // segment start of every block of this tile
    __shared__ int Bk[T_MAXB];          // column-block id of every block of this tile
    __shared__ int n_items_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];
    const int nb = tb1 - tb0;
    if (nb == 0) {      // a tile whose rows are all empty: nothing to stage (the loaders' lookahead needs >= 1 block)
        const int n4z = min(T_R, t.n_dst - tile * T_R) * 4;
        float4* dz = reinterpret_cast<float4*>(Y + (size_t)tile * T_R * 16);
        if (!(ABL & 16))
            for (int i = tid; i < n4z; i += T_THREADS) dz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }

    for (int i = tid; i < T_R * 4; i += T_THREADS) Ya[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid <= nb) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * T_R];
    if (tid < nb) Bk[tid] = t.blk_id[tb0 + tid];
    if (tid == 0) n_items_s = 0;
    __syncthreads();
    if (tid < nb) {
        const int len = Sg[tid + 1] - Sg[tid];
        atomicAdd(&n_items_s, max(1, (len + T_ECAP - 1) / T_ECAP));
    }
    __syncthreads();
    const int n_items = n_items_s;
    const int n_even = (n_items + 1) & ~1;   // every role runs this many stage / barrier rounds

    unsigned cyc[5] = {0u, 0u, 0u, 0u, 0u};
    const unsigned start_ = (ABL & 16) ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    unsigned last_ = start_;
#define W_TICK(K)                                                                                           \
    if (ABL & 16) {                                                                                         \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();                                       \
        cyc[K] += now_ - last_;                                                                             \
        last_ = now_;                                                                                       \
    }
    // item = (block index inside the tile, window start); all roles step through the same sequence
#define W_ADVANCE(B, W)                                                                                     \
    {                                                                                                       \
        W += T_ECAP;                                                                                        \
        if (W >= Sg[B + 1] - Sg[B]) { W = 0; B += 1; }                                                      \
    }

    if (wave >= W_NW + W_LT / 64) {
        // ------------------------------------------------ entry loaders ------------------------------
        const int lid = tid - (W_NW * 64 + W_LT);
        int2 peA0, peA1, peA2, peA3, peA4, peA5, peA6, peA7, peA8, peA9, peA10, peA11, peA12, peA13, peA14, peA15, peA16, peA17, peA18, peA19, peA20, peA21, peA22, peA23;
        int2 peB0, peB1, peB2, peB3, peB4, peB5, peB6, peB7, peB8, peB9, peB10, peB11, peB12, peB13, peB14, peB15, peB16, peB17, peB18, peB19, peB20, peB21, peB22, peB23;
        int2 ppA = make_int2(0, 0), pmA = ppA, ppB = ppA, pmB = ppA;
        int lenA = 0, lenB = 0, segA = 0, segB = 0;
        static_assert(W_EPT == 24, "the entry-loader macros are written for 24 loads per thread");
        // (scalar variables, not arrays: arrays indexed in macro loops were demoted to scratch memory)
#define W_LDE(S, K) pe##S##K = es_[(unsigned)min(lid + K * W_LT, wl_ - 1)];
#define W_STE(S, K) Es[lid + K * W_LT] = pe##S##K;
#define W_LOADE(S, B, W)                                                                                    \
    {                                                                                                       \
        const int b_ = min((B), nb - 1);                       /* past the end: refetch the last block */  \
        const int w_ = (B) < nb ? (W) : 0;                                                                  \
        seg##S = __builtin_amdgcn_readfirstlane(Sg[b_]);                                                    \
        len##S = __builtin_amdgcn_readfirstlane(Sg[b_ + 1]) - seg##S;                                       \
        const int wl_ = max(min(T_ECAP, len##S - w_), 1);                                                   \
        const int2* es_ = t.ent + seg##S + w_;                                                              \
        /* (offsets are made segment-relative at stage time: touching the loaded value here would wait for it) */ \
        pp##S = reinterpret_cast<const int2*>(t.ptr2 + (size_t)(tb0 + b_) * T_R)[lid];                      \
        pm##S = reinterpret_cast<const int2*>(t.perm + (size_t)(tb0 + b_) * T_R)[lid];                      \
        W_LDE(S, 0) W_LDE(S, 1) W_LDE(S, 2) W_LDE(S, 3)                                                      \
        W_LDE(S, 4) W_LDE(S, 5) W_LDE(S, 6) W_LDE(S, 7)                                                      \
        W_LDE(S, 8) W_LDE(S, 9) W_LDE(S, 10) W_LDE(S, 11)                                                    \
        W_LDE(S, 12) W_LDE(S, 13) W_LDE(S, 14) W_LDE(S, 15)                                                  \
        W_LDE(S, 16) W_LDE(S, 17) W_LDE(S, 18) W_LDE(S, 19)                                                  \
        W_LDE(S, 20) W_LDE(S, 21) W_LDE(S, 22) W_LDE(S, 23)                                                  \
    }
#define W_STAGEE(S)                                                                                         \
    {                                                                                                       \
        W_TICK(3)                                                                                           \
        __syncthreads();                     /* A: the walkers are done with the previous image */          \
        W_TICK(2)                                                                                           \
        W_STE(S, 0) W_STE(S, 1) W_STE(S, 2) W_STE(S, 3)                                                      \
        W_STE(S, 4) W_STE(S, 5) W_STE(S, 6) W_STE(S, 7)                                                      \
        W_STE(S, 8) W_STE(S, 9) W_STE(S, 10) W_STE(S, 11)                                                    \
        W_STE(S, 12) W_STE(S, 13) W_STE(S, 14) W_STE(S, 15)                                                  \
        W_STE(S, 16) W_STE(S, 17) W_STE(S, 18) W_STE(S, 19)                                                  \
        W_STE(S, 20) W_STE(S, 21) W_STE(S, 22) W_STE(S, 23)                                                  \
        reinterpret_cast<int4*>(PP)[lid] = make_int4(pp##S.x - seg##S, pm##S.x, pp##S.y - seg##S, pm##S.y); \
        if (lid == 0) PP[T_R] = make_int2(len##S, 0);                                                       \
        W_TICK(1)                                                                                           \
        __syncthreads();                     /* B: image ready */                                           \
        W_TICK(0)                                                                                           \
    }
        int be = 0, we = 0;          // item whose entries are loaded next
        W_LOADE(A, be, we)
        W_ADVANCE(be, we)
        __builtin_amdgcn_sched_barrier(0);   // keep set A older than set B for the counted vmcnt waits
        W_LOADE(B, be, we)
        if (be < nb) W_ADVANCE(be, we)
        // Straight-line pairs: with an `if` or a `break` between the halves the structurised loop has a path on
        // which set A is the youngest at the loop top, and the compiler's vmcnt bookkeeping then waits for everything.
        for (int it = 0; it < n_even; it += 2) {
            W_STAGEE(A)
            W_LOADE(A, be, we)
            if (be < nb) W_ADVANCE(be, we)
            W_STAGEE(B)
            W_LOADE(B, be, we)
            if (be < nb) W_ADVANCE(be, we)
        }
#undef W_LDE
#undef W_STE
#undef W_LOADE
#undef W_STAGEE
    } else if (wave >= W_NW) {
        // ------------------------------------------------ H loaders ----------------------------------
        const int lid = tid - W_NW * 64;
        float4 px0 = {}, px1 = {}, px2 = {}, px3 = {}, px4 = {}, px5 = {}, px6 = {}, px7 = {}, px8 = {}, px9 = {}, px10 = {},
               px11 = {}, px12 = {}, px13 = {}, px14 = {}, px15 = {};
        static_assert(W_XPT == 16, "the H-loader macros are written for 16 loads per thread");
#define W_LDX(K) if (!BF || K < W_XPT / 2) px##K = src_[(unsigned)min(lid + K * W_LT, c4_ - 1)];
#define W_STX(K) if (!BF || K < W_XPT / 2) Xs[lid + K * W_LT] = px##K;
#define W_LOADX(B)                                                                                          \
    {                                                                                                       \
        const int b_ = min((B), nb - 1);                                                                    \
        const int c0_ = __builtin_amdgcn_readfirstlane(Bk[b_]) * T_CB;   /* wave-uniform: SGPR base */     \
        const int c4_ = min(T_CB, t.n_src - c0_) * (BF ? 2 : 4);          /* 16-byte pieces of the block */  \
        const float4* src_ = reinterpret_cast<const float4*>(X + (size_t)c0_ * (BF ? 8 : 16));             \
        W_LDX(0) W_LDX(1) W_LDX(2) W_LDX(3)                                                                  \
        W_LDX(4) W_LDX(5) W_LDX(6) W_LDX(7)                                                                  \
        W_LDX(8) W_LDX(9) W_LDX(10) W_LDX(11)                                                                \
        W_LDX(12) W_LDX(13) W_LDX(14) W_LDX(15)                                                              \
    }
        int bx = 0, wx = 0;          // item whose H block is loaded next
        W_LOADX(bx)
        W_ADVANCE(bx, wx)
        for (int it = 0; it < n_even; ++it) {
            W_TICK(3)
            __syncthreads();                     // A
            W_TICK(2)
            W_STX(0) W_STX(1) W_STX(2) W_STX(3)
            W_STX(4) W_STX(5) W_STX(6) W_STX(7)
            W_STX(8) W_STX(9) W_STX(10) W_STX(11)
            W_STX(12) W_STX(13) W_STX(14) W_STX(15)
            W_TICK(1)
            __syncthreads();                     // B
            W_TICK(0)
            W_LOADX(bx)
            if (bx < nb) W_ADVANCE(bx, wx)
        }
#undef W_LDX
#undef W_STX
#undef W_LOADX
    } else {
        // ------------------------------------------------ walkers ------------------------------------
        const int quad = lane >> 2, part = lane & 3, part16 = part * 16;
        int bi = 0, w0 = 0;          // current item
        for (int it = 0; it < n_even; ++it) {
            W_TICK(4)
            __syncthreads();         // A
            W_TICK(0)
            __syncthreads();         // B
            W_TICK(1)
            if (it >= n_items) break;            // the padding round of an odd item count: barriers only
            const int w1 = w0 + T_ECAP;
            // headers of both passes (offsets, rows, accumulators, first entries) are read up front, so the second
            // pass starts without dependent LDS round trips.  (Handing the pairs out from an LDS counter, longest
            // first, was slower and did not shorten the wait at the barrier.)
            // second bundle pair of each walker: measured per-pair costs (tools/phase_cycles.py) balanced by longest-
            // processing-time assignment; the plain snake (15 - wave) left walker 4 the slowest by 5 %
            const int k0 = wave * 32 + quad, k1 = (int)((0xADCE98BFu >> (4 * wave)) & 15u) * 32 + quad;
            const int2 hA0 = PP[k0], hA0n = PP[k0 + 1], hB0 = PP[k0 + 16], hB0n = PP[k0 + 17];
            const int2 hA1 = PP[k1], hA1n = PP[k1 + 1], hB1 = PP[k1 + 16], hB1n = PP[k1 + 17];
#define W_ROWS(P)                                                                                           \
    const int sA##P = max(hA##P.x, w0) - w0, eA##P = min(hA##P##n.x, w1) - w0;                              \
    const int sB##P = max(hB##P.x, w0) - w0, eB##P = min(hB##P##n.x, w1) - w0;                              \
    float4 aA##P = Ya[hA##P.y * 4 + part], aB##P = Ya[hB##P.y * 4 + part];                                 \
    int2 nA##P = Es[min(sA##P + part, T_ECAP - 1)], nB##P = Es[min(sB##P + part, T_ECAP - 1)];
            W_ROWS(0)
            W_ROWS(1)
#undef W_ROWS
#define W_WALK(P)                                                                                           \
    if ((sA##P < eA##P || sB##P < eB##P) && !(ABL & 4)) {                                                   \
        const int n_it = max(eA##P - sA##P, eB##P - sB##P);                                                 \
        f32x2 cA0 = {aA##P.x, aA##P.y}, cA1 = {aA##P.z, aA##P.w};                                           \
        f32x2 cB0 = {aB##P.x, aB##P.y}, cB1 = {aB##P.z, aB##P.w};                                           \
        int2 nA = nA##P, nB = nB##P;                                                                        \
        for (int j = 0; j < n_it; j += 4) {                                                                 \
            /* four entries of each row, one per lane of the quad; lanes past the end contribute 0 * H[0] */ \
            const int2 mA = nA, mB = nB;                                                                    \
            const int pA = sA##P + j + part, pB = sB##P + j + part;                                         \
            nA = Es[min(pA + 4, T_ECAP - 1)];                                                               \
            nB = Es[min(pB + 4, T_ECAP - 1)];                                                               \
            const int colA = pA < eA##P ? mA.x : 0, valA = pA < eA##P ? mA.y : 0;                           \
            const int colB = pB < eB##P ? mB.x : 0, valB = pB < eB##P ? mB.y : 0;                           \
            const float4 x0 = lds_hrow<BF>(Xs, quad_bcast<0>(colA) + part16), x1 = lds_hrow<BF>(Xs, quad_bcast<1>(colA) + part16), \
                         x2 = lds_hrow<BF>(Xs, quad_bcast<2>(colA) + part16), x3 = lds_hrow<BF>(Xs, quad_bcast<3>(colA) + part16); \
            const float4 y0 = lds_hrow<BF>(Xs, quad_bcast<0>(colB) + part16), y1 = lds_hrow<BF>(Xs, quad_bcast<1>(colB) + part16), \
                         y2 = lds_hrow<BF>(Xs, quad_bcast<2>(colB) + part16), y3 = lds_hrow<BF>(Xs, quad_bcast<3>(colB) + part16); \
            pk_fma4(__int_as_float(quad_bcast<0>(valA)), x0, cA0, cA1);                                     \
            pk_fma4(__int_as_float(quad_bcast<1>(valA)), x1, cA0, cA1);                                     \
            pk_fma4(__int_as_float(quad_bcast<2>(valA)), x2, cA0, cA1);                                     \
            pk_fma4(__int_as_float(quad_bcast<3>(valA)), x3, cA0, cA1);                                     \
            pk_fma4(__int_as_float(quad_bcast<0>(valB)), y0, cB0, cB1);                                     \
            pk_fma4(__int_as_float(quad_bcast<1>(valB)), y1, cB0, cB1);                                     \
            pk_fma4(__int_as_float(quad_bcast<2>(valB)), y2, cB0, cB1);                                     \
            pk_fma4(__int_as_float(quad_bcast<3>(valB)), y3, cB0, cB1);                                     \
        }                                                                                                   \
        Ya[hA##P.y * 4 + part] = make_float4(cA0.x, cA0.y, cA1.x, cA1.y);                                   \
        Ya[hB##P.y * 4 + part] = make_float4(cB0.x, cB0.y, cB1.x, cB1.y);                                   \
    }
            W_WALK(0)
            W_WALK(1)
#undef W_WALK
            W_ADVANCE(bi, w0)
        }
    }
#undef W_ADVANCE
    __syncthreads();
    if (ABL & 16) {
        const unsigned total_ = (unsigned)__builtin_amdgcn_s_memtime() - start_;
        int* acc = reinterpret_cast<int*>(PP);
        if (tid < 16) acc[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            const int o = wave >= W_NW ? 8 : 0;
            for (int k = 0; k < 5; ++k) atomicAdd(&acc[o + k], (int)cyc[k]);
            atomicAdd(&acc[o + 5], (int)total_);
        }
        __syncthreads();
        if (tid < 16 && 2 * tile + 1 < t.n_dst) Y[(size_t)(2 * tile) * 16 + tid] = (float)acc[tid];
        if (lane == 0 && wave < W_NW && 2 * tile + 1 < t.n_dst) {
            Y[(size_t)(2 * tile + 1) * 16 + wave] = (float)cyc[4];
            Y[(size_t)(2 * tile + 1) * 16 + 8 + wave] = (float)cyc[0];
        }
        return;
    }
#undef W_TICK
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
                        const float4 x0 = Xs[(e0.x >> 4) + part], x1 = Xs[(e1.x >> 4) + part];
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
                        // the logits are the same in the four lanes of the quad: lanes 0, 2 take the exponential of the
                        // first entry, lanes 1, 3 of the second, and share them by DPP (the exponential is the most
                        // expensive operation of the walk; same arithmetic per value, two instead of four per lane-pair)
                        const float pmine = exp_acc_t(((part & 1) ? d1 : d0) - m);
                        const float p0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(pmine), 0xA0, 0xF, 0xF, true));   // quad_perm [0,0,2,2]
                        const float p1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(pmine), 0xF5, 0xF, 0xF, true));   // quad_perm [1,1,3,3]
                        L += p0 + p1;
                        u = fmaf(p0, a0, fmaf(p1, a1, u));
                        fma4(p0, x0, z);
                        fma4(p1, x1, z);
                    }
                    if (p < pe_) {
                        const int2 en = Es[p];
                        const float4 x = Xs[(en.x >> 4) + part];
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

// =================================================================================================
// Attention backward, source-major, in the tiled form: dX_j = sum_i alpha_ij gv_i + dl_ij q'_i.
// Rows of this sweep are SOURCE nodes j (the orientation opposite to the conv's destination-major one); the
// column blocks hold the 160-byte backward RECORDS of the destination nodes {q'[16], gv[16], t, rowmax, 1/rowsum,
// ge, c}.  The generic sweep gathers 148 bytes per nonzero from L2 (the largest item of the training step on the
// synthetic batch); here a block of 384 records (60 KB, contiguous in memory) is staged in LDS, and the tile keeps
// x_j and the running dX_j of its 512 rows in LDS.  Geometry (variant 2): 512 rows x 384 columns, 3072-entry
// windows.  Per nonzero a quad recomputes the logit (4 FMAs + DPP quad sum), alpha, dl (second dot product) and
// adds alpha gv + dl q' (8 FMAs).  Same staging scheme as the attention forward kernel above.
// =================================================================================================
constexpr int S_R = 512;
constexpr int S_BUNDLES = S_R / 16;
constexpr int S_CB = 384;
constexpr int S_ECAP = 3 * T_THREADS;      // 3072 entries (24 KB) per window
constexpr int S_REC4 = REC_W / 4;          // float4 per record
static_assert(REC_W == 40, "record layout: q'[16] gv[16] {t, rowmax, rinv, ge} c ...");
static_assert(S_CB * S_REC4 <= 4 * T_THREADS, "record block staged with 4 float4 per thread");

struct BwdSrcTiledArgs {
    const float* __restrict__ X;        // [n_rows, 16] features of the source nodes (rows of this sweep)
    const float* __restrict__ rec;      // [n_cols, REC_W] backward records of the destination nodes
    float* __restrict__ dX;             // [n_rows, 16]
    int accumulate;
};

__global__ __launch_bounds__(T_THREADS) void bwdsrc16_tiled_kernel(TiledDev t, BwdSrcTiledArgs a) {
    __shared__ float4 Rs[S_CB * S_REC4];    // 60 KB  staged records of the column block
    __shared__ int2 Es[S_ECAP];             // 24 KB  entry segment (window): {record byte offset, value bits}
    __shared__ float4 Xj[S_R * 4];          // 32 KB  x_j of the tile's rows
    __shared__ float4 Ac[S_R * 4];          // 32 KB  running dX_j
    __shared__ int Ps[S_R + 16];
    __shared__ int Pm[S_R];
    __shared__ int Sg[T_MAXB + 1];
    __shared__ int Bk[T_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = lane >> 2, part = lane & 3;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];
    const int row0 = tile * S_R;
    const int n_rows = min(S_R, t.n_dst - row0);

    for (int i = tid; i < S_R * 4; i += T_THREADS) {
        Xj[i] = (i >> 2) < n_rows ? reinterpret_cast<const float4*>(a.X + (size_t)row0 * 16)[i]
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        Ac[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid <= tb1 - tb0) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * S_R];
    if (tid < tb1 - tb0) Bk[tid] = t.blk_id[tb0 + tid];
    __syncthreads();

#define S_PREFETCH(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbx_ = (TB);                                                                              \
        const int c0_ = Bk[tbx_ - tb0] * S_CB;                                                              \
        const int c4_ = min(S_CB, t.n_src - c0_) * S_REC4;                                                  \
        const float4* src_ = reinterpret_cast<const float4*>(a.rec + (size_t)c0_ * REC_W);                 \
        pr##S##0 = src_[min(tid, c4_ - 1)];                                                                 \
        pr##S##1 = src_[min(tid + T_THREADS, c4_ - 1)];                                                     \
        pr##S##2 = src_[min(tid + 2 * T_THREADS, c4_ - 1)];                                                 \
        pr##S##3 = src_[min(tid + 3 * T_THREADS, c4_ - 1)];                                                 \
        seg0##S = Sg[tbx_ - tb0];                                                                           \
        len##S = Sg[tbx_ - tb0 + 1] - seg0##S;                                                              \
        pp##S = t.ptr2[(size_t)tbx_ * S_R + min(tid, S_R - 1)] - seg0##S;                                   \
        pm##S = t.perm[(size_t)tbx_ * S_R + min(tid, S_R - 1)];                                             \
        pe##S##0 = t.ent[seg0##S + min(tid, max(len##S - 1, 0))];                                           \
        pe##S##1 = t.ent[seg0##S + min(tid + T_THREADS, max(len##S - 1, 0))];                               \
        pe##S##2 = t.ent[seg0##S + min(tid + 2 * T_THREADS, max(len##S - 1, 0))];                           \
    }
#define S_DO_BLOCK(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbc_ = (TB);                                                                              \
        __syncthreads();                                                                                    \
        Rs[tid] = pr##S##0; Rs[tid + T_THREADS] = pr##S##1; Rs[tid + 2 * T_THREADS] = pr##S##2;             \
        if (tid + 3 * T_THREADS < S_CB * S_REC4) Rs[tid + 3 * T_THREADS] = pr##S##3;                        \
        Es[tid] = pe##S##0; Es[tid + T_THREADS] = pe##S##1; Es[tid + 2 * T_THREADS] = pe##S##2;             \
        if (tid < S_R) {                                                                                    \
            Ps[tid] = pp##S;                                                                                \
            Pm[tid] = pm##S;                                                                                \
        }                                                                                                   \
        if (tid == 0) Ps[S_R] = len##S;                                                                     \
        const int cur_seg0 = seg0##S, cur_len = len##S;                                                     \
        __syncthreads();                                                                                    \
        const int tb_next = min(tbc_ + 2, tb1 - 1);                                                         \
        if (early) S_PREFETCH(S, tb_next)                                                                   \
        walk(cur_seg0, cur_len);                                                                            \
        if (!early) S_PREFETCH(S, tb_next)                                                                  \
    }

    // one nonzero of row j against the staged record at byte offset `off`
    auto edge = [&](int off, float av, const float4& xj, float4& acc) {
        const float4* r = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Rs) + off);
        const float4 qp = r[part], gv = r[4 + part], s0 = r[8];       // s0 = {t, rowmax, rinv, ge}
        const float cc = reinterpret_cast<const float*>(r)[36];
        const float l = fmaf(av, s0.x, quad_sum4(dot4(qp, xj)));
        const float alpha = exp_acc_t(l - s0.y) * s0.z;
        const float dl = alpha * (quad_sum4(dot4(gv, xj)) + fmaf(av, s0.w, cc));
        fma4(alpha, gv, acc);
        fma4(dl, qp, acc);
    };
    // two nonzeros at once: the logits are the same in the four lanes of a quad, so lanes 0, 2 take the exponential of the
    // first and lanes 1, 3 of the second and share them by DPP (same values, same order of the accumulation as two edge() calls)
    auto edge2 = [&](int off0, float av0, int off1, float av1, const float4& xj, float4& acc) {
        const float4* r0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Rs) + off0);
        const float4* r1 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Rs) + off1);
        const float4 qp0 = r0[part], gv0 = r0[4 + part], s0 = r0[8];
        const float4 qp1 = r1[part], gv1 = r1[4 + part], s1 = r1[8];
        const float cc0 = reinterpret_cast<const float*>(r0)[36], cc1 = reinterpret_cast<const float*>(r1)[36];
        const float l0 = fmaf(av0, s0.x, quad_sum4(dot4(qp0, xj)));
        const float l1 = fmaf(av1, s1.x, quad_sum4(dot4(qp1, xj)));
        const float em = exp_acc_t((part & 1) ? l1 - s1.y : l0 - s0.y);
        const float e0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(em), 0xA0, 0xF, 0xF, true));   // quad_perm [0,0,2,2]
        const float e1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(em), 0xF5, 0xF, 0xF, true));   // quad_perm [1,1,3,3]
        const float alpha0 = e0 * s0.z, alpha1 = e1 * s1.z;
        const float dl0 = alpha0 * (quad_sum4(dot4(gv0, xj)) + fmaf(av0, s0.w, cc0));
        const float dl1 = alpha1 * (quad_sum4(dot4(gv1, xj)) + fmaf(av1, s1.w, cc1));
        fma4(alpha0, gv0, acc);
        fma4(dl0, qp0, acc);
        fma4(alpha1, gv1, acc);
        fma4(dl1, qp1, acc);
    };
    auto walk = [&](int cur_seg0, int cur_len) {
        for (int w0 = 0; w0 < cur_len; w0 += S_ECAP) {
            if (w0 > 0) {   // rare: segment longer than one window
                __syncthreads();
                for (int i = tid; i < min(S_ECAP, cur_len - w0); i += T_THREADS) Es[i] = t.ent[cur_seg0 + w0 + i];
                __syncthreads();
            }
            const int w1 = w0 + S_ECAP;
#pragma unroll
            for (int pass = 0; pass < S_BUNDLES / T_WAVES; ++pass) {
                const int bundle = (pass & 1) ? (pass + 1) * T_WAVES - 1 - wave : pass * T_WAVES + wave;
                const int k = bundle * 16 + quad;
                const int s = max(Ps[k], w0), e = min(Ps[k + 1], w1);
                if (s < e) {
                    const int rl = Pm[k];
                    const float4 xj = Xj[rl * 4 + part];
                    float4 acc = Ac[rl * 4 + part];
                    int p = s - w0;
                    const int pe_ = e - w0;
                    for (; p + 1 < pe_; p += 2) {      // two entries per pass: independent LDS reads and dot products
                        const int2 e0 = Es[p], e1 = Es[p + 1];
                        edge2(e0.x, __int_as_float(e0.y), e1.x, __int_as_float(e1.y), xj, acc);
                    }
                    if (p < pe_) {
                        const int2 e0 = Es[p];
                        edge(e0.x, __int_as_float(e0.y), xj, acc);
                    }
                    Ac[rl * 4 + part] = acc;
                }
            }
        }
    };

    float4 prA0, prA1, prA2, prA3, prB0, prB1, prB2, prB3;
    int2 peA0, peA1, peA2, peB0, peB1, peB2;
    int ppA = 0, pmA = 0, seg0A = 0, lenA = 0, ppB = 0, pmB = 0, seg0B = 0, lenB = 0;
    const bool early = wave < T_WAVES / 2;
    if (tb0 < tb1) {
        S_PREFETCH(A, tb0)
        S_PREFETCH(B, min(tb0 + 1, tb1 - 1))
    }
    for (int tb = tb0; tb < tb1; tb += 2) {
        S_DO_BLOCK(A, tb)
        if (tb + 1 < tb1) S_DO_BLOCK(B, tb + 1)
    }
#undef S_PREFETCH
#undef S_DO_BLOCK
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(a.dX + (size_t)row0 * 16);
    for (int i = tid; i < n_rows * 4; i += T_THREADS) {
        float4 v = Ac[i];
        if (a.accumulate) {
            const float4 old = dst[i];
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        dst[i] = v;
    }
}

int launch_bwdsrc16_tiled(const Tiled& tl, int n_rows, int n_cols, const float* rec, const float* x_rows, float* dx,
                          int accumulate, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    TiledDev d;
    d.tile_blk = tl.tile_blk; d.blk_id = tl.blk_id; d.ptr2 = tl.ptr2; d.perm = tl.perm;
    d.ent = reinterpret_cast<const int2*>(tl.ent);
    d.n_tiles = tl.n_tiles; d.n_dst = n_rows; d.n_src = n_cols;
    BwdSrcTiledArgs a{x_rows, rec, dx, accumulate};
    hipLaunchKernelGGL(bwdsrc16_tiled_kernel, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwdsrc16_tiled");
}

// =================================================================================================
// Attention backward, destination-major, ONE LANE PER ROW: ds_i, dt_i, dq'_i and the destination-side input gradient
// (sweep_kernels.hip::BwdDst16Op).  The quad-per-row sweeps spend 3.5 wave instructions per nonzero (two dot products
// reduced over the quad by DPP, an exp, masks); the first tiled version of this sweep (224 bytes of row state in LDS,
// 256-row tiles) was no faster than the generic L2 gather (2.65 ms at 256 M nonzeros).  Here a lane OWNS a row of the
// tile for all its column blocks: the record {q'[16], gv[16], t, rowmax, 1/rowsum, ge, c} and the running
// {dq'[16], ds, dt} stay in its registers (no row state in LDS, no per-block row order), and one nonzero costs the lane
// 24 packed FMAs + an exp.  Measured 1.84 ms at 256 M nonzeros (the walk is bound by LDS round trips: packed FMAs gave
// 3 %, a bank-aware entry order 1 %; 12 walker wavefronts with 4 loader wavefronts were slower, 2.6 ms -- the loaders'
// register sets no longer fit).
//   * geometry (variant 4): 512 rows x 512 columns, 3072-entry windows; rows keep their order (lane l of walker w =
//     row 64 w + l of the tile).  The entries of every 64-row chunk of a (tile, block) are stored by STEP: entry j of the
//     rows that have one, in lane order -- step j of a wavefront reads one contiguous run, a lane's entry sits at
//     S_j + (number of lower lanes that still have an entry): mbcnt of the ballot.  Lanes whose row has ended read a
//     dummy entry (the image's zero row) and are masked.
//   * roles as in the SpMM kernel: 8 walker wavefronts, 4 H loaders, 4 entry loaders; everything staged exists twice in
//     LDS (2 x 32 KB of H, 2 x 24 KB of entries, 2 x 2 KB of row offsets): the loaders write image b + 1 from registers
//     (loaded one or two blocks earlier) while the walkers walk image b -- one barrier per block.
//   * the four 16-byte pieces of an H row are read in a per-lane rotation ((k + lane) & 3), which spreads the 16 lanes of
//     a ds_read_b128 lane group over the bank quarters; q', gv and dq' are held in the same rotated order.
//   * the rows' 16 x 16 epilogue (dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt) runs 16 lanes per row from an LDS
//     copy of dq' when all blocks are done.
// =================================================================================================
constexpr int D_R = 512, D_CB = 512, D_ECAP = 3072;
constexpr int D_XPT = (D_CB * 4) / W_LT;    // 8 float4 of H per H-loader thread
constexpr int D_EPT = D_ECAP / W_LT;        // 12 entries per entry-loader thread
constexpr int D_ZERO = D_CB * 64;           // byte offset of the image's zero row = column offset of the dummy entry
static_assert(D_R == 8 * 64 && D_R == 2 * W_LT, "role split of the lane-per-row kernel");

struct BwdDstTiledArgs {
    const float* __restrict__ X;        // [n_src, 16]
    const float* __restrict__ rec;      // [n_dst, REC_W]
    const float* __restrict__ g;        // [n_dst, 16] relu-masked output gradient
    const float* __restrict__ derived;
    float* __restrict__ dqp;            // [n_dst, 16]
    float* __restrict__ dsdt;           // [n_dst, 2]
    float* __restrict__ dx_dst;         // [n_dst, 16] or nullptr
    int accumulate;
};

// 16-term dot product as eight packed FMAs: the two lanes of the result still have to be added
__device__ __forceinline__ f32x2 pk_dot16(const float4& a0, const float4& a1, const float4& a2, const float4& a3,
                                          const float4& b0, const float4& b1, const float4& b2, const float4& b3) {
    f32x2 s = f32x2{a0.x, a0.y} * f32x2{b0.x, b0.y};
    f32x2 u = f32x2{a2.x, a2.y} * f32x2{b2.x, b2.y};
    s = __builtin_elementwise_fma(f32x2{a0.z, a0.w}, f32x2{b0.z, b0.w}, s);
    u = __builtin_elementwise_fma(f32x2{a2.z, a2.w}, f32x2{b2.z, b2.w}, u);
    s = __builtin_elementwise_fma(f32x2{a1.x, a1.y}, f32x2{b1.x, b1.y}, s);
    u = __builtin_elementwise_fma(f32x2{a3.x, a3.y}, f32x2{b3.x, b3.y}, u);
    s = __builtin_elementwise_fma(f32x2{a1.z, a1.w}, f32x2{b1.z, b1.w}, s);
    u = __builtin_elementwise_fma(f32x2{a3.z, a3.w}, f32x2{b3.z, b3.w}, u);
    return s + u;
}

__global__ __launch_bounds__(T_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void bwddst16_lane_kernel(TiledDev t, BwdDstTiledArgs a) {
    __shared__ float4 Xs[2][D_CB * 4 + 4];  // 2 x 32 KB  staged column blocks of H, each followed by a row of zeros
    __shared__ int2 Es[2][D_ECAP + 1];      // 2 x 24 KB  entry windows, each followed by the dummy entry {zero row, 0}
    __shared__ int PP[2][D_R + 8];          // offset of every row inside the segment; [D_R] = segment length
    __shared__ int Sg[T_MAXB + 1];          // segment start of every block of this tile
    __shared__ int Bk[T_MAXB];              // column-block id of every block of this tile
    __shared__ int n_items_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];
    const int nb = tb1 - tb0;
    const int row0 = tile * D_R;
    const int n_rows = min(D_R, t.n_dst - row0);
    const float* X = a.X;

    // the walkers' row state (the other roles carry zeros)
    const int rot = lane & 3;
    float4 q0 = {}, q1 = {}, q2 = {}, q3 = {}, g0 = {}, g1 = {}, g2 = {}, g3 = {};
    f32x2 d0l = {0.f, 0.f}, d0h = d0l, d1l = d0l, d1h = d0l, d2l = d0l, d2h = d0l, d3l = d0l, d3h = d0l;
    float rt = 0.0f, rm = 0.0f, rinv = 0.0f, rge = 0.0f, rcc = 0.0f, ds = 0.0f, dt = 0.0f;
    if (wave < W_NW && wave * 64 + lane < n_rows) {
        const float* r = a.rec + (size_t)(row0 + wave * 64 + lane) * REC_W;
        q0 = ld4(r + 4 * ((0 + rot) & 3)); q1 = ld4(r + 4 * ((1 + rot) & 3));
        q2 = ld4(r + 4 * ((2 + rot) & 3)); q3 = ld4(r + 4 * ((3 + rot) & 3));
        g0 = ld4(r + 16 + 4 * ((0 + rot) & 3)); g1 = ld4(r + 16 + 4 * ((1 + rot) & 3));
        g2 = ld4(r + 16 + 4 * ((2 + rot) & 3)); g3 = ld4(r + 16 + 4 * ((3 + rot) & 3));
        const float4 s0 = ld4(r + 32);
        rt = s0.x; rm = s0.y; rinv = s0.z; rge = s0.w;
        rcc = r[36];
    }

    if (tid <= nb) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * D_R];
    if (tid < nb) Bk[tid] = t.blk_id[tb0 + tid];
    if (tid == 0) n_items_s = 0;
    if (tid < 8) Xs[tid >> 2][D_CB * 4 + (tid & 3)] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 2) Es[tid][D_ECAP] = make_int2(D_ZERO, 0);
    __syncthreads();
    if (tid < nb) {
        const int len = Sg[tid + 1] - Sg[tid];
        atomicAdd(&n_items_s, max(1, (len + D_ECAP - 1) / D_ECAP));
    }
    __syncthreads();
    const int n_items = nb > 0 ? n_items_s : 0;
    const int n_even = (n_items + 1) & ~1;   // every role runs this many rounds (one barrier each)
#define D_ADVANCE(B, W)                                                                                     \
    {                                                                                                       \
        W += D_ECAP;                                                                                        \
        if (W >= Sg[B + 1] - Sg[B]) { W = 0; B += 1; }                                                      \
    }

    if (n_items == 0) {
        // a tile whose rows have no nonzeros: only the epilogue below
    } else if (wave >= W_NW + W_LT / 64) {
        // ------------------------------------------------ entry loaders ------------------------------
        const int lid = tid - (W_NW * 64 + W_LT);
        int2 peA0, peA1, peA2, peA3, peA4, peA5, peA6, peA7, peA8, peA9, peA10, peA11;
        int2 peB0, peB1, peB2, peB3, peB4, peB5, peB6, peB7, peB8, peB9, peB10, peB11;
        int2 ppA = make_int2(0, 0), ppB = ppA;
        int lenA = 0, lenB = 0, segA = 0, segB = 0;
        static_assert(D_EPT == 12, "the entry-loader macros are written for 12 loads per thread");
#define D_LDE(S, K) pe##S##K = es_[(unsigned)min(lid + K * W_LT, wl_ - 1)];
#define D_STE(S, K, BUF) Es[BUF][lid + K * W_LT] = pe##S##K;
#define D_LOADE(S, B, W)                                                                                    \
    {                                                                                                       \
        const int b_ = min((B), nb - 1);                       /* past the end: refetch the last block */  \
        const int w_ = (B) < nb ? (W) : 0;                                                                  \
        seg##S = __builtin_amdgcn_readfirstlane(Sg[b_]);                                                    \
        len##S = __builtin_amdgcn_readfirstlane(Sg[b_ + 1]) - seg##S;                                       \
        const int wl_ = max(min(D_ECAP, len##S - w_), 1);                                                   \
        const int2* es_ = t.ent + seg##S + w_;                                                              \
        pp##S = reinterpret_cast<const int2*>(t.ptr2 + (size_t)(tb0 + b_) * D_R)[lid];                      \
        D_LDE(S, 0) D_LDE(S, 1) D_LDE(S, 2) D_LDE(S, 3) D_LDE(S, 4) D_LDE(S, 5)                              \
        D_LDE(S, 6) D_LDE(S, 7) D_LDE(S, 8) D_LDE(S, 9) D_LDE(S, 10) D_LDE(S, 11)                            \
    }
#define D_STAGEE(S, BUF)                                                                                    \
    {                                                                                                       \
        D_STE(S, 0, BUF) D_STE(S, 1, BUF) D_STE(S, 2, BUF) D_STE(S, 3, BUF) D_STE(S, 4, BUF) D_STE(S, 5, BUF) \
        D_STE(S, 6, BUF) D_STE(S, 7, BUF) D_STE(S, 8, BUF) D_STE(S, 9, BUF) D_STE(S, 10, BUF) D_STE(S, 11, BUF) \
        reinterpret_cast<int2*>(PP[BUF])[lid] = make_int2(pp##S.x - seg##S, pp##S.y - seg##S);              \
        if (lid == 0) PP[BUF][D_R] = len##S;                                                                \
    }
        int be = 0, we = 0;          // item whose entries are loaded next
        D_LOADE(A, be, we)
        D_ADVANCE(be, we)
        __builtin_amdgcn_sched_barrier(0);   // keep set A older than set B for the counted vmcnt waits
        D_LOADE(B, be, we)
        if (be < nb) D_ADVANCE(be, we)
        D_STAGEE(A, 0)                       // item 0 -> image 0
        D_LOADE(A, be, we)
        if (be < nb) D_ADVANCE(be, we)
        __syncthreads();                     // image 0 ready
        // straight-line pairs (see spmm_tiled_ws_kernel): item it + 1 -> image 1 under the walk of image 0, ...
        for (int it = 0; it < n_even; it += 2) {
            D_STAGEE(B, 1)
            D_LOADE(B, be, we)
            if (be < nb) D_ADVANCE(be, we)
            __syncthreads();
            D_STAGEE(A, 0)
            D_LOADE(A, be, we)
            if (be < nb) D_ADVANCE(be, we)
            __syncthreads();
        }
#undef D_LDE
#undef D_STE
#undef D_LOADE
#undef D_STAGEE
    } else if (wave >= W_NW) {
        // ------------------------------------------------ H loaders ----------------------------------
        const int lid = tid - W_NW * 64;
        float4 px0, px1, px2, px3, px4, px5, px6, px7;
        static_assert(D_XPT == 8, "the H-loader macros are written for 8 loads per thread");
#define D_LDX(K) px##K = src_[(unsigned)min(lid + K * W_LT, c4_ - 1)];
#define D_STX(K, BUF) Xs[BUF][lid + K * W_LT] = px##K;
#define D_LOADX(B)                                                                                          \
    {                                                                                                       \
        const int b_ = min((B), nb - 1);                                                                    \
        const int c0_ = __builtin_amdgcn_readfirstlane(Bk[b_]) * D_CB;   /* wave-uniform: SGPR base */     \
        const int c4_ = min(D_CB, t.n_src - c0_) * 4;                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(X + (size_t)c0_ * 16);                        \
        D_LDX(0) D_LDX(1) D_LDX(2) D_LDX(3) D_LDX(4) D_LDX(5) D_LDX(6) D_LDX(7)                              \
    }
#define D_STAGEX(BUF)                                                                                       \
    { D_STX(0, BUF) D_STX(1, BUF) D_STX(2, BUF) D_STX(3, BUF) D_STX(4, BUF) D_STX(5, BUF) D_STX(6, BUF) D_STX(7, BUF) }
        int bx = 0, wx = 0;          // item whose H block is loaded next
        D_LOADX(bx)
        D_ADVANCE(bx, wx)
        D_STAGEX(0)
        D_LOADX(bx)
        if (bx < nb) D_ADVANCE(bx, wx)
        __syncthreads();                     // image 0 ready
        for (int it = 0; it < n_even; it += 2) {
            D_STAGEX(1)
            D_LOADX(bx)
            if (bx < nb) D_ADVANCE(bx, wx)
            __syncthreads();
            D_STAGEX(0)
            D_LOADX(bx)
            if (bx < nb) D_ADVANCE(bx, wx)
            __syncthreads();
        }
#undef D_LDX
#undef D_STX
#undef D_LOADX
#undef D_STAGEX
    } else {
        // ------------------------------------------------ walkers ------------------------------------
        int bi = 0, w0 = 0;          // current item
        __syncthreads();             // image 0 ready
        for (int it = 0; it < n_even; ++it) {
            if (it < n_items) {
                const int buf = it & 1;
                const int* Pb = PP[buf];
                const int off = Pb[wave * 64 + lane];
                const int len = Pb[wave * 64 + lane + 1] - off;
                int n = len;                                                 // steps of this chunk: its longest row
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) n = max(n, __shfl_xor(n, o, 64));
                n = __builtin_amdgcn_readfirstlane(n);
                if (n > 0) {
                    int S = __builtin_amdgcn_readfirstlane(off) - w0;       // the chunk's first entry, window-relative
                    const char* ebase = reinterpret_cast<const char*>(Es[buf]);
                    const char* xbase = reinterpret_cast<const char*>(Xs[buf]);
                    const char* c0 = xbase + ((0 + rot) & 3) * 16; const char* c1 = xbase + ((1 + rot) & 3) * 16;
                    const char* c2 = xbase + ((2 + rot) & 3) * 16; const char* c3 = xbase + ((3 + rot) & 3) * 16;
                    // entry of step J for this lane (the dummy entry when the row has ended or the window does not hold it)
#define D_ENT(E, J)                                                                                         \
    {                                                                                                       \
        const bool on_ = (J) < len;                                                                         \
        const unsigned long long m_ = __ballot(on_);                                                        \
        const int p_ = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0u)); \
        S += __popcll(m_);                                                                                  \
        const int i_ = (on_ && (unsigned)p_ < (unsigned)D_ECAP) ? p_ : D_ECAP;                              \
        E = *reinterpret_cast<const int2*>(ebase + 8 * i_);                                                 \
    }
#define D_ROWX(X_, E)                                                                                       \
    X_##0 = *reinterpret_cast<const float4*>(c0 + E.x); X_##1 = *reinterpret_cast<const float4*>(c1 + E.x); \
    X_##2 = *reinterpret_cast<const float4*>(c2 + E.x); X_##3 = *reinterpret_cast<const float4*>(c3 + E.x);
                    // one nonzero: alpha recomputed from the saved rowmax / rowsum, dl, and the three accumulations
#define D_FMA(E, X_)                                                                                        \
    {                                                                                                       \
        const float a_ = __int_as_float(E.y);                                                               \
        const f32x2 ql_ = pk_dot16(q0, q1, q2, q3, X_##0, X_##1, X_##2, X_##3);                             \
        const f32x2 gl_ = pk_dot16(g0, g1, g2, g3, X_##0, X_##1, X_##2, X_##3);                             \
        const float l_ = fmaf(a_, rt, ql_.x + ql_.y);                                                       \
        const float gx_ = gl_.x + gl_.y;                                                                    \
        const float al_ = E.x != D_ZERO ? exp_acc(l_ - rm) * rinv : 0.0f;                                   \
        const float dl_ = al_ * (gx_ + fmaf(a_, rge, rcc));                                                 \
        ds += dl_;                                                                                          \
        dt = fmaf(dl_, a_, dt);                                                                             \
        pk_fma4(dl_, X_##0, d0l, d0h); pk_fma4(dl_, X_##1, d1l, d1h);                                       \
        pk_fma4(dl_, X_##2, d2l, d2h); pk_fma4(dl_, X_##3, d3l, d3h);                                       \
    }
                    int2 e0, e1, e2, e3;
                    float4 p0, p1, p2, p3, r0, r1, r2, r3;
                    D_ENT(e0, 0) D_ENT(e1, 1) D_ENT(e2, 2) D_ENT(e3, 3)
                    D_ROWX(p, e0)
#define D_FENCE __builtin_amdgcn_sched_barrier(0);     // keep the issue order: next step's reads, THEN this step's math
                    for (int j = 0;; j += 4) {
                        D_ROWX(r, e1) D_FENCE D_FMA(e0, p) D_FENCE if (j + 1 >= n) break;
                        D_ENT(e0, j + 4) D_FENCE
                        D_ROWX(p, e2) D_FENCE D_FMA(e1, r) D_FENCE if (j + 2 >= n) break;
                        D_ENT(e1, j + 5) D_FENCE
                        D_ROWX(r, e3) D_FENCE D_FMA(e2, p) D_FENCE if (j + 3 >= n) break;
                        D_ENT(e2, j + 6) D_FENCE
                        D_ROWX(p, e0) D_FENCE D_FMA(e3, r) D_FENCE if (j + 4 >= n) break;
                        D_ENT(e3, j + 7) D_FENCE
                    }
#undef D_FENCE
#undef D_ENT
#undef D_ROWX
#undef D_FMA
                }
                D_ADVANCE(bi, w0)
            }
            __syncthreads();
        }
    }
#undef D_ADVANCE
    __syncthreads();
    // ---- the rows' results -> LDS (the images are free now): dq' [D_R][16] over Xs[0], {ds, dt} over Es[0] ----
    float4* Dq = Xs[0];
    float2* Sd = reinterpret_cast<float2*>(Es[0]);
    if (wave < W_NW) {
        const int r = wave * 64 + lane;
        Dq[r * 4 + ((0 + rot) & 3)] = make_float4(d0l.x, d0l.y, d0h.x, d0h.y);
        Dq[r * 4 + ((1 + rot) & 3)] = make_float4(d1l.x, d1l.y, d1h.x, d1h.y);
        Dq[r * 4 + ((2 + rot) & 3)] = make_float4(d2l.x, d2l.y, d2h.x, d2h.y);
        Dq[r * 4 + ((3 + rot) & 3)] = make_float4(d3l.x, d3l.y, d3h.x, d3h.y);
        Sd[r] = make_float2(ds, dt);
    }
    __syncthreads();

    // ---- epilogue: 16 lanes per row (lane gl = input channel), 64 rows per pass ----
    //      dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt          (sweep_kernels.hip::BwdDst16Op::epilogue)
    const int gl = tid & 15;
    const float* D = a.derived;
    float wsT[16], pqT[16];
    float pb = 0.f, pt = 0.f;
    if (a.dx_dst) {
        load_row16(D + OFF_WST + gl * 16, wsT);
        load_row16(D + OFF_PQT + gl * 16, pqT);
        pb = D[OFF_PB + gl];
        pt = D[OFF_PT + gl];
    }
    for (int r = tid >> 4; r < n_rows; r += T_THREADS / 16) {
        const size_t row = (size_t)row0 + r;
        const float2 sd = Sd[r];
        float da[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = Dq[r * 4 + q];
            da[4 * q] = v.x; da[4 * q + 1] = v.y; da[4 * q + 2] = v.z; da[4 * q + 3] = v.w;
        }
        a.dqp[row * 16 + gl] = select16(da, gl);
        if (gl == 0) reinterpret_cast<float2*>(a.dsdt)[row] = make_float2(sd.x, sd.y);
        if (a.dx_dst) {
            float gr[16];
            load_row16(a.g + row * 16, gr);
            float v = dot16(wsT, gr, 0.0f);
            v = fmaf(sd.x, pb, v);
            v = fmaf(sd.y, pt, v);
            v = dot16(pqT, da, v);
            float* dst = a.dx_dst + row * 16 + gl;
            *dst = a.accumulate ? *dst + v : v;
        }
    }
}

int launch_bwddst16_tiled(const Tiled& tl, int n_dst, int n_src, const ConvWs& w, const float* x_src, const float* g,
                          float* dx_dst, int accumulate, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    TiledDev d;
    d.tile_blk = tl.tile_blk; d.blk_id = tl.blk_id; d.ptr2 = tl.ptr2; d.perm = tl.perm;
    d.ent = reinterpret_cast<const int2*>(tl.ent);
    d.n_tiles = tl.n_tiles; d.n_dst = n_dst; d.n_src = n_src;
    BwdDstTiledArgs a{x_src, w.rec, g, w.derived, w.dqp, w.dsdt, dx_dst, accumulate};
    hipLaunchKernelGGL(bwddst16_lane_kernel, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwddst16_tiled");
}

// =================================================================================================
// Layer-1 attention sweeps (scalar node features, reference linear_program_methods.py:90-91, 241-242) in the
// tiled form.  The generic sweeps gather one 4-byte x_j per nonzero and lane from L2 (one request per nonzero:
// 4.4 ms forward, 3.1 ms backward at 512 M nonzeros).  With one channel a column block of 1024 sources is only
// 4 KB, so the workgroup is small: 256 threads = one 256-row tile, ~40 KB of LDS, four workgroups per CU that
// overlap each other's staging and walking.  ONE LANE PER ROW: a lane walks the entries of its length-sorted row
// (entry: ds_read_b64, x_j: ds_read_b32, a handful of FMAs, one exp) with no cross-lane traffic at all; wave w
// takes sorted positions 64w .. 64w+63, i.e. rows of nearly equal length.  Geometry (variant 3): 256 rows x 1024
// columns, 3072-entry windows; entries {byte offset of x_j in the block = 4 * local column, value}.
// =================================================================================================
constexpr int C_THREADS = 256;
constexpr int C_R = 256;
constexpr int C_CB = 1024;
constexpr int C_ECAP = 12 * C_THREADS;     // 3072 entries (24 KB) per window

template <bool BWD, class Args>
__global__ __launch_bounds__(C_THREADS) void scalar_tiled_kernel(TiledDev t, Args a) {
    __shared__ float Xs[C_CB];              //  4 KB  staged x_j of the column block
    __shared__ int2 Es[C_ECAP];             // 24 KB  entry segment (window)
    __shared__ float4 S0[C_R];              //  4 KB  running state of the rows
    __shared__ float4 S1[C_R];              //  4 KB  per-row inputs
    __shared__ float4 S2[BWD ? C_R : 1];    //  4 KB  (backward only)
    __shared__ int Ps[C_R + 16];
    __shared__ int Pm[C_R];
    __shared__ int Sg[T_MAXB + 1];
    __shared__ int Bk[T_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], tb1 = t.tile_blk[tile + 1];
    const int row0 = tile * C_R;
    const int n_rows = min(C_R, t.n_dst - row0);
    (void)lane;

    if constexpr (BWD) {
        BwdDst1T::init_row(a, row0 + tid, tid < n_rows, &S0[tid], &S1[tid]);
        S2[tid] = BwdDst1T::init_row2(a, row0 + tid, tid < n_rows);
    } else {
        Fwd1T::init_row(a, row0 + tid, tid < n_rows, &S0[tid], &S1[tid]);
    }
    if (tid <= tb1 - tb0) Sg[tid] = t.ptr2[(size_t)(tb0 + tid) * C_R];
    if (tid < tb1 - tb0) Bk[tid] = t.blk_id[tb0 + tid];
    __syncthreads();

#define C_LDE(S, K) pe##S##K = t.ent[seg0##S + min(tid + K * C_THREADS, max(len##S - 1, 0))];
#define C_STE(S, K) Es[tid + K * C_THREADS] = pe##S##K;
#define C_PREFETCH(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbx_ = (TB);                                                                              \
        const int c0_ = Bk[tbx_ - tb0] * C_CB;                                                              \
        const int c4_ = (min(C_CB, t.n_src - c0_) + 3) / 4;                                                 \
        /* (the last block of x may end inside a float4: the tail is read element-wise) */                 \
        const float* src_ = a.X + c0_;                                                                      \
        const int i4_ = min(tid, c4_ - 1) * 4;                                                              \
        const int lim_ = t.n_src - c0_ - 1;                                                                 \
        px##S = make_float4(src_[min(i4_, lim_)], src_[min(i4_ + 1, lim_)], src_[min(i4_ + 2, lim_)],       \
                            src_[min(i4_ + 3, lim_)]);                                                      \
        seg0##S = Sg[tbx_ - tb0];                                                                           \
        len##S = Sg[tbx_ - tb0 + 1] - seg0##S;                                                              \
        pp##S = t.ptr2[(size_t)tbx_ * C_R + tid] - seg0##S;                                                 \
        pm##S = t.perm[(size_t)tbx_ * C_R + tid];                                                           \
        C_LDE(S, 0) C_LDE(S, 1) C_LDE(S, 2) C_LDE(S, 3) C_LDE(S, 4) C_LDE(S, 5)                             \
        C_LDE(S, 6) C_LDE(S, 7) C_LDE(S, 8) C_LDE(S, 9) C_LDE(S, 10) C_LDE(S, 11)                           \
    }
#define C_DO_BLOCK(S, TB)                                                                                   \
    {                                                                                                       \
        const int tbc_ = (TB);                                                                              \
        __syncthreads();                                                                                    \
        reinterpret_cast<float4*>(Xs)[tid] = px##S;                                                         \
        C_STE(S, 0) C_STE(S, 1) C_STE(S, 2) C_STE(S, 3) C_STE(S, 4) C_STE(S, 5)                             \
        C_STE(S, 6) C_STE(S, 7) C_STE(S, 8) C_STE(S, 9) C_STE(S, 10) C_STE(S, 11)                           \
        Ps[tid] = pp##S;                                                                                    \
        Pm[tid] = pm##S;                                                                                    \
        if (tid == 0) Ps[C_R] = len##S;                                                                     \
        const int cur_seg0 = seg0##S, cur_len = len##S;                                                     \
        __syncthreads();                                                                                    \
        const int tb_next = min(tbc_ + 2, tb1 - 1);                                                         \
        if (early) C_PREFETCH(S, tb_next)                                                                   \
        walk(cur_seg0, cur_len);                                                                            \
        if (!early) C_PREFETCH(S, tb_next)                                                                  \
    }

    auto walk = [&](int cur_seg0, int cur_len) {
        for (int w0 = 0; w0 < cur_len; w0 += C_ECAP) {
            if (w0 > 0) {   // rare: segment longer than one window
                __syncthreads();
                for (int i = tid; i < min(C_ECAP, cur_len - w0); i += C_THREADS) Es[i] = t.ent[cur_seg0 + w0 + i];
                __syncthreads();
            }
            const int w1 = w0 + C_ECAP;
            const int k = tid;                           // sorted position: wave w walks positions 64w .. 64w + 63
            const int s = max(Ps[k], w0), e = min(Ps[k + 1], w1);
            if (s < e) {
                const int rl = Pm[k];
                int p = s - w0;
                const int pe_ = e - w0;
                const char* xb = reinterpret_cast<const char*>(Xs);
                if constexpr (BWD) {
                    BwdDst1T op;
                    op.load(S0[rl], S1[rl], S2[rl]);
                    for (; p + 1 < pe_; p += 2) {
                        const int2 e0 = Es[p], e1 = Es[p + 1];
                        const float x0 = *reinterpret_cast<const float*>(xb + e0.x);
                        const float x1 = *reinterpret_cast<const float*>(xb + e1.x);
                        op.edge1(x0, __int_as_float(e0.y));
                        op.edge1(x1, __int_as_float(e1.y));
                    }
                    if (p < pe_) {
                        const int2 e0 = Es[p];
                        op.edge1(*reinterpret_cast<const float*>(xb + e0.x), __int_as_float(e0.y));
                    }
                    op.store(&S0[rl]);
                } else {
                    Fwd1T op;
                    op.load(S0[rl], S1[rl]);
                    for (; p + 1 < pe_; p += 2) {
                        const int2 e0 = Es[p], e1 = Es[p + 1];
                        const float x0 = *reinterpret_cast<const float*>(xb + e0.x);
                        const float x1 = *reinterpret_cast<const float*>(xb + e1.x);
                        op.edge2(x0, __int_as_float(e0.y), x1, __int_as_float(e1.y));
                    }
                    if (p < pe_) {
                        const int2 e0 = Es[p];
                        op.edge1(*reinterpret_cast<const float*>(xb + e0.x), __int_as_float(e0.y));
                    }
                    op.store(&S0[rl]);
                }
            }
        }
    };

    float4 pxA, pxB;
    int2 peA0, peA1, peA2, peA3, peA4, peA5, peA6, peA7, peA8, peA9, peA10, peA11;
    int2 peB0, peB1, peB2, peB3, peB4, peB5, peB6, peB7, peB8, peB9, peB10, peB11;
    int ppA = 0, pmA = 0, seg0A = 0, lenA = 0, ppB = 0, pmB = 0, seg0B = 0, lenB = 0;
    const bool early = wave < C_THREADS / 128;
    if (tb0 < tb1) {
        C_PREFETCH(A, tb0)
        C_PREFETCH(B, min(tb0 + 1, tb1 - 1))
    }
    for (int tb = tb0; tb < tb1; tb += 2) {
        C_DO_BLOCK(A, tb)
        if (tb + 1 < tb1) C_DO_BLOCK(B, tb + 1)
    }
#undef C_LDE
#undef C_STE
#undef C_PREFETCH
#undef C_DO_BLOCK
    __syncthreads();
    if (tid < n_rows) {
        if constexpr (BWD) BwdDst1T::epilogue(a, row0 + tid, S0[tid]);
        else Fwd1T::epilogue(a, row0 + tid, S0[tid], S1[tid]);
    }
}

static TiledDev tiled_dev(const Tiled& tl, int n_dst, int n_src) {
    TiledDev d;
    d.tile_blk = tl.tile_blk; d.blk_id = tl.blk_id; d.ptr2 = tl.ptr2; d.perm = tl.perm;
    d.ent = reinterpret_cast<const int2*>(tl.ent);
    d.n_tiles = tl.n_tiles; d.n_dst = n_dst; d.n_src = n_src;
    return d;
}

int launch_fwd1_tiled(const Tiled& tl, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                      const float* x_src, const float* x_dst, float* h_out, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    Fwd1TiledArgs a;
    a.X = x_src; a.xd = x_dst; a.derived = w.derived; a.p = conv_params_at(conv_params, 1);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    hipLaunchKernelGGL((scalar_tiled_kernel<false, Fwd1TiledArgs>), dim3((unsigned)tl.n_tiles), dim3(C_THREADS), 0, s,
                       tiled_dev(tl, n_dst, n_src), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "fwd1_tiled");
}

int launch_bwddst1_tiled(const Tiled& tl, int n_dst, int n_src, const ConvWs& w, const float* x_src, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    BwdDst1TiledArgs a{x_src, w.rec, w.dqp, w.dsdt};
    hipLaunchKernelGGL((scalar_tiled_kernel<true, BwdDst1TiledArgs>), dim3((unsigned)tl.n_tiles), dim3(C_THREADS), 0, s,
                       tiled_dev(tl, n_dst, n_src), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwddst1_tiled");
}

// The product library has exactly one SpMM path and reads no environment variables.  `make timing` builds
// libmllp_hip_timing.so with -DMLLP_TIMING_BUILD: the same kernel with its timing-only instantiations (results are
// WRONG there when the mask is not 0: 4 = no walk, 16 = cycle counters instead of Y), selected by MLLP_TILED_ABLATION
// and used only by tools/phase_cycles.py.
#ifdef MLLP_TIMING_BUILD
static int g_tiled_ablation = [] {
    const char* e = getenv("MLLP_TILED_ABLATION");
    return e ? atoi(e) : 0;
}();
#endif

int launch_spmm_tiled(const Tiled& tl, int n_dst, int n_src, const float* H, float* Y, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    const TiledDev d = tiled_dev(tl, n_dst, n_src);
#ifdef MLLP_TIMING_BUILD
    switch (g_tiled_ablation) {
        case 0: hipLaunchKernelGGL(spmm_tiled_ws_kernel<0>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 4: hipLaunchKernelGGL(spmm_tiled_ws_kernel<4>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        case 16: hipLaunchKernelGGL(spmm_tiled_ws_kernel<16>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y); break;
        default: return fail(MLLP_EINVAL, "unknown ablation mask");
    }
#else
    hipLaunchKernelGGL(spmm_tiled_ws_kernel<0>, dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d, H, Y);
#endif
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "spmm_tiled");
}

// H: bf16 [n_src, 16]
int launch_spmm_tiled_bf16(const Tiled& tl, int n_dst, int n_src, const void* H, float* Y, hipStream_t s) {
    if (tl.n_tiles == 0) return MLLP_OK;
    const TiledDev d = tiled_dev(tl, n_dst, n_src);
    hipLaunchKernelGGL((spmm_tiled_ws_kernel<0, true>), dim3((unsigned)tl.n_tiles), dim3(T_THREADS), 0, s, d,
                       reinterpret_cast<const float*>(H), Y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "spmm_tiled_ws");
}

int tiled_max_blocks_per_tile() { return T_MAXB; }

int tiled_geometry(int variant, int* rows_per_tile, int* cols_per_block, int* bundle_capacity) {
    if (variant == 0) {          // plain SpMM (both kernels)
        *rows_per_tile = T_R; *cols_per_block = T_CB; *bundle_capacity = T_ECAP;
    } else if (variant == 1) {   // attention forward sweep
        *rows_per_tile = F_R; *cols_per_block = F_CB; *bundle_capacity = F_ECAP;
    } else if (variant == 2) {   // attention backward, source-major: columns are 160-byte records
        *rows_per_tile = S_R; *cols_per_block = S_CB; *bundle_capacity = S_ECAP;
    } else if (variant == 3) {   // layer-1 (scalar) attention sweeps, destination-major
        *rows_per_tile = C_R; *cols_per_block = C_CB; *bundle_capacity = C_ECAP;
    } else {                     // attention backward, destination-major
        *rows_per_tile = D_R; *cols_per_block = D_CB; *bundle_capacity = D_ECAP;
    }
    return MLLP_OK;
}

}  // namespace mllp
