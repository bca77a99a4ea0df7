// stream_spmm.hip -- the roofline kernel of round 3: Y = A H (16 fp32 channels) on the STREAMED copy of an orientation
// (layout, geometry and the reasons: stream_layout.h).  The reference's equivalent is the gather / scale / scatter-add
// of PyG's message passing over `edge_index`, `edge_attr` (linear_program_methods.py:241-247).
//
// Workgroup = one 512-row tile = 768 threads on one CU:
//   wavefronts 0..7   WALKERS.  Each holds, in registers, its own entries of the current (tile, block) (S_K groups of
//                     4 steps, loaded one block ahead with coalesced 1 KB non-temporal loads: the entries never touch
//                     LDS) and walks them: per step and quad one entry of row A and one of row B, shared inside the
//                     quad by DPP, one ds_read_b128 of the source row per entry, two packed FMAs per lane.
//                     The accumulators of the two rows are read from / written back to LDS once per pass.
//   wavefronts 8..11  STAGERS.  global_load_lds_dwordx4 (LDS-DMA: no registers, no ds_write) of the NEXT block's
//                     1000 x 64 B of H into the image the walkers are not reading.
//   LDS               image 0 | image 1 (64 000 B + one all-zero row each) | accumulators of the tile (32 KB).
//   one barrier per block: behind it image b+1 and every walker's entries of block b+1 have landed (vmcnt(0) on the
//                     issuing wavefronts, MI355X_MICROARCH.md "Two waves per SIMD" item 7) and the walkers' reads of
//                     image b are done, so block b+2 may overwrite it.
// Deterministic: a row of a (tile, block) belongs to one quad, blocks are walked in order, no atomics.
#include <cstdlib>

#include "device_utils.h"
#include "internal.h"
#include "stream_layout.h"

namespace mllp {

constexpr int SK_THREADS = 768;
constexpr int SK_STAGERS = SK_THREADS / 64 - S_NW;           // 4
constexpr int SK_IMG = S_CB * S_ROW_BYTES + S_ROW_BYTES;      // 64 064: image + the all-zero row
constexpr int SK_YA = 2 * SK_IMG;                             // accumulators behind the two images
constexpr int SK_LDS = SK_YA + S_R * S_ROW_BYTES;             // 160 896
constexpr int SK_PIECES = (S_CB * S_ROW_BYTES + 1023) / 1024; // 63 LDS-DMA pieces of 1 KB per image
static_assert(SK_LDS <= 163840, "LDS of one CU");
static_assert(SK_PIECES <= 16 * SK_STAGERS, "pieces per stager");

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

struct StreamDev {
    const int* __restrict__ tile_blk;
    const int* __restrict__ blk_id;
    const i32x4* __restrict__ rec;
    const i32x4* __restrict__ ent;
    int n_tiles, n_dst, n_src;
};

template <int K>
__device__ __forceinline__ int qbcast(int v) {
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ void pk4(float v, const float4& x, f32x2s& lo, f32x2s& hi) {
    const f32x2s vv = {v, v};
    lo = __builtin_elementwise_fma(vv, f32x2s{x.x, x.y}, lo);
    hi = __builtin_elementwise_fma(vv, f32x2s{x.z, x.w}, hi);
}

// ABL (timing build only): 1 = no walk, 2 = no staging, 16 = cycle stamps instead of the result
template <int ABL>
__global__ __launch_bounds__(SK_THREADS) void spmm_stream_kernel(StreamDev t, const float* __restrict__ X,
                                                                 float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) char smem[SK_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], nb = t.tile_blk[tile + 1] - tb0;
    const int row0 = tile * S_R;
    const int n4 = min(S_R, t.n_dst - row0) * 4;
    float4* dst = reinterpret_cast<float4*>(Y + (size_t)row0 * 16);
    if (nb == 0) {      // every row of the tile is empty
        for (int i = tid; i < n4; i += SK_THREADS) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float4* Ya = reinterpret_cast<float4*>(smem + SK_YA);
    for (int i = tid; i < S_R * 4; i += SK_THREADS) Ya[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 8) *reinterpret_cast<float4*>(smem + (tid >> 2) * SK_IMG + S_ZERO_OFF + (tid & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    unsigned cyc[4] = {0u, 0u, 0u, 0u};
    unsigned last_ = (ABL & 16) ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const unsigned start_ = last_;
#define SK_TICK(K)                                                                                          \
    if (ABL & 16) {                                                                                         \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();                                       \
        cyc[K] += now_ - last_;                                                                             \
        last_ = now_;                                                                                       \
    }

    if (wave >= S_NW) {
        // ------------------------------------------------ stagers ------------------------------------
        const int d = wave - S_NW;
        auto stage = [&](int b, int img) {
            const int c0 = __builtin_amdgcn_readfirstlane(t.blk_id[tb0 + b]) * S_CB;
            const int nbytes = min(S_CB, t.n_src - c0) * S_ROW_BYTES;
            const char* src = reinterpret_cast<const char*>(X + (size_t)c0 * 16) + lane * 16;
            char* img_base = smem + img * SK_IMG;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int piece = d + SK_STAGERS * i;
                if (piece * 1024 + lane * 16 < nbytes && !(ABL & 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + piece * 1024),
                        (__attribute__((address_space(3))) void*)(img_base + piece * 1024), 16, 0, 0);
            }
        };
        stage(0, 0);
        SK_TICK(0)
        __syncthreads();
        SK_TICK(1)
        for (int k = 0; k < nb; ++k) {
            if (k + 1 < nb) stage(k + 1, (k + 1) & 1);
            SK_TICK(0)
            __syncthreads();
            SK_TICK(1)
        }
    } else {
        // ------------------------------------------------ walkers ------------------------------------
        const int quad = lane >> 2, part = lane & 3;
        const i32x4* recp = t.rec + ((size_t)tb0 * S_NW + wave) * 16 + quad;      // + 128 per block
        i32x4 rc = __builtin_nontemporal_load(recp);
        i32x4 rn = __builtin_nontemporal_load(recp + 128 * min(1, nb - 1));
        i32x4 eX[S_K], eY[S_K];
        auto loadset = [&](i32x4 (&buf)[S_K], int first_group) {
            const i32x4* p = t.ent + (size_t)first_group * 64 + lane;
#pragma unroll
            for (int j = 0; j < S_K; ++j) buf[j] = __builtin_nontemporal_load(p + 64 * j);
        };
        loadset(eX, __builtin_amdgcn_readfirstlane(rc.z) >> 2);

        // one group of four steps: the lanes of a quad hold steps 0..3 of the group; `v` = this lane's step is inside
        // the pass (steps of the neighbouring pass / block in the same group are replaced by padding entries)
        auto group4 = [&](const i32x4& e, bool v, unsigned pb, f32x2s& a0, f32x2s& a1, f32x2s& b0, f32x2s& b1) {
            const int oA = v ? e.x : S_ZERO_OFF, vA = v ? e.y : 0;
            const int oB = v ? e.z : S_ZERO_OFF, vB = v ? e.w : 0;
            const float4 x0 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<0>(oA)));
            const float4 x1 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<1>(oA)));
            const float4 x2 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<2>(oA)));
            const float4 x3 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<3>(oA)));
            const float4 y0 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<0>(oB)));
            const float4 y1 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<1>(oB)));
            const float4 y2 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<2>(oB)));
            const float4 y3 = *reinterpret_cast<const float4*>(smem + (pb + qbcast<3>(oB)));
            pk4(__int_as_float(qbcast<0>(vA)), x0, a0, a1);
            pk4(__int_as_float(qbcast<1>(vA)), x1, a0, a1);
            pk4(__int_as_float(qbcast<2>(vA)), x2, a0, a1);
            pk4(__int_as_float(qbcast<3>(vA)), x3, a0, a1);
            pk4(__int_as_float(qbcast<0>(vB)), y0, b0, b1);
            pk4(__int_as_float(qbcast<1>(vB)), y1, b0, b1);
            pk4(__int_as_float(qbcast<2>(vB)), y2, b0, b1);
            pk4(__int_as_float(qbcast<3>(vB)), y3, b0, b1);
        };
        // one pass: steps [a, b) (absolute) of rows `rows` = A | B << 16; `cur` holds the S_K groups from g0 on.
        // Steps beyond them (a row with hundreds of entries inside one block) are fetched group by group: slow path,
        // kept apart so that the common path never waits for a load inside the walk.
        auto pass = [&](const i32x4 (&cur)[S_K], int g0, int rows, int a, int b, unsigned pb) {
            if (a >= b || (ABL & 1)) return;
            const int rA = (rows & 0xffff) * 4 + part, rB = ((rows >> 16) & 0xffff) * 4 + part;
            const float4 ya = Ya[rA], yb = Ya[rB];
            f32x2s a0 = {ya.x, ya.y}, a1 = {ya.z, ya.w}, b0 = {yb.x, yb.y}, b1 = {yb.z, yb.w};
            const int ra = a - 4 * g0, rb = b - 4 * g0;
#pragma unroll
            for (int j = 0; j < S_K; ++j) {
                if (4 * j + 4 <= ra || 4 * j >= rb) continue;       // wave-uniform
                const int st = 4 * j + part;
                group4(cur[j], st >= ra && st < rb, pb, a0, a1, b0, b1);
            }
            for (int g = max(ra >> 2, S_K); 4 * g < rb; ++g) {
                const i32x4 e = __builtin_nontemporal_load(t.ent + (size_t)(g0 + g) * 64 + lane);
                const int st = 4 * g + part;
                group4(e, st >= ra && st < rb, pb, a0, a1, b0, b1);
            }
            Ya[rA] = make_float4(a0.x, a0.y, a1.x, a1.y);
            Ya[rB] = make_float4(b0.x, b0.y, b1.x, b1.y);
        };
        SK_TICK(0)
        __syncthreads();
        SK_TICK(1)
#define SK_ITER(CUR, NXT, KK)                                                                               \
    {                                                                                                       \
        const i32x4 r2 = __builtin_nontemporal_load(recp + 128 * min((KK) + 2, nb - 1));                    \
        loadset(NXT, __builtin_amdgcn_readfirstlane(rn.z) >> 2);                                            \
        const int S_ = __builtin_amdgcn_readfirstlane(rc.z), c_ = __builtin_amdgcn_readfirstlane(rc.w);     \
        const int n0_ = c_ & 0xffff, n1_ = (int)((unsigned)c_ >> 16);                                       \
        const int g0_ = S_ >> 2;                                                                            \
        const unsigned pb_ = (unsigned)(((KK) & 1) * SK_IMG + part * 16);                                   \
        pass(CUR, g0_, rc.x, S_, S_ + n0_, pb_);                                                            \
        pass(CUR, g0_, rc.y, S_ + n0_, S_ + n0_ + n1_, pb_);                                                \
        SK_TICK(2)                                                                                          \
        rc = rn;                                                                                            \
        rn = r2;                                                                                            \
        __syncthreads();                                                                                    \
        SK_TICK(1)                                                                                          \
    }
        for (int k = 0; k < nb; k += 2) {
            SK_ITER(eX, eY, k)
            if (k + 1 >= nb) break;
            SK_ITER(eY, eX, k + 1)
        }
#undef SK_ITER
    }
    if (ABL & 16) {
        // stamps instead of the result: row 2*tile = walkers {[0] prologue, [1] barrier wait, [2] walk, [3] total},
        // row 2*tile+1 = stagers {[0] issue + landing wait, [1] barrier wait, .., [3] total}, summed over wavefronts
        const unsigned total_ = (unsigned)__builtin_amdgcn_s_memtime() - start_;
        __syncthreads();
        int* acc = reinterpret_cast<int*>(smem);
        if (tid < 32) acc[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            const int o = wave >= S_NW ? 16 : 0;
            for (int k = 0; k < 3; ++k) atomicAdd(&acc[o + k], (int)cyc[k]);
            atomicAdd(&acc[o + 3], (int)total_);
        }
        __syncthreads();
        if (tid < 32 && 2 * tile + 1 < t.n_dst) Y[(size_t)(2 * tile) * 16 + tid] = (float)acc[tid];
        return;
    }
#undef SK_TICK
    // (the last barrier of the loop made every walker's accumulators visible)
    for (int i = tid; i < n4; i += SK_THREADS) dst[i] = Ya[i];
}

int launch_spmm_stream(const StreamCopy& sc, int n_dst, int n_src, const float* H, float* Y, hipStream_t s) {
    if (n_dst == 0) return MLLP_OK;
    StreamDev t;
    t.tile_blk = sc.tile_blk;
    t.blk_id = sc.blk_id;
    t.rec = reinterpret_cast<const i32x4*>(sc.rec);
    t.ent = reinterpret_cast<const i32x4*>(sc.ent);
    t.n_tiles = sc.n_tiles;
    t.n_dst = n_dst;
    t.n_src = n_src;
    int abl = 0;
#ifdef MLLP_TIMING_BUILD
    if (const char* e = getenv("MLLP_STREAM_ABLATION")) abl = atoi(e);
    if (abl == 1) hipLaunchKernelGGL(spmm_stream_kernel<1>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 2) hipLaunchKernelGGL(spmm_stream_kernel<2>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 16) hipLaunchKernelGGL(spmm_stream_kernel<16>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
#endif
    if (abl == 0) hipLaunchKernelGGL(spmm_stream_kernel<0>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    MLLP_HIP_TRY(hipGetLastError());
    return MLLP_OK;
}

}  // namespace mllp
