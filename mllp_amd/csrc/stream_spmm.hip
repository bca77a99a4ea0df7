// stream_spmm.hip -- the roofline kernel of round 3: Y = A H (16 fp32 channels) on the STREAMED copy of an orientation
// (layout, geometry and the reasons: stream_layout.h).  The reference's equivalent is the gather / scale / scatter-add
// of PyG's message passing over `edge_index`, `edge_attr` (linear_program_methods.py:241-247).
//
// Workgroup = one row tile (at most 1024 rows of one instance) = 8 wavefronts on one CU, all with the same role:
//   prefetch   at the top of block b: this wavefront's share (5 pieces of 1 KB) of the NEXT block's 625 x 64 B of H
//              by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) into the image that is not being read,
//              and its own entries of block b + 1 (S_K groups of 2 steps, coalesced 1 KB non-temporal loads) into the
//              second register set: the entries never touch LDS.
//   walk       two passes over its entries of block b, four rows per quad: per step and quad one entry of each row,
//              shared inside the quad by DPP, one ds_read_b128 of the source row per entry, two packed FMAs per lane.
//              The steps are software-pipelined by hand (reads of step s + 2 are issued before the FMAs of step s:
//              up to 12 ds_read_b128 in flight per wavefront); the accumulators of the four rows are read from /
//              written back to LDS once per pass.
//   barrier    one per block: behind it image b + 1 and everybody's entries of block b + 1 have landed (vmcnt(0) on
//              every wavefront, MI355X_MICROARCH.md "Two waves per SIMD" item 7) and the reads of image b are done.
//   LDS        image 0 | image 1 (40 000 B + one all-zero row each) | accumulators of the tile (64 KB).
// The LDS-DMA is issued from inline asm: hipcc orders every later LDS read behind a visible global_load_lds
// (s_waitcnt vmcnt(0) in front of each ds_read: the destination may alias), which would serialise prefetch and walk.
// Deterministic: a row of a (tile, block) belongs to one quad, blocks are walked in order, no atomics.
#include <cstdlib>

#include "device_utils.h"
#include "internal.h"
#include "stream_layout.h"

namespace mllp {

constexpr int SK_THREADS = 64 * S_NW;
constexpr int SK_IMG = S_CB * S_ROW_BYTES + S_ROW_BYTES;      // image + the all-zero row
constexpr int SK_YA = 2 * SK_IMG;                             // accumulators behind the two images
constexpr int SK_LDS = SK_YA + S_R * S_ROW_BYTES;
constexpr int SK_PIECES = (S_CB * S_ROW_BYTES + 1023) / 1024; // LDS-DMA pieces of 1 KB per image
constexpr int SK_PPW = (SK_PIECES + S_NW - 1) / S_NW;         // pieces per wavefront
constexpr int SK_STEPS = 2 * S_K;                             // steps held in one register set
static_assert(SK_LDS <= 163840, "LDS of one CU");

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

struct StreamDev {
    const int* __restrict__ tile_row;
    const int* __restrict__ tile_blk;
    const int* __restrict__ blk_id;
    const i32x4* __restrict__ rows;
    const i32x4* __restrict__ hdr;
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
// 16 bytes per lane, global -> LDS at (wave-uniform) lds_dst + 16 * lane, without the compiler's knowledge
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
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
    const int row0 = t.tile_row[tile];
    const int n4 = (t.tile_row[tile + 1] - row0) * 4;
    float4* dst = reinterpret_cast<float4*>(Y + (size_t)row0 * 16);
    if (nb == 0) {      // every row of the tile is empty
        for (int i = tid; i < n4; i += SK_THREADS) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float4* Ya = reinterpret_cast<float4*>(smem + SK_YA);
    for (int i = tid; i < S_R * 4; i += SK_THREADS) Ya[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 8) *reinterpret_cast<float4*>(smem + (tid >> 2) * SK_IMG + S_ZERO_OFF + (tid & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    unsigned cyc[4] = {0u, 0u, 0u, 0u};
    unsigned last_ = (ABL & 16) ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const unsigned start_ = last_;
#define SK_TICK(K)                                                                                          \
    if (ABL & 16) {                                                                                         \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();                                       \
        cyc[K] += now_ - last_;                                                                             \
        last_ = now_;                                                                                       \
    }

    // this wavefront's pieces of the H rows of column block `blk` -> image `img`
    auto stage = [&](int blk, int img) {
        const int c0 = __builtin_amdgcn_readfirstlane(blk) * S_CB;
        const int nbytes = min(S_CB, t.n_src - c0) * S_ROW_BYTES;
        const char* src = reinterpret_cast<const char*>(X + (size_t)c0 * 16) + lane * 16;
        const unsigned img_base = lds0 + (unsigned)(img * SK_IMG);
#pragma unroll
        for (int i = 0; i < SK_PPW; ++i) {
            const int piece = wave + S_NW * i;
            if (piece * 1024 + lane * 16 < nbytes && !(ABL & 2))
                glds16(src + piece * 1024, __builtin_amdgcn_readfirstlane(img_base + piece * 1024));
        }
    };

    const int quad = lane >> 2, part = lane & 3;
    const i32x4* rowp = t.rows + ((size_t)tb0 * S_NW + wave) * 16 + quad;     // + 128 per block
    const i32x4* hdrp = t.hdr + (size_t)tb0 * S_NW + wave;                    // + 8 per block
    i32x4 rc = __builtin_nontemporal_load(rowp), hc = __builtin_nontemporal_load(hdrp);
    stage(hc.z, 0);
    i32x4 rn = __builtin_nontemporal_load(rowp + 128 * min(1, nb - 1));
    i32x4 hn = __builtin_nontemporal_load(hdrp + 8 * min(1, nb - 1));
    i32x4 eX[S_K], eY[S_K];
    auto loadset = [&](i32x4 (&buf)[S_K], int first_group) {
        const i32x4* p = t.ent + (size_t)first_group * 64 + lane;
#pragma unroll
        for (int j = 0; j < S_K; ++j) buf[j] = __builtin_nontemporal_load(p + 64 * j);
    };
    loadset(eX, __builtin_amdgcn_readfirstlane(hc.x) >> 1);

    // the four source rows of one step (lane = 4 channels of the quad's four rows; `o` = this lane's entry offset)
    auto issue = [&](int o, unsigned pb, float4 (&x)[4]) {
        x[0] = *reinterpret_cast<const float4*>(smem + (pb + qbcast<0>(o)));
        x[1] = *reinterpret_cast<const float4*>(smem + (pb + qbcast<1>(o)));
        x[2] = *reinterpret_cast<const float4*>(smem + (pb + qbcast<2>(o)));
        x[3] = *reinterpret_cast<const float4*>(smem + (pb + qbcast<3>(o)));
    };
    auto fma = [&](int v, const float4 (&x)[4], f32x2s (&acc)[8]) {
        pk4(__int_as_float(qbcast<0>(v)), x[0], acc[0], acc[1]);
        pk4(__int_as_float(qbcast<1>(v)), x[1], acc[2], acc[3]);
        pk4(__int_as_float(qbcast<2>(v)), x[2], acc[4], acc[5]);
        pk4(__int_as_float(qbcast<3>(v)), x[3], acc[6], acc[7]);
    };
    // one pass: steps [a, b) (absolute) of the quad's rows r01 = row0 | row1 << 16, r23; `cur` holds the SK_STEPS steps
    // from 2 * g0 on.  Steps beyond them (a row with hundreds of entries inside one block) are fetched group by
    // group: slow path, kept apart so that the common path never waits for a load inside the walk.
    auto pass = [&](const i32x4 (&cur)[S_K], int g0, int r01, int r23, int a, int b, unsigned pb) {
        if (a >= b || (ABL & 1)) return;
        const int i0 = (r01 & 0xffff) * 4 + part, i1 = ((r01 >> 16) & 0xffff) * 4 + part;
        const int i2 = (r23 & 0xffff) * 4 + part, i3 = ((r23 >> 16) & 0xffff) * 4 + part;
        const float4 y0 = Ya[i0], y1 = Ya[i1], y2 = Ya[i2], y3 = Ya[i3];
        f32x2s acc[8] = {{y0.x, y0.y}, {y0.z, y0.w}, {y1.x, y1.y}, {y1.z, y1.w},
                         {y2.x, y2.y}, {y2.z, y2.w}, {y3.x, y3.y}, {y3.z, y3.w}};
        const int ra = a - 2 * g0, rb = b - 2 * g0;
        float4 x[3][4];
#pragma unroll
        for (int s = 0; s < SK_STEPS + 2; ++s) {
            if (s < SK_STEPS && s >= ra && s < rb)                          // wave-uniform
                issue((s & 1) ? cur[s >> 1].z : cur[s >> 1].x, pb, x[s % 3]);
            const int f = s - 2;
            if (f >= 0 && f >= ra && f < rb) fma((f & 1) ? cur[f >> 1].w : cur[f >> 1].y, x[f % 3], acc);
        }
        for (int st = max(ra, SK_STEPS); st < rb; ++st) {
            const i32x4 e = __builtin_nontemporal_load(t.ent + (size_t)(g0 + (st >> 1)) * 64 + lane);
            float4 xs[4];
            issue((st & 1) ? e.z : e.x, pb, xs);
            fma((st & 1) ? e.w : e.y, xs, acc);
        }
        Ya[i0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
        Ya[i1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        Ya[i2] = make_float4(acc[4].x, acc[4].y, acc[5].x, acc[5].y);
        Ya[i3] = make_float4(acc[6].x, acc[6].y, acc[7].x, acc[7].y);
    };
    SK_TICK(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): image 0 and the first entries have landed
    __syncthreads();
    SK_TICK(1)
#define SK_ITER(CUR, NXT, KK)                                                                               \
    {                                                                                                       \
        if ((KK) + 1 < nb) stage(hn.z, ((KK) + 1) & 1);                                                   \
        const int nx_ = min((KK) + 2, nb - 1);                                                              \
        const i32x4 r2 = __builtin_nontemporal_load(rowp + 128 * nx_);                                      \
        const i32x4 h2 = __builtin_nontemporal_load(hdrp + 8 * nx_);                                        \
        loadset(NXT, __builtin_amdgcn_readfirstlane(hn.x) >> 1);                                            \
        SK_TICK(3)                                                                                          \
        const int S_ = __builtin_amdgcn_readfirstlane(hc.x);                                                \
        const unsigned c_ = (unsigned)__builtin_amdgcn_readfirstlane(hc.y);                                 \
        const int n0_ = (int)(c_ & 0xffffu), n1_ = (int)(c_ >> 16), g0_ = S_ >> 1;                          \
        const unsigned pb_ = (unsigned)(((KK) & 1) * SK_IMG + part * 16);                                   \
        pass(CUR, g0_, rc.x, rc.y, S_, S_ + n0_, pb_);                                                      \
        pass(CUR, g0_, rc.z, rc.w, S_ + n0_, S_ + n0_ + n1_, pb_);                                          \
        SK_TICK(2)                                                                                          \
        __builtin_amdgcn_s_waitcnt(0x0F70);  /* vmcnt(0): image and entries of the next block have landed */ \
        rc = rn; hc = hn;                                                                                   \
        rn = r2; hn = h2;                                                                                   \
        SK_TICK(0)                                                                                          \
        __syncthreads();                                                                                    \
        SK_TICK(1)                                                                                          \
    }
    for (int k = 0; k < nb; k += 2) {
        SK_ITER(eX, eY, k)
        if (k + 1 >= nb) break;
        SK_ITER(eY, eX, k + 1)
    }
#undef SK_ITER
    if (ABL & 16) {
        // stamps instead of the result, summed over the wavefronts: row 2 * tile = {[0] wait for the loads to land,
        // [1] wait at the barrier, [2] walk, [3] prefetch issue, [4] total}
        const unsigned total_ = (unsigned)__builtin_amdgcn_s_memtime() - start_;
        __syncthreads();
        int* acc = reinterpret_cast<int*>(smem);
        if (tid < 16) acc[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) atomicAdd(&acc[k], (int)cyc[k]);
            atomicAdd(&acc[4], (int)total_);
        }
        __syncthreads();
        if (tid < 16 && 2 * tile + 1 < t.n_dst) Y[(size_t)(2 * tile) * 16 + tid] = (float)acc[tid];
        return;
    }
#undef SK_TICK
    // (the last barrier of the loop made every wavefront's accumulators visible)
    for (int i = tid; i < n4; i += SK_THREADS) dst[i] = Ya[i];
}

int launch_spmm_stream(const StreamCopy& sc, int n_dst, int n_src, const float* H, float* Y, hipStream_t s) {
    if (n_dst == 0) return MLLP_OK;
    StreamDev t;
    t.tile_row = sc.tile_row;
    t.tile_blk = sc.tile_blk;
    t.blk_id = sc.blk_id;
    t.rows = reinterpret_cast<const i32x4*>(sc.rows);
    t.hdr = reinterpret_cast<const i32x4*>(sc.hdr);
    t.ent = reinterpret_cast<const i32x4*>(sc.ent);
    t.n_tiles = sc.n_tiles;
    t.n_dst = n_dst;
    t.n_src = n_src;
    int abl = 0;
#ifdef MLLP_TIMING_BUILD
    if (const char* e = getenv("MLLP_STREAM_ABLATION")) abl = atoi(e);
    if (abl == 1) hipLaunchKernelGGL(spmm_stream_kernel<1>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 3) hipLaunchKernelGGL(spmm_stream_kernel<3>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 2) hipLaunchKernelGGL(spmm_stream_kernel<2>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 16) hipLaunchKernelGGL(spmm_stream_kernel<16>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
#endif
    if (abl == 0) hipLaunchKernelGGL(spmm_stream_kernel<0>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    MLLP_HIP_TRY(hipGetLastError());
    return MLLP_OK;
}

}  // namespace mllp
