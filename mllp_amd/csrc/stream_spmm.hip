// stream_spmm.hip -- the roofline kernel of round 3: Y = A H (16 fp32 channels) on the STREAMED copy of an orientation
// (layout, geometry and the reasons: stream_layout.h).  The reference's equivalent is the gather / scale / scatter-add
// of PyG's message passing over `edge_index`, `edge_attr` (linear_program_methods.py:241-247).
//
// Workgroup = one row tile (at most 1024 rows of one instance) = 768 threads on one CU:
//   wavefronts 0..7   WALKERS.  Each holds, in registers, its own entries of the current (tile, block): one register
//              set per pass (S_K0 / S_K1 groups of 2 steps, 12 bytes per lane and group); a group is reloaded with the
//              NEXT block's entries (coalesced non-temporal loads, one per site) as soon as its last step is done, so
//              one set serves both blocks and the entries never touch LDS.  Two passes per block, four rows per quad:
//              per step and quad one entry of each row, shared inside the quad by DPP, one ds_read_b128 of the source
//              row per entry, two packed FMAs per lane.  The steps are static code, software-pipelined by hand (the
//              reads of step s + SK_D are issued before the FMAs of step s: 4 (SK_D + 1) ds_read_b128 in flight per
//              wavefront); the accumulators of the four rows are read from / written back to LDS once per pass.
//   wavefronts 8..11  STAGERS.  global_load_lds_dwordx4 (LDS-DMA: no registers, no ds_write) of the NEXT block's
//              625 x 64 B of H into the image the walkers are not reading, paced with s_sleep: a burst would fill the
//              CU's vector-memory queue and block the walkers where they issue their own loads.
//   barrier    one per block: behind it image b + 1 has landed (vmcnt(0) on the stagers, MI355X_MICROARCH.md "Two
//              waves per SIMD" item 7) and the reads of image b are done.
//   LDS        image 0 | image 1 (40 000 B + one all-zero row each) | accumulators of the tile (64 KB).
// (An 8-wavefront version in which every wavefront walked AND staged was slower, 1.77 vs 1.50 ms: all eight stood in
// the issue of their 19 loads for 3 200 cycles per block, profiles/r03_stream_experiments.txt.)
// Deterministic: a row of a (tile, block) belongs to one quad, blocks are walked in order, no atomics.
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "internal.h"
#include "stream_layout.h"

namespace mllp {

constexpr int SK_STAGERS = 4;
constexpr int SK_THREADS = 64 * (S_NW + SK_STAGERS);
constexpr int SK_IMG = S_CB * S_ROW_BYTES + S_ROW_BYTES;      // image + the all-zero row
constexpr int SK_YA = 2 * SK_IMG;                             // accumulators behind the two images
constexpr int SK_LDS = SK_YA + S_R * S_ROW_BYTES;
constexpr int SK_PIECES = (S_CB * S_ROW_BYTES + 1023) / 1024; // LDS-DMA pieces of 1 KB per image
constexpr int SK_PPS = (SK_PIECES + SK_STAGERS - 1) / SK_STAGERS;   // pieces per stager
static_assert(SK_LDS + 256 <= 163840, "LDS of one CU");

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) Ent3 {      // one lane's share of a group: two steps of its row slot
    int o;        // byte offset of the source row of step 2 g | of step 2 g + 1 << 16
    int v0, v1;   // value bits
};
struct StreamDev {
    const int* __restrict__ tile_row;
    const int* __restrict__ tile_blk;
    const i32x4* __restrict__ rows;
    const i32x4* __restrict__ hdr;
    const Ent3* __restrict__ ent;
    int n_tiles, n_dst, n_src;
};

template <int K>
__device__ __forceinline__ int qbcast(int v) {
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
template <class V4>
__device__ __forceinline__ void pk4(float v, const V4& x, f32x2s& lo, f32x2s& hi) {
    const f32x2s vv = {v, v};
    lo = __builtin_elementwise_fma(vv, f32x2s{x.x, x.y}, lo);
    hi = __builtin_elementwise_fma(vv, f32x2s{x.z, x.w}, hi);
}
// LDS reads of the walk and their waits, by hand.  hipcc's own s_waitcnt insertion merges the LDS counter state of the
// conditionally executed sites into "wait for everything" (lgkmcnt(0) in front of every group of reads and of every
// group of FMAs: no read was ever in flight beside an FMA, profiles/r03_stream_experiments.txt).  A read issued from
// asm is invisible to that pass; every consumer of its result is made data-dependent on the matching counted wait
// ("+v" ties), which is the only thing that orders it behind the return of the data.
__device__ __forceinline__ f32x4s lds_read16(unsigned addr) {
    f32x4s v;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
// four 4-byte LDS writes (a lane's own word of a dummy region) that only keep the count of outstanding LDS operations
// uniform.  Writes, not reads: an asm read's destination register is written when the data returns, long after the
// statement, and nothing would keep the compiler from reusing a dummy destination in between.
__device__ __forceinline__ void lds_pad4(unsigned addr) {
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1\n\tds_write_b32 %0, %1\n\tds_write_b32 %0, %1"
                 :
                 : "v"(addr), "v"(0.0f)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}
// (no instruction: the values read by asm may only be used behind this point, i.e. behind the wait in front of it)
__device__ __forceinline__ void lds_tie4(f32x4s& a, f32x4s& b, f32x4s& c, f32x4s& d) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
// Static sites of a pass (compile-time recursion: the register sets must be indexed by constants): site S issues the
// four reads of step S, does the FMAs of step S - D behind a wait for "at most 4 D reads outstanding", and calls
// hook(S) -- unconditionally, so that the loads the hook issues (the next block's entries, one per site: a burst at the
// top of the block filled the CU's vector-memory queue and every walker stood 2 800 cycles per block in the issue of
// its loads) sit at fixed program points and the compiler's vmcnt counts stay exact.  Steps [ra, rb) of the 2 K steps
// held in registers, ra = 0 or 1.  A site that has no real step to issue (step 0 when ra = 1, the D sites behind the
// last step) issues four 4-byte dummy reads instead, so the wait count is the same constant at every site: the
// instruction count per site is what bounds the walk (a wavefront issues one instruction per 4 cycles; the first
// version spent 60 % of its instructions on scalar bookkeeping around the waits).
constexpr int SK_D = 2;         // pipeline depth in steps (4 ds_read_b128 each): D + 1 sets of read results
template <int S, int K, class FI, class FP, class FF, class FH>
__device__ __forceinline__ void stream_sites(int ra, int rb, FI&& issue_s, FP&& pad_s, FF&& fma_s, FH&& hook) {
    if constexpr (S < 2 * K + SK_D) {
        if (S < rb + SK_D) {                                             // wave-uniform
            bool real = false;
            if constexpr (S < 2 * K) real = S < rb && (S > 0 || ra == 0);
            if (real) {
                if constexpr (S < 2 * K) issue_s(std::integral_constant<int, S>());
            } else {
                pad_s(std::integral_constant<int, S>());
            }
            if constexpr (S >= SK_D) {
                if (S > SK_D || ra == 0) fma_s(std::integral_constant<int, S - SK_D>());
            }
        }
        hook(std::integral_constant<int, S>());
        stream_sites<S + 1, K>(ra, rb, issue_s, pad_s, fma_s, hook);
    }
}

// ABL (timing build only): 1 = no walk, 2 = no staging, 16 = cycle stamps instead of the result
template <int ABL>
__global__ __launch_bounds__(SK_THREADS) void spmm_stream_kernel(StreamDev t, const float* __restrict__ X,
                                                                 float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) char smem[SK_LDS + 256];     // + a dummy word per lane (lds_pad4)
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
        const i32x4* hdrp = t.hdr + (size_t)tb0 * S_NW;        // any wavefront's header carries the block id
        auto stage = [&](int b, int img, bool paced) {
            const int c0 = __builtin_amdgcn_readfirstlane(hdrp[8 * b].z) * S_CB;
            const int nbytes = min(S_CB, t.n_src - c0) * S_ROW_BYTES;
            const char* src = reinterpret_cast<const char*>(X + (size_t)c0 * 16) + lane * 16;
            char* img_base = smem + img * SK_IMG;
#pragma unroll
            for (int i = 0; i < SK_PPS; ++i) {
                const int piece = d + SK_STAGERS * i;
                if (piece * 1024 + lane * 16 < nbytes && !(ABL & 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + piece * 1024),
                        (__attribute__((address_space(3))) void*)(img_base + piece * 1024), 16, 0, 0);
                if (paced) __builtin_amdgcn_s_sleep(3);
            }
        };
        stage(0, 0, false);
        SK_TICK(0)
        __syncthreads();
        SK_TICK(1)
        for (int k = 0; k < nb; ++k) {
            if (k + 1 < nb) stage(k + 1, (k + 1) & 1, true);
            SK_TICK(0)
            __syncthreads();
            SK_TICK(1)
        }
    } else {
        // ------------------------------------------------ walkers ------------------------------------
        const int quad = lane >> 2, part = lane & 3;
        const i32x4* rowp = t.rows + ((size_t)tb0 * S_NW + wave) * 16 + quad;     // + 128 per block
        const i32x4* hdrp = t.hdr + (size_t)tb0 * S_NW + wave;                    // + 8 per block (wave-uniform)
        i32x4 rc = __builtin_nontemporal_load(rowp), hc = hdrp[0];
        i32x4 hn = hdrp[8 * min(1, nb - 1)];
        Ent3 ea[S_K0], eb[S_K1];
        auto ld3 = [&](const Ent3* p) {
            Ent3 e;
            e.o = __builtin_nontemporal_load(&p->o);
            e.v0 = __builtin_nontemporal_load(&p->v0);
            e.v1 = __builtin_nontemporal_load(&p->v1);
            return e;
        };
        // entries of a block: pass 0 from the group of its first step, pass 1 from the group of ITS first step
        auto set_base = [&](const i32x4& h, int pass) {
            const int S = __builtin_amdgcn_readfirstlane(h.x), n0 = __builtin_amdgcn_readfirstlane(h.y) & 0xffff;
            return t.ent + (size_t)((S + (pass ? n0 : 0)) >> 1) * 64 + lane;
        };
        {
            const Ent3* p = set_base(hc, 0);
            const Ent3* q = set_base(hc, 1);
#pragma unroll
            for (int j = 0; j < S_K0; ++j) ea[j] = ld3(p + 64 * j);
#pragma unroll
            for (int j = 0; j < S_K1; ++j) eb[j] = ld3(q + 64 * j);
        }

        // the four source rows of one step (lane = 4 channels of the quad's four rows; `o` = this lane's entry offset)
        auto issue = [&](int o, unsigned pb, f32x4s (&x)[4]) {
            if (ABL & 4) {      // timing only: no LDS reads
                x[0] = x[1] = x[2] = x[3] = f32x4s{1.f, 1.f, 1.f, (float)o};
                return;
            }
            x[0] = lds_read16(pb + qbcast<0>(o));
            x[1] = lds_read16(pb + qbcast<1>(o));
            x[2] = lds_read16(pb + qbcast<2>(o));
            x[3] = lds_read16(pb + qbcast<3>(o));
        };
        auto fma = [&](int v, const auto (&x)[4], f32x2s (&acc)[8]) {
            if (ABL & 8) {      // timing only: one FMA instead of the eight packed ones
                acc[0].x += x[0].x * __int_as_float(v) + x[1].y + x[2].z + x[3].w;
                return;
            }
            pk4(__int_as_float(qbcast<0>(v)), x[0], acc[0], acc[1]);
            pk4(__int_as_float(qbcast<1>(v)), x[1], acc[2], acc[3]);
            pk4(__int_as_float(qbcast<2>(v)), x[2], acc[4], acc[5]);
            pk4(__int_as_float(qbcast<3>(v)), x[3], acc[6], acc[7]);
        };
        // one pass: steps [a, b) (absolute) of the quad's rows r01 = row0 | row1 << 16, r23; `cur` holds 2 K steps from
        // step (a & ~1) on.  Static sites: site s issues the reads of step s and does the FMAs of step s - SK_D, which
        // wait until at most 4 x (steps issued behind it) reads are outstanding.  The rows' accumulators are read at
        // the start and added at the end.  Steps beyond the register set (a row with dozens of entries inside one
        // block) are fetched group by group: slow path, kept apart so that the common path never waits for a load.
        // `np`: where the groups of the same pass of the NEXT block start (group j is reloaded behind its last step)
        auto pass = [&](auto& cur, const Ent3* np, auto kk, int r01, int r23, int a, int b, unsigned pb) {
            constexpr int K = decltype(kk)::value;
            if (ABL & 1) a = b;
            const int i0 = (r01 & 0xffff) * 4 + part, i1 = ((r01 >> 16) & 0xffff) * 4 + part;
            const int i2 = (r23 & 0xffff) * 4 + part, i3 = ((r23 >> 16) & 0xffff) * 4 + part;
            f32x4s y0 = lds_read16(SK_YA + i0 * 16), y1 = lds_read16(SK_YA + i1 * 16);
            f32x4s y2 = lds_read16(SK_YA + i2 * 16), y3 = lds_read16(SK_YA + i3 * 16);
            f32x2s acc[8] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            const int ra = a & 1, rb = b - (a & ~1);
            f32x4s x[SK_D + 1][4];
            const unsigned pad_addr = (unsigned)(SK_LDS + lane * 4);
            stream_sites<0, K>(
                ra, rb,
                [&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    issue((s & 1) ? (int)((unsigned)cur[s >> 1].o >> 16) : (cur[s >> 1].o & 0xffff), pb, x[s % (SK_D + 1)]);
                },
                [&](auto) {
                    if (!(ABL & 4)) lds_pad4(pad_addr);
                },
                [&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    f32x4s(&xf)[4] = x[f % (SK_D + 1)];
                    lds_wait<4 * SK_D>();
                    lds_tie4(xf[0], xf[1], xf[2], xf[3]);
                    fma((f & 1) ? cur[f >> 1].v1 : cur[f >> 1].v0, xf, acc);
                },
                [&](auto hc_) {
                    constexpr int h = decltype(hc_)::value - 1 - SK_D;      // site of the last FMA of group h / 2
                    if constexpr (h >= 0 && h % 2 == 0 && h / 2 < K) {
                        if (!(ABL & 64)) cur[h / 2] = ld3(np + 64 * (h / 2));      // (64: timing only, no reloads)
                    }
                });
            lds_wait<0>();
            lds_tie4(y0, y1, y2, y3);
            acc[0] += f32x2s{y0.x, y0.y}; acc[1] += f32x2s{y0.z, y0.w};
            acc[2] += f32x2s{y1.x, y1.y}; acc[3] += f32x2s{y1.z, y1.w};
            acc[4] += f32x2s{y2.x, y2.y}; acc[5] += f32x2s{y2.z, y2.w};
            acc[6] += f32x2s{y3.x, y3.y}; acc[7] += f32x2s{y3.z, y3.w};
            for (int st = 2 * K; st < rb; ++st) {
                const Ent3 e = ld3(t.ent + (size_t)((a >> 1) + (st >> 1)) * 64 + lane);
                f32x4s xs[4];
                issue((st & 1) ? (int)((unsigned)e.o >> 16) : (e.o & 0xffff), pb, xs);
                lds_wait<0>();
                lds_tie4(xs[0], xs[1], xs[2], xs[3]);
                fma((st & 1) ? e.v1 : e.v0, xs, acc);
            }
            Ya[i0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            Ya[i1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
            Ya[i2] = make_float4(acc[4].x, acc[4].y, acc[5].x, acc[5].y);
            Ya[i3] = make_float4(acc[6].x, acc[6].y, acc[7].x, acc[7].y);
        };
        SK_TICK(0)
        __syncthreads();
        SK_TICK(1)
        for (int k = 0; k < nb; ++k) {
            // rows of the next block, header of the one after it (the walk below needs the next header's addresses)
            const i32x4 rn = __builtin_nontemporal_load(rowp + 128 * min(k + 1, nb - 1));
            const i32x4 h2 = hdrp[8 * min(k + 2, nb - 1)];
            // (32: timing only, every block reloads the tile's first groups: cache hits)
            const Ent3* npa = (ABL & 32) ? t.ent + lane : set_base(hn, 0);
            const Ent3* npb = (ABL & 32) ? t.ent + lane : set_base(hn, 1);
            SK_TICK(3)
            const int S_ = __builtin_amdgcn_readfirstlane(hc.x);
            const unsigned c_ = (unsigned)__builtin_amdgcn_readfirstlane(hc.y);
            const int n0_ = (int)(c_ & 0xffffu), n1_ = (int)(c_ >> 16);
            const unsigned pb_ = (unsigned)((k & 1) * SK_IMG + part * 16);
            pass(ea, npa, std::integral_constant<int, S_K0>(), rc.x, rc.y, S_, S_ + n0_, pb_);
            pass(eb, npb, std::integral_constant<int, S_K1>(), rc.z, rc.w, S_ + n0_, S_ + n0_ + n1_, pb_);
            SK_TICK(2)
            rc = rn; hc = hn; hn = h2;
            __syncthreads();
            SK_TICK(1)
        }
    }
    if (ABL & 16) {
        // stamps instead of the result, summed over the wavefronts of a role: row 2 * tile = walkers {[0] prologue,
        // [1] wait at the barrier, [2] walk, [3] prefetch issue, [4] total}, row 2 * tile + 1 = stagers {[0] LDS-DMA issue
        // + pacing + landing, [1] wait at the barrier, [4] total}
        const unsigned total_ = (unsigned)__builtin_amdgcn_s_memtime() - start_;
        __syncthreads();
        int* acc = reinterpret_cast<int*>(smem);
        if (tid < 32) acc[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            const int o = wave >= S_NW ? 16 : 0;
            for (int k = 0; k < 4; ++k) atomicAdd(&acc[o + k], (int)cyc[k]);
            atomicAdd(&acc[o + 4], (int)total_);
        }
        __syncthreads();
        if (tid < 32 && 2 * tile + 1 < t.n_dst) Y[(size_t)(2 * tile) * 16 + tid] = (float)acc[tid];
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
    t.rows = reinterpret_cast<const i32x4*>(sc.rows);
    t.hdr = reinterpret_cast<const i32x4*>(sc.hdr);
    t.ent = reinterpret_cast<const Ent3*>(sc.ent);
    t.n_tiles = sc.n_tiles;
    t.n_dst = n_dst;
    t.n_src = n_src;
    int abl = 0;
#ifdef MLLP_TIMING_BUILD
    if (const char* e = getenv("MLLP_STREAM_ABLATION")) abl = atoi(e);
    if (abl == 1) hipLaunchKernelGGL(spmm_stream_kernel<1>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 3) hipLaunchKernelGGL(spmm_stream_kernel<3>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 2) hipLaunchKernelGGL(spmm_stream_kernel<2>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 4) hipLaunchKernelGGL(spmm_stream_kernel<4>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 8) hipLaunchKernelGGL(spmm_stream_kernel<8>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 12) hipLaunchKernelGGL(spmm_stream_kernel<12>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 20) hipLaunchKernelGGL(spmm_stream_kernel<20>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 24) hipLaunchKernelGGL(spmm_stream_kernel<24>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 28) hipLaunchKernelGGL(spmm_stream_kernel<28>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 48) hipLaunchKernelGGL(spmm_stream_kernel<48>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 80) hipLaunchKernelGGL(spmm_stream_kernel<80>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 92) hipLaunchKernelGGL(spmm_stream_kernel<92>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    else if (abl == 16) hipLaunchKernelGGL(spmm_stream_kernel<16>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
#endif
    if (abl == 0) hipLaunchKernelGGL(spmm_stream_kernel<0>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    MLLP_HIP_TRY(hipGetLastError());
    return MLLP_OK;
}

}  // namespace mllp
