// stream_spmm.hip -- the roofline kernel of round 3: Y = A H (16 fp32 channels) on the STREAMED copy of an orientation
// (layout, geometry and the reasons: stream_layout.h).  The reference's equivalent is the gather / scale / scatter-add
// of PyG's message passing over `edge_index`, `edge_attr` (linear_program_methods.py:241-247).
//
// Workgroup = one row tile (at most 960 rows of one instance) on one CU: 8 walking wavefronts + 4 staging wavefronts.
//   entries    each walking wavefront holds, in registers, its own entries of the current (tile, block): one register
//              set per pass (S_K0 / S_K1 groups, 12 bytes per lane and group); a group is reloaded with the NEXT block's
//              entries (coalesced non-temporal loads, one per site) as soon as its last step is done, so the entries
//              never touch LDS.  The pass-1 set is double-buffered (its reload is hidden behind pass 0 of the next block).
//   walk       two passes per block, S_RQ = 4 rows per quad: per step and quad one entry of each row, shared inside the
//              quad by DPP, one ds_read_b128 of the source row per entry, two packed FMAs per lane.  The steps are
//              static code, software-pipelined by hand, one inline-asm statement per step (`sk_site`): the reads of step
//              s are issued before the FMAs of step s - 1.  The rows' accumulators live in fixed registers during a pass:
//              read from LDS at its start, written back at its end.
//   staging    the NEXT block's 750 x 64 B of H by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) into
//              the image that is not being read, by the 4 staging wavefronts, paced with s_sleep.
//   barrier    one per block: behind it image b + 1 and everybody's entries of block b + 1 have landed (vmcnt(0) on
//              every wavefront, MI355X_MICROARCH.md "Two waves per SIMD" item 7) and the reads of image b are done.
//   LDS        image 0 | image 1 (48 000 B + one all-zero row each) | accumulators of the tile (64 KB).
// What is done by hand because hipcc's register allocation and s_waitcnt insertion cannot see through it
// (profiles/r03_stream_experiments.txt): see the comment at `sk_site`.
// Deterministic: a row of a (tile, block) belongs to one quad, blocks are walked in order, no atomics.
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "internal.h"
#include "stream_layout.h"

namespace mllp {

constexpr int SK_STAGERS = S_NW == 8 ? 4 : 0;      // separate staging wavefronts (16 walkers stage for themselves)
constexpr int SK_THREADS = 64 * (S_NW + SK_STAGERS);
constexpr int SK_IMG = S_CB * S_ROW_BYTES + S_ROW_BYTES;      // image + the all-zero row
constexpr int SK_YA = 2 * SK_IMG;                             // accumulators behind the two images
constexpr int SK_LDS = SK_YA + S_R * S_ROW_BYTES;
constexpr int SK_PIECES = (S_CB * S_ROW_BYTES + 1023) / 1024; // LDS-DMA pieces of 1 KB per image
constexpr int SK_NSTG = SK_STAGERS ? SK_STAGERS : S_NW;       // wavefronts that stage
constexpr int SK_PPW = (SK_PIECES + SK_NSTG - 1) / SK_NSTG;   // pieces per staging wavefront
#ifndef MLLP_SK_D
#define MLLP_SK_D 1
#endif
#ifndef MLLP_SK_SLEEP
#define MLLP_SK_SLEEP 3
#endif
constexpr int SK_D = MLLP_SK_D; // pipeline depth in steps: D + 1 sets of read results
constexpr int SK_NQ = (SK_D + 1) * S_RQ;        // register quads that the asm owns: D + 1 sets of S_RQ rows
constexpr int SK_VGPRS = 114;                   // what is left for the compiler: the asm owns v128-v167
static_assert(SK_D == 1 && S_RQ == 4 && S_GS == 2, "the hand-written sites are for one step in flight, four rows per quad");
static_assert(SK_NQ <= 16, "SK_QUADS names 16 quads");
static_assert(SK_LDS + 256 <= 163840 && SK_THREADS <= 1024, "one CU");

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) Ent3 {      // one lane's share of a group: two steps of its row slot
    int o;        // byte offset of the source row of the first step | of the second << 16
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
__device__ __forceinline__ void pk4(float v, const f32x4s& x, f32x2s& lo, f32x2s& hi) {
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
// The walk is hand-written: one asm statement per SITE (the reads of step s, then the FMAs of step s - 1 behind a counted
// wait), on FIXED registers that the compiler never sees -- the kernel is compiled for SK_VGPRS = 114 VGPRs
// (amdgpu_num_vgpr) and everything above is named in the asm text:
//   v[164:167] v[160:163] v[156:159] v[152:155]   source rows of an EVEN step (rows 0-3 of the quad)
//   v[148:151] v[144:147] v[140:143] v[136:139]   source rows of an ODD step
//   v135  this lane's offset of the step; v134  LDS address; v[132:133], v[114:115]  a row's value as pk_fma operand
//   v[116:131]  the accumulators of the quad's four rows (this lane's four channels of each)
// Why by hand (profiles/r03_stream_experiments.txt): an asm read's destination is written when the data returns, long
// after the statement -- with compiler-allocated destinations hipcc copied such values at control-flow joins before
// the wait (stale data; the late write then landed in a register it had reused: a memory fault); with compiler-issued
// reads it merges the LDS counter state of conditionally executed sites into lgkmcnt(0); and between DPP instructions
// that it allocates to one register it inserts s_nop's for the `old` operand that bound_ctrl never reads.  A wavefront
// issues about one instruction per 4-5 cycles whatever it is, so the instruction count per step bounds the walk: 22 per
// site here (+ 2 scalar for the guard and the hook's conditional load), 43 in the compiler-scheduled version.
// tools/check_asm_reads.py verifies on the generated assembly, at every build, that no compiler-generated instruction
// names a reserved register and that no VALU result is read by a DPP instruction of the asm within two instructions.
#if defined(MLLP_TIMING_BUILD) && defined(MLLP_SK_WAIT8)      // (timing build only: a deeper pipeline; the results are wrong)
#define SK_WAITN "s_waitcnt lgkmcnt(8)\n\t"
#else
#define SK_WAITN "s_waitcnt lgkmcnt(4)\n\t"
#endif
#define SK_DPP(K) " quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SK_READS(Q0, Q1, Q2, Q3)                                                   \
    "v_add_u32_dpp v134, v135, %[pb]" SK_DPP(0) "ds_read_b128 v[" Q0 "], v134\n\t"  \
    "v_add_u32_dpp v134, v135, %[pb]" SK_DPP(1) "ds_read_b128 v[" Q1 "], v134\n\t"  \
    "v_add_u32_dpp v134, v135, %[pb]" SK_DPP(2) "ds_read_b128 v[" Q2 "], v134\n\t"  \
    "v_add_u32_dpp v134, v135, %[pb]" SK_DPP(3) "ds_read_b128 v[" Q3 "], v134\n\t"
#define SK_VAL(K, P) "v_mov_b32_dpp v" P ", %[val]" SK_DPP(K)
#define SK_FMA2(P, A, B, X0, X1)                                                   \
    "v_pk_fma_f32 v[" A "], v[" P "], v[" X0 "], v[" A "] op_sel_hi:[0,1,1]\n\t"       \
    "v_pk_fma_f32 v[" B "], v[" P "], v[" X1 "], v[" B "] op_sel_hi:[0,1,1]\n\t"
// (the values of rows 0 and 1 are broadcast in front of the reads -- SK_VAL01: they fill the two wait states between the
// write of v135 and its first DPP read)
#define SK_VAL01 SK_VAL(0, "132") SK_VAL(1, "114")
#define SK_FMAS(A0, A1, B0, B1, C0, C1, D0, D1)                                                                  \
    SK_FMA2("132:133", "116:117", "118:119", A0, A1) SK_FMA2("114:115", "120:121", "122:123", B0, B1)            \
    SK_VAL(2, "132") SK_VAL(3, "114")                                                                            \
    SK_FMA2("132:133", "124:125", "126:127", C0, C1) SK_FMA2("114:115", "128:129", "130:131", D0, D1)
#define SK_READS_EVEN SK_READS("164:167", "160:163", "156:159", "152:155")
#define SK_READS_ODD SK_READS("148:151", "144:147", "140:143", "136:139")
#define SK_FMAS_EVEN SK_FMAS("164:165", "166:167", "160:161", "162:163", "156:157", "158:159", "152:153", "154:155")
#define SK_FMAS_ODD SK_FMAS("148:149", "150:151", "144:145", "146:147", "140:141", "142:143", "136:137", "138:139")
#define SK_CLOBBER                                                                                                    \
    "memory", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", \
        "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",      \
        "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154",      \
        "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"

// site of step S (parity PAR = S & 1): reads of step S, FMAs of step S - 1.  `o`: the lane's offset word of the group of
// step S, `val`: its value of step S - 1.  MODE 0 = both, 1 = reads only (first step of a pass), 2 = FMAs only (behind the
// last step).  ABL (timing build): 4 = no LDS reads, 8 = no FMAs.
template <int PAR, int MODE, int ABL>
__device__ __forceinline__ void sk_site(int o, int val, unsigned pb) {
    constexpr bool RD = MODE != 2 && !(ABL & 4), FM = MODE != 1 && !(ABL & 8);
    if constexpr (RD && FM) {
        if constexpr (PAR == 0)
            asm volatile("v_and_b32 v135, 0xffff, %[o]\n\t" SK_VAL01 SK_READS_EVEN SK_WAITN SK_FMAS_ODD
                         : : [o] "v"(o), [val] "v"(val), [pb] "v"(pb) : SK_CLOBBER);
        else
            asm volatile("v_lshrrev_b32 v135, 16, %[o]\n\t" SK_VAL01 SK_READS_ODD SK_WAITN SK_FMAS_EVEN
                         : : [o] "v"(o), [val] "v"(val), [pb] "v"(pb) : SK_CLOBBER);
    } else if constexpr (RD) {
        if constexpr (PAR == 0)
            asm volatile("v_and_b32 v135, 0xffff, %[o]\n\ts_nop 1\n\t" SK_READS_EVEN : : [o] "v"(o), [pb] "v"(pb) : SK_CLOBBER);
        else
            asm volatile("v_lshrrev_b32 v135, 16, %[o]\n\ts_nop 1\n\t" SK_READS_ODD : : [o] "v"(o), [pb] "v"(pb) : SK_CLOBBER);
    } else if constexpr (FM) {
        if constexpr (PAR == 0)     // (the step whose FMAs these are is odd)
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" SK_VAL01 SK_FMAS_ODD : : [val] "v"(val) : SK_CLOBBER);
        else
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" SK_VAL01 SK_FMAS_EVEN : : [val] "v"(val) : SK_CLOBBER);
    }
}
// the accumulators of the quad's four rows (v[116:131]): read from / written back to the tile's accumulators in LDS
__device__ __forceinline__ void sk_acc_load(unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("ds_read_b128 v[116:119], %0\n\tds_read_b128 v[120:123], %1\n\tds_read_b128 v[124:127], %2\n\t"
                 "ds_read_b128 v[128:131], %3" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : SK_CLOBBER);
}
__device__ __forceinline__ void sk_acc_store(unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b128 %0, v[116:119]\n\tds_write_b128 %1, v[120:123]\n\t"
                 "ds_write_b128 %2, v[124:127]\n\tds_write_b128 %3, v[128:131]" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : SK_CLOBBER);
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}

// Static sites of a pass (compile-time recursion: the entry registers must be indexed by constants).  `cur` holds S_GS K
// steps from the group of the pass's first step on; the pass walks steps [ra, rb) of them, ra = 0 or 1.  hook(S) runs at
// every site, behind the walk's end too: what it issues (the next block's entries, one conditional load per site -- a
// burst at the top of the block fills the CU's vector-memory queue and every wavefront stands in the issue of its
// loads) sits at fixed program points.  Steps behind the register set are walked by the slow path of `pass`.
template <int S, int K, int ABL, class Cur, class FH>
__device__ __forceinline__ void stream_sites(int ra, int rb, const Cur& cur, unsigned pb, FH&& hook) {
    constexpr int N = S_GS * K;
    if constexpr (S <= N) {
        constexpr int g = (S < N ? S : N - 1) / S_GS, gp = (S > 0 ? S - 1 : 0) / S_GS;
        const int o = cur[g].o;                                                   // offsets of step S (S < N)
        const int v = ((S - 1) & 1) ? cur[gp].v1 : cur[gp].v0;                    // value of step S - 1 (S > 0)
        if constexpr (S == 0) {
            if (ra == 0 && 0 < rb) sk_site<0, 1, ABL>(o, 0, pb);
        } else if constexpr (S == 1) {
            if (1 < rb) {
                if (ra == 0) sk_site<1, 0, ABL>(o, v, pb);
                else sk_site<1, 1, ABL>(o, 0, pb);
            } else if (ra == 0 && rb == 1) {
                sk_site<1, 2, ABL>(0, v, pb);
            }
        } else if constexpr (S < N) {
            if (__builtin_expect(S < rb, 1)) sk_site<S & 1, 0, ABL>(o, v, pb);
            if (__builtin_expect(S == rb, 0)) sk_site<S & 1, 2, ABL>(0, v, pb);
        } else {
            if (N <= rb) sk_site<S & 1, 2, ABL>(0, v, pb);                    // (the slow path continues from here)
        }
        hook(std::integral_constant<int, S>());
        stream_sites<S + 1, K, ABL>(ra, rb, cur, pb, hook);
    }
}

// ABL (timing build only): 1 = no walk, 2 = no staging, 4 = no LDS reads, 8 = no FMAs, 16 = cycle stamps instead of the
// result, 64 = no entry reloads
template <int ABL>
__global__ __launch_bounds__(SK_THREADS) __attribute__((amdgpu_num_vgpr(SK_VGPRS))) void spmm_stream_kernel(StreamDev t, const float* __restrict__ X,
                                                                 float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) char smem[SK_LDS + 256];     // + a dummy word per lane (lds_pad)
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

    if (SK_STAGERS && wave >= S_NW) {
        // ------------------------------------------------ stagers (8-walker configuration) ----------------
        const int d = wave - S_NW;
        const i32x4* hp = t.hdr + (size_t)tb0 * S_NW;          // any wavefront's header carries the block id
        auto stage = [&](int b, int img, bool paced) {
            const int c0 = __builtin_amdgcn_readfirstlane(hp[S_NW * b].z) * S_CB;
            const int nbytes = min(S_CB, t.n_src - c0) * S_ROW_BYTES;
            const char* src = reinterpret_cast<const char*>(X + (size_t)c0 * 16) + lane * 16;
            char* img_base = smem + img * SK_IMG;
#pragma unroll
            for (int i = 0; i < SK_PPW; ++i) {
                const int piece = d + SK_NSTG * i;
                if (piece * 1024 + lane * 16 < nbytes && !(ABL & 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + piece * 1024),
                        (__attribute__((address_space(3))) void*)(img_base + piece * 1024), 16, 0, 0);
                if (paced) __builtin_amdgcn_s_sleep(MLLP_SK_SLEEP);      // a burst would block the walkers in their own loads
            }
        };
        stage(0, 0, false);
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            if (k + 1 < nb) stage(k + 1, (k + 1) & 1, true);
            __syncthreads();
        }
        if (ABL & 16) {
            __syncthreads();
            __syncthreads();
            __syncthreads();
            return;
        }
        for (int i = tid; i < n4; i += SK_THREADS) dst[i] = Ya[i];
        return;
    }
    const int quad = lane >> 2, part = lane & 3;
    const i32x4* rowp = t.rows + ((size_t)tb0 * S_NW + wave) * 16 + quad;     // + 16 S_NW per block
    const i32x4* hdrp = t.hdr + (size_t)tb0 * S_NW + wave;                    // + S_NW per block (wave-uniform)
    i32x4 rc = __builtin_nontemporal_load(rowp), hc = hdrp[0];
    i32x4 hn = hdrp[S_NW * min(1, nb - 1)];

    // piece i (of SK_PPW) of this wavefront's share of column block `blk` -> image `img`
    auto stage_piece = [&](int blk, int img, int i) {
        const int c0 = __builtin_amdgcn_readfirstlane(blk) * S_CB;
        const int nbytes = min(S_CB, t.n_src - c0) * S_ROW_BYTES;
        const int off = (wave + SK_NSTG * i) * 1024;
        if (!SK_STAGERS && off + lane * 16 < nbytes && !(ABL & 2))
            glds16(reinterpret_cast<const char*>(X + (size_t)c0 * 16) + off + lane * 16,
                   __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(img * SK_IMG + off)));
    };
    if (!SK_STAGERS)
        for (int i = 0; i < SK_PPW; ++i) stage_piece(hc.z, 0, i);

    Ent3 ea[S_K0], eb[S_K1], eb2[S_K1];     // pass 1 has two sets: the next block's is loaded during pass 0 (see `block`)
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
        return t.ent + (size_t)((S + (pass ? n0 : 0)) / S_GS) * 64 + lane;
    };
    {
        const Ent3* p = set_base(hc, 0);
        const Ent3* q = set_base(hc, 1);
#pragma unroll
        for (int j = 0; j < S_K0; ++j) ea[j] = ld3(p + 64 * j);
#pragma unroll
        for (int j = 0; j < S_K1; ++j) eb[j] = ld3(q + 64 * j);
    }

    // One pass: steps [a, b) (absolute) of the quad's rows (r01 = slot 0 | slot 1 << 16, r23 likewise); `cur` holds
    // S_GS K steps from the group of step a on; `np`: where the groups of the same pass of the NEXT block start
    // (group j is reloaded behind its last step); `extra(site)`: what else the hooks issue.  Steps beyond the register
    // set (a row with dozens of entries inside one block) are fetched group by group: slow path, kept apart so that
    // the common path never waits for a load.
    auto pass = [&](auto& cur, const Ent3* np, int gn, auto kk, int r01, int r23, int a, int b, unsigned pb, auto&& extra) {
        constexpr int K = decltype(kk)::value;
        if (ABL & 1) a = b;
        if ((ABL & 256) && wave != 0) a = b;      // timing only: one wavefront walks alone
        unsigned ya[S_RQ];      // LDS byte addresses of the rows' accumulators (this lane's 16-byte piece)
#pragma unroll
        for (int r = 0; r < S_RQ; ++r)
            ya[r] = (unsigned)(SK_YA + (((((r & 2) ? r23 : r01) >> (16 * (r & 1))) & 0xffff) * 4 + part) * 16);
        const int ra = a % S_GS, rb = b - (a - ra);
        const bool any = rb > ra;                                                   // wave-uniform
        if (any) sk_acc_load(ya[0], ya[1], ya[2], ya[3]);
        stream_sites<0, K, ABL>(ra, rb, cur, pb, [&](auto hc_) {
            constexpr int site = decltype(hc_)::value;
            constexpr int h = site - (S_GS - 1) - SK_D;             // site of the last FMA of group h / S_GS
            if constexpr (h >= 0 && h % S_GS == 0 && h / S_GS < K) {
                // (only the groups the next block's pass will use: the CU's vector-memory path, ~30 B/clk, is what
                // bounds the kernel, and loading the whole set every block tripled the entry bytes through it)
                if (__builtin_expect(h / S_GS < gn, 1) && !(ABL & 64)) cur[h / S_GS] = ld3(np + 64 * (h / S_GS));
            }
            extra(hc_);
        });
        if (any) sk_acc_store(ya[0], ya[1], ya[2], ya[3]);      // (waits for lgkmcnt(0) first)
        if (rb > S_GS * K) {
            // slow path: the steps behind the register set (a row with dozens of entries inside one block), fetched
            // group by group and accumulated in ordinary registers, added to the rows' accumulators in LDS at the end
            // (nothing of the asm's is live here: under register pressure the compiler may use v116-v167 between the two
            // markers, and only there -- tools/check_asm_reads.py)
            asm volatile("; SK_SLOW_BEGIN" : : : "memory");
            f32x2s acc[2 * S_RQ];
#pragma unroll
            for (int r = 0; r < 2 * S_RQ; ++r) acc[r] = f32x2s{0.f, 0.f};
            for (int st = S_GS * K; st < rb; ++st) {
                const Ent3 e = ld3(t.ent + (size_t)(a / S_GS + st / S_GS) * 64 + lane);
                const int o = (st & 1) ? (int)((unsigned)e.o >> 16) : (e.o & 0xffff), v = (st & 1) ? e.v1 : e.v0;
#pragma unroll
                for (int r = 0; r < S_RQ; ++r) {
                    const int src = (lane & ~3) | r;
                    const f32x4s xs = *reinterpret_cast<const f32x4s*>(smem + (pb + __shfl(o, src, 64)));
                    pk4(__int_as_float(__shfl(v, src, 64)), xs, acc[2 * r], acc[2 * r + 1]);
                }
            }
#pragma unroll
            for (int r = 0; r < S_RQ; ++r) {
                float4* yp = reinterpret_cast<float4*>(smem + ya[r]);
                const float4 y = *yp;
                *yp = make_float4(y.x + acc[2 * r].x, y.y + acc[2 * r].y, y.z + acc[2 * r + 1].x, y.w + acc[2 * r + 1].y);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): no compiler-tracked LDS operation stays pending
            asm volatile("; SK_SLOW_END" : : : "memory");
        }
    };
    SK_TICK(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): image 0 and the first entries have landed
    __syncthreads();
    SK_TICK(1)
    // One block.  The pass-0 set `ea` is reloaded in place (group j behind its last step); the pass-1 set of the NEXT
    // block goes into the other of eb / eb2 during pass 0: reloaded in place, its loads would be the last ones of the
    // block and every wavefront stood ~700 cycles per block at the barrier waiting for them to land.
    auto block = [&](int k, auto& ebc, auto& ebn) {
        // rows of the next block, header of the one after it (the walk below needs the next header's addresses)
        const i32x4 rn = __builtin_nontemporal_load(rowp + 16 * S_NW * min(k + 1, nb - 1));
        const i32x4 h2 = hdrp[S_NW * min(k + 2, nb - 1)];
        const Ent3* npa = set_base(hn, 0);
        const Ent3* npb = set_base(hn, 1);
        const int nS_ = __builtin_amdgcn_readfirstlane(hn.x);
        const unsigned nc_ = (unsigned)__builtin_amdgcn_readfirstlane(hn.y);
        const bool more = k + 1 < nb;
        // groups the next block's passes use
        const int gna = more ? (nS_ % S_GS + (int)(nc_ & 0xffffu) + S_GS - 1) / S_GS : 0;
        const int gnb = more ? ((nS_ + (int)(nc_ & 0xffffu)) % S_GS + (int)(nc_ >> 16) + S_GS - 1) / S_GS : 0;
        const int nblk = hn.z, nimg = (k + 1) & 1;
        SK_TICK(3)
        const int S_ = __builtin_amdgcn_readfirstlane(hc.x);
        const unsigned c_ = (unsigned)__builtin_amdgcn_readfirstlane(hc.y);
        const int n0_ = (int)(c_ & 0xffffu), n1_ = (int)(c_ >> 16);
        const unsigned pb_ = (unsigned)((k & 1) * SK_IMG + part * 16);
        pass(ea, npa, gna, std::integral_constant<int, S_K0>(), rc.x, rc.y, S_, S_ + n0_, pb_, [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s % 2 == 0 && s / 2 < S_K1) {     // the next block's pass-1 entries, one group per even site
                if (__builtin_expect(s / 2 < gnb, 1) && !(ABL & 64)) ebn[s / 2] = ld3(npb + 64 * (s / 2));
            }
            if constexpr (!SK_STAGERS && s % 3 == 1 && s / 3 < SK_PPW) {    // the LDS-DMA pieces of the next image
                if (more) stage_piece(nblk, nimg, s / 3);
            }
        });
        pass(ebc, npb, 0, std::integral_constant<int, S_K1>(), rc.z, rc.w, S_ + n0_, S_ + n0_ + n1_, pb_, [&](auto) {});
        SK_TICK(2)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next entries (and, when this wavefront stages, the image) have landed
        rc = rn; hc = hn; hn = h2;
        SK_TICK(0)
        __syncthreads();
        SK_TICK(1)
    };
    for (int k = 0; k < nb; k += 2) {
        block(k, eb, eb2);
        if (k + 1 >= nb) break;
        block(k + 1, eb2, eb);
    }
    if (ABL & 16) {
        // stamps instead of the result, summed over the wavefronts: row 2 * tile = {[0] wait for the prefetch to land,
        // [1] wait at the barrier, [2] walk, [3] top of the block, [4] total}
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
    t.rows = reinterpret_cast<const i32x4*>(sc.rows);
    t.hdr = reinterpret_cast<const i32x4*>(sc.hdr);
    t.ent = reinterpret_cast<const Ent3*>(sc.ent);
    t.n_tiles = sc.n_tiles;
    t.n_dst = n_dst;
    t.n_src = n_src;
    int abl = 0;
#ifdef MLLP_TIMING_BUILD
    if (const char* e = getenv("MLLP_STREAM_ABLATION")) abl = atoi(e);
#define SK_LAUNCH(A) \
    if (abl == A) hipLaunchKernelGGL(spmm_stream_kernel<A>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    SK_LAUNCH(1) SK_LAUNCH(2) SK_LAUNCH(3) SK_LAUNCH(4) SK_LAUNCH(8) SK_LAUNCH(12) SK_LAUNCH(16) SK_LAUNCH(18) SK_LAUNCH(20) SK_LAUNCH(24) SK_LAUNCH(28) SK_LAUNCH(80) SK_LAUNCH(272) SK_LAUNCH(284)
#undef SK_LAUNCH
#endif
    if (abl == 0) hipLaunchKernelGGL(spmm_stream_kernel<0>, dim3(sc.n_tiles), dim3(SK_THREADS), 0, s, t, H, Y);
    MLLP_HIP_TRY(hipGetLastError());
    return MLLP_OK;
}

}  // namespace mllp
