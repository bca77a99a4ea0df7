// stream_attn.hip -- the attention sweeps of the throughput regime on the STREAMED layout (stream_layout.h, geometries
// AttnGeom / BdstGeom / BsrcGeom): the design that took the plain SpMM from 0.41 to 0.55 of the HBM roofline
// (stream_spmm.hip) carried to the kernels the training step actually runs.  Reference: the five TransformerConv calls of
// GNNModel.forward (linear_program_methods.py:241-247) and autograd of them (linear_program_experiment.py:141); per
// destination i:
//   forward   l_ij = <q'_i, x_j> + a_ij t_i,  alpha = softmax_j(l),  Z_i = sum_j alpha_ij x_j,  u_i = sum_j alpha_ij a_ij,
//             o_i = relu(Wv Z_i + S_i bv + u_i w_e + Ws x_i + bs)
//   backward  dl_ij = alpha_ij (<gv_i, x_j> + a_ij ge_i + c_i);   destination-major: dq'_i = sum_j dl_ij x_j, dt_i = sum_j dl_ij a_ij
//             source-major (rows = sources j, staged items = the destinations' 160-byte records): dx_j = sum_i alpha_ij gv_i + dl_ij q'_i
//
// What is the same as in the SpMM: the nonzeros stream HBM -> registers in the order the lanes consume them (never
// LDS), one block ahead; two images of the gathered items filled by LDS-DMA under the walk; one barrier per block; the walk
// is one hand-written asm statement per step on FIXED registers the compiler never sees (tools/check_asm_reads.py).
// What is different:
//   * state   a row carries state from block to block (forward: q', the running Z, {L, u, m, t}: 144 B; destination-major
//             backward: q', gv, dq', 6 scalars: 224 B; source-major backward: x_j, dx_j: 128 B) in LDS, so the tile is 480
//             rows and the column blocks are what is left of the 160 KB (720 / 432 / 312 items).
//   * rows    TWO rows per quad and pass (16 bundles of 32 rows: eight wavefronts still walk a long and a short bundle
//             each, two wavefronts per SIMD).  The pair is read in a ROTATED pattern: in slot 0 lanes 0, 1 of the quad
//             read pieces 0, 1 of row 0's item and lanes 2, 3 pieces 2, 3 of row 1's; slot 1 the other way round.
//             After the per-lane partial dot products, one DPP add across the pairs and one inside the pair leave row
//             0's sums in lanes 0, 1 and row 1's in lanes 2, 3: the scalar arithmetic (exponential, L, u / alpha, dl) runs
//             ONCE per lane for the lane's own row instead of once per row in every lane, and only the own row's
//             coefficients cross lanes again (one DPP move each) for the packed accumulations of the other row.
//   * softmax (forward) no running maximum in the walk.  Every row keeps a REFERENCE m (log2 units, 0 until a slow pass moves
//             it): p = 2^(l - m), and a pass whose |l - m| ever exceeds 64 is thrown away and redone by the exact
//             (compiled) slow pass, which rebases m; alpha = p / L does not depend on m, so the result is the
//             reference's softmax (its detached max only guards against overflow, as this does).  One v_exp_f32, no
//             branch and no rescale inside the step: 31 instructions per 32 nonzeros of a wavefront (39 / 47 backward).
//   * padding an entry of a row that is shorter than its bundle's longest points at the all-zero item; its lane is
//             recognised by that address and contributes nothing.
//   * long rows  a pass longer than the register set continues on synchronously loaded groups (same sites).
// Deterministic: a row of a (tile, block) belongs to one quad, blocks are walked in order, no atomics.
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "internal.h"
#include "stream_layout.h"

namespace mllp {

namespace {

constexpr int MODE_FWD = 0, MODE_BDST = 1, MODE_BSRC = 2;
constexpr int A_STAGERS = 4;
constexpr int A_RS = AttnGeom::RR + 1;                   // state rows: the tile's rows + one dummy row for the slots above them
constexpr float A_LOG2E = 1.4426950408889634f, A_LN2 = 0.6931471805599453f;
#ifndef MLLP_AK_SLEEP
#define MLLP_AK_SLEEP 3
#endif

template <int MODE>
struct AM;
template <>
struct AM<MODE_FWD> {
    using G = AttnGeom;
    static constexpr int IMG = G::CB * G::ITEM + G::ITEM;     // image + the all-zero item
    static constexpr int QA = 2 * IMG, ZA = QA + A_RS * 64, SA = ZA + A_RS * 64, LDS = SA + A_RS * 16;
    static constexpr int VGPRS = 120;                         // what is left for the compiler: the asm owns v120-v167
};
template <>
struct AM<MODE_BDST> {
    using G = BdstGeom;
    static constexpr int IMG = G::CB * G::ITEM + G::ITEM;
    static constexpr int QA = 2 * IMG, GA = QA + A_RS * 64, DA = GA + A_RS * 64, SA = DA + A_RS * 64, LDS = SA + A_RS * 32;
    static constexpr int VGPRS = 106;                         // the asm owns v106-v167
};
template <>
struct AM<MODE_BSRC> {
    using G = BsrcGeom;
    static constexpr int IMG = G::CB * G::ITEM + G::ITEM;
    static constexpr int XA = 2 * IMG, DA = XA + A_RS * 64, LDS = DA + A_RS * 64;
    static constexpr int VGPRS = 94;                          // the asm owns v94-v167
};
constexpr int A_THREADS = 64 * (AttnGeom::NW + A_STAGERS);
static_assert(AttnGeom::NW == 8 && BdstGeom::NW == 8 && BsrcGeom::NW == 8 && AttnGeom::RR == BdstGeom::RR && AttnGeom::RR == BsrcGeom::RR, "one scaffold");
static_assert(AM<0>::LDS + 256 <= 163840 && AM<1>::LDS + 256 <= 163840 && AM<2>::LDS + 256 <= 163840, "one CU");
static_assert(REC_W * 4 == BsrcGeom::ITEM, "the staged items of the source-major sweep are the backward records");

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) Ent3 {      // one lane's share of a group: two steps of its row slot
    int o;        // byte offset of the item of the first step | of the second << 16
    int v0, v1;   // value bits
};
struct AttnStreamDev {
    const int* __restrict__ tile_row;
    const int* __restrict__ tile_blk;
    const i32x4* __restrict__ rows;
    const i32x4* __restrict__ hdr;
    const Ent3* __restrict__ ent;
    int n_tiles, n_dst, n_src;
};
struct AttnArgs {
    const float* __restrict__ items;    // what is gathered: X [n_src, 16] (forward, destination-major backward), rec [n_src, 40]
    // forward
    const float* __restrict__ xd;       // [n_dst, 16]
    const float* __restrict__ qp;       // [n_dst, 16]
    const float* __restrict__ tq;       // [n_dst]
    ConvParams p;
    float* __restrict__ h;              // [n_dst, 16]
    float* __restrict__ Z;              // [n_dst, 16]
    float* __restrict__ aux;            // [n_dst, 4]
    // destination-major backward
    const float* __restrict__ rec;      // [n_dst, REC_W]
    const float* __restrict__ g;        // [n_dst, 16] relu-masked output gradient
    const float* __restrict__ derived;
    float* __restrict__ dqp;            // [n_dst, 16]
    float* __restrict__ dsdt;           // [n_dst, 2]
    float* __restrict__ dx;             // [n_dst, 16] or nullptr (destination-major); [n_rows, 16] (source-major)
    const float* __restrict__ x_rows;   // source-major: [n_rows, 16] features of the rows
    int accumulate;
};

// ---- the hand-written steps -----------------------------------------------------------------------------------------
// "own row" of lane 4 q + p: row slot p >> 1 of quad q.  Step s of the quad's rows 0, 1 sits in lanes B, B + 1 of the
// quad with B = 2 ((s % 4) >> 1) (stream_layout.h: a group is four steps, lane p holds steps 4 g + 2 (p >> 1) and the one
// behind it of row slot p & 1).  A site reads the items of step S and does the arithmetic of step S - 1.
#define AK_QP(a, b, c, d) " quad_perm:[" #a "," #b "," #c "," #d "] row_mask:0xf bank_mask:0xf\n\t"
#define AK_OWN0 AK_QP(0, 0, 1, 1)     // the own row's lane when the step sits in lanes 0, 1
#define AK_OWN2 AK_QP(2, 2, 3, 3)     // ... in lanes 2, 3
#define AK_OTH0 AK_QP(1, 1, 0, 0)
#define AK_OTH2 AK_QP(3, 3, 2, 2)
// offset of an even / odd step out of the lane's offset word W (an asm operand name)
#define AK_OFF_EVEN(OFFR, W) "v_and_b32 " OFFR ", 0xffff, %[" W "]\n\t"
#define AK_OFF_ODD(OFFR, W) "v_lshrrev_b32 " OFFR ", 16, %[" W "]\n\t"
#define AK_PKMUL(D, A, B) "v_pk_mul_f32 " D ", " A ", " B "\n\t"
#define AK_PKFMA(D, A, B) "v_pk_fma_f32 " D ", " A ", " B ", " D "\n\t"
#define AK_PKACC(D, S, X) "v_pk_fma_f32 " D ", " S ", " X ", " D " op_sel_hi:[0,1,1]\n\t"      // D += S.lo * X (both halves)

struct AkConst {      // per-lane / per-wavefront constants and the running flags of a pass
    unsigned pb, pz, pbs;         // LDS address of the lane's piece of item 0 of the image; of the all-zero item; of item 0's scalars
    int ninf;                     // -inf
    float k;                      // source-major backward: log2(e)
    unsigned long long real_e, real_o;
    unsigned long long seen;      // forward: lanes whose own row had a real entry in this pass
};
#define AK_CLOBBER_TOP                                                                                                \
    "memory", "vcc", "scc", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",   \
        "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147",      \
        "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161",      \
        "v162", "v163", "v164", "v165", "v166", "v167"
#define AK_CLOBBER_F AK_CLOBBER_TOP
#define AK_CLOBBER_D AK_CLOBBER_TOP, "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"
#define AK_CLOBBER_B AK_CLOBBER_D, "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105"
#define AK_REAL(REALR, ADR) "v_cmp_ne_u32_e64 " REALR ", " ADR ", %[pz]\n\t"
// Ablations (variant libraries of the timing build only, `make -C mllp_amd/csrc abl ABL=n`; the results are wrong): bit 0 = no
// LDS reads in the sites, bit 1 = no arithmetic in the sites, bit 2 = no staging, bit 3 = no state load / store per pass,
// bit 4 = no reloads of the entry registers, bit 5 = stagers without pacing
#if defined(MLLP_TIMING_BUILD) && defined(MLLP_AK_ABL)
constexpr int AK_ABL = MLLP_AK_ABL;
#else
constexpr int AK_ABL = 0;
#endif
#if defined(MLLP_TIMING_BUILD) && defined(MLLP_AK_ABL) && (MLLP_AK_ABL & 1)
#define AK_DSREAD(X) ""
#else
#define AK_DSREAD(X) X
#endif
#if defined(MLLP_TIMING_BUILD) && defined(MLLP_AK_ABL) && (MLLP_AK_ABL & 2)
#define AK_ARITH(X) ""
#else
#define AK_ARITH(X) X
#endif

// ===== forward =====  fixed registers (compiled for 120 VGPRs):
//   v[120:123] v[124:127]   item pieces of an EVEN step: slot 0 (the lane's own row), slot 1 (the pair's other row)
//   v[128:131] v[132:135]   the same of an ODD step
//   v[136:139] v[140:143]   q' pieces (log2 units): own row, other row          v[144:147] v[148:151]   running Z likewise
//   v152 L   v153 u   v154 m   v155 t            (the lane's own row)
//   v156  the lane's offsets of the step to read        v158  LDS address        v159  partial sum -> logit -> l - m
//   v[160:161] v[162:163]   dot-product pairs, then a t - m (v161), the other pair's partial (v162), a (v163)
//   v164 p (own row; v[164:165] as pk_fma operand, which reads the low half only)
//   v[166:167] p of the other row
// A site = the arithmetic of step S - 1, then the reads of step S + 1 into the registers that step S - 1 has just left: the
// reads run TWO steps ahead on two register sets, every step has ONE copy of its code (a first version with separate
// "both / reads only / arithmetic only" copies and the block body inlined twice was 48-63 KB per kernel: the instruction
// cache that two CUs share holds 64 KB, and a site took 600 cycles whatever it contained -- profiles/r04_attn_stream_cycles.txt).
// The pass starts with the reads of its first two steps (AK_PRO) and ends with two reads too many (of whatever follows in
// the stream: harmless, waited for before the state is stored).
#define FK_ADR0(OFFR, OWNR) "v_add_u32_dpp v158, " OFFR ", %[pb]" OWNR
#define FK_READS_(OFFR, XR0, XR1, OTHR, REALR)      /* (the first address is in v158 already) */                     \
    AK_DSREAD("ds_read_b128 " XR0 ", v158\n\t") AK_REAL(REALR, "v158")                                             \
    "v_add_u32_dpp v158, " OFFR ", %[pb]" OTHR AK_DSREAD("ds_read_b128 " XR1 ", v158\n\t")
#define FK_READS(OFFR, XR0, XR1, OWNR, OTHR, REALR) FK_ADR0(OFFR, OWNR) FK_READS_(OFFR, XR0, XR1, OTHR, REALR)
#define FK_DOT(C0L, C0H, C1L, C1H)                                                                        \
    AK_PKMUL("v[160:161]", C0L, "v[136:137]") AK_PKMUL("v[162:163]", C1L, "v[140:141]")                  \
    AK_PKFMA("v[160:161]", C0H, "v[138:139]") AK_PKFMA("v[162:163]", C1H, "v[142:143]")
// partial sums of the own / the other row; a of the own row; both summed over the lane pair; a t - m; (OFFN); + the other
// pair's sum of the own row; l - m; padding -> -inf; p = 2^(l - m) (+ its wait state); L, u; the other row's p; Z
#define FK_RED(OWNC, OFFN, REALC, ADR0)                                   \
    AK_ARITH("v_add_f32 v159, v160, v161\n\t"                             \
    "v_add_f32 v162, v162, v163\n\t"                                      \
    "v_mov_b32_dpp v163, %[val]" OWNC                                     \
    "v_add_f32_dpp v159, v159, v159" AK_QP(1, 0, 3, 2)                    \
    "v_add_f32_dpp v162, v162, v162" AK_QP(1, 0, 3, 2)                    \
    "v_fma_f32 v161, v163, v155, -v154\n\t") OFFN                         \
    AK_ARITH("v_add_f32_dpp v159, v162, v159" AK_QP(2, 3, 0, 1)           \
    "v_add_f32 v159, v159, v161\n\t"                                      \
    "v_cndmask_b32_e64 v164, %[ninf], v159, " REALC "\n\t"                \
    "s_or_b64 %[seen], %[seen], " REALC "\n\t"                            \
    "v_exp_f32_e32 v164, v164\n\t") ADR0
#define FK_ACC(C0L, C0H, C1L, C1H)                                        \
    "v_add_f32 v152, v152, v164\n\t"                                      \
    "v_fmac_f32 v153, v164, v163\n\t"                                     \
    "v_mov_b32_dpp v166, v164" AK_QP(2, 3, 0, 1)                          \
    AK_PKACC("v[144:145]", "v[164:165]", C0L) AK_PKACC("v[146:147]", "v[164:165]", C0H) \
    AK_PKACC("v[148:149]", "v[166:167]", C1L) AK_PKACC("v[150:151]", "v[166:167]", C1H)
#define FK_SITE(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)                          \
    "s_waitcnt lgkmcnt(2)\n\t" AK_ARITH(FK_DOT(C0L, C0H, C1L, C1H)) FK_RED(OWNC, OFFN, REALC, FK_ADR0("v156", OWNR)) \
        AK_ARITH(FK_ACC(C0L, C0H, C1L, C1H)) FK_READS_("v156", XR0, XR1, OTHR, REALR)
#define FK_PRO(OFFA, XA0, XA1, OWNA, OTHA, REALA, OFFB, XB0, XB1, OWNB, OTHB, REALB)                       \
    OFFA "s_nop 1\n\t" FK_READS("v156", XA0, XA1, OWNA, OTHA, REALA) OFFB "s_nop 1\n\t" FK_READS("v156", XB0, XB1, OWNB, OTHB, REALB)
#define FK_E "v[120:121]", "v[122:123]", "v[124:125]", "v[126:127]"
#define FK_O "v[128:129]", "v[130:131]", "v[132:133]", "v[134:135]"

// ===== destination-major backward =====  fixed registers (compiled for 106 VGPRs):
//   v106 a   v107 a t - m
//   v[108:111] v[112:115]   item pieces of an EVEN step (own row, other row)       v[116:119] v[120:123]   of an ODD step
//   v[124:127] v[128:131] q' (log2 units)   v[132:135] v[136:139] gv   v[140:143] v[144:147] running dq'    (own, other)
//   v148 t   v149 m   v150 1/L   v151 ge   v152 c   v153 dt       (the lane's own row)
//   v154 offsets of the step to read       v164 LDS address
//   v[156:157] l-dot own -> l - m -> 2^(l - m) -> dl     v[158:159] l-dot other     v[160:161] dalpha-dot own -> e     v[162:163] other
//   v[166:167] dl of the other row
#define DK_READS(OFFR, XR0, XR1, OWNR, OTHR, REALR)                                                       \
    "v_add_u32_dpp v164, " OFFR ", %[pb]" OWNR AK_DSREAD("ds_read_b128 " XR0 ", v164\n\t") AK_REAL(REALR, "v164")     \
    "v_add_u32_dpp v164, " OFFR ", %[pb]" OTHR AK_DSREAD("ds_read_b128 " XR1 ", v164\n\t")
#define DK_DOT(C0L, C0H, C1L, C1H)                                                                          \
    AK_PKMUL("v[156:157]", C0L, "v[124:125]") AK_PKMUL("v[158:159]", C1L, "v[128:129]")                    \
    AK_PKMUL("v[160:161]", C0L, "v[132:133]") AK_PKMUL("v[162:163]", C1L, "v[136:137]")                    \
    AK_PKFMA("v[156:157]", C0H, "v[126:127]") AK_PKFMA("v[158:159]", C1H, "v[130:131]")                    \
    AK_PKFMA("v[160:161]", C0H, "v[134:135]") AK_PKFMA("v[162:163]", C1H, "v[138:139]")
// partial sums; a; += the other pair's partials of the own row; a t - m; a ge + c; (OFFN); sums over the pair; l - m;
// e = dalpha + c; padding -> -inf; 2^(l - m); e / L (also the wait state of the exponential); dl
#define DK_RED(OWNC, OFFN, REALC)                                         \
    "v_add_f32 v156, v156, v157\n\t"                                      \
    "v_add_f32 v158, v158, v159\n\t"                                      \
    "v_add_f32 v160, v160, v161\n\t"                                      \
    "v_add_f32 v162, v162, v163\n\t"                                      \
    "v_mov_b32_dpp v106, %[val]" OWNC                                     \
    "v_add_f32_dpp v156, v158, v156" AK_QP(2, 3, 0, 1)                    \
    "v_add_f32_dpp v160, v162, v160" AK_QP(2, 3, 0, 1)                    \
    "v_fma_f32 v107, v106, v148, -v149\n\t"                               \
    "v_fma_f32 v157, v106, v151, v152\n\t" OFFN                           \
    "v_add_f32_dpp v156, v156, v156" AK_QP(1, 0, 3, 2)                    \
    "v_add_f32_dpp v160, v160, v160" AK_QP(1, 0, 3, 2)                    \
    "v_add_f32 v156, v156, v107\n\t"                                      \
    "v_add_f32 v160, v160, v157\n\t"                                      \
    "v_cndmask_b32_e64 v156, %[ninf], v156, " REALC "\n\t"                \
    "v_exp_f32_e32 v156, v156\n\t"                                        \
    "v_mul_f32 v160, v160, v150\n\t"                                      \
    "v_mul_f32 v156, v156, v160\n\t"
#define DK_ACC(C0L, C0H, C1L, C1H)                                        \
    "v_fmac_f32 v153, v156, v106\n\t"                                     \
    AK_PKACC("v[140:141]", "v[156:157]", C0L)                             \
    "v_mov_b32_dpp v166, v156" AK_QP(2, 3, 0, 1)                          \
    AK_PKACC("v[142:143]", "v[156:157]", C0H)                             \
    AK_PKACC("v[144:145]", "v[166:167]", C1L) AK_PKACC("v[146:147]", "v[166:167]", C1H)
#define DK_SITE(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)                          \
    "s_waitcnt lgkmcnt(2)\n\t" DK_DOT(C0L, C0H, C1L, C1H) DK_RED(OWNC, OFFN, REALC) DK_ACC(C0L, C0H, C1L, C1H) \
        DK_READS("v154", XR0, XR1, OWNR, OTHR, REALR)
#define DK_PRO(OFFA, XA0, XA1, OWNA, OTHA, REALA, OFFB, XB0, XB1, OWNB, OTHB, REALB)                       \
    OFFA "s_nop 1\n\t" DK_READS("v154", XA0, XA1, OWNA, OTHA, REALA) OFFB "s_nop 1\n\t" DK_READS("v154", XB0, XB1, OWNB, OTHB, REALB)
#define DK_E "v[108:109]", "v[110:111]", "v[112:113]", "v[114:115]"
#define DK_O "v[116:117]", "v[118:119]", "v[120:121]", "v[122:123]"

// ===== source-major backward =====  fixed registers (compiled for 94 VGPRs).  The item of an entry is the 160-byte record of
// its destination {q'[16], gv[16], t, rowmax, 1/L, ge, c}; the rows of the sweep are the sources (x_j, the running dx_j).
//   v94 a   v95 address of the record's scalars, then a t - rowmax   v117 LDS address
//   EVEN step: v[96:99] q' own  v[100:103] gv own  v[104:107] q' other  v[108:111] gv other  v[112:115] {t, rowmax, 1/L, ge}  v116 c
//   ODD step:  v[118:121]       v[122:125]         v[126:129]           v[130:133]           v[134:137]                      v138
//   v[140:143] v[144:147] x_j (own row, other row)      v[148:151] v[152:155] running dx_j
//   v[156:157] l-dot own -> l - rowmax -> 2^.. -> dl    v[158:159] l-dot other -> alpha    v[160:161] dalpha-dot own -> e -> dl of the other row
//   v[162:163] dalpha-dot other -> alpha of the other row        v164 offsets of the step to read
#define BK_DOT(Q0L, Q0H, G0L, G0H, Q1L, Q1H, G1L, G1H)                                                     \
    AK_PKMUL("v[156:157]", Q0L, "v[140:141]") AK_PKMUL("v[158:159]", Q1L, "v[144:145]")                    \
    AK_PKMUL("v[160:161]", G0L, "v[140:141]") AK_PKMUL("v[162:163]", G1L, "v[144:145]")                    \
    AK_PKFMA("v[156:157]", Q0H, "v[142:143]") AK_PKFMA("v[158:159]", Q1H, "v[146:147]")                    \
    AK_PKFMA("v[160:161]", G0H, "v[142:143]") AK_PKFMA("v[162:163]", G1H, "v[146:147]")
// T, RM, RI, GE, CC: the scalars of the own row's record of the computed step
#define BK_RED(OWNC, OFFN, REALC, T, RM, RI, GE, CC)                      \
    "v_add_f32 v156, v156, v157\n\t"                                      \
    "v_add_f32 v158, v158, v159\n\t"                                      \
    "v_add_f32 v160, v160, v161\n\t"                                      \
    "v_add_f32 v162, v162, v163\n\t"                                      \
    "v_mov_b32_dpp v94, %[val]" OWNC                                      \
    "v_add_f32_dpp v156, v158, v156" AK_QP(2, 3, 0, 1)                    \
    "v_add_f32_dpp v160, v162, v160" AK_QP(2, 3, 0, 1)                    \
    "v_fma_f32 v95, v94, " T ", -" RM "\n\t"                              \
    "v_fma_f32 v157, v94, " GE ", " CC "\n\t" OFFN                        \
    "v_add_f32_dpp v156, v156, v156" AK_QP(1, 0, 3, 2)                    \
    "v_add_f32_dpp v160, v160, v160" AK_QP(1, 0, 3, 2)                    \
    "v_add_f32 v156, v156, v95\n\t"                                       \
    "v_mul_f32 v156, %[k], v156\n\t"                                      \
    "v_cndmask_b32_e64 v156, %[ninf], v156, " REALC "\n\t"                \
    "v_exp_f32_e32 v156, v156\n\t"                                        \
    "v_add_f32 v160, v160, v157\n\t"                                      \
    "v_mul_f32 v158, v156, " RI "\n\t"                                    \
    "v_mul_f32 v156, v158, v160\n\t"
#define BK_ACC(Q0L, Q0H, G0L, G0H, Q1L, Q1H, G1L, G1H)                    \
    AK_PKACC("v[148:149]", "v[158:159]", G0L)                             \
    "v_mov_b32_dpp v162, v158" AK_QP(2, 3, 0, 1)                          \
    AK_PKACC("v[150:151]", "v[158:159]", G0H)                             \
    "v_mov_b32_dpp v160, v156" AK_QP(2, 3, 0, 1)                          \
    AK_PKACC("v[148:149]", "v[156:157]", Q0L) AK_PKACC("v[150:151]", "v[156:157]", Q0H)                    \
    AK_PKACC("v[152:153]", "v[162:163]", G1L) AK_PKACC("v[154:155]", "v[162:163]", G1H)                    \
    AK_PKACC("v[152:153]", "v[160:161]", Q1L) AK_PKACC("v[154:155]", "v[160:161]", Q1H)
// (the scalars of a record sit at its byte 128 for every lane: their address leaves out the lane's piece offset)
#define BK_READS(OFFR, RQ0, RG0, RSC, RC, RQ1, RG1, OWNR, OTHR, REALR)                                   \
    "v_add_u32_dpp v117, " OFFR ", %[pb]" OWNR AK_DSREAD("ds_read_b128 " RQ0 ", v117\n\tds_read_b128 " RG0 ", v117 offset:64\n\t") \
    "v_add_u32_dpp v95, " OFFR ", %[pbs]" OWNR AK_DSREAD("ds_read_b128 " RSC ", v95\n\tds_read_b32 " RC ", v95 offset:16\n\t")     \
    AK_REAL(REALR, "v117")                                                                                \
    "v_add_u32_dpp v117, " OFFR ", %[pb]" OTHR AK_DSREAD("ds_read_b128 " RQ1 ", v117\n\tds_read_b128 " RG1 ", v117 offset:64\n\t")
// register names of a step's items: E = even, O = odd
#define BK_RE "v[96:99]", "v[100:103]", "v[112:115]", "v116"
#define BK_RE1 "v[104:107]", "v[108:111]"
#define BK_RO "v[118:121]", "v[122:125]", "v[134:137]", "v138"
#define BK_RO1 "v[126:129]", "v[130:133]"
#define BK_CE "v[96:97]", "v[98:99]", "v[100:101]", "v[102:103]", "v[104:105]", "v[106:107]", "v[108:109]", "v[110:111]"
#define BK_CO "v[118:119]", "v[120:121]", "v[122:123]", "v[124:125]", "v[126:127]", "v[128:129]", "v[130:131]", "v[132:133]"
#define BK_SE "v112", "v113", "v114", "v115", "v116"
#define BK_SO "v134", "v135", "v136", "v137", "v138"
// (the packs above are passed by name and expand one level down)
#define BK_SITE(C, S, OWNC, REALC, OFFN, R, R1, OWNR, OTHR, REALR) BK_SITE_X(C, S, OWNC, REALC, OFFN, R, R1, OWNR, OTHR, REALR)
#define BK_SITE_X(Q0L, Q0H, G0L, G0H, Q1L, Q1H, G1L, G1H, T, RM, RI, GE, CC, OWNC, REALC, OFFN, RQ0, RG0, RSC, RC, RQ1, RG1, OWNR, OTHR, REALR) \
    "s_waitcnt lgkmcnt(6)\n\t" BK_DOT(Q0L, Q0H, G0L, G0H, Q1L, Q1H, G1L, G1H) BK_RED(OWNC, OFFN, REALC, T, RM, RI, GE, CC)                       \
        BK_ACC(Q0L, Q0H, G0L, G0H, Q1L, Q1H, G1L, G1H) BK_READS("v164", RQ0, RG0, RSC, RC, RQ1, RG1, OWNR, OTHR, REALR)
#define BK_PRO(OFFA, RA, RA1, OWNA, OTHA, REALA, OFFB, RB, RB1, OWNB, OTHB, REALB) BK_PRO_X(OFFA, RA, RA1, OWNA, OTHA, REALA, OFFB, RB, RB1, OWNB, OTHB, REALB)
#define BK_PRO_X(OFFA, AQ0, AG0, ASC, AC, AQ1, AG1, OWNA, OTHA, REALA, OFFB, BQ0, BG0, BSC, BC, BQ1, BG1, OWNB, OTHB, REALB) \
    OFFA "s_nop 1\n\t" BK_READS("v164", AQ0, AG0, ASC, AC, AQ1, AG1, OWNA, OTHA, REALA)                                    \
    OFFB "s_nop 1\n\t" BK_READS("v164", BQ0, BG0, BSC, BC, BQ1, BG1, OWNB, OTHB, REALB)

// forwarding macros: the F / D families take their computed pieces as one macro argument (FK_E, ...)
#define FK_SITE_(C, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR) FK_SITE_X(C, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)
#define FK_SITE_X(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR) FK_SITE(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)
#define DK_SITE_(C, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR) DK_SITE_X(C, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)
#define DK_SITE_X(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR) DK_SITE(C0L, C0H, C1L, C1H, OWNC, REALC, OFFN, XR0, XR1, OWNR, OTHR, REALR)

#define AK_IO_(CLOB) : [re] "+s"(c.real_e), [ro] "+s"(c.real_o), [seen] "+s"(c.seen) \
                     : [oa] "v"(oa), [ob] "v"(ob), [val] "v"(val), [pb] "v"(c.pb), [pz] "v"(c.pz), [pbs] "v"(c.pbs), [k] "s"(c.k), [ninf] "v"(c.ninf) : CLOB
// Site S (S counted from the first step of the register set's first group; S4 = S mod 4): arithmetic of step S - 1 (`val`: the
// lane's value of that step), reads of step S + 1 (`ob`: the lane's offset word of the group of step S + 1).
template <int MODE, int S4>
__device__ __forceinline__ void ak_site(int ob, int val, AkConst& c) {
    const int oa = 0;
    constexpr bool CEV = (S4 & 1) == 1;            // the computed step S - 1 (and the read step S + 1) is even
    constexpr bool COWN0 = ((S4 + 3) & 3) < 2;     // the computed step sits in lanes 0, 1 of the quad
    constexpr bool ROWN0 = ((S4 + 1) & 3) < 2;     // the read step ...
#define AK_EMIT(OWNC, OWNR, OTHR)                                                                                          \
    if constexpr (MODE == MODE_FWD) {                                                                                      \
        if constexpr (CEV) asm volatile(FK_SITE_(FK_E, OWNC, "%[re]", AK_OFF_EVEN("v156", "ob"), "v[120:123]", "v[124:127]", OWNR, OTHR, "%[re]") AK_IO_(AK_CLOBBER_F)); \
        else asm volatile(FK_SITE_(FK_O, OWNC, "%[ro]", AK_OFF_ODD("v156", "ob"), "v[128:131]", "v[132:135]", OWNR, OTHR, "%[ro]") AK_IO_(AK_CLOBBER_F));               \
    } else if constexpr (MODE == MODE_BDST) {                                                                              \
        if constexpr (CEV) asm volatile(DK_SITE_(DK_E, OWNC, "%[re]", AK_OFF_EVEN("v154", "ob"), "v[108:111]", "v[112:115]", OWNR, OTHR, "%[re]") AK_IO_(AK_CLOBBER_D)); \
        else asm volatile(DK_SITE_(DK_O, OWNC, "%[ro]", AK_OFF_ODD("v154", "ob"), "v[116:119]", "v[120:123]", OWNR, OTHR, "%[ro]") AK_IO_(AK_CLOBBER_D));               \
    } else {                                                                                                               \
        if constexpr (CEV) asm volatile(BK_SITE(BK_CE, BK_SE, OWNC, "%[re]", AK_OFF_EVEN("v164", "ob"), BK_RE, BK_RE1, OWNR, OTHR, "%[re]") AK_IO_(AK_CLOBBER_B)); \
        else asm volatile(BK_SITE(BK_CO, BK_SO, OWNC, "%[ro]", AK_OFF_ODD("v164", "ob"), BK_RO, BK_RO1, OWNR, OTHR, "%[ro]") AK_IO_(AK_CLOBBER_B));               \
    }
    if constexpr (COWN0 && ROWN0) { AK_EMIT(AK_OWN0, AK_OWN0, AK_OTH0) }
    else if constexpr (COWN0) { AK_EMIT(AK_OWN0, AK_OWN2, AK_OTH2) }
    else if constexpr (ROWN0) { AK_EMIT(AK_OWN2, AK_OWN0, AK_OTH0) }
    else { AK_EMIT(AK_OWN2, AK_OWN2, AK_OTH2) }
#undef AK_EMIT
}
// Start of a pass at step RA (0 .. 3 inside its group): the reads of steps RA and RA + 1.  `oa`: offset word of the group of
// step RA, `ob`: of step RA + 1 (the next group's when RA = 3).
template <int MODE, int RA>
__device__ __forceinline__ void ak_prologue(int oa, int ob, AkConst& c) {
    const int val = 0;
    constexpr bool AEV = (RA & 1) == 0;
#define AK_PA(M, E0, E1, O0, O1)                                                                                           \
    if constexpr (RA == 0) asm volatile(M(AK_OFF_EVEN(OFFREG, "oa"), E0, E1, AK_OWN0, AK_OTH0, "%[re]", AK_OFF_ODD(OFFREG, "ob"), O0, O1, AK_OWN0, AK_OTH0, "%[ro]") AK_IO_(CLOB)); \
    if constexpr (RA == 1) asm volatile(M(AK_OFF_ODD(OFFREG, "oa"), O0, O1, AK_OWN0, AK_OTH0, "%[ro]", AK_OFF_EVEN(OFFREG, "ob"), E0, E1, AK_OWN2, AK_OTH2, "%[re]") AK_IO_(CLOB)); \
    if constexpr (RA == 2) asm volatile(M(AK_OFF_EVEN(OFFREG, "oa"), E0, E1, AK_OWN2, AK_OTH2, "%[re]", AK_OFF_ODD(OFFREG, "ob"), O0, O1, AK_OWN2, AK_OTH2, "%[ro]") AK_IO_(CLOB)); \
    if constexpr (RA == 3) asm volatile(M(AK_OFF_ODD(OFFREG, "oa"), O0, O1, AK_OWN2, AK_OTH2, "%[ro]", AK_OFF_EVEN(OFFREG, "ob"), E0, E1, AK_OWN0, AK_OTH0, "%[re]") AK_IO_(CLOB));
    (void)AEV;
    if constexpr (MODE == MODE_FWD) {
#define OFFREG "v156"
#define CLOB AK_CLOBBER_F
        AK_PA(FK_PRO, "v[120:123]", "v[124:127]", "v[128:131]", "v[132:135]")
#undef OFFREG
#undef CLOB
    } else if constexpr (MODE == MODE_BDST) {
#define OFFREG "v154"
#define CLOB AK_CLOBBER_D
        AK_PA(DK_PRO, "v[108:111]", "v[112:115]", "v[116:119]", "v[120:123]")
#undef OFFREG
#undef CLOB
    } else {
#define OFFREG "v164"
#define CLOB AK_CLOBBER_B
#define BK_PA_(OA, RA_, RA1_, OWA, OTA, REA, OB, RB_, RB1_, OWB, OTB, REB) BK_PRO(OA, RA_, RA1_, OWA, OTA, REA, OB, RB_, RB1_, OWB, OTB, REB)
        if constexpr (RA == 0) asm volatile(BK_PRO(AK_OFF_EVEN(OFFREG, "oa"), BK_RE, BK_RE1, AK_OWN0, AK_OTH0, "%[re]", AK_OFF_ODD(OFFREG, "ob"), BK_RO, BK_RO1, AK_OWN0, AK_OTH0, "%[ro]") AK_IO_(CLOB));
        if constexpr (RA == 1) asm volatile(BK_PRO(AK_OFF_ODD(OFFREG, "oa"), BK_RO, BK_RO1, AK_OWN0, AK_OTH0, "%[ro]", AK_OFF_EVEN(OFFREG, "ob"), BK_RE, BK_RE1, AK_OWN2, AK_OTH2, "%[re]") AK_IO_(CLOB));
        if constexpr (RA == 2) asm volatile(BK_PRO(AK_OFF_EVEN(OFFREG, "oa"), BK_RE, BK_RE1, AK_OWN2, AK_OTH2, "%[re]", AK_OFF_ODD(OFFREG, "ob"), BK_RO, BK_RO1, AK_OWN2, AK_OTH2, "%[ro]") AK_IO_(CLOB));
        if constexpr (RA == 3) asm volatile(BK_PRO(AK_OFF_ODD(OFFREG, "oa"), BK_RO, BK_RO1, AK_OWN2, AK_OTH2, "%[ro]", AK_OFF_EVEN(OFFREG, "ob"), BK_RE, BK_RE1, AK_OWN0, AK_OTH0, "%[re]") AK_IO_(CLOB));
#undef BK_PA_
#undef OFFREG
#undef CLOB
    }
#undef AK_PA
}
#undef AK_IO_

// state of the lane's two rows: LDS -> the fixed registers, and back.  `own` / `oth`: byte offset of the lane's 16-byte piece
// of its own / the other row inside a [rows][64 B] array; `sc`: byte address of the own row's scalars.
template <int MODE>
__device__ __forceinline__ void ak_state_load(unsigned own, unsigned oth, unsigned sc) {
    using M = AM<MODE>;
    if constexpr (MODE == MODE_FWD)
        asm volatile("ds_read_b128 v[136:139], %0\n\tds_read_b128 v[140:143], %1\n\t"
                     "ds_read_b128 v[144:147], %2\n\tds_read_b128 v[148:151], %3\n\tds_read_b128 v[152:155], %4"
                     : : "v"(own + M::QA), "v"(oth + M::QA), "v"(own + M::ZA), "v"(oth + M::ZA), "v"(sc) : AK_CLOBBER_F);
    else if constexpr (MODE == MODE_BDST)
        asm volatile("ds_read_b128 v[124:127], %0\n\tds_read_b128 v[128:131], %1\n\t"
                     "ds_read_b128 v[132:135], %2\n\tds_read_b128 v[136:139], %3\n\t"
                     "ds_read_b128 v[140:143], %4\n\tds_read_b128 v[144:147], %5\n\t"
                     "ds_read_b128 v[148:151], %6\n\tds_read_b64 v[152:153], %6 offset:16"
                     : : "v"(own + M::QA), "v"(oth + M::QA), "v"(own + M::GA), "v"(oth + M::GA), "v"(own + M::DA), "v"(oth + M::DA), "v"(sc)
                     : AK_CLOBBER_D);
    else
        asm volatile("ds_read_b128 v[140:143], %0\n\tds_read_b128 v[144:147], %1\n\t"
                     "ds_read_b128 v[148:151], %2\n\tds_read_b128 v[152:155], %3"
                     : : "v"(own + M::XA), "v"(oth + M::XA), "v"(own + M::DA), "v"(oth + M::DA) : AK_CLOBBER_B);
}
template <int MODE>
__device__ __forceinline__ void ak_state_store(unsigned own, unsigned oth, unsigned sc) {
    using M = AM<MODE>;
    if constexpr (MODE == MODE_FWD)
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b128 %0, v[144:147]\n\tds_write_b128 %1, v[148:151]\n\t"
                     "ds_write_b64 %2, v[152:153]"
                     : : "v"(own + M::ZA), "v"(oth + M::ZA), "v"(sc) : AK_CLOBBER_F);
    else if constexpr (MODE == MODE_BDST)
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b128 %0, v[140:143]\n\tds_write_b128 %1, v[144:147]\n\t"
                     "ds_write_b32 %2, v153 offset:20"
                     : : "v"(own + M::DA), "v"(oth + M::DA), "v"(sc) : AK_CLOBBER_D);
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b128 %0, v[148:151]\n\tds_write_b128 %1, v[152:155]"
                     : : "v"(own + M::DA), "v"(oth + M::DA) : AK_CLOBBER_B);
}

// Static sites of a pass (compile-time recursion: the entry registers must be indexed by constants).  `cur` holds GS K
// steps from the group of the pass's first step on; the pass walks steps [ra, rb) of them, 0 <= ra < GS, ra < rb <= GS K:
// sites ra + 1 .. rb.  hook(S) runs for every S = 0 .. GS K whether site S is walked or not: what it issues (the next block's
// entries, one conditional load per site) sits at fixed program points of the walk.  The chain is LEFT behind the last step (it
// returns the S it stopped at) and ak_tail runs the hooks from there on, one copy for all exits -- every instruction of a
// wavefront costs an issue slot, scalar ones too.
template <int S, int K, class FH>
__device__ __forceinline__ void ak_tail(int from, FH&& hook) {
    if constexpr (S <= 4 * K) {
        if (S >= from) hook(std::integral_constant<int, S>());      // (the hooks that do nothing fold away with their test)
        ak_tail<S + 1, K>(from, hook);
    }
}
template <int MODE, int S, int K, class Cur, class FH>
__device__ __forceinline__ int ak_sites(int ra, int rb, const Cur& cur, AkConst& c, FH&& hook) {
    constexpr int GS = 4, N = GS * K;
    if constexpr (S <= N) {
        constexpr int gr = (S + 1 < N ? S + 1 : N - 1) / GS, gp = (S - 1) / GS;
        // (ONE call of the next level per level: two would double the code of everything behind them)
        if (S > GS || S > ra) {             // (the first group: the pass starts at step ra, its first site is ra + 1)
            if (__builtin_expect(S > rb, 0)) return S;
            ak_site<MODE, S & 3>(cur[gr].o, ((S - 1) & 1) ? cur[gp].v1 : cur[gp].v0, c);
        }
        hook(std::integral_constant<int, S>());
        return ak_sites<MODE, S + 1, K>(ra, rb, cur, c, hook);
    } else {
        return S;
    }
}
// the whole pass (rb > ra): the reads of the first two steps, the walk, then the hooks behind its end
template <int MODE, int K, class Cur, class FH, class FT>
__device__ __forceinline__ void ak_walk(int ra, int rb, const Cur& cur, AkConst& c, FH&& hook, FT&& tick) {
    if (rb > ra) {
        const int o0 = cur[0].o, o1 = cur[K > 1 ? 1 : 0].o;
        if (ra == 0) ak_prologue<MODE, 0>(o0, o0, c);
        else if (ra == 1) ak_prologue<MODE, 1>(o0, o0, c);
        else if (ra == 2) ak_prologue<MODE, 2>(o0, o0, c);
        else ak_prologue<MODE, 3>(o0, o1, c);
    }
    tick(2);
    hook(std::integral_constant<int, 0>());
    const int ex = rb > ra ? ak_sites<MODE, 1, K>(ra, rb, cur, c, hook) : 1;
    tick(3);
    ak_tail<1, K>(ex, hook);
    tick(4);
}

// (the kernels below declare the LDS block and call this; the asm addresses LDS by absolute byte offsets, so the block must
// be the kernel's only __shared__ object: checked at run time, one scalar compare)
// STAMP (timing build only): cycle counters per phase instead of the results (tools/attn_stream_cycles.py)
template <int MODE, bool STAMP = false>
__device__ __forceinline__ void attn_stream_body(const AttnStreamDev& t, const AttnArgs& a, char* smem) {
    using M = AM<MODE>;
    using G = typename M::G;
    if (__builtin_expect((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u, 0)) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int tb0 = t.tile_blk[tile], nb = t.tile_blk[tile + 1] - tb0;
    const int row0 = t.tile_row[tile];
    const int n_rows = t.tile_row[tile + 1] - row0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned cyc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    unsigned last_ = STAMP ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const unsigned start_ = last_;
#define AK_TICK(K)                                                      \
    if (STAMP) {                                                        \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();   \
        cyc[K] += now_ - last_;                                         \
        last_ = now_;                                                   \
    }

    // (detail variant of the stamps, -DMLLP_AK_STAMP_DETAIL: [1] top of the block, [2] start of a pass (addresses, state
    // loads, first reads), [3] the chain of sites, [4] hooks behind the walk + refills, [5] end of a pass (check, state
    // store) + wait for the prefetch + barrier)
#ifdef MLLP_AK_STAMP_DETAIL
#define AK_TICKC(K)
#define AK_TICKD(K) AK_TICK(K)
#else
#define AK_TICKC(K) AK_TICK(K)
#define AK_TICKD(K)
#endif

    // ---- the tile's state ----
    if constexpr (MODE == MODE_FWD) {       // q' and t in log2 units, Z = 0, {L, u, m} = 0
        const float4* qsrc = reinterpret_cast<const float4*>(a.qp + (size_t)row0 * 16);
        float4* Q = reinterpret_cast<float4*>(smem + M::QA);
        float4* Zs = reinterpret_cast<float4*>(smem + M::ZA);
        float4* St = reinterpret_cast<float4*>(smem + M::SA);
        for (int i = tid; i < A_RS * 4; i += A_THREADS) {
            float4 q = zero4;
            if ((i >> 2) < n_rows) {
                q = qsrc[i];
                q.x *= A_LOG2E; q.y *= A_LOG2E; q.z *= A_LOG2E; q.w *= A_LOG2E;
            }
            Q[i] = q;
            Zs[i] = zero4;
        }
        for (int r = tid; r < A_RS; r += A_THREADS)
            St[r] = make_float4(0.f, 0.f, 0.f, r < n_rows ? a.tq[row0 + r] * A_LOG2E : 0.f);
    } else if constexpr (MODE == MODE_BDST) {      // from the records: q', t, rowmax in log2 units; gv; {1/L, ge, c}; dq' = dt = 0
        float4* Q = reinterpret_cast<float4*>(smem + M::QA);
        float4* Gv = reinterpret_cast<float4*>(smem + M::GA);
        float4* Dq = reinterpret_cast<float4*>(smem + M::DA);
        float4* St = reinterpret_cast<float4*>(smem + M::SA);
        for (int i = tid; i < A_RS * 4; i += A_THREADS) {
            float4 q = zero4, gv = zero4;
            const int r = i >> 2;
            if (r < n_rows) {
                const float4* rp = reinterpret_cast<const float4*>(a.rec + (size_t)(row0 + r) * REC_W);
                q = rp[i & 3];
                gv = rp[4 + (i & 3)];
                q.x *= A_LOG2E; q.y *= A_LOG2E; q.z *= A_LOG2E; q.w *= A_LOG2E;
            }
            Q[i] = q;
            Gv[i] = gv;
            Dq[i] = zero4;
        }
        for (int r = tid; r < A_RS; r += A_THREADS) {
            float4 s0 = zero4;
            float cc = 0.f;
            if (r < n_rows) {
                const float* rp = a.rec + (size_t)(row0 + r) * REC_W;
                s0 = *reinterpret_cast<const float4*>(rp + 32);      // {t, rowmax, 1/L, ge}
                s0.x *= A_LOG2E; s0.y *= A_LOG2E;
                cc = rp[36];
            }
            St[2 * r] = s0;
            St[2 * r + 1] = make_float4(cc, 0.f, 0.f, 0.f);          // {c, dt, -, -}
        }
    } else {                                        // x_j of the tile's rows, dx_j = 0
        const float4* xsrc = reinterpret_cast<const float4*>(a.x_rows + (size_t)row0 * 16);
        float4* Xs = reinterpret_cast<float4*>(smem + M::XA);
        float4* Dx = reinterpret_cast<float4*>(smem + M::DA);
        for (int i = tid; i < A_RS * 4; i += A_THREADS) {
            Xs[i] = (i >> 2) < n_rows ? xsrc[i] : zero4;
            Dx[i] = zero4;
        }
    }
    for (int i = tid; i < 2 * (G::ITEM / 16); i += A_THREADS)      // the all-zero item behind each image
        *reinterpret_cast<float4*>(smem + (i / (G::ITEM / 16)) * M::IMG + G::ZERO_OFF + (i % (G::ITEM / 16)) * 16) = zero4;

    if (wave >= G::NW) {
        // ------------------------------------------------ stagers -------------------------------------------------
        constexpr int PIECES = (G::CB * G::ITEM + 1023) / 1024, PPW = (PIECES + A_STAGERS - 1) / A_STAGERS;
        const int d = wave - G::NW;
        const i32x4* hp = t.hdr + (size_t)tb0 * G::NW;          // any wavefront's header carries the block id
        auto stage = [&](int b, int img, bool paced) {
            const int c0 = __builtin_amdgcn_readfirstlane(hp[G::NW * b].z) * G::CB;
            const int nbytes = min(G::CB, t.n_src - c0) * G::ITEM;
            const char* src = reinterpret_cast<const char*>(a.items) + (size_t)c0 * G::ITEM + lane * 16;
            char* img_base = smem + img * M::IMG;
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int piece = d + A_STAGERS * i;
                if (piece * 1024 + lane * 16 < nbytes && !(AK_ABL & 4))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + piece * 1024),
                        (__attribute__((address_space(3))) void*)(img_base + piece * 1024), 16, 0, 0);
                if (paced && !(AK_ABL & 32)) __builtin_amdgcn_s_sleep(MLLP_AK_SLEEP);      // a burst would block the walkers in their own loads
            }
        };
        if (nb > 0) stage(0, 0, false);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        AK_TICK(0)
        for (int k = 0; k < nb; ++k) {
            if (k + 1 < nb) stage(k + 1, (k + 1) & 1, true);
            AK_TICK(1)
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the pieces have landed
            AK_TICK(2)
            __syncthreads();
            AK_TICK(3)
        }
    } else {
        // ------------------------------------------------ walkers -------------------------------------------------
        const int quad = lane >> 2, part = lane & 3;
        const i32x4* rowp = t.rows + ((size_t)tb0 * G::NW + wave) * 16 + quad;     // + 16 NW per block
        const i32x4* hdrp = t.hdr + (size_t)tb0 * G::NW + wave;                    // + NW per block (wave-uniform)
        i32x4 rc = {0, 0, 0, 0}, hc = {0, 0, 0, 0}, hn = {0, 0, 0, 0};
        if (nb > 0) {
            rc = __builtin_nontemporal_load(rowp);
            hc = hdrp[0];
            hn = hdrp[G::NW * min(1, nb - 1)];
        }
        Ent3 ea[G::K0], eb[G::K1], eb2[G::K1];     // pass 1 has two sets: the next block's is loaded during pass 0
        auto ld3 = [&](const Ent3* p) {
            Ent3 e;
            e.o = __builtin_nontemporal_load(&p->o);
            e.v0 = __builtin_nontemporal_load(&p->v0);
            e.v1 = __builtin_nontemporal_load(&p->v1);
            return e;
        };
        auto set_base = [&](const i32x4& h, int pass) {
            const int S = __builtin_amdgcn_readfirstlane(h.x), n0 = __builtin_amdgcn_readfirstlane(h.y) & 0xffff;
            return t.ent + (size_t)((S + (pass ? n0 : 0)) / G::GS) * 64 + lane;
        };
        if (nb > 0) {
            const Ent3* p = set_base(hc, 0);
            const Ent3* q = set_base(hc, 1);
#pragma unroll
            for (int j = 0; j < G::K0; ++j) ea[j] = ld3(p + 64 * j);
#pragma unroll
            for (int j = 0; j < G::K1; ++j) eb[j] = ld3(q + 64 * j);
        }

        // Forward only.  Exact pass, compiled: lanes 0, 1 of a quad walk steps [a, b) of the quad's rows 0, 1 with the
        // running maximum (rebasing m).  Runs after the fast pass found |l - m| > 64 (and left the state untouched).
        auto slow_pass = [&](int rr, int a0, int b0, unsigned img) {
            if constexpr (MODE == MODE_FWD) {
                if (part < 2) {
                    // (few registers on purpose: the row's q' and Z stay in LDS, 16 bytes at a time -- this path is rare and
                    // shares the kernel's 120 registers with the entry sets of the fast path)
                    const int row = min((rr >> (16 * part)) & 0xffff, G::RR);
                    const float4* Q = reinterpret_cast<const float4*>(smem + M::QA) + row * 4;
                    float4* Zs = reinterpret_cast<float4*>(smem + M::ZA) + row * 4;
                    float4* St = reinterpret_cast<float4*>(smem + M::SA) + row;
                    const float4 st = *St;
                    float L = st.x, u = st.y, m = st.z;
                    const float tq = st.w;
                    for (int s = a0; s < b0; ++s) {
                        const int* e = reinterpret_cast<const int*>(t.ent) + G::ent_index(s, quad, part);
                        const int off = (s & 1) ? (int)((unsigned)e[0] >> 16) : (e[0] & 0xffff);
                        if (off == G::ZERO_OFF) continue;
                        const float av = __int_as_float(e[1 + (s & 1)]);
                        const float4* xs = reinterpret_cast<const float4*>(smem + img + off);
                        float l = av * tq;
#pragma unroll 1
                        for (int i = 0; i < 4; ++i) l += dot4(Q[i], xs[i]);
                        float cs = 1.0f;
                        if (L == 0.0f) {
                            m = l;
                        } else if (l > m) {
                            cs = __builtin_amdgcn_exp2f(m - l);
                            L *= cs; u *= cs;
                            m = l;
                        }
                        const float p = __builtin_amdgcn_exp2f(l - m);
                        L += p;
                        u = fmaf(p, av, u);
#pragma unroll 1
                        for (int i = 0; i < 4; ++i) {
                            float4 z = Zs[i];
                            const float4 x = xs[i];
                            z.x = fmaf(p, x.x, z.x * cs); z.y = fmaf(p, x.y, z.y * cs);
                            z.z = fmaf(p, x.z, z.z * cs); z.w = fmaf(p, x.w, z.w * cs);
                            if (row < G::RR) Zs[i] = z;
                        }
                    }
                    if (row < G::RR) *St = make_float4(L, u, m, tq);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): no compiler-tracked LDS operation stays pending
            }
        };

        // One pass: steps [a, b) (absolute) of the quad's two rows (rr = row 0 | row 1 << 16); `cur` holds GS K steps from
        // the group of step a on; `np`: where the groups of the same pass of the NEXT block start (group j is reloaded
        // behind its last step); `extra(site)`: what else the hooks issue.
        auto pass = [&](auto& cur, const Ent3* np, int gn, auto kk, int rr, int a0, int b0, unsigned img, auto&& extra) {
            constexpr int K = decltype(kk)::value;
            const int ra = a0 % G::GS, rb = b0 - (a0 - ra);
            const bool any = rb > ra;                                                   // wave-uniform
            // (row slots above the tile's last row sort last and have no entries: their lanes work on the dummy state row)
            const int r_own = min((rr >> (16 * (part >> 1))) & 0xffff, G::RR), r_oth = min((rr >> (16 * ((part >> 1) ^ 1))) & 0xffff, G::RR);
            const unsigned own = (unsigned)((r_own * 4 + part) * 16), oth = (unsigned)((r_oth * 4 + part) * 16);
            unsigned sc = 0;
            if constexpr (MODE == MODE_FWD) sc = (unsigned)(M::SA + r_own * 16);
            if constexpr (MODE == MODE_BDST) sc = (unsigned)(M::SA + r_own * 32);
            AkConst c;
            c.pb = img + part * 16; c.pz = c.pb + G::ZERO_OFF; c.pbs = img + 128; c.ninf = (int)0xff800000u;
            c.k = A_LOG2E;
            c.real_e = c.real_o = c.seen = 0ull;
            asm volatile("" : "+v"(c.ninf), "+v"(c.pbs));     // (opaque: they stay in VGPRs instead of being re-made per site)
            if (any && !(AK_ABL & 8)) ak_state_load<MODE>(own, oth, sc);
            ak_walk<MODE, K>(ra, any ? min(rb, G::GS * K) : ra, cur, c, [&](auto hc_) {
                constexpr int site = decltype(hc_)::value;
                constexpr int h = site - G::GS;             // site of the last arithmetic of group h / GS
                if constexpr (h >= 0 && h % G::GS == 0 && h / G::GS < K) {
                    if (__builtin_expect(h / G::GS < gn, 1) && !(AK_ABL & 16)) cur[h / G::GS] = ld3(np + 64 * (h / G::GS));
                }
                extra(hc_);
            }, [&](int k_) { (void)k_; AK_TICKD(k_) });
            if (__builtin_expect(rb > G::GS * K, 0)) {
                // rare: the pass is longer than its register set (a row with dozens of entries inside one block): the rest
                // runs on groups that are loaded here, K1 at a time -- the same sites, the state stays in the registers
                const Ent3* base = t.ent + (size_t)(a0 / G::GS) * 64 + lane;
                for (int s0 = G::GS * K; s0 < rb; s0 += G::GS * G::K1) {
                    Ent3 tmp[G::K1];
#pragma unroll
                    for (int j = 0; j < G::K1; ++j) tmp[j] = ld3(base + 64 * (s0 / G::GS + j));
                    ak_walk<MODE, G::K1>(0, min(rb - s0, G::GS * G::K1), tmp, c, [&](auto) {}, [&](int) {});
                }
            }
            bool redo = false;
            if constexpr (MODE == MODE_FWD) {
                // Did some l - m leave the window of the fast pass?  Not checked per step: a p = 2^(l - m) above 2^64 shows in L
                // (L > 2^63 or inf; a slow pass leaves L <= the row's length, the fast passes behind it add p < 2^64 each), and a
                // row ALL of whose p underflowed shows as L < 2^-63 with a real entry seen in this pass (single tiny p next to
                // normal ones are below fp32's resolution of L anyway).  A NaN in L fails both tests: it belongs to the data.
                unsigned long long hi, lo;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_cmp_gt_f32_e64 %0, v152, %2\n\tv_cmp_lt_f32_e64 %1, v152, %3\n\t"
                             "s_and_b64 %1, %1, %4"
                             : "=&s"(hi), "=&s"(lo) : "s"(9.223372e18f), "s"(1.0842022e-19f), "s"(c.seen) : AK_CLOBBER_F);
                redo = any && (hi | lo) != 0ull && AK_ABL == 0;       // (an ablated walk computes on garbage: no redo)
            }
            if (any && !redo && !(AK_ABL & 8)) ak_state_store<MODE>(own, oth, sc);
            if (any && redo) {
                asm volatile("; SK_SLOW_BEGIN" : : : "memory");
                slow_pass(rr, a0, b0, img);
                asm volatile("; SK_SLOW_END" : : : "memory");
            }
        };

        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the first entries have landed
        __syncthreads();                         // image 0 and the tile's state are in LDS
        AK_TICK(0)
        auto block = [&](int k, auto& ebc, auto& ebn) {
            const i32x4 rn = __builtin_nontemporal_load(rowp + 16 * G::NW * min(k + 1, nb - 1));
            const i32x4 h2 = hdrp[G::NW * min(k + 2, nb - 1)];
            const Ent3* npa = set_base(hn, 0);
            const Ent3* npb = set_base(hn, 1);
            const int nS_ = __builtin_amdgcn_readfirstlane(hn.x);
            const unsigned nc_ = (unsigned)__builtin_amdgcn_readfirstlane(hn.y);
            const bool more = k + 1 < nb;
            const int gna = more ? (nS_ % G::GS + (int)(nc_ & 0xffffu) + G::GS - 1) / G::GS : 0;
            const int gnb = more ? ((nS_ + (int)(nc_ & 0xffffu)) % G::GS + (int)(nc_ >> 16) + G::GS - 1) / G::GS : 0;
            const int S_ = __builtin_amdgcn_readfirstlane(hc.x);
            const unsigned c_ = (unsigned)__builtin_amdgcn_readfirstlane(hc.y);
            const int n0_ = (int)(c_ & 0xffffu), n1_ = (int)(c_ >> 16);
            const unsigned img = (unsigned)((k & 1) * M::IMG);
            AK_TICK(1)
            pass(ea, npa, gna, std::integral_constant<int, G::K0>(), rc.x, S_, S_ + n0_, img, [&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                if constexpr (s % 2 == 0 && s / 2 < G::K1) {     // the next block's pass-1 entries, one group per even site
                    if (__builtin_expect(s / 2 < gnb, 1) && !(AK_ABL & 16)) ebn[s / 2] = ld3(npb + 64 * (s / 2));
                }
            });
            AK_TICKC(2)
            pass(ebc, npb, 0, std::integral_constant<int, G::K1>(), rc.z, S_ + n0_, S_ + n0_ + n1_, img, [&](auto) {});
            AK_TICKC(3)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next entries have landed
            rc = rn; hc = hn; hn = h2;
            AK_TICKC(4)
            __syncthreads();
            AK_TICK(5)
        };
        for (int k = 0; k < nb; ++k) {       // (one copy of the block body: the pass-1 sets change roles by a register copy)
            block(k, eb, eb2);
#pragma unroll
            for (int j = 0; j < G::K1; ++j) eb[j] = eb2[j];
        }
    }
    // (the last barrier of the loop made every wavefront's state visible)
    if constexpr (STAMP) {
        // stamps instead of the result: row 2 * tile of the output = the walkers' sums {[0] init, [1] top of the block, [2] pass
        // 0, [3] pass 1, [4] wait for the prefetch, [5] barrier, [6] total}, row 2 * tile + 1 = the stagers' {[0] init, [1] issue,
        // [2] wait for the pieces, [3] barrier, [6] total}
        const unsigned total_ = (unsigned)__builtin_amdgcn_s_memtime() - start_;
        __syncthreads();
        int* acc = reinterpret_cast<int*>(smem);
        if (tid < 32) acc[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            const int o = wave >= G::NW ? 16 : 0;
            for (int k = 0; k < 6; ++k) atomicAdd(&acc[o + k], (int)cyc[k]);
            atomicAdd(&acc[o + 6], (int)total_);
        }
        __syncthreads();
        float* out = MODE == MODE_FWD ? a.h : MODE == MODE_BDST ? a.dqp : a.dx;
        if (tid < 32 && 2 * tile + 1 < t.n_dst) out[(size_t)(2 * tile) * 16 + tid] = (float)acc[tid];
        return;
    }
#undef AK_TICK

    if constexpr (MODE == MODE_FWD) {
        // ---- epilogue: 16 lanes per row (lane gl = output channel) ----
        const int gl = tid & 15;
        float wv[16], ws[16];
        load_row16(a.p.Wv + gl * 16, wv);
        load_row16(a.p.Ws + gl * 16, ws);
        const float b_s = a.p.bs[gl], b_v = a.p.bv[gl], w_e = a.p.we[gl];
        const float4* Zs = reinterpret_cast<const float4*>(smem + M::ZA);
        const float4* St = reinterpret_cast<const float4*>(smem + M::SA);
        for (int r = tid >> 4; r < n_rows; r += A_THREADS / 16) {
            const float4 st = St[r];      // {L, u, m, t}
            const float rinv = st.x > 0.0f ? 1.0f / st.x : 1.0e16f;
            const float S = st.x > 0.0f ? 1.0f : 0.0f, un = st.y * rinv;
            float zn[16], xr[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 zq = Zs[r * 4 + q];
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
            // saved for the backward sweeps: alpha_ij = exp(l_ij - rowmax) * rinv with rowmax = the reference in natural units
            if (gl == 0) reinterpret_cast<float4*>(a.aux)[row] = make_float4(un, st.z * A_LN2, rinv, S);
        }
    } else if constexpr (MODE == MODE_BDST) {
        // ---- epilogue: 16 lanes per row (lane gl = input channel):  dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt ----
        // (ds_i = sum_j dl_ij = D_i (1 - S_i) is 0 up to 1e-16 for every row with an entry and for the others: written as 0)
        const int gl = tid & 15;
        const float* D = a.derived;
        float wsT[16], pqT[16];
        float pt = 0.f;
        if (a.dx) {
            load_row16(D + OFF_WST + gl * 16, wsT);
            load_row16(D + OFF_PQT + gl * 16, pqT);
            pt = D[OFF_PT + gl];
        }
        const float4* Dq = reinterpret_cast<const float4*>(smem + M::DA);
        const float4* St = reinterpret_cast<const float4*>(smem + M::SA);
        for (int r = tid >> 4; r < n_rows; r += A_THREADS / 16) {
            const size_t row = (size_t)row0 + r;
            const float dt = St[2 * r + 1].y;
            float da[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = Dq[r * 4 + q];
                da[4 * q] = v.x; da[4 * q + 1] = v.y; da[4 * q + 2] = v.z; da[4 * q + 3] = v.w;
            }
            a.dqp[row * 16 + gl] = select16(da, gl);
            if (gl == 0) reinterpret_cast<float2*>(a.dsdt)[row] = make_float2(0.0f, dt);
            if (a.dx) {
                float gr[16];
                load_row16(a.g + row * 16, gr);
                float v = dot16(wsT, gr, 0.0f);
                v = fmaf(dt, pt, v);
                v = dot16(pqT, da, v);
                float* dst = a.dx + row * 16 + gl;
                *dst = a.accumulate ? *dst + v : v;
            }
        }
    } else {
        const float4* Dx = reinterpret_cast<const float4*>(smem + M::DA);
        float4* dst = reinterpret_cast<float4*>(a.dx + (size_t)row0 * 16);
        for (int i = tid; i < n_rows * 4; i += A_THREADS) {
            float4 v = Dx[i];
            if (a.accumulate) {
                const float4 o = dst[i];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            dst[i] = v;
        }
    }
}

// (amdgpu_num_vgpr takes a literal: one kernel per sweep around the common body)
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(120))) void fwd16_stream_kernel(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_FWD>::LDS + 256];
    attn_stream_body<MODE_FWD>(t, a, smem);
}
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(106))) void bwddst16_stream_kernel(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_BDST>::LDS + 256];
    attn_stream_body<MODE_BDST>(t, a, smem);
}
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(94))) void bwdsrc16_stream_kernel(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_BSRC>::LDS + 256];
    attn_stream_body<MODE_BSRC>(t, a, smem);
}
static_assert(AM<MODE_FWD>::VGPRS == 120 && AM<MODE_BDST>::VGPRS == 106 && AM<MODE_BSRC>::VGPRS == 94, "the literals above");
#ifdef MLLP_TIMING_BUILD
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(120))) void fwd16_stream_stamps(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_FWD>::LDS + 256];
    attn_stream_body<MODE_FWD, true>(t, a, smem);
}
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(106))) void bwddst16_stream_stamps(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_BDST>::LDS + 256];
    attn_stream_body<MODE_BDST, true>(t, a, smem);
}
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_num_vgpr(94))) void bwdsrc16_stream_stamps(AttnStreamDev t, AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[AM<MODE_BSRC>::LDS + 256];
    attn_stream_body<MODE_BSRC, true>(t, a, smem);
}
static bool attn_stamps() { const char* e = getenv("MLLP_ATTN_STAMPS"); return e && atoi(e) != 0; }
#define AK_LAUNCH(K, KS, ...) hipLaunchKernelGGL(attn_stamps() ? KS : K, __VA_ARGS__)
#else
#define AK_LAUNCH(K, KS, ...) hipLaunchKernelGGL(K, __VA_ARGS__)
#endif

AttnStreamDev stream_dev(const StreamCopy& sc, int n_dst, int n_src) {
    AttnStreamDev t;
    t.tile_row = sc.tile_row;
    t.tile_blk = sc.tile_blk;
    t.rows = reinterpret_cast<const i32x4*>(sc.rows);
    t.hdr = reinterpret_cast<const i32x4*>(sc.hdr);
    t.ent = reinterpret_cast<const Ent3*>(sc.ent);
    t.n_tiles = sc.n_tiles;
    t.n_dst = n_dst;
    t.n_src = n_src;
    return t;
}

}  // namespace

int launch_fwd16_stream(const StreamCopy& sc, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                        const float* x_src, const float* x_dst, float* h_out, hipStream_t s) {
    if (n_dst == 0 || sc.n_tiles == 0) return MLLP_OK;
    AttnArgs a = {};
    a.items = x_src; a.xd = x_dst; a.qp = w.qp; a.tq = w.t; a.p = conv_params_at(conv_params, 16);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    AK_LAUNCH(fwd16_stream_kernel, fwd16_stream_stamps, dim3((unsigned)sc.n_tiles), dim3(A_THREADS), 0, s, stream_dev(sc, n_dst, n_src), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "fwd16_stream");
}

int launch_bwddst16_stream(const StreamCopy& sc, int n_dst, int n_src, const ConvWs& w, const float* x_src, const float* g,
                           float* dx_dst, int accumulate, hipStream_t s) {
    if (n_dst == 0 || sc.n_tiles == 0) return MLLP_OK;
    AttnArgs a = {};
    a.items = x_src; a.rec = w.rec; a.g = g; a.derived = w.derived; a.dqp = w.dqp; a.dsdt = w.dsdt; a.dx = dx_dst;
    a.accumulate = accumulate;
    AK_LAUNCH(bwddst16_stream_kernel, bwddst16_stream_stamps, dim3((unsigned)sc.n_tiles), dim3(A_THREADS), 0, s, stream_dev(sc, n_dst, n_src), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwddst16_stream");
}

// rows of the sweep = the conv's SOURCE nodes (n_rows), columns = its destinations (n_cols records)
int launch_bwdsrc16_stream(const StreamCopy& sc, int n_rows, int n_cols, const float* rec, const float* x_rows, float* dx,
                           int accumulate, hipStream_t s) {
    if (n_rows == 0 || sc.n_tiles == 0) return MLLP_OK;
    AttnArgs a = {};
    a.items = rec; a.x_rows = x_rows; a.dx = dx; a.accumulate = accumulate;
    AK_LAUNCH(bwdsrc16_stream_kernel, bwdsrc16_stream_stamps, dim3((unsigned)sc.n_tiles), dim3(A_THREADS), 0, s, stream_dev(sc, n_rows, n_cols), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwdsrc16_stream");
}

}  // namespace mllp
