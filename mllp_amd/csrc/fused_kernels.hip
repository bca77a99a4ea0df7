// fused_kernels.hip -- the latency-regime path of the whole model (batches below 32 M nonzeros: real Netlib).
//
// Why a second set of kernels: on the Netlib batch (1.07 M nonzeros, 366 k nodes, median row length 2-4) a training
// step of the generic path is ~45 launches whose sweeps are bound by dependent memory round trips at low occupancy
// (profiles/r01_netlib_kernel_stats.md: 91 us for a 1 M-nonzero attention sweep, 68 % of wave time in s_waitcnt), and
// per conv it writes and re-reads q', t, dq', ds/dt and runs a separate statistics pass.  Here a conv is ONE sweep
// launch forward (q' = Pq x, the attention sweep, the output GEMMs, ReLU -- and for the last conv fc +
// BCEWithLogits + dL/dh) and TWO backward (destination-major: ReLU mask, gv = Wv^T g, record, sweep, input gradient,
// parameter statistics on the MFMA; source-major: the transposed sweep); the two convs of a layer share a launch.
// reference: linear_program_methods.py:238-251, linear_program_experiment.py:139-141; formulas SURVEY.md appendix
// A.3 / A.4 (oracle/spmm_form.py).
//
// Data: the path works in RENUMBERED node ids (host_graph.h::HostFusedOrient): constraints ordered by row length,
// variables by column length, descending.  Row k of a sweep is node k, so every per-node tensor of the model is
// read and written with unit stride, the rows a wavefront shares have (nearly) equal length, and the row tiers are
// contiguous id ranges.  Both orientations are stored once in that numbering as {source id, value} pairs; the
// inputs (coefs, rhs, labels) are copied into it when they are bound, and because layer 1's node features ARE the
// inputs, its per-nonzero source feature is stored beside the value ({a_ij, x_j}): the layer-1 sweeps gather
// nothing.  Only the logits leave in the caller's variable order.
//
// Mapping: persistent workgroups of 1024 threads, one per CU.  A WAVEFRONT takes an item = 16 rows, one QUAD of
// lanes per row (lane `part` owns channels 4 part .. 4 part + 3; a gather of a 64-byte source row is one 16-byte
// load per lane of the quad): 16 rows and 64 gathers in flight per wavefront, and the entries of the next four
// nonzeros are fetched while the current four are gathered.  Rows longer than 16 nonzeros take 4 quads, longer than
// 64 a whole wavefront, longer than 1024 the whole workgroup (states merged by DPP / shuffles / LDS).  Scalar
// (layer-1) sweeps use one LANE per row.  The 16x16 per-node GEMMs and the parameter statistics run on the MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32) through LDS tiles that convert between the quad layout and the MFMA layouts.
// Nothing here uses atomics: partial sums have a fixed owner and a fixed order, so two runs give identical bits.
#include <algorithm>
#include <vector>

#include "device_utils.h"
#include "host_graph.h"
#include "internal.h"
#include "node_bodies.h"

namespace mllp {

typedef float f32x4m __attribute__((ext_vector_type(4)));

// Timing build only (`make timing`, libmllp_hip_timing.so): MLLP_FUSED_ABL masks parts of fused_bwd16_kernel out
// (results WRONG): 1 = no sweep, 2 = no statistics, 4 = no input-gradient GEMMs, 8 = no record store, 16 = no
// prologue GEMMs.  The product build compiles FUSED_ABL(bit) to false.
#ifdef MLLP_TIMING_BUILD
static int g_fused_abl = [] {
    const char* e = getenv("MLLP_FUSED_ABL");
    return e ? atoi(e) : 0;
}();
#define FUSED_ABL(bit) ((J.abl & (bit)) != 0)
// per-wavefront cycle sums of the segments of fwd16_row (s_memtime), read back by tools/abl_fused.py
__device__ unsigned long long g_fused_stamps[8192 * 8];
__device__ unsigned long long g_fused_stamps1[8192 * 8];     // the same for bwd1_row
#define FUSED_STAMP(k)                                                                      \
    {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        stamp_sum[k] += now_ - stamp_last;                                                  \
        stamp_last = now_;                                                                  \
    }
#else
#define FUSED_ABL(bit) false
#define FUSED_STAMP(k)
#endif

constexpr int FT = 768;             // threads per workgroup of the sweep kernels: 3 wavefronts per SIMD leave 168 VGPRs
                                    // each -- at 1024 threads (128 VGPRs) the statistics accumulators were spilled to
                                    // scratch and re-loaded per item (42 of the 75 us of fused_bwd16_kernel)
constexpr int FW = FT / 64;         // wavefronts per workgroup
constexpr int RT = 1024;            // threads of the small reduction / head kernels
constexpr int MAXJOBS = 2;

constexpr int NP = FUSED_NP;        // partitions (one per XCD: workgroup b works on partition b % NP)
static_assert(NP == FUSED_PARTS, "partition count");

struct PartTiers {
    int row0, n_block, n_wave, n_group, n_base;
};
struct ItemsDev {
    const int* __restrict__ sptr;    // [n_dst + 1]
    const int2* __restrict__ sent;   // [nnz] {source id, value bits}
    const float2* __restrict__ sax;  // [nnz] {a, x_src} (layer 1)
    int n_dst, nnz;
    const int* __restrict__ order;   // [NP][nw][L] items of every wavefront, -1 = none (host_graph.h::HostWaveLists)
    int nw, L;
    PartTiers part[NP];
};

// which == 0: 16-channel lists at one workgroup per CU, 1: 1-channel lists, 2: 16-channel lists at two workgroups per CU
static ItemsDev items_dev(const FusedOrient& o, int which) {
    const bool scalar = which == 1;
    ItemsDev d;
    d.order = which == 0 ? o.lst16 : which == 1 ? o.lst1 : o.lst16x2;
    d.nw = which == 0 ? o.nw16 : which == 1 ? o.nw1 : o.nw16x2;
    d.L = which == 0 ? o.L16 : which == 1 ? o.L1 : o.L16x2;
    d.sptr = o.sptr;
    d.sent = reinterpret_cast<const int2*>(o.sent);
    d.sax = reinterpret_cast<const float2*>(o.sax);
    d.n_dst = o.n_dst;
    d.nnz = o.nnz;
    for (int q = 0; q < NP; ++q) {
        const FusedTiersDev& t = scalar ? o.t1[q] : o.t16[q];
        d.part[q].row0 = o.row0[q];
        d.part[q].n_block = t.n_block; d.part[q].n_wave = t.n_wave; d.part[q].n_group = t.n_group; d.part[q].n_base = t.n_base;
    }
    return d;
}

// ---- 16x16 per-node GEMMs on the MFMA, rows of a wavefront through an LDS tile ---------------------------
// A wavefront's item is 16 rows x 16 channels, held in the QUAD layout (lane 4 q + p: row q, channels 4 p .. 4 p + 3).
// v_mfma_f32_16x16x4_f32 wants A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15] and returns
// D[m = 4 (lane >> 4) + j][n = lane & 15], j = 0..3.  A tile in LDS (row stride 17 floats: conflict-free for both
// read patterns) converts between the layouts; the LDS operations of one wavefront complete in order, so a wavefront
// needs no barrier around its own tile.  The weights are the B operand: 4 registers per matrix, loaded once per job.
constexpr int TS = 17;                // floats per tile row
constexpr int TILE = 16 * TS;         // floats per tile
__device__ __forceinline__ void tile_put(float* tile, const float4& v, int lane) {
    float* d = tile + (lane >> 2) * TS + 4 * (lane & 3);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
__device__ __forceinline__ float4 tile_get(const float* tile, int lane) {
    const float* d = tile + (lane >> 2) * TS + 4 * (lane & 3);
    return make_float4(d[0], d[1], d[2], d[3]);
}
// A operand with m = row, k = channel (input of a per-row GEMM)
__device__ __forceinline__ void tile_rows(const float* tile, int lane, float (&a)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = tile[(lane & 15) * TS + 4 * s + (lane >> 4)];
}
// operand with m (or n) = channel, k = row (statistics: sums over the rows)
__device__ __forceinline__ void tile_cols(const float* tile, int lane, float (&a)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = tile[(4 * s + (lane >> 4)) * TS + (lane & 15)];
}
__device__ __forceinline__ void tile_put_result(float* tile, const f32x4m& d, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[(4 * (lane >> 4) + j) * TS + (lane & 15)] = d[j];
}
struct MatB {
    float w[4];
};
// y[n] = sum_k W[n][k] x[k]  (W row-major, leading dimension ld):  B[k][n] = W[n][4 s + k]
__device__ __forceinline__ MatB matB(const float* __restrict__ W, int ld, int lane) {
    MatB b;
#pragma unroll
    for (int s = 0; s < 4; ++s) b.w[s] = W[(lane & 15) * ld + 4 * s + (lane >> 4)];
    return b;
}
// y[k] = sum_o W[o][k] g[o]:  B[o'][n = k] = W[4 s + o'][k]
__device__ __forceinline__ MatB matB_t(const float* __restrict__ W, int ld, int lane) {
    MatB b;
#pragma unroll
    for (int s = 0; s < 4; ++s) b.w[s] = W[(4 * s + (lane >> 4)) * ld + (lane & 15)];
    return b;
}
// the B operands live in LDS between uses (one float4 per lane and matrix): registers are scarce across the sweeps
__device__ __forceinline__ void matB_store(float* lds, const MatB& b, int lane) {
    *reinterpret_cast<float4*>(lds + 4 * lane) = make_float4(b.w[0], b.w[1], b.w[2], b.w[3]);
}
__device__ __forceinline__ MatB matB_lds(const float* lds, int lane) {
    const float4 v = *reinterpret_cast<const float4*>(lds + 4 * lane);
    MatB b;
    b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
    return b;
}
__device__ __forceinline__ f32x4m mat_apply(const float (&a)[4], const MatB& b, f32x4m acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.w[s], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4m splat4(float v) { return (f32x4m){v, v, v, v}; }
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4scale(const float4& a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <int J>
__device__ __forceinline__ int quad_bcast_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true);
}
// sum over the 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}


// which rows a wavefront works on: an item of the wave loop, or a block-tier row shared by the workgroup
struct RowSlot {
    int row;      // -1: no row for this quad / lane
    int first;    // index of this unit's first nonzero
    int stride;   // distance between its nonzeros
    int end;      // end of the row
    int mode;     // 0 = row per unit (quad / lane), 1 = row per group of units, 2 = row per wavefront, 3 = per workgroup
    bool writer;  // this unit stores the row's results (and counts it in the statistics)
};
// The slot of an item is computed one item ahead (software pipeline below).  Its two row pointers are LOADED and nothing
// may consume them where they are requested -- hipcc waits for a load in front of its first consumer, and an add right
// behind the load (first = sptr[row] + q) put `s_waitcnt vmcnt(0)` at the top of every item, which also waited for
// the prefetches and stores in flight (1.5 k cycles per item in the in-kernel stamps).  SlotReq keeps the raw values;
// slot_make() turns them into a RowSlot an item later.
struct SlotReq {
    int row, q, stride, mode;
    int sb, se;   // raw loads
    bool writer;
};
// G lanes per unit: 4 (16-channel sweeps: a quad) or 1 (scalar sweeps: a lane)
template <int G>
struct Geo {
    static constexpr int U = 64 / G;              // units per wavefront
    static constexpr int QG = 4;                  // units per row in the group tier: 4 quads (16-channel sweeps) or 4 lanes
                                                  // (1-channel sweeps; with 16 lanes per row an item held only 4 rows and
                                                  // half of all 1-channel items of the Netlib batch were such items)
    static constexpr int RG = U / QG;             // rows per wavefront in the group tier: 4 / 16
};
template <int G>
__device__ __forceinline__ int wave_items(const PartTiers& s) {
    return s.n_wave + (s.n_group + Geo<G>::RG - 1) / Geo<G>::RG + (s.n_base + Geo<G>::U - 1) / Geo<G>::U;
}
template <int G>
__device__ __forceinline__ SlotReq item_request(const ItemsDev& S, const PartTiers& s, int item, int lane) {
    constexpr int U = Geo<G>::U, QG = Geo<G>::QG, RG = Geo<G>::RG;
    const int unit = lane / G;
    const int n_gitems = (s.n_group + RG - 1) / RG;
    SlotReq r;
    bool valid;
    if (item < s.n_wave) {
        r.row = s.row0 + s.n_block + item; r.q = unit; r.stride = U; r.mode = 2; valid = true; r.writer = unit == 0;
    } else if (item < s.n_wave + n_gitems) {
        const int rl = (item - s.n_wave) * RG + unit / QG;
        valid = rl < s.n_group;
        r.row = s.row0 + s.n_block + s.n_wave + rl; r.q = unit % QG; r.stride = QG; r.mode = 1; r.writer = valid && r.q == 0;
    } else {
        const int rl = (item - s.n_wave - n_gitems) * U + unit;
        valid = rl < s.n_base;
        r.row = s.row0 + s.n_block + s.n_wave + s.n_group + rl; r.q = 0; r.stride = 1; r.mode = 0; r.writer = valid;
    }
    // unconditional loads (row 0 for a unit without a row): a load under `if (valid)` is merged with the other path's
    // constant by a register COPY right behind the load, and hipcc waits for the load there (vmcnt(0), a full round trip
    // at the spot where the request was meant to leave and be forgotten)
    const int rc = valid ? r.row : 0;
    r.sb = S.sptr[rc];
    r.se = S.sptr[rc + 1];
    if (!valid) r.row = -1;
    return r;
}
__device__ __forceinline__ SlotReq empty_request() {
    SlotReq r;
    r.row = -1; r.q = 0; r.stride = 1; r.mode = 0; r.sb = 0; r.se = 0; r.writer = false;
    return r;
}
__device__ __forceinline__ RowSlot slot_make(const SlotReq& q) {
    RowSlot r;
    r.row = q.row; r.first = q.sb + q.q; r.stride = q.stride; r.end = q.row >= 0 ? q.se : 0; r.mode = q.mode; r.writer = q.writer;
    // the two loaded values are taken HERE (requested an item ago: no stall).  Left as aliases of the load's registers
    // they were first read by the register copies at the loop's back edge, where hipcc can only wait with vmcnt(0):
    // that also waited for the next item's prefetch, issued moments before (1.3 k cycles per item, in-kernel stamps)
    asm volatile("" : "+v"(r.first), "+v"(r.end));
    return r;
}
template <int G>
__device__ __forceinline__ RowSlot item_slot(const ItemsDev& S, const PartTiers& s, int item, int lane) {
    return slot_make(item_request<G>(S, s, item, lane));
}
template <int G>
__device__ __forceinline__ RowSlot block_slot(const ItemsDev& s, int k, int tid) {     // k: renumbered row id
    RowSlot r;
    r.row = k;
    r.mode = 3;
    r.first = s.sptr[k] + tid / G;
    r.stride = FT / G;
    r.end = s.sptr[k + 1];
    r.writer = tid < G;
    return r;
}
__device__ __forceinline__ int slot_count(const RowSlot& r) {
    return r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
}
__device__ __forceinline__ RowSlot empty_slot() {
    RowSlot r;
    r.row = -1; r.first = 0; r.stride = 1; r.end = 0; r.mode = 0; r.writer = false;
    return r;
}
// The items of one wavefront (static assignment by estimated cost: host_graph.h::HostWaveLists), 64 at a time in a
// VGPR.  A chunk is loaded and WAITED FOR outside the item loop (wave_list_chunk): with the reload inside the loop
// hipcc put `s_waitcnt vmcnt(0)` in front of every v_readlane, which also waited for the stores of the item before
// (1.3 k cycles per item in the stamps).  The item pipelines drain at a chunk's end (one cold start per 64 items).
struct WaveList {
    const int* p;
    int L, chunk;
};
__device__ __forceinline__ WaveList wave_list(const ItemsDev& S, int px, int gw) {
    WaveList w;
    w.L = gw < S.nw ? S.L : 0;
    w.p = S.order + ((size_t)px * S.nw + gw) * S.L;
    w.chunk = -1;
    return w;
}
// items c0 .. c0 + 63 of the list
__device__ __forceinline__ void wave_list_chunk(WaveList& w, int c0, int lane) {
    int v = c0 + lane < w.L ? w.p[c0 + lane] : -1;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));      // taken here, not at the first v_readlane of the item loop
    w.chunk = v;
}
// k-th item of the current chunk or -1 (k is wave-uniform)
__device__ __forceinline__ int wave_list_get(const WaveList& w, int k) {
    return k < 64 ? __builtin_amdgcn_readlane(w.chunk, __builtin_amdgcn_readfirstlane(k)) : -1;
}
// Software pipeline over the items of a wavefront.  Its memory round trips are what a sweep waits for (PMC of the
// first version: 69 % of the wave cycles in s_waitcnt at 4 wavefronts per SIMD), so an item's dependent chain
// "row pointers -> entries -> gathers" and its row data are fetched one item ahead: the slot of item i + 1 is
// computed (row pointers requested) when item i starts, its first entries and its row data are requested when the
// sweep of item i ends -- they land during item i's epilogue -- and item i + 1 starts by issuing its gathers.
// `lanes_per_entry`: the quad lanes that fetch distinct entries (4: slots part, 2: slots part & 1)
template <int LPE>
__device__ __forceinline__ int2 first_entries(const ItemsDev& s, const RowSlot& r, int part) {
    const int k = LPE == 4 ? part : (part & 1);
    return k < slot_count(r) ? s.sent[r.first + k * r.stride] : make_int2(0, 0);
}
// the same without a branch around the load (see item_request): lanes without an entry read a valid index and their
// value is never used (the sweeps test k < slot_count before they use an entry)
template <int LPE>
__device__ __forceinline__ int2 first_entries_any(const ItemsDev& s, const RowSlot& r, int part) {
    const int k = LPE == 4 ? part : (part & 1);
    return s.sent[max(min(r.first + k * r.stride, s.nnz - 1), 0)];
}
// sums / maxima over the units that share a row (quad layout: values replicated inside a quad or owned per part)
__device__ __forceinline__ float shared_sum4(float v, int mode) {
    return mode == 1 ? quads_sum<16>(v) : quads_sum<64>(v);
}
__device__ __forceinline__ float shared_max4(float v, int mode) {
    return mode == 1 ? quads_max<16>(v) : quads_max<64>(v);
}
__device__ __forceinline__ float shared_sum1(float v, int mode) { return mode == 1 ? quad_sum(v) : wave_sum(v); }
__device__ __forceinline__ float shared_max1(float v, int mode) { return mode == 1 ? quad_max(v) : group_max<64>(v); }

// ====================================================================================================
// forward, 16 source channels
// ====================================================================================================
struct FwdJob16 {
    ItemsDev s;
    const float* __restrict__ x_src;   // [n_src, 16]
    const float* __restrict__ x_dst;   // [n_dst, 16]
    const float* __restrict__ D;       // folded weights (param_prep)
    ConvParams p;
    float* __restrict__ h;             // [n_dst, 16]   (not written when head == 2)
    float* __restrict__ Z;             // [n_dst, 16]
    float* __restrict__ aux;           // [n_dst, 4]
    int head;                          // 0: plain conv, 1: + logits, 2: + BCEWithLogits, dL/dh (masked) and fc partials
    const float* __restrict__ fcw;
    const float* __restrict__ fcb;
    const float* __restrict__ inv_n;   // [n_dst] renumbered
    const float* __restrict__ labels;  // [n_dst] renumbered
    const int* __restrict__ perm;      // [n_dst] renumbered -> caller's variable id (where the logit goes)
    float inv_batch;
    float* __restrict__ logits;        // [n_dst] caller's order
    float* __restrict__ g_out;         // [n_dst, 16]  dL/dh of the head, already ReLU-masked
    float* __restrict__ head_part;     // [grid, 18]   {dW_fc[16], db_fc, loss} per workgroup
#ifdef MLLP_TIMING_BUILD
    int abl;
#endif
};
struct FwdLaunch16 {
    FwdJob16 job[MAXJOBS];
    int n_jobs;
};
struct FwdW16 {        // per job in LDS: the B operands of the three 16x16 GEMMs, small vectors (read as float4 at 4 part)
    float BPq[256], BWv[256], BWs[256];
    float pq0[16], Pt[16], bv[16], we[16], bs[16], fcw[16];
    float pt0, fcb;
};
struct SoftState {
    float4 Z;
    float m, L, u;
};

// four gathered source rows of a quad and their matrix entries
struct Gather4 {
    float4 x0, x1, x2, x3;
    float a0, a1, a2, a3;
};
// issue the gathers of slots k0 .. k0 + 3 from the entries `en` (one per lane of the quad), then fetch the next
// four entries into `en`
__device__ __forceinline__ void gather4_issue(const ItemsDev& s, const float* __restrict__ X, const RowSlot& r, int n_mine,
                                              int k0, int part, int2& en, Gather4& g) {
    const int colm = en.x;
    const float am = __int_as_float(en.y);
    const int c0 = quad_bcast_i<0>(colm), c1 = quad_bcast_i<1>(colm), c2 = quad_bcast_i<2>(colm), c3 = quad_bcast_i<3>(colm);
    g.a0 = quad_bcast<0>(am); g.a1 = quad_bcast<1>(am); g.a2 = quad_bcast<2>(am); g.a3 = quad_bcast<3>(am);
    // unconditional: a slot past the row's end carries the entry {0, 0} and gathers row 0, which the steps mask out
    // (sixteen zeroing moves and four exec-masked branches less per step than loads under `if (k < n_mine)`)
    g.x0 = ld4(X + (size_t)c0 * 16 + 4 * part);
    g.x1 = ld4(X + (size_t)c1 * 16 + 4 * part);
    g.x2 = ld4(X + (size_t)c2 * 16 + 4 * part);
    g.x3 = ld4(X + (size_t)c3 * 16 + 4 * part);
    const int kn = k0 + 4 + part;
    en = kn < n_mine ? s.sent[r.first + kn * r.stride] : make_int2(0, 0);
}

// online segment softmax over four gathered nonzeros
__device__ __forceinline__ void fwd16_step(const Gather4& g, int k0, int n_mine, const float4& qp, float t, SoftState& st) {
    const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine, ok2 = k0 + 2 < n_mine, ok3 = k0 + 3 < n_mine;
    const float d0 = ok0 ? fmaf(g.a0, t, quad_sum(dot4(qp, g.x0))) : NEG_BIG;
    const float d1 = ok1 ? fmaf(g.a1, t, quad_sum(dot4(qp, g.x1))) : NEG_BIG;
    const float d2 = ok2 ? fmaf(g.a2, t, quad_sum(dot4(qp, g.x2))) : NEG_BIG;
    const float d3 = ok3 ? fmaf(g.a3, t, quad_sum(dot4(qp, g.x3))) : NEG_BIG;
    const float mi = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
    if (__any(mi > st.m)) {          // some row of this wavefront moves its running max: rescale those rows
        const float mn = fmaxf(st.m, mi);
        const float sc = exp_acc(st.m - mn);
        st.L *= sc; st.u *= sc;
        st.Z = f4scale(st.Z, sc);
        st.m = mn;
    }
    const float p0 = ok0 ? exp_acc(d0 - st.m) : 0.0f, p1 = ok1 ? exp_acc(d1 - st.m) : 0.0f;
    const float p2 = ok2 ? exp_acc(d2 - st.m) : 0.0f, p3 = ok3 ? exp_acc(d3 - st.m) : 0.0f;
    st.L += (p0 + p1) + (p2 + p3);
    st.u = fmaf(p0, g.a0, fmaf(p1, g.a1, fmaf(p2, g.a2, fmaf(p3, g.a3, st.u))));
    fma4(p0, g.x0, st.Z);
    fma4(p1, g.x1, st.Z);
    fma4(p2, g.x2, st.Z);
    fma4(p3, g.x3, st.Z);
}

// the units that share a row hold partial states: every lane ends with the row's state
__device__ __forceinline__ void soft_merge(SoftState& st, int mode) {
    const float M = shared_max4(st.m, mode);
    const float f = exp_acc(st.m - M);      // partial without nonzeros: exp(-huge) == 0
    st.m = M;
    st.L = shared_sum4(st.L * f, mode);
    st.u = shared_sum4(st.u * f, mode);
    st.Z.x = shared_sum4(st.Z.x * f, mode);
    st.Z.y = shared_sum4(st.Z.y * f, mode);
    st.Z.z = shared_sum4(st.Z.z * f, mode);
    st.Z.w = shared_sum4(st.Z.w * f, mode);
}

struct HeadAcc {
    float4 w;      // dW_fc of this lane's four channels
    float b, l;    // db_fc, loss (part-0 lanes)
};

// Results of an item, stored one item LATER (after the next item's sweep).  Stored at the item's end they were still
// on their way at the loop's back edge, where the first use of the prefetched row data makes hipcc wait with vmcnt(0)
// -- 1.3-1.6 k cycles per item waiting for store acknowledgements (in-kernel stamps, "between items").
struct FwdPending {
    float4 zn, hv, g, aux;
    float z;
    int row, lrow;      // row < 0: nothing pending
};
__device__ __forceinline__ void fwd16_flush(const FwdJob16& J, FwdPending& p, int part) {
    if (p.row >= 0 && !FUSED_ABL(2048)) {
        *reinterpret_cast<float4*>(J.Z + (size_t)p.row * 16 + 4 * part) = p.zn;
        if (part == 0) reinterpret_cast<float4*>(J.aux)[p.row] = p.aux;
        if (J.head != 2) *reinterpret_cast<float4*>(J.h + (size_t)p.row * 16 + 4 * part) = p.hv;
        if (J.head && part == 0) J.logits[p.lrow] = p.z;
        if (J.head == 2) *reinterpret_cast<float4*>(J.g_out + (size_t)p.row * 16 + 4 * part) = p.g;
    }
    p.row = -1;
}

// per-row prologue + sweep + epilogue of one job.  Two-deep pipeline over a wavefront's items: every item costs two
// DEPENDENT round trips (row pointers -> first entries / row data), about 2.5 k cycles each, and with the second one
// requested behind the sweep an item waited ~1.5 k cycles for it at its start (in-kernel stamps, "between items").
// Now item i, right behind its own first gathers, turns the row pointers of item i + 1 (requested during item i - 1:
// q1) into that item's slot and requests its row data and first entries (nx), and requests the row pointers of item
// i + 2 (it2): both round trips have a whole item to complete.
struct FwdNext {
    RowSlot r;
    float4 xd;
    int2 en;
    int lrow;
    SlotReq q;      // row pointers of the item after the next
};
__device__ __forceinline__ void fwd16_row(const FwdJob16& J, const FwdW16& W, const PartTiers& P, const RowSlot& r,
                                          const float4& xd, int2& en, int lrow, const SlotReq& q1, int it2, FwdNext& nx,
                                          FwdPending& pend, int part, int lane, float* merge_lds, float* tiles, HeadAcc& ha
#ifdef MLLP_TIMING_BUILD
                                          , unsigned long long (&stamp_sum)[8], unsigned long long& stamp_last
#endif
) {
    FUSED_STAMP(0)      // between items: slot of the next item, loop overhead
    const int n_mine = FUSED_ABL(256) ? 0 : slot_count(r);
    Gather4 gt;
    gather4_issue(J.s, J.x_src, r, n_mine, 0, part, en, gt);      // the first gathers leave before anything else
    nx.r = slot_make(q1);
    nx.xd = ld4(J.x_dst + (size_t)max(nx.r.row, 0) * 16 + 4 * part);      // row 0 for a quad without a row: never stored
    nx.en = first_entries_any<4>(J.s, nx.r, part);
    nx.lrow = J.perm[max(nx.r.row, 0)];
    nx.q = item_request<4>(J.s, P, max(it2, 0), lane);                     // it2 < 0: the item loop ends before it is used
    float4 qp;
    float t;
    {   // q' = Pq x + pq0 on the MFMA (rows of the wavefront through tile 0, kept for the epilogue; result through tile 1)
        float ax_[4];
        tile_put(tiles, xd, lane);
        tile_rows(tiles, lane, ax_);
        if (!FUSED_ABL(1024))
        tile_put_result(tiles + TILE, mat_apply(ax_, matB_lds(W.BPq, lane), splat4(W.pq0[lane & 15])), lane);
        qp = tile_get(tiles + TILE, lane);
        t = quad_sum(dot4(lds4(W.Pt + 4 * part), xd)) + W.pt0;
    }
    FUSED_STAMP(1)      // first gathers issued, prologue GEMM
    SoftState st;
    st.Z = f4zero(); st.m = NEG_BIG; st.L = 0.0f; st.u = 0.0f;
    for (int k0 = 0;;) {
        fwd16_step(gt, k0, n_mine, qp, t, st);
        k0 += 4;
        if (!__any(k0 < n_mine)) break;
        gather4_issue(J.s, J.x_src, r, n_mine, k0, part, en, gt);
    }
    FUSED_STAMP(2)      // sweep
    // The sweep's last wait for gathers has just drained the memory pipe: the next item's data (requested at this item's
    // start) is TAKEN here, where waiting is free.  Left pending, hipcc waits for it at the register copies of the loop's
    // back edge with vmcnt(0), which also waits for the stores below (1.3 k cycles per item in the stamps).
    asm volatile("" : "+v"(nx.xd.x), "+v"(nx.xd.y), "+v"(nx.xd.z), "+v"(nx.xd.w), "+v"(nx.en.x), "+v"(nx.en.y),
                      "+v"(nx.lrow), "+v"(nx.q.sb), "+v"(nx.q.se));
    fwd16_flush(J, pend, part);     // the previous item's results leave now: acknowledged long before the back edge
    const int lrow_cur = lrow;      // where this row's logit goes (caller's variable order), fetched with the row
    FUSED_STAMP(3)      // prefetch issue

    if (r.mode >= 1) soft_merge(st, r.mode);
    if (r.mode == 3) {       // merge the 16 wavefronts of the workgroup through LDS (fixed order)
        const int wave = threadIdx.x >> 6;
        if (lane < 4) {
            float* slot = merge_lds + wave * 20;
            if (part == 0) *reinterpret_cast<float4*>(slot) = make_float4(st.m, st.L, st.u, 0.0f);
            *reinterpret_cast<float4*>(slot + 4 + 4 * part) = st.Z;
        }
        __syncthreads();
        float M = NEG_BIG;
        for (int w = 0; w < FW; ++w) M = fmaxf(M, merge_lds[w * 20]);
        float L = 0.0f, u = 0.0f;
        float4 Z = f4zero();
        for (int w = 0; w < FW; ++w) {
            const float4 hd = lds4(merge_lds + w * 20);
            const float f = exp_acc(hd.x - M);
            L = fmaf(f, hd.y, L);
            u = fmaf(f, hd.z, u);
            fma4(f, lds4(merge_lds + w * 20 + 4 + 4 * part), Z);
        }
        st.m = M; st.L = L; st.u = u; st.Z = Z;
        __syncthreads();     // the slots are free for the next block-tier row
    }
    const bool writer = r.writer;
    // epilogue: o = Wv Zn + Ws x + (bs + S bv + un we)
    const float rinv = 1.0f / (st.L + 1e-16f);   // torch_geometric.utils.softmax: sum + 1e-16
    const float S = st.L * rinv, un = st.u * rinv;
    const float4 zn = f4scale(st.Z, rinv);
    float4 o;
    {
        float ax_[4], az_[4];
        tile_rows(tiles, lane, ax_);
        tile_put(tiles + TILE, zn, lane);
        tile_rows(tiles + TILE, lane, az_);
        if (!FUSED_ABL(512))
        tile_put_result(tiles + 2 * TILE,
                        mat_apply(ax_, matB_lds(W.BWs, lane), mat_apply(az_, matB_lds(W.BWv, lane), splat4(0.0f))), lane);
        o = tile_get(tiles + 2 * TILE, lane);
    }
    o = f4add(o, lds4(W.bs + 4 * part));
    fma4(S, lds4(W.bv + 4 * part), o);
    fma4(un, lds4(W.we + 4 * part), o);
    const float4 hv = make_float4(fmaxf(o.x, 0.0f), fmaxf(o.y, 0.0f), fmaxf(o.z, 0.0f), fmaxf(o.w, 0.0f));
    FUSED_STAMP(4)      // merges, epilogue GEMM
    pend.row = writer ? r.row : -1;
    pend.zn = zn; pend.hv = hv; pend.lrow = lrow_cur;
    pend.aux = make_float4(un, st.L > 0.0f ? st.m : 0.0f, rinv, S);
    pend.z = 0.0f; pend.g = f4zero();
    if (J.head) {        // fc (16 -> 1) on the conv's output row, reference linear_program_methods.py:250
        const float4 fw = lds4(W.fcw + 4 * part);
        const float z = quad_sum(dot4(hv, fw)) + W.fcb;
        pend.z = z;
        if (J.head == 2 && writer) {      // BCEWithLogitsLoss, mean per instance / batch: linear_program_experiment.py:41,139-140
            const float y = J.labels[r.row];
            const float wn = J.inv_n[r.row] * J.inv_batch;
            const float e = expf(-fabsf(z));
            const float sig = z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
            const float dz = wn * (sig - y);
            pend.g = make_float4(hv.x > 0.0f ? dz * fw.x : 0.0f, hv.y > 0.0f ? dz * fw.y : 0.0f,
                                 hv.z > 0.0f ? dz * fw.z : 0.0f, hv.w > 0.0f ? dz * fw.w : 0.0f);
            fma4(dz, hv, ha.w);
            if (part == 0) {
                ha.b += dz;
                ha.l += wn * (fmaxf(z, 0.0f) - z * y + log1pf(e));
            }
        }
    }
    FUSED_STAMP(5)      // stores, head
}

// {dW_fc[16], db, loss} of the workgroup -> head_part[blockIdx][18] (wave order)
template <int NW>
__device__ __forceinline__ void head_partials_store(HeadAcc& ha, float* head_lds, float* head_part, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    ha.w.x = quads_sum<64>(ha.w.x); ha.w.y = quads_sum<64>(ha.w.y);
    ha.w.z = quads_sum<64>(ha.w.z); ha.w.w = quads_sum<64>(ha.w.w);
    ha.b = wave_sum(ha.b);
    ha.l = wave_sum(ha.l);
    if (lane < 4) {
        float* d = head_lds + wave * 18 + 4 * lane;
        d[0] = ha.w.x; d[1] = ha.w.y; d[2] = ha.w.z; d[3] = ha.w.w;
    }
    if (lane == 0) { head_lds[wave * 18 + 16] = ha.b; head_lds[wave * 18 + 17] = ha.l; }
    __syncthreads();
    if (tid < 18) {
        float v = 0.0f;
        for (int w = 0; w < NW; ++w) v += head_lds[w * 18 + tid];
        head_part[(size_t)blockIdx.x * 18 + tid] = v;
    }
}

__global__ __launch_bounds__(FT) void fused_fwd16_kernel(FwdLaunch16 A) {
    __shared__ FwdW16 Ws_[MAXJOBS];
    __shared__ float merge_lds[FW * 20];
    __shared__ float head_lds[FW * 18];
    __shared__ float tiles_[FW * 3 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, part = lane & 3;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob16& J = A.job[j];
        FwdW16& W = Ws_[j];
        if (wave == 0) matB_store(W.BPq, matB(J.D + OFF_PQ, 16, lane), lane);
        if (wave == 1) matB_store(W.BWv, matB(J.p.Wv, 16, lane), lane);
        if (wave == 2) matB_store(W.BWs, matB(J.p.Ws, 16, lane), lane);
        if (tid < 16) {
            W.pq0[tid] = J.D[OFF_PQ0 + tid];
            W.Pt[tid] = J.D[OFF_PT + tid];
            W.bv[tid] = J.p.bv[tid];
            W.we[tid] = J.p.we[tid];
            W.bs[tid] = J.p.bs[tid];
            W.fcw[tid] = J.head ? J.fcw[tid] : 0.0f;
        }
        if (tid == 0) {
            W.pt0 = J.D[OFF_PT0];
            W.fcb = J.head ? J.fcb[0] : 0.0f;
        }
    }
    __syncthreads();
    float* tiles = tiles_ + wave * 3 * TILE;
    HeadAcc ha;
    ha.w = f4zero(); ha.b = 0.0f; ha.l = 0.0f;
#ifdef MLLP_TIMING_BUILD
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_begin = stamp_last;
#define STAMP_ARGS , stamp_sum, stamp_last
#else
#define STAMP_ARGS
#endif
    // this workgroup's partition of the instances, its rank among the partition's workgroups, their wavefronts
    const int px = blockIdx.x % NP, bi = blockIdx.x / NP, gp = gridDim.x / NP;
    const int gw = bi * FW + wave;
    float* head_part = nullptr;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob16& J = A.job[j];
        const PartTiers P = J.s.part[px];
        if (J.head == 2) head_part = J.head_part;
        // block tier: the whole workgroup walks one long row at a time (longest rows first)
        const RowSlot none = empty_slot();
        FwdPending pend;
        pend.row = -1;
        for (int k = bi; k < P.n_block; k += gp) {
            const RowSlot r = block_slot<4>(J.s, P.row0 + k, tid);
            const float4 xd = ld4(J.x_dst + (size_t)r.row * 16 + 4 * part);
            int2 en = first_entries<4>(J.s, r, part);
            const int lrow = J.head ? J.perm[r.row] : 0;
            FwdNext nx;
            fwd16_row(J, Ws_[j], P, r, xd, en, lrow, empty_request(), -1, nx, pend, part, lane, merge_lds, tiles, ha STAMP_ARGS);
        }
        // wave loop: this wavefront's items (static assignment by estimated cost)
        WaveList wl = wave_list(J.s, px, gw);
        for (int c0 = 0; c0 < wl.L; c0 += 64) {
            wave_list_chunk(wl, c0, lane);
            int it = wave_list_get(wl, 0), it1 = wave_list_get(wl, 1);
            RowSlot r = it >= 0 ? item_slot<4>(J.s, P, it, lane) : none;
            float4 xd = r.row >= 0 ? ld4(J.x_dst + (size_t)r.row * 16 + 4 * part) : f4zero();
            int2 en = first_entries<4>(J.s, r, part);
            int lrow = (J.head && r.row >= 0) ? J.perm[r.row] : 0;
            SlotReq q1 = item_request<4>(J.s, P, max(it1, 0), lane);
            if (it1 < 0) q1.row = -1;
            // cold start: everything requested above is waited for HERE -- left pending into the loop, the copies at its
            // header would wait with vmcnt(0) in every iteration (the waits of a loop header cover its entry path too)
            asm volatile("" : "+v"(xd.x), "+v"(xd.y), "+v"(xd.z), "+v"(xd.w), "+v"(en.x), "+v"(en.y), "+v"(lrow),
                              "+v"(q1.sb), "+v"(q1.se));
            for (int k = 0; it >= 0; ++k) {
                const int it2 = wave_list_get(wl, k + 2);
                FwdNext nx;
                fwd16_row(J, Ws_[j], P, r, xd, en, lrow, q1, it2, nx, pend, part, lane, merge_lds, tiles, ha STAMP_ARGS);
#ifdef MLLP_TIMING_BUILD
                stamp_sum[7] += 1;      // items
#endif
                r = nx.r; xd = nx.xd; en = nx.en; lrow = nx.lrow; q1 = nx.q;
                it = it1; it1 = it2;
            }
        }
        fwd16_flush(J, pend, part);
    }
#ifdef MLLP_TIMING_BUILD
    if (lane == 0 && A.job[0].abl >= 0) {
        stamp_sum[6] = __builtin_amdgcn_s_memtime() - stamp_begin;      // whole kernel of this wavefront
        unsigned long long* d = g_fused_stamps + (size_t)(blockIdx.x * FW + wave) * 8;
        for (int k = 0; k < 8; ++k) d[k] = stamp_sum[k];
    }
#endif
    if (head_part) head_partials_store<FW>(ha, head_lds, head_part, tid);     // (uniform: one job at most has a head)
}

// ====================================================================================================
// forward, 1 source channel (layer 1): one lane per row; the source feature rides with the value
// ====================================================================================================
struct FwdJob1 {
    ItemsDev s;
    const float* __restrict__ x_dst;   // [n_dst] renumbered
    const float* __restrict__ D;
    ConvParams p;
    float* __restrict__ h;             // [n_dst, 16]
    float* __restrict__ Z;             // [n_dst]
    float* __restrict__ aux;           // [n_dst, 4]
};
struct FwdLaunch1 {
    FwdJob1 job[MAXJOBS];
    int n_jobs;
};
struct alignas(16) FwdW1 {
    float wv[16], ws[16], bv[16], we[16], bs[16];
    float pq, pq0, pt, pt0;
};
struct Soft1 {
    float m, L, u, Z;
};

// Eight entries {a_ij, x_src} of a unit (lane / 16 lanes / wavefront per row), requested together.  The loads are
// unconditional (clamped index; a slot past the row's end is masked where it is used): the first version fetched four
// entries per step under `k < n ? load : 0`, one round trip to L2 per step of four nonzeros, and a 16-nonzero row
// cost four of them in sequence (the 1-channel sweeps do a handful of FMAs per nonzero: they are pure latency).
struct Ent8 {
    float2 e0, e1, e2, e3, e4, e5, e6, e7;
};
__device__ __forceinline__ void ent8_load(const ItemsDev& s, const RowSlot& r, int k0, Ent8& E) {
    const int last = max(s.nnz - 1, 0), d = r.stride, b = r.first + k0 * d;
    E.e0 = s.sax[min(b, last)];         E.e1 = s.sax[min(b + d, last)];
    E.e2 = s.sax[min(b + 2 * d, last)]; E.e3 = s.sax[min(b + 3 * d, last)];
    E.e4 = s.sax[min(b + 4 * d, last)]; E.e5 = s.sax[min(b + 5 * d, last)];
    E.e6 = s.sax[min(b + 6 * d, last)]; E.e7 = s.sax[min(b + 7 * d, last)];
}

// online softmax over four entries (slots k0 .. k0 + 3 of the unit)
__device__ __forceinline__ void fwd1_step4(const float2& e0, const float2& e1, const float2& e2, const float2& e3, int k0,
                                           int n_mine, float qp, float t, Soft1& st) {
    const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine, ok2 = k0 + 2 < n_mine, ok3 = k0 + 3 < n_mine;
    const float d0 = ok0 ? fmaf(qp, e0.y, e0.x * t) : NEG_BIG, d1 = ok1 ? fmaf(qp, e1.y, e1.x * t) : NEG_BIG;
    const float d2 = ok2 ? fmaf(qp, e2.y, e2.x * t) : NEG_BIG, d3 = ok3 ? fmaf(qp, e3.y, e3.x * t) : NEG_BIG;
    const float mi = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
    if (__any(mi > st.m)) {
        const float mn = fmaxf(st.m, mi);
        const float sc = exp_acc(st.m - mn);
        st.L *= sc; st.u *= sc; st.Z *= sc;
        st.m = mn;
    }
    const float p0 = ok0 ? exp_acc(d0 - st.m) : 0.0f, p1 = ok1 ? exp_acc(d1 - st.m) : 0.0f;
    const float p2 = ok2 ? exp_acc(d2 - st.m) : 0.0f, p3 = ok3 ? exp_acc(d3 - st.m) : 0.0f;
    st.L += (p0 + p1) + (p2 + p3);
    st.u = fmaf(p0, e0.x, fmaf(p1, e1.x, fmaf(p2, e2.x, fmaf(p3, e3.x, st.u))));
    st.Z = fmaf(p0, e0.y, fmaf(p1, e1.y, fmaf(p2, e2.y, fmaf(p3, e3.y, st.Z))));
}
// E: slots 0 .. 7, requested by the caller before the row's other loads
__device__ __forceinline__ void fwd1_edges(const ItemsDev& s, const RowSlot& r, float qp, float t, Soft1& st, Ent8& E) {
    const int n_mine = slot_count(r);
    for (int k0 = 0; __any(k0 < n_mine); k0 += 8) {
        Ent8 N;
        const bool more = __any(k0 + 8 < n_mine);
        if (more) ent8_load(s, r, k0 + 8, N);
        fwd1_step4(E.e0, E.e1, E.e2, E.e3, k0, n_mine, qp, t, st);
        if (__any(k0 + 4 < n_mine)) fwd1_step4(E.e4, E.e5, E.e6, E.e7, k0 + 4, n_mine, qp, t, st);
        if (more) E = N;
    }
}

__device__ __forceinline__ void fwd1_row(const FwdJob1& J, const FwdW1& W, const RowSlot& r, int lane, float* merge_lds) {
    Ent8 E;
    ent8_load(J.s, r, 0, E);
    const float x = J.x_dst[max(r.row, 0)];      // row 0 for a lane without a row: never stored
    const float qp = fmaf(W.pq, x, W.pq0), t = fmaf(W.pt, x, W.pt0);
    Soft1 st;
    st.m = NEG_BIG; st.L = 0.0f; st.u = 0.0f; st.Z = 0.0f;
    fwd1_edges(J.s, r, qp, t, st, E);
    if (r.mode >= 1) {
        const float M = shared_max1(st.m, r.mode);
        const float f = exp_acc(st.m - M);
        st.m = M;
        st.L = shared_sum1(st.L * f, r.mode);
        st.u = shared_sum1(st.u * f, r.mode);
        st.Z = shared_sum1(st.Z * f, r.mode);
    }
    if (r.mode == 3) {
        const int wave = threadIdx.x >> 6;
        if (lane == 0) *reinterpret_cast<float4*>(merge_lds + wave * 4) = make_float4(st.m, st.L, st.u, st.Z);
        __syncthreads();
        float M = NEG_BIG;
        for (int w = 0; w < FW; ++w) M = fmaxf(M, merge_lds[w * 4]);
        float L = 0.0f, u = 0.0f, Z = 0.0f;
        for (int w = 0; w < FW; ++w) {
            const float4 hd = lds4(merge_lds + w * 4);
            const float f = exp_acc(hd.x - M);
            L = fmaf(f, hd.y, L);
            u = fmaf(f, hd.z, u);
            Z = fmaf(f, hd.w, Z);
        }
        st.m = M; st.L = L; st.u = u; st.Z = Z;
        __syncthreads();
    }
    const float rinv = 1.0f / (st.L + 1e-16f);
    const float S = st.L * rinv, un = st.u * rinv, zn = st.Z * rinv;
    if (r.mode == 0) {
        // Base tier: the item's 64 output rows are consecutive -> h is stored COALESCED (lane l writes float4 number
        // l + 64 q of the item's 4 KB: row (l + 64 q) / 4), the row's four scalars fetched from its lane by ds_bpermute.
        // A lane storing its own 64-byte row made every store instruction touch 64 cache lines.
        const int base_row = __builtin_amdgcn_readfirstlane(r.row);          // lane 0 of a base item always has a row
        const int n_valid = __popcll(__ballot(r.row >= 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, row_l = idx >> 2, ch = (idx & 3) * 4;
            const float Sr = __shfl(S, row_l, 64), ur = __shfl(un, row_l, 64), zr = __shfl(zn, row_l, 64), xr = __shfl(x, row_l, 64);
            float4 v = lds4(W.bs + ch);
            fma4(Sr, lds4(W.bv + ch), v);
            fma4(ur, lds4(W.we + ch), v);
            fma4(zr, lds4(W.wv + ch), v);
            fma4(xr, lds4(W.ws + ch), v);
            if (row_l < n_valid)
                *reinterpret_cast<float4*>(J.h + (size_t)(base_row + row_l) * 16 + ch) =
                    make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
        }
        if (r.writer) {
            J.Z[r.row] = zn;
            reinterpret_cast<float4*>(J.aux)[r.row] = make_float4(un, st.L > 0.0f ? st.m : 0.0f, rinv, S);
        }
        return;
    }
    if (!r.writer) return;
    float o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = W.bs[c];
        v = fmaf(S, W.bv[c], v);
        v = fmaf(un, W.we[c], v);
        v = fmaf(W.wv[c], zn, v);
        v = fmaf(W.ws[c], x, v);
        o[c] = fmaxf(v, 0.0f);
    }
    float4* hd = reinterpret_cast<float4*>(J.h + (size_t)r.row * 16);
    hd[0] = make_float4(o[0], o[1], o[2], o[3]);
    hd[1] = make_float4(o[4], o[5], o[6], o[7]);
    hd[2] = make_float4(o[8], o[9], o[10], o[11]);
    hd[3] = make_float4(o[12], o[13], o[14], o[15]);
    J.Z[r.row] = zn;
    reinterpret_cast<float4*>(J.aux)[r.row] = make_float4(un, st.L > 0.0f ? st.m : 0.0f, rinv, S);
}

__global__ __launch_bounds__(FT) void fused_fwd1_kernel(FwdLaunch1 A) {
    __shared__ FwdW1 Ws_[MAXJOBS];
    __shared__ float merge_lds[FW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob1& J = A.job[j];
        FwdW1& W = Ws_[j];
        if (tid < 16) {
            W.wv[tid] = J.p.Wv[tid]; W.ws[tid] = J.p.Ws[tid];
            W.bv[tid] = J.p.bv[tid]; W.we[tid] = J.p.we[tid]; W.bs[tid] = J.p.bs[tid];
        }
        if (tid == 0) {
            W.pq = J.D[OFF_PQ]; W.pq0 = J.D[OFF_PQ0]; W.pt = J.D[OFF_PT]; W.pt0 = J.D[OFF_PT0];
        }
    }
    __syncthreads();
    const int px = blockIdx.x % NP, bi = blockIdx.x / NP, gp = gridDim.x / NP;
    const int gw = bi * FW + wave;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob1& J = A.job[j];
        const PartTiers P = J.s.part[px];
        for (int k = bi; k < P.n_block; k += gp)
            fwd1_row(J, Ws_[j], block_slot<1>(J.s, P.row0 + k, tid), lane, merge_lds);
        WaveList wl = wave_list(J.s, px, gw);
        for (int c0 = 0; c0 < wl.L; c0 += 64) {
            wave_list_chunk(wl, c0, lane);
            // row pointers of item k + 1 requested before item k is worked on
            int it = wave_list_get(wl, 0);
            SlotReq q = item_request<1>(J.s, P, max(it, 0), lane);
            for (int k = 0; it >= 0; ++k) {
                const int itn = wave_list_get(wl, k + 1);
                const RowSlot r = slot_make(q);
                q = item_request<1>(J.s, P, max(itn, 0), lane);
                fwd1_row(J, Ws_[j], r, lane, merge_lds);
                it = itn;
            }
        }
    }
}

// ====================================================================================================
// backward, destination-major, 16 channels: ReLU mask, record, sweep, input gradient, parameter statistics
//   rec_i = { q'_i[16], gv_i[16], t_i, rowmax_i, rinv_i, ge_i, c_i, 0, 0, 0 }
//   alpha_ij = exp(l_ij - rowmax_i) rinv_i ;  dl_ij = alpha_ij ( <gv_i, X_j> + a_ij ge_i + c_i )
//   dq'_i = sum_j dl_ij X_j, ds_i = sum_j dl_ij, dt_i = sum_j dl_ij a_ij
//   dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt
//   statistics T0..T6 (node_kernels.hip::param_stats16_kernel) accumulated on the MFMA over the rows of the wavefront
// ====================================================================================================
struct BwdJob16 {
    ItemsDev s;
    const float* __restrict__ x_src;   // [n_src, 16]
    const float* __restrict__ x_dst;   // [n_dst, 16]
    const float* __restrict__ D;
    ConvParams p;
    const float* __restrict__ h;       // [n_dst, 16] conv output (ReLU mask); nullptr: dh_a is already masked
    const float* __restrict__ dh_a;    // [n_dst, 16]
    const float* __restrict__ dh_b;    // [n_dst, 16] or nullptr (second contribution to dL/dh)
    const float* __restrict__ Z;
    const float* __restrict__ aux;
    float* __restrict__ rec;           // [n_dst, REC_W] or nullptr (no source-major sweep follows)
    float* __restrict__ dx_dst;        // [n_dst, 16] or nullptr
    int mask_dx;                       // dx_dst is stored times (x_dst > 0): x_dst is a ReLU output and the consumer of
                                       // dx_dst (the backward of the layer below) would re-read it only to apply this mask
    float* __restrict__ stats;         // [grid, STAT_FLOATS]
#ifdef MLLP_TIMING_BUILD
    int abl;
#endif
};
struct BwdLaunch16 {
    BwdJob16 job[MAXJOBS];
    int n_jobs;
};
struct BwdW16 {        // per job in LDS: B operands (gv = Wv^T g, q' = Pq x, dx = Ws^T g + Pq^T dq'), small vectors
    float BWvT[256], BPq[256], BWsT[256], BPqT[256];
    float pq0[16], Pt[16], Pb[16], bv[16], we[16];
    float pt0;
};
struct BwdState {
    float4 dq;
    float ds, dt;
};

__device__ __forceinline__ void bwd16_step(const Gather4& g, int k0, int n_mine, const float4& qp, const float4& gv, float t,
                                           float m, float rinv, float ge, float cc, BwdState& st) {
#define MLLP_BWD_SLOT(K, A_, X_)                                                           \
    {                                                                                      \
        const float l_ = fmaf(A_, t, quad_sum(dot4(qp, X_)));                              \
        const float al_ = (k0 + K < n_mine) ? exp_acc(l_ - m) * rinv : 0.0f;               \
        const float dl_ = al_ * (quad_sum(dot4(gv, X_)) + fmaf(A_, ge, cc));               \
        st.ds += dl_;                                                                      \
        st.dt = fmaf(dl_, A_, st.dt);                                                      \
        fma4(dl_, X_, st.dq);                                                              \
    }
    MLLP_BWD_SLOT(0, g.a0, g.x0)
    MLLP_BWD_SLOT(1, g.a1, g.x1)
    MLLP_BWD_SLOT(2, g.a2, g.x2)
    MLLP_BWD_SLOT(3, g.a3, g.x3)
#undef MLLP_BWD_SLOT
}

// the row data of a destination-major backward item (fetched one item ahead)
struct BwdRow {
    float4 ga, gb, h, x, Z, ax;
};
__device__ __forceinline__ void bwd16_fetch(const BwdJob16& J, const RowSlot& r, int part, BwdRow& d) {
    const bool have = r.row >= 0;
    const size_t ro = have ? (size_t)r.row * 16 + 4 * part : 0;
    const float4 z4 = f4zero();
    d.ga = have ? ld4(J.dh_a + ro) : z4;
    d.gb = (J.dh_b && have) ? ld4(J.dh_b + ro) : z4;
    d.h = (J.h && have) ? ld4(J.h + ro) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    d.x = have ? ld4(J.x_dst + ro) : z4;
    d.Z = have ? ld4(J.Z + ro) : z4;
    d.ax = have ? reinterpret_cast<const float4*>(J.aux)[r.row] : z4;   // {u, rowmax, rinv, S}
}

// tiles of a wavefront in the destination-major backward sweep
constexpr int TB_G = 0, TB_X = 1, TB_Z = 2, TB_DQ = 3, TB_SC = 4, TB_E = 5, TB_N = 6;

// itn: the wavefront's next item (-1: none).  Its row pointers are requested BEHIND this item's first gathers: requested
// in front of them (by the caller), they were the youngest loads when the first use of the prefetched row data made
// hipcc wait with vmcnt(0) -- a full round trip at the top of every item.
__device__ __forceinline__ void bwd16_row(const BwdJob16& J, const BwdW16& W, const PartTiers& P, const RowSlot& r,
                                          BwdRow& rd, int2& en, int itn, RowSlot& rn, int part,
                                          int lane, float* merge_lds, float* tiles, f32x4m (&acc)[STAT_TILES]) {
    const bool have = r.row >= 0;
    const bool writer = r.writer;
    const int n_mine = FUSED_ABL(1) ? 0 : slot_count(r);
    Gather4 gt;
    gather4_issue(J.s, J.x_src, r, n_mine, 0, part, en, gt);      // the first gathers leave before anything else
    SlotReq qn = item_request<4>(J.s, P, max(itn, 0), lane);       // consumed behind the sweep
    if (itn < 0) qn.row = -1;
    const int xpos = (rd.x.x > 0.0f ? 1 : 0) | (rd.x.y > 0.0f ? 2 : 0) | (rd.x.z > 0.0f ? 4 : 0) | (rd.x.w > 0.0f ? 8 : 0);
    float4 qp, gv;
    float t, m, rinv, ge, cc;
    {   // everything of the row that the sweep does not need stays in the tiles G, X, Z, E until the statistics
        float4 g = f4add(rd.ga, rd.gb);
        g = make_float4(rd.h.x > 0.0f ? g.x : 0.0f, rd.h.y > 0.0f ? g.y : 0.0f, rd.h.z > 0.0f ? g.z : 0.0f, rd.h.w > 0.0f ? g.w : 0.0f);
        const float4 xd = rd.x, Zn = rd.Z, ax = rd.ax;
        // gv = Wv^T g and q' = Pq x + pq0 on the MFMA; the results come back through the (still unused) tiles DQ and SC
        float ag_[4], ax_[4];
        tile_put(tiles + TB_G * TILE, g, lane);
        tile_put(tiles + TB_X * TILE, xd, lane);
        tile_put(tiles + TB_Z * TILE, Zn, lane);
        tile_put(tiles + TB_E * TILE, (have && part == 0) ? make_float4(1.0f, ax.w, ax.x, 0.0f) : f4zero(), lane);
        tile_rows(tiles + TB_G * TILE, lane, ag_);
        tile_rows(tiles + TB_X * TILE, lane, ax_);
        if (!FUSED_ABL(16)) {
            tile_put_result(tiles + TB_DQ * TILE, mat_apply(ag_, matB_lds(W.BWvT, lane), splat4(0.0f)), lane);
            tile_put_result(tiles + TB_SC * TILE, mat_apply(ax_, matB_lds(W.BPq, lane), splat4(W.pq0[lane & 15])), lane);
        }
        gv = tile_get(tiles + TB_DQ * TILE, lane);
        qp = tile_get(tiles + TB_SC * TILE, lane);
        t = quad_sum(dot4(lds4(W.Pt + 4 * part), xd)) + W.pt0;
        ge = quad_sum(dot4(g, lds4(W.we + 4 * part)));
        const float gb = quad_sum(dot4(g, lds4(W.bv + 4 * part)));
        const float Dn = quad_sum(dot4(gv, Zn)) + gb * ax.w + ge * ax.x;
        cc = gb - Dn;
        m = ax.y; rinv = ax.z;
    }
    BwdState st;
    st.dq = f4zero(); st.ds = 0.0f; st.dt = 0.0f;
    for (int k0 = 0;;) {
        bwd16_step(gt, k0, n_mine, qp, gv, t, m, rinv, ge, cc, st);
        k0 += 4;
        if (!__any(k0 < n_mine)) break;
        gather4_issue(J.s, J.x_src, r, n_mine, k0, part, en, gt);
    }
    // the record leaves behind the sweep: stored in front of it, the sweep's gather waits (vmcnt(0)) waited for the
    // stores' acknowledgements as well
    if (writer && J.rec && !FUSED_ABL(8)) {
        float* rr = J.rec + (size_t)r.row * REC_W;
        *reinterpret_cast<float4*>(rr + 4 * part) = qp;
        *reinterpret_cast<float4*>(rr + 16 + 4 * part) = gv;
        if (part == 0) *reinterpret_cast<float4*>(rr + 32) = make_float4(t, m, rinv, ge);
        if (part == 1) *reinterpret_cast<float4*>(rr + 36) = make_float4(cc, 0.0f, 0.0f, 0.0f);
    }
    // the next item's row data and first entries travel while this one finishes
    rn = slot_make(qn);                // the next item's row pointers were requested when this item started
    bwd16_fetch(J, rn, part, rd);      // (rd and en are dead here: the next item's data land in the same registers)
    en = first_entries<4>(J.s, rn, part);
    if (r.mode >= 1) {
        st.ds = shared_sum4(st.ds, r.mode);
        st.dt = shared_sum4(st.dt, r.mode);
        st.dq.x = shared_sum4(st.dq.x, r.mode); st.dq.y = shared_sum4(st.dq.y, r.mode);
        st.dq.z = shared_sum4(st.dq.z, r.mode); st.dq.w = shared_sum4(st.dq.w, r.mode);
    }
    if (r.mode == 3) {
        const int wave = threadIdx.x >> 6;
        if (lane < 4) {
            float* slot = merge_lds + wave * 20;
            if (part == 0) *reinterpret_cast<float4*>(slot) = make_float4(st.ds, st.dt, 0.0f, 0.0f);
            *reinterpret_cast<float4*>(slot + 4 + 4 * part) = st.dq;
        }
        __syncthreads();
        float ds = 0.0f, dt = 0.0f;
        float4 dq = f4zero();
        for (int w = 0; w < FW; ++w) {
            ds += merge_lds[w * 20];
            dt += merge_lds[w * 20 + 1];
            dq = f4add(dq, lds4(merge_lds + w * 20 + 4 + 4 * part));
        }
        st.ds = ds; st.dt = dt; st.dq = dq;
        __syncthreads();
        if (threadIdx.x >= 64) return;      // wave-uniform: the row's outputs and statistics belong to wavefront 0
    }
    tile_put(tiles + TB_DQ * TILE, st.dq, lane);
    if (J.dx_dst && !FUSED_ABL(4)) {          // dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt
        float ag_[4], adq_[4];
        tile_rows(tiles + TB_G * TILE, lane, ag_);
        tile_rows(tiles + TB_DQ * TILE, lane, adq_);
        tile_put_result(tiles + TB_SC * TILE,
                        mat_apply(adq_, matB_lds(W.BPqT, lane), mat_apply(ag_, matB_lds(W.BWsT, lane), splat4(0.0f))), lane);
        float4 v = tile_get(tiles + TB_SC * TILE, lane);
        fma4(st.ds, lds4(W.Pb + 4 * part), v);
        fma4(st.dt, lds4(W.Pt + 4 * part), v);
        if (J.mask_dx) {
            if (!(xpos & 1)) v.x = 0.0f;
            if (!(xpos & 2)) v.y = 0.0f;
            if (!(xpos & 4)) v.z = 0.0f;
            if (!(xpos & 8)) v.w = 0.0f;
        }
        if (writer) *reinterpret_cast<float4*>(J.dx_dst + (size_t)r.row * 16 + 4 * part) = v;
    }
    // statistics (node_kernels.hip::param_stats16_kernel): operands with m / n = channel, k = row.  A row shared by
    // several quads is counted once: only its writer keeps it, the other rows of the tiles become zeros
    if (FUSED_ABL(2)) return;
    if (r.mode >= 1 && !writer) {
        const float4 z4 = f4zero();
        tile_put(tiles + TB_G * TILE, z4, lane);
        tile_put(tiles + TB_X * TILE, z4, lane);
        tile_put(tiles + TB_Z * TILE, z4, lane);
        tile_put(tiles + TB_E * TILE, z4, lane);
        tile_put(tiles + TB_DQ * TILE, z4, lane);
    }
    tile_put(tiles + TB_SC * TILE, (writer && part == 0) ? make_float4(st.ds, st.dt, 0.0f, 0.0f) : f4zero(), lane);
    float cg[4], cdq[4], csc[4], cx[4], cz[4], ce[4];
    if (FUSED_ABL(64)) {
        for (int s = 0; s < 4; ++s) { cg[s] = cdq[s] = csc[s] = cx[s] = cz[s] = ce[s] = (float)lane; }
    } else {
    tile_cols(tiles + TB_G * TILE, lane, cg);
    tile_cols(tiles + TB_DQ * TILE, lane, cdq);
    tile_cols(tiles + TB_SC * TILE, lane, csc);
    tile_cols(tiles + TB_X * TILE, lane, cx);
    tile_cols(tiles + TB_Z * TILE, lane, cz);
    tile_cols(tiles + TB_E * TILE, lane, ce);
    }
    if (FUSED_ABL(32)) {
        asm volatile("" :: "v"(cg[0]), "v"(cdq[1]), "v"(csc[2]), "v"(cx[3]), "v"(cz[0]), "v"(ce[1]));
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[s], cx[s], acc[0], 0, 0, 0);    // T0 g x^T
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[s], cz[s], acc[1], 0, 0, 0);    // T1 g Z^T
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[s], ce[s], acc[2], 0, 0, 0);    // T2 g e^T
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cdq[s], cx[s], acc[3], 0, 0, 0);   // T3 dq' x^T
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(cdq[s], ce[s], acc[4], 0, 0, 0);   // T4 dq' e^T
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(csc[s], cx[s], acc[5], 0, 0, 0);   // T5 sc x^T
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(csc[s], ce[s], acc[6], 0, 0, 0);   // T6 sc e^T
    }
}

__global__ __launch_bounds__(FT) void fused_bwd16_kernel(BwdLaunch16 A) {
    __shared__ BwdW16 Ws_[MAXJOBS];
    __shared__ float merge_lds[FW * 20];
    __shared__ float tiles_[FW * TB_N * TILE];       // 6.5 KB per wavefront; reused for the tile reduction at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, part = lane & 3;
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob16& J = A.job[j];
        BwdW16& W = Ws_[j];
        if (wave == 0) matB_store(W.BWvT, matB_t(J.p.Wv, 16, lane), lane);
        if (wave == 1) matB_store(W.BPq, matB(J.D + OFF_PQ, 16, lane), lane);
        if (wave == 2) matB_store(W.BWsT, matB_t(J.p.Ws, 16, lane), lane);
        if (wave == 3) matB_store(W.BPqT, matB_t(J.D + OFF_PQ, 16, lane), lane);
        if (tid < 16) {
            W.pq0[tid] = J.D[OFF_PQ0 + tid];
            W.Pt[tid] = J.D[OFF_PT + tid];
            W.Pb[tid] = J.D[OFF_PB + tid];
            W.bv[tid] = J.p.bv[tid];
            W.we[tid] = J.p.we[tid];
        }
        if (tid == 0) W.pt0 = J.D[OFF_PT0];
    }
    __syncthreads();
    float* tiles = tiles_ + wave * TB_N * TILE;
    const int px = blockIdx.x % NP, bi = blockIdx.x / NP, gp = gridDim.x / NP;
    const int gw = bi * FW + wave;
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob16& J = A.job[j];
        const PartTiers P = J.s.part[px];
        f32x4m acc[STAT_TILES];
#pragma unroll
        for (int i = 0; i < STAT_TILES; ++i) acc[i] = splat4(0.0f);
        const RowSlot none = empty_slot();
        for (int k = bi; k < P.n_block; k += gp) {
            const RowSlot r = block_slot<4>(J.s, P.row0 + k, tid);
            BwdRow rd;
            bwd16_fetch(J, r, part, rd);
            int2 en = first_entries<4>(J.s, r, part);
            RowSlot rn;
            bwd16_row(J, Ws_[j], P, r, rd, en, -1, rn, part, lane, merge_lds, tiles, acc);
        }
        WaveList wl = wave_list(J.s, px, gw);
        for (int c0 = 0; c0 < wl.L; c0 += 64) {
            wave_list_chunk(wl, c0, lane);
            int it = wave_list_get(wl, 0);
            RowSlot r = it >= 0 ? item_slot<4>(J.s, P, it, lane) : none;
            BwdRow rd;
            bwd16_fetch(J, r, part, rd);
            int2 en = first_entries<4>(J.s, r, part);
            for (int k = 0; it >= 0; ++k) {
                const int itn = wave_list_get(wl, k + 1);
                RowSlot rn;
                bwd16_row(J, Ws_[j], P, r, rd, en, itn, rn, part, lane, merge_lds, tiles, acc);
                r = rn;
                it = itn;
            }
        }
        // the workgroup's partial statistics: the 16 wavefronts' tiles summed in wave order, one tile at a time
        __syncthreads();
        float* red = tiles_;                   // [FW][256]
#pragma unroll
        for (int i = 0; i < STAT_TILES; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave * 256 + ((lane >> 4) * 4 + q) * 16 + (lane & 15)] = acc[i][q];
            __syncthreads();
            if (tid < 256) {
                float v = 0.0f;
                for (int w = 0; w < FW; ++w) v += red[w * 256 + tid];
                J.stats[(size_t)blockIdx.x * STAT_FLOATS + i * 256 + tid] = v;
            }
            __syncthreads();
        }
    }
}

// ====================================================================================================
// backward, source-major, 16 channels: dX_j = sum_i alpha_ij gv_i + dl_ij q'_i   (rows = SOURCE nodes j of the conv,
// i.e. this sweep walks the orientation opposite to the conv's; the gathered records belong to destinations i)
// ====================================================================================================
struct SrcJob16 {
    ItemsDev s;                        // rows = source nodes
    const float* __restrict__ x;       // [n_rows, 16] features of the rows
    const float* __restrict__ rec;     // [n_cols, REC_W]
    float* __restrict__ dx;            // [n_rows, 16]
    // the rows' features are ReLU outputs and dx is the gradient with respect to them: stored as
    // (x > 0) * (dx + addw * addp[row]) -- the layer below then reads ONE pre-masked gradient per node instead of two
    // gradient parts and the activations (addp is always readable: x itself with addw = 0 when there is nothing to add)
    const float* __restrict__ addp;
    float addw;
    int mask;
};
struct SrcLaunch16 {
    SrcJob16 job[MAXJOBS];
    int n_jobs;
};

// two gathered destination records of a quad
__device__ __forceinline__ void src16_row(const SrcJob16& J, const RowSlot& r, float4& xj_io, int2& en, const SlotReq& qn,
                                          RowSlot& rn, int part, int lane, float* merge_lds) {
    const int n_mine = slot_count(r);
    const float4 xj = xj_io;
    const float4 addv = ld4(J.addp + (size_t)max(r.row, 0) * 16 + 4 * part);     // lands during the sweep
    float4 acc = f4zero();
    for (int k0 = 0; __any(k0 < n_mine); k0 += 2) {
        const int colm = en.x;
        const float am = __int_as_float(en.y);
        const int c0 = quad_bcast_i<0>(colm), c1 = quad_bcast_i<1>(colm);
        const float a0 = quad_bcast<0>(am), a1 = quad_bcast<1>(am);
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine;
        // unconditional (see gather4_issue): a slot past the end reads record 0 and is masked out through `al`
        const float* rr0 = J.rec + (size_t)c0 * REC_W;
        const float* rr1 = J.rec + (size_t)c1 * REC_W;
        const float4 q0 = ld4(rr0 + 4 * part), g0 = ld4(rr0 + 16 + 4 * part), s0 = ld4(rr0 + 32);
        const float cc0 = rr0[36];
        const float4 q1 = ld4(rr1 + 4 * part), g1 = ld4(rr1 + 16 + 4 * part), s1 = ld4(rr1 + 32);
        const float cc1 = rr1[36];
        const int kn = k0 + 2 + (part & 1);
        en = kn < n_mine ? J.s.sent[r.first + kn * r.stride] : make_int2(0, 0);
        {
            const float l = fmaf(a0, s0.x, quad_sum(dot4(q0, xj)));
            const float al = ok0 ? exp_acc(l - s0.y) * s0.z : 0.0f;
            const float dl = al * (quad_sum(dot4(g0, xj)) + fmaf(a0, s0.w, cc0));
            fma4(al, g0, acc);
            fma4(dl, q0, acc);
        }
        {
            const float l = fmaf(a1, s1.x, quad_sum(dot4(q1, xj)));
            const float al = ok1 ? exp_acc(l - s1.y) * s1.z : 0.0f;
            const float dl = al * (quad_sum(dot4(g1, xj)) + fmaf(a1, s1.w, cc1));
            fma4(al, g1, acc);
            fma4(dl, q1, acc);
        }
    }
    rn = slot_make(qn);
    xj_io = rn.row >= 0 ? ld4(J.x + (size_t)rn.row * 16 + 4 * part) : f4zero();
    en = first_entries<2>(J.s, rn, part);
    if (r.mode >= 1) {
        acc.x = shared_sum4(acc.x, r.mode); acc.y = shared_sum4(acc.y, r.mode);
        acc.z = shared_sum4(acc.z, r.mode); acc.w = shared_sum4(acc.w, r.mode);
    }
    if (r.mode == 3) {
        const int wave = threadIdx.x >> 6;
        if (lane < 4) *reinterpret_cast<float4*>(merge_lds + wave * 20 + 4 * part) = acc;
        __syncthreads();
        float4 v = f4zero();
        for (int w = 0; w < FW; ++w) v = f4add(v, lds4(merge_lds + w * 20 + 4 * part));
        acc = v;
        __syncthreads();
    }
    fma4(J.addw, addv, acc);
    if (J.mask) {
        if (!(xj.x > 0.0f)) acc.x = 0.0f;
        if (!(xj.y > 0.0f)) acc.y = 0.0f;
        if (!(xj.z > 0.0f)) acc.z = 0.0f;
        if (!(xj.w > 0.0f)) acc.w = 0.0f;
    }
    if (r.writer) *reinterpret_cast<float4*>(J.dx + (size_t)r.row * 16 + 4 * part) = acc;
}

__global__ __launch_bounds__(FT) void fused_src16_kernel(SrcLaunch16 A) {
    __shared__ float merge_lds[FW * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, part = lane & 3;
    const int px = blockIdx.x % NP, bi = blockIdx.x / NP, gp = gridDim.x / NP;
    const int gw = bi * FW + wave;
    for (int j = 0; j < A.n_jobs; ++j) {
        const SrcJob16& J = A.job[j];
        const PartTiers P = J.s.part[px];
        const RowSlot none = empty_slot();
        for (int k = bi; k < P.n_block; k += gp) {
            const RowSlot r = block_slot<4>(J.s, P.row0 + k, tid);
            float4 xj = ld4(J.x + (size_t)r.row * 16 + 4 * part);
            int2 en = first_entries<2>(J.s, r, part);
            RowSlot rn;
            src16_row(J, r, xj, en, empty_request(), rn, part, lane, merge_lds);
        }
        WaveList wl = wave_list(J.s, px, gw);
        for (int c0 = 0; c0 < wl.L; c0 += 64) {
            wave_list_chunk(wl, c0, lane);
            int it = wave_list_get(wl, 0);
            RowSlot r = it >= 0 ? item_slot<4>(J.s, P, it, lane) : none;
            float4 xj = r.row >= 0 ? ld4(J.x + (size_t)r.row * 16 + 4 * part) : f4zero();
            int2 en = first_entries<2>(J.s, r, part);
            for (int k = 0; it >= 0; ++k) {
                const int itn = wave_list_get(wl, k + 1);
                const SlotReq qn = itn >= 0 ? item_request<4>(J.s, P, itn, lane) : empty_request();
                RowSlot rn;
                src16_row(J, r, xj, en, qn, rn, part, lane, merge_lds);
                r = rn;
                it = itn;
            }
        }
    }
}

// ====================================================================================================
// backward, destination-major, 1 channel (layer 1: inputs are data, no input gradients): one lane per row
//   statistics in the layout of node_kernels.hip::param_stats1_kernel: T[o][n] = sum_rows g_o R_n with
//   R = {x, Z, 1, S, u} on the MFMA (the wavefront's 64 rows through two LDS tiles), six scalar sums on the VALU
// ====================================================================================================
struct BwdJob1 {
    ItemsDev s;
    const float* __restrict__ x_dst;   // [n_dst] renumbered
    const float* __restrict__ D;
    ConvParams p;
    const float* __restrict__ g;       // [n_dst, 16] gradient of the conv's output, ReLU mask applied by its producer
    const float* __restrict__ Z;       // [n_dst]
    const float* __restrict__ aux;     // [n_dst, 4]
    float* __restrict__ stats;         // [grid, STAT_FLOATS]
};
struct BwdLaunch1 {
    BwdJob1 job[MAXJOBS];
    int n_jobs;
};
struct alignas(16) BwdW1 {
    float wv[16], bv[16], we[16];
    float pq, pq0, pt, pt0;
};
constexpr int G1S = 17, R1S = 9;                       // row strides of the g tile (16 channels) and the R tile (5 columns)
constexpr int TILE1 = 64 * G1S + 64 * R1S;             // floats per wavefront

__device__ __forceinline__ void bwd1_row(const BwdJob1& J, const BwdW1& W, const RowSlot& r, int lane, float* merge_lds,
                                         float* tiles, f32x4m& accT, float (&accS)[6]
#ifdef MLLP_TIMING_BUILD
                                         , unsigned long long (&stamp_sum)[8], unsigned long long& stamp_last
#endif
) {
    FUSED_STAMP(0)      // between items
    const bool writer = r.writer;
    Ent8 E;
    ent8_load(J.s, r, 0, E);           // the first eight entries leave before the row's own data
    const int rowc = max(r.row, 0);    // row 0 for a lane without a row: nothing of it is counted (writer is false)
    float gv = 0.0f, ge = 0.0f, gb = 0.0f;
    if (r.mode == 0) {
        // Base tier (a lane per row, the item's 64 rows are consecutive): dh_a / dh_b / h are read COALESCED, lane l
        // taking float4 number l + 64 q of the item's 4 KB (row (l + 64 q) / 4, channels 4 ((l + 64 q) & 3) ...), the
        // three dot products reduced over the quad and handed to the row's own lane through the free columns of the
        // R tile.  One lane reading its own 64-byte row made every load instruction touch 64 cache lines: 11.4 k of
        // the 18.5 k cycles of an item went into this block (in-kernel stamps), the vector-memory path being shared by
        // the CU's 12 wavefronts.
        const int base_row = __builtin_amdgcn_readfirstlane(r.row);          // lane 0 of a base item always has a row
        const int n_valid = __popcll(__ballot(r.row >= 0));
        float* gt = tiles;
        float* rt = tiles + 64 * G1S;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q, row_l = idx >> 2, ch = (idx & 3) * 4;
            const size_t ro = (size_t)min(base_row + row_l, J.s.n_dst - 1) * 16 + ch;
            const float4 a = ld4(J.g + ro);
            const bool on = row_l < n_valid;
            const float4 g = make_float4(on ? a.x : 0.0f, on ? a.y : 0.0f, on ? a.z : 0.0f, on ? a.w : 0.0f);
            const float pv = quad_sum(dot4(g, lds4(W.wv + ch))), pe = quad_sum(dot4(g, lds4(W.we + ch)));
            const float pb = quad_sum(dot4(g, lds4(W.bv + ch)));
            float* d = gt + row_l * G1S + ch;
            d[0] = g.x; d[1] = g.y; d[2] = g.z; d[3] = g.w;
            if ((lane & 3) == 0) { rt[row_l * R1S + 5] = pv; rt[row_l * R1S + 6] = pe; rt[row_l * R1S + 7] = pb; }
        }
        gv = rt[lane * R1S + 5]; ge = rt[lane * R1S + 6]; gb = rt[lane * R1S + 7];      // same wavefront: LDS in order
    } else {   // the row of g goes to the tile (zeros unless this lane owns the row)
        const float4* pa = reinterpret_cast<const float4*>(J.g + (size_t)rowc * 16);
        float* gt = tiles + lane * G1S;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = pa[q];
            const float g0 = a.x, g1 = a.y, g2 = a.z, g3 = a.w;
            gv = fmaf(g0, W.wv[4 * q], fmaf(g1, W.wv[4 * q + 1], fmaf(g2, W.wv[4 * q + 2], fmaf(g3, W.wv[4 * q + 3], gv))));
            ge = fmaf(g0, W.we[4 * q], fmaf(g1, W.we[4 * q + 1], fmaf(g2, W.we[4 * q + 2], fmaf(g3, W.we[4 * q + 3], ge))));
            gb = fmaf(g0, W.bv[4 * q], fmaf(g1, W.bv[4 * q + 1], fmaf(g2, W.bv[4 * q + 2], fmaf(g3, W.bv[4 * q + 3], gb))));
            gt[4 * q] = writer ? g0 : 0.0f;
            gt[4 * q + 1] = writer ? g1 : 0.0f;
            gt[4 * q + 2] = writer ? g2 : 0.0f;
            gt[4 * q + 3] = writer ? g3 : 0.0f;
        }
    }
    const float x = J.x_dst[rowc];
    const float Zn = J.Z[rowc];
    const float4 ax = reinterpret_cast<const float4*>(J.aux)[rowc];   // {u, rowmax, rinv, S}
    {
        float* rt = tiles + 64 * G1S + lane * R1S;
        rt[0] = writer ? x : 0.0f;
        rt[1] = writer ? Zn : 0.0f;
        rt[2] = writer ? 1.0f : 0.0f;
        rt[3] = writer ? ax.w : 0.0f;
        rt[4] = writer ? ax.x : 0.0f;
    }
    const float Dn = gv * Zn + gb * ax.w + ge * ax.x;
    const float cc = gb - Dn;
    const float qp = fmaf(W.pq, x, W.pq0), t = fmaf(W.pt, x, W.pt0);
    float ds = 0.0f, dt = 0.0f, dq = 0.0f;
    const int n_mine = slot_count(r);
    FUSED_STAMP(1)      // row data, g tile
    for (int k0 = 0; __any(k0 < n_mine); k0 += 8) {
        Ent8 N;
        const bool more = __any(k0 + 8 < n_mine);
        if (more) ent8_load(J.s, r, k0 + 8, N);
#define MLLP_BWD1_SLOT(K, E_)                                                              \
    {                                                                                      \
        const float l_ = fmaf(qp, E_.y, E_.x * t);                                         \
        const float al_ = (K) < n_mine ? exp_acc(l_ - ax.y) * ax.z : 0.0f;                 \
        const float dl_ = al_ * fmaf(gv, E_.y, fmaf(E_.x, ge, cc));                        \
        ds += dl_;                                                                         \
        dt = fmaf(dl_, E_.x, dt);                                                          \
        dq = fmaf(dl_, E_.y, dq);                                                          \
    }
        MLLP_BWD1_SLOT(k0, E.e0)
        MLLP_BWD1_SLOT(k0 + 1, E.e1)
        MLLP_BWD1_SLOT(k0 + 2, E.e2)
        MLLP_BWD1_SLOT(k0 + 3, E.e3)
        if (__any(k0 + 4 < n_mine)) {
            MLLP_BWD1_SLOT(k0 + 4, E.e4)
            MLLP_BWD1_SLOT(k0 + 5, E.e5)
            MLLP_BWD1_SLOT(k0 + 6, E.e6)
            MLLP_BWD1_SLOT(k0 + 7, E.e7)
        }
#undef MLLP_BWD1_SLOT
        if (more) E = N;
    }
    FUSED_STAMP(2)      // sweep
    if (r.mode >= 1) {
        ds = shared_sum1(ds, r.mode); dt = shared_sum1(dt, r.mode); dq = shared_sum1(dq, r.mode);
    }
    if (r.mode == 3) {
        const int wave = threadIdx.x >> 6;
        if (lane == 0) *reinterpret_cast<float4*>(merge_lds + wave * 4) = make_float4(ds, dt, dq, 0.0f);
        __syncthreads();
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int w = 0; w < FW; ++w) {
            const float4 v = lds4(merge_lds + w * 4);
            a += v.x; b += v.y; c += v.z;
        }
        ds = a; dt = b; dq = c;
        __syncthreads();
        if (threadIdx.x >= 64) return;      // wave-uniform: the row is counted by wavefront 0
    }
    if (writer) {
        accS[0] = fmaf(dq, x, accS[0]);     // T3[0][0]
        accS[1] += dq;                      // T4[0][0]
        accS[2] = fmaf(ds, x, accS[2]);     // T5[0][0]
        accS[3] = fmaf(dt, x, accS[3]);     // T5[1][0]
        accS[4] += ds;                      // T6[0][0]
        accS[5] += dt;                      // T6[1][0]
    }
    FUSED_STAMP(3)      // merges, scalar sums
    // T[o][n] += sum over the 64 rows of the wavefront: 16 MFMA steps of 4 rows
    const float* gt = tiles;
    const float* rt = tiles + 64 * G1S;
    const int kq = lane >> 4, rr = lane & 15;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float a = gt[(4 * s + kq) * G1S + rr];
        const float b = rr < 5 ? rt[(4 * s + kq) * R1S + rr] : 0.0f;
        accT = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accT, 0, 0, 0);
    }
    FUSED_STAMP(4)      // statistics on the MFMA
}

__global__ __launch_bounds__(FT) void fused_bwd1_kernel(BwdLaunch1 A) {
    __shared__ BwdW1 Ws_[MAXJOBS];
    __shared__ float merge_lds[FW * 4];
    __shared__ float tiles_[FW * TILE1];     // 6.5 KB per wavefront; reused for the reduction at the end of a job
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob1& J = A.job[j];
        BwdW1& W = Ws_[j];
        if (tid < 16) { W.wv[tid] = J.p.Wv[tid]; W.bv[tid] = J.p.bv[tid]; W.we[tid] = J.p.we[tid]; }
        if (tid == 0) { W.pq = J.D[OFF_PQ]; W.pq0 = J.D[OFF_PQ0]; W.pt = J.D[OFF_PT]; W.pt0 = J.D[OFF_PT0]; }
    }
    __syncthreads();
    float* tiles = tiles_ + wave * TILE1;
    const int px = blockIdx.x % NP, bi = blockIdx.x / NP, gp = gridDim.x / NP;
    const int gw = bi * FW + wave;
#ifdef MLLP_TIMING_BUILD
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_begin = stamp_last;
#define STAMP_ARGS1 , stamp_sum, stamp_last
#else
#define STAMP_ARGS1
#endif
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob1& J = A.job[j];
        const PartTiers P = J.s.part[px];
        f32x4m accT = splat4(0.0f);
        float accS[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = bi; k < P.n_block; k += gp)
            bwd1_row(J, Ws_[j], block_slot<1>(J.s, P.row0 + k, tid), lane, merge_lds, tiles, accT, accS STAMP_ARGS1);
        WaveList wl = wave_list(J.s, px, gw);
        for (int c0 = 0; c0 < wl.L; c0 += 64) {
            wave_list_chunk(wl, c0, lane);
            int it = wave_list_get(wl, 0);
            SlotReq q = item_request<1>(J.s, P, max(it, 0), lane);
            for (int k = 0; it >= 0; ++k) {
                const int itn = wave_list_get(wl, k + 1);
                const RowSlot r = slot_make(q);
                q = item_request<1>(J.s, P, max(itn, 0), lane);
                bwd1_row(J, Ws_[j], r, lane, merge_lds, tiles, accT, accS STAMP_ARGS1);
#ifdef MLLP_TIMING_BUILD
                stamp_sum[7] += 1;
#endif
                it = itn;
            }
        }
        // wavefronts -> workgroup (LDS, wave order), one partial per workgroup in the layout of param_stats1_kernel
        __syncthreads();
        float* red = tiles_;                 // [FW][256 + 6]
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave * 262 + ((lane >> 4) * 4 + q) * 16 + (lane & 15)] = accT[q];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float v = wave_sum(accS[i]);
            if (lane == 0) red[wave * 262 + 256 + i] = v;
        }
        __syncthreads();
        float* dst = J.stats + (size_t)blockIdx.x * STAT_FLOATS;
        for (int i = tid; i < STAT_FLOATS; i += FT) dst[i] = 0.0f;
        __syncthreads();
        if (tid < 262) {
            float v = 0.0f;
            for (int w = 0; w < FW; ++w) v += red[w * 262 + tid];
            const int o = tid >> 4, n = tid & 15;
            if (tid >= 256) {
                const int i = tid - 256;     // T3[0][0], T4[0][0], T5[0][0], T5[1][0], T6[0][0], T6[1][0]
                const int slot = i == 0 ? 768 : i == 1 ? 1024 : i == 2 ? 1280 : i == 3 ? 1280 + 16 : i == 4 ? 1536 : 1536 + 16;
                dst[slot] = v;
            } else if (n == 0) dst[o * 16] = v;                 // T0[o][0] = sum g_o x
            else if (n == 1) dst[256 + o * 16] = v;             // T1[o][0] = sum g_o Z
            else if (n < 5) dst[512 + o * 16 + (n - 2)] = v;    // T2[o][0..2] = sum g_o {1, S, u}
        }
        __syncthreads();
#ifdef MLLP_TIMING_BUILD
        stamp_last = __builtin_amdgcn_s_memtime();      // the job's reduction is not part of "between items"
#endif
    }
#ifdef MLLP_TIMING_BUILD
    if (lane == 0) {
        stamp_sum[6] = __builtin_amdgcn_s_memtime() - stamp_begin;
        unsigned long long* d = g_fused_stamps1 + (size_t)(blockIdx.x * FW + wave) * 8;
        for (int k = 0; k < 8; ++k) d[k] = stamp_sum[k];
    }
#endif
#undef STAMP_ARGS1
}

// ====================================================================================================
// head backward from given dL/dlogits (caller's variable order): dh3 = dz w_fc (unmasked), fc gradient partials
// ====================================================================================================
__global__ __launch_bounds__(RT) void fused_head_bwd_kernel(int n, const float* __restrict__ h3, const float* __restrict__ fcw,
                                                            const float* __restrict__ dlogits, const int* __restrict__ perm,
                                                            float* __restrict__ dh3, float* __restrict__ head_part) {
    __shared__ float head_lds[(RT / 64) * 18];
    const int tid = threadIdx.x, lane = tid & 63, part = lane & 3;
    HeadAcc ha;
    ha.w = f4zero(); ha.b = 0.0f; ha.l = 0.0f;
    const float4 fw = ld4(fcw + 4 * part);
    for (int row = blockIdx.x * (RT / 4) + (tid >> 2); row < n; row += gridDim.x * (RT / 4)) {
        const float dz = dlogits[perm[row]];
        const float4 hv = ld4(h3 + (size_t)row * 16 + 4 * part);
        *reinterpret_cast<float4*>(dh3 + (size_t)row * 16 + 4 * part) = f4scale(fw, dz);
        fma4(dz, hv, ha.w);
        if (part == 0) ha.b += dz;
    }
    head_partials_store<RT / 64>(ha, head_lds, head_part, tid);
}

// ====================================================================================================
// reduction of the per-workgroup partials: out[c][i] = sum_b stats_c[b][i], fixed order.
// grid = convs x 7 tiles (1024 threads = 4 slices of the partials x 256 columns) + one workgroup for the fc partials
// ====================================================================================================
struct ReduceArgs {
    const float* stats[MODEL_CONVS];
    float* out[MODEL_CONVS];
    int nblk;
    const float* head_part;   // [nblk][18] or nullptr
    float* head_out;          // [17] gradient of fc
    float* loss_out;          // nullable
};

__device__ __forceinline__ void fused_reduce_body(const ReduceArgs& A, int n_conv) {
    __shared__ float sh[32][32];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == n_conv * STAT_TILES) {     // fc partials: 32 slices x 18 columns
        if (!A.head_part) return;
        const int col = tid & 31, slice = tid >> 5;
        float v = 0.0f;
        if (col < 18)
            for (int b = slice; b < A.nblk; b += 32) v += A.head_part[(size_t)b * 18 + col];
        sh[slice][col] = v;
        __syncthreads();
        if (tid < 18) {
            float t = 0.0f;
            for (int q = 0; q < 32; ++q) t += sh[q][tid];
            if (tid < 17) A.head_out[tid] = t;
            else if (A.loss_out) A.loss_out[0] = t;
        }
        return;
    }
    float* sh4 = &sh[0][0];       // [4][256]
    const int col = tid & 255, slice = tid >> 8;
    const int c = blockIdx.x / STAT_TILES, tile = blockIdx.x % STAT_TILES;
    const float* src = A.stats[c] + tile * 256 + col;
    float v = 0.0f;
    int b = slice;
    // the partials were written by other XCDs: every load is a 2 k-cycle trip to the Infinity Cache, so 32 of them are
    // in flight at once (a thread of the 256-workgroup grid has 64 to add, in a fixed order)
    for (; b + 124 < A.nblk; b += 128) {
        float t[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) t[q] = src[(size_t)(b + 4 * q) * STAT_FLOATS];
#pragma unroll
        for (int q = 0; q < 32; ++q) v += t[q];
    }
    for (; b + 28 < A.nblk; b += 32) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = src[(size_t)(b + 4 * q) * STAT_FLOATS];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += t[q];
    }
    for (; b < A.nblk; b += 4) v += src[(size_t)b * STAT_FLOATS];
    sh4[slice * 256 + col] = v;
    __syncthreads();
    if (tid < 256) A.out[c][tile * 256 + tid] = (sh4[tid] + sh4[256 + tid]) + (sh4[512 + tid] + sh4[768 + tid]);
}
__global__ __launch_bounds__(RT) void fused_reduce_kernel(ReduceArgs A, int n_conv) { fused_reduce_body(A, n_conv); }

// ====================================================================================================
// The tail of a single-rank training step in ONE launch (mllp_gnn_train_step): reduce -> gradients of the five convs ->
// Adam -> folded weights of the next step.  All 36 workgroups sum the partials, ONE grid barrier (the grid is far smaller
// than the GPU, so every workgroup is resident; the spin is bounded all the same), then a workgroup per conv does
// gradients -> Adam -> folded weights of its conv with workgroup barriers only.  Separately the four launches take
// 27 us of a 342 us step; a first version with a grid barrier between each of the four phases took the same 28 us
// (a barrier = an L2 write-back + invalidate on 8 XCDs), this one ~22 us.
// ====================================================================================================
struct TailArgs {
    ReduceArgs R;
    ConvParams p[MODEL_CONVS];      // (point into `params`: read by phase B before Adam, by phase D after it)
    int cin[MODEL_CONVS];
    float* grads[MODEL_CONVS];
    float* zero;                    // gradient of the never-used gconv3_s2w
    int n_zero;
    float* params; const float* grads_all; float* m; float* v; float* state;
    float eps;
    int n_params;
    float* D[MODEL_CONVS];          // folded weights (the batch's workspace)
    unsigned long long* sync;       // [0] arrivals (cumulative over launches), [1] launches so far
    unsigned* err;                  // host-mapped word: != 0 once a barrier has timed out (checked by the host at the next call)
};
// Grid barrier of an ordinary (non-cooperative) launch: 36 workgroups on 256 CUs are co-resident unless something else
// holds the device.  The spin is bounded so that a workgroup that is never scheduled cannot hang the others; a timeout is
// NOT silent: the workgroup raises the error word and the kernel skips Adam and the weight folding -- also in
// the workgroups that arrive late and find the count complete -- and the host refuses the next call.
// The error word the workgroups LOOK at lives in device memory (word 2 of the barrier state): the first version read the
// host-mapped word after every barrier -- one PCIe round trip per workgroup on the step's critical path, 20.7 -> 27.3 us for
// the kernel.  The host-mapped word is only written, on a timeout, for the host's check before the next launch.
__device__ __forceinline__ bool tail_barrier(unsigned long long* cnt, unsigned long long target, unsigned* err_host) {
    __shared__ int ok_s;
    unsigned long long* err_dev = cnt + 2;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(8);
        if (spins >= (1 << 22)) {
            __hip_atomic_store(err_dev, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence();
        ok_s = __hip_atomic_load(err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
    }
    __syncthreads();
    return ok_s != 0;
}
__global__ __launch_bounds__(RT) void fused_tail_kernel(TailArgs A, int n_conv) {
    const unsigned long long G = gridDim.x, epoch = A.sync[1];
    const int b = blockIdx.x;
    // (every workgroup reads the optimizer's scalars before the barrier; one of them writes the step count behind it)
    const float step = A.state[0] + 1.0f, lr = A.state[1], b1 = A.state[2], b2 = A.state[3];
    fused_reduce_body(A.R, n_conv);                                         // all workgroups: the partials
    if (!tail_barrier(A.sync, (epoch + 1) * G, A.err)) return;
    // From here on a workgroup works on ITS conv alone: gradients -> Adam on the conv's parameter range -> folded
    // weights; no conv needs another one's result (the never-used gconv3_s2w has zero gradients and zero moments: Adam
    // leaves it where it is, exactly as the separate launch does).
    if (b < n_conv) {
        finalize_conv_body(A.cin[b], A.p[b], A.R.out[b], 1, A.grads[b]);
        __syncthreads();
        const int off = (int)(A.grads[b] - A.grads_all), n = 4 * FEAT * A.cin[b] + 5 * FEAT;
        adam_slice(A.params + off, A.grads_all + off, A.m + off, A.v + off, step, lr, b1, b2, A.eps, 1.0f, n);
        __syncthreads();
        if (threadIdx.x < BLOCK) param_prep_body(A.p[b], A.cin[b], A.D[b]);
    } else if (b == n_conv) {
        for (int k = threadIdx.x; k < A.n_zero; k += RT) A.zero[k] = 0.0f;
        const int off = (int)(A.R.head_out - A.grads_all);                  // fc: 16 weights + bias, behind the convs
        adam_slice(A.params + off, A.grads_all + off, A.m + off, A.v + off, step, lr, b1, b2, A.eps, 1.0f, A.n_params - off);
        if (threadIdx.x == 0) {
            A.state[0] = step;
            A.sync[1] = epoch + 1;
        }
    }
}

// ====================================================================================================
// building the renumbered graph, binding inputs
// ====================================================================================================
// one wavefront per renumbered row k: entries of the original row perm[k], source ids renumbered
__global__ __launch_bounds__(256) void fused_fill_kernel(int n_dst, const int* __restrict__ perm, const int* __restrict__ ptr,
                                                         const int* __restrict__ idx, const float* __restrict__ val,
                                                         const int* __restrict__ inv_src, const int* __restrict__ sptr,
                                                         int2* __restrict__ sent) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_dst) return;
    const int r = perm[k], beg = ptr[r], deg = ptr[r + 1] - beg, o = sptr[k];
    for (int e = lane; e < deg; e += 64) sent[o + e] = make_int2(inv_src[idx[beg + e]], __float_as_int(val[beg + e]));
}
__global__ void fused_permute_kernel(int n, const int* __restrict__ perm, const float* __restrict__ src, float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[perm[i]];
}
// sax[e] = {a_e, x_src[source of e]}: layer 1's per-nonzero source feature beside the value
__global__ void fused_pregather_kernel(int64_t nnz, const int2* __restrict__ sent, const float* __restrict__ xs,
                                       float2* __restrict__ sax) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int2 v = sent[e];
        sax[e] = make_float2(__int_as_float(v.y), xs[v.x]);
    }
}

#ifdef MLLP_TIMING_BUILD
extern "C" int mllp_timing_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fused_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int mllp_timing_read_stamps1(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fused_stamps1), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, what);
}

template <class T>
static int dev_alloc(mllp_graph* g, size_t count, T** out, const T* host = nullptr) {
    void* p = nullptr;
    MLLP_HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    g->allocs.push_back(p);
    if (host && count) MLLP_HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<T*>(p);
    return MLLP_OK;
}

static void set_tiers(FusedTiersDev& d, const FusedTiers& t) {
    d.n_block = t.n_block; d.n_wave = t.n_wave; d.n_group = t.n_group; d.n_base = t.n_base;
}

int fused_graph_build(mllp_graph* g, const int* h_csr_ptr, const int* h_csc_ptr) {
    if (g->fused_built) return MLLP_OK;
    if ((int64_t)g->h_csr_ptr.size() != g->M + 1 && !h_csr_ptr) return fail(MLLP_EINVAL, "fused path: no host row pointers");
    HostFusedOrient hc, hv;
    {
#ifdef MLLP_SCALAR_PARTITION       // (experiment: the one-cost rule of rounds 2-3)
        const std::vector<int> part = host_partition_instances(host_instance_cost(h_csr_ptr, g->h_inst_ptr_m),
                                                               host_instance_cost(h_csc_ptr, g->h_inst_ptr_n), FUSED_PARTS);
#else
        const std::vector<int> part = host_partition_instances_v(host_instance_loads(h_csr_ptr, g->h_inst_ptr_m),
                                                                 host_instance_loads(h_csc_ptr, g->h_inst_ptr_n), FUSED_PARTS);
#endif
        host_build_fused_orient(h_csr_ptr, (int)g->M, g->h_inst_ptr_m, part, &hc);     // constraints by (partition, row length)
        host_build_fused_orient(h_csc_ptr, (int)g->N, g->h_inst_ptr_n, part, &hv);     // variables by (partition, column length)
    }
    int rc;
    if ((rc = dev_alloc(g, (size_t)g->M, &g->perm_c, hc.perm.data()))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N, &g->perm_v, hv.perm.data()))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->M, &g->inv_c, hc.inv.data()))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N, &g->inv_v, hv.inv.data()))) return rc;
    FusedOrient& A = g->FA;
    FusedOrient& At = g->FAt;
    A.n_dst = (int)g->M; A.n_src = (int)g->N; A.nnz = (int)g->nnz;
    At.n_dst = (int)g->N; At.n_src = (int)g->M; At.nnz = (int)g->nnz;
    for (int q = 0; q < NP; ++q) {
        set_tiers(A.t16[q], hc.t16[q]); set_tiers(A.t1[q], hc.t1[q]);
        set_tiers(At.t16[q], hv.t16[q]); set_tiers(At.t1[q], hv.t1[q]);
    }
    for (int q = 0; q <= NP; ++q) { A.row0[q] = hc.row0[q]; At.row0[q] = hv.row0[q]; }
    {   // items of every wavefront at the grids the sweeps are launched with (fused_grid: fixed per device)
        const int gp = fused_grid(g) / NP;
        const HostFusedOrient* ho[2] = {&hc, &hv};
        FusedOrient* dv[2] = {&A, &At};
        for (int k = 0; k < 2; ++k) {
            HostWaveLists l16, l16x2, l1;
            host_build_wave_lists(*ho[k], false, gp * FW, FW, 4, &l16);
            host_build_wave_lists(*ho[k], false, 2 * gp * FW, FW, 2, &l16x2);      // fused_src16_kernel: 2 nonzeros per step
            host_build_wave_lists(*ho[k], true, gp * FW, FW, 8, &l1);
            FusedOrient& d = *dv[k];
            d.L16 = l16.L; d.nw16 = l16.waves_per_part;
            d.L16x2 = l16x2.L; d.nw16x2 = l16x2.waves_per_part;
            d.L1 = l1.L; d.nw1 = l1.waves_per_part;
            if ((rc = dev_alloc(g, l16.order.size(), &d.lst16, l16.order.data()))) return rc;
            if ((rc = dev_alloc(g, l16x2.order.size(), &d.lst16x2, l16x2.order.data()))) return rc;
            if ((rc = dev_alloc(g, l1.order.size(), &d.lst1, l1.order.data()))) return rc;
        }
    }
    if ((rc = dev_alloc(g, (size_t)g->M + 1, &A.sptr, hc.sptr.data()))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N + 1, &At.sptr, hv.sptr.data()))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->nnz * 2, &A.sent))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->nnz * 2, &At.sent))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->nnz * 2, &A.sax))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->nnz * 2, &At.sax))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N, &g->inv_n_p))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N, &g->x1_p))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->M, &g->x2_p))) return rc;
    if ((rc = dev_alloc(g, (size_t)g->N, &g->labels_p))) return rc;
    if (!g->tail_sync) {
        if ((rc = dev_alloc(g, (size_t)4, &g->tail_sync))) return rc;
        MLLP_HIP_TRY(hipMemset(g->tail_sync, 0, 32));
        MLLP_HIP_TRY(hipHostMalloc((void**)&g->tail_err_host, 64, hipHostMallocMapped));
        *g->tail_err_host = 0u;
        MLLP_HIP_TRY(hipHostGetDevicePointer((void**)&g->tail_err_dev, g->tail_err_host, 0));
    }
    if (g->M > 0) {
        hipLaunchKernelGGL(fused_fill_kernel, dim3((unsigned)((g->M + 3) / 4)), dim3(256), 0, 0, (int)g->M, g->perm_c, g->A.ptr,
                           g->A.idx, g->A.val, g->inv_v, A.sptr, reinterpret_cast<int2*>(A.sent));
        if ((rc = check_launch("fused_fill A"))) return rc;
    }
    if (g->N > 0) {
        hipLaunchKernelGGL(fused_fill_kernel, dim3((unsigned)((g->N + 3) / 4)), dim3(256), 0, 0, (int)g->N, g->perm_v, g->At.ptr,
                           g->At.idx, g->At.val, g->inv_c, At.sptr, reinterpret_cast<int2*>(At.sent));
        if ((rc = check_launch("fused_fill At"))) return rc;
        hipLaunchKernelGGL(fused_permute_kernel, dim3(256), dim3(256), 0, 0, (int)g->N, g->perm_v, g->inv_n, g->inv_n_p);
        if ((rc = check_launch("fused_permute inv_n"))) return rc;
    }
    MLLP_HIP_TRY(hipDeviceSynchronize());
    g->fused_built = true;
    return MLLP_OK;
}

// renumbered copies of the inputs, made when the caller's pointers change (the contents are taken as constant while
// the pointers are: the model's inputs are data -- reference linear_program_methods.py:90-91; the contract and
// mllp_graph_invalidate_inputs are in include/mllp_hip.h).  While the stream is being captured into a hipGraph the
// copies are made inside the capture on every call and the cache is left empty: a captured launch has not run, so a
// later eager call with the same pointers must not find the inputs "bound".
int fused_bind(mllp_graph* g, const float* x1, const float* x2, const float* labels, hipStream_t s) {
    int rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (capturing) g->bound_x1 = g->bound_x2 = g->bound_labels = nullptr;
    if (x1 != g->bound_x1 || x2 != g->bound_x2) {
        if (g->N > 0) hipLaunchKernelGGL(fused_permute_kernel, dim3(256), dim3(256), 0, s, (int)g->N, g->perm_v, x1, g->x1_p);
        if (g->M > 0) hipLaunchKernelGGL(fused_permute_kernel, dim3(256), dim3(256), 0, s, (int)g->M, g->perm_c, x2, g->x2_p);
        if (g->nnz > 0) {
            // FA: rows = constraints, sources = variables (x1); FAt: rows = variables, sources = constraints (x2)
            hipLaunchKernelGGL(fused_pregather_kernel, dim3(1024), dim3(256), 0, s, g->nnz,
                               reinterpret_cast<const int2*>(g->FA.sent), g->x1_p, reinterpret_cast<float2*>(g->FA.sax));
            hipLaunchKernelGGL(fused_pregather_kernel, dim3(1024), dim3(256), 0, s, g->nnz,
                               reinterpret_cast<const int2*>(g->FAt.sent), g->x2_p, reinterpret_cast<float2*>(g->FAt.sax));
        }
        if ((rc = check_launch("fused_bind inputs"))) return rc;
        if (!capturing) { g->bound_x1 = x1; g->bound_x2 = x2; }
    }
    if (labels && labels != g->bound_labels) {
        if (g->N > 0) hipLaunchKernelGGL(fused_permute_kernel, dim3(256), dim3(256), 0, s, (int)g->N, g->perm_v, labels, g->labels_p);
        if ((rc = check_launch("fused_bind labels"))) return rc;
        if (!capturing) g->bound_labels = labels;
    }
    return MLLP_OK;
}

// ====================================================================================================
// the whole model on the fused path
// ====================================================================================================
// one workgroup per CU, a multiple of the partition count (workgroup b works on partition b % NP)
int fused_grid(const mllp_graph* g) { return std::max(std::min(g->n_cu, STAT_BLOCKS_MAX) / NP, 1) * NP; }

static FwdJob16 fwd_job16(const FusedOrient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                          float* h) {
    FwdJob16 J = {};
    J.s = items_dev(o, 0);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 16);
    J.h = h; J.Z = w.Z; J.aux = w.aux;
    J.head = 0;
#ifdef MLLP_TIMING_BUILD
    J.abl = g_fused_abl;
#endif
    return J;
}
static FwdJob1 fwd_job1(const FusedOrient& o, const float* cp, const ConvWs& w, const float* x_dst, float* h) {
    FwdJob1 J = {};
    J.s = items_dev(o, 1);
    J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 1);
    J.h = h; J.Z = w.Z; J.aux = w.aux;
    return J;
}

int fused_forward(mllp_graph* g, const FusedModel& m, int head_mode, hipStream_t s, bool skip_prep) {
    const int G = fused_grid(g);
    int rc;
    if ((rc = fused_bind(g, m.x1, m.x2, head_mode == 2 ? m.labels : nullptr, s))) return rc;
    if (!skip_prep) {   // folded weights of all five convs (skipped when the previous step's tail left them in this workspace)
        const float* cps[MODEL_CONVS] = {m.cp[0], m.cp[1], m.cp[2], m.cp[3], m.cp[4]};
        const int cins[MODEL_CONVS] = {1, 1, 16, 16, 16};
        float* ders[MODEL_CONVS] = {m.c[0].derived, m.c[1].derived, m.c[2].derived, m.c[3].derived, m.c[4].derived};
        if ((rc = launch_param_prep_batch(MODEL_CONVS, cps, cins, ders, s))) return rc;
    }
    {   // linear_program_methods.py:241-242  layer 1, both directions
        FwdLaunch1 L = {};
        L.n_jobs = 2;
        L.job[0] = fwd_job1(g->FAt, m.cp[0], m.c[0], g->x1_p, m.h1v);     // w2s: dst = variables
        L.job[1] = fwd_job1(g->FA, m.cp[1], m.c[1], g->x2_p, m.h1c);      // s2w: dst = constraints
        hipLaunchKernelGGL(fused_fwd1_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd1"))) return rc;
    }
    {   // :244-245  layer 2 (simultaneous update)
        FwdLaunch16 L = {};
        L.n_jobs = 2;
        L.job[0] = fwd_job16(g->FAt, m.cp[2], m.c[2], m.h1c, m.h1v, m.h2v);
        L.job[1] = fwd_job16(g->FA, m.cp[3], m.c[3], m.h1v, m.h1c, m.h2c);
        L.job[0].perm = g->perm_v; L.job[1].perm = g->perm_c;      // read (unconditional prefetch), used by the head only
        hipLaunchKernelGGL(fused_fwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd16 layer 2"))) return rc;
    }
    {   // :247 layer 3 (variables only) + :250 fc (+ loss)
        FwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = fwd_job16(g->FAt, m.cp[4], m.c[4], m.h2c, m.h2v, m.h3v);
        FwdJob16& J = L.job[0];
        J.head = head_mode;
        J.fcw = m.fcw; J.fcb = m.fcb; J.inv_n = g->inv_n_p; J.labels = g->labels_p; J.perm = g->perm_v;
        J.inv_batch = m.inv_batch;
        J.logits = m.logits; J.g_out = m.d3v; J.head_part = m.head_part;
        hipLaunchKernelGGL(fused_fwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd16 layer 3"))) return rc;
    }
    return MLLP_OK;
}

int fused_head_backward(const mllp_graph* g, const FusedModel& m, const float* dlogits, hipStream_t s) {
    const int G = fused_grid(g);
    hipLaunchKernelGGL(fused_head_bwd_kernel, dim3(G), dim3(RT), 0, s, (int)g->N, m.h3v, m.fcw, dlogits, g->perm_v, m.d3v,
                       m.head_part);
    return check_launch("fused_head_bwd");
}

static BwdJob16 bwd_job16(const FusedOrient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                          const float* h, const float* dh_a, const float* dh_b, float* dx_dst, bool need_rec, bool mask_dx) {
    BwdJob16 J = {};
    J.s = items_dev(o, 0);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 16);
    J.h = h; J.dh_a = dh_a; J.dh_b = dh_b; J.Z = w.Z; J.aux = w.aux;
    J.rec = need_rec ? w.rec : nullptr;
    J.dx_dst = dx_dst;
    J.mask_dx = mask_dx ? 1 : 0;
    J.stats = w.stats;
#ifdef MLLP_TIMING_BUILD
    J.abl = g_fused_abl;
#endif
    return J;
}
static SrcJob16 src_job16(const FusedOrient& o_src_major, const ConvWs& w, const float* x_rows, float* dx, const float* add) {
    SrcJob16 J = {};
    J.s = items_dev(o_src_major, 2);
    J.x = x_rows; J.rec = w.rec; J.dx = dx;
    J.addp = add ? add : x_rows; J.addw = add ? 1.0f : 0.0f; J.mask = 1;
    return J;
}
static BwdJob1 bwd_job1(const FusedOrient& o, const float* cp, const ConvWs& w, const float* x_dst, const float* g_masked) {
    BwdJob1 J = {};
    J.s = items_dev(o, 1);
    J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 1);
    J.g = g_masked; J.Z = w.Z; J.aux = w.aux; J.stats = w.stats;
    return J;
}

// Backward chain (linear_program_methods.py:241-247 read backwards), one stream.  Every gradient of a hidden activation is
// stored already multiplied by the ReLU mask of that activation by the kernel that produces it (it has the activation
// in registers as its own x), and the two parts of the layer-1 gradients are summed by the second producer: the
// consumers read one array per node and no activations for masking.
//   K1  C3  dst  (dh = d3v [premasked when it came from the fused head]) -> rec3, d2v (masked by h2v)
//   K2  C3  src  -> d2c (masked by h2c)                 K2' C2V dst (dh = d2v) -> rec2v, d1v (masked by h1v)
//   K3  C2C dst  (dh = d2c) -> rec2c, d1c (masked)      K3' C2V src -> d1c_b = (h1c > 0) (dX + d1c)
//   K4  C2C src  -> d1v_b = (h1v > 0) (dX + d1v)
//   K5  C1C dst (g = d1c_b) and C1V dst (g = d1v_b) in one launch
//   K6  reduce the statistics partials (+ fc partials), K7 finalize (node_kernels.hip)
int fused_backward(const mllp_graph* g, const FusedModel& m, bool premasked, float* grads, float* loss, hipStream_t s,
                   const FusedAdam* adam) {
    const int G = fused_grid(g);
    int rc;
    const FusedOrient& A = g->FA;      // rows = constraints
    const FusedOrient& At = g->FAt;    // rows = variables
    {   // K1
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(At, m.cp[4], m.c[4], m.h2c, m.h2v, premasked ? nullptr : m.h3v, m.d3v, nullptr, m.d2v, true, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C3"))) return rc;
    }
    {   // K2: C3 source-major (rows = constraints) and C2V destination-major
        SrcLaunch16 S = {};
        S.n_jobs = 1;
        S.job[0] = src_job16(A, m.c[4], m.h2c, m.d2c, nullptr);
        hipLaunchKernelGGL(fused_src16_kernel, dim3(2 * G), dim3(FT), 0, s, S);   // 72 VGPRs: two workgroups per CU
        if ((rc = check_launch("fused_src16 C3"))) return rc;
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(At, m.cp[2], m.c[2], m.h1c, m.h1v, nullptr, m.d2v, nullptr, m.d1v, true, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C2V"))) return rc;
    }
    {   // K3: C2C destination-major (rows = constraints) and C2V source-major (rows = constraints)
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(A, m.cp[3], m.c[3], m.h1v, m.h1c, nullptr, m.d2c, nullptr, m.d1c, true, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C2C"))) return rc;
        SrcLaunch16 S = {};
        S.n_jobs = 2;
        S.job[0] = src_job16(A, m.c[2], m.h1c, m.d1c_b, m.d1c);      // d1c_b = (h1c > 0) (dX + d1c): the whole gradient of h1c
        S.job[1] = src_job16(At, m.c[3], m.h1v, m.d1v_b, m.d1v);     // K4: C2C source-major (rows = variables)
        hipLaunchKernelGGL(fused_src16_kernel, dim3(2 * G), dim3(FT), 0, s, S);   // 72 VGPRs: two workgroups per CU
        if ((rc = check_launch("fused_src16 C2V + C2C"))) return rc;
    }
    {   // K5: layer 1, both convs (inputs are data: no input gradients)
        BwdLaunch1 L = {};
        L.n_jobs = 2;
        L.job[0] = bwd_job1(A, m.cp[1], m.c[1], g->x2_p, m.d1c_b);     // pre-masked, pre-summed by K3' / K4
        L.job[1] = bwd_job1(At, m.cp[0], m.c[0], g->x1_p, m.d1v_b);
        hipLaunchKernelGGL(fused_bwd1_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd1"))) return rc;
    }
    ReduceArgs R = {};
    for (int i = 0; i < MODEL_CONVS; ++i) { R.stats[i] = m.c[i].stats; R.out[i] = m.c[i].red; }
    R.nblk = G;
    R.head_part = m.have_head_part ? m.head_part : nullptr;
    R.head_out = grads + 4704;
    R.loss_out = loss;
    const int cins[MODEL_CONVS] = {1, 1, 16, 16, 16};
    float* grs[MODEL_CONVS] = {grads + 0, grads + 144, grads + 288, grads + 1392, grads + 2496};
    if (adam) {     // K6 + K7 + Adam + the next step's folded weights in one launch (single-rank step)
        TailArgs T = {};
        T.R = R;
        for (int i = 0; i < MODEL_CONVS; ++i) {
            T.p[i] = conv_params_at(m.cp[i], cins[i]);
            T.cin[i] = cins[i];
            T.grads[i] = grs[i];
            T.D[i] = m.c[i].derived;
        }
        T.zero = grads + 3600;
        T.n_zero = 1104;
        T.params = adam->params; T.grads_all = grads; T.m = adam->m; T.v = adam->v; T.state = adam->state;
        T.eps = adam->eps;
        T.n_params = adam->n;
        T.sync = g->tail_sync;
        T.err = g->tail_err_dev;
        if (g->tail_err_host && *(volatile unsigned*)g->tail_err_host != 0u)
            return fail(MLLP_EHIP, "the grid barrier of fused_tail_kernel timed out in an earlier step (the device was shared with "
                                   "other work): parameters and optimizer state since then are invalid; recreate the graph");
        hipLaunchKernelGGL(fused_tail_kernel, dim3(MODEL_CONVS * STAT_TILES + 1), dim3(RT), 0, s, T, MODEL_CONVS);
        return check_launch("fused_tail");
    }
    {   // K6
        hipLaunchKernelGGL(fused_reduce_kernel, dim3(MODEL_CONVS * STAT_TILES + 1), dim3(RT), 0, s, R, MODEL_CONVS);
        if ((rc = check_launch("fused_reduce"))) return rc;
    }
    {   // K7: the 16x16 algebra of every conv, and the zero gradient of the never-used gconv3_s2w
        const float* cps[MODEL_CONVS] = {m.cp[0], m.cp[1], m.cp[2], m.cp[3], m.cp[4]};
        const float* sts[MODEL_CONVS] = {m.c[0].red, m.c[1].red, m.c[2].red, m.c[3].red, m.c[4].red};
        const int nbs[MODEL_CONVS] = {1, 1, 1, 1, 1};
        if ((rc = launch_finalize_batch(MODEL_CONVS, cps, cins, sts, nbs, grs, grads + 3600, 1104, s))) return rc;
    }
    return MLLP_OK;
}

}  // namespace mllp
