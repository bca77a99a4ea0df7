// internal.h -- shared declarations of libmllp_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mllp_hip.h"

namespace mllp {

constexpr int FEAT = 16;
constexpr int BLOCK = 256;     // threads per workgroup (4 wavefronts of 64)
constexpr int REC_W = 40;      // floats per destination record read by the source-major backward sweep
constexpr int STAT_TILES = 7;  // 16x16 outer-product tiles reduced over nodes per conv
constexpr int STAT_FLOATS = STAT_TILES * 256;
constexpr int STAT_BLOCKS_MAX = 256;

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define MLLP_HIP_TRY(expr)                                              \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) return ::mllp::hip_fail(_e, #expr);       \
    } while (0)

// One traversal orientation: destination-major CSR over the sparsity pattern of A (dst = constraint
// rows) or of A^T (dst = variable columns).  Rows are split in three tiers by nonzero count:
//   group tier: 16 lanes per row (4 rows per wavefront), wave tier: 64 lanes per row,
//   block tier: one 256-thread workgroup per row.
// optional LDS-tiled copy of an orientation (tiled_kernels.hip); arrays are borrowed from the caller
// (mllp_graph_attach_tiled) or owned by the library (mllp_graph_build_tiled: tiled_build.hip)
struct Tiled {
    int n_tiles = 0, n_tb = 0, max_nbt = 0, max_run = 0;   // max_run: longest (tile, block) segment (library-built copies)
    bool owned = false;
    const int* tile_blk = nullptr;   // [n_tiles + 1]
    const int* blk_id = nullptr;     // [n_tb]
    const int* ptr2 = nullptr;       // [n_tb * rows_per_tile + 1] offsets of the length-sorted positions
    const int* perm = nullptr;       // [n_tb * rows_per_tile] row of each sorted position
    const int* ent = nullptr;        // [nnz][2] {col_local * 64 (byte offset of the source row in the staged block), value bits}
};

// streamed copy of an orientation (stream_layout.h): library-owned device arrays, built by mllp_graph_build_spmm_copy
struct StreamCopy {
    int n_tiles = 0, n_tb = 0;
    int64_t n_groups = 0;       // groups of 2 steps, without the padding groups at the end
    int64_t step_slots = 0;     // 128 x groups: entry slots of the stream, padding included
    int* tile_row = nullptr;    // [n_tiles + 1]
    int* tile_blk = nullptr;    // [n_tiles + 1]
    int* blk_id = nullptr;      // [n_tb]
    int* rows = nullptr;        // [n_tb * 8 * 16] int4
    int* hdr = nullptr;         // [n_tb * 8] int4
    int* ent = nullptr;         // [(n_groups + S_K0) * 64 * S_ENT]
    double build_seconds = 0.0;
};

// lane-per-row streamed copy of one orientation for the layer-1 sweeps (lane_layout.h, lane_stream.hip)
struct LaneCopy {
    int n_tiles = 0, n_tb = 0;
    int64_t n_groups = 0;       // groups of 4 steps, without the padding groups at the end
    int64_t nnz = 0;
    int* tile_row = nullptr;    // [n_tiles + 1]
    int* tile_blk = nullptr;    // [n_tiles + 1]
    int* tile_col = nullptr;    // [n_tiles][2]
    int* rows = nullptr;        // [n_tiles][L1_R]
    int* whdr = nullptr;        // [n_tb][L1_NW][2]
    unsigned* offs = nullptr;   // [(n_groups + L1_PADG) * 64 * 2]
    float* vals = nullptr;      // [(n_groups + L1_PADG) * 64 * 4]
    double build_seconds = 0.0;
};

struct Orient {
    LaneCopy lane1;     // geometry 4: layer-1 sweeps, destination-major (lane_stream.hip); takes precedence over tiled_scalar
    StreamCopy stream;  // plain SpMM on the streamed copy (stream_spmm.hip); takes precedence over `tiled`
    StreamCopy stream_attn;  // geometry 1: attention forward (stream_attn.hip); takes precedence over tiled_attn
    StreamCopy stream_bdst;  // geometry 3: destination-major attention backward; over tiled_bdst
    StreamCopy stream_bsrc;  // geometry 2: source-major attention backward (this orientation = its rows); over tiled_bsrc
    Tiled tiled;        // variant 0: geometry of the plain SpMM
    Tiled tiled_attn;   // variant 1: geometry of the attention forward sweep
    Tiled tiled_bdst;   // variant 4: geometry of the destination-major attention backward sweep
    Tiled tiled_scalar; // variant 3: geometry of the layer-1 (one channel) attention sweeps, destination-major
    Tiled tiled_bsrc;   // variant 2: geometry of the source-major attention backward sweep (this orientation = its rows)
    int n_dst = 0, n_src = 0;
    int* ptr = nullptr;    // [n_dst + 1]
    int* idx = nullptr;    // [nnz] source ids
    float* val = nullptr;  // [nnz] a_ij
    int* rows_group = nullptr;  // nullptr => identity (every row is in the group tier)
    int* rows_wave = nullptr;
    int* chunks = nullptr;      // [n_chunk] int4 {row, beg, end, slot}; slot < 0: the row's only chunk
    int* split = nullptr;       // [n_split] int4 {row, first_slot, n_chunks, 0}: rows cut into several chunks
    int n_group = 0, n_wave = 0, n_chunk = 0, n_split = 0, n_slots = 0;
    float* scratch = nullptr;   // [n_slots x SCRATCH_NS] partial states of this orientation's split rows
    int tier_wave = 0;          // rows with more nonzeros than this are skipped by the group tier
    bool short_rows = false;    // mean row length <= 16: group tier runs one nonzero slot per pass
};
// One orientation of the fused latency-regime path in RENUMBERED node ids (host_graph.h::HostFusedOrient)
struct FusedTiersDev {
    int n_block = 0, n_wave = 0, n_group = 0, n_base = 0;
};
constexpr int FUSED_NP = 8;     // partitions of the instances (host_graph.h::FUSED_PARTS): one per XCD
struct FusedOrient {
    int n_dst = 0, n_src = 0, nnz = 0;
    int* sptr = nullptr;        // [n_dst + 1]
    int* sent = nullptr;        // [nnz][2] {source id (renumbered), value bits}
    float* sax = nullptr;       // [nnz][2] {a_ij, x_src} of the bound layer-1 inputs (scalar node features are data)
    int row0[FUSED_NP + 1] = {};
    FusedTiersDev t16[FUSED_NP], t1[FUSED_NP];
    // items of every wavefront (host_graph.h::HostWaveLists), [FUSED_NP][nw][L]: 16-channel sweeps at one workgroup per
    // CU (lst16) and at two (lst16x2: fused_src16_kernel), 1-channel sweeps (lst1)
    int* lst16 = nullptr; int* lst16x2 = nullptr; int* lst1 = nullptr;
    int L16 = 0, L16x2 = 0, L1 = 0, nw16 = 0, nw16x2 = 0, nw1 = 0;
};
constexpr int SCRATCH_NS = 20;  // floats per partial-state slot of a split row

}  // namespace mllp

struct mllp_graph {
    int64_t M = 0, N = 0, nnz = 0, n_inst = 0;
    mllp::Orient A;   // dst = constraints, src = variables
    mllp::Orient At;  // dst = variables,   src = constraints
    float* inv_n = nullptr;      // [N] 1 / n_k of the owning instance
    int* inst_ptr_n = nullptr;   // [n_inst + 1] device
    int* inst_ptr_m = nullptr;   // [n_inst + 1] device
    std::vector<int64_t> h_inst_ptr_n, h_inst_ptr_m;
    std::vector<int> h_csr_ptr, h_csc_ptr;     // host copies of the row pointers (the fused path is built from them)
    int tier_wave = 0, tier_block = 0, chunk_nnz = 0;
    int max_inst_n = 0;
    int n_cu = 256;              // compute units of the device: grid size of the persistent fused kernels
    int path = 0;                // whole-model path: 0 = by size (fused below 32 M nonzeros), 1 = generic / tiled, 2 = fused
    // fused path: renumbered copies of both orientations, permutations, renumbered copies of the bound inputs
    bool fused_built = false;
    mllp::FusedOrient FA, FAt;   // FA: rows = constraints, FAt: rows = variables
    int* perm_v = nullptr;       // [N] renumbered variable id -> original
    int* perm_c = nullptr;       // [M]
    int* inv_v = nullptr;        // [N] original -> renumbered
    int* inv_c = nullptr;        // [M]
    float* inv_n_p = nullptr;    // [N] 1 / n_k in renumbered order
    unsigned long long* tail_sync = nullptr;   // [4] grid-barrier state of fused_tail_kernel {arrivals, launches, error word, -}
    unsigned* tail_err_host = nullptr;         // host-mapped word that the kernel raises when its barrier times out
    unsigned* tail_err_dev = nullptr;          // ... its device address
    float* x1_p = nullptr;       // [N] bound inputs in renumbered order
    float* x2_p = nullptr;       // [M]
    float* labels_p = nullptr;   // [N]
    const void* bound_x1 = nullptr;
    const void* bound_x2 = nullptr;
    const void* bound_labels = nullptr;
    int ws_path = -1;            // which whole-model path wrote the workspace last: 0 generic / tiled, 1 fused (backward checks it)
    const void* ws_ptr = nullptr;
    // which {workspace, parameters} hold the folded weights that fused_tail_kernel left behind for the NEXT step
    // (mllp_gnn_train_step flags bit 0 is honoured only when both match; every other whole-model call, a path switch
    // and the generic branch of train_step clear the record)
    const void* folded_ws = nullptr;
    const void* folded_params = nullptr;
    // second stream + events: the two convs of a layer (one per orientation) and the single-workgroup
    // finalize kernels run beside the main stream (fork/join by events, also under hipGraph capture)
    hipStream_t aux = nullptr;
    hipEvent_t ev[16] = {};
    std::vector<void*> allocs;   // everything to hipFree on destroy
};

namespace mllp {

// ---- per-conv parameter views into the flat state_dict-ordered buffer ---------------------------
struct ConvParams {
    const float *Wk, *bk, *Wq, *bq, *Wv, *bv, *we, *Ws, *bs;
};
inline int conv_param_count(int cin) { return 4 * FEAT * cin + 5 * FEAT; }
inline ConvParams conv_params_at(const float* base, int cin) {
    ConvParams p;
    const float* q = base;
    p.Wk = q; q += FEAT * cin;
    p.bk = q; q += FEAT;
    p.Wq = q; q += FEAT * cin;
    p.bq = q; q += FEAT;
    p.Wv = q; q += FEAT * cin;
    p.bv = q; q += FEAT;
    p.we = q; q += FEAT;
    p.Ws = q; q += FEAT * cin;
    p.bs = q;
    return p;
}

// folded weights written by the param_prep kernel (per conv; DERIVED_W floats, same layout for cin 1/16)
//   Pq [k][d]  = sum_c Wk[c][k] Wq[c][d] / 4      q'_i = Pq x_i + pq0   (logit scale 1/sqrt(16) folded in)
//   PqT[d][k]  = Pq[k][d]
//   WsT[d][o]  = Ws[o][d]          WvT[k][o] = Wv[o][k]
//   pq0[k]     = sum_c Wk[c][k] bq[c] / 4
//   Pt [d]     = sum_c we[c] Wq[c][d] / 4 ,  pt0 = <bq, we> / 4        t_i = <Pt, x_i> + pt0
//   Pb [d]     = sum_c bk[c] Wq[c][d] / 4                              (backward only)
constexpr int OFF_PQ = 0, OFF_PQT = 256, OFF_WST = 512, OFF_WVT = 768;
constexpr int OFF_PQ0 = 1024, OFF_PT = 1040, OFF_PB = 1056, OFF_PT0 = 1072;
constexpr int DERIVED_W = 1088;

// workspace of one conv (floats): what forward saves for backward + backward scratch
struct ConvWs {
    float* derived;  // [DERIVED_W]
    float* qp;       // [n_dst, cin]
    float* t;        // [n_dst]
    float* Z;        // [n_dst, cin]   normalised attention-weighted source sum
    float* aux;      // [n_dst, 4]     {u, rowmax, 1/(rowsum+1e-16), S}
    float* rec;      // [n_dst, REC_W] backward record (cin = 16) / [n_dst, 8] (cin = 1)
    float* dqp;      // [n_dst, cin]
    float* dsdt;     // [n_dst, 2]
    float* stats;    // [STAT_BLOCKS_MAX, STAT_FLOATS] per-workgroup partial statistics
    float* red;      // [STAT_FLOATS] the summed statistics (fused path)
};
int64_t conv_ws_floats(int64_t n_dst, int cin);
ConvWs conv_ws_carve(float* base, int64_t n_dst, int cin);

// ---- launchers (sweep_kernels.hip / node_kernels.hip) -------------------------------------------
int launch_spmm(const Orient& o, const float* H, float* Y, float* scratch, hipStream_t s);
int build_stream_device(const Orient& o, int64_t nnz, const std::vector<int>& tile_row, StreamCopy& sc, hipStream_t s,
                        int geom = 0);   // stream_build.hip (geom: stream_layout.h::STREAM_GEOM_*)
void stream_copy_free(StreamCopy& sc);                                                  // stream_api.cpp
int build_tiled_device(const Orient& o, int64_t nnz, int variant, Tiled& out, hipStream_t s);   // tiled_build.hip
void tiled_free(Tiled& tl);                                                             // tiled_build.hip (owned arrays only)
int launch_spmm_stream(const StreamCopy& sc, int n_dst, int n_src, const float* H, float* Y, hipStream_t s);
struct ConvWs;
int launch_fwd16_stream(const StreamCopy& sc, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                        const float* x_src, const float* x_dst, float* h_out, hipStream_t s);      // stream_attn.hip
int launch_bwddst16_stream(const StreamCopy& sc, int n_dst, int n_src, const ConvWs& w, const float* x_src, const float* g,
                           float* dx_dst, int accumulate, hipStream_t s);
struct ConvWs;
int launch_fwd1_lane(const LaneCopy& lc, int n_dst, int n_src, const float* conv_params, const ConvWs& w, const float* x_src,
                     const float* x_dst, float* h_out, hipStream_t s);                              // lane_stream.hip
int launch_bwddst1_lane(const LaneCopy& lc, int n_dst, int n_src, const ConvWs& w, const float* x_src, hipStream_t s);
int build_lane_copy(const Orient& o, int64_t nnz, const std::vector<int64_t>& seg, LaneCopy& lc, hipStream_t s);
void lane_copy_free(LaneCopy& lc);
int launch_bwdsrc16_stream(const StreamCopy& sc, int n_rows, int n_cols, const float* rec, const float* x_rows, float* dx,
                           int accumulate, hipStream_t s);
int launch_spmm_tiled(const Tiled& tl, int n_dst, int n_src, const float* H, float* Y, hipStream_t s);
int launch_spmm_tiled_bf16(const Tiled& tl, int n_dst, int n_src, const void* H_bf16, float* Y, hipStream_t s);
struct ConvWs;
int launch_fwd16_tiled(const Tiled& tl, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                       const float* x_src, const float* x_dst, float* h_out, hipStream_t s);
int launch_fwd1_tiled(const Tiled& tl, int n_dst, int n_src, const float* conv_params, const ConvWs& w,
                      const float* x_src, const float* x_dst, float* h_out, hipStream_t s);
int launch_bwddst1_tiled(const Tiled& tl, int n_dst, int n_src, const ConvWs& w, const float* x_src, hipStream_t s);
int launch_bwddst16_tiled(const Tiled& tl, int n_dst, int n_src, const ConvWs& w, const float* x_src, const float* g,
                          float* dx_dst, int accumulate, hipStream_t s);
int launch_bwdsrc16_tiled(const Tiled& tl, int n_rows, int n_cols, const float* rec, const float* x_rows, float* dx,
                          int accumulate, hipStream_t s);
int tiled_geometry(int variant, int* rows_per_tile, int* cols_per_block, int* bundle_capacity);
int tiled_max_blocks_per_tile();
constexpr int MODEL_CONVS = 5;   // convs of GNNModel that are used (gconv3_s2w is not)
int launch_param_prep_batch(int n, const float* const* conv_params, const int* cin, float* const* derived, hipStream_t s);
int launch_finalize_batch(int n, const float* const* conv_params, const int* cin, const float* const* stats,
                          const int* n_stat_blocks, float* const* grads, float* zero, int n_zero, hipStream_t s);
int launch_param_prep(const float* conv_params, int cin, float* derived, hipStream_t s);
int launch_node_qp(const float* x_dst, int64_t n_dst, const float* derived, float* qp, float* t, hipStream_t s);
int launch_attn_fwd(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                    const float* x_dst, float* h_out, float* scratch, hipStream_t s);
int launch_bwd_pre(int64_t n_dst, int cin, const float* conv_params, const ConvWs& w, const float* x_dst,
                   const float* h_out, float* dh, hipStream_t s);
int launch_attn_bwd_dst(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                        const float* g, float* dx_dst, int accumulate, float* scratch, hipStream_t s);
int launch_attn_bwd_src(const Orient& o_src_major, const ConvWs& w, const float* x_src, float* dx_src,
                        int accumulate, float* scratch, hipStream_t s);
int launch_param_stats(int cin, int64_t n_dst, const ConvWs& w, const float* x_dst, const float* g, hipStream_t s);
int launch_finalize_conv(int cin, const float* conv_params, const float* stats, int n_stat_blocks, float* grads,
                         hipStream_t s);
int stat_blocks_for(int64_t n_dst);

// head: mode 0 = logits only, 1 = backward from given dlogits, 2 = fused BCE forward+backward
int launch_head(int mode, int64_t n, const float* h3v, const float* fc_w, const float* fc_b, const float* inv_n,
                const float* labels, float inv_batch, const float* dlogits_in, float* logits, float* dh3v,
                float* partials, hipStream_t s);
int launch_head_finalize(const float* partials, int n_blocks, float* grad_fc /*17*/, float* loss /*nullable*/,
                         hipStream_t s);
int head_blocks_for(int64_t n);
int launch_adam(float* p, const float* g, float* m, float* v, float* state, float eps, float gscale, int64_t n,
                hipStream_t s);
int launch_fill_zero(float* p, int64_t n, hipStream_t s);
int launch_topm_metrics(const mllp_graph* g, const float* logits, const float* labels, void* scratch, float* out,
                        hipStream_t s);

// ---- fused latency-regime path of the whole model (fused_kernels.hip) ---------------------------------
struct FusedModel {
    const float* cp[MODEL_CONVS];   // parameters of gconv1_w2s, gconv1_s2w, gconv2_w2s, gconv2_s2w, gconv3_w2s
    ConvWs c[MODEL_CONVS];
    const float *x1, *x2, *fcw, *fcb, *labels;
    float inv_batch;
    float *h1v, *h1c, *h2v, *h2c, *h3v;
    float *d3v, *d2v, *d2c, *d1v, *d1c, *d1v_b, *d1c_b;
    float *logits, *head_part;
    bool have_head_part;         // fused_backward sums the fc partials (from fused_forward mode 2 or fused_head_backward)
};
int fused_graph_build(mllp_graph* g, const int* h_csr_ptr, const int* h_csc_ptr);   // allocates: creation / set_path only
int fused_grid(const mllp_graph* g);
int fused_bind(mllp_graph* g, const float* x1, const float* x2, const float* labels, hipStream_t s);
// head_mode 1: logits (h3v kept), 2: logits + BCE + masked dL/dh3v in d3v + fc partials
int fused_forward(mllp_graph* g, const FusedModel& m, int head_mode, hipStream_t s, bool skip_prep = false);
// d3v = dlogits (original variable order) x fc weight in renumbered order, fc gradient partials
int fused_head_backward(const mllp_graph* g, const FusedModel& m, const float* dlogits, hipStream_t s);
// premasked: d3v and the fc partials come from fused_forward(head_mode 2); else d3v = dL/dh3v (unmasked) and the
// caller has produced the fc gradient itself
// adam != nullptr: the single-rank tail (reduce + gradients + Adam + the next step's folded weights in one launch)
struct FusedAdam {
    float *params, *m, *v, *state;
    float eps;
    int n;
};
int fused_backward(const mllp_graph* g, const FusedModel& m, bool premasked, float* grads, float* loss, hipStream_t s,
                   const FusedAdam* adam = nullptr);

}  // namespace mllp
