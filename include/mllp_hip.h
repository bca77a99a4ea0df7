/*
 * mllp_hip.h -- C ABI of libmllp_hip.so: the MI355X (gfx950) implementation of mllp's learned-LP
 * hot path (bipartite TransformerConv message passing over the LP constraint matrix, forward and
 * backward, fused BCE head, flat Adam).
 *
 * The reference (HAHHHD/mllp) has no FFI/plugin interface: it is pure Python on PyTorch +
 * PyTorch-Geometric.  The boundary a maintainer binds is therefore this
 * header, loaded with ctypes from the Python modules that keep the reference's names
 * (INTEGRATION.md shows the stub).  Each entry point cites the reference lines it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative MLLP_E* code on failure;
 *     mllp_last_error() returns a thread-local message for the last failure on this thread.
 *   - "device pointer" arguments are caller-owned HIP device memory (e.g. torch tensors); the
 *     library never frees or reallocates them and allocates nothing after mllp_graph_create_*.
 *   - every launch is asynchronous on the hipStream_t passed as `stream` (void*; NULL = default
 *     stream); no entry point synchronises except mllp_graph_create_* / mllp_graph_export.
 *   - all launch functions are hipGraph-capturable (no malloc/free/sync inside).
 *   - a mllp_graph_t is immutable after creation except for an internal scratch buffer used by
 *     rows that are split over several workgroups: calls on ONE graph must be stream-ordered.
 *   - feature width is fixed at 16 (reference linear_program_methods.py:206-211), fp32 everywhere.
 *   - there is NO CPU fallback: without a HIP device every launch function fails with MLLP_EHIP.
 */
#ifndef MLLP_HIP_H
#define MLLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLLP_ABI_VERSION 3 /* 2: streamed SpMM copy, device-built tiled copies, mllp_gnn_train_step, mllp_graph_invalidate_inputs; 3: streamed copies of the attention sweeps (mllp_graph_*_stream_copy) */
#define MLLP_FEAT 16
#define MLLP_NUM_PARAMS 4721 /* GNNModel.state_dict(), SURVEY.md appendix A.2 */

#define MLLP_OK 0
#define MLLP_EINVAL (-1) /* bad argument (null pointer, shape mismatch, unsorted indices, ...) */
#define MLLP_EHIP (-2)   /* a HIP runtime call failed (message has hipGetErrorString) */
#define MLLP_ENOMEM (-3)
#define MLLP_ERANGE (-4) /* sizes exceed int32 indexing */

typedef struct mllp_graph mllp_graph_t;

const char* mllp_last_error(void);
int mllp_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Graph: a block-diagonal batch of LP constraint matrices, resident in HBM in both orientations.
 * Replaces the per-step Python graph build `build_graph_from_weights_sets`
 * (linear_program_methods.py:89-103) and the batching rule `BipartiteData.__inc__`
 * (linear_program_methods.py:60-72): variable ids of instance k are offset by sum_{j<k} n_j,
 * constraint ids by sum_{j<k} m_j.  Built ONCE; the reference rebuilds it every step
 * (linear_program_experiment.py:124).
 * ---------------------------------------------------------------------------------------------- */

/* From host CSR pieces exactly as the reference loader returns them
 * (linear_program_data.py:75-77: scipy CSR indptr/indices/data per instance).
 *   indptr  : concatenation over instances of (m_k + 1) LOCAL row offsets (each block starts at 0)
 *   indices : concatenation of LOCAL column ids (row-major, sorted within a row, no duplicates)
 *   values  : float64 entries a_ij (cast to fp32 as linear_program_methods.py:100 does)
 *   tier_wave, tier_block : rows with more than tier_wave nonzeros are processed by one 64-lane
 *     wavefront, more than tier_block by 256-thread workgroups (one per chunk of at most
 *     4 * tier_block nonzeros; longer rows are split and merged); 0 = choose automatically.    */
int mllp_graph_create_host(int64_t n_inst, const int64_t* inst_m, const int64_t* inst_n,
                           const int64_t* indptr, const int32_t* indices, const double* values,
                           int32_t tier_wave, int32_t tier_block, mllp_graph_t** out);

/* From device arrays that already hold the batch in global ids, both orientations (CSR of A:
 * rows = constraints; CSR of A^T: rows = variables, row ids ascending within a column).  The
 * arrays are copied.  inst_ptr_m / inst_ptr_n are HOST arrays of n_inst + 1 offsets.           */
/* CSR(A) -> CSR(A^T) on the device (counting + scatter + per-column ordering by row id: the one stable transposition,
 * whatever order the scatter's atomics produced; mllp_amd/csrc/transpose.hip): the second orientation that
 * mllp_graph_create_device wants, for batches that were generated or loaded on the GPU.  All arrays are caller-owned
 * device memory: d_ptr [n_rows + 1], d_idx / d_val [nnz] (column ids ascending inside a row), outputs d_t_ptr
 * [n_cols + 1], d_t_idx / d_t_val [nnz].  Allocates scratch and synchronises `stream` (not a launch function).
 * The input is checked on the device first: d_ptr must ascend from 0 to nnz and the column ids of a row must be
 * strictly ascending and smaller than n_cols (no duplicate entries, which the per-column ordering could not keep
 * apart); anything else returns MLLP_EINVAL and writes nothing.                                                */
int mllp_csr_transpose_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* d_ptr, const int32_t* d_idx,
                              const float* d_val, int32_t* d_t_ptr, int32_t* d_t_idx, float* d_t_val, void* stream);

int mllp_graph_create_device(int64_t n_inst, const int64_t* inst_ptr_m, const int64_t* inst_ptr_n,
                             int64_t nnz,
                             const int32_t* d_csr_ptr, const int32_t* d_csr_idx, const float* d_csr_val,
                             const int32_t* d_csc_ptr, const int32_t* d_csc_idx, const float* d_csc_val,
                             int32_t tier_wave, int32_t tier_block, void* stream, mllp_graph_t** out);

int mllp_graph_destroy(mllp_graph_t* g);

/* dims[0..11] = M (constraints), N (variables), nnz, n_inst,
 *               group-tier rows / wave-tier rows / chunk work items for A (dst = constraints),
 *               the same three for A^T (dst = variables),
 *               rows split over several chunks for A, for A^T                                  */
int mllp_graph_dims(const mllp_graph_t* g, int64_t dims[12]);

/* Copy one device array of the graph back to host memory (synchronises; tests/debugging).
 * which: 0 csr_ptr(int32,M+1) 1 csr_idx(int32,nnz) 2 csr_val(f32,nnz)
 *        3 csc_ptr(int32,N+1) 4 csc_idx(int32,nnz) 5 csc_val(f32,nnz) 6 inv_n(f32,N)            */
int mllp_graph_export(const mllp_graph_t* g, int which, void* host_dst, int64_t capacity_bytes);

/* Which kernels the whole-model entry points (mllp_gnn_forward / _backward / _loss_step) use:
 *   0  by size (default): the fused latency-regime kernels below 32 M nonzeros (real Netlib: one sweep launch
 *      per conv forward, two backward, 16 rows per wavefront), the generic / LDS-tiled sweeps above;
 *   1  always the generic / LDS-tiled sweeps;   2  always the fused kernels.
 * Both paths compute the same quantities (tests compare them with each other and with the oracle).        */
int mllp_graph_set_path(mllp_graph_t* g, int path);

/* INPUT CONTRACT of the whole-model entry points on the fused path: the graph keeps renumbered copies of d_x1, d_x2
 * and d_labels (and the layer-1 source features gathered beside the nonzeros), keyed on the POINTER VALUES -- the
 * model's inputs are data (linear_program_methods.py:90-91), constant for the life of a batch.  A caller that changes
 * their contents in place, or frees a buffer and gets the same address back for other data, must call
 * mllp_graph_invalidate_inputs before the next mllp_gnn_* call on this graph (the generic / tiled path reads the
 * pointers on every call and needs nothing).  Calls made while the stream is captured into a hipGraph re-make the
 * copies inside the capture and leave the cache empty.  mllp_gnn_backward must follow an mllp_gnn_forward on the same
 * workspace with the same path (MLLP_EINVAL otherwise: the two paths lay the workspace out differently).          */
int mllp_graph_invalidate_inputs(mllp_graph_t* g);

/* ------------------------------------------------------------------------------------------------
 * Plain CSR SpMM (the roofline kernel named in BASELINE.json's metric):
 *   transpose == 0:  Y[M,16] = A   * H[N,16]       transpose == 1:  Y[N,16] = A^T * H[M,16]
 * This is the unweighted skeleton of the message passing in linear_program_methods.py:241-247
 * (gather source rows by edge, scale by a_ij, add into the destination row).
 * ---------------------------------------------------------------------------------------------- */
int mllp_spmm_csr_f32(const mllp_graph_t* g, int transpose, const float* d_H, float* d_Y, void* stream);
/* Opt-in bf16 feature image (SURVEY.md 8b `mllp_spmm_csr_f32/_bf16`, BASELINE.json configs[2] "bf16"): the same
 * product with H given as bf16 [n_src,16] (32-byte rows: half the H traffic and half the LDS image), fp32
 * values, fp32 accumulation, fp32 Y.  NOT the parity path: the result is the fp32 product of A with H rounded
 * to bf16 (relative error of an element of H up to 2^-8; the accumulation itself is exact-fp32 as above).
 * Runs on the LDS-tiled copy only (mllp_graph_attach_tiled, variant 0); MLLP_EINVAL without it.            */
int mllp_spmm_csr_bf16(const mllp_graph_t* g, int transpose, const void* d_H_bf16, float* d_Y, void* stream);

/* Streamed copy of one orientation for mllp_spmm_csr_f32 on large batches (rows of ~100+ nonzeros; layout and
 * rationale: mllp_amd/csrc/stream_layout.h).  The nonzeros are re-blocked ONCE into row tiles of at most 960 rows (never
 * across two LP instances) x 750-column blocks and stored in the order the kernel's wavefronts consume them, so that they stream HBM -> registers while LDS
 * holds two images of H (double-buffered by LDS-DMA).  Replaces the edge list the reference rebuilds every step
 * (linear_program_methods.py:89-103).  The copy is LIBRARY-owned device memory (these three are not launch functions:
 * they allocate, free and synchronise), ~8.6 bytes per nonzero; mllp_spmm_csr_f32 uses it when present.
 *   where: 0 = built on the device (counting + placement kernels), 1 = built by the host reference builder
 *          (same bytes; tests compare the two).
 *   mllp_graph_spmm_copy_info: info[0..7] = row tiles, (tile, block) pairs, 1 KB entry groups, entry slots (padding
 *          included; compare with nnz), bytes of the copy, microseconds the build took, rows per tile, columns per block
 *          | wavefronts per tile << 16.
 *   mllp_graph_export_spmm_copy (tests): which = 0 tile_blk (int32, tiles + 1), 1 blk_id (int32), 2 row records
 *          (int32 x 4 per (tile-block, wavefront, quad)), 3 entry stream (int32 x 4 per (group, lane)), 4 tile_row
 *          (int32, tiles + 1), 5 headers (int32 x 4 per (tile-block, wavefront)).                                 */
int mllp_graph_build_spmm_copy(mllp_graph_t* g, int transpose, int where, void* stream);
int mllp_graph_drop_spmm_copy(mllp_graph_t* g, int transpose);
int mllp_graph_spmm_copy_info(const mllp_graph_t* g, int transpose, int64_t info[8]);
int mllp_graph_export_spmm_copy(const mllp_graph_t* g, int transpose, int which, void* host_dst, int64_t capacity_bytes);

/* Streamed copies for the ATTENTION sweeps of the throughput regime (round 4; mllp_amd/csrc/stream_attn.hip): the same
 * layout and builders in other geometries, one library-owned copy per (orientation, geometry).  The four functions above
 * are these with geom = 0.
 *   geom 0  plain SpMM: 960-row tiles x 750-column blocks, four rows per quad (mllp_spmm_csr_f32)
 *   geom 1  attention forward sweep of a 16-channel TransformerConv whose DESTINATIONS are the rows of this orientation
 *           (transpose = 0: constraints, 1: variables): 480-row tiles x 720-column blocks, two rows per quad, 64-byte
 *           staged items.  When present, mllp_tconv_fwd and the mllp_gnn_* calls of the generic / tiled path use it in
 *           place of the LDS-tiled variant 1.
 *   geom 2  source-major backward sweep (rows of this orientation = the conv's SOURCE nodes): 312-column blocks of the
 *           160-byte destination records; replaces LDS-tiled variant 2.
 *   geom 3  destination-major backward sweep (rows = the conv's destinations): 432-column blocks; replaces variant 4.
 *   geom 4  the LAYER-1 sweeps (one input channel, destination-major forward and backward; mllp_amd/csrc/lane_stream.hip,
 *           layout lane_layout.h): 512-row tiles, ONE LANE PER ROW, 20 000-column blocks of 4-byte items (a whole instance
 *           of the synthetic batch); replaces LDS-tiled variant 3.  where = 1: host reference builder, same bytes.  Export: which = 0
 *           tile_blk, 1 tile_col (int32 x 2 per tile), 2 rows (int32 x 512 per tile), 3 column offsets (uint32 x 2 per
 *           (group, lane)), 4 tile_row, 5 headers (int32 x 2 per (tile-block, wavefront)), 6 values (float x 4 per
 *           (group, lane)); info[6] = 512 | 1 << 16 | 4 << 24.
 *   mllp_graph_stream_copy_info: info[0..5] as mllp_graph_spmm_copy_info; info[6] = row slots per tile | rows per quad
 *          << 16 | bytes per staged item << 24; info[7] = columns per block | wavefronts << 16 | padding groups << 24. */
int mllp_graph_build_stream_copy(mllp_graph_t* g, int transpose, int geom, int where, void* stream);
int mllp_graph_drop_stream_copy(mllp_graph_t* g, int transpose, int geom);
int mllp_graph_stream_copy_info(const mllp_graph_t* g, int transpose, int geom, int64_t info[8]);
int mllp_graph_export_stream_copy(const mllp_graph_t* g, int transpose, int geom, int which, void* host_dst,
                                  int64_t capacity_bytes);

/* Optional LDS-tiled copy of one orientation for large batches (rows of ~100+ nonzeros): the nonzeros
 * re-blocked into row tiles x column blocks so that source rows are read from LDS instead of L2.
 *   mllp_tiled_geometry: rows per tile, source nodes per column block, and the number of entries of one
 *     (tile, block) segment that fit the kernel's LDS window (longer segments take several windows).
 *   d_tile_blk [n_tiles+1]  first (tile, block) index of each row tile (a tile lists the contiguous
 *                           range of column blocks it touches)
 *   d_blk_id   [n_tb]       global column-block id of each (tile, block)
 *   d_ptr2     [n_tb * rows_per_tile + 1]  entry offsets: inside a (tile, block) the rows are ordered by their
 *                           number of entries in that block, descending (position k = k-th longest)
 *   d_perm     [n_tb * rows_per_tile]      row (inside its tile) of every sorted position
 *   d_ent      [nnz][2]     {byte offset of the column's staged item inside its block (64 * local column for
 *                           variants 0, 1, 4), fp32 value bits}, ordered by (tile, block, position); inside a row
 *                           round-robin over (local column mod 4) (LDS bank spreading)
 * The arrays are BORROWED: the caller keeps them alive while attached (n_tiles = 0 detaches).
 * variant 0 is the geometry of the plain SpMM (mllp_spmm_csr_f32 uses it when attached); variant 1 the attention
 * forward sweep with 16 channels (per-row query and softmax state in LDS, hence smaller column blocks); variant 2
 * the source-major attention backward sweep (attach it to the orientation whose ROWS are the conv's source nodes;
 * the staged items are the 160-byte backward records, d_ent[.][0] = 160 * local column); variant 3 the layer-1
 * (one channel) forward and destination-major backward sweeps (staged items are scalars, d_ent[.][0] = 4 * local
 * column); variant 4 the destination-major attention backward sweep with one lane per row: d_perm is the
 * identity (rows keep their order inside a tile, d_ptr2 their offsets) and the entries of every 64-row chunk of a
 * (tile, block) are ordered by step -- entry j of the rows that have one, in row order -- instead of row by row
 * (mllp_amd/graph.py::build_tiled_arrays builds every variant).  mllp_tconv_* / mllp_gnn_* use whichever
 * copies are attached and fall back to the generic sweeps (HIP as well) otherwise.                    */
int mllp_tiled_geometry(int variant, int32_t* rows_per_tile, int32_t* cols_per_block, int32_t* bundle_capacity);
int mllp_graph_attach_tiled(mllp_graph_t* g, int transpose, int variant, int64_t n_tiles, int64_t n_tb,
                            int32_t max_blocks_per_tile /* largest tile_blk[t+1]-tile_blk[t]; at most 255 */,
                            const int32_t* d_tile_blk, const int32_t* d_blk_id, const int32_t* d_ptr2,
                            const int32_t* d_perm, const int32_t* d_ent);
/* The same copies built by the library on the device from the graph's own CSR (counting passes + one placement pass,
 * mllp_amd/csrc/tiled_build.hip; not a launch function: allocates and synchronises `stream`).  The arrays are
 * library-owned: a later build / attach / detach of the same (orientation, variant) and mllp_graph_destroy free them.
 *   mllp_graph_tiled_info: info[0..4] = row tiles, (tile, block) pairs, most blocks in one tile, 1 if library-owned,
 *   longest (tile, block) segment in entries (0 for caller-built copies).
 *   mllp_graph_export_tiled (tests): device-to-device copy of array `which` = 0 tile_blk, 1 blk_id, 2 ptr2, 3 perm,
 *   4 ent ((nnz + 1) x 2 int32); `count` int32 elements must equal the array's length.                       */
int mllp_graph_build_tiled(mllp_graph_t* g, int transpose, int variant, void* stream);
int mllp_graph_tiled_info(const mllp_graph_t* g, int transpose, int variant, int64_t info[5]);
int mllp_graph_export_tiled(const mllp_graph_t* g, int transpose, int variant, int which, int32_t* d_dst, int64_t count,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * One torch_geometric.nn.TransformerConv((cin, cin), 16, edge_dim=1) followed by ReLU, as called at
 * linear_program_methods.py:241-247 (construction :206-211).  dst_is_var = 1 for the *_w2s convs
 * (source = constraints, destination = variables), 0 for *_s2w.
 *   d_conv_params : the conv's 9 tensors, flat, in state_dict order (lin_key.weight, lin_key.bias,
 *                   lin_query.weight, lin_query.bias, lin_value.weight, lin_value.bias,
 *                   lin_edge.weight, lin_skip.weight, lin_skip.bias): 144 floats (cin=1) or 1104.
 *   d_x_src [N_src, cin], d_x_dst [N_dst, cin], d_h_out [N_dst, 16] = relu(conv(...))
 *   d_ws : mllp_tconv_workspace_floats() floats; holds what backward needs (kept by the caller
 *          between forward and backward).
 * Backward: d_dh [N_dst,16] = dL/dh_out (overwritten with the ReLU-masked gradient);
 *   d_dx_dst [N_dst,cin], d_dx_src [N_src,cin] may be NULL (then not computed; cin = 1 inputs are
 *   data, linear_program_methods.py:90-91); accumulate bit 0 / bit 1 = add into d_dx_dst / d_dx_src
 *   instead of overwriting; d_param_grads: same layout as d_conv_params (overwritten).
 * ---------------------------------------------------------------------------------------------- */
int mllp_tconv_workspace_floats(const mllp_graph_t* g, int dst_is_var, int cin, int64_t* n_floats);
int mllp_tconv_fwd(const mllp_graph_t* g, int dst_is_var, int cin, const float* d_conv_params,
                   const float* d_x_src, const float* d_x_dst, float* d_h_out, float* d_ws, void* stream);
int mllp_tconv_bwd(const mllp_graph_t* g, int dst_is_var, int cin, const float* d_conv_params,
                   const float* d_x_src, const float* d_x_dst, const float* d_h_out, float* d_ws,
                   float* d_dh, float* d_dx_dst, float* d_dx_src, int accumulate,
                   float* d_param_grads, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole model: GNNModel.forward (linear_program_methods.py:238-251) and its backward
 * (autograd in the reference: linear_program_experiment.py:141).
 *   d_params : MLLP_NUM_PARAMS floats, GNNModel.state_dict() order
 *   d_x1 [N] = objective coefficients (x1, methods.py:90), d_x2 [M] = right-hand sides (x2, :91)
 *   d_ws     : mllp_gnn_workspace_bytes() bytes, kept between forward and backward
 *   d_logits [N] : per-variable logits (methods.py:250-251)
 * mllp_gnn_backward: d_dlogits [N] -> d_grads [MLLP_NUM_PARAMS] (overwritten; the never-called
 *   gconv3_s2w block, methods.py:248, is written as zeros).
 * mllp_gnn_loss_step: forward + BCEWithLogitsLoss + backward in one call, loss =
 *   inv_batch * sum_k mean_i BCE(logit_i, label_i) over the instances k of this graph
 *   (linear_program_experiment.py:41,139-141 with batch size 1 and inv_batch = 1).
 *   d_labels [N] float 0/1; d_loss: 1 float.
 * ---------------------------------------------------------------------------------------------- */
int mllp_gnn_workspace_bytes(const mllp_graph_t* g, int64_t* bytes);
int mllp_gnn_forward(const mllp_graph_t* g, const float* d_params, const float* d_x1, const float* d_x2,
                     void* d_ws, float* d_logits, void* stream);
int mllp_gnn_backward(const mllp_graph_t* g, const float* d_params, const float* d_x1, const float* d_x2,
                      void* d_ws, const float* d_dlogits, float* d_grads, void* stream);
int mllp_gnn_loss_step(const mllp_graph_t* g, const float* d_params, const float* d_x1, const float* d_x2,
                       const float* d_labels, float inv_batch, void* d_ws, float* d_logits,
                       float* d_loss, float* d_grads, void* stream);

/* ------------------------------------------------------------------------------------------------
 * torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8), no weight decay
 * (linear_program_experiment.py:119,143-144) on flat buffers.
 *   d_state: 4 floats on the device {step (as float, incremented by the kernel), lr, beta1, beta2};
 *            eps is passed by value.  grad_scale multiplies the gradient first (data-parallel
 *            averaging).  Parameters whose gradient is exactly 0 with zero moments do not move,
 *            which reproduces torch skipping `grad is None` parameters (gconv3_s2w).
 * ---------------------------------------------------------------------------------------------- */
int mllp_adam_step(float* d_params, const float* d_grads, float* d_exp_avg, float* d_exp_avg_sq,
                   float* d_state, float eps, float grad_scale, int64_t n, void* stream);

/* One whole training step of a single rank: mllp_gnn_loss_step followed by mllp_adam_step (grad_scale 1) on the
 * MLLP_NUM_PARAMS parameters -- the body of the reference's loop, linear_program_experiment.py:139-144 -- with the same
 * results bit for bit.  On the latency-regime path (batches below 32 M nonzeros) the end of the step runs as ONE launch:
 * reduction of the statistics, the gradients of the five convs, Adam, and the folded weights of the NEXT step, which
 * are left in the workspace.  flags bit 0: "the folded weights in d_ws are current" -- set it when the previous call on
 * this d_ws was mllp_gnn_train_step with these d_params and nothing else has written d_params since; the forward then
 * skips its weight-folding launch.  With several ranks the gradient all-reduce sits between the two halves, so a
 * data-parallel caller keeps calling mllp_gnn_loss_step and mllp_adam_step.                                      */
int mllp_gnn_train_step(const mllp_graph_t* g, float* d_params, const float* d_x1, const float* d_x2,
                        const float* d_labels, float inv_batch, void* d_ws, float* d_logits, float* d_loss,
                        float* d_grads, float* d_exp_avg, float* d_exp_avg_sq, float* d_state, float eps, int flags,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Prediction + metrics (linear_program_experiment.py:146-151): per instance k, mark the m_k largest
 * logits, correct_k = |pred & basis|, f1_k = 2TP / (2TP + FP + FN).
 *   d_out [n_inst, 2] = {correct_k, f1_k}.   d_scratch: mllp_metrics_scratch_bytes() bytes.
 * ---------------------------------------------------------------------------------------------- */
int mllp_metrics_scratch_bytes(const mllp_graph_t* g, int64_t* bytes);
int mllp_topm_metrics(const mllp_graph_t* g, const float* d_logits, const float* d_labels,
                      void* d_scratch, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MPS -> the tensors the reference's loader reads (host only, no GPU needed).
 * Replaces the preprocessing that produced /root/reference/dataset/netlib_mps{,_norm}/<name>_{constrs.npz,
 * coefs.npy,rhs.npy} from /root/reference/netlib_mps/<name>.mps (input format: afiro.mps:1-83; consumer:
 * linear_program_data.py:58-80).  The reference ships the tensors but not the script; the rules (recovered from
 * the data, reproducing all 97 instances) are stated in mllp_amd/csrc/mps_reader.cpp and oracle/mps_norm.py.
 *   normalize == 0: the raw stage (A, c, b; one extra column per RANGES entry)
 *   normalize != 0: slack column per L / G row, rows scaled to unit 2-norm with the right-hand side capped at 5,
 *                   objective scaled to unit 2-norm.
 * dims[0..5] = m, n (all columns), nnz, structural columns, range columns, slack columns.
 * mllp_lp_export: any pointer may be null; indptr [m+1] int64, indices [nnz] int32 (ascending in a row),
 *   values [nnz], coefs [n], rhs [m] float64, slack_rows [slack columns] int32 (the row of each slack, in order:
 *   basis = [v ; c[slack_rows]] assembles the label vector from a solver's variable / constraint statuses).   */
typedef struct mllp_lp mllp_lp_t;
int mllp_mps_read(const char* path, int normalize, mllp_lp_t** out);
int mllp_lp_dims(const mllp_lp_t* lp, int64_t dims[6]);
int mllp_lp_export(const mllp_lp_t* lp, int64_t* indptr, int32_t* indices, double* values, double* coefs,
                   double* rhs, int32_t* slack_rows);
int mllp_lp_free(mllp_lp_t* lp);

/* ------------------------------------------------------------------------------------------------
 * `AngleModel` of the `angleNet` method (reference linear_program_methods.py:187-200, loop
 * linear_program_experiment.py:81-114): three TransformerConv layers (2 -> F, F -> F, and the same F -> F layer
 * again) + Linear(F, 1) on the COMPLETE directed graph over the N = n + 1 nodes of one instance, edge attribute =
 * cosine similarity (build_graph_from_Q_sets, :119-130).  Dense attention with a scalar edge bias:
 *   d_cos    [N, N]  cosine matrix, row = target node, column = source node; SYMMETRIC (the kernels read either triangle);
 *                    the diagonal is ignored: no self loops
 *   d_x      [N, 2]  node features {coef, |Q row|}
 *   d_params flat fp32 in PyG state_dict order: gconv1, gconv2, gconv3 (each lin_key {W [F,C], b [F]}, lin_query,
 *            lin_value, lin_edge {W [F,1]}, lin_skip {W, b}; C = 2 for gconv1, F otherwise), fc {W [1,F], b [1]}
 *            (mllp_angle_num_params floats; gconv3 is never called by the reference's forward: its gradient is 0)
 *   d_ws     mllp_angle_workspace_floats floats, kept between forward and backward
 *   forward : d_logits [N - 1] = fc(h)[:-1]            backward: d_dlogits [N - 1] -> d_grads (layout of d_params)
 * Hand-written kernels on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): flash-attention-style forward / backward
 * sweeps that keep no N x N matrix in HBM (the backward recomputes the weights from the saved row max / row sum) and one
 * strided GEMM for the projections and their gradients; no BLAS library is linked or loaded.  feat_dim must be 16, 32,
 * 64, 128 or 256 (MLLP_EINVAL otherwise); N <= 46340; the workspace is O(N * feat_dim * ranges), ranges <= 32.       */
int mllp_angle_num_params(int feat_dim, int64_t* out);
int mllp_angle_workspace_floats(int64_t n_nodes, int feat_dim, int64_t* out);
int mllp_angle_forward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                       float* d_ws, float* d_logits, void* stream);
int mllp_angle_backward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                        float* d_ws, const float* d_dlogits, float* d_grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLLP_HIP_H */
