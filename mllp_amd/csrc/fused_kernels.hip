// fused_kernels.hip -- the latency-regime path of the whole model (batches below 32 M nonzeros: real Netlib).
//
// Why a second set of kernels: on the Netlib batch (1.07 M nonzeros, 366 k nodes, median row length 2-4) a training
// step of the generic path is ~45 launches whose sweeps are bound by dependent memory round trips at low occupancy
// (profiles/r01_netlib_kernel_stats.md: 91 us for a 1 M-nonzero attention sweep, 68 % of wave time in s_waitcnt), and
// per conv it writes and re-reads q', t, dq', ds/dt and a separate statistics pass.  Here a conv is ONE sweep launch
// forward (weight folding products, q' = Pq x, the attention sweep, the output GEMMs, ReLU -- and for the last conv
// fc + BCEWithLogits + dL/dh) and TWO backward (destination-major: ReLU mask, gv = Wv^T g, record, sweep, input
// gradient, parameter statistics on the MFMA; source-major: the transposed sweep), and the two convs of a layer /
// the independent sweeps of the backward chain share launches.  reference: linear_program_methods.py:238-251,
// linear_program_experiment.py:139-141; formulas SURVEY.md appendix A.3 / A.4 (oracle/spmm_form.py).
//
// Mapping: persistent workgroups of 1024 threads, one per CU.  Rows of an orientation are ordered by length
// (host_graph.h::HostItems); a WAVEFRONT takes an item = 16 rows of (nearly) equal length, one QUAD of lanes per row
// (lane `part` owns channels 4 part .. 4 part + 3; a gather of a 64-byte source row is one 16-byte load per lane of
// the quad), so 16 rows and up to 64 gathers are in flight per wavefront -- four times the rows per wave of the
// generic 16-lanes-per-row tier.  Rows longer than 32 nonzeros take a whole wavefront (16 quads stride the row,
// states merged by DPP / shuffles), rows longer than 512 the whole workgroup (merged through LDS).  Scalar (layer-1)
// convs use one LANE per row.  The 16x16 per-node GEMMs run in the quad layout: the 16 inputs of a row are spread
// over the quad, fetched with quad_perm DPP, weights come from LDS (rows padded to 80 bytes so that the four parts
// read four different bank groups).  Nothing here uses atomics: partial sums have a fixed owner and a fixed order, so
// two runs give identical bits.
#include <algorithm>

#include "device_utils.h"
#include "internal.h"

namespace mllp {

typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr int FT = 1024;            // threads per workgroup
constexpr int FW = FT / 64;         // wavefronts per workgroup
constexpr int WSTR = 20;            // floats per padded 16-float weight row in LDS
constexpr int MAXJOBS = 2;

struct ItemsDev {
    const int* __restrict__ ptr;
    const int* __restrict__ idx;
    const float* __restrict__ val;
    const int* __restrict__ rows;   // rows by length, descending: [block tier | wave tier | quad tier]
    int n_dst, n_block, n_wave, n_quad;
};

static ItemsDev items_dev(const Orient& o) {
    ItemsDev d;
    d.ptr = o.ptr; d.idx = o.idx; d.val = o.val; d.rows = o.item_rows;
    d.n_dst = o.n_dst; d.n_block = o.n_iblock; d.n_wave = o.n_iwave; d.n_quad = o.n_iquad;
    return d;
}

__device__ __forceinline__ int wave_items16(const ItemsDev& s) { return s.n_wave + ((s.n_quad + 15) >> 4); }
__device__ __forceinline__ int wave_items64(const ItemsDev& s) { return s.n_wave + ((s.n_quad + 63) >> 6); }

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
    int first;    // index of this quad's (lane's) first nonzero
    int stride;   // distance between its nonzeros
    int end;      // end of the row
    int mode;     // 0 = row per quad (lane), 1 = row per wavefront, 2 = row per workgroup
};
// G lanes per row in the quad tier: 4 (16-channel sweeps) or 1 (scalar sweeps)
template <int G>
__device__ __forceinline__ RowSlot item_slot(const ItemsDev& s, int item, int lane) {
    constexpr int RPW = 64 / G;                    // rows per wavefront in the quad tier
    const int unit = lane / G;                     // quad (or lane) inside the wavefront
    RowSlot r;
    if (item < s.n_wave) {
        r.row = s.rows[s.n_block + item];
        r.mode = 1;
        r.first = s.ptr[r.row] + unit;
        r.stride = RPW;
        r.end = s.ptr[r.row + 1];
    } else {
        const int k = s.n_block + s.n_wave + (item - s.n_wave) * RPW + unit;
        r.mode = 0;
        r.stride = 1;
        if (k < s.n_dst) {
            r.row = s.rows[k];
            r.first = s.ptr[r.row];
            r.end = s.ptr[r.row + 1];
        } else {
            r.row = -1; r.first = 0; r.end = 0;
        }
    }
    return r;
}
template <int G>
__device__ __forceinline__ RowSlot block_slot(const ItemsDev& s, int k, int tid) {
    RowSlot r;
    r.row = s.rows[k];
    r.mode = 2;
    r.first = s.ptr[r.row] + tid / G;
    r.stride = FT / G;
    r.end = s.ptr[r.row + 1];
    return r;
}

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
    const float* __restrict__ inv_n;
    const float* __restrict__ labels;
    float inv_batch;
    float* __restrict__ logits;        // [n_dst]
    float* __restrict__ g_out;         // [n_dst, 16]  dL/dh of the head, already ReLU-masked
    float* __restrict__ head_part;     // [grid, 18]   {dW_fc[16], db_fc, loss} per workgroup
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

// the attention sweep of one quad over its nonzeros (first, first + stride, ... < end): online segment softmax
__device__ __forceinline__ void fwd16_edges(const ItemsDev& s, const float* __restrict__ X, const RowSlot& r,
                                            const float4& qp, float t, int part, SoftState& st) {
    const int n_mine = r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
    for (int k0 = 0; __any(k0 < n_mine); k0 += 4) {
        // the four lanes of the quad fetch four consecutive entries of the row and share them by DPP
        const int km = k0 + part;
        const bool okm = km < n_mine;
        const int em = r.first + km * r.stride;
        const int colm = okm ? s.idx[em] : 0;
        const float am = okm ? s.val[em] : 0.0f;
        const int c0 = quad_bcast_i<0>(colm), c1 = quad_bcast_i<1>(colm), c2 = quad_bcast_i<2>(colm), c3 = quad_bcast_i<3>(colm);
        const float a0 = quad_bcast<0>(am), a1 = quad_bcast<1>(am), a2 = quad_bcast<2>(am), a3 = quad_bcast<3>(am);
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine, ok2 = k0 + 2 < n_mine, ok3 = k0 + 3 < n_mine;
        float4 x0 = f4zero(), x1 = f4zero(), x2 = f4zero(), x3 = f4zero();
        if (ok0) x0 = ld4(X + (size_t)c0 * 16 + 4 * part);
        if (ok1) x1 = ld4(X + (size_t)c1 * 16 + 4 * part);
        if (ok2) x2 = ld4(X + (size_t)c2 * 16 + 4 * part);
        if (ok3) x3 = ld4(X + (size_t)c3 * 16 + 4 * part);
        const float d0 = ok0 ? fmaf(a0, t, quad_sum(dot4(qp, x0))) : NEG_BIG;
        const float d1 = ok1 ? fmaf(a1, t, quad_sum(dot4(qp, x1))) : NEG_BIG;
        const float d2 = ok2 ? fmaf(a2, t, quad_sum(dot4(qp, x2))) : NEG_BIG;
        const float d3 = ok3 ? fmaf(a3, t, quad_sum(dot4(qp, x3))) : NEG_BIG;
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
        st.u = fmaf(p0, a0, fmaf(p1, a1, fmaf(p2, a2, fmaf(p3, a3, st.u))));
        fma4(p0, x0, st.Z);
        fma4(p1, x1, st.Z);
        fma4(p2, x2, st.Z);
        fma4(p3, x3, st.Z);
    }
}

// all 16 quads of the wavefront hold partial states of ONE row: every lane ends with the row's state
__device__ __forceinline__ void soft_merge_wave(SoftState& st) {
    const float M = quads_max<64>(st.m);
    const float f = exp_acc(st.m - M);      // partial without nonzeros: exp(-huge) == 0
    st.m = M;
    st.L = quads_sum<64>(st.L * f);
    st.u = quads_sum<64>(st.u * f);
    st.Z.x = quads_sum<64>(st.Z.x * f);
    st.Z.y = quads_sum<64>(st.Z.y * f);
    st.Z.z = quads_sum<64>(st.Z.z * f);
    st.Z.w = quads_sum<64>(st.Z.w * f);
}

struct HeadAcc {
    float4 w;      // dW_fc of this lane's four channels
    float b, l;    // db_fc, loss (part-0 lanes)
};

// per-row prologue + sweep + epilogue of one job
__device__ __forceinline__ void fwd16_row(const FwdJob16& J, const FwdW16& W, const RowSlot& r, int part,
                                          int lane, float* merge_lds, float* tiles, HeadAcc& ha) {
    const bool have = r.row >= 0;
    float4 qp;
    float t;
    {   // q' = Pq x + pq0 on the MFMA (rows of the wavefront through tile 0, kept for the epilogue; result through tile 1)
        const float4 xd = have ? ld4(J.x_dst + (size_t)r.row * 16 + 4 * part) : f4zero();
        float ax_[4];
        tile_put(tiles, xd, lane);
        tile_rows(tiles, lane, ax_);
        tile_put_result(tiles + TILE, mat_apply(ax_, matB_lds(W.BPq, lane), splat4(W.pq0[lane & 15])), lane);
        qp = tile_get(tiles + TILE, lane);
        t = quad_sum(dot4(lds4(W.Pt + 4 * part), xd)) + W.pt0;
    }
    SoftState st;
    st.Z = f4zero(); st.m = NEG_BIG; st.L = 0.0f; st.u = 0.0f;
    fwd16_edges(J.s, J.x_src, r, qp, t, part, st);
    bool writer = have;
    if (r.mode >= 1) {
        soft_merge_wave(st);
        writer = (lane >> 2) == 0;
    }
    if (r.mode == 2) {       // merge the 16 wavefronts of the workgroup through LDS (fixed order)
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
        writer = threadIdx.x < 4;
        __syncthreads();     // the slots are free for the next block-tier row
    }
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
        tile_put_result(tiles + 2 * TILE,
                        mat_apply(ax_, matB_lds(W.BWs, lane), mat_apply(az_, matB_lds(W.BWv, lane), splat4(0.0f))), lane);
        o = tile_get(tiles + 2 * TILE, lane);
    }
    o = f4add(o, lds4(W.bs + 4 * part));
    fma4(S, lds4(W.bv + 4 * part), o);
    fma4(un, lds4(W.we + 4 * part), o);
    const float4 hv = make_float4(fmaxf(o.x, 0.0f), fmaxf(o.y, 0.0f), fmaxf(o.z, 0.0f), fmaxf(o.w, 0.0f));
    if (writer) {
        *reinterpret_cast<float4*>(J.Z + (size_t)r.row * 16 + 4 * part) = zn;
        if (part == 0) reinterpret_cast<float4*>(J.aux)[r.row] = make_float4(un, st.L > 0.0f ? st.m : 0.0f, rinv, S);
        if (J.head != 2) *reinterpret_cast<float4*>(J.h + (size_t)r.row * 16 + 4 * part) = hv;
    }
    if (J.head) {        // fc (16 -> 1) on the conv's output row, reference linear_program_methods.py:250
        const float4 fw = lds4(W.fcw + 4 * part);
        const float z = quad_sum(dot4(hv, fw)) + W.fcb;
        if (writer && part == 0) J.logits[r.row] = z;
        if (J.head == 2 && writer) {      // BCEWithLogitsLoss, mean per instance / batch: linear_program_experiment.py:41,139-140
            const float y = J.labels[r.row];
            const float wn = J.inv_n[r.row] * J.inv_batch;
            const float e = expf(-fabsf(z));
            const float sig = z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
            const float dz = wn * (sig - y);
            const float4 g = make_float4(hv.x > 0.0f ? dz * fw.x : 0.0f, hv.y > 0.0f ? dz * fw.y : 0.0f,
                                         hv.z > 0.0f ? dz * fw.z : 0.0f, hv.w > 0.0f ? dz * fw.w : 0.0f);
            *reinterpret_cast<float4*>(J.g_out + (size_t)r.row * 16 + 4 * part) = g;
            fma4(dz, hv, ha.w);
            if (part == 0) {
                ha.b += dz;
                ha.l += wn * (fmaxf(z, 0.0f) - z * y + log1pf(e));
            }
        }
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
    const int gw = blockIdx.x * FW + wave, nw = gridDim.x * FW;
    int base = 0;
    bool any_head = false;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob16& J = A.job[j];
        any_head |= J.head == 2;
        // block tier: the whole workgroup walks one long row at a time (longest rows first)
        for (int k = blockIdx.x; k < J.s.n_block; k += gridDim.x)
            fwd16_row(J, Ws_[j], block_slot<4>(J.s, k, tid), part, lane, merge_lds, tiles, ha);
        // wave loop: the items of all jobs form one sequence dealt round-robin over the wavefronts
        const int n_items = wave_items16(J.s);
        int it = (gw - base % nw + nw) % nw;
        for (; it < n_items; it += nw)
            fwd16_row(J, Ws_[j], item_slot<4>(J.s, it, lane), part, lane, merge_lds, tiles, ha);
        base += n_items;
    }
    // fc partial sums of this workgroup (one job at most has a head)
    if (!any_head) return;
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
        for (int w = 0; w < FW; ++w) v += head_lds[w * 18 + tid];
        for (int j = 0; j < A.n_jobs; ++j)
            if (A.job[j].head == 2) A.job[j].head_part[(size_t)blockIdx.x * 18 + tid] = v;
    }
}

// ====================================================================================================
// forward, 1 source channel (layer 1): one lane per row
// ====================================================================================================
struct FwdJob1 {
    ItemsDev s;
    const float* __restrict__ x_src;   // [n_src]
    const float* __restrict__ x_dst;   // [n_dst]
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
struct FwdW1 {
    float wv[16], ws[16], bv[16], we[16], bs[16];
    float pq, pq0, pt, pt0;
};
struct Soft1 {
    float m, L, u, Z;
};

__device__ __forceinline__ void fwd1_edges(const ItemsDev& s, const float* __restrict__ X, const RowSlot& r, float qp,
                                           float t, Soft1& st) {
    const int n_mine = r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
    for (int k0 = 0; __any(k0 < n_mine); k0 += 2) {
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine;
        const int e0 = r.first + k0 * r.stride, e1 = e0 + r.stride;
        const int c0 = ok0 ? s.idx[e0] : 0, c1 = ok1 ? s.idx[e1] : 0;
        const float a0 = ok0 ? s.val[e0] : 0.0f, a1 = ok1 ? s.val[e1] : 0.0f;
        const float x0 = ok0 ? X[c0] : 0.0f, x1 = ok1 ? X[c1] : 0.0f;
        const float d0 = ok0 ? fmaf(qp, x0, a0 * t) : NEG_BIG;
        const float d1 = ok1 ? fmaf(qp, x1, a1 * t) : NEG_BIG;
        const float mi = fmaxf(d0, d1);
        if (__any(mi > st.m)) {
            const float mn = fmaxf(st.m, mi);
            const float sc = exp_acc(st.m - mn);
            st.L *= sc; st.u *= sc; st.Z *= sc;
            st.m = mn;
        }
        const float p0 = ok0 ? exp_acc(d0 - st.m) : 0.0f, p1 = ok1 ? exp_acc(d1 - st.m) : 0.0f;
        st.L += p0 + p1;
        st.u = fmaf(p0, a0, fmaf(p1, a1, st.u));
        st.Z = fmaf(p0, x0, fmaf(p1, x1, st.Z));
    }
}
__device__ __forceinline__ void soft1_merge_wave(Soft1& st) {
    const float M = group_max<64>(st.m);
    const float f = exp_acc(st.m - M);
    st.m = M;
    st.L = wave_sum(st.L * f);
    st.u = wave_sum(st.u * f);
    st.Z = wave_sum(st.Z * f);
}

__device__ __forceinline__ void fwd1_row(const FwdJob1& J, const FwdW1& W, const RowSlot& r, int lane, float* merge_lds) {
    const bool have = r.row >= 0;
    const float x = have ? J.x_dst[r.row] : 0.0f;
    const float qp = fmaf(W.pq, x, W.pq0), t = fmaf(W.pt, x, W.pt0);
    Soft1 st;
    st.m = NEG_BIG; st.L = 0.0f; st.u = 0.0f; st.Z = 0.0f;
    fwd1_edges(J.s, J.x_src, r, qp, t, st);
    bool writer = have;
    if (r.mode >= 1) {
        soft1_merge_wave(st);
        writer = lane == 0;
    }
    if (r.mode == 2) {
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
        writer = threadIdx.x == 0;
        __syncthreads();
    }
    if (!writer) return;
    const float rinv = 1.0f / (st.L + 1e-16f);
    const float S = st.L * rinv, un = st.u * rinv, zn = st.Z * rinv;
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
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob1& J = A.job[j];
        for (int k = blockIdx.x; k < J.s.n_block; k += gridDim.x)
            fwd1_row(J, Ws_[j], block_slot<1>(J.s, k, tid), lane, merge_lds);
    }
    const int gw = blockIdx.x * FW + wave, nw = gridDim.x * FW;
    int base = 0;
    for (int j = 0; j < A.n_jobs; ++j) {
        const FwdJob1& J = A.job[j];
        const int n_items = wave_items64(J.s);
        int it = (gw - base % nw + nw) % nw;
        for (; it < n_items; it += nw) fwd1_row(J, Ws_[j], item_slot<1>(J.s, it, lane), lane, merge_lds);
        base += n_items;
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
    float* __restrict__ stats;         // [grid, STAT_FLOATS]
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

__device__ __forceinline__ void bwd16_edges(const ItemsDev& s, const float* __restrict__ X, const RowSlot& r,
                                            const float4& qp, const float4& gv, float t, float m, float rinv, float ge,
                                            float cc, int part, BwdState& st) {
    const int n_mine = r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
    for (int k0 = 0; __any(k0 < n_mine); k0 += 4) {
        const int km = k0 + part;
        const bool okm = km < n_mine;
        const int em = r.first + km * r.stride;
        const int colm = okm ? s.idx[em] : 0;
        const float am = okm ? s.val[em] : 0.0f;
        const int c0 = quad_bcast_i<0>(colm), c1 = quad_bcast_i<1>(colm), c2 = quad_bcast_i<2>(colm), c3 = quad_bcast_i<3>(colm);
        const float a0 = quad_bcast<0>(am), a1 = quad_bcast<1>(am), a2 = quad_bcast<2>(am), a3 = quad_bcast<3>(am);
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine, ok2 = k0 + 2 < n_mine, ok3 = k0 + 3 < n_mine;
        float4 x0 = f4zero(), x1 = f4zero(), x2 = f4zero(), x3 = f4zero();
        if (ok0) x0 = ld4(X + (size_t)c0 * 16 + 4 * part);
        if (ok1) x1 = ld4(X + (size_t)c1 * 16 + 4 * part);
        if (ok2) x2 = ld4(X + (size_t)c2 * 16 + 4 * part);
        if (ok3) x3 = ld4(X + (size_t)c3 * 16 + 4 * part);
#define MLLP_BWD_SLOT(OK, A_, X_)                                                          \
    {                                                                                      \
        const float l_ = fmaf(A_, t, quad_sum(dot4(qp, X_)));                              \
        const float al_ = OK ? exp_acc(l_ - m) * rinv : 0.0f;                              \
        const float dl_ = al_ * (quad_sum(dot4(gv, X_)) + fmaf(A_, ge, cc));               \
        st.ds += dl_;                                                                      \
        st.dt = fmaf(dl_, A_, st.dt);                                                      \
        fma4(dl_, X_, st.dq);                                                              \
    }
        MLLP_BWD_SLOT(ok0, a0, x0)
        MLLP_BWD_SLOT(ok1, a1, x1)
        MLLP_BWD_SLOT(ok2, a2, x2)
        MLLP_BWD_SLOT(ok3, a3, x3)
#undef MLLP_BWD_SLOT
    }
}

// tiles of a wavefront in the destination-major backward sweep
constexpr int TB_G = 0, TB_X = 1, TB_Z = 2, TB_DQ = 3, TB_SC = 4, TB_E = 5, TB_N = 6;

__device__ __forceinline__ void bwd16_row(const BwdJob16& J, const BwdW16& W, const RowSlot& r, int part,
                                          int lane, float* merge_lds, float* tiles, f32x4m (&acc)[STAT_TILES]) {
    const bool have = r.row >= 0;
    float4 qp, gv;
    float t, m, rinv, ge, cc;
    bool writer = have;
    if (r.mode >= 1) writer = (lane >> 2) == 0;
    if (r.mode == 2) writer = threadIdx.x < 4;
    {   // everything of the row that the sweep does not need stays in the tiles G, X, Z, E until the statistics
        const size_t ro = have ? (size_t)r.row * 16 + 4 * part : 0;
        float4 g = have ? ld4(J.dh_a + ro) : f4zero();
        if (J.dh_b && have) g = f4add(g, ld4(J.dh_b + ro));
        if (J.h && have) {
            const float4 hv = ld4(J.h + ro);
            g = make_float4(hv.x > 0.0f ? g.x : 0.0f, hv.y > 0.0f ? g.y : 0.0f, hv.z > 0.0f ? g.z : 0.0f, hv.w > 0.0f ? g.w : 0.0f);
        }
        const float4 xd = have ? ld4(J.x_dst + ro) : f4zero();
        const float4 Zn = have ? ld4(J.Z + ro) : f4zero();
        const float4 ax = have ? reinterpret_cast<const float4*>(J.aux)[r.row] : f4zero();   // {u, rowmax, rinv, S}
        // gv = Wv^T g and q' = Pq x + pq0 on the MFMA; the results come back through the (still unused) tiles DQ and SC
        float ag_[4], ax_[4];
        tile_put(tiles + TB_G * TILE, g, lane);
        tile_put(tiles + TB_X * TILE, xd, lane);
        tile_put(tiles + TB_Z * TILE, Zn, lane);
        tile_put(tiles + TB_E * TILE, (have && part == 0) ? make_float4(1.0f, ax.w, ax.x, 0.0f) : f4zero(), lane);
        tile_rows(tiles + TB_G * TILE, lane, ag_);
        tile_rows(tiles + TB_X * TILE, lane, ax_);
        tile_put_result(tiles + TB_DQ * TILE, mat_apply(ag_, matB_lds(W.BWvT, lane), splat4(0.0f)), lane);
        tile_put_result(tiles + TB_SC * TILE, mat_apply(ax_, matB_lds(W.BPq, lane), splat4(W.pq0[lane & 15])), lane);
        gv = tile_get(tiles + TB_DQ * TILE, lane);
        qp = tile_get(tiles + TB_SC * TILE, lane);
        t = quad_sum(dot4(lds4(W.Pt + 4 * part), xd)) + W.pt0;
        ge = quad_sum(dot4(g, lds4(W.we + 4 * part)));
        const float gb = quad_sum(dot4(g, lds4(W.bv + 4 * part)));
        const float Dn = quad_sum(dot4(gv, Zn)) + gb * ax.w + ge * ax.x;
        cc = gb - Dn;
        m = ax.y; rinv = ax.z;
        if (writer && J.rec) {
            float* rr = J.rec + (size_t)r.row * REC_W;
            *reinterpret_cast<float4*>(rr + 4 * part) = qp;
            *reinterpret_cast<float4*>(rr + 16 + 4 * part) = gv;
            if (part == 0) *reinterpret_cast<float4*>(rr + 32) = make_float4(t, m, rinv, ge);
            if (part == 1) *reinterpret_cast<float4*>(rr + 36) = make_float4(cc, 0.0f, 0.0f, 0.0f);
        }
    }
    BwdState st;
    st.dq = f4zero(); st.ds = 0.0f; st.dt = 0.0f;
    bwd16_edges(J.s, J.x_src, r, qp, gv, t, m, rinv, ge, cc, part, st);
    if (r.mode >= 1) {
        st.ds = quads_sum<64>(st.ds);
        st.dt = quads_sum<64>(st.dt);
        st.dq.x = quads_sum<64>(st.dq.x); st.dq.y = quads_sum<64>(st.dq.y);
        st.dq.z = quads_sum<64>(st.dq.z); st.dq.w = quads_sum<64>(st.dq.w);
    }
    if (r.mode == 2) {
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
    if (J.dx_dst) {          // dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt
        float ag_[4], adq_[4];
        tile_rows(tiles + TB_G * TILE, lane, ag_);
        tile_rows(tiles + TB_DQ * TILE, lane, adq_);
        tile_put_result(tiles + TB_SC * TILE,
                        mat_apply(adq_, matB_lds(W.BPqT, lane), mat_apply(ag_, matB_lds(W.BWsT, lane), splat4(0.0f))), lane);
        float4 v = tile_get(tiles + TB_SC * TILE, lane);
        fma4(st.ds, lds4(W.Pb + 4 * part), v);
        fma4(st.dt, lds4(W.Pt + 4 * part), v);
        if (writer) *reinterpret_cast<float4*>(J.dx_dst + (size_t)r.row * 16 + 4 * part) = v;
    }
    // statistics (node_kernels.hip::param_stats16_kernel): operands with m / n = channel, k = row.  A row shared by
    // the wavefront (workgroup) is counted once: only quad 0 keeps it, the other 15 rows of the tiles become zeros
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
    tile_cols(tiles + TB_G * TILE, lane, cg);
    tile_cols(tiles + TB_DQ * TILE, lane, cdq);
    tile_cols(tiles + TB_SC * TILE, lane, csc);
    tile_cols(tiles + TB_X * TILE, lane, cx);
    tile_cols(tiles + TB_Z * TILE, lane, cz);
    tile_cols(tiles + TB_E * TILE, lane, ce);
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
    const int gw = blockIdx.x * FW + wave, nw = gridDim.x * FW;
    int base = 0;
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob16& J = A.job[j];
        f32x4m acc[STAT_TILES];
#pragma unroll
        for (int i = 0; i < STAT_TILES; ++i) acc[i] = splat4(0.0f);
        for (int k = blockIdx.x; k < J.s.n_block; k += gridDim.x)
            bwd16_row(J, Ws_[j], block_slot<4>(J.s, k, tid), part, lane, merge_lds, tiles, acc);
        const int n_items = wave_items16(J.s);
        int it = (gw - base % nw + nw) % nw;
        for (; it < n_items; it += nw)
            bwd16_row(J, Ws_[j], item_slot<4>(J.s, it, lane), part, lane, merge_lds, tiles, acc);
        base += n_items;
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
};

__device__ __forceinline__ void src16_row(const SrcJob16& J, const RowSlot& r, int part, int lane, float* merge_lds) {
    const bool have = r.row >= 0;
    const float4 xj = have ? ld4(J.x + (size_t)r.row * 16 + 4 * part) : f4zero();
    float4 acc = f4zero();
    const int n_mine = r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
    for (int k0 = 0; __any(k0 < n_mine); k0 += 2) {
        const int km = k0 + (part & 1);
        const bool okm = km < n_mine;
        const int em = r.first + km * r.stride;
        const int colm = okm ? J.s.idx[em] : 0;
        const float am = okm ? J.s.val[em] : 0.0f;
        const int c0 = quad_bcast_i<0>(colm), c1 = quad_bcast_i<1>(colm);
        const float a0 = quad_bcast<0>(am), a1 = quad_bcast<1>(am);
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine;
        float4 q0 = f4zero(), g0 = f4zero(), s0 = f4zero(), q1 = f4zero(), g1 = f4zero(), s1 = f4zero();
        float cc0 = 0.0f, cc1 = 0.0f;
        if (ok0) {
            const float* rr = J.rec + (size_t)c0 * REC_W;
            q0 = ld4(rr + 4 * part); g0 = ld4(rr + 16 + 4 * part); s0 = ld4(rr + 32); cc0 = rr[36];
        }
        if (ok1) {
            const float* rr = J.rec + (size_t)c1 * REC_W;
            q1 = ld4(rr + 4 * part); g1 = ld4(rr + 16 + 4 * part); s1 = ld4(rr + 32); cc1 = rr[36];
        }
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
    bool writer = have;
    if (r.mode >= 1) {
        acc.x = quads_sum<64>(acc.x); acc.y = quads_sum<64>(acc.y);
        acc.z = quads_sum<64>(acc.z); acc.w = quads_sum<64>(acc.w);
        writer = (lane >> 2) == 0;
    }
    if (r.mode == 2) {
        const int wave = threadIdx.x >> 6;
        if (lane < 4) *reinterpret_cast<float4*>(merge_lds + wave * 20 + 4 * part) = acc;
        __syncthreads();
        float4 v = f4zero();
        for (int w = 0; w < FW; ++w) v = f4add(v, lds4(merge_lds + w * 20 + 4 * part));
        acc = v;
        writer = threadIdx.x < 4;
        __syncthreads();
    }
    if (writer) *reinterpret_cast<float4*>(J.dx + (size_t)r.row * 16 + 4 * part) = acc;
}

// ====================================================================================================
// backward, destination-major, 1 channel (layer 1: inputs are data, no input gradients): one lane per row
//   statistics in the layout of node_kernels.hip::param_stats1_kernel
// ====================================================================================================
struct BwdJob1 {
    ItemsDev s;
    const float* __restrict__ x_src;   // [n_src]
    const float* __restrict__ x_dst;   // [n_dst]
    const float* __restrict__ D;
    ConvParams p;
    const float* __restrict__ h;       // [n_dst, 16]
    const float* __restrict__ dh_a;    // [n_dst, 16]
    const float* __restrict__ dh_b;    // [n_dst, 16] or nullptr
    const float* __restrict__ Z;       // [n_dst]
    const float* __restrict__ aux;     // [n_dst, 4]
    float* __restrict__ stats;         // [grid, STAT_FLOATS]
};
struct BwdW1 {
    float wv[16], bv[16], we[16];
    float pq, pq0, pt, pt0;
};
// statistics of a scalar conv: T[o][n] = sum_rows g_o R_n with R = {x, Z, 1, S, u} on the MFMA (the wavefront's 64
// rows through two LDS tiles), and six scalar sums on the VALU
constexpr int G1S = 17, R1S = 9;                       // row strides of the g tile (16 channels) and the R tile (5 columns)
constexpr int TILE1 = 64 * G1S + 64 * R1S;             // floats per wavefront

__device__ __forceinline__ void bwd1_row(const BwdJob1& J, const BwdW1& W, const RowSlot& r, int lane, float* merge_lds,
                                         float* tiles, f32x4m& accT, float (&accS)[6]) {
    const bool have = r.row >= 0;
    bool writer = have;
    if (r.mode >= 1) writer = lane == 0;
    if (r.mode == 2) writer = threadIdx.x == 0;
    float gv = 0.0f, ge = 0.0f, gb = 0.0f;
    {   // g = (dh_a + dh_b) * (h > 0); the row of g goes to the tile (zeros unless this lane owns the row)
        const size_t ro = (size_t)(have ? r.row : 0) * 16;
        const float4* pa = reinterpret_cast<const float4*>(J.dh_a + ro);
        const float4* pb = reinterpret_cast<const float4*>((J.dh_b ? J.dh_b : J.dh_a) + ro);
        const float4* ph = reinterpret_cast<const float4*>(J.h + ro);
        float* gt = tiles + lane * G1S;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a = have ? pa[q] : f4zero();
            if (J.dh_b && have) a = f4add(a, pb[q]);
            const float4 hv = have ? ph[q] : f4zero();
            const float g0 = hv.x > 0.0f ? a.x : 0.0f, g1 = hv.y > 0.0f ? a.y : 0.0f;
            const float g2 = hv.z > 0.0f ? a.z : 0.0f, g3 = hv.w > 0.0f ? a.w : 0.0f;
            gv = fmaf(g0, W.wv[4 * q], fmaf(g1, W.wv[4 * q + 1], fmaf(g2, W.wv[4 * q + 2], fmaf(g3, W.wv[4 * q + 3], gv))));
            ge = fmaf(g0, W.we[4 * q], fmaf(g1, W.we[4 * q + 1], fmaf(g2, W.we[4 * q + 2], fmaf(g3, W.we[4 * q + 3], ge))));
            gb = fmaf(g0, W.bv[4 * q], fmaf(g1, W.bv[4 * q + 1], fmaf(g2, W.bv[4 * q + 2], fmaf(g3, W.bv[4 * q + 3], gb))));
            gt[4 * q] = writer ? g0 : 0.0f;
            gt[4 * q + 1] = writer ? g1 : 0.0f;
            gt[4 * q + 2] = writer ? g2 : 0.0f;
            gt[4 * q + 3] = writer ? g3 : 0.0f;
        }
    }
    const float x = have ? J.x_dst[r.row] : 0.0f;
    const float Zn = have ? J.Z[r.row] : 0.0f;
    const float4 ax = have ? reinterpret_cast<const float4*>(J.aux)[r.row] : f4zero();   // {u, rowmax, rinv, S}
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
    const int n_mine = r.first < r.end ? (r.end - r.first + r.stride - 1) / r.stride : 0;
    for (int k0 = 0; __any(k0 < n_mine); k0 += 2) {
        const bool ok0 = k0 < n_mine, ok1 = k0 + 1 < n_mine;
        const int e0 = r.first + k0 * r.stride, e1 = e0 + r.stride;
        const int c0 = ok0 ? J.s.idx[e0] : 0, c1 = ok1 ? J.s.idx[e1] : 0;
        const float a0 = ok0 ? J.s.val[e0] : 0.0f, a1 = ok1 ? J.s.val[e1] : 0.0f;
        const float x0 = ok0 ? J.x_src[c0] : 0.0f, x1 = ok1 ? J.x_src[c1] : 0.0f;
        const float l0 = fmaf(qp, x0, a0 * t), l1 = fmaf(qp, x1, a1 * t);
        const float al0 = ok0 ? exp_acc(l0 - ax.y) * ax.z : 0.0f, al1 = ok1 ? exp_acc(l1 - ax.y) * ax.z : 0.0f;
        const float dl0 = al0 * fmaf(gv, x0, fmaf(a0, ge, cc)), dl1 = al1 * fmaf(gv, x1, fmaf(a1, ge, cc));
        ds += dl0 + dl1;
        dt = fmaf(dl0, a0, fmaf(dl1, a1, dt));
        dq = fmaf(dl0, x0, fmaf(dl1, x1, dq));
    }
    if (r.mode >= 1) {
        ds = wave_sum(ds); dt = wave_sum(dt); dq = wave_sum(dq);
    }
    if (r.mode == 2) {
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
}

// ---- one launch of the backward chain: up to two destination-major jobs (16- or 1-channel) and up to two
//      source-major jobs; all of them only read what earlier launches wrote --------------------------------------
struct BwdLaunch1 {
    BwdJob1 job[MAXJOBS];
    int n_jobs;
};
struct SrcLaunch16 {
    SrcJob16 job[MAXJOBS];
    int n_jobs;
};

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
    const int gw = blockIdx.x * FW + wave, nw = gridDim.x * FW;
    int base = 0;
    for (int j = 0; j < A.n_jobs; ++j) {
        const BwdJob1& J = A.job[j];
        f32x4m accT = splat4(0.0f);
        float accS[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = blockIdx.x; k < J.s.n_block; k += gridDim.x)
            bwd1_row(J, Ws_[j], block_slot<1>(J.s, k, tid), lane, merge_lds, tiles, accT, accS);
        const int n_items = wave_items64(J.s);
        int it = (gw - base % nw + nw) % nw;
        for (; it < n_items; it += nw) bwd1_row(J, Ws_[j], item_slot<1>(J.s, it, lane), lane, merge_lds, tiles, accT, accS);
        base += n_items;
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
    }
}

__global__ __launch_bounds__(FT) void fused_src16_kernel(SrcLaunch16 A) {
    __shared__ float merge_lds[FW * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, part = lane & 3;
    for (int j = 0; j < A.n_jobs; ++j) {
        const SrcJob16& J = A.job[j];
        for (int k = blockIdx.x; k < J.s.n_block; k += gridDim.x)
            src16_row(J, block_slot<4>(J.s, k, tid), part, lane, merge_lds);
    }
    const int gw = blockIdx.x * FW + wave, nw = gridDim.x * FW;
    int base = 0;
    for (int j = 0; j < A.n_jobs; ++j) {
        const SrcJob16& J = A.job[j];
        const int n_items = wave_items16(J.s);
        int it = (gw - base % nw + nw) % nw;
        for (; it < n_items; it += nw) src16_row(J, item_slot<4>(J.s, it, lane), part, lane, merge_lds);
        base += n_items;
    }
}

// ====================================================================================================
// reduction of the per-workgroup statistics partials: out[c][i] = sum_b stats_c[b][i], fixed order.
// grid = (convs x 7 tiles); 1024 threads = 4 slices of the partials x 256 columns
// ====================================================================================================
struct ReduceArgs {
    const float* stats[MODEL_CONVS];
    float* out[MODEL_CONVS];
    int nblk;
    // the fc partials ride along: workgroup (n_conv * 7) sums head_part[nblk][18] into head_out[18]
    const float* head_part;
    float* head_out;     // [17] gradient of fc, then the loss
    float* loss_out;     // nullable
};

__global__ __launch_bounds__(FT) void fused_reduce_kernel(ReduceArgs A, int n_conv) {
    __shared__ float sh[4][256];
    const int tid = threadIdx.x, col = tid & 255, slice = tid >> 8;
    if ((int)blockIdx.x == n_conv * STAT_TILES) {     // fc partials
        float v = 0.0f;
        if (slice == 0 && col < 18 && A.head_part)
            for (int b = 0; b < A.nblk; ++b) v += A.head_part[(size_t)b * 18 + col];
        if (slice == 0 && col < 18 && A.head_part) {
            if (col < 17) A.head_out[col] = v;
            else if (A.loss_out) A.loss_out[0] = v;
        }
        return;
    }
    const int c = blockIdx.x / STAT_TILES, tile = blockIdx.x % STAT_TILES;
    const float* src = A.stats[c] + tile * 256 + col;
    float v = 0.0f;
    int b = slice;
    for (; b + 28 < A.nblk; b += 32) {      // eight partials in flight, fixed order
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = src[(size_t)(b + 4 * q) * STAT_FLOATS];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += t[q];
    }
    for (; b < A.nblk; b += 4) v += src[(size_t)b * STAT_FLOATS];
    sh[slice][col] = v;
    __syncthreads();
    if (tid < 256) A.out[c][tile * 256 + tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}

// ====================================================================================================
// host side: the whole model on the fused path
// ====================================================================================================
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, what);
}

int fused_grid(const mllp_graph* g) { return std::min(std::max(g->n_cu, 1), STAT_BLOCKS_MAX); }

static FwdJob16 fwd_job16(const Orient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                          float* h) {
    FwdJob16 J = {};
    J.s = items_dev(o);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 16);
    J.h = h; J.Z = w.Z; J.aux = w.aux;
    J.head = 0;
    return J;
}
static FwdJob1 fwd_job1(const Orient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                        float* h) {
    FwdJob1 J = {};
    J.s = items_dev(o);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 1);
    J.h = h; J.Z = w.Z; J.aux = w.aux;
    return J;
}

int fused_forward(const mllp_graph* g, const FusedModel& m, int head_mode, hipStream_t s) {
    const int G = fused_grid(g);
    int rc;
    {   // folded weights of all five convs
        const float* cps[MODEL_CONVS] = {m.cp[0], m.cp[1], m.cp[2], m.cp[3], m.cp[4]};
        const int cins[MODEL_CONVS] = {1, 1, 16, 16, 16};
        float* ders[MODEL_CONVS] = {m.c[0].derived, m.c[1].derived, m.c[2].derived, m.c[3].derived, m.c[4].derived};
        if ((rc = launch_param_prep_batch(MODEL_CONVS, cps, cins, ders, s))) return rc;
    }
    {   // linear_program_methods.py:241-242  layer 1, both directions
        FwdLaunch1 L = {};
        L.n_jobs = 2;
        L.job[0] = fwd_job1(g->At, m.cp[0], m.c[0], m.x2, m.x1, m.h1v);     // w2s: dst = variables
        L.job[1] = fwd_job1(g->A, m.cp[1], m.c[1], m.x1, m.x2, m.h1c);      // s2w: dst = constraints
        hipLaunchKernelGGL(fused_fwd1_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd1"))) return rc;
    }
    {   // :244-245  layer 2 (simultaneous update)
        FwdLaunch16 L = {};
        L.n_jobs = 2;
        L.job[0] = fwd_job16(g->At, m.cp[2], m.c[2], m.h1c, m.h1v, m.h2v);
        L.job[1] = fwd_job16(g->A, m.cp[3], m.c[3], m.h1v, m.h1c, m.h2c);
        hipLaunchKernelGGL(fused_fwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd16 layer 2"))) return rc;
    }
    {   // :247 layer 3 (variables only) + :250 fc (+ loss)
        FwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = fwd_job16(g->At, m.cp[4], m.c[4], m.h2c, m.h2v, m.h3v);
        FwdJob16& J = L.job[0];
        J.head = head_mode;
        J.fcw = m.fcw; J.fcb = m.fcb; J.inv_n = g->inv_n; J.labels = m.labels; J.inv_batch = m.inv_batch;
        J.logits = m.logits; J.g_out = m.d3v; J.head_part = m.head_part;
        hipLaunchKernelGGL(fused_fwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_fwd16 layer 3"))) return rc;
    }
    return MLLP_OK;
}

static BwdJob16 bwd_job16(const Orient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                          const float* h, const float* dh_a, const float* dh_b, float* dx_dst, bool need_rec) {
    BwdJob16 J = {};
    J.s = items_dev(o);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 16);
    J.h = h; J.dh_a = dh_a; J.dh_b = dh_b; J.Z = w.Z; J.aux = w.aux;
    J.rec = need_rec ? w.rec : nullptr;
    J.dx_dst = dx_dst;
    J.stats = w.stats;
    return J;
}
static SrcJob16 src_job16(const Orient& o_src_major, const ConvWs& w, const float* x_rows, float* dx) {
    SrcJob16 J = {};
    J.s = items_dev(o_src_major);
    J.x = x_rows; J.rec = w.rec; J.dx = dx;
    return J;
}
static BwdJob1 bwd_job1(const Orient& o, const float* cp, const ConvWs& w, const float* x_src, const float* x_dst,
                        const float* h, const float* dh_a, const float* dh_b) {
    BwdJob1 J = {};
    J.s = items_dev(o);
    J.x_src = x_src; J.x_dst = x_dst; J.D = w.derived; J.p = conv_params_at(cp, 1);
    J.h = h; J.dh_a = dh_a; J.dh_b = dh_b; J.Z = w.Z; J.aux = w.aux; J.stats = w.stats;
    return J;
}

// Backward chain (linear_program_methods.py:241-247 read backwards), one stream, independent sweeps share a launch:
//   K1  C3  dst  (dh = d3v [premasked when it came from the fused head]) -> rec3, d2v
//   K2  C3  src  -> d2c                      K2' C2V dst (dh = d2v) -> rec2v, d1v_a
//   K3  C2C dst  (dh = d2c) -> rec2c, d1c_a  K3' C2V src -> d1c_b
//   K4  C2C src  -> d1v_b                    K4' C1C dst (dh = d1c_a + d1c_b)
//   K5  C1V dst  (dh = d1v_a + d1v_b)
//   K6  reduce the statistics partials (+ fc partials), K7 finalize (node_kernels.hip)
int fused_backward(const mllp_graph* g, const FusedModel& m, bool premasked, float* grads, float* loss, hipStream_t s) {
    const int G = fused_grid(g);
    int rc;
    const Orient& A = g->A;      // rows = constraints
    const Orient& At = g->At;    // rows = variables
    {   // K1
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(At, m.cp[4], m.c[4], m.h2c, m.h2v, premasked ? nullptr : m.h3v, m.d3v, nullptr, m.d2v, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C3"))) return rc;
    }
    {   // K2: C3 source-major (rows = constraints = A) and C2V destination-major
        SrcLaunch16 S = {};
        S.n_jobs = 1;
        S.job[0] = src_job16(A, m.c[4], m.h2c, m.d2c);
        hipLaunchKernelGGL(fused_src16_kernel, dim3(G), dim3(FT), 0, s, S);
        if ((rc = check_launch("fused_src16 C3"))) return rc;
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(At, m.cp[2], m.c[2], m.h1c, m.h1v, m.h2v, m.d2v, nullptr, m.d1v, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C2V"))) return rc;
    }
    {   // K3: C2C destination-major (rows = constraints) and C2V source-major (rows = constraints = A)
        BwdLaunch16 L = {};
        L.n_jobs = 1;
        L.job[0] = bwd_job16(A, m.cp[3], m.c[3], m.h1v, m.h1c, m.h2c, m.d2c, nullptr, m.d1c, true);
        hipLaunchKernelGGL(fused_bwd16_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd16 C2C"))) return rc;
        SrcLaunch16 S = {};
        S.n_jobs = 1;
        S.job[0] = src_job16(A, m.c[2], m.h1c, m.d1c_b);
        hipLaunchKernelGGL(fused_src16_kernel, dim3(G), dim3(FT), 0, s, S);
        if ((rc = check_launch("fused_src16 C2V"))) return rc;
    }
    {   // K4: C2C source-major (rows = variables = At) and C1C destination-major
        SrcLaunch16 S = {};
        S.n_jobs = 1;
        S.job[0] = src_job16(At, m.c[3], m.h1v, m.d1v_b);
        hipLaunchKernelGGL(fused_src16_kernel, dim3(G), dim3(FT), 0, s, S);
        if ((rc = check_launch("fused_src16 C2C"))) return rc;
    }
    {   // K4' + K5: layer 1, both convs (inputs are data: no input gradients)
        BwdLaunch1 L = {};
        L.n_jobs = 2;
        L.job[0] = bwd_job1(A, m.cp[1], m.c[1], m.x1, m.x2, m.h1c, m.d1c, m.d1c_b);
        L.job[1] = bwd_job1(At, m.cp[0], m.c[0], m.x2, m.x1, m.h1v, m.d1v, m.d1v_b);
        hipLaunchKernelGGL(fused_bwd1_kernel, dim3(G), dim3(FT), 0, s, L);
        if ((rc = check_launch("fused_bwd1"))) return rc;
    }
    {   // K6
        ReduceArgs R = {};
        for (int i = 0; i < MODEL_CONVS; ++i) { R.stats[i] = m.c[i].stats; R.out[i] = m.c[i].red; }
        R.nblk = G;
        R.head_part = premasked ? m.head_part : nullptr;
        R.head_out = grads + 4704;
        R.loss_out = loss;
        hipLaunchKernelGGL(fused_reduce_kernel, dim3(MODEL_CONVS * STAT_TILES + 1), dim3(FT), 0, s, R, MODEL_CONVS);
        if ((rc = check_launch("fused_reduce"))) return rc;
    }
    {   // K7: the 16x16 algebra of every conv, and the zero gradient of the never-used gconv3_s2w
        const float* cps[MODEL_CONVS] = {m.cp[0], m.cp[1], m.cp[2], m.cp[3], m.cp[4]};
        const int cins[MODEL_CONVS] = {1, 1, 16, 16, 16};
        const float* sts[MODEL_CONVS] = {m.c[0].red, m.c[1].red, m.c[2].red, m.c[3].red, m.c[4].red};
        const int nbs[MODEL_CONVS] = {1, 1, 1, 1, 1};
        float* grs[MODEL_CONVS] = {grads + 0, grads + 144, grads + 288, grads + 1392, grads + 2496};
        if ((rc = launch_finalize_batch(MODEL_CONVS, cps, cins, sts, nbs, grs, grads + 3600, 1104, s))) return rc;
    }
    return MLLP_OK;
}

}  // namespace mllp
