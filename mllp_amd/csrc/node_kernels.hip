// node_kernels.hip -- dense per-node work around the sparse sweeps (gfx950):
//   param_prep      fold the conv weights once per step (q' = Wk^T q / 4 etc.)
//   node_qp         Q' = X Pq^T + pq0, t = X Pt + pt0            (MFMA v_mfma_f32_16x16x4_f32, exact fp32)
//   bwd_pre         ReLU mask, gv = Wv^T g, ge, c -> per-destination backward record
//   param_stats     sum over nodes of the outer products that make up the parameter gradients
//                   (MFMA 16x16x4 with K = nodes; deterministic two-stage reduction, no atomics)
//   finalize_conv   partial statistics -> the conv's 9 gradient tensors
//   head            fc (16 -> 1) + BCEWithLogits forward/backward, fused
//   adam            torch.optim.Adam on the flat parameter buffer
//   topm_metrics    per-instance top-m prediction, correct count and F1
// Reference call sites: linear_program_methods.py:241-251 (layers, relu, fc),
// linear_program_experiment.py:41,139-151 (loss, Adam, top-k metrics).
#include "device_utils.h"
#include "internal.h"
#include "node_bodies.h"

namespace mllp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// param_prep: one workgroup per conv
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void param_prep_kernel(ConvParams p, int cin, float* __restrict__ D) {
    param_prep_body(p, cin, D);
}

// all convs of the model in one launch (workgroup = conv): four launches fewer on the forward critical path
struct PrepBatch {
    ConvParams p[MODEL_CONVS];
    int cin[MODEL_CONVS];
    float* D[MODEL_CONVS];
};
__global__ __launch_bounds__(BLOCK) void param_prep_batch_kernel(PrepBatch b) {
    param_prep_body(b.p[blockIdx.x], b.cin[blockIdx.x], b.D[blockIdx.x]);
}

int launch_param_prep_batch(int n, const float* const* conv_params, const int* cin, float* const* derived, hipStream_t s) {
    if (n < 1 || n > MODEL_CONVS) return fail(MLLP_EINVAL, "param_prep_batch: conv count");
    PrepBatch b;
    for (int i = 0; i < n; ++i) {
        b.p[i] = conv_params_at(conv_params[i], cin[i]);
        b.cin[i] = cin[i];
        b.D[i] = derived[i];
    }
    hipLaunchKernelGGL(param_prep_batch_kernel, dim3(n), dim3(BLOCK), 0, s, b);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "param_prep_batch");
}

int launch_param_prep(const float* conv_params, int cin, float* derived, hipStream_t s) {
    hipLaunchKernelGGL(param_prep_kernel, dim3(1), dim3(BLOCK), 0, s, conv_params_at(conv_params, cin), cin, derived);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "param_prep");
}

// -------------------------------------------------------------------------------------------------
// node_qp (cin = 16): one wavefront per 16 nodes, 4 MFMA 16x16x4 (K = 16 input channels)
//   A[i = lane & 15][k = lane >> 4] = X[node0 + i][4 s + k]     B[k][j = lane & 15] = PqT[4 s + k][j]
//   D[row = 4 (lane >> 4) + r][col = lane & 15]
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void node_qp_kernel(const float* __restrict__ X, int n, const float* __restrict__ D,
                                                        float* __restrict__ qp, float* __restrict__ t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int node0 = (blockIdx.x * 4 + wave) * 16;
    if (node0 >= n) return;
    const int r = lane & 15, kq = lane >> 4;
    const bool ok = node0 + r < n;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float tpart = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int d = 4 * s + kq;
        const float a = ok ? X[(size_t)(node0 + r) * 16 + d] : 0.0f;
        const float b = D[OFF_PQT + d * 16 + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        tpart = fmaf(a, D[OFF_PT + d], tpart);
    }
    tpart += __shfl_xor(tpart, 16, 64);
    tpart += __shfl_xor(tpart, 32, 64);
    if (kq == 0 && ok) t[node0 + r] = tpart + D[OFF_PT0];
    const float bias = D[OFF_PQ0 + r];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int node = node0 + kq * 4 + j;
        if (node < n) qp[(size_t)node * 16 + r] = acc[j] + bias;
    }
}

int launch_node_qp(const float* x_dst, int64_t n_dst, const float* derived, float* qp, float* t, hipStream_t s) {
    if (n_dst == 0) return MLLP_OK;
    const int64_t blocks = (n_dst + 63) / 64;
    hipLaunchKernelGGL(node_qp_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, s, x_dst, (int)n_dst, derived, qp, t);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "node_qp");
}

// -------------------------------------------------------------------------------------------------
// bwd_pre: 16 lanes per destination node.  g = dh * (h > 0) written back in place; record for the sweeps.
// -------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(BLOCK) void bwd_pre_kernel(int n, ConvParams p, const float* __restrict__ D,
                                                        const float* __restrict__ h, float* __restrict__ dh,
                                                        const float* __restrict__ xd, const float* __restrict__ qp,
                                                        const float* __restrict__ t, const float* __restrict__ Z,
                                                        const float* __restrict__ aux, float* __restrict__ rec) {
    // grid-stride over groups of 16 nodes: the lane's weights (a row of Wv^T, we_k, bv_k) are loaded once, and the
    // masked g row every lane needs (gv_k = sum_o g_o Wv[o][k]) is gathered from the 16 lanes of the group by DPP row
    // broadcasts instead of two more 64-byte row loads per lane (0.80 -> ... ms for 5.1 M nodes)
    const int k = threadIdx.x & 15;
    float wvt[16];
    if (CIN == 16) load_row16(D + OFF_WVT + k * 16, wvt);
    const float wek = p.we[k], bvk = p.bv[k];
    for (int node = blockIdx.x * 16 + (threadIdx.x >> 4); node < n; node += gridDim.x * 16) {
    const float hk = h[(size_t)node * 16 + k];
    const float gk = hk > 0.0f ? dh[(size_t)node * 16 + k] : 0.0f;
    float gr[16];
    if (CIN == 16) {
        gr[0] = dpp_mov<0x150>(gk);  gr[1] = dpp_mov<0x151>(gk);  gr[2] = dpp_mov<0x152>(gk);  gr[3] = dpp_mov<0x153>(gk);
        gr[4] = dpp_mov<0x154>(gk);  gr[5] = dpp_mov<0x155>(gk);  gr[6] = dpp_mov<0x156>(gk);  gr[7] = dpp_mov<0x157>(gk);
        gr[8] = dpp_mov<0x158>(gk);  gr[9] = dpp_mov<0x159>(gk);  gr[10] = dpp_mov<0x15A>(gk); gr[11] = dpp_mov<0x15B>(gk);
        gr[12] = dpp_mov<0x15C>(gk); gr[13] = dpp_mov<0x15D>(gk); gr[14] = dpp_mov<0x15E>(gk); gr[15] = dpp_mov<0x15F>(gk);
    }
    dh[(size_t)node * 16 + k] = gk;
    const float4 ax = reinterpret_cast<const float4*>(aux)[node];  // {u, rowmax, rinv, S}
    const float ge = row16_sum(gk * wek);
    const float gb = row16_sum(gk * bvk);
    if (CIN == 16) {
        const float gv = dot16(wvt, gr, 0.0f);
        const float Dn = row16_sum(gv * Z[(size_t)node * 16 + k]) + gb * ax.w + ge * ax.x;
        const float cc = gb - Dn;
        float* r = rec + (size_t)node * REC_W;
        r[k] = qp[(size_t)node * 16 + k];
        r[16 + k] = gv;
        if (k < 8) {
            const float tv = t[node];
            float v = 0.0f;
            v = k == 0 ? tv : v;
            v = k == 1 ? ax.y : v;
            v = k == 2 ? ax.z : v;
            v = k == 3 ? ge : v;
            v = k == 4 ? cc : v;
            r[32 + k] = v;
        }
    } else {
        const float gv = row16_sum(gk * p.Wv[k]);
        const float Dn = gv * Z[node] + gb * ax.w + ge * ax.x;
        const float cc = gb - Dn;
        const float x = xd[node];
        const float q1 = fmaf(D[OFF_PQ], x, D[OFF_PQ0]);
        const float t1 = fmaf(D[OFF_PT], x, D[OFF_PT0]);
        if (k < 8) {
            float v = 0.0f;
            v = k == 0 ? q1 : v;
            v = k == 1 ? gv : v;
            v = k == 2 ? t1 : v;
            v = k == 3 ? ax.y : v;
            v = k == 4 ? ax.z : v;
            v = k == 5 ? ge : v;
            v = k == 6 ? cc : v;
            rec[(size_t)node * 8 + k] = v;
        }
    }
    }
}

int launch_bwd_pre(int64_t n_dst, int cin, const float* conv_params, const ConvWs& w, const float* x_dst,
                     const float* h_out, float* dh, hipStream_t s) {
    if (n_dst == 0) return MLLP_OK;
    const int64_t blocks = std::min<int64_t>((n_dst + 15) / 16, 8192);      // grid-stride: 16 nodes per pass and block
    ConvParams p = conv_params_at(conv_params, cin);
    if (cin == 16)
        hipLaunchKernelGGL(bwd_pre_kernel<16>, dim3((unsigned)blocks), dim3(BLOCK), 0, s, (int)n_dst, p, w.derived,
                           h_out, dh, x_dst, w.qp, w.t, w.Z, w.aux, w.rec);
    else
        hipLaunchKernelGGL(bwd_pre_kernel<1>, dim3((unsigned)blocks), dim3(BLOCK), 0, s, (int)n_dst, p, w.derived, h_out,
                           dh, x_dst, w.qp, w.t, w.Z, w.aux, w.rec);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwd_pre");
}

// -------------------------------------------------------------------------------------------------
// param_stats: T[p][q] = sum_nodes L_node[p] * R_node[q] for seven 16x16 tiles
//   T0 g x^T   T1 g Z^T   T2 g e^T   T3 dq' x^T   T4 dq' e^T   T5 sc x^T   T6 sc e^T
//   e = [1, S, u, 0...],  sc = [ds, dt, 0...]
// cin = 16: MFMA with K = 4 nodes per instruction: A[p = lane & 15][kk = lane >> 4] = L[node0 + kk][p],
//           B[kk][q = lane & 15] = R[node0 + kk][q]  -- both are plain coalesced dword loads.
// -------------------------------------------------------------------------------------------------
int stat_blocks_for(int64_t n_dst) {
    int64_t b = (n_dst + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > STAT_BLOCKS_MAX) b = STAT_BLOCKS_MAX;
    return (int)b;
}

// 1024-thread workgroups: the grid is capped at STAT_BLOCKS_MAX partials (one per CU), and with 256 threads that was
// ONE wavefront per SIMD streaming 280 bytes per node with 24 dword loads in flight -- 1.25 ms for 2.5-5 M nodes, a
// quarter of what HBM allows.  16 wavefronts per workgroup, 8 chunks per pass.
constexpr int PS_BLOCK = 1024, PS_WAVES = PS_BLOCK / 64;
__global__ __launch_bounds__(PS_BLOCK) void param_stats16_kernel(int n, const float* __restrict__ g,
                                                              const float* __restrict__ x, const float* __restrict__ Z,
                                                              const float* __restrict__ aux,
                                                              const float* __restrict__ dqp,
                                                              const float* __restrict__ dsdt, float* __restrict__ out) {
    __shared__ float sh[4][STAT_FLOATS];      // the 16 wavefronts fold into it four at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc[STAT_TILES];
#pragma unroll
    for (int i = 0; i < STAT_TILES; ++i) acc[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // a pass covers CH consecutive chunks of 4 nodes: all loads of the pass are issued before its MFMAs
    constexpr int CH = 8;
    const int npass = (n + 4 * CH - 1) / (4 * CH);
    for (int ps = blockIdx.x * PS_WAVES + wave; ps < npass; ps += gridDim.x * PS_WAVES) {
        float ag[CH], adq[CH], asc[CH], bx[CH], bz[CH], be[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int node = (ps * CH + c) * 4 + kq;
            ag[c] = adq[c] = asc[c] = bx[c] = bz[c] = be[c] = 0.0f;
            if (node < n) {
                ag[c] = g[(size_t)node * 16 + r];
                adq[c] = dqp[(size_t)node * 16 + r];
                bx[c] = x[(size_t)node * 16 + r];
                bz[c] = Z[(size_t)node * 16 + r];
                const float2 sd = reinterpret_cast<const float2*>(dsdt)[node];
                const float4 ax = reinterpret_cast<const float4*>(aux)[node];
                asc[c] = r == 0 ? sd.x : (r == 1 ? sd.y : 0.0f);
                be[c] = r == 0 ? 1.0f : (r == 1 ? ax.w : (r == 2 ? ax.x : 0.0f));
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[c], bx[c], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[c], bz[c], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[c], be[c], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(adq[c], bx[c], acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(adq[c], be[c], acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(asc[c], bx[c], acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(asc[c], be[c], acc[6], 0, 0, 0);
        }
    }
    // fixed order: wavefronts 0-3 store, 4-7 / 8-11 / 12-15 add in turn
    for (int round = 0; round < PS_WAVES / 4; ++round) {
        if ((wave >> 2) == round) {
#pragma unroll
            for (int i = 0; i < STAT_TILES; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* d = &sh[wave & 3][i * 256 + (kq * 4 + j) * 16 + r];
                    *d = round == 0 ? acc[i][j] : *d + acc[i][j];
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < STAT_FLOATS; i += PS_BLOCK)
        out[(size_t)blockIdx.x * STAT_FLOATS + i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

// cin = 1: node features, Z, dq' are scalars; 16 lanes per node (lane o owns g_o); same tile layout out.
// (1024 threads and four nodes in flight per 16-lane group for the same reason as param_stats16_kernel)
__global__ __launch_bounds__(PS_BLOCK) void param_stats1_kernel(int n, const float* __restrict__ g,
                                                             const float* __restrict__ x, const float* __restrict__ Z,
                                                             const float* __restrict__ aux,
                                                             const float* __restrict__ dqp,
                                                             const float* __restrict__ dsdt, float* __restrict__ out) {
    constexpr int NG = PS_BLOCK / 16;       // 16-lane groups per workgroup
    __shared__ float sh[NG][16][12];
    const int grp = threadIdx.x >> 4, o = threadIdx.x & 15;
    float a[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) a[i] = 0.0f;
    const int stride = gridDim.x * NG;
    for (int node0 = blockIdx.x * NG + grp; node0 < n; node0 += 4 * stride) {
        float go[4], xv[4], zv[4], dq[4];
        float2 sd[4];
        float4 ax[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {        // all loads of the four nodes leave before the first is used
            const int node = min(node0 + k * stride, n - 1);
            go[k] = g[(size_t)node * 16 + o];
            xv[k] = x[node]; zv[k] = Z[node]; dq[k] = dqp[node];
            sd[k] = reinterpret_cast<const float2*>(dsdt)[node];
            ax[k] = reinterpret_cast<const float4*>(aux)[node];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (node0 + k * stride >= n) break;
            a[0] = fmaf(go[k], xv[k], a[0]);        // T0[o][0]
            a[1] = fmaf(go[k], zv[k], a[1]);        // T1[o][0]
            a[2] += go[k];                          // T2[o][0]
            a[3] = fmaf(go[k], ax[k].w, a[3]);      // T2[o][1]
            a[4] = fmaf(go[k], ax[k].x, a[4]);      // T2[o][2]
            a[5] = fmaf(dq[k], xv[k], a[5]);        // T3[0][0]
            a[6] += dq[k];                          // T4[0][0]
            a[7] = fmaf(sd[k].x, xv[k], a[7]);      // T5[0][0]
            a[8] = fmaf(sd[k].y, xv[k], a[8]);      // T5[1][0]
            a[9] += sd[k].x;                        // T6[0][0]
            a[10] += sd[k].y;                       // T6[1][0]
        }
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) sh[grp][o][i] = a[i];
    __syncthreads();
    float* dst = out + (size_t)blockIdx.x * STAT_FLOATS;
    for (int i = threadIdx.x; i < STAT_FLOATS; i += PS_BLOCK) dst[i] = 0.0f;
    __syncthreads();
    if (threadIdx.x < 16 * 11) {
        const int oo = threadIdx.x / 11, i = threadIdx.x % 11;
        float v = 0.0f;
        for (int gq = 0; gq < NG; ++gq) v += sh[gq][oo][i];
        if (i == 0) dst[0 * 256 + oo * 16 + 0] = v;
        else if (i == 1) dst[1 * 256 + oo * 16 + 0] = v;
        else if (i <= 4) dst[2 * 256 + oo * 16 + (i - 2)] = v;
        else if (oo == 0) {   // scalar statistics are identical in every lane of a group: take lane 0
            if (i == 5) dst[3 * 256] = v;
            else if (i == 6) dst[4 * 256] = v;
            else if (i == 7) dst[5 * 256] = v;
            else if (i == 8) dst[5 * 256 + 16] = v;
            else if (i == 9) dst[6 * 256] = v;
            else dst[6 * 256 + 16] = v;
        }
    }
}

int launch_param_stats(int cin, int64_t n_dst, const ConvWs& w, const float* x_dst, const float* g, hipStream_t s) {
    const int blocks = stat_blocks_for(n_dst);
    if (cin == 16)
        hipLaunchKernelGGL(param_stats16_kernel, dim3(blocks), dim3(PS_BLOCK), 0, s, (int)n_dst, g, x_dst, w.Z, w.aux,
                           w.dqp, w.dsdt, w.stats);
    else
        hipLaunchKernelGGL(param_stats1_kernel, dim3(blocks), dim3(PS_BLOCK), 0, s, (int)n_dst, g, x_dst, w.Z, w.aux, w.dqp,
                           w.dsdt, w.stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "param_stats");
}

// -------------------------------------------------------------------------------------------------
// finalize_conv: fixed-order sum of the per-workgroup partial tiles, then the small matrix algebra
// (oracle/spmm_form.py::conv_bwd "grads = {...}").  One workgroup.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void finalize_conv_kernel(int cin, ConvParams p, const float* __restrict__ stats,
                                                             int nblk, float* __restrict__ grads) {
    finalize_conv_body(cin, p, stats, nblk, grads);
}

// all convs of the model in one launch (workgroup = conv) + one workgroup that zeroes a gradient range (the
// never-used gconv3_s2w): the single-workgroup tail of the step shrinks from five launches on two streams to one
struct FinalizeBatch {
    ConvParams p[MODEL_CONVS];
    int cin[MODEL_CONVS];
    const float* stats[MODEL_CONVS];
    int nblk[MODEL_CONVS];
    float* grads[MODEL_CONVS];
    float* zero;
    int n_zero;
};
__global__ __launch_bounds__(1024) void finalize_batch_kernel(FinalizeBatch b, int n) {
    const int i = blockIdx.x;
    if (i < n) {
        finalize_conv_body(b.cin[i], b.p[i], b.stats[i], b.nblk[i], b.grads[i]);
    } else {
        for (int k = threadIdx.x; k < b.n_zero; k += 1024) b.zero[k] = 0.0f;
    }
}

int launch_finalize_batch(int n, const float* const* conv_params, const int* cin, const float* const* stats,
                          const int* n_stat_blocks, float* const* grads, float* zero, int n_zero, hipStream_t s) {
    if (n < 1 || n > MODEL_CONVS) return fail(MLLP_EINVAL, "finalize_batch: conv count");
    FinalizeBatch b;
    for (int i = 0; i < n; ++i) {
        b.p[i] = conv_params_at(conv_params[i], cin[i]);
        b.cin[i] = cin[i];
        b.stats[i] = stats[i];
        b.nblk[i] = n_stat_blocks[i];
        b.grads[i] = grads[i];
    }
    b.zero = zero;
    b.n_zero = n_zero;
    hipLaunchKernelGGL(finalize_batch_kernel, dim3(n + (n_zero > 0 ? 1 : 0)), dim3(1024), 0, s, b, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "finalize_batch");
}

int launch_finalize_conv(int cin, const float* conv_params, const float* stats, int n_stat_blocks, float* grads,
                         hipStream_t s) {
    hipLaunchKernelGGL(finalize_conv_kernel, dim3(1), dim3(1024), 0, s, cin, conv_params_at(conv_params, cin), stats,
                       n_stat_blocks, grads);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "finalize_conv");
}

// -------------------------------------------------------------------------------------------------
// head: z_i = <h_i, w_fc> + b_fc ; BCEWithLogits ; dh_i = dz_i w_fc ; partial sums of dW, db, loss
// 16 lanes per variable.  partials: [blocks][18] = {dW[16], db, loss}
// -------------------------------------------------------------------------------------------------
constexpr int HEAD_BLOCKS_MAX = 1024;
int head_blocks_for(int64_t n) {
    int64_t b = (n + 63) / 64;
    if (b < 1) b = 1;
    if (b > HEAD_BLOCKS_MAX) b = HEAD_BLOCKS_MAX;
    return (int)b;
}

__global__ __launch_bounds__(BLOCK) void head_kernel(int mode, int n, const float* __restrict__ h,
                                                     const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                     const float* __restrict__ inv_n, const float* __restrict__ labels,
                                                     float inv_batch, const float* __restrict__ dz_in,
                                                     float* __restrict__ logits, float* __restrict__ dh,
                                                     float* __restrict__ partials) {
    // A wavefront takes 64 consecutive nodes per round.  Round part 1: sixteen steps, step u reads the 16 channels of nodes
    // base + 4 u + q (q = the 16-lane group inside the wavefront: 256 contiguous bytes per load) and sums <h, w>; lane (q, c)
    // keeps the logit of step u = c, i.e. of node base + 4 c + q.  Part 2: every lane does the scalar arithmetic of ITS node
    // once (sigmoid, loss: with 16 lanes per node all repeating it the kernel was VALU-bound, 0.45 ms for 0.7 GB at 5.12 M
    // nodes).  Part 3: sixteen steps again, dz of step u comes from lane u of the group (DPP row broadcast), dh = dz w.
    __shared__ float sh[BLOCK / 64][18];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, c = lane & 15;
    const float w = fcw[c], b = fcb[0];
    float accw = 0.0f, accb = 0.0f, accl = 0.0f;
    const int per_round = (BLOCK / 64) * 64;
    for (int base = blockIdx.x * per_round + wave * 64; base < n; base += gridDim.x * per_round) {
        float hv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) hv[u] = h[(size_t)min(base + 4 * u + q, n - 1) * 16 + c];
        float z = 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float zu = row16_sum(hv[u] * w) + b;
            z = c == u ? zu : z;
        }
        const int node = base + 4 * c + q;
        const bool ok = node < n;
        if (ok && logits) logits[node] = z;
        if (mode == 0) continue;
        float dz = 0.0f;
        if (ok) {
            if (mode == 1) {
                dz = dz_in[node];
            } else {
                const float y = labels[node];
                const float wn = inv_n[node] * inv_batch;
                const float e = expf(-fabsf(z));
                const float sig = z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
                dz = wn * (sig - y);
                accl += wn * (fmaxf(z, 0.0f) - z * y + log1pf(e));
            }
        }
        accb += dz;
        float dzu[16];
        dzu[0] = dpp_mov<0x150>(dz);  dzu[1] = dpp_mov<0x151>(dz);  dzu[2] = dpp_mov<0x152>(dz);  dzu[3] = dpp_mov<0x153>(dz);
        dzu[4] = dpp_mov<0x154>(dz);  dzu[5] = dpp_mov<0x155>(dz);  dzu[6] = dpp_mov<0x156>(dz);  dzu[7] = dpp_mov<0x157>(dz);
        dzu[8] = dpp_mov<0x158>(dz);  dzu[9] = dpp_mov<0x159>(dz);  dzu[10] = dpp_mov<0x15A>(dz); dzu[11] = dpp_mov<0x15B>(dz);
        dzu[12] = dpp_mov<0x15C>(dz); dzu[13] = dpp_mov<0x15D>(dz); dzu[14] = dpp_mov<0x15E>(dz); dzu[15] = dpp_mov<0x15F>(dz);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int nu = base + 4 * u + q;
            if (nu < n) dh[(size_t)nu * 16 + c] = dzu[u] * w;
            accw = fmaf(dzu[u], hv[u], accw);
        }
    }
    if (mode == 0) return;
    // per channel: the four groups of a wavefront, then the wavefronts; db and the loss: all lanes
    accw += __shfl_xor(accw, 16, 64);
    accw += __shfl_xor(accw, 32, 64);
    for (int o = 32; o > 0; o >>= 1) { accb += __shfl_xor(accb, o, 64); accl += __shfl_xor(accl, o, 64); }
    if (lane < 16) sh[wave][lane] = accw;
    if (lane == 0) { sh[wave][16] = accb; sh[wave][17] = accl; }
    __syncthreads();
    if (threadIdx.x < 18) {
        float v = 0.0f;
        for (int gq = 0; gq < BLOCK / 64; ++gq) v += sh[gq][threadIdx.x];
        partials[(size_t)blockIdx.x * 18 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void head_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                              float* __restrict__ grad_fc, float* __restrict__ loss) {
    __shared__ float sh[8][32];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    float v = 0.0f;
    if (col < 18) {
        int b = slice;
        for (; b + 24 < nblk; b += 32) {      // four loads in flight, fixed order
            const float t0 = partials[(size_t)b * 18 + col], t1 = partials[(size_t)(b + 8) * 18 + col];
            const float t2 = partials[(size_t)(b + 16) * 18 + col], t3 = partials[(size_t)(b + 24) * 18 + col];
            v = (((v + t0) + t1) + t2) + t3;
        }
        for (; b < nblk; b += 8) v += partials[(size_t)b * 18 + col];
    }
    sh[slice][col] = v;
    __syncthreads();
    if (threadIdx.x < 18) {
        float t = 0.0f;
        for (int q = 0; q < 8; ++q) t += sh[q][threadIdx.x];
        if (threadIdx.x < 17) grad_fc[threadIdx.x] = t;
        else if (loss) loss[0] = t;
    }
}

int launch_head(int mode, int64_t n, const float* h3v, const float* fc_w, const float* fc_b, const float* inv_n,
                const float* labels, float inv_batch, const float* dlogits_in, float* logits, float* dh3v,
                float* partials, hipStream_t s) {
    const int blocks = head_blocks_for(n);
    hipLaunchKernelGGL(head_kernel, dim3(blocks), dim3(BLOCK), 0, s, mode, (int)n, h3v, fc_w, fc_b, inv_n, labels,
                       inv_batch, dlogits_in, logits, dh3v, partials);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "head");
}

int launch_head_finalize(const float* partials, int n_blocks, float* grad_fc, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(BLOCK), 0, s, partials, n_blocks, grad_fc, loss);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "head_finalize");
}

// -------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults apart from lr): one workgroup, the step counter lives on the device so
// that a captured hipGraph can be replayed.  state = {step, lr, beta1, beta2}
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    float* __restrict__ state, float eps, float gscale, int n) {
    adam_body(p, g, m, v, state, eps, gscale, n);
}

// many parameters (AngleModel at feat_dim 256: 529 k): slices of 4096 elements per workgroup; the workgroups only READ the
// step counter, a one-thread launch behind them advances it (stream order; capturable like the single launch)
constexpr int ADAM_SLICE = 4096;
__global__ __launch_bounds__(1024) void adam_wide_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ state, float eps, float gscale, int n) {
    const int o = blockIdx.x * ADAM_SLICE;
    adam_slice(p + o, g + o, m + o, v + o, state[0] + 1.0f, state[1], state[2], state[3], eps, gscale, min(ADAM_SLICE, n - o));
}
__global__ void adam_tick_kernel(float* __restrict__ state) { state[0] += 1.0f; }

int launch_adam(float* p, const float* g, float* m, float* v, float* state, float eps, float gscale, int64_t n,
                hipStream_t s) {
    if (n > 4 * ADAM_SLICE) {
        hipLaunchKernelGGL(adam_wide_kernel, dim3((unsigned)((n + ADAM_SLICE - 1) / ADAM_SLICE)), dim3(1024), 0, s, p, g, m, v, state,
                           eps, gscale, (int)n);
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, state);
    } else
    hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(1024), 0, s, p, g, m, v, state, eps, gscale, (int)n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "adam");
}

__global__ void fill_zero_kernel(float* __restrict__ p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 0.0f;
}
int launch_fill_zero(float* p, int64_t n, hipStream_t s) {
    if (n <= 0) return MLLP_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "fill_zero");
}

// -------------------------------------------------------------------------------------------------
// top-m metrics: one workgroup per instance; 4-pass radix select (8 bits per pass) of the m-th largest
// logit, then TP / F1.  Ties at the threshold are taken in index order.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned orderable(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TOPM_T = 1024;
__global__ __launch_bounds__(TOPM_T) void topm_metrics_kernel(const int* __restrict__ ptr_n, const int* __restrict__ ptr_m,
                                                              const float* __restrict__ logits,
                                                              const float* __restrict__ labels, float* __restrict__ out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need, s_wtot[4];
    __shared__ float s_red[4][TOPM_T / 64];
    __shared__ unsigned s_cnt[TOPM_T / 64];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = ptr_n[k], n = ptr_n[k + 1] - beg;
    int m = ptr_m[k + 1] - ptr_m[k];
    if (m > n) m = n;
    const float* z = logits + beg;
    const float* y = labels + beg;
    if (m <= 0 || n <= 0) {
        if (tid == 0) { out[2 * k] = 0.0f; out[2 * k + 1] = 0.0f; }
        return;
    }
    if (tid == 0) { s_prefix = 0u; s_need = (unsigned)m; }
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix, need = s_need;
        if (pass == 3) {
            // top byte (sign + 7 exponent bits): a handful of distinct values, so plain LDS atomics serialise 64 lanes
            // on one address (the kernel took 93 us for the Netlib batch, all of it here).  Wave-aggregated instead:
            // one atomic per distinct bin of the wavefront.
            for (int i0 = 0; i0 < n; i0 += TOPM_T) {
                const int i = i0 + tid;
                const bool have = i < n;
                const unsigned bin = have ? (orderable(z[i]) >> 24) : 0u;
                unsigned long long todo = __ballot(have);
                while (todo) {
                    const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                    const unsigned long long same = __ballot(have && bin == b) & todo;
                    if (lane == leader) atomicAdd(&hist[b], (unsigned)__popcll(same));
                    todo &= ~same;
                }
            }
        } else {
            for (int i = tid; i < n; i += TOPM_T) {
                const unsigned key = orderable(z[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        // the bin that holds the need-th largest key: suffix sums over the 256 bins in four wavefronts (one thread
        // walking the bins took 256 dependent LDS reads per pass: most of the kernel's 90 us on the Netlib batch)
        unsigned h = 0u, sfx = 0u;
        if (tid < 256) {
            h = hist[tid];
            sfx = h;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_down(sfx, o, 64);
                if (lane + o < 64) sfx += v;
            }
            if (lane == 0) s_wtot[wave] = sfx;
        }
        __syncthreads();
        if (tid < 256) {
            unsigned above = 0u;
            for (int w = wave + 1; w < 4; ++w) above += s_wtot[w];
            const unsigned ge = sfx + above, gt = ge - h;      // keys of this prefix with bin >= tid / > tid
            if (ge >= need && gt < need) {                     // exactly one bin
                s_need = need - gt;
                s_prefix = prefix | ((unsigned)tid << shift);
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned thr = s_prefix;
    const unsigned need_eq = s_need;   // how many keys equal to thr belong to the top-m (>= 1)
    float vals[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // tp (key > thr), sum y, count (key == thr), sum y (key == thr)
    for (int i = tid; i < n; i += TOPM_T) {
        const unsigned key = orderable(z[i]);
        const float yi = y[i];
        vals[1] += yi;
        if (key > thr) vals[0] += yi;
        else if (key == thr) { vals[2] += 1.0f; vals[3] += yi; }
    }
    float tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = vals[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_red[q][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = 0.0f;
        for (int w = 0; w < TOPM_T / 64; ++w) v += s_red[q][w];
        tot[q] = v;
    }
    float TP = tot[0];
    if ((unsigned)tot[2] == need_eq) {
        TP += tot[3];
    } else {
        // tie at the threshold: the first need_eq equal keys in INDEX order belong to the top-m
        // (ordered block scan with wave ballots; identical columns make this common on real LPs)
        unsigned running = 0u;
        float tie = 0.0f;
        for (int base = 0; base < n && running < need_eq; base += TOPM_T) {
            const int i = base + tid;
            const bool f = i < n && orderable(z[i]) == thr;
            const unsigned long long bal = __ballot(f);
            const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == 0) s_cnt[wave] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned off = 0u, total = 0u;
            for (int w = 0; w < TOPM_T / 64; ++w) {
                const unsigned c = s_cnt[w];
                if (w < wave) off += c;
                total += c;
            }
            if (f && running + off + before < need_eq) tie += y[i];
            running += total;
        }
        for (int o = 32; o > 0; o >>= 1) tie += __shfl_xor(tie, o, 64);
        __syncthreads();
        if (lane == 0) s_red[0][wave] = tie;
        __syncthreads();
        for (int w = 0; w < TOPM_T / 64; ++w) TP += s_red[0][w];
    }
    if (tid == 0) {
        const float FP = (float)m - TP, FN = tot[1] - TP;
        const float den = 2.0f * TP + FP + FN;
        out[2 * k] = TP;
        out[2 * k + 1] = den > 0.0f ? 2.0f * TP / den : 0.0f;
    }
}

int launch_topm_metrics(const mllp_graph* g, const float* logits, const float* labels, void* scratch, float* out,
                        hipStream_t s) {
    (void)scratch;
    if (g->n_inst == 0) return MLLP_OK;
    hipLaunchKernelGGL(topm_metrics_kernel, dim3((unsigned)g->n_inst), dim3(TOPM_T), 0, s, g->inst_ptr_n, g->inst_ptr_m,
                       logits, labels, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "topm_metrics");
}

}  // namespace mllp
