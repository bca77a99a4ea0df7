// angle.hip -- SURVEY.md section 8f-4: the `angleNet` method of the reference.
//
// `AngleModel` (reference linear_program_methods.py:187-200) runs three PyG TransformerConv layers (2 -> F, F -> F and
// the SAME F -> F layer again; `gconv3` is constructed but never called) and Linear(F, 1) over the COMPLETE directed
// graph on the N = n + 1 "variables" of one LP instance (its columns plus the right-hand side), whose edge attribute is
// the cosine similarity of two rows of the Q factor of [A | b]^T (build_graph_from_Q_sets, :119-130; F = 256 in
// linear_program_experiment.py:83).  On a complete graph the layer is dense attention with a scalar edge bias:
//     Q = X Wq^T + bq, K = X Wk^T + bk, V = X Wv^T + bv, R = X Ws^T + bs
//     L_ij = (Q_i . K_j + (Q_i . we) A_ij) / sqrt(F)      for j != i           (key_j + lin_edge(a_ij), target i, source j)
//     alpha = softmax_j(L)   (torch_geometric.utils.softmax: exp(L - max) / (sum + 1e-16))
//     O_i  = sum_j alpha_ij V_j + (sum_j alpha_ij A_ij) we + R_i ,   H = relu(O)
// A [N, N] is the dense, SYMMETRIC cosine matrix (diagonal ignored: the graph has no self loops).
//
// Round 4: everything is hand-written for the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 157 TFLOP/s peak);
// rocBLAS is neither linked nor loaded, and no N x N matrix exists in HBM:
//   * attn_kernel<F, MODE>   one wavefront per (block of 16 "Y" nodes, range of "X" blocks); four wavefronts share the X
//       tiles, staged one block ahead into two LDS buffers.  Every product is computed
//       TRANSPOSED so that no tile ever changes its register layout: T[x][y] = X_x . Y_y lands with y in the lane and x in
//       the register, which is exactly the B operand of the accumulation  acc[f][y] += W[x][f] T'[x][y].
//         FWD  X = keys, Y = queries:  T = K Q^T -> online softmax over the X blocks (edge bias (Q . we) A, diagonal
//              masked, side sum u = sum p A) -> acc += V^T p;  partial {m, l, u, acc} per range, merged by fwd_combine
//         BQ   X = keys, Y = queries:  p from the saved row max / 1 / sum, dp = V dO^T + (dO . we) A,
//              dz = p (dp - D) / sqrt(F) -> acc += K^T dz (dQ), r = sum dz A
//         BKV  X = queries, Y = keys:  the same p, dz with the roles exchanged -> accV += dO^T p (dV), accK += Q^T dz (dK)
//       The backward recomputes p from {row max, 1 / (sum + 1e-16)} as the sparse path does; D_i = dO_i . (alpha V)_i + u_i s_i.
//   * gemm_kernel            one 64 x 64 x 16 LDS-tiled MFMA GEMM with general strides (X W^T, dY^T X, dY W), K split
//       over workgroups where the output is small (weight gradients: K = N), a ones column for the bias gradients.
// The backward pass is hand-derived (tests/test_angle.py checks it against fp64 autograd of the oracle's literal
// TransformerConv on the edge list).  Deterministic: fixed ranges, fixed summation orders, no atomics.
#include <algorithm>
#include <cmath>

#include "device_utils.h"
#include "internal.h"

namespace mllp {
namespace {

constexpr int AT = 256;     // threads of the row kernels
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 f4zero() { return f4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < AT / 64; ++w) t += sh[w];
    return t;
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < AT / 64; ++w) t += sh[w];
    return t;
}

// =================================================================================================== GEMM
// C[m][n] = sum over pairs p, k of A_p(m, k) B_p(n, k)  (+ bias[n]),  A_p(m, k) = A_p[m sam + k sak], B_p(n, k) = B_p[n sbn + k sbk]
struct GemmProb {
    const float* A[4];
    const float* B[4];
    float* C;                 // [M][ldc]           (ksplits == 1)
    const float* bias;        // [N] or nullptr     (ksplits == 1)
    int npairs;
};
struct GemmBatch {
    GemmProb p[4];
    int M, N, K;              // the same for every problem of the batch
    long long sam, sak, sbn, sbk;
    int ldc;
    int ones_col;             // B(N - 1, k) = 1: the last output column is the column sum of A (bias gradients)
    int ksplits, kchunk;      // K is cut into ksplits ranges of kchunk (a multiple of 16); > 1: partial sums
    float* partial;           // [problem][ksplit][M][N]
};
constexpr int GT = 64, GK = 32, GS = GK + 1;

// 64 x 64 output tile, K in steps of 32: the 16 elements a thread stages per step are loaded one step ahead (registers),
// so the L2 latency of a step hides behind the 32 MFMAs per wavefront of the previous one.
template <bool AK1, bool BK1>
__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch g) {
    __shared__ float As[GT * GS], Bs[GT * GS];
    const GemmProb& pr = g.p[blockIdx.z];
    const int tiles_n = (g.N + GT - 1) / GT;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * GT, n0 = tn * GT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int k_lo = blockIdx.y * g.kchunk, k_hi = min(g.K, k_lo + g.kchunk);
    const int steps_per_pair = k_hi > k_lo ? (k_hi - k_lo + GK - 1) / GK : 0;
    const int steps = steps_per_pair * pr.npairs;
    f4 acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
    float ra[8], rb[8];
    auto fetch = [&](int step) {
        const int p = step / steps_per_pair, k0 = k_lo + (step % steps_per_pair) * GK;
        const float* __restrict__ A = pr.A[p];
        const float* __restrict__ B = pr.B[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (AK1) { m = t >> 2; k = (t & 3) * 8 + i; } else { k = t >> 3; m = (t & 7) * 8 + i; }
            const bool ok = m0 + m < g.M && k0 + k < k_hi;
            ra[i] = ok ? A[(long long)(m0 + m) * g.sam + (long long)(k0 + k) * g.sak] : 0.0f;
            int n, kb;
            if (BK1) { n = t >> 2; kb = (t & 3) * 8 + i; } else { kb = t >> 3; n = (t & 7) * 8 + i; }
            const bool okb = n0 + n < g.N && k0 + kb < k_hi;
            float v = 0.0f;
            if (okb) v = (g.ones_col && n0 + n == g.N - 1) ? 1.0f : B[(long long)(n0 + n) * g.sbn + (long long)(k0 + kb) * g.sbk];
            rb[i] = v;
        }
    };
    if (steps > 0) fetch(0);
    for (int step = 0; step < steps; ++step) {
        __syncthreads();                    // everybody has left the tiles of the previous step
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (AK1) { m = t >> 2; k = (t & 3) * 8 + i; } else { k = t >> 3; m = (t & 7) * 8 + i; }
            As[m * GS + k] = ra[i];
            int n, kb;
            if (BK1) { n = t >> 2; kb = (t & 3) * 8 + i; } else { kb = t >> 3; n = (t & 7) * 8 + i; }
            Bs[n * GS + kb] = rb[i];
        }
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const float a = As[(16 * w + (lane & 15)) * GS + 4 * kk + (lane >> 4)];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4(a, Bs[(16 * nt + (lane & 15)) * GS + 4 * kk + (lane >> 4)], acc[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * w + 4 * (lane >> 4) + r, n = n0 + 16 * nt + (lane & 15);
            if (m < g.M && n < g.N) {
                if (g.ksplits == 1 && !g.ones_col) pr.C[(long long)m * g.ldc + n] = acc[nt][r] + (pr.bias ? pr.bias[n] : 0.0f);
                else g.partial[(((long long)blockIdx.z * g.ksplits + blockIdx.y) * g.M + m) * g.N + n] = acc[nt][r];
            }
        }
}

// The same tile with 16-byte global loads (every operand's contiguous dimension is a multiple of 4 elements from an aligned
// base: the F x F layers), 32-bit offsets and a row stride of 36 floats in LDS (16-byte aligned rows, conflict-free MFMA
// operand reads): a quarter of the load / address instructions of the scalar kernel, which stays for the 2-column layer.
constexpr int GV = 36;
template <bool AK1, bool BK1>
__global__ __launch_bounds__(256) void gemm_vec_kernel(GemmBatch g) {
    __shared__ __attribute__((aligned(16))) float As[GT * GV], Bs[GT * GV];
    const GemmProb& pr = g.p[blockIdx.z];
    const int tiles_n = (g.N + GT - 1) / GT;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * GT, n0 = tn * GT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int k_lo = blockIdx.y * g.kchunk, k_hi = min(g.K, k_lo + g.kchunk);
    const int steps_per_pair = k_hi > k_lo ? (k_hi - k_lo + GK - 1) / GK : 0;
    const int steps = steps_per_pair * pr.npairs;
    const int sam = (int)g.sam, sak = (int)g.sak, sbn = (int)g.sbn, sbk = (int)g.sbk;
    f4 acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
    f4 ra[2], rb[2];
    // one 16-byte piece: K1: elements (r, k .. k + 3) of P[r sr + k]; else elements (r .. r + 3, k) of P[k sk + r]
    auto piece = [&](const float* __restrict__ P, bool k1, int r, int R, int k, int sr, int sk, bool ones) -> f4 {
        if (k1) {
            if (r < R && k + 3 < k_hi && !(ones && r == R - 1)) return *reinterpret_cast<const f4*>(P + r * sr + k);
            f4 v = f4zero();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r < R && k + j < k_hi) v[j] = (ones && r == R - 1) ? 1.0f : P[r * sr + k + j];
            return v;
        }
        if (k < k_hi && r + 3 < R - (ones ? 1 : 0)) return *reinterpret_cast<const f4*>(P + k * sk + r);
        f4 v = f4zero();
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k < k_hi && r + j < R) v[j] = (ones && r + j == R - 1) ? 1.0f : P[k * sk + r + j];
        return v;
    };
    auto fetch = [&](int step) {
        const int p = step / steps_per_pair, k0 = k_lo + (step % steps_per_pair) * GK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i;
            if (AK1) ra[i] = piece(pr.A[p], true, m0 + (idx >> 3), g.M, k0 + 4 * (idx & 7), sam, 1, false);
            else ra[i] = piece(pr.A[p], false, m0 + 4 * (idx & 15), g.M, k0 + (idx >> 4), 1, sak, false);
            if (BK1) rb[i] = piece(pr.B[p], true, n0 + (idx >> 3), g.N, k0 + 4 * (idx & 7), sbn, 1, g.ones_col != 0);
            else rb[i] = piece(pr.B[p], false, n0 + 4 * (idx & 15), g.N, k0 + (idx >> 4), 1, sbk, g.ones_col != 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i;
            if (AK1) *reinterpret_cast<f4*>(&As[(idx >> 3) * GV + 4 * (idx & 7)]) = ra[i];
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(4 * (idx & 15) + j) * GV + (idx >> 4)] = ra[i][j];
            if (BK1) *reinterpret_cast<f4*>(&Bs[(idx >> 3) * GV + 4 * (idx & 7)]) = rb[i];
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(4 * (idx & 15) + j) * GV + (idx >> 4)] = rb[i][j];
        }
    };
    if (steps > 0) fetch(0);
    for (int step = 0; step < steps; ++step) {
        __syncthreads();                    // everybody has left the tiles of the previous step
        stage();
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const float a = As[(16 * w + (lane & 15)) * GV + 4 * kk + (lane >> 4)];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma4(a, Bs[(16 * nt + (lane & 15)) * GV + 4 * kk + (lane >> 4)], acc[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * w + 4 * (lane >> 4) + r, n = n0 + 16 * nt + (lane & 15);
            if (m < g.M && n < g.N) {
                if (g.ksplits == 1 && !g.ones_col) pr.C[(long long)m * g.ldc + n] = acc[nt][r] + (pr.bias ? pr.bias[n] : 0.0f);
                else g.partial[(((long long)blockIdx.z * g.ksplits + blockIdx.y) * g.M + m) * g.N + n] = acc[nt][r];
            }
        }
}

// C[m][n] = beta C[m][n] + sum over the K ranges; with a ones column: column N - 1 goes to colsum[m] instead
struct ReduceBatch {
    float* C[4];
    float* colsum[4];
    int M, N, ldc, ksplits, ones_col;
    float beta;
    const float* partial;
};
__global__ __launch_bounds__(AT) void gemm_reduce_kernel(ReduceBatch g) {
    const long long total = (long long)g.M * g.N;
    for (long long i = (long long)blockIdx.x * AT + threadIdx.x; i < total; i += (long long)gridDim.x * AT) {
        const int m = (int)(i / g.N), n = (int)(i % g.N);
        float v = 0.0f;
        for (int s = 0; s < g.ksplits; ++s) v += g.partial[(((long long)blockIdx.z * g.ksplits + s) * g.M + m) * g.N + n];
        float* dst = (g.ones_col && n == g.N - 1) ? g.colsum[blockIdx.z] + m : g.C[blockIdx.z] + (long long)m * g.ldc + n;
        *dst = g.beta != 0.0f ? g.beta * *dst + v : v;
    }
}

int check(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, what);
}

// out[f] = beta out[f] + sum_k (A1[k][f] b1[k] + A2[k][f] b2[k]): a GEMM with one output column would leave 63 of 64 tile
// columns empty; chunks of 32 rows, then the same reduction as the split GEMMs
constexpr int WC_ROWS = 32;
__global__ __launch_bounds__(AT) void wcolsum_kernel(int K, int F, const float* __restrict__ A1, const float* __restrict__ b1,
                                                     const float* __restrict__ A2, const float* __restrict__ b2,
                                                     float* __restrict__ partial) {
    const int k0 = blockIdx.x * WC_ROWS, k1 = min(K, k0 + WC_ROWS);
    for (int c = threadIdx.x; c < F; c += AT) {
        float acc = 0.0f;
        for (int k = k0; k < k1; k += 8) {          // eight rows of loads in flight
            float a1[8], a2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = min(k + i, k1 - 1);
                a1[i] = A1[(size_t)kk * F + c];
                a2[i] = A2 ? A2[(size_t)kk * F + c] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k + i < k1) {
                    acc = fmaf(a1[i], b1[k + i], acc);
                    if (A2) acc = fmaf(a2[i], b2[k + i], acc);
                }
        }
        partial[(size_t)blockIdx.x * F + c] = acc;
    }
}
int launch_wcolsum(hipStream_t s, int K, int F, const float* A1, const float* b1, const float* A2, const float* b2, float beta,
                   float* out, float* partial);

// strides of the three forms on row-major operands
struct GemmShape {
    int M, N, K;
    long long sam, sak, sbn, sbk;
    int ldc;
};
int launch_gemm(hipStream_t s, const GemmShape& sh, int nprob, const GemmProb* probs, int ones_col, int ksplits, float beta,
                float* partial, float* const* colsum) {
    if (sh.M == 0 || sh.N == 0) return MLLP_OK;
    GemmBatch g;
    for (int i = 0; i < nprob; ++i) g.p[i] = probs[i];
    g.M = sh.M; g.N = sh.N; g.K = sh.K; g.sam = sh.sam; g.sak = sh.sak; g.sbn = sh.sbn; g.sbk = sh.sbk; g.ldc = sh.ldc;
    g.ones_col = ones_col;
    g.ksplits = ksplits;
    g.kchunk = ((sh.K + ksplits - 1) / ksplits + GK - 1) / GK * GK;
    g.partial = partial;
    const dim3 grid((unsigned)(((sh.M + GT - 1) / GT) * ((sh.N + GT - 1) / GT)), (unsigned)ksplits, (unsigned)nprob);
    const bool ak1 = sh.sak == 1, bk1 = sh.sbk == 1;
    bool vec = (ak1 ? sh.sam : sh.sak) % 4 == 0 && (bk1 ? sh.sbn : sh.sbk) % 4 == 0 && (long long)sh.M * std::max(sh.sam, sh.sak) < (1ll << 31);
    for (int i = 0; i < nprob && vec; ++i)
        for (int q = 0; q < probs[i].npairs; ++q)
            vec = vec && (reinterpret_cast<uintptr_t>(probs[i].A[q]) & 15) == 0 && (reinterpret_cast<uintptr_t>(probs[i].B[q]) & 15) == 0;
    if (vec) {
        if (ak1 && bk1) hipLaunchKernelGGL((gemm_vec_kernel<true, true>), grid, dim3(256), 0, s, g);
        else if (ak1) hipLaunchKernelGGL((gemm_vec_kernel<true, false>), grid, dim3(256), 0, s, g);
        else if (bk1) hipLaunchKernelGGL((gemm_vec_kernel<false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_vec_kernel<false, false>), grid, dim3(256), 0, s, g);
    } else if (ak1 && bk1) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (ak1) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (bk1) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, s, g);
    int rc;
    if ((rc = check("angle gemm"))) return rc;
    if (ksplits > 1 || ones_col) {
        ReduceBatch r;
        for (int i = 0; i < nprob; ++i) { r.C[i] = probs[i].C; r.colsum[i] = colsum ? colsum[i] : nullptr; }
        r.M = sh.M; r.N = sh.N; r.ldc = sh.ldc; r.ksplits = ksplits; r.ones_col = ones_col; r.beta = beta; r.partial = partial;
        const long long total = (long long)sh.M * sh.N;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)std::min<long long>((total + AT - 1) / AT, 2048), 1, (unsigned)nprob),
                           dim3(AT), 0, s, r);
        return check("angle gemm reduce");
    }
    return MLLP_OK;
}

int launch_wcolsum(hipStream_t s, int K, int F, const float* A1, const float* b1, const float* A2, const float* b2, float beta,
                   float* out, float* partial) {
    const int chunks = (K + WC_ROWS - 1) / WC_ROWS;
    hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)chunks), dim3(AT), 0, s, K, F, A1, b1, A2, b2, partial);
    ReduceBatch r{};
    r.C[0] = out; r.M = F; r.N = 1; r.ldc = 1; r.ksplits = chunks; r.ones_col = 0; r.beta = beta; r.partial = partial;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((F + AT - 1) / AT), 1, 1), dim3(AT), 0, s, r);
    return check("angle wcolsum");
}

// =================================================================================================== attention
constexpr int MODE_FWD = 0, MODE_BQ = 1, MODE_BKV = 2;
struct AttnArgs {
    const float *Y1, *Y2;      // rows of the Y nodes   FWD: Q, -     BQ: Q, dO     BKV: K, V
    const float *X1, *X2;      // rows of the X nodes   FWD: K, -     BQ: K, V      BKV: Q, dO
    const float *W1, *W2;      // accumulated rows      FWD: V, -     BQ: K, -      BKV: dO (-> dV), Q (-> dK)
    const float* cos;          // [N][N]
    const float* we;           // [F]
    const float *qe, *m, *inv, *u, *D;     // per node (written by FWD / the row kernels; read by BQ / BKV)
    float *part1, *part2;      // [ranges][N][F]
    float* stats;              // [ranges][N][4]  FWD: {m, l, u}   BQ: {r}
    float* qe_out;             // FWD: [N]
    int N, xb_per_range;
    float scale;
};

// rows row0 .. row0 + 15 of M as MFMA operands of a dot product over the features: lane l holds row l & 15, features
// 16 j + 4 (l >> 4) .. + 3 -- MFMA (j, c) then multiplies feature 16 j + 4 (l >> 4) + c of both operands
template <int F>
__device__ __forceinline__ void load_frag(f4 (&fr)[F / 16], const float* __restrict__ M, int row0, int N, int lane) {
    const int row = row0 + (lane & 15);
    const float* p = M + (size_t)row * F + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < F / 16; ++j) fr[j] = row < N ? *reinterpret_cast<const f4*>(p + 16 * j) : f4zero();
}
// The 16 rows of an X block live in LDS as a tile of row stride F + 4 floats: both read patterns below are conflict-free.
// T[x][y] = X_x . Y_y:  lane l, register r  <->  x = 4 (l >> 4) + r,  y = l & 15
template <int F>
__device__ __forceinline__ f4 tile_dot(const float* __restrict__ tile, const f4 (&yb)[F / 16], int lane) {
    const float* p = tile + (lane & 15) * (F + 4) + 4 * (lane >> 4);
    // four independent chains: with two, a dependent MFMA issued every ~53 cycles instead of 32 (measured: the dot
    // products ran at 0.6 of the rate of the 16-chain accumulation)
    f4 t0 = f4zero(), t1 = f4zero(), t2 = f4zero(), t3 = f4zero();
    // The four chains must stay INTERLEAVED: left alone, the scheduler ran two chains at a time (to shorten the live range of
    // the operand rows), and a dependent MFMA every second slot issues every ~57 cycles instead of 32 (cycle stamps:
    // profiles/r04_angle_cycles.txt).  Hence: all reads of the tile row first (they return in order), then the MFMAs with a
    // scheduling barrier behind every group of four.
    f4 xa[F / 16];
#pragma unroll
    for (int j = 0; j < F / 16; ++j) xa[j] = *reinterpret_cast<const f4*>(p + 16 * j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < F / 16; ++j) {
        t0 = mfma4(xa[j][0], yb[j][0], t0);
        t1 = mfma4(xa[j][1], yb[j][1], t1);
        t2 = mfma4(xa[j][2], yb[j][2], t2);
        t3 = mfma4(xa[j][3], yb[j][3], t3);
        __builtin_amdgcn_sched_barrier(0);
    }
    return (t0 + t1) + (t2 + t3);
}
// acc[f][y] += sum_x W[x][f] t[x][y]:  MFMA r takes x = 4 (l >> 4) + r as its k slot; the A operand of lane (g, i) is
// W[x][64 u + 4 i + c] (one 16-byte read gives c = 0..3, i.e. the four tiles (u, c)); tile (u, c), register rr of lane (g, y)
// then holds f = 64 u + 16 g + 4 rr + c.
template <int F>
__device__ __forceinline__ void accumulate(f4 (&acc)[(F + 63) / 64 * 4], const float* __restrict__ tile, int lane, const f4& t) {
    constexpr int FU = (F + 63) / 64;
    const int i4 = 4 * (lane & 15);
    f4 w[2][FU];                               // the reads of k slot r + 1 are in flight while slot r multiplies
    auto rd = [&](int r, f4 (&d)[FU]) {
        const float* p = tile + (4 * (lane >> 4) + r) * (F + 4) + i4;
#pragma unroll
        for (int u = 0; u < FU; ++u) d[u] = 64 * u + i4 < F ? *reinterpret_cast<const f4*>(p + 64 * u) : f4zero();
    };
    rd(0, w[0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < 3) rd(r + 1, w[(r + 1) & 1]);
#pragma unroll
        for (int u = 0; u < FU; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[4 * u + c] = mfma4(w[r & 1][u][c], t[r], acc[4 * u + c]);
    }
}
template <int F>
__device__ __forceinline__ void store_acc(const f4 (&acc)[(F + 63) / 64 * 4], float* __restrict__ out, int y, int N, int lane) {
    constexpr int FU = (F + 63) / 64;
    if (y >= N) return;
#pragma unroll
    for (int u = 0; u < FU; ++u)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int f0 = 64 * u + 16 * (lane >> 4) + 4 * rr;
            if (f0 < F) *reinterpret_cast<f4*>(out + (size_t)y * F + f0) = f4{acc[4 * u][rr], acc[4 * u + 1][rr], acc[4 * u + 2][rr], acc[4 * u + 3][rr]};
        }
}
__device__ __forceinline__ float group_max(float v) {      // over the four lanes l, l ^ 16, l ^ 32, l ^ 48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

constexpr int ATT_W = 4;                  // wavefronts (= Y blocks of 16 nodes) per workgroup: they share the X tiles
#ifndef MLLP_ANGLE_ABL                    // timing experiments only (tools/variant_lib.sh): 1 no dot products, 2 no accumulation,
#define MLLP_ANGLE_ABL 0                  // 4 no softmax arithmetic, 8 no staging of the next tiles
#endif
constexpr int AABL = MLLP_ANGLE_ABL;
#ifdef MLLP_TIMING_BUILD
// cycle stamps of the attention kernels (timing library only; tools/angle_cycles.py): per MODE {wavefronts, prologue, dot
// products, arithmetic, accumulation, wait for the next tiles, barrier, epilogue, issue of the next block's loads, first
// tile read} summed over the wavefronts (s_memtime)
}  // namespace
__device__ unsigned long long g_angle_stamps[3][10];
namespace {
#define ANGLE_TICK(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tprev; tprev = t_; }
#else
#define ANGLE_TICK(k)
#endif

// One workgroup = 4 wavefronts = 64 Y nodes, one range of X blocks.  The two matrices of an X block (FWD: K, V; BQ: K, V;
// BKV: Q, dO -- each serves as the operand of a dot product AND of an accumulation) are staged by LDS-DMA one block ahead
// (two buffers, one barrier per block).
template <int F, int MODE>
__global__ __launch_bounds__(64 * ATT_W, MODE == MODE_FWD ? 2 : 1) void attn_kernel(AttnArgs a) {
    constexpr int FJ = F / 16, NA = (F + 63) / 64 * 4, RS = F + 4, TILE = 16 * RS;
    __shared__ __attribute__((aligned(16))) float sm[2][2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4;
#ifdef MLLP_TIMING_BUILD
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif
    const int N = a.N;
    const int y0 = (blockIdx.x * ATT_W + wv) * 16, y = y0 + (lane & 15);
    const int n_xb = (N + 15) / 16;
    const int xb_lo = blockIdx.y * a.xb_per_range, xb_hi = min(n_xb, xb_lo + a.xb_per_range);
    const bool yok = y < N;
    const float* __restrict__ cosrow = a.cos + (size_t)min(y, N - 1) * N;
    const float* __restrict__ XA = a.X1;
    const float* __restrict__ XB = MODE == MODE_FWD ? a.W1 : a.X2;

    // LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no registers, no ds_write): one instruction
    // copies one row of a tile (F / 4 lanes), wavefront w the rows w, w + 4, ...  A row behind N is row N - 1 again: finite
    // data whose products are masked (p = dz = 0 for x >= N).
    auto fetch = [&](int xb, int buf) {
        if (lane < F / 4) {
#pragma unroll
            for (int i = 0; i < 16 / ATT_W; ++i) {
                const int row = wv + ATT_W * i;
                const int off = min(xb * 16 + row, N - 1) * F + 4 * lane;        // (N F < 2^31: 32-bit address arithmetic)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(XA + off),
                                                 (__attribute__((address_space(3))) void*)(&sm[buf][0][row * RS]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(XB + off),
                                                 (__attribute__((address_space(3))) void*)(&sm[buf][1][row * RS]), 16, 0, 0);
            }
        }
    };
    // the edge attributes of the block (and, BKV, the scalars of its X nodes) also come one block ahead
    // (no guards, hence no branches: clamped addresses; whatever belongs to a node >= N is masked where it is used)
    float ncv[4], nst[MODE == MODE_BKV ? 20 : 1];
    auto fetch_small = [&](int xb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ncv[r] = cosrow[min(xb * 16 + 4 * g + r, N - 1)];     // (symmetric: A[y][x] serves both orientations)
        if constexpr (MODE == MODE_BKV) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xc = min(xb * 16 + 4 * g + r, N - 1);
                nst[r] = a.qe[xc]; nst[4 + r] = a.m[xc]; nst[8 + r] = a.inv[xc]; nst[12 + r] = a.u[xc]; nst[16 + r] = a.D[xc];
            }
        }
    };
    if (xb_lo < xb_hi) { fetch(xb_lo, 0); fetch_small(xb_lo); }

    f4 y1[FJ], y2[MODE == MODE_FWD ? 1 : FJ];
    load_frag<F>(y1, a.Y1, y0, N, lane);
    if constexpr (MODE != MODE_FWD) load_frag<F>(y2, a.Y2, y0, N, lane);
    f4 acc1[NA], acc2[MODE == MODE_BKV ? NA : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc1[i] = f4zero();
    if constexpr (MODE == MODE_BKV)
#pragma unroll
        for (int i = 0; i < NA; ++i) acc2[i] = f4zero();

    // per-Y scalars
    float qe_y = 0.f, m_y = 0.f, inv_y = 0.f, u_y = 0.f, D_y = 0.f;
    if constexpr (MODE == MODE_FWD) {          // qe = Q . we for the own rows (every lane of the row ends with the total)
        float d = 0.f;
        const float* wp = a.we + 4 * g;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const f4 wv4 = *reinterpret_cast<const f4*>(wp + 16 * j);
            d = fmaf(y1[j][0], wv4[0], d); d = fmaf(y1[j][1], wv4[1], d); d = fmaf(y1[j][2], wv4[2], d); d = fmaf(y1[j][3], wv4[3], d);
        }
        qe_y = group_sum(d);
        if (blockIdx.y == 0 && g == 0 && yok) a.qe_out[y] = qe_y;
    } else if constexpr (MODE == MODE_BQ) {
        if (yok) { qe_y = a.qe[y]; m_y = a.m[y]; inv_y = a.inv[y]; u_y = a.u[y]; D_y = a.D[y]; }
    }
    float run_m = NEG_BIG, run_l = 0.f, run_u = 0.f, run_r = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wavefront's rows of the first tiles have landed
    __syncthreads();
    ANGLE_TICK(1)

    for (int xb = xb_lo; xb < xb_hi; ++xb) {
        const int buf = (xb - xb_lo) & 1;
        const float* tA = sm[buf][0];
        const float* tB = sm[buf][1];
        float cv[4], st[MODE == MODE_BKV ? 20 : 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[r] = ncv[r];
        if constexpr (MODE == MODE_BKV)
#pragma unroll
            for (int r = 0; r < 20; ++r) st[r] = nst[r];
        if (xb + 1 < xb_hi && !(AABL & 8)) { fetch(xb + 1, buf ^ 1); fetch_small(xb + 1); }
        ANGLE_TICK(8)
#ifdef MLLP_TIMING_BUILD
        {   // how long the first read of the current tile takes behind the DMA issue (the compiler orders LDS reads behind
            // global_load_lds with vmcnt(0))
            float probe = tA[lane];
            asm volatile("s_nop 0" : "+v"(probe));
            ANGLE_TICK(9)
        }
#endif
        const int x0 = xb * 16, xg = x0 + 4 * g;
        f4 t1 = (AABL & 1) ? f4{cv[0], cv[1], cv[2], cv[3]} : tile_dot<F>(tA, y1, lane);
        f4 t2 = f4zero();
        if constexpr (MODE != MODE_FWD) t2 = (AABL & 1) ? t1 : tile_dot<F>(tB, y2, lane);
#ifdef MLLP_TIMING_BUILD
        asm volatile("s_nop 0" : "+v"(t1), "+v"(t2));      // (the stamp behind the MFMAs' results)
#endif
        ANGLE_TICK(2)
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int x = xg + r;
            ok[r] = yok && x < N && x != y;
        }
        if constexpr (MODE == MODE_FWD) {
            float L[4], lm = NEG_BIG;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                L[r] = ok[r] ? fmaf(qe_y, cv[r], t1[r]) * a.scale : NEG_BIG;
                lm = fmaxf(lm, L[r]);
            }
            if (!(AABL & 4)) lm = group_max(lm);
            // the reference of the running sums moves only when a logit exceeds it by more than 8 (p <= e^8 is harmless in
            // fp32 and alpha = p / l does not depend on the reference): the accumulators are rescaled a few times per range
            const float mn = lm > run_m + 8.0f ? lm : run_m;
            const float al = exp_acc(run_m - mn);
            run_m = mn;
            f4 p;
            float ps = 0.f, pu = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = (AABL & 4) ? L[r] : (ok[r] ? exp_acc(L[r] - mn) : 0.0f);
                ps += p[r];
                pu = fmaf(p[r], cv[r], pu);
            }
            run_l = fmaf(run_l, al, ps);
            run_u = fmaf(run_u, al, pu);
#ifdef MLLP_TIMING_BUILD
            asm volatile("s_nop 0" : "+v"(p), "+v"(run_l));
#endif
            ANGLE_TICK(3)
            if (__builtin_expect(__any(al != 1.0f), 0)) {
#pragma unroll
                for (int i = 0; i < NA; ++i) acc1[i] *= al;
            }
            if (!(AABL & 2)) accumulate<F>(acc1, tB, lane, p);
            else acc1[0] += p;
        } else {
            float qe_[4], m_[4], inv_[4], u_[4], D_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (MODE == MODE_BQ) { qe_[r] = qe_y; m_[r] = m_y; inv_[r] = inv_y; u_[r] = u_y; D_[r] = D_y; }
                else { qe_[r] = st[r]; m_[r] = st[4 + r]; inv_[r] = st[8 + r]; u_[r] = st[12 + r]; D_[r] = st[16 + r]; }
            }
            f4 p, dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float L = fmaf(qe_[r], cv[r], t1[r]) * a.scale;
                p[r] = (AABL & 4) ? L : (ok[r] ? exp_acc(L - m_[r]) * inv_[r] : 0.0f);
                const float dp = fmaf(u_[r], cv[r], t2[r]);
                dz[r] = ok[r] ? p[r] * (dp - D_[r]) * a.scale : 0.0f;      // (the scalars of a node >= N are whatever the padding holds)
                run_r = fmaf(dz[r], cv[r], run_r);
            }
#ifdef MLLP_TIMING_BUILD
            asm volatile("s_nop 0" : "+v"(p), "+v"(dz));
#endif
            ANGLE_TICK(3)
            if (AABL & 2) { acc1[0] += dz; acc1[1] += p; }
            else if constexpr (MODE == MODE_BQ) accumulate<F>(acc1, tA, lane, dz);      // dQ += K^T dz
            else {
                accumulate<F>(acc1, tB, lane, p);                                   // dV += dO^T p
                accumulate<F>(acc2, tA, lane, dz);                                  // dK += Q^T dz
            }
        }
#ifdef MLLP_TIMING_BUILD
        asm volatile("s_nop 0" : "+v"(acc1[0]), "+v"(acc1[NA - 1]));
#endif
        ANGLE_TICK(4)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // the next tiles (and this block's prefetched scalars) have landed
        ANGLE_TICK(5)
        __syncthreads();
        ANGLE_TICK(6)
    }
    // The partial accumulators leave through LDS (the tiles are free now; every wavefront has a 16 x F region of its own): in
    // registers a store instruction covers 16 rows x 64 bytes, out of LDS one whole row -- the 64-byte pieces made the
    // epilogue 13-24 % of the kernel (cycle stamps, profiles/r04_angle_cycles.txt).
    const size_t slab = (size_t)blockIdx.y * N;
    float* region = &sm[0][0][0] + wv * TILE;
    auto store_rows = [&](const f4 (&acc)[NA], float* __restrict__ out) {
#pragma unroll
        for (int u = 0; u < (F + 63) / 64; ++u)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int f0 = 64 * u + 16 * g + 4 * rr;
                if (f0 < F) *reinterpret_cast<f4*>(region + (lane & 15) * RS + f0) = f4{acc[4 * u][rr], acc[4 * u + 1][rr], acc[4 * u + 2][rr], acc[4 * u + 3][rr]};
            }
        const int c4 = min(4 * lane, F - 4);          // (F < 256: the lanes behind the row repeat its last 16 bytes)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (y0 + i < N) *reinterpret_cast<f4*>(out + (size_t)(y0 + i) * F + c4) = *reinterpret_cast<const f4*>(region + i * RS + c4);
    };
    store_rows(acc1, a.part1 + slab * F);
    if constexpr (MODE == MODE_BKV) store_rows(acc2, a.part2 + slab * F);
    if constexpr (MODE == MODE_FWD) {
        run_l = group_sum(run_l);
        run_u = group_sum(run_u);
        if (g == 0 && yok) *reinterpret_cast<f4*>(a.stats + (slab + y) * 4) = f4{run_m, run_l, run_u, 0.f};
    } else if constexpr (MODE == MODE_BQ) {
        run_r = group_sum(run_r);
        if (g == 0 && yok) a.stats[(slab + y) * 4] = run_r;
    }
#ifdef MLLP_TIMING_BUILD
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ANGLE_TICK(7)
    if (lane == 0) {
        atomicAdd(&g_angle_stamps[MODE][0], 1ull);
        for (int k = 1; k < 10; ++k) atomicAdd(&g_angle_stamps[MODE][k], tacc[k]);
    }
#endif
}

// merge of the forward ranges and the layer's epilogue: Oa = alpha V, s = sum alpha A, H = relu(Oa + s we + R).
// Four rows per workgroup, 64 lanes per row, 16 bytes per lane (F <= 256).
constexpr int RK_ROWS = AT / 64;
__global__ __launch_bounds__(AT) void fwd_combine_kernel(int N, int F, int ranges, const float* __restrict__ part,
                                                         const float* __restrict__ stats, const float* __restrict__ R,
                                                         const float* __restrict__ we, float* __restrict__ Oa,
                                                         float* __restrict__ H, float* __restrict__ m_out,
                                                         float* __restrict__ inv_out, float* __restrict__ s_out) {
    const int i = blockIdx.x * RK_ROWS + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (i >= N) return;
    float M = NEG_BIG;
    for (int s = 0; s < ranges; ++s) M = fmaxf(M, stats[((size_t)s * N + i) * 4]);
    float l = 0.f, u = 0.f;
    f4 o = f4zero();
    const bool col = c < F;
#pragma unroll 4
    for (int s = 0; s < ranges; ++s) {
        const f4 st = *reinterpret_cast<const f4*>(stats + ((size_t)s * N + i) * 4);
        const float w = exp_acc(st[0] - M);
        l = fmaf(st[1], w, l);
        u = fmaf(st[2], w, u);
        if (col) o += *reinterpret_cast<const f4*>(part + ((size_t)s * N + i) * F + c) * w;
    }
    const float inv = 1.0f / (l + 1e-16f);        // torch_geometric.utils.softmax
    const float sv = u * inv;
    if ((threadIdx.x & 63) == 0) { m_out[i] = M; inv_out[i] = inv; s_out[i] = sv; }
    if (col) {
        o *= inv;
        *reinterpret_cast<f4*>(Oa + (size_t)i * F + c) = o;
        const f4 r = *reinterpret_cast<const f4*>(R + (size_t)i * F + c), wv = *reinterpret_cast<const f4*>(we + c);
        f4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = fmaxf(fmaf(sv, wv[k], o[k] + r[k]), 0.0f);
        *reinterpret_cast<f4*>(H + (size_t)i * F + c) = h;
    }
}

// out[i][:] = sum over the ranges of part (+ r_i w, r_i = sum over the ranges of stats); blockIdx.y = 1: the second pair
__global__ __launch_bounds__(AT) void sum_ranges_kernel(int N, int F, int ranges, const float* __restrict__ part,
                                                        const float* __restrict__ stats, const float* __restrict__ w,
                                                        float* __restrict__ out, float* __restrict__ r_out,
                                                        const float* __restrict__ part_b, float* __restrict__ out_b) {
    const int i = blockIdx.x * RK_ROWS + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (i >= N) return;
    if (blockIdx.y == 1) { part = part_b; out = out_b; }
    float r = 0.f;
    if (stats) {
        for (int s = 0; s < ranges; ++s) r += stats[((size_t)s * N + i) * 4];
        if ((threadIdx.x & 63) == 0) r_out[i] = r;
    }
    if (c >= F) return;
    f4 o = f4zero();
#pragma unroll 8
    for (int s = 0; s < ranges; ++s) o += *reinterpret_cast<const f4*>(part + ((size_t)s * N + i) * F + c);
    if (stats) o += *reinterpret_cast<const f4*>(w + c) * r;
    *reinterpret_cast<f4*>(out + (size_t)i * F + c) = o;
}

// dO = dH * (H > 0) in place; u[i] = dO_i . we; D[i] = dO_i . Oa_i + u_i s_i
__global__ __launch_bounds__(AT) void relu_bwd_kernel(int F, float* __restrict__ dH, const float* __restrict__ H,
                                                      const float* __restrict__ Oa, const float* __restrict__ we,
                                                      const float* __restrict__ s, float* __restrict__ u, float* __restrict__ D) {
    __shared__ double shd[AT / 64];
    float* row = dH + (size_t)blockIdx.x * F;
    const float* h = H + (size_t)blockIdx.x * F;
    const float* oa = Oa + (size_t)blockIdx.x * F;
    // (D in double: the dz of a row must sum to zero, and they do only as far as D equals sum_j p_ij dp_ij -- the key
    // gradients are what is left of that cancellation)
    double au = 0.0, ad = 0.0;
    for (int c = threadIdx.x; c < F; c += AT) {
        const float v = h[c] > 0.0f ? row[c] : 0.0f;
        row[c] = v;
        au += (double)v * (double)we[c];
        ad += (double)v * (double)oa[c];
    }
    const double tu = block_sum_d(au, shd);
    const double td = block_sum_d(ad, shd);
    if (threadIdx.x == 0) { u[blockIdx.x] = (float)tu; D[blockIdx.x] = (float)(tu * (double)s[blockIdx.x] + td); }
}

// logits[i] = H[i, :] . w + b  for i < n_out
__global__ __launch_bounds__(AT) void fc_kernel(int F, const float* __restrict__ H, const float* __restrict__ w,
                                                const float* __restrict__ b, float* __restrict__ logits) {
    __shared__ float sh[AT / 64];
    const float* row = H + (size_t)blockIdx.x * F;
    float acc = 0.0f;
    for (int c = threadIdx.x; c < F; c += AT) acc = fmaf(row[c], w[c], acc);
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) logits[blockIdx.x] = t + b[0];
}
// dH[i, :] = dlogit_i * w  (i < n_out), 0 for the last row
__global__ __launch_bounds__(AT) void fc_bwd_kernel(int F, int64_t n_out, const float* __restrict__ dlogits,
                                                    const float* __restrict__ w, float* __restrict__ dH) {
    float* row = dH + (size_t)blockIdx.x * F;
    const float d = (int64_t)blockIdx.x < n_out ? dlogits[blockIdx.x] : 0.0f;
    for (int c = threadIdx.x; c < F; c += AT) row[c] = d * w[c];
}
__global__ __launch_bounds__(AT) void sum_kernel(int64_t n, const float* __restrict__ v, float* __restrict__ out) {
    __shared__ float sh[AT / 64];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += AT) acc += v[i];
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = t;
}

// flat parameters in PyG state_dict order (mllp_amd/angle.py::AngleModel): per conv lin_key {W [F,C], b [F]}, lin_query,
// lin_value, lin_edge {W [F,1]}, lin_skip {W, b}; then fc {W [1,F], b [1]}
struct ConvP {
    const float *Wk, *bk, *Wq, *bq, *Wv, *bv, *we, *Ws, *bs;
};
struct ConvG {
    float *Wk, *bk, *Wq, *bq, *Wv, *bv, *we, *Ws, *bs;
};
int64_t conv_size(int C, int F) { return (int64_t)4 * F * C + 5 * F; }
template <class P, class T>
P conv_at(T* base, int C, int F) {
    P p;
    T* q = base;
    p.Wk = q; q += (int64_t)F * C; p.bk = q; q += F;
    p.Wq = q; q += (int64_t)F * C; p.bq = q; q += F;
    p.Wv = q; q += (int64_t)F * C; p.bv = q; q += F;
    p.we = q; q += F;
    p.Ws = q; q += (int64_t)F * C; p.bs = q;
    return p;
}

// ranges of X blocks per group of 64 Y nodes: one workgroup per CU (the backward kernels hold 320-430 registers; the ~10 us
// a workgroup spends before its first and after its last block are paid once; two forward workgroups per CU -- 228
// registers allow it -- measured 55 -> 52 us but doubled the merge of the ranges: no gain); never more ranges than X
// blocks, at most 32
int attn_ranges(int64_t N, bool fwd = true) {
    (void)fwd;
    const int64_t nb = (N + 15) / 16, yg = (nb + ATT_W - 1) / ATT_W;
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nb, 32), 256 / std::min<int64_t>(yg, 256)));
}
int gemm_ksplits(int64_t K) { return (int)std::max<int64_t>(1, std::min<int64_t>(32, K / 256)); }

// per layer: Q, K, V, R, Oa, H [N, F]; m, inv, s, qe [N]
struct LayerWs {
    float *Q, *K, *V, *R, *Oa, *H, *m, *inv, *s, *qe;
};
struct AngleWs {
    LayerWs L[3];
    float *dA, *dB;            // [N, F] gradient of a layer's output / of its input
    float *dQ, *dK, *dV;       // [N, F]
    float *u, *r, *D;          // [N]
    float *part1, *part2;      // [ranges][N][F]
    float* stats;              // [ranges][N][4]
    float* gpart;              // [4][ksplits][F][F + 1]
};
int64_t up(int64_t x) { return (x + 63) & ~int64_t(63); }
int64_t angle_ws_floats(int64_t N, int F) {
    const int64_t R = attn_ranges(N), ks = gemm_ksplits(N);
    return 3 * (6 * up(N * F) + 4 * up(N)) + 5 * up(N * F) + 3 * up(N) + 2 * up(R * N * F) + up(R * N * 4) + up(4 * ks * F * (F + 1));
}
AngleWs angle_carve(float* base, int64_t N, int F) {
    const int64_t R = attn_ranges(N);
    AngleWs w;
    float* p = base;
    for (int l = 0; l < 3; ++l) {
        LayerWs& L = w.L[l];
        L.Q = p; p += up(N * F); L.K = p; p += up(N * F); L.V = p; p += up(N * F); L.R = p; p += up(N * F);
        L.Oa = p; p += up(N * F); L.H = p; p += up(N * F);
        L.m = p; p += up(N); L.inv = p; p += up(N); L.s = p; p += up(N); L.qe = p; p += up(N);
    }
    w.dA = p; p += up(N * F); w.dB = p; p += up(N * F);
    w.dQ = p; p += up(N * F); w.dK = p; p += up(N * F); w.dV = p; p += up(N * F);
    w.u = p; p += up(N); w.r = p; p += up(N); w.D = p; p += up(N);
    w.part1 = p; p += up(R * N * F); w.part2 = p; p += up(R * N * F);
    w.stats = p; p += up(R * N * 4);
    w.gpart = p;
    return w;
}

template <int MODE>
int launch_attn(hipStream_t s, int F, const AttnArgs& a) {
    const int nb = (a.N + 15) / 16;
    const dim3 grid((unsigned)((nb + ATT_W - 1) / ATT_W), (unsigned)((nb + a.xb_per_range - 1) / a.xb_per_range)), block(64 * ATT_W);
    switch (F) {
        case 16: hipLaunchKernelGGL((attn_kernel<16, MODE>), grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL((attn_kernel<32, MODE>), grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL((attn_kernel<64, MODE>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((attn_kernel<128, MODE>), grid, block, 0, s, a); break;
        case 256: hipLaunchKernelGGL((attn_kernel<256, MODE>), grid, block, 0, s, a); break;
        default: return fail(MLLP_EINVAL, "AngleModel: feat_dim must be 16, 32, 64, 128 or 256");
    }
    return check("angle attention");
}
AttnArgs attn_base(int64_t N, int F, const float* cos, const ConvP& p, const LayerWs& L, const AngleWs& w, bool fwd) {
    AttnArgs a{};
    a.cos = cos; a.we = p.we;
    a.qe = L.qe; a.m = L.m; a.inv = L.inv; a.u = w.u; a.D = w.D;
    a.part1 = w.part1; a.part2 = w.part2; a.stats = w.stats; a.qe_out = L.qe;
    a.N = (int)N;
    const int nb = (int)((N + 15) / 16), R = attn_ranges(N, fwd);
    a.xb_per_range = (nb + R - 1) / R;
    a.scale = 1.0f / sqrtf((float)F);
    return a;
}

int conv_forward(hipStream_t s, int64_t N, int C, int F, const float* A, const float* X, const ConvP& p, const LayerWs& L,
                 const AngleWs& w) {
    int rc;
    // Q, K, V, R = X W^T + b in one launch
    GemmProb pr[4] = {};
    const float* Ws_[4] = {p.Wq, p.Wk, p.Wv, p.Ws};
    const float* bs_[4] = {p.bq, p.bk, p.bv, p.bs};
    float* out[4] = {L.Q, L.K, L.V, L.R};
    for (int i = 0; i < 4; ++i) { pr[i].A[0] = X; pr[i].B[0] = Ws_[i]; pr[i].C = out[i]; pr[i].bias = bs_[i]; pr[i].npairs = 1; }
    if ((rc = launch_gemm(s, GemmShape{(int)N, F, C, C, 1, C, 1, F}, 4, pr, 0, 1, 0.0f, nullptr, nullptr))) return rc;
    AttnArgs a = attn_base(N, F, A, p, L, w, true);
    a.Y1 = L.Q; a.X1 = L.K; a.W1 = L.V;
    if ((rc = launch_attn<MODE_FWD>(s, F, a))) return rc;
    const int ranges = (int)(((N + 15) / 16 + a.xb_per_range - 1) / a.xb_per_range);
    hipLaunchKernelGGL(fwd_combine_kernel, dim3((unsigned)((N + RK_ROWS - 1) / RK_ROWS)), dim3(AT), 0, s, (int)N, F, ranges, w.part1, w.stats, L.R, p.we, L.Oa,
                       L.H, L.m, L.inv, L.s);
    return check("angle fwd_combine");
}

// dH (in: gradient of the layer's output, overwritten with dO) -> parameter gradients (accumulated when acc) and, if dX,
// the gradient of the layer's input
int conv_backward(hipStream_t s, int64_t N, int C, int F, const float* A, const float* X, const ConvP& p, const LayerWs& L,
                  const AngleWs& w, float* dH, float* dX, const ConvG& g, bool acc) {
    int rc;
    const float beta = acc ? 1.0f : 0.0f;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, dH, L.H, L.Oa, p.we, L.s, w.u, w.D);   // dO, u, D
    if ((rc = check("angle relu_bwd"))) return rc;
    AttnArgs a = attn_base(N, F, A, p, L, w, false);
    const int ranges = (int)(((N + 15) / 16 + a.xb_per_range - 1) / a.xb_per_range);
    // dQ = dz K + r we^T,  r = sum dz A
    a.Y1 = L.Q; a.Y2 = dH; a.X1 = L.K; a.X2 = L.V; a.W1 = L.K;
    if ((rc = launch_attn<MODE_BQ>(s, F, a))) return rc;
    hipLaunchKernelGGL(sum_ranges_kernel, dim3((unsigned)((N + RK_ROWS - 1) / RK_ROWS)), dim3(AT), 0, s, (int)N, F, ranges, w.part1, w.stats, p.we, w.dQ, w.r,
                       (const float*)nullptr, (float*)nullptr);
    // dV = alpha^T dO,  dK = dz^T Q
    a.Y1 = L.K; a.Y2 = L.V; a.X1 = L.Q; a.X2 = dH; a.W1 = dH; a.W2 = L.Q;
    if ((rc = launch_attn<MODE_BKV>(s, F, a))) return rc;
    hipLaunchKernelGGL(sum_ranges_kernel, dim3((unsigned)((N + RK_ROWS - 1) / RK_ROWS), 2), dim3(AT), 0, s, (int)N, F, ranges, w.part1, (const float*)nullptr,
                       (const float*)nullptr, w.dV, (float*)nullptr, w.part2, w.dK);
    if ((rc = check("angle sum_ranges"))) return rc;
    // (K splits of the weight-gradient GEMM: 7 at N = 1 877 = 560 workgroups; 3 splits -- one round of workgroups -- measured
    // 44 us against 32, 14 splits 31.5 with a reduction twice as long)
    const int ks = gemm_ksplits(N);
    {   // dW = dY^T X and db = 1^T dY of the four projections in one launch (K = N, split; the ones column gives db)
        GemmProb pr[4] = {};
        const float* dY[4] = {w.dQ, w.dK, w.dV, dH};
        float* dW[4] = {g.Wq, g.Wk, g.Wv, g.Ws};
        float* db[4] = {g.bq, g.bk, g.bv, g.bs};
        for (int i = 0; i < 4; ++i) { pr[i].A[0] = dY[i]; pr[i].B[0] = X; pr[i].C = dW[i]; pr[i].npairs = 1; }
        if ((rc = launch_gemm(s, GemmShape{F, C + 1, (int)N, 1, F, 1, C, C}, 4, pr, 1, ks, beta, w.gpart, db))) return rc;
    }
    if ((rc = launch_wcolsum(s, (int)N, F, dH, L.s, L.Q, w.r, beta, g.we, w.gpart))) return rc;       // dwe = dO^T s + Q^T r
    if (dX) {   // dX = dO Ws + dQ Wq + dK Wk + dV Wv
        GemmProb pr[1] = {};
        pr[0].A[0] = dH; pr[0].B[0] = p.Ws; pr[0].A[1] = w.dQ; pr[0].B[1] = p.Wq; pr[0].A[2] = w.dK; pr[0].B[2] = p.Wk;
        pr[0].A[3] = w.dV; pr[0].B[3] = p.Wv; pr[0].C = dX; pr[0].npairs = 4;
        // (few output tiles: the feature range of every pair is cut in up to four, partial sums in the idle range buffer)
        const int ksx = std::min(std::min(4, attn_ranges(N)), std::max(1, F / 32));
        if ((rc = launch_gemm(s, GemmShape{(int)N, C, F, F, 1, 1, C, C}, 1, pr, 0, ksx, 0.0f, w.part1, nullptr))) return rc;
    }
    return MLLP_OK;
}

bool feat_ok(int F) { return F == 16 || F == 32 || F == 64 || F == 128 || F == 256; }

}  // namespace
}  // namespace mllp
#ifdef MLLP_TIMING_BUILD
// timing library only: read (and clear) the attention kernels' cycle stamps, [3 modes][10]
extern "C" int mllp_debug_angle_stamps(unsigned long long* host) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(mllp::g_angle_stamps), sizeof(unsigned long long) * 30) != hipSuccess) return -1;
    static const unsigned long long zero[30] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(mllp::g_angle_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
namespace mllp {
namespace {

}  // namespace
}  // namespace mllp

using namespace mllp;

#define REQUIRE(cond, msg) \
    do {                   \
        if (!(cond)) return fail(MLLP_EINVAL, msg); \
    } while (0)

extern "C" int mllp_angle_num_params(int feat_dim, int64_t* out) {
    REQUIRE(out && feat_dim >= 1, "bad argument");
    *out = conv_size(2, feat_dim) + 2 * conv_size(feat_dim, feat_dim) + feat_dim + 1;
    return MLLP_OK;
}

extern "C" int mllp_angle_workspace_floats(int64_t n_nodes, int feat_dim, int64_t* out) {
    REQUIRE(out && n_nodes >= 2, "bad argument");
    REQUIRE(feat_ok(feat_dim), "AngleModel: feat_dim must be 16, 32, 64, 128 or 256");
    REQUIRE(n_nodes <= 46340, "n_nodes^2 must fit 32-bit indexing");
    *out = angle_ws_floats(n_nodes, feat_dim);
    return MLLP_OK;
}

extern "C" int mllp_angle_forward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                                  float* d_ws, float* d_logits, void* stream) {
    REQUIRE(d_cos && d_x && d_params && d_ws && d_logits, "null argument");
    REQUIRE(n_nodes >= 2 && n_nodes <= 46340, "bad size");
    REQUIRE(feat_ok(feat_dim), "AngleModel: feat_dim must be 16, 32, 64, 128 or 256");
    const int64_t N = n_nodes;
    const int F = feat_dim;
    hipStream_t s = (hipStream_t)stream;
    const AngleWs w = angle_carve(d_ws, N, F);
    const ConvP p1 = conv_at<ConvP>(d_params, 2, F);
    const ConvP p2 = conv_at<ConvP>(d_params + conv_size(2, F), F, F);
    const float* fcw = d_params + conv_size(2, F) + 2 * conv_size(F, F);
    int rc;
    // linear_program_methods.py:196-198: gconv1, gconv2, gconv2 (again)
    if ((rc = conv_forward(s, N, 2, F, d_cos, d_x, p1, w.L[0], w))) return rc;
    if ((rc = conv_forward(s, N, F, F, d_cos, w.L[0].H, p2, w.L[1], w))) return rc;
    if ((rc = conv_forward(s, N, F, F, d_cos, w.L[1].H, p2, w.L[2], w))) return rc;
    // :199-200 fc, all nodes but the last
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)(N - 1)), dim3(AT), 0, s, F, w.L[2].H, fcw, fcw + F, d_logits);
    return check("angle fc");
}

extern "C" int mllp_angle_backward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                                   float* d_ws, const float* d_dlogits, float* d_grads, void* stream) {
    REQUIRE(d_cos && d_x && d_params && d_ws && d_dlogits && d_grads, "null argument");
    REQUIRE(n_nodes >= 2 && n_nodes <= 46340, "bad size");
    REQUIRE(feat_ok(feat_dim), "AngleModel: feat_dim must be 16, 32, 64, 128 or 256");
    const int64_t N = n_nodes;
    const int F = feat_dim;
    hipStream_t s = (hipStream_t)stream;
    const AngleWs w = angle_carve(d_ws, N, F);
    const int64_t o2 = conv_size(2, F), o3 = o2 + conv_size(F, F), ofc = o3 + conv_size(F, F);
    const ConvP p1 = conv_at<ConvP>(d_params, 2, F);
    const ConvP p2 = conv_at<ConvP>(d_params + o2, F, F);
    const ConvG g1 = conv_at<ConvG>(d_grads, 2, F);
    const ConvG g2 = conv_at<ConvG>(d_grads + o2, F, F);
    const float* fcw = d_params + ofc;
    int rc;
    // gconv3 is never called (reference :198 applies gconv2 twice): its gradient is zero
    MLLP_HIP_TRY(hipMemsetAsync(d_grads + o3, 0, (size_t)conv_size(F, F) * sizeof(float), s));
    // fc: dW = sum_i dlogit_i H3_i, db = sum_i dlogit_i, dH3 = dlogit w (last node: 0)
    if ((rc = launch_wcolsum(s, (int)(N - 1), F, w.L[2].H, d_dlogits, nullptr, nullptr, 0.0f, d_grads + ofc, w.gpart))) return rc;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(AT), 0, s, N - 1, d_dlogits, d_grads + ofc + F);
    hipLaunchKernelGGL(fc_bwd_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, N - 1, d_dlogits, fcw, w.dA);
    if ((rc = check("angle fc_bwd"))) return rc;
    // third layer (gconv2, second use) -> dH2 in dB; second layer (gconv2, first use, accumulates) -> dH1 in dA; first layer
    if ((rc = conv_backward(s, N, F, F, d_cos, w.L[1].H, p2, w.L[2], w, w.dA, w.dB, g2, false))) return rc;
    if ((rc = conv_backward(s, N, F, F, d_cos, w.L[0].H, p2, w.L[1], w, w.dB, w.dA, g2, true))) return rc;
    return conv_backward(s, N, 2, F, d_cos, d_x, p1, w.L[0], w, w.dA, nullptr, g1, false);
}
