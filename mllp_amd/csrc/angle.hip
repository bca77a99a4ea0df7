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
// so the hot operations are N x N x F GEMMs (rocBLAS sgemm, loaded on first use: plain library GEMMs, exact fp32) and the fused
// row kernels below (bias / row dot, masked softmax with the edge term, softmax backward with the edge reductions).
// A [N, N] is the dense cosine matrix (diagonal ignored: the graph has no self loops).  The backward pass is hand-derived
// (tests/test_angle.py checks it against fp64 autograd of the oracle's literal TransformerConv on the edge list).
#include <dlfcn.h>

#include <cmath>

#include "device_utils.h"
#include "internal.h"

namespace mllp {
namespace {

constexpr int AT = 256;     // threads of the row kernels

// rocBLAS is loaded on first use (dlopen): the sparse hot path of this library neither links nor loads it.  The four
// entry points and the enum values below are rocBLAS' public C API (rocblas/internal/rocblas-types.h).
struct Blas {
    void* handle = nullptr;
    int (*set_stream)(void*, hipStream_t) = nullptr;
    int (*sgemm)(void*, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*,
                 int) = nullptr;
};
constexpr int ROCBLAS_OP_N = 111, ROCBLAS_OP_T = 112, ROCBLAS_ATOMICS_NOT_ALLOWED = 0;

const Blas& blas() {
    static Blas b = [] {
        Blas x;
        // the copy that is already in the process first (torch loads its own under a versioned SONAME): two rocBLAS
        // instances in one process would each keep their own kernels and handles
        void* lib = nullptr;
        for (const char* name : {"librocblas.so.5", "librocblas.so.4", "librocblas.so"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
        if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return x;
        auto create = reinterpret_cast<int (*)(void**)>(dlsym(lib, "rocblas_create_handle"));
        auto atomics = reinterpret_cast<int (*)(void*, int)>(dlsym(lib, "rocblas_set_atomics_mode"));
        x.set_stream = reinterpret_cast<int (*)(void*, hipStream_t)>(dlsym(lib, "rocblas_set_stream"));
        x.sgemm = reinterpret_cast<decltype(x.sgemm)>(dlsym(lib, "rocblas_sgemm"));
        if (!create || !atomics || !x.set_stream || !x.sgemm || create(&x.handle) != 0) {
            x.handle = nullptr;
            return x;
        }
        atomics(x.handle, ROCBLAS_ATOMICS_NOT_ALLOWED);      // deterministic sums (no split-K atomics)
        return x;
    }();
    return b;
}

// row-major C[M, N] = alpha * op(A) * op(B) + beta * C   (op(A): M x K, op(B): K x N)
int gemm_rm(hipStream_t s, bool ta, bool tb, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
            const float* B, int64_t ldb, float beta, float* C, int64_t ldc) {
    if (M == 0 || N == 0) return MLLP_OK;
    const Blas& b = blas();
    if (!b.handle) return fail(MLLP_EHIP, "rocBLAS (librocblas.so) could not be loaded: AngleModel needs it for its dense GEMMs");
    b.set_stream(b.handle, s);
    // a row-major matrix is its transpose in column-major storage: C^T = op(B)^T op(A)^T
    const int st = b.sgemm(b.handle, tb ? ROCBLAS_OP_T : ROCBLAS_OP_N, ta ? ROCBLAS_OP_T : ROCBLAS_OP_N, (int)N, (int)M, (int)K,
                           &alpha, B, (int)ldb, A, (int)lda, &beta, C, (int)ldc);
    return st == 0 ? MLLP_OK : fail(MLLP_EHIP, "rocblas_sgemm failed");
}

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
__device__ __forceinline__ float block_max(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = sh[0];
    for (int w = 1; w < AT / 64; ++w) t = fmaxf(t, sh[w]);
    return t;
}

// C[r, :] += b ; optionally d[r] = C[r, :] . w      (one workgroup per row)
__global__ __launch_bounds__(AT) void bias_dot_kernel(int F, float* __restrict__ C, const float* __restrict__ b,
                                                      const float* __restrict__ w, float* __restrict__ d) {
    __shared__ float sh[AT / 64];
    float* row = C + (size_t)blockIdx.x * F;
    float acc = 0.0f;
    for (int c = threadIdx.x; c < F; c += AT) {
        const float v = row[c] + (b ? b[c] : 0.0f);
        row[c] = v;
        if (w) acc = fmaf(v, w[c], acc);
    }
    if (d) {
        const float t = block_sum(acc, sh);
        if (threadIdx.x == 0) d[blockIdx.x] = t;
    }
}

// Z[i, :] (= Q_i . K_j) -> alpha[i, :] in place, s[i] = sum_j alpha_ij A_ij     (row i: target, diagonal masked)
__global__ __launch_bounds__(AT) void softmax_edge_kernel(int N, float scale, float* __restrict__ Z,
                                                          const float* __restrict__ A, const float* __restrict__ qe,
                                                          float* __restrict__ s_out) {
    __shared__ float sh[AT / 64];
    const int i = blockIdx.x;
    float* z = Z + (size_t)i * N;
    const float* a = A + (size_t)i * N;
    const float q = qe[i];
    float m = -3.0e38f;
    for (int j = threadIdx.x; j < N; j += AT)
        if (j != i) m = fmaxf(m, fmaf(q, a[j], z[j]) * scale);
    m = block_max(m, sh);
    float sum = 0.0f;
    for (int j = threadIdx.x; j < N; j += AT) {
        float e = 0.0f;
        if (j != i) e = exp_acc(fmaf(q, a[j], z[j]) * scale - m);
        z[j] = e;
        sum += e;
    }
    sum = block_sum(sum, sh);
    const float inv = 1.0f / (sum + 1e-16f);        // torch_geometric.utils.softmax
    float sa = 0.0f;
    for (int j = threadIdx.x; j < N; j += AT) {
        const float al = z[j] * inv;
        z[j] = al;
        sa = fmaf(al, a[j], sa);                    // alpha_ii = 0
    }
    sa = block_sum(sa, sh);
    if (threadIdx.x == 0) s_out[i] = sa;
}

// H[i, :] = relu(H[i, :] + bs + s_i we)
__global__ __launch_bounds__(AT) void out_relu_kernel(int F, float* __restrict__ H, const float* __restrict__ bs,
                                                      const float* __restrict__ we, const float* __restrict__ s) {
    float* row = H + (size_t)blockIdx.x * F;
    const float si = s[blockIdx.x];
    for (int c = threadIdx.x; c < F; c += AT) row[c] = fmaxf(fmaf(si, we[c], row[c] + bs[c]), 0.0f);
}

// dO = dH * (H > 0) in place; u[i] = dO_i . we
__global__ __launch_bounds__(AT) void relu_bwd_kernel(int F, float* __restrict__ dH, const float* __restrict__ H,
                                                      const float* __restrict__ we, float* __restrict__ u) {
    __shared__ float sh[AT / 64];
    float* row = dH + (size_t)blockIdx.x * F;
    const float* h = H + (size_t)blockIdx.x * F;
    float acc = 0.0f;
    for (int c = threadIdx.x; c < F; c += AT) {
        const float v = h[c] > 0.0f ? row[c] : 0.0f;
        row[c] = v;
        acc = fmaf(v, we[c], acc);
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) u[blockIdx.x] = t;
}

// G[i, :] (= dO_i . V_j) -> dZ[i, :] in place; r[i] = sum_j dZ_ij A_ij
//   g_ij = G_ij + u_i A_ij ,  dL_ij = alpha_ij (g_ij - sum_k alpha_ik g_ik) ,  dZ = dL * scale
__global__ __launch_bounds__(AT) void softmax_bwd_kernel(int N, float scale, float* __restrict__ G,
                                                         const float* __restrict__ alpha, const float* __restrict__ A,
                                                         const float* __restrict__ u, float* __restrict__ r_out) {
    __shared__ float sh[AT / 64];
    const int i = blockIdx.x;
    float* g = G + (size_t)i * N;
    const float* al = alpha + (size_t)i * N;
    const float* a = A + (size_t)i * N;
    const float ui = u[i];
    float t = 0.0f;
    for (int j = threadIdx.x; j < N; j += AT) t = fmaf(al[j], fmaf(ui, a[j], g[j]), t);
    t = block_sum(t, sh);
    float ra = 0.0f;
    for (int j = threadIdx.x; j < N; j += AT) {
        const float dz = al[j] * (fmaf(ui, a[j], g[j]) - t) * scale;
        g[j] = dz;
        ra = fmaf(dz, a[j], ra);
    }
    ra = block_sum(ra, sh);
    if (threadIdx.x == 0) r_out[i] = ra;
}

// C[i, :] += r_i * w
__global__ __launch_bounds__(AT) void add_outer_kernel(int F, float* __restrict__ C, const float* __restrict__ r,
                                                       const float* __restrict__ w) {
    float* row = C + (size_t)blockIdx.x * F;
    const float ri = r[blockIdx.x];
    for (int c = threadIdx.x; c < F; c += AT) row[c] = fmaf(ri, w[c], row[c]);
}

__global__ __launch_bounds__(AT) void fill_kernel(int64_t n, float v, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) out[i] = v;
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

int check(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, what);
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

// per layer: Q, K, V, H [N, F]; alpha [N, N]; s [N]
struct LayerWs {
    float *Q, *K, *V, *H, *alpha, *s;
};
struct AngleWs {
    LayerWs L[3];
    float *dA, *dB;            // [N, F] gradient of a layer's output / of its input
    float *dQ, *dK, *dV;       // [N, F]
    float *G;                  // [N, N]
    float *u, *r, *qe, *ones;  // [N]
};
int64_t up(int64_t x) { return (x + 63) & ~int64_t(63); }
int64_t angle_ws_floats(int64_t N, int F) {
    return 3 * (4 * up(N * F) + up(N * N) + up(N)) + 5 * up(N * F) + up(N * N) + 4 * up(N);
}
AngleWs angle_carve(float* base, int64_t N, int F) {
    AngleWs w;
    float* p = base;
    for (int l = 0; l < 3; ++l) {
        w.L[l].Q = p; p += up(N * F); w.L[l].K = p; p += up(N * F); w.L[l].V = p; p += up(N * F);
        w.L[l].H = p; p += up(N * F); w.L[l].alpha = p; p += up(N * N); w.L[l].s = p; p += up(N);
    }
    w.dA = p; p += up(N * F); w.dB = p; p += up(N * F);
    w.dQ = p; p += up(N * F); w.dK = p; p += up(N * F); w.dV = p; p += up(N * F);
    w.G = p; p += up(N * N);
    w.u = p; p += up(N); w.r = p; p += up(N); w.qe = p; p += up(N); w.ones = p;
    return w;
}

int conv_forward(hipStream_t s, int64_t N, int C, int F, const float* A, const float* X, const ConvP& p, const LayerWs& L,
                 float* qe) {
    int rc;
    const float scale = 1.0f / sqrtf((float)F);
    if ((rc = gemm_rm(s, false, true, N, F, C, 1.0f, X, C, p.Wq, C, 0.0f, L.Q, F))) return rc;
    if ((rc = gemm_rm(s, false, true, N, F, C, 1.0f, X, C, p.Wk, C, 0.0f, L.K, F))) return rc;
    if ((rc = gemm_rm(s, false, true, N, F, C, 1.0f, X, C, p.Wv, C, 0.0f, L.V, F))) return rc;
    if ((rc = gemm_rm(s, false, true, N, F, C, 1.0f, X, C, p.Ws, C, 0.0f, L.H, F))) return rc;       // R, completed below
    hipLaunchKernelGGL(bias_dot_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, L.Q, p.bq, p.we, qe);
    hipLaunchKernelGGL(bias_dot_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, L.K, p.bk, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(bias_dot_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, L.V, p.bv, (const float*)nullptr, (float*)nullptr);
    if ((rc = check("angle bias"))) return rc;
    if ((rc = gemm_rm(s, false, true, N, N, F, 1.0f, L.Q, F, L.K, F, 0.0f, L.alpha, N))) return rc;   // Z = Q K^T
    hipLaunchKernelGGL(softmax_edge_kernel, dim3((unsigned)N), dim3(AT), 0, s, (int)N, scale, L.alpha, A, qe, L.s);
    if ((rc = check("angle softmax"))) return rc;
    if ((rc = gemm_rm(s, false, false, N, F, N, 1.0f, L.alpha, N, L.V, F, 1.0f, L.H, F))) return rc;  // H = R + alpha V
    hipLaunchKernelGGL(out_relu_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, L.H, p.bs, p.we, L.s);
    return check("angle out");
}

// dH (in: gradient of the layer's output, overwritten with dO) -> parameter gradients (accumulated when acc) and, if dX,
// the gradient of the layer's input
int conv_backward(hipStream_t s, int64_t N, int C, int F, const float* A, const float* X, const ConvP& p, const LayerWs& L,
                  const AngleWs& w, float* dH, float* dX, const ConvG& g, bool acc) {
    int rc;
    const float scale = 1.0f / sqrtf((float)F);
    const float beta = acc ? 1.0f : 0.0f;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, dH, L.H, p.we, w.u);    // dO, u = dO . we
    if ((rc = check("angle relu_bwd"))) return rc;
    // skip path and value path
    if ((rc = gemm_rm(s, true, false, F, C, N, 1.0f, dH, F, X, C, beta, g.Ws, C))) return rc;         // dWs = dO^T X
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, w.ones, N, dH, F, beta, g.bs, F))) return rc;     // dbs = 1^T dO
    if ((rc = gemm_rm(s, true, false, N, F, N, 1.0f, L.alpha, N, dH, F, 0.0f, w.dV, F))) return rc;   // dV = alpha^T dO
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, L.s, N, dH, F, beta, g.we, F))) return rc;      // dwe = s^T dO
    // attention weights
    if ((rc = gemm_rm(s, false, true, N, N, F, 1.0f, dH, F, L.V, F, 0.0f, w.G, N))) return rc;        // G = dO V^T
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)N), dim3(AT), 0, s, (int)N, scale, w.G, L.alpha, A, w.u, w.r);
    if ((rc = check("angle softmax_bwd"))) return rc;
    if ((rc = gemm_rm(s, false, false, N, F, N, 1.0f, w.G, N, L.K, F, 0.0f, w.dQ, F))) return rc;     // dQ = dZ K + r we^T
    hipLaunchKernelGGL(add_outer_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, w.dQ, w.r, p.we);
    if ((rc = gemm_rm(s, true, false, N, F, N, 1.0f, w.G, N, L.Q, F, 0.0f, w.dK, F))) return rc;      // dK = dZ^T Q
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, w.r, N, L.Q, F, 1.0f, g.we, F))) return rc;     // dwe += r^T Q
    // weights of the three projections
    if ((rc = gemm_rm(s, true, false, F, C, N, 1.0f, w.dQ, F, X, C, beta, g.Wq, C))) return rc;
    if ((rc = gemm_rm(s, true, false, F, C, N, 1.0f, w.dK, F, X, C, beta, g.Wk, C))) return rc;
    if ((rc = gemm_rm(s, true, false, F, C, N, 1.0f, w.dV, F, X, C, beta, g.Wv, C))) return rc;
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, w.ones, N, w.dQ, F, beta, g.bq, F))) return rc;
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, w.ones, N, w.dK, F, beta, g.bk, F))) return rc;
    if ((rc = gemm_rm(s, false, false, 1, F, N, 1.0f, w.ones, N, w.dV, F, beta, g.bv, F))) return rc;
    if (dX) {                                                                                         // dX = sum d* W*
        if ((rc = gemm_rm(s, false, false, N, C, F, 1.0f, dH, F, p.Ws, C, 0.0f, dX, C))) return rc;
        if ((rc = gemm_rm(s, false, false, N, C, F, 1.0f, w.dQ, F, p.Wq, C, 1.0f, dX, C))) return rc;
        if ((rc = gemm_rm(s, false, false, N, C, F, 1.0f, w.dK, F, p.Wk, C, 1.0f, dX, C))) return rc;
        if ((rc = gemm_rm(s, false, false, N, C, F, 1.0f, w.dV, F, p.Wv, C, 1.0f, dX, C))) return rc;
    }
    return MLLP_OK;
}

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
    REQUIRE(out && n_nodes >= 2 && feat_dim >= 1, "bad argument");
    REQUIRE(n_nodes <= 46340, "n_nodes^2 must fit 32-bit GEMM dimensions");
    *out = angle_ws_floats(n_nodes, feat_dim);
    return MLLP_OK;
}

extern "C" int mllp_angle_forward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                                  float* d_ws, float* d_logits, void* stream) {
    REQUIRE(d_cos && d_x && d_params && d_ws && d_logits, "null argument");
    REQUIRE(n_nodes >= 2 && n_nodes <= 46340 && feat_dim >= 1, "bad size");
    const int64_t N = n_nodes;
    const int F = feat_dim;
    hipStream_t s = (hipStream_t)stream;
    const AngleWs w = angle_carve(d_ws, N, F);
    const ConvP p1 = conv_at<ConvP>(d_params, 2, F);
    const ConvP p2 = conv_at<ConvP>(d_params + conv_size(2, F), F, F);
    const float* fcw = d_params + conv_size(2, F) + 2 * conv_size(F, F);
    int rc;
    // linear_program_methods.py:196-198: gconv1, gconv2, gconv2 (again)
    if ((rc = conv_forward(s, N, 2, F, d_cos, d_x, p1, w.L[0], w.qe))) return rc;
    if ((rc = conv_forward(s, N, F, F, d_cos, w.L[0].H, p2, w.L[1], w.qe))) return rc;
    if ((rc = conv_forward(s, N, F, F, d_cos, w.L[1].H, p2, w.L[2], w.qe))) return rc;
    // :199-200 fc, all nodes but the last
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)(N - 1)), dim3(AT), 0, s, F, w.L[2].H, fcw, fcw + F, d_logits);
    return check("angle fc");
}

extern "C" int mllp_angle_backward(int64_t n_nodes, int feat_dim, const float* d_cos, const float* d_x, const float* d_params,
                                   float* d_ws, const float* d_dlogits, float* d_grads, void* stream) {
    REQUIRE(d_cos && d_x && d_params && d_ws && d_dlogits && d_grads, "null argument");
    REQUIRE(n_nodes >= 2 && n_nodes <= 46340 && feat_dim >= 1, "bad size");
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
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(AT), 0, s, N, 1.0f, w.ones);
    // gconv3 is never called (reference :198 applies gconv2 twice): its gradient is zero
    MLLP_HIP_TRY(hipMemsetAsync(d_grads + o3, 0, (size_t)conv_size(F, F) * sizeof(float), s));
    // fc: dW = sum_i dlogit_i H3_i, db = sum_i dlogit_i, dH3 = dlogit w (last node: 0)
    if ((rc = gemm_rm(s, false, false, 1, F, N - 1, 1.0f, d_dlogits, N - 1, w.L[2].H, F, 0.0f, d_grads + ofc, F))) return rc;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(AT), 0, s, N - 1, d_dlogits, d_grads + ofc + F);
    hipLaunchKernelGGL(fc_bwd_kernel, dim3((unsigned)N), dim3(AT), 0, s, F, N - 1, d_dlogits, fcw, w.dA);
    if ((rc = check("angle fc_bwd"))) return rc;
    // third layer (gconv2, second use) -> dH2 in dB; second layer (gconv2, first use, accumulates) -> dH1 in dA; first layer
    if ((rc = conv_backward(s, N, F, F, d_cos, w.L[1].H, p2, w.L[2], w, w.dA, w.dB, g2, false))) return rc;
    if ((rc = conv_backward(s, N, F, F, d_cos, w.L[0].H, p2, w.L[1], w, w.dB, w.dA, g2, true))) return rc;
    return conv_backward(s, N, 2, F, d_cos, d_x, p1, w.L[0], w, w.dA, nullptr, g1, false);
}
