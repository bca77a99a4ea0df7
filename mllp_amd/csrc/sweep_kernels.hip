// sweep_kernels.hip -- the HBM-bound sparse sweeps over the constraint-matrix pattern (gfx950).
//
// Every sweep walks a destination-major CSR (of A, or of A^T) once:
//   spmm          Y_i   = sum_j a_ij X_j                                  (plain CSR SpMM, roofline kernel)
//   attn_fwd      o_i   = relu( Wv Z_i + S_i bv + u_i we + Ws x_i + bs ), Z_i = sum_j alpha_ij X_j
//                 with the online segment-softmax alpha_ij over l_ij = <q'_i, X_j> + a_ij t_i
//   attn_bwd_dst  dq'_i = sum_j dl_ij X_j, ds_i, dt_i, and the destination-side input gradient
//   attn_bwd_src  dX_j  = sum_i alpha_ij gv_i + dl_ij q'_i   (walks the OPPOSITE orientation)
// (reference call sites linear_program_methods.py:241-247, PyG
// TransformerConv.message / utils.softmax restated in oracle/pyg_restatement.py.)
//
// Mapping: one LANE per nonzero.  A lane streams (idx, val) coalesced, gathers the whole 64-byte
// source row X_j with four 16-byte loads, does the dot products and accumulations in its own
// registers, and only the per-row totals cross lanes (DPP all-reduce inside the 16-lane row, then
// the LDS crossbar for a 64-lane wave, then LDS for a workgroup).  Rows are tiered by length:
//   group tier  : 16 lanes per row, 4 rows per wavefront (short rows / throughput regime)
//   wave tier   : 64 lanes per row
//   block tier  : 256 threads per row, four per-wave partial states merged through LDS
// Tiles of the group tier are remapped so that each XCD walks a contiguous row range: the source
// rows gathered by neighbouring tiles then come out of one XCD's L2.
#include "device_utils.h"
#include "internal.h"

namespace mllp {

struct OrientDev {
    const int* __restrict__ ptr;
    const int* __restrict__ idx;
    const float* __restrict__ val;
    const int* __restrict__ rows_group;
    const int* __restrict__ rows_wave;
    const int* __restrict__ rows_block;
    int n_group, n_wave, n_block, nbA, nbB;
};

static OrientDev make_dev(const Orient& o) {
    OrientDev d;
    d.ptr = o.ptr; d.idx = o.idx; d.val = o.val;
    d.rows_group = o.rows_group; d.rows_wave = o.rows_wave; d.rows_block = o.rows_block;
    d.n_group = o.n_group; d.n_wave = o.n_wave; d.n_block = o.n_block;
    d.nbA = (o.n_group + 15) / 16;
    d.nbB = (o.n_wave + 3) / 4;
    return d;
}

// -------------------------------------------------------------------------------------------------
// generic tiered driver.  Op provides:
//   Args                         kernel arguments (POD)
//   NS                           floats of per-wave state exchanged through LDS in the block tier
//   load_row(args,row,deg)       per-row inputs (uniform over the lanes of the row)
//   edges<G,U>(args,o,beg,end,first,stride,gl)   accumulate this lane's nonzeros
//   reduce<G>()                  all-reduce the state over the G lanes
//   to_lds(float*) / merge_from_lds(const float*, nwaves)
//   epilogue(args,row,deg,gl)    called with gl in [0,16) on lanes holding the reduced state
// -------------------------------------------------------------------------------------------------
template <class Op, int UA, int UB>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(typename Op::Args args, OrientDev o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    if (b < o.nbA) {
        const int tile = xcd_tile(b, o.nbA);
        const int slot = tile * 16 + wave * 4 + (lane >> 4);
        if (slot < o.n_group) {
            const int row = o.rows_group ? o.rows_group[slot] : slot;
            const int beg = o.ptr[row], end = o.ptr[row + 1];
            Op op;
            op.load_row(args, row, end - beg);
            op.template edges<16, UA>(args, o, beg, end, 0, 16 * UA, lane & 15);
            op.template reduce<16>();
            op.epilogue(args, row, end - beg, lane & 15);
        }
    } else if (b < o.nbA + o.nbB) {
        const int slot = (b - o.nbA) * 4 + wave;
        if (slot < o.n_wave) {
            const int row = o.rows_wave[slot];
            const int beg = o.ptr[row], end = o.ptr[row + 1];
            Op op;
            op.load_row(args, row, end - beg);
            op.template edges<64, UB>(args, o, beg, end, 0, 64 * UB, lane);
            op.template reduce<64>();
            if (lane < 16) op.epilogue(args, row, end - beg, lane);
        }
    } else {
        __shared__ float sh[4 * Op::NS];
        const int row = o.rows_block[b - o.nbA - o.nbB];
        const int beg = o.ptr[row], end = o.ptr[row + 1];
        Op op;
        op.load_row(args, row, end - beg);
        op.template edges<64, UB>(args, o, beg, end, wave * 64 * UB, 4 * 64 * UB, lane);
        op.template reduce<64>();
        if (lane == 0) op.to_lds(sh + wave * Op::NS);
        __syncthreads();
        if (wave == 0) {
            op.merge_from_lds(sh, 4);
            if (lane < 16) op.epilogue(args, row, end - beg, lane);
        }
    }
}

// load U (idx,val) pairs for this lane; invalid slots get idx 0 / val 0 and ok = false
template <int G, int U>
__device__ __forceinline__ void load_edges(const OrientDev& o, int e0, int end, int gl, int (&col)[U], float (&a)[U],
                                           bool (&ok)[U]) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int e = e0 + k * G + gl;
        ok[k] = e < end;
        col[k] = ok[k] ? o.idx[e] : 0;
        a[k] = ok[k] ? o.val[e] : 0.0f;
    }
}

// =================================================================================================
// plain CSR SpMM, 16 channels
// =================================================================================================
struct SpmmOp {
    struct Args {
        const float* __restrict__ X;
        float* __restrict__ Y;
    };
    static constexpr int NS = 16;
    float acc[16];

    __device__ __forceinline__ void load_row(const Args&, int, int) {
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
            float x[U][16];
#pragma unroll
            for (int k = 0; k < U; ++k) load_row16(args.X + (size_t)col[k] * 16, x[k]);
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(a[k], x[k][c], acc[c]);
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = group_sum<G>(acc[c]);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c] = acc[c];
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = 0.0f;
            for (int w = 0; w < nw; ++w) v += s[w * NS + c];
            acc[c] = v;
        }
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        args.Y[(size_t)row * 16 + gl] = select16(acc, gl);
    }
};

// =================================================================================================
// attention forward, 16 source channels
// =================================================================================================
struct FwdArgs {
    const float* __restrict__ X;        // [n_src, C]
    const float* __restrict__ xd;       // [n_dst, C]
    const float* __restrict__ qp;       // [n_dst, C]   (C = 16 only; C = 1 computes it inline)
    const float* __restrict__ t;        // [n_dst]
    const float* __restrict__ derived;  // folded weights
    ConvParams p;
    float* __restrict__ h;              // [n_dst, 16]
    float* __restrict__ Z;              // [n_dst, C]
    float* __restrict__ aux;            // [n_dst, 4]
};

struct Fwd16Op {
    using Args = FwdArgs;
    static constexpr int NS = 19;
    float qp[16], t;
    float m, L, u, Z[16];

    __device__ __forceinline__ void load_row(const Args& args, int row, int) {
        load_row16(args.qp + (size_t)row * 16, qp);
        t = args.t[row];
        m = NEG_BIG;
        L = 0.0f;
        u = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) Z[c] = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
            float x[U][16];
#pragma unroll
            for (int k = 0; k < U; ++k) load_row16(args.X + (size_t)col[k] * 16, x[k]);
            float l[U], mi = NEG_BIG;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float d = dot16(qp, x[k], a[k] * t);
                l[k] = ok[k] ? d : NEG_BIG;
                mi = fmaxf(mi, l[k]);
            }
            mi = group_max<G>(mi);                 // uniform over the lanes of the row
            const float m_new = fmaxf(m, mi);
            const float scale = expf(m - m_new);   // 0 on the first pass (m = NEG_BIG), 1 when the max holds
            if (__any(scale != 1.0f)) {            // wave-uniform branch: rescale only when some row's max moved
                L *= scale;
                u *= scale;
#pragma unroll
                for (int c = 0; c < 16; ++c) Z[c] *= scale;
            }
            m = m_new;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float p = expf(l[k] - m_new);  // invalid slots: exp(-huge) == 0
                L += p;
                u = fmaf(p, a[k], u);
#pragma unroll
                for (int c = 0; c < 16; ++c) Z[c] = fmaf(p, x[k][c], Z[c]);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        // a lane that saw no nonzero still holds m = NEG_BIG with zero sums; inside one row every lane
        // shares m by construction (group_max), so plain sums are correct
        m = group_max<G>(m);
        L = group_sum<G>(L);
        u = group_sum<G>(u);
#pragma unroll
        for (int c = 0; c < 16; ++c) Z[c] = group_sum<G>(Z[c]);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
        s[0] = m; s[1] = L; s[2] = u;
#pragma unroll
        for (int c = 0; c < 16; ++c) s[3 + c] = Z[c];
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
        float M = NEG_BIG;
        for (int w = 0; w < nw; ++w) M = fmaxf(M, s[w * NS]);
        float l = 0.0f, uu = 0.0f, z[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) z[c] = 0.0f;
        for (int w = 0; w < nw; ++w) {
            const float f = expf(s[w * NS] - M);   // waves without nonzeros: exp(-huge) == 0
            l = fmaf(f, s[w * NS + 1], l);
            uu = fmaf(f, s[w * NS + 2], uu);
#pragma unroll
            for (int c = 0; c < 16; ++c) z[c] = fmaf(f, s[w * NS + 3 + c], z[c]);
        }
        m = M; L = l; u = uu;
#pragma unroll
        for (int c = 0; c < 16; ++c) Z[c] = z[c];
    }
    // lane gl = output channel
    __device__ __forceinline__ void epilogue(const Args& args, int row, int deg, int gl) {
        const float rinv = 1.0f / (L + 1e-16f);   // torch_geometric.utils.softmax: sum + 1e-16
        const float S = L * rinv;
        const float un = u * rinv;
        float zn[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) zn[c] = Z[c] * rinv;
        float xr[16], wv[16], ws[16];
        load_row16(args.xd + (size_t)row * 16, xr);
        load_row16(args.p.Wv + gl * 16, wv);
        load_row16(args.p.Ws + gl * 16, ws);
        float o = args.p.bs[gl];
        o = fmaf(S, args.p.bv[gl], o);
        o = fmaf(un, args.p.we[gl], o);
        o = dot16(wv, zn, o);
        o = dot16(ws, xr, o);
        args.h[(size_t)row * 16 + gl] = fmaxf(o, 0.0f);
        args.Z[(size_t)row * 16 + gl] = select16(zn, gl);
        if (gl == 0) {
            float4 ax = make_float4(un, deg > 0 ? m : 0.0f, rinv, S);
            reinterpret_cast<float4*>(args.aux)[row] = ax;
        }
    }
};

// =================================================================================================
// attention forward, 1 source channel (layer 1: scalar node features, reference methods.py:90-91)
// =================================================================================================
struct Fwd1Op {
    using Args = FwdArgs;
    static constexpr int NS = 4;
    float qp, t;
    float m, L, u, Z;

    __device__ __forceinline__ void load_row(const Args& args, int row, int) {
        const float* D = args.derived;
        const float x = args.xd[row];
        qp = fmaf(D[OFF_PQ], x, D[OFF_PQ0]);
        t = fmaf(D[OFF_PT], x, D[OFF_PT0]);
        m = NEG_BIG;
        L = 0.0f;
        u = 0.0f;
        Z = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
            float x[U], l[U], mi = NEG_BIG;
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = args.X[col[k]];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float d = fmaf(qp, x[k], a[k] * t);
                l[k] = ok[k] ? d : NEG_BIG;
                mi = fmaxf(mi, l[k]);
            }
            mi = group_max<G>(mi);
            const float m_new = fmaxf(m, mi);
            const float scale = expf(m - m_new);
            L *= scale;
            u *= scale;
            Z *= scale;
            m = m_new;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float p = expf(l[k] - m_new);
                L += p;
                u = fmaf(p, a[k], u);
                Z = fmaf(p, x[k], Z);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        m = group_max<G>(m);
        L = group_sum<G>(L);
        u = group_sum<G>(u);
        Z = group_sum<G>(Z);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
        s[0] = m; s[1] = L; s[2] = u; s[3] = Z;
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
        float M = NEG_BIG;
        for (int w = 0; w < nw; ++w) M = fmaxf(M, s[w * NS]);
        float l = 0.0f, uu = 0.0f, z = 0.0f;
        for (int w = 0; w < nw; ++w) {
            const float f = expf(s[w * NS] - M);
            l = fmaf(f, s[w * NS + 1], l);
            uu = fmaf(f, s[w * NS + 2], uu);
            z = fmaf(f, s[w * NS + 3], z);
        }
        m = M; L = l; u = uu; Z = z;
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int deg, int gl) {
        const float rinv = 1.0f / (L + 1e-16f);
        const float S = L * rinv, un = u * rinv, zn = Z * rinv;
        const float x = args.xd[row];
        float o = args.p.bs[gl];
        o = fmaf(S, args.p.bv[gl], o);
        o = fmaf(un, args.p.we[gl], o);
        o = fmaf(args.p.Wv[gl], zn, o);
        o = fmaf(args.p.Ws[gl], x, o);
        args.h[(size_t)row * 16 + gl] = fmaxf(o, 0.0f);
        if (gl == 0) {
            args.Z[row] = zn;
            float4 ax = make_float4(un, deg > 0 ? m : 0.0f, rinv, S);
            reinterpret_cast<float4*>(args.aux)[row] = ax;
        }
    }
};

// =================================================================================================
// backward, destination-major: dq'_i, ds_i, dt_i and the destination-side input gradient
//   rec_i = { q'_i[16], gv_i[16], t_i, rowmax_i, rinv_i, ge_i, c_i, 0, 0, 0 }   (bwd_pre kernel)
//   alpha_ij = exp(l_ij - rowmax_i) * rinv_i ;  dl_ij = alpha_ij ( <gv_i, X_j> + a_ij ge_i + c_i )
// =================================================================================================
struct BwdDstArgs {
    const float* __restrict__ X;        // [n_src, C]
    const float* __restrict__ rec;      // [n_dst, REC_W] (C = 16) or [n_dst, 8] (C = 1)
    const float* __restrict__ g;        // [n_dst, 16] relu-masked output gradient
    const float* __restrict__ derived;
    float* __restrict__ dqp;            // [n_dst, C]
    float* __restrict__ dsdt;           // [n_dst, 2]
    float* __restrict__ dx_dst;         // [n_dst, C] or nullptr
    int accumulate;
};

struct BwdDst16Op {
    using Args = BwdDstArgs;
    static constexpr int NS = 18;
    float qp[16], gv[16], t, m, rinv, ge, cc;
    float dqp[16], ds, dt;

    __device__ __forceinline__ void load_row(const Args& args, int row, int) {
        const float* r = args.rec + (size_t)row * REC_W;
        load_row16(r, qp);
        load_row16(r + 16, gv);
        const float4 s0 = reinterpret_cast<const float4*>(r + 32)[0];
        const float4 s1 = reinterpret_cast<const float4*>(r + 32)[1];
        t = s0.x; m = s0.y; rinv = s0.z; ge = s0.w; cc = s1.x;
        ds = 0.0f;
        dt = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) dqp[c] = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
            float x[U][16];
#pragma unroll
            for (int k = 0; k < U; ++k) load_row16(args.X + (size_t)col[k] * 16, x[k]);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float l = dot16(qp, x[k], a[k] * t);
                const float alpha = ok[k] ? expf(l - m) * rinv : 0.0f;
                const float dl = alpha * dot16(gv, x[k], fmaf(a[k], ge, cc));
                ds += dl;
                dt = fmaf(dl, a[k], dt);
#pragma unroll
                for (int c = 0; c < 16; ++c) dqp[c] = fmaf(dl, x[k][c], dqp[c]);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        ds = group_sum<G>(ds);
        dt = group_sum<G>(dt);
#pragma unroll
        for (int c = 0; c < 16; ++c) dqp[c] = group_sum<G>(dqp[c]);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
        s[0] = ds; s[1] = dt;
#pragma unroll
        for (int c = 0; c < 16; ++c) s[2 + c] = dqp[c];
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
        float a = 0.0f, b = 0.0f, z[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) z[c] = 0.0f;
        for (int w = 0; w < nw; ++w) {
            a += s[w * NS];
            b += s[w * NS + 1];
#pragma unroll
            for (int c = 0; c < 16; ++c) z[c] += s[w * NS + 2 + c];
        }
        ds = a; dt = b;
#pragma unroll
        for (int c = 0; c < 16; ++c) dqp[c] = z[c];
    }
    // dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt        (lane gl = input channel)
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        args.dqp[(size_t)row * 16 + gl] = select16(dqp, gl);
        if (gl == 0) reinterpret_cast<float2*>(args.dsdt)[row] = make_float2(ds, dt);
        if (args.dx_dst) {
            const float* D = args.derived;
            float gr[16], wsT[16], pqT[16];
            load_row16(args.g + (size_t)row * 16, gr);
            load_row16(D + OFF_WST + gl * 16, wsT);   // WsT[gl][o] = Ws[o][gl]
            load_row16(D + OFF_PQT + gl * 16, pqT);   // PqT[gl][k] = Pq[k][gl]
            float v = ds * D[OFF_PB + gl];
            v = fmaf(dt, D[OFF_PT + gl], v);
            v = dot16(wsT, gr, v);
            v = dot16(pqT, dqp, v);
            float* dst = args.dx_dst + (size_t)row * 16 + gl;
            *dst = args.accumulate ? *dst + v : v;
        }
    }
};

// C = 1: layer-1 convs.  Their inputs are data (no input gradient), only dq', ds, dt are produced.
//   rec8_i = { q'_i, gv_i, t_i, rowmax_i, rinv_i, ge_i, c_i, 0 }
struct BwdDst1Op {
    using Args = BwdDstArgs;
    static constexpr int NS = 3;
    float qp, gv, t, m, rinv, ge, cc;
    float dqp, ds, dt;

    __device__ __forceinline__ void load_row(const Args& args, int row, int) {
        const float4 s0 = reinterpret_cast<const float4*>(args.rec + (size_t)row * 8)[0];
        const float4 s1 = reinterpret_cast<const float4*>(args.rec + (size_t)row * 8)[1];
        qp = s0.x; gv = s0.y; t = s0.z; m = s0.w;
        rinv = s1.x; ge = s1.y; cc = s1.z;
        dqp = 0.0f; ds = 0.0f; dt = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
            float x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = args.X[col[k]];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float l = fmaf(qp, x[k], a[k] * t);
                const float alpha = ok[k] ? expf(l - m) * rinv : 0.0f;
                const float dl = alpha * fmaf(gv, x[k], fmaf(a[k], ge, cc));
                ds += dl;
                dt = fmaf(dl, a[k], dt);
                dqp = fmaf(dl, x[k], dqp);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        ds = group_sum<G>(ds);
        dt = group_sum<G>(dt);
        dqp = group_sum<G>(dqp);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
        s[0] = ds; s[1] = dt; s[2] = dqp;
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int w = 0; w < nw; ++w) {
            a += s[w * NS];
            b += s[w * NS + 1];
            c += s[w * NS + 2];
        }
        ds = a; dt = b; dqp = c;
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        if (gl == 0) {
            args.dqp[row] = dqp;
            reinterpret_cast<float2*>(args.dsdt)[row] = make_float2(ds, dt);
        }
    }
};

// =================================================================================================
// backward, source-major (walks the opposite orientation): dX_j = sum_i alpha_ij gv_i + dl_ij q'_i
// rows of this sweep are SOURCE nodes j; the gathered records belong to destination nodes i.
// =================================================================================================
struct BwdSrcArgs {
    const float* __restrict__ X;    // [n_src, 16] features of the nodes that are rows here
    const float* __restrict__ rec;  // [n_dst, REC_W]
    float* __restrict__ dX;         // [n_src, 16]
    int accumulate;
};

struct BwdSrc16Op {
    using Args = BwdSrcArgs;
    static constexpr int NS = 16;
    float xj[16], acc[16];

    __device__ __forceinline__ void load_row(const Args& args, int row, int) {
        load_row16(args.X + (size_t)row * 16, xj);
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U>(o, e0, end, gl, col, a, ok);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float* r = args.rec + (size_t)col[k] * REC_W;
                float qp[16], gv[16];
                load_row16(r, qp);
                load_row16(r + 16, gv);
                const float4 s0 = reinterpret_cast<const float4*>(r + 32)[0];
                const float cc = r[36];
                const float l = dot16(qp, xj, a[k] * s0.x);
                const float alpha = ok[k] ? expf(l - s0.y) * s0.z : 0.0f;
                const float dl = alpha * dot16(gv, xj, fmaf(a[k], s0.w, cc));
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(alpha, gv[c], fmaf(dl, qp[c], acc[c]));
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = group_sum<G>(acc[c]);
    }
    __device__ __forceinline__ void to_lds(float* s) const {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c] = acc[c];
    }
    __device__ __forceinline__ void merge_from_lds(const float* s, int nw) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = 0.0f;
            for (int w = 0; w < nw; ++w) v += s[w * NS + c];
            acc[c] = v;
        }
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        float* dst = args.dX + (size_t)row * 16 + gl;
        const float v = select16(acc, gl);
        *dst = args.accumulate ? *dst + v : v;
    }
};

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
template <class Op, int UA, int UB>
static int launch_sweep(const Orient& o, const typename Op::Args& args, hipStream_t s, const char* name) {
    OrientDev d = make_dev(o);
    const int64_t blocks = (int64_t)d.nbA + d.nbB + d.n_block;
    if (blocks == 0) return MLLP_OK;
    if (blocks >= INT32_MAX) return fail(MLLP_ERANGE, "too many workgroups");
    hipLaunchKernelGGL((sweep_kernel<Op, UA, UB>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MLLP_OK;
}

int launch_spmm(const Orient& o, const float* H, float* Y, hipStream_t s) {
    SpmmOp::Args a{H, Y};
    return launch_sweep<SpmmOp, 2, 2>(o, a, s, "spmm_csr");
}

int launch_attn_fwd(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                    const float* x_dst, float* h_out, hipStream_t s) {
    FwdArgs a;
    a.X = x_src; a.xd = x_dst; a.qp = w.qp; a.t = w.t; a.derived = w.derived;
    a.p = conv_params_at(conv_params, cin);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    if (cin == 16) return launch_sweep<Fwd16Op, 2, 2>(o, a, s, "attn_fwd16");
    return launch_sweep<Fwd1Op, 2, 2>(o, a, s, "attn_fwd1");
}

int launch_attn_bwd_dst(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                        const float* g, float* dx_dst, int accumulate, hipStream_t s) {
    (void)conv_params;
    BwdDstArgs a;
    a.X = x_src; a.rec = w.rec; a.g = g; a.derived = w.derived;
    a.dqp = w.dqp; a.dsdt = w.dsdt; a.dx_dst = dx_dst; a.accumulate = accumulate;
    if (cin == 16) return launch_sweep<BwdDst16Op, 2, 2>(o, a, s, "attn_bwd_dst16");
    a.dx_dst = nullptr;
    return launch_sweep<BwdDst1Op, 2, 2>(o, a, s, "attn_bwd_dst1");
}

int launch_attn_bwd_src(const Orient& o_src_major, const ConvWs& w, const float* x_src, float* dx_src, int accumulate,
                        hipStream_t s) {
    BwdSrcArgs a{x_src, w.rec, dx_src, accumulate};
    return launch_sweep<BwdSrc16Op, 1, 1>(o_src_major, a, s, "attn_bwd_src16");
}

}  // namespace mllp
