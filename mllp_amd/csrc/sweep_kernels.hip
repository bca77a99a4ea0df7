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
// Mapping (16-channel sweeps): one QUAD of lanes per nonzero, lane `part` = 0..3 owning channels
// 4*part .. 4*part+3.  The four lanes of a quad read the four 16-byte pieces of ONE 64-byte source row,
// so a wave-wide gather instruction touches 16 rows as 16 whole 64-byte segments (the vector memory
// pipeline looks up one line per quad instead of one per lane: the first version, one lane per nonzero
// with four dwordx4 loads each, ran the gathers at ~10 B/clk/CU).  Dot products are 4 FMAs + a 2-step
// DPP quad reduction; accumulators are 4 floats per lane; per-row totals are combined over the quads of
// the row's lane group with DPP row rotations.  Scalar (layer-1) sweeps keep one lane per nonzero.
// Rows are tiered by length so that no work item loops more than a few times in the latency-bound
// regime (real Netlib) while the throughput regime (synthetic) keeps 16 lanes per row:
//   group tier : 16 lanes (4 quads) per row, 4 rows per wavefront
//   wave tier  : 64 lanes (16 quads) per row
//   chunk tier : 256 threads per chunk of a row; a row longer than the chunk size is split over several
//                workgroups whose partial states go to a scratch buffer and are merged by combine_kernel
// Tiles of the group tier are remapped so that each XCD walks a contiguous row range: the source
// rows gathered by neighbouring tiles then come out of one XCD's L2.
#include "device_utils.h"
#include "internal.h"

namespace mllp {

struct OrientDev {
    const int* __restrict__ ptr;
    const int* __restrict__ idx;
    const float* __restrict__ val;
    const int* __restrict__ rows_wave;
    const int4* __restrict__ chunks;   // {row, beg, end, slot}; slot < 0: the row's only chunk
    const int4* __restrict__ split;    // {row, first_slot, n_chunks, 0}
    int n_dst, n_wave, n_chunk, n_split, nbA, nbB, tier_wave;
};

static OrientDev make_dev(const Orient& o) {
    OrientDev d;
    d.ptr = o.ptr; d.idx = o.idx; d.val = o.val;
    d.rows_wave = o.rows_wave;
    d.chunks = reinterpret_cast<const int4*>(o.chunks);
    d.split = reinterpret_cast<const int4*>(o.split);
    d.n_dst = o.n_dst; d.n_wave = o.n_wave; d.n_chunk = o.n_chunk; d.n_split = o.n_split;
    d.tier_wave = o.tier_wave;
    d.nbA = (o.n_dst + 15) / 16;   // the group tier visits every row and skips the long ones: no index list to chase
    d.nbB = (o.n_wave + 3) / 4;
    return d;
}

// -------------------------------------------------------------------------------------------------
// generic tiered driver.  Op provides:
//   Args, NS (floats of exchanged state), LPN (lanes per nonzero: 4 or 1)
//   load_row(args,row,deg,gl)   per-row inputs
//   edges<G,U>(args,o,beg,end,first,stride,gl)   accumulate this lane's nonzeros; one pass of a G-lane
//                               group covers (G / LPN) * U nonzeros
//   reduce<G>()                 all-reduce the state over the G lanes of the row
//   to_mem(float*)              lanes 0..3 of a reduced group write the state
//   merge_from(const float*, n) rebuild the state from n stored states (LDS or global)
//   epilogue(args,row,deg,gl)   gl in [0,16): write the row's outputs
// -------------------------------------------------------------------------------------------------
template <class Op, int UA, int UB>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(typename Op::Args args, OrientDev o, float* __restrict__ scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroups are dispatched in blockIdx order: the long rows (wave and chunk tiers) go first so that they run
    // next to the mass of short rows instead of forming the tail of the kernel
    const int heavy = o.nbB + o.n_chunk;
    const int b = (int)blockIdx.x < heavy ? (int)blockIdx.x + o.nbA : (int)blockIdx.x - heavy;
    if (b < o.nbA) {
        const int tile = xcd_tile(b, o.nbA);
        const int row = tile * 16 + wave * 4 + (lane >> 4);
        int beg = 0, end = 0;
        if (row < o.n_dst) {
            beg = o.ptr[row];
            end = o.ptr[row + 1];
        }
        if (row < o.n_dst && end - beg <= o.tier_wave) {   // longer rows belong to the wave / chunk tiers
            Op op;
            op.load_row(args, row, end - beg, lane & 15);
            // one nonzero slot per pass suits the short rows that dominate real LPs (every unrolled slot is executed
            // even when masked off), but a 40-nonzero row then takes 10 dependent passes and its wave becomes the
            // tail: waves that hold such a row take UB slots per pass, like the wave tier
            if (UA == 1 && UB > 1 && __any(end - beg > 2 * (16 / Op::LPN)))
                op.template edges<16, UB>(args, o, beg, end, 0, (16 / Op::LPN) * UB, lane & 15);
            else
                op.template edges<16, UA>(args, o, beg, end, 0, (16 / Op::LPN) * UA, lane & 15);
            op.template reduce<16>();
            op.epilogue(args, row, end - beg, lane & 15);
        }
    } else if (b < o.nbA + o.nbB) {
        const int slot = (b - o.nbA) * 4 + wave;
        if (slot < o.n_wave) {
            const int row = o.rows_wave[slot];
            const int beg = o.ptr[row], end = o.ptr[row + 1];
            Op op;
            op.load_row(args, row, end - beg, lane);
            op.template edges<64, UB>(args, o, beg, end, 0, (64 / Op::LPN) * UB, lane);
            op.template reduce<64>();
            if (lane < 16) op.epilogue(args, row, end - beg, lane);
        }
    } else {
        __shared__ float sh[4 * Op::NS];
        const int4 ck = o.chunks[b - o.nbA - o.nbB];
        const int row = ck.x, beg = ck.y, end = ck.z;
        constexpr int PER = (64 / Op::LPN) * UB;   // nonzeros per wave and pass
        Op op;
        op.load_row(args, row, end - beg, lane);
        op.template edges<64, UB>(args, o, beg, end, wave * PER, 4 * PER, lane);
        op.template reduce<64>();
        if (lane < 4) op.to_mem(sh + wave * Op::NS);
        __syncthreads();
        if (wave == 0) {
            op.merge_from(sh, 4);
            if (ck.w < 0) {
                if (lane < 16) op.epilogue(args, row, o.ptr[row + 1] - o.ptr[row], lane);
            } else if (lane < 4) {
                op.to_mem(scratch + (size_t)ck.w * Op::NS);
            }
        }
    }
}

// rows that were split over several chunk workgroups: merge their partial states in chunk order
template <class Op>
__global__ __launch_bounds__(BLOCK) void combine_kernel(typename Op::Args args, OrientDev o,
                                                        const float* __restrict__ scratch) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= o.n_split) return;
    const int4 sp = o.split[i];
    const int gl = threadIdx.x & 15;
    const int deg = o.ptr[sp.x + 1] - o.ptr[sp.x];
    Op op;
    op.load_row(args, sp.x, deg, gl);
    op.merge_from(scratch + (size_t)sp.y * Op::NS, sp.z);
    op.epilogue(args, sp.x, deg, gl);
}

// load U (idx,val) pairs for this lane's nonzero slots; invalid slots get idx 0 / val 0 and ok = false
template <int G, int U, int LPN>
__device__ __forceinline__ void load_edges(const OrientDev& o, int e0, int end, int gl, int (&col)[U], float (&a)[U],
                                           bool (&ok)[U]) {
    const int q = (LPN == 4) ? (gl >> 2) : gl;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int e = e0 + k * (G / LPN) + q;
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
    static constexpr int NS = 16, LPN = 4;
    float4 acc;
    int part;

    __device__ __forceinline__ void load_row(const Args&, int, int, int gl) {
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        part = gl & 3;
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
            float4 x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = ld4(args.X + (size_t)col[k] * 16 + 4 * part);
#pragma unroll
            for (int k = 0; k < U; ++k) fma4(a[k], x[k], acc);
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        acc.x = quads_sum<G>(acc.x);
        acc.y = quads_sum<G>(acc.y);
        acc.z = quads_sum<G>(acc.z);
        acc.w = quads_sum<G>(acc.w);
    }
    __device__ __forceinline__ void to_mem(float* s) const { *reinterpret_cast<float4*>(s + 4 * part) = acc; }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < n; ++w) {
            const float4 t = ld4(s + w * NS + 4 * part);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        acc = v;
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        if (gl < 4) *reinterpret_cast<float4*>(args.Y + (size_t)row * 16 + 4 * gl) = acc;
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
    static constexpr int NS = 20, LPN = 4;   // {m, L, u, -, Z[16]}
    float4 qp, Z;
    float t, m, L, u, skip;
    int part;

    __device__ __forceinline__ void load_row(const Args& args, int row, int, int gl) {
        part = gl & 3;
        qp = ld4(args.qp + (size_t)row * 16 + 4 * part);
        t = args.t[row];
        skip = 0.0f;   // (hoisting the root term Ws x_i + bs here cost 26 VGPRs and 14 % on the large batches)
        m = NEG_BIG;
        L = 0.0f;
        u = 0.0f;
        Z = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
            float4 x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = ld4(args.X + (size_t)col[k] * 16 + 4 * part);
            float l[U], mi = NEG_BIG;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float d = fmaf(a[k], t, quad_sum(dot4(qp, x[k])));
                l[k] = ok[k] ? d : NEG_BIG;
                mi = fmaxf(mi, l[k]);
            }
            mi = quads_max<G>(mi);                 // uniform over the lanes of the row
            const float m_new = fmaxf(m, mi);
            const float scale = exp_acc(m - m_new);   // 0 on the first pass (m = NEG_BIG), 1 when the max holds
            L *= scale;
            u *= scale;
            Z.x *= scale; Z.y *= scale; Z.z *= scale; Z.w *= scale;
            m = m_new;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float p = exp_acc(l[k] - m_new);  // invalid slots: exp(-huge) == 0
                L += p;
                u = fmaf(p, a[k], u);
                fma4(p, x[k], Z);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        // every lane of the row shares m (quads_max); L and u are replicated inside a quad, Z is owned per part:
        // summing over the quads (lanes 4 apart) is the row total for all of them
        L = quads_sum<G>(L);
        u = quads_sum<G>(u);
        Z.x = quads_sum<G>(Z.x);
        Z.y = quads_sum<G>(Z.y);
        Z.z = quads_sum<G>(Z.z);
        Z.w = quads_sum<G>(Z.w);
    }
    __device__ __forceinline__ void to_mem(float* s) const {
        if (part == 0) *reinterpret_cast<float4*>(s) = make_float4(m, L, u, 0.0f);
        *reinterpret_cast<float4*>(s + 4 + 4 * part) = Z;
    }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float M = NEG_BIG;
        for (int w = 0; w < n; ++w) M = fmaxf(M, s[w * NS]);
        float l = 0.0f, uu = 0.0f;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < n; ++w) {
            const float4 h0 = ld4(s + w * NS);
            const float f = exp_acc(h0.x - M);       // states without nonzeros: exp(-huge) == 0
            l = fmaf(f, h0.y, l);
            uu = fmaf(f, h0.z, uu);
            fma4(f, ld4(s + w * NS + 4 + 4 * part), z);
        }
        m = M; L = l; u = uu; Z = z;
    }
    // lane gl = output channel
    __device__ __forceinline__ void epilogue(const Args& args, int row, int deg, int gl) {
        const float rinv = 1.0f / (L + 1e-16f);   // torch_geometric.utils.softmax: sum + 1e-16
        const float S = L * rinv;
        const float un = u * rinv;
        const float4 zn = make_float4(Z.x * rinv, Z.y * rinv, Z.z * rinv, Z.w * rinv);
        float za[16], xr[16], wv[16], ws[16];
        quad_allgather(zn, za);
        load_row16(args.xd + (size_t)row * 16, xr);
        load_row16(args.p.Wv + gl * 16, wv);
        load_row16(args.p.Ws + gl * 16, ws);
        float o = args.p.bs[gl] + skip;
        o = fmaf(S, args.p.bv[gl], o);
        o = fmaf(un, args.p.we[gl], o);
        o = dot16(wv, za, o);
        o = dot16(ws, xr, o);
        args.h[(size_t)row * 16 + gl] = fmaxf(o, 0.0f);
        if (gl < 4) *reinterpret_cast<float4*>(args.Z + (size_t)row * 16 + 4 * gl) = zn;
        if (gl == 0) reinterpret_cast<float4*>(args.aux)[row] = make_float4(un, deg > 0 ? m : 0.0f, rinv, S);
    }
};

// =================================================================================================
// attention forward, 1 source channel (layer 1: scalar node features, reference methods.py:90-91):
// one lane per nonzero
// =================================================================================================
struct Fwd1Op {
    using Args = FwdArgs;
    static constexpr int NS = 4, LPN = 1;
    float qp, t;
    float m, L, u, Z;

    __device__ __forceinline__ void load_row(const Args& args, int row, int, int) {
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
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
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
            const float scale = exp_acc(m - m_new);
            L *= scale;
            u *= scale;
            Z *= scale;
            m = m_new;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float p = exp_acc(l[k] - m_new);
                L += p;
                u = fmaf(p, a[k], u);
                Z = fmaf(p, x[k], Z);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        L = group_sum<G>(L);
        u = group_sum<G>(u);
        Z = group_sum<G>(Z);
    }
    __device__ __forceinline__ void to_mem(float* s) const {
        *reinterpret_cast<float4*>(s) = make_float4(m, L, u, Z);   // lanes 0..3 write the same value
    }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float M = NEG_BIG;
        for (int w = 0; w < n; ++w) M = fmaxf(M, s[w * NS]);
        float l = 0.0f, uu = 0.0f, z = 0.0f;
        for (int w = 0; w < n; ++w) {
            const float4 h0 = ld4(s + w * NS);
            const float f = exp_acc(h0.x - M);
            l = fmaf(f, h0.y, l);
            uu = fmaf(f, h0.z, uu);
            z = fmaf(f, h0.w, z);
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
            reinterpret_cast<float4*>(args.aux)[row] = make_float4(un, deg > 0 ? m : 0.0f, rinv, S);
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
    static constexpr int NS = 20, LPN = 4;   // {ds, dt, -, -, dqp[16]}
    float4 qp, gv, dqp;
    float t, m, rinv, ge, cc, ds, dt, v0;
    int part;

    __device__ __forceinline__ void load_row(const Args& args, int row, int, int gl) {
        part = gl & 3;
        v0 = 0.0f;
        if (args.dx_dst) {   // Ws^T g_i of input channel (gl & 15): independent of the sweep, loaded up front
            float gr[16], wsT[16];
            load_row16(args.g + (size_t)row * 16, gr);
            load_row16(args.derived + OFF_WST + (gl & 15) * 16, wsT);
            v0 = dot16(wsT, gr, 0.0f);
        }
        const float* r = args.rec + (size_t)row * REC_W;
        qp = ld4(r + 4 * part);
        gv = ld4(r + 16 + 4 * part);
        const float4 s0 = ld4(r + 32);
        t = s0.x; m = s0.y; rinv = s0.z; ge = s0.w;
        cc = r[36];
        ds = 0.0f;
        dt = 0.0f;
        dqp = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
            float4 x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = ld4(args.X + (size_t)col[k] * 16 + 4 * part);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float l = fmaf(a[k], t, quad_sum(dot4(qp, x[k])));
                const float alpha = ok[k] ? exp_acc(l - m) * rinv : 0.0f;
                const float dl = alpha * (quad_sum(dot4(gv, x[k])) + fmaf(a[k], ge, cc));
                ds += dl;
                dt = fmaf(dl, a[k], dt);
                fma4(dl, x[k], dqp);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        ds = quads_sum<G>(ds);
        dt = quads_sum<G>(dt);
        dqp.x = quads_sum<G>(dqp.x);
        dqp.y = quads_sum<G>(dqp.y);
        dqp.z = quads_sum<G>(dqp.z);
        dqp.w = quads_sum<G>(dqp.w);
    }
    __device__ __forceinline__ void to_mem(float* s) const {
        if (part == 0) *reinterpret_cast<float4*>(s) = make_float4(ds, dt, 0.0f, 0.0f);
        *reinterpret_cast<float4*>(s + 4 + 4 * part) = dqp;
    }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float a = 0.0f, b = 0.0f;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < n; ++w) {
            a += s[w * NS];
            b += s[w * NS + 1];
            const float4 v = ld4(s + w * NS + 4 + 4 * part);
            z.x += v.x; z.y += v.y; z.z += v.z; z.w += v.w;
        }
        ds = a; dt = b; dqp = z;
    }
    // dx_i = Ws^T g_i + Pq^T dq'_i + ds_i Pb + dt_i Pt        (lane gl = input channel)
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        if (gl < 4) *reinterpret_cast<float4*>(args.dqp + (size_t)row * 16 + 4 * gl) = dqp;
        if (gl == 0) reinterpret_cast<float2*>(args.dsdt)[row] = make_float2(ds, dt);
        if (args.dx_dst) {
            const float* D = args.derived;
            float da[16], pqT[16];
            quad_allgather(dqp, da);
            load_row16(D + OFF_PQT + gl * 16, pqT);   // PqT[gl][k] = Pq[k][gl]
            float v = fmaf(ds, D[OFF_PB + gl], v0);
            v = fmaf(dt, D[OFF_PT + gl], v);
            v = dot16(pqT, da, v);
            float* dst = args.dx_dst + (size_t)row * 16 + gl;
            *dst = args.accumulate ? *dst + v : v;
        }
    }
};

// C = 1: layer-1 convs.  Their inputs are data (no input gradient), only dq', ds, dt are produced.
//   rec8_i = { q'_i, gv_i, t_i, rowmax_i, rinv_i, ge_i, c_i, 0 }
struct BwdDst1Op {
    using Args = BwdDstArgs;
    static constexpr int NS = 4, LPN = 1;
    float qp, gv, t, m, rinv, ge, cc;
    float dqp, ds, dt;

    __device__ __forceinline__ void load_row(const Args& args, int row, int, int) {
        const float4 s0 = ld4(args.rec + (size_t)row * 8);
        const float4 s1 = ld4(args.rec + (size_t)row * 8 + 4);
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
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
            float x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = args.X[col[k]];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float l = fmaf(qp, x[k], a[k] * t);
                const float alpha = ok[k] ? exp_acc(l - m) * rinv : 0.0f;
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
    __device__ __forceinline__ void to_mem(float* s) const {
        *reinterpret_cast<float4*>(s) = make_float4(ds, dt, dqp, 0.0f);
    }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int w = 0; w < n; ++w) {
            const float4 v = ld4(s + w * NS);
            a += v.x; b += v.y; c += v.z;
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
    static constexpr int NS = 16, LPN = 4;
    float4 xj, acc;
    int part;

    __device__ __forceinline__ void load_row(const Args& args, int row, int, int gl) {
        part = gl & 3;
        xj = ld4(args.X + (size_t)row * 16 + 4 * part);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int G, int U>
    __device__ __forceinline__ void edges(const Args& args, const OrientDev& o, int beg, int end, int first, int stride,
                                          int gl) {
        for (int e0 = beg + first; e0 < end; e0 += stride) {
            int col[U];
            float a[U];
            bool ok[U];
            load_edges<G, U, LPN>(o, e0, end, gl, col, a, ok);
            float4 qp[U], gv[U], s0[U];
            float cc[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float* r = args.rec + (size_t)col[k] * REC_W;
                qp[k] = ld4(r + 4 * part);
                gv[k] = ld4(r + 16 + 4 * part);
                s0[k] = ld4(r + 32);        // {t, rowmax, rinv, ge}
                cc[k] = r[36];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float l = fmaf(a[k], s0[k].x, quad_sum(dot4(qp[k], xj)));
                const float alpha = ok[k] ? exp_acc(l - s0[k].y) * s0[k].z : 0.0f;
                const float dl = alpha * (quad_sum(dot4(gv[k], xj)) + fmaf(a[k], s0[k].w, cc[k]));
                fma4(alpha, gv[k], acc);
                fma4(dl, qp[k], acc);
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void reduce() {
        acc.x = quads_sum<G>(acc.x);
        acc.y = quads_sum<G>(acc.y);
        acc.z = quads_sum<G>(acc.z);
        acc.w = quads_sum<G>(acc.w);
    }
    __device__ __forceinline__ void to_mem(float* s) const { *reinterpret_cast<float4*>(s + 4 * part) = acc; }
    __device__ __forceinline__ void merge_from(const float* s, int n) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < n; ++w) {
            const float4 t = ld4(s + w * NS + 4 * part);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        acc = v;
    }
    __device__ __forceinline__ void epilogue(const Args& args, int row, int, int gl) {
        if (gl < 4) {
            float4* dst = reinterpret_cast<float4*>(args.dX + (size_t)row * 16 + 4 * gl);
            float4 v = acc;
            if (args.accumulate) {
                const float4 old = *dst;
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            *dst = v;
        }
    }
};

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
// UA: nonzero slots per lane group and pass in the group tier.  Short rows dominate the latency regime (real
// Netlib: median 2-4 nonzeros per row), where every unrolled slot is executed instruction for instruction even
// when it is masked off, so UA = 1 there; the throughput regime (rows of 100+) wants the loads of 4 slots in flight.
template <class Op, int UA, int UB>
static int launch_sweep(const Orient& o, const typename Op::Args& args, float* scratch, hipStream_t s,
                        const char* name) {
    static_assert(Op::NS <= SCRATCH_NS, "scratch slot too small");
    OrientDev d = make_dev(o);
    const int64_t blocks = (int64_t)d.nbA + d.nbB + d.n_chunk;
    if (blocks == 0) return MLLP_OK;
    if (blocks >= INT32_MAX) return fail(MLLP_ERANGE, "too many workgroups");
    if (o.short_rows)
        hipLaunchKernelGGL((sweep_kernel<Op, 1, UB>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args, d, scratch);
    else
        hipLaunchKernelGGL((sweep_kernel<Op, UA, UB>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args, d, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    if (d.n_split > 0) {
        hipLaunchKernelGGL((combine_kernel<Op>), dim3((unsigned)((d.n_split + 15) / 16)), dim3(BLOCK), 0, s, args, d,
                           scratch);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, name);
    }
    return MLLP_OK;
}

int launch_spmm(const Orient& o, const float* H, float* Y, float* scratch, hipStream_t s) {
    if (o.stream.n_tiles > 0) return launch_spmm_stream(o.stream, o.n_dst, o.n_src, H, Y, s);
    if (o.tiled.n_tiles > 0) return launch_spmm_tiled(o.tiled, o.n_dst, o.n_src, H, Y, s);
    SpmmOp::Args a{H, Y};
    return launch_sweep<SpmmOp, 4, 4>(o, a, scratch, s, "spmm_csr");
}

int launch_attn_fwd(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                    const float* x_dst, float* h_out, float* scratch, hipStream_t s) {
    FwdArgs a;
    a.X = x_src; a.xd = x_dst; a.qp = w.qp; a.t = w.t; a.derived = w.derived;
    a.p = conv_params_at(conv_params, cin);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    if (cin == 16 && o.stream_attn.n_tiles > 0)
        return launch_fwd16_stream(o.stream_attn, o.n_dst, o.n_src, conv_params, w, x_src, x_dst, h_out, s);
    if (cin == 16 && o.tiled_attn.n_tiles > 0)
        return launch_fwd16_tiled(o.tiled_attn, o.n_dst, o.n_src, conv_params, w, x_src, x_dst, h_out, s);
    if (cin == 16) return launch_sweep<Fwd16Op, 4, 4>(o, a, scratch, s, "attn_fwd16");
    if (o.lane1.n_tiles > 0) return launch_fwd1_lane(o.lane1, o.n_dst, o.n_src, conv_params, w, x_src, x_dst, h_out, s);
    if (o.tiled_scalar.n_tiles > 0)
        return launch_fwd1_tiled(o.tiled_scalar, o.n_dst, o.n_src, conv_params, w, x_src, x_dst, h_out, s);
    return launch_sweep<Fwd1Op, 2, 2>(o, a, scratch, s, "attn_fwd1");
}

int launch_attn_bwd_dst(const Orient& o, int cin, const float* conv_params, const ConvWs& w, const float* x_src,
                        const float* g, float* dx_dst, int accumulate, float* scratch, hipStream_t s) {
    (void)conv_params;
    BwdDstArgs a;
    a.X = x_src; a.rec = w.rec; a.g = g; a.derived = w.derived;
    a.dqp = w.dqp; a.dsdt = w.dsdt; a.dx_dst = dx_dst; a.accumulate = accumulate;
    if (cin == 16 && o.stream_bdst.n_tiles > 0)
        return launch_bwddst16_stream(o.stream_bdst, o.n_dst, o.n_src, w, x_src, g, dx_dst, accumulate, s);
    if (cin == 16 && o.tiled_bdst.n_tiles > 0)
        return launch_bwddst16_tiled(o.tiled_bdst, o.n_dst, o.n_src, w, x_src, g, dx_dst, accumulate, s);
    if (cin == 16) return launch_sweep<BwdDst16Op, 4, 4>(o, a, scratch, s, "attn_bwd_dst16");
    a.dx_dst = nullptr;
    if (o.lane1.n_tiles > 0) return launch_bwddst1_lane(o.lane1, o.n_dst, o.n_src, w, x_src, s);
    if (o.tiled_scalar.n_tiles > 0) return launch_bwddst1_tiled(o.tiled_scalar, o.n_dst, o.n_src, w, x_src, s);
    return launch_sweep<BwdDst1Op, 2, 2>(o, a, scratch, s, "attn_bwd_dst1");
}

int launch_attn_bwd_src(const Orient& o_src_major, const ConvWs& w, const float* x_src, float* dx_src, int accumulate,
                        float* scratch, hipStream_t s) {
    if (o_src_major.stream_bsrc.n_tiles > 0)
        return launch_bwdsrc16_stream(o_src_major.stream_bsrc, o_src_major.n_dst, o_src_major.n_src, w.rec, x_src, dx_src,
                                      accumulate, s);
    if (o_src_major.tiled_bsrc.n_tiles > 0)
        return launch_bwdsrc16_tiled(o_src_major.tiled_bsrc, o_src_major.n_dst, o_src_major.n_src, w.rec, x_src, dx_src,
                                     accumulate, s);
    BwdSrcArgs a{x_src, w.rec, dx_src, accumulate};
    return launch_sweep<BwdSrc16Op, 2, 2>(o_src_major, a, scratch, s, "attn_bwd_src16");
}

}  // namespace mllp
