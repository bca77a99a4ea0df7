// scalar_ops.h -- per-row arithmetic of the layer-1 (one input channel) attention sweeps, destination-major: the state a
// row carries, what one nonzero does to it, and the row's epilogue.  Shared by the LDS-tiled kernel
// (tiled_kernels.hip::scalar_tiled_kernel) and the lane-per-row streamed kernel (lane_stream.hip); the arithmetic is the
// generic sweep's (sweep_kernels.hip::Fwd1Op / BwdDst1Op; reference linear_program_methods.py:90-91, 241-242).
#pragma once
#include "device_utils.h"
#include "internal.h"

namespace mllp {

struct Fwd1TiledArgs {
    const float* __restrict__ X;        // [n_src]
    const float* __restrict__ xd;       // [n_dst]
    const float* __restrict__ derived;
    ConvParams p;
    float* __restrict__ h;              // [n_dst, 16]
    float* __restrict__ Z;              // [n_dst]
    float* __restrict__ aux;            // [n_dst, 4]
};
struct BwdDst1TiledArgs {
    const float* __restrict__ X;        // [n_src]
    const float* __restrict__ rec;      // [n_dst, 8] {q', gv, t, rowmax, rinv, ge, c, 0}
    float* __restrict__ dqp;            // [n_dst]
    float* __restrict__ dsdt;           // [n_dst, 2]
};

// forward: per row {rowmax, L, u, Z} and {q', t}
struct Fwd1T {
    using Args = Fwd1TiledArgs;
    float4 st;      // m, L, u, Z
    float qp, t;
    __device__ __forceinline__ static void init_row(const Args& a, int row, bool valid, float4* S0, float4* S1) {
        const float* D = a.derived;
        const float x = valid ? a.xd[row] : 0.0f;
        *S0 = make_float4(NEG_BIG, 0.f, 0.f, 0.f);
        *S1 = make_float4(fmaf(D[OFF_PQ], x, D[OFF_PQ0]), fmaf(D[OFF_PT], x, D[OFF_PT0]), x, 0.f);
    }
    __device__ __forceinline__ void load(const float4& s0, const float4& s1) { st = s0; qp = s1.x; t = s1.y; }
    __device__ __forceinline__ void edge2(float x0, float a0, float x1, float a1) {
        const float d0 = fmaf(qp, x0, a0 * t), d1 = fmaf(qp, x1, a1 * t);
        const float dm = fmaxf(d0, d1);
        if (__any(dm > st.x)) {             // some row of this wave moves its max: rescale those rows
            const float mn = fmaxf(st.x, dm);
            const float sc = exp_acc(st.x - mn);
            st.y *= sc; st.z *= sc; st.w *= sc;
            st.x = mn;
        }
        const float p0 = exp_acc(d0 - st.x), p1 = exp_acc(d1 - st.x);
        st.y += p0 + p1;
        st.z = fmaf(p0, a0, fmaf(p1, a1, st.z));
        st.w = fmaf(p0, x0, fmaf(p1, x1, st.w));
    }
    __device__ __forceinline__ void edge1(float x0, float a0) {
        const float d0 = fmaf(qp, x0, a0 * t);
        if (__any(d0 > st.x)) {
            const float mn = fmaxf(st.x, d0);
            const float sc = exp_acc(st.x - mn);
            st.y *= sc; st.z *= sc; st.w *= sc;
            st.x = mn;
        }
        const float p0 = exp_acc(d0 - st.x);
        st.y += p0;
        st.z = fmaf(p0, a0, st.z);
        st.w = fmaf(p0, x0, st.w);
    }
    __device__ __forceinline__ void store(float4* s0) const { *s0 = st; }
    // o = relu(Wv Z + S bv + u we + Ws x + bs), Z, aux                      (sweep_kernels.hip::Fwd1Op::epilogue)
    __device__ __forceinline__ static void epilogue(const Args& a, int row, const float4& s0, const float4& s1) {
        const float rinv = 1.0f / (s0.y + 1e-16f);
        const float S = s0.y * rinv, un = s0.z * rinv, zn = s0.w * rinv;
        const float x = s1.z;
        float o[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = a.p.bs[c];
            v = fmaf(S, a.p.bv[c], v);
            v = fmaf(un, a.p.we[c], v);
            v = fmaf(a.p.Wv[c], zn, v);
            v = fmaf(a.p.Ws[c], x, v);
            o[c] = fmaxf(v, 0.0f);
        }
        float4* hd = reinterpret_cast<float4*>(a.h + (size_t)row * 16);
        hd[0] = make_float4(o[0], o[1], o[2], o[3]);
        hd[1] = make_float4(o[4], o[5], o[6], o[7]);
        hd[2] = make_float4(o[8], o[9], o[10], o[11]);
        hd[3] = make_float4(o[12], o[13], o[14], o[15]);
        a.Z[row] = zn;
        reinterpret_cast<float4*>(a.aux)[row] = make_float4(un, s0.y > 0.0f ? s0.x : 0.0f, rinv, S);
    }
};

// backward, destination-major: ds_i, dt_i, dq'_i                             (sweep_kernels.hip::BwdDst1Op)
struct BwdDst1T {
    using Args = BwdDst1TiledArgs;
    float4 acc;     // ds, dt, dqp, -
    float qp, gv, t, m, rinv, ge, cc;
    __device__ __forceinline__ static void init_row(const Args& a, int row, bool valid, float4* S0, float4* S1) {
        // S0 = accumulators, S1 = {q', gv, t, rowmax}; {rinv, ge, c} ride in a third table
        *S0 = make_float4(0.f, 0.f, 0.f, 0.f);
        *S1 = valid ? ld4(a.rec + (size_t)row * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ static float4 init_row2(const Args& a, int row, bool valid) {
        return valid ? ld4(a.rec + (size_t)row * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void load(const float4& s0, const float4& s1, const float4& s2) {
        acc = s0; qp = s1.x; gv = s1.y; t = s1.z; m = s1.w; rinv = s2.x; ge = s2.y; cc = s2.z;
    }
    __device__ __forceinline__ void edge1(float x0, float a0) {
        const float l = fmaf(qp, x0, a0 * t);
        const float alpha = exp_acc(l - m) * rinv;
        const float dl = alpha * fmaf(gv, x0, fmaf(a0, ge, cc));
        acc.x += dl;
        acc.y = fmaf(dl, a0, acc.y);
        acc.z = fmaf(dl, x0, acc.z);
    }
    __device__ __forceinline__ void store(float4* s0) const { *s0 = acc; }
    __device__ __forceinline__ static void epilogue(const Args& a, int row, const float4& s0) {
        a.dqp[row] = s0.z;
        reinterpret_cast<float2*>(a.dsdt)[row] = make_float2(s0.x, s0.y);
    }
};

}  // namespace mllp
