// device_utils.h -- gfx950 device helpers: DPP cross-lane reductions inside 16-lane rows and
// 64-lane wavefronts, XCD-aware tile remap, small vector helpers.
#pragma once
#include <hip/hip_runtime.h>

namespace mllp {

constexpr float NEG_BIG = -3.0e38f;  // "minus infinity" sentinel that never produces inf - inf

// ---- DPP (data-parallel primitives): lane permutations inside a row of 16 lanes ------------------
// dpp_ctrl encodings (CDNA ISA): quad_perm = 0x00..0xFF, row_mirror = 0x140, row_half_mirror = 0x141
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// all-reduce over the 16 lanes of a DPP row (every lane ends with the result)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2] : lane ^ 1
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1] : lane ^ 2
    v += dpp_mov<0x141>(v);  // row_half_mirror     : quad 0 <-> quad 1 inside each 8
    v += dpp_mov<0x140>(v);  // row_mirror          : half 0 <-> half 1 inside the 16
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

// G = 16: one DPP row.  G = 64: the whole wavefront (two more butterfly steps through the LDS crossbar).
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    v = row16_sum(v);
    if (G == 64) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
    }
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
    v = row16_max(v);
    if (G == 64) {
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
    }
    return v;
}

// value v[c] for a lane-dependent c in [0,16): 15 v_cndmask.  The operands are copied to prvalues first:
// `cond ? v[a] : v[b]` on lvalues makes clang select the ADDRESS, which turns the register array into a
// dynamically indexed alloca (scratch / LDS-promoted) -- measured: 16 KB of LDS per workgroup.
__device__ __forceinline__ float select16(const float (&v)[16], int c) {
    float a[8], b[4], d[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float hi = v[2 * i + 1], lo = v[2 * i];
        a[i] = (c & 1) ? hi : lo;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float hi = a[2 * i + 1], lo = a[2 * i];
        b[i] = (c & 2) ? hi : lo;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float hi = b[2 * i + 1], lo = b[2 * i];
        d[i] = (c & 4) ? hi : lo;
    }
    const float hi = d[1], lo = d[0];
    return (c & 8) ? hi : lo;
}

__device__ __forceinline__ void load_row16(const float* __restrict__ p, float (&r)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 t = q[i];
        r[4 * i + 0] = t.x;
        r[4 * i + 1] = t.y;
        r[4 * i + 2] = t.z;
        r[4 * i + 3] = t.w;
    }
}

__device__ __forceinline__ float dot16(const float (&a)[16], const float (&b)[16], float init) {
    float s = init;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = fmaf(a[i], b[i], s);
    return s;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void fma4(float s, const float4& x, float4& acc) {
    acc.x = fmaf(s, x.x, acc.x);
    acc.y = fmaf(s, x.y, acc.y);
    acc.z = fmaf(s, x.z, acc.z);
    acc.w = fmaf(s, x.w, acc.w);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// XCD-aware, bijective block -> tile remap (guide T1): blocks b and b+8 share an XCD (and its L2),
// so give each XCD a contiguous range of tiles; consecutive tiles then gather the same source rows
// from one L2.  Valid for any number of tiles.
__device__ __forceinline__ int xcd_tile(int b, int n) {
    int q = n >> 3, r = n & 7, x = b & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

// all-reduce over the quads of a G-lane group, for values that are replicated or owned per `part`
// (lanes 4 apart hold the same channel): G = 16 -> rotate by 8 and 4 inside the DPP row
template <int G>
__device__ __forceinline__ float quads_sum(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    if (G == 64) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
    }
    return v;
}
template <int G>
__device__ __forceinline__ float quads_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    if (G == 64) {
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
    }
    return v;
}
// sum over the 4 lanes of a quad (every lane gets the total)
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
// max over the 4 lanes of a quad (every lane gets it)
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    return v;
}
// value held by lane J of the own quad
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {
    return dpp_mov<J * 0x55>(v);   // quad_perm [J,J,J,J]
}
// all 16 channels from the 4 per-part float4 of a quad
__device__ __forceinline__ void quad_allgather(const float4& v, float (&r)[16]) {
    r[0] = quad_bcast<0>(v.x);  r[1] = quad_bcast<0>(v.y);  r[2] = quad_bcast<0>(v.z);  r[3] = quad_bcast<0>(v.w);
    r[4] = quad_bcast<1>(v.x);  r[5] = quad_bcast<1>(v.y);  r[6] = quad_bcast<1>(v.z);  r[7] = quad_bcast<1>(v.w);
    r[8] = quad_bcast<2>(v.x);  r[9] = quad_bcast<2>(v.y);  r[10] = quad_bcast<2>(v.z); r[11] = quad_bcast<2>(v.w);
    r[12] = quad_bcast<3>(v.x); r[13] = quad_bcast<3>(v.y); r[14] = quad_bcast<3>(v.z); r[15] = quad_bcast<3>(v.w);
}

// exp(x) for x <= ~0 with ~1-2 ulp: hardware 2^t (v_exp_f32) on t = x*log2(e), with the rounding error
// of that product (and of the constant) fed back as a first-order correction.  The plain
// v_exp_f32(x * log2e) loses |x| * 6e-8 relative accuracy, too much against the 1e-5 parity budget.
__device__ __forceinline__ float exp_acc(float x) {
    const float L2E_HI = 1.44269502163e+00f, L2E_LO = 1.92596299112e-08f, LN2 = 0.693147180560f;
    x = fmaxf(x, -150.0f);   // exp(-150) is exactly 0 in fp32; keeps NEG_BIG sentinels away from inf - inf
    const float t = x * L2E_HI;
    float r = fmaf(x, L2E_HI, -t);
    r = fmaf(x, L2E_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * LN2, e);
}

}  // namespace mllp
