// node_bodies.h -- device bodies of the single-workgroup kernels at the two ends of a training step (folded weights,
// statistics -> gradients, Adam), shared by node_kernels.hip (one launch each) and fused_kernels.hip (the fused path's
// tail runs them inside ONE launch: fused_tail_kernel).  Included by .hip files only.
#pragma once
#include "device_utils.h"
#include "internal.h"

namespace mllp {

// param_prep: one workgroup (the first BLOCK threads) per conv
__device__ __forceinline__ void param_prep_body(const ConvParams& p, int cin, float* __restrict__ D) {
    const int tid = threadIdx.x;
    const int k = tid >> 4, d = tid & 15;
    float pq = 0.0f, wst = 0.0f, wvt = 0.0f;
    if (k < cin && d < cin) {
        for (int c = 0; c < 16; ++c) pq = fmaf(p.Wk[c * cin + k], p.Wq[c * cin + d], pq);
        pq *= 0.25f;
    }
    if (k < cin) {            // here: k = input channel, d = output channel
        wst = p.Ws[d * cin + k];
        wvt = p.Wv[d * cin + k];
    }
    D[OFF_PQ + k * 16 + d] = pq;
    D[OFF_PQT + d * 16 + k] = pq;
    D[OFF_WST + k * 16 + d] = wst;
    D[OFF_WVT + k * 16 + d] = wvt;
    if (tid < 16) {
        float pq0 = 0.0f, pt = 0.0f, pb = 0.0f;
        if (tid < cin) {
            for (int c = 0; c < 16; ++c) {
                pq0 = fmaf(p.Wk[c * cin + tid], p.bq[c], pq0);
                pt = fmaf(p.we[c], p.Wq[c * cin + tid], pt);
                pb = fmaf(p.bk[c], p.Wq[c * cin + tid], pb);
            }
        }
        float pt0 = 0.0f;
        if (tid == 0)
            for (int c = 0; c < 16; ++c) pt0 = fmaf(p.bq[c], p.we[c], pt0);
        D[OFF_PQ0 + tid] = 0.25f * pq0;
        D[OFF_PT + tid] = 0.25f * pt;
        D[OFF_PB + tid] = 0.25f * pb;
        D[OFF_PT0 + tid] = 0.25f * pt0;
    }
}


// finalize_conv: fixed-order sum of the per-workgroup partial tiles, then the small matrix algebra
// (oracle/spmm_form.py::conv_bwd "grads = {...}").  One workgroup of 1024 threads.
__device__ __forceinline__ void finalize_conv_body(int cin, const ConvParams& p, const float* __restrict__ stats,
                                                   int nblk, float* __restrict__ grads) {
    __shared__ float P[4][STAT_FLOATS];
    __shared__ float T[STAT_FLOATS];
    {
        const int slice = threadIdx.x >> 8, col = threadIdx.x & 255;
        float v[STAT_TILES];
#pragma unroll
        for (int i = 0; i < STAT_TILES; ++i) v[i] = 0.0f;
        // fixed order b = slice, slice + 4, ...; four partials (28 independent loads) in flight per thread: the sum is
        // bound by memory round trips, not by the additions
        int b = slice;
        for (; b + 12 < nblk; b += 16) {
            float t0[STAT_TILES], t1[STAT_TILES], t2[STAT_TILES], t3[STAT_TILES];
            const float* src = stats + (size_t)b * STAT_FLOATS + col;
#pragma unroll
            for (int i = 0; i < STAT_TILES; ++i) {
                t0[i] = src[i * 256];
                t1[i] = src[(size_t)4 * STAT_FLOATS + i * 256];
                t2[i] = src[(size_t)8 * STAT_FLOATS + i * 256];
                t3[i] = src[(size_t)12 * STAT_FLOATS + i * 256];
            }
#pragma unroll
            for (int i = 0; i < STAT_TILES; ++i) v[i] = (((v[i] + t0[i]) + t1[i]) + t2[i]) + t3[i];
        }
        for (; b < nblk; b += 4) {
            const float* src = stats + (size_t)b * STAT_FLOATS + col;
#pragma unroll
            for (int i = 0; i < STAT_TILES; ++i) v[i] += src[i * 256];
        }
#pragma unroll
        for (int i = 0; i < STAT_TILES; ++i) P[slice][i * 256 + col] = v[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < STAT_FLOATS; i += 1024) T[i] = (P[0][i] + P[1][i]) + (P[2][i] + P[3][i]);
    __syncthreads();
    if (threadIdx.x >= BLOCK) return;
    const float* T0 = T; const float* T1 = T + 256; const float* T2 = T + 512; const float* T3 = T + 768;
    const float* T4 = T + 1024; const float* T5 = T + 1280; const float* T6 = T + 1536;
    float* gWk = grads;
    float* gbk = gWk + 16 * cin;
    float* gWq = gbk + 16;
    float* gbq = gWq + 16 * cin;
    float* gWv = gbq + 16;
    float* gbv = gWv + 16 * cin;
    float* gwe = gbv + 16;
    float* gWs = gwe + 16;
    float* gbs = gWs + 16 * cin;
    const int c = threadIdx.x >> 4, j = threadIdx.x & 15;
    if (j < cin) {
        gWs[c * cin + j] = T0[c * 16 + j];
        gWv[c * cin + j] = T1[c * 16 + j];
        float wk = p.bq[c] * T4[j * 16];           // bq[c] * s_dqp[k=j]
        float wq = fmaf(p.bk[c], T5[j], p.we[c] * T5[16 + j]);
        for (int d = 0; d < cin; ++d) wk = fmaf(p.Wq[c * cin + d], T3[j * 16 + d], wk);
        for (int k = 0; k < cin; ++k) wq = fmaf(p.Wk[c * cin + k], T3[k * 16 + j], wq);
        gWk[c * cin + j] = 0.25f * wk;
        gWq[c * cin + j] = 0.25f * wq;
    }
    if (j == 0) {
        gbs[c] = T2[c * 16 + 0];
        gbv[c] = T2[c * 16 + 1];
        float bk = p.bq[c] * T6[0], we = p.bq[c] * T6[16];
        float bq = fmaf(p.bk[c], T6[0], p.we[c] * T6[16]);
        for (int d = 0; d < cin; ++d) {
            bk = fmaf(p.Wq[c * cin + d], T5[d], bk);
            we = fmaf(p.Wq[c * cin + d], T5[16 + d], we);
        }
        for (int k = 0; k < cin; ++k) bq = fmaf(p.Wk[c * cin + k], T4[k * 16], bq);
        gbk[c] = 0.25f * bk;
        gwe[c] = T2[c * 16 + 2] + 0.25f * we;
        gbq[c] = 0.25f * bq;
    }
}


// Adam (torch.optim.Adam defaults apart from lr): one workgroup, the step counter lives on the device so that a captured
// hipGraph can be replayed.  state = {step, lr, beta1, beta2}
// elements [0, n) of the given (offset) pointers with the step's scalars; `step` = the incremented step count
__device__ __forceinline__ void adam_slice(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, float step, float lr, float b1, float b2, float eps,
                                           float gscale, int n) {
    const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
    const float step_size = lr / bc1, rs_bc2 = 1.0f / sqrtf(bc2);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rs_bc2 + eps;
        p[i] -= step_size * (mi / denom);
    }
}
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, float* __restrict__ state, float eps, float gscale, int n) {
    const float step = state[0] + 1.0f, lr = state[1], b1 = state[2], b2 = state[3];
    __syncthreads();
    adam_slice(p, g, m, v, step, lr, b1, b2, eps, gscale, n);
    if (threadIdx.x == 0) state[0] = step;
}

}  // namespace mllp
