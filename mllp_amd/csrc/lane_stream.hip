// lane_stream.hip -- the layer-1 (one input channel) attention sweeps, destination-major, on the lane-per-row streamed copy
// (lane_layout.h), and the device builder of that copy.
//
// Reference: linear_program_methods.py:90-91 (TransformerConv(1, 16, edge_dim=1)) called at :241-242; arithmetic of a row
// as in sweep_kernels.hip::Fwd1Op / BwdDst1Op (scalar_ops.h).
//
// One workgroup of 512 threads per row tile, two workgroups per CU.  For every column block of the tile (one, when the
// instance has at most 20 000 columns): the block's x is staged into LDS with coalesced 16-byte loads, then every lane
// walks ITS row: the wavefront loads one group (4 steps x 64 rows: 512 B of column offsets + 1 KB of values, contiguous,
// four groups ahead in registers), a lane reads x_j with one ds_read_b32 per entry and updates the row's online softmax
// (forward: {m, L, u, Z}) or gradient sums (backward: {ds, dt, dq'}) in its own registers.  No cross-lane traffic, no
// per-row state in LDS, no barrier after the image has landed; the epilogue is the lane's.
// Bound: HBM, 6 bytes per nonzero (+ padding: rows of a wavefront are length-sorted neighbours) -- 3.07 GB at 512 M
// nonzeros = 0.38 ms at 8 TB/s.  A padding entry has column offset L1_PAD and logit -inf: p = 0.
// Deterministic: a row is walked by one lane in CSR order.
#include <algorithm>
#include <chrono>
#include <climits>
#include <vector>

#include "device_utils.h"
#include "internal.h"
#include "lane_layout.h"
#include "scalar_ops.h"

namespace mllp {

namespace {

template <class T>
struct LbBuf {
    T* p = nullptr;
    ~LbBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? 0 : 1; }
};

// ------------------------------------------------------------------------------------------------ builder
// rows of the tile ordered by their number of entries (descending, ties by row id); the tile's column range
__global__ __launch_bounds__(L1_R) void lb_tile_sort(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                      const int* __restrict__ tile_row, int* __restrict__ rows,
                                                      int* __restrict__ tile_col, int* __restrict__ nblk) {
    __shared__ unsigned long long key[L1_R];
    __shared__ int smin[L1_NW], smax[L1_NW];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int r0 = tile_row[t], nr = tile_row[t + 1] - r0;
    const bool valid = tid < nr;
    const int b = valid ? ptr[r0 + tid] : 0, e = valid ? ptr[r0 + tid + 1] : 0;
    int cmin = e > b ? idx[b] : INT_MAX, cmax = e > b ? idx[e - 1] : -1;
    for (int o = 32; o > 0; o >>= 1) {
        cmin = min(cmin, __shfl_xor(cmin, o, 64));
        cmax = max(cmax, __shfl_xor(cmax, o, 64));
    }
    if ((tid & 63) == 0) { smin[tid >> 6] = cmin; smax[tid >> 6] = cmax; }
    key[tid] = (unsigned long long)(valid ? (unsigned)(e - b) + 1u : 0u) << 32 | (unsigned)(L1_R - 1 - tid);
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < L1_NW; ++w) { cmin = min(cmin, smin[w]); cmax = max(cmax, smax[w]); }
        const int c0 = cmax < 0 ? 0 : cmin & ~3;
        tile_col[2 * t] = c0;
        tile_col[2 * t + 1] = cmax;
        nblk[t] = cmax < 0 ? 0 : (cmax - c0) / L1_CB + 1;
    }
    // bitonic sort, descending
    for (int k = 2; k <= L1_R; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int other = tid ^ j;
            if (other > tid) {
                const unsigned long long x = key[tid], y = key[other];
                const bool desc = (tid & k) == 0;
                if (desc ? x < y : x > y) { key[tid] = y; key[other] = x; }
            }
            __syncthreads();
        }
    const unsigned long long kk = key[tid];
    rows[(size_t)t * L1_R + tid] = (kk >> 32) == 0 ? -1 : L1_R - 1 - (int)(unsigned)kk;
}

__device__ __forceinline__ int lb_lower_bound(const int* __restrict__ idx, int lo, int hi, int col) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (idx[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// FILL = false: groups of every (tile, wavefront, block) into whdr[..][1];  FILL = true: the entries
template <bool FILL>
__global__ __launch_bounds__(L1_R) void lb_walk(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                 const float* __restrict__ val, const int* __restrict__ tile_row,
                                                 const int* __restrict__ tile_blk, const int* __restrict__ tile_col,
                                                 const int* __restrict__ rows, int* __restrict__ whdr,
                                                 uint2* __restrict__ offs, float4* __restrict__ vals) {
    const int t = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int tb0 = tile_blk[t], nb = tile_blk[t + 1] - tb0;
    const int local = rows[(size_t)t * L1_R + tid];
    const int r = tile_row[t] + max(local, 0);
    int pos = local >= 0 ? ptr[r] : 0;
    const int end = local >= 0 ? ptr[r + 1] : 0;
    const int c0 = tile_col[2 * t];
    for (int b = 0; b < nb; ++b) {
        const int cb = c0 + b * L1_CB;
        const int stop = b + 1 == nb ? end : lb_lower_bound(idx, pos, end, cb + L1_CB);
        int* h = whdr + ((size_t)tb0 * L1_NW + (size_t)w * nb + b) * 2;
        if constexpr (!FILL) {
            int n = stop - pos;
            for (int o = 32; o > 0; o >>= 1) n = max(n, __shfl_xor(n, o, 64));
            if (lane == 0) h[1] = (n + L1_GS - 1) / L1_GS;
        } else {
            const int g0 = h[0], ng = h[1];
            for (int g = 0; g < ng; ++g) {
                unsigned o[4];
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = pos + 4 * g + k;
                    const bool real = e < stop;
                    o[k] = real ? (unsigned)(idx[e] - cb) : (unsigned)L1_PAD;
                    v[k] = real ? val[e] : 0.0f;
                }
                offs[((size_t)g0 + g) * 64 + lane] = make_uint2(o[0] | o[1] << 16, o[2] | o[3] << 16);
                vals[((size_t)g0 + g) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        pos = stop;
    }
}

__global__ void lb_pad(uint2* __restrict__ offs, float4* __restrict__ vals, long long first) {
    const size_t i = (size_t)first * 64 + (size_t)blockIdx.x * 64 + threadIdx.x;
    offs[i] = make_uint2((unsigned)L1_PAD | (unsigned)L1_PAD << 16, (unsigned)L1_PAD | (unsigned)L1_PAD << 16);
    vals[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------ sweeps
struct LaneDev {
    const int* __restrict__ tile_row;
    const int* __restrict__ tile_blk;
    const int* __restrict__ tile_col;
    const int* __restrict__ rows;
    const int2* __restrict__ whdr;
    const uint2* __restrict__ offs;
    const float4* __restrict__ vals;
    int n_tiles, n_src;
};

#ifndef MLLP_L1_AHEAD
#define MLLP_L1_AHEAD 4
#endif
constexpr int L1_AHEAD = MLLP_L1_AHEAD;     // groups a wavefront holds in registers
static_assert(L1_AHEAD <= L1_PADG, "a wavefront without groups still loads its register set");
constexpr float L1_NINF = -__builtin_inff();

typedef unsigned l1_u2 __attribute__((ext_vector_type(2)));
typedef float l1_f4 __attribute__((ext_vector_type(4)));
struct L1Grp {
    l1_u2 o;
    l1_f4 v;
};

// forward: the row's online softmax, four entries at a time (Fwd1T::edge2 widened; a padding entry has d = -inf)
struct L1Fwd : Fwd1T {
    float4 s1;
    __device__ __forceinline__ void init(const Args& a, int row, bool valid) {
        float4 s0;
        init_row(a, row, valid, &s0, &s1);
        load(s0, s1);
    }
    __device__ __forceinline__ void group(const float (&x)[4], const float (&av)[4], const bool (&pad)[4]) {
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = pad[k] ? L1_NINF : fmaf(qp, x[k], av[k] * t);
        const float dm = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]));
        if (__any(dm > st.x)) {             // some row of this wavefront moves its max: rescale those rows
            const float mn = fmaxf(st.x, dm);
            const float sc = exp_acc(st.x - mn);
            st.y *= sc; st.z *= sc; st.w *= sc;
            st.x = mn;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float p = exp_acc(d[k] - st.x);
            st.y += p;
            st.z = fmaf(p, av[k], st.z);
            st.w = fmaf(p, x[k], st.w);
        }
    }
    __device__ __forceinline__ void finish(const Args& a, int row) const { epilogue(a, row, st, s1); }
};

// backward, destination-major
struct L1Bwd : BwdDst1T {
    __device__ __forceinline__ void init(const Args& a, int row, bool valid) {
        float4 s0, s1;
        init_row(a, row, valid, &s0, &s1);
        load(s0, s1, init_row2(a, row, valid));
    }
    __device__ __forceinline__ void group(const float (&x)[4], const float (&av)[4], const bool (&pad)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float l = pad[k] ? L1_NINF : fmaf(qp, x[k], av[k] * t);
            const float alpha = exp_acc(l - m) * rinv;
            const float dl = alpha * fmaf(gv, x[k], fmaf(av[k], ge, cc));
            acc.x += dl;
            acc.y = fmaf(dl, av[k], acc.y);
            acc.z = fmaf(dl, x[k], acc.z);
        }
    }
    __device__ __forceinline__ void finish(const Args& a, int row) const { epilogue(a, row, acc); }
};

template <class Op>
__global__ __launch_bounds__(L1_R, 4) void lane1_kernel(LaneDev t, typename Op::Args a, int aligned) {
    __shared__ __attribute__((aligned(16))) float img[L1_CB + 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = xcd_tile(blockIdx.x, t.n_tiles);
    const int r0 = t.tile_row[tile];
    const int local = t.rows[(size_t)tile * L1_R + tid];
    const bool valid = local >= 0;
    const int row = r0 + max(local, 0);
    const int tb0 = t.tile_blk[tile], nb = t.tile_blk[tile + 1] - tb0;
    const int c_first = t.tile_col[2 * tile], c_last = t.tile_col[2 * tile + 1];
    Op op;
    op.init(a, row, valid);
    if (tid < 4) img[L1_CB + tid] = 0.0f;
    const char* ib = reinterpret_cast<const char*>(img);
    const float* __restrict__ X = a.X;

    for (int b = 0; b < nb; ++b) {
        const int2 h = t.whdr[(size_t)tb0 * L1_NW + (size_t)w * nb + b];
        const int g0 = __builtin_amdgcn_readfirstlane(h.x), ng = __builtin_amdgcn_readfirstlane(h.y);
        const l1_u2* po = reinterpret_cast<const l1_u2*>(t.offs) + (size_t)g0 * 64 + lane;
        const l1_f4* pv = reinterpret_cast<const l1_f4*>(t.vals) + (size_t)g0 * 64 + lane;
        const int glast = max(ng - 1, 0);           // the read-ahead stops at the last group (re-read from the cache, never used)
        L1Grp ring[L1_AHEAD];
#pragma unroll
        for (int j = 0; j < L1_AHEAD; ++j) {
            ring[j].o = __builtin_nontemporal_load(po + 64 * min(j, glast));
            ring[j].v = __builtin_nontemporal_load(pv + 64 * min(j, glast));
        }
        if (b > 0) __syncthreads();                 // everybody has left the previous image
        {
            const int cb = c_first + b * L1_CB;
            const int width = min(L1_CB, c_last + 1 - cb);
            if (aligned) {
                for (int i = tid * 4; i < width; i += L1_R * 4) {
                    const int c = cb + i;
                    float4 v;
                    if (c + 3 < t.n_src) v = ld4(X + c);
                    else v = make_float4(X[min(c, t.n_src - 1)], X[min(c + 1, t.n_src - 1)], X[min(c + 2, t.n_src - 1)], X[min(c + 3, t.n_src - 1)]);
                    *reinterpret_cast<float4*>(img + i) = v;
                }
            } else {
                for (int i = tid; i < width; i += L1_R) img[i] = X[cb + i];
            }
        }
        __syncthreads();
        for (int g = 0; g < ng; g += L1_AHEAD) {
#pragma unroll
            for (int j = 0; j < L1_AHEAD; ++j) {
                const L1Grp c = ring[j];
                ring[j].o = __builtin_nontemporal_load(po + 64 * min(g + L1_AHEAD + j, glast));
                ring[j].v = __builtin_nontemporal_load(pv + 64 * min(g + L1_AHEAD + j, glast));
                if (g + j < ng) {
                    const unsigned o[4] = {c.o.x & 0xFFFFu, c.o.x >> 16, c.o.y & 0xFFFFu, c.o.y >> 16};
                    const float av[4] = {c.v.x, c.v.y, c.v.z, c.v.w};
                    float x[4];
                    bool pad[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        x[k] = *reinterpret_cast<const float*>(ib + o[k] * 4u);
                        pad[k] = o[k] == (unsigned)L1_PAD;
                    }
                    op.group(x, av, pad);
                }
            }
        }
    }
    if (valid) op.finish(a, row);
}

std::vector<int> lane_tiles(const std::vector<int64_t>& seg, int64_t n_dst) {
    std::vector<int> tr;
    int64_t covered = 0;
    for (size_t i = 0; i + 1 < seg.size(); ++i) {
        for (int64_t r = seg[i]; r < seg[i + 1]; r += L1_R) tr.push_back((int)r);
        covered = seg[i + 1];
    }
    for (int64_t r = covered; r < n_dst; r += L1_R) tr.push_back((int)r);     // (rows behind the last instance: none in a valid graph)
    tr.push_back((int)n_dst);
    return tr;
}

LaneDev lane_dev(const LaneCopy& lc, int n_src) {
    LaneDev d;
    d.tile_row = lc.tile_row; d.tile_blk = lc.tile_blk; d.tile_col = lc.tile_col; d.rows = lc.rows;
    d.whdr = reinterpret_cast<const int2*>(lc.whdr);
    d.offs = reinterpret_cast<const uint2*>(lc.offs);
    d.vals = reinterpret_cast<const float4*>(lc.vals);
    d.n_tiles = lc.n_tiles; d.n_src = n_src;
    return d;
}

}  // namespace

void lane_copy_free(LaneCopy& lc) {
    for (void* p : {(void*)lc.tile_row, (void*)lc.tile_blk, (void*)lc.tile_col, (void*)lc.rows, (void*)lc.whdr, (void*)lc.offs, (void*)lc.vals})
        if (p) (void)hipFree(p);
    lc = LaneCopy();
}

int build_lane_copy(const Orient& o, int64_t nnz, const std::vector<int64_t>& seg, LaneCopy& lc, hipStream_t s) {
    const std::vector<int> tile_row = lane_tiles(seg, o.n_dst);
    const int n_tiles = (int)tile_row.size() - 1;
    if (n_tiles <= 0) return MLLP_OK;
    LbBuf<int> nblk;
    if (nblk.alloc(n_tiles)) return fail(MLLP_ENOMEM, "lane copy: hipMalloc failed");
    MLLP_HIP_TRY(hipMalloc((void**)&lc.tile_row, ((size_t)n_tiles + 1) * 4));
    MLLP_HIP_TRY(hipMalloc((void**)&lc.tile_blk, ((size_t)n_tiles + 1) * 4));
    MLLP_HIP_TRY(hipMalloc((void**)&lc.tile_col, (size_t)n_tiles * 8));
    MLLP_HIP_TRY(hipMalloc((void**)&lc.rows, (size_t)n_tiles * L1_R * 4));
    MLLP_HIP_TRY(hipMemcpyAsync(lc.tile_row, tile_row.data(), ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lb_tile_sort, dim3(n_tiles), dim3(L1_R), 0, s, o.ptr, o.idx, lc.tile_row, lc.rows, lc.tile_col, nblk.p);
    std::vector<int> h_nb(n_tiles), h_tb((size_t)n_tiles + 1, 0);
    MLLP_HIP_TRY(hipMemcpyAsync(h_nb.data(), nblk.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    int64_t n_tb = 0;
    for (int t = 0; t < n_tiles; ++t) {
        h_tb[t] = (int)n_tb;
        n_tb += h_nb[t];
        if (n_tb >= (1 << 26)) return fail(MLLP_ERANGE, "lane copy: too many (tile, block) pairs");
    }
    h_tb[n_tiles] = (int)n_tb;
    MLLP_HIP_TRY(hipMemcpyAsync(lc.tile_blk, h_tb.data(), ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s));
    const size_t n_h = (size_t)n_tb * L1_NW;
    MLLP_HIP_TRY(hipMalloc((void**)&lc.whdr, std::max<size_t>(n_h, 1) * 8));
    hipLaunchKernelGGL(lb_walk<false>, dim3(n_tiles), dim3(L1_R), 0, s, o.ptr, o.idx, o.val, lc.tile_row, lc.tile_blk, lc.tile_col,
                       lc.rows, lc.whdr, (uint2*)nullptr, (float4*)nullptr);
    std::vector<int> h_h(std::max<size_t>(n_h, 1) * 2, 0);
    if (n_h) MLLP_HIP_TRY(hipMemcpyAsync(h_h.data(), lc.whdr, n_h * 8, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    int64_t n_groups = 0;
    for (size_t i = 0; i < n_h; ++i) {
        h_h[2 * i] = (int)n_groups;
        n_groups += h_h[2 * i + 1];
        if (n_groups >= (1ll << 31) / 64) return fail(MLLP_ERANGE, "lane copy: the stream exceeds int32 indexing");
    }
    if (n_h) MLLP_HIP_TRY(hipMemcpyAsync(lc.whdr, h_h.data(), n_h * 8, hipMemcpyHostToDevice, s));
    MLLP_HIP_TRY(hipMalloc((void**)&lc.offs, (size_t)(n_groups + L1_PADG) * 64 * 8));
    MLLP_HIP_TRY(hipMalloc((void**)&lc.vals, (size_t)(n_groups + L1_PADG) * 64 * 16));
    hipLaunchKernelGGL(lb_walk<true>, dim3(n_tiles), dim3(L1_R), 0, s, o.ptr, o.idx, o.val, lc.tile_row, lc.tile_blk, lc.tile_col,
                       lc.rows, lc.whdr, reinterpret_cast<uint2*>(lc.offs), reinterpret_cast<float4*>(lc.vals));
    hipLaunchKernelGGL(lb_pad, dim3(L1_PADG), dim3(64), 0, s, reinterpret_cast<uint2*>(lc.offs), reinterpret_cast<float4*>(lc.vals),
                       (long long)n_groups);
    MLLP_HIP_TRY(hipGetLastError());
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    lc.n_tiles = n_tiles;
    lc.n_tb = (int)n_tb;
    lc.n_groups = n_groups;
    lc.nnz = nnz;
    return MLLP_OK;
}

int launch_fwd1_lane(const LaneCopy& lc, int n_dst, int n_src, const float* conv_params, const ConvWs& w, const float* x_src,
                     const float* x_dst, float* h_out, hipStream_t s) {
    (void)n_dst;
    if (lc.n_tiles == 0) return MLLP_OK;
    Fwd1TiledArgs a;
    a.X = x_src; a.xd = x_dst; a.derived = w.derived; a.p = conv_params_at(conv_params, 1);
    a.h = h_out; a.Z = w.Z; a.aux = w.aux;
    hipLaunchKernelGGL(lane1_kernel<L1Fwd>, dim3((unsigned)lc.n_tiles), dim3(L1_R), 0, s, lane_dev(lc, n_src), a,
                       (int)((reinterpret_cast<uintptr_t>(x_src) & 15) == 0));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "fwd1_lane");
}

int launch_bwddst1_lane(const LaneCopy& lc, int n_dst, int n_src, const ConvWs& w, const float* x_src, hipStream_t s) {
    (void)n_dst;
    if (lc.n_tiles == 0) return MLLP_OK;
    BwdDst1TiledArgs a{x_src, w.rec, w.dqp, w.dsdt};
    hipLaunchKernelGGL(lane1_kernel<L1Bwd>, dim3((unsigned)lc.n_tiles), dim3(L1_R), 0, s, lane_dev(lc, n_src), a,
                       (int)((reinterpret_cast<uintptr_t>(x_src) & 15) == 0));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLLP_OK : hip_fail(e, "bwddst1_lane");
}

}  // namespace mllp
