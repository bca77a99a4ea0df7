// tiled_build.hip -- device builder of the LDS-tiled copies (variants 0-4, layout: include/mllp_hip.h,
// mllp_graph_attach_tiled; reference builder: mllp_amd/graph.py::build_tiled_arrays, torch ops, 5-15 s per copy at 512 M
// nonzeros).  Replaces, once per batch, orientation and kernel family, what the reference redoes every step
// (build_graph_from_weights_sets, linear_program_methods.py:89-103).  Counting passes + one placement pass, no sort of
// the nonzeros, well under a second per copy:
//   tb_tile_range   per row tile: first / last column block its rows touch (a tile lists the whole range)
//   tb_count        per (tile, block, row): entries of the row in the block, CSR position of the first one
//   tb_sort         per (tile, block): rows by entry count, descending, ties by row (variants 0-3; variant 4 keeps the
//                   rows in place); offsets of the sorted positions inside the (tile, block) segment
//   tb_fill         entries.  Variants 0, 1 (a quad of lanes per row, 16 sorted positions per wavefront) and 4 (a lane
//                   per row): the four rows that read in the same LDS cycle of a ds_read_b128 are ordered JOINTLY over
//                   (column mod 4), one thread per such team; variants 2, 3 (and any matrix with a (row, block) run of
//                   more than 512 entries): round-robin over (column mod 4) per row, starting at the class of the row's
//                   quad; variant 4 stores the entries of every 64-row chunk by step.
// Every array comes out bit-identical to graph.py::build_tiled_arrays (tests/test_hip_parity.py).
// The arrays are library-owned (freed on rebuild / detach / destroy).
#include <algorithm>
#include <climits>
#include <vector>

#include "internal.h"

namespace mllp {

void tiled_free(Tiled& tl) {
    if (tl.owned) {
        (void)hipFree((void*)tl.tile_blk);
        (void)hipFree((void*)tl.blk_id);
        (void)hipFree((void*)tl.ptr2);
        (void)hipFree((void*)tl.perm);
        (void)hipFree((void*)tl.ent);
    }
    tl = Tiled();
}

namespace {

constexpr int TEAMS[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

__global__ void tb_tile_range(const int* __restrict__ ptr, const int* __restrict__ idx, int n_dst, int R, int CB,
                              int* __restrict__ lo, int* __restrict__ hi) {
    __shared__ int s_lo, s_hi;
    if (threadIdx.x == 0) { s_lo = INT_MAX; s_hi = -1; }
    __syncthreads();
    const int r = blockIdx.x * R + threadIdx.x;
    if ((int)threadIdx.x < R && r < n_dst) {
        const int b = ptr[r], e = ptr[r + 1];
        if (e > b) {
            atomicMin(&s_lo, idx[b] / CB);
            atomicMax(&s_hi, idx[e - 1] / CB);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { lo[blockIdx.x] = s_lo; hi[blockIdx.x] = s_hi; }
}

__global__ void tb_count(const int* __restrict__ ptr, const int* __restrict__ idx, int n_dst, int R, int CB,
                         const int* __restrict__ lo, const int* __restrict__ tile_blk, int* __restrict__ cnt,
                         int* __restrict__ start) {
    const int t = blockIdx.x, r = t * R + threadIdx.x;
    if ((int)threadIdx.x >= R || r >= n_dst) return;
    const int l = lo[t], tb0 = tile_blk[t];
    int last = -1, c = 0;
    size_t slot = 0;
    for (int e = ptr[r]; e < ptr[r + 1]; ++e) {
        const int b = idx[e] / CB;
        if (b != last) {
            if (c) cnt[slot] = c;
            last = b;
            c = 0;
            slot = (size_t)(tb0 + b - l) * R + threadIdx.x;
            start[slot] = e;
        }
        ++c;
    }
    if (c) cnt[slot] = c;
}

// perm[tb][k] = row at sorted position k; off[tb][k] = offset of position k inside the segment; seg[tb] = its length
__global__ void tb_sort(const int* __restrict__ cnt, int R, int keep_order, int* __restrict__ perm, int* __restrict__ off,
                        int* __restrict__ seg, int* __restrict__ longest) {
    extern __shared__ int sh[];          // c[R], sorted counts / scan[R]
    int* c = sh;
    int* sc = sh + R;
    const size_t tb = blockIdx.x;
    const int r = threadIdx.x;
    const int mine = cnt[tb * R + r];
    c[r] = mine;
    __syncthreads();
    int rank = r;
    if (!keep_order) {
        rank = 0;
        for (int k = 0; k < R; ++k) {
            const int o = c[k];
            rank += (o > mine || (o == mine && k < r)) ? 1 : 0;
        }
    }
    perm[tb * R + rank] = r;
    sc[rank] = mine;
    __syncthreads();
    // exclusive scan of the sorted counts (R <= 512: Hillis-Steele in LDS)
    int v = sc[r];
    for (int d = 1; d < R; d <<= 1) {
        const int add = r >= d ? sc[r - d] : 0;
        __syncthreads();
        v += add;
        sc[r] = v;
        __syncthreads();
    }
    off[tb * R + r] = r ? sc[r - 1] : 0;
    if (r == R - 1) seg[tb] = v;
    if (r == 0) {                        // longest (row, block) run: the joint entry order is used up to 512 (as graph.py)
        int m = 0;
        for (int k = 0; k < R; ++k) m = max(m, c[k]);
        atomicMax(longest, m);
    }
}

__global__ void tb_ptr2(const int* __restrict__ off, const int* __restrict__ seg_start, int R, long long n,
                        int nnz, int* __restrict__ ptr2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ptr2[i] = seg_start[i / R] + off[i];
    if (i == n) ptr2[n] = nnz;
}

// thread = sorted position k of the (tile, block).  joint = 1 (variants 0, 1): the threads of position 16 b + TEAMS[t][0]
// order the four rows of their team jointly, the others return.  chunked = 1 (variant 4): step-major inside 64-row chunks.
__global__ void tb_fill(const int* __restrict__ idx, const float* __restrict__ val, const int* __restrict__ blk_id,
                        const int* __restrict__ cnt, const int* __restrict__ start, const int* __restrict__ perm,
                        const int* __restrict__ ptr2, int R, int CB, int item_bytes, int joint, int chunked,
                        int2* __restrict__ ent) {
    extern __shared__ int sh[];          // len[R] by position (variant 4: by row = position)
    const size_t tb = blockIdx.x;
    const int k = threadIdx.x;
    const int row = perm[tb * R + k];
    const int len = cnt[tb * R + row];
    sh[k] = len;
    __syncthreads();
    const int c0 = blk_id[tb] * CB;
    if (joint) {
        // the four positions of this thread's team (the thread of the team's first position does the work)
        int pos4[4];
        if (chunked) {      // a lane per row: lanes of one ds_read_b128 lane group that read with the same rotation (graph.py::LANE_GROUPS)
            const int lane = k & 63, c64 = k & ~63;
            if ((lane & 31) >= 8) return;
            const int r = lane & 3, h = lane & 32;
            if ((lane & 7) < 4) { pos4[0] = r; pos4[1] = 12 + r; pos4[2] = 20 + r; pos4[3] = 24 + r; }
            else { pos4[0] = 4 + r; pos4[1] = 8 + r; pos4[2] = 16 + r; pos4[3] = 28 + r; }
#pragma unroll
            for (int i = 0; i < 4; ++i) pos4[i] += c64 + h;
        } else {
            const int q = k & 15, b16 = k & ~15;
            int tm = -1;
            for (int t = 0; t < 4; ++t)
                if (TEAMS[t][0] == q) tm = t;
            if (tm < 0) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) pos4[i] = b16 + TEAMS[tm][i];
        }
        int beg[4], rem[4], nxt[4][4], cl[4][4], base[4];
        int maxlen = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = pos4[i];
            const int rr = perm[tb * R + kk];
            beg[i] = start[tb * R + rr];
            rem[i] = sh[kk];
            base[i] = chunked ? ptr2[tb * R + (kk & ~63)] : ptr2[tb * R + kk];
#pragma unroll
            for (int c = 0; c < 4; ++c) { nxt[i][c] = 0; cl[i][c] = 0; }
            for (int e = 0; e < rem[i]; ++e) {
                const int c = (idx[beg[i] + e] - c0) & 3;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) cl[i][cc] += (cc == c) ? 1 : 0;
            }
            maxlen = max(maxlen, rem[i]);
        }
        for (int p = 0; p < maxlen; ++p) {
            unsigned used = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i != ((j + p) & 3)) continue;
                    if (rem[i] == 0) continue;
                    int pick = -1, pick_any = -1, best = 0, best_any = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ck = cl[i][c];
                        if (ck == 0) continue;
                        if (pick_any < 0 || ck > best_any) { pick_any = c; best_any = ck; }
                        if (!(used >> c & 1u) && (pick < 0 || ck > best)) { pick = c; best = ck; }
                    }
                    if (pick < 0) pick = pick_any;
                    int e = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) e = (c == pick) ? nxt[i][c] : e;
                    while (((idx[beg[i] + e] - c0) & 3) != pick) ++e;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c == pick) { nxt[i][c] = e + 1; cl[i][c] -= 1; }
                    rem[i] -= 1;
                    used |= 1u << pick;
                    int dest = base[i] + p;
                    if (chunked) {       // step p of lane l: behind steps < p of every lane of the chunk and step p of the lanes < l
                        const int c64 = pos4[i] & ~63;
                        dest = base[i];
                        for (int l = 0; l < 64; ++l) {
                            const int ll = sh[c64 + l];
                            dest += min(ll, p) + ((c64 + l < pos4[i] && ll > p) ? 1 : 0);
                        }
                    }
                    ent[dest] = make_int2((idx[beg[i] + e] - c0) * item_bytes, __float_as_int(val[beg[i] + e]));
                }
            }
        }
        return;
    }
    // per row: round-robin over the classes, starting at the class of the row's quad inside its lane group
    if (len == 0) return;
    const int bg = start[tb * R + row];
    int cc[4] = {0, 0, 0, 0};
    for (int e = 0; e < len; ++e) {
        const int c = (idx[bg + e] - c0) & 3;
#pragma unroll
        for (int x = 0; x < 4; ++x) cc[x] += (x == c) ? 1 : 0;
    }
    const int g = (k & 7) >> 1;
    int rot[4];                          // count of the class with rotated index x: class (x + g) & 3
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        int v = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y) v = (y == ((x + g) & 3)) ? cc[y] : v;
        rot[x] = v;
    }
    int seen[4] = {0, 0, 0, 0};
    const int chunk0 = chunked ? (k & ~63) : 0;
    const int base = chunked ? ptr2[tb * R + chunk0] : ptr2[tb * R + k];
    for (int e = 0; e < len; ++e) {
        const int col = idx[bg + e] - c0, c = col & 3, x = (c - g) & 3;
        int rho = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            if (y == c) { rho = seen[y]; seen[y] += 1; }
        }
        int pos = 0;                     // entries with a smaller (rank in class, rotated class)
#pragma unroll
        for (int y = 0; y < 4; ++y) pos += min(rot[y], rho) + ((y < x && rot[y] > rho) ? 1 : 0);
        int dest = base + pos;
        if (chunked) {                   // entry `pos` of lane l = k - chunk0: behind steps < pos of every lane and step pos of lanes < l
            dest = base;
            for (int l = 0; l < 64 && chunk0 + l < R; ++l) {
                const int ll = sh[chunk0 + l];
                dest += min(ll, pos) + ((chunk0 + l < k && ll > pos) ? 1 : 0);
            }
        }
        ent[dest] = make_int2(col * item_bytes, __float_as_int(val[bg + e]));
    }
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? 0 : 1; }
};

}  // namespace

int build_tiled_device(const Orient& o, int64_t nnz, int variant, Tiled& out, hipStream_t s) {
    int R, CB, CAP;
    tiled_geometry(variant, &R, &CB, &CAP);
    const int item_bytes = variant == 2 ? 160 : variant == 3 ? 4 : 64;
    const int n_tiles = (int)(((int64_t)o.n_dst + R - 1) / R);
    if (nnz == 0 || n_tiles == 0) return fail(MLLP_EINVAL, "tiled copy: the matrix has no nonzeros");
    DevBuf<int> lo, hi, cnt, start, off, seg, seg_start, longest;
    if (lo.alloc(n_tiles) || hi.alloc(n_tiles)) return fail(MLLP_ENOMEM, "tiled copy: hipMalloc failed");
    const int T = ((R + 63) / 64) * 64;
    hipLaunchKernelGGL(tb_tile_range, dim3(n_tiles), dim3(T), 0, s, o.ptr, o.idx, o.n_dst, R, CB, lo.p, hi.p);
    std::vector<int> h_lo(n_tiles), h_hi(n_tiles), h_tile_blk((size_t)n_tiles + 1, 0);
    MLLP_HIP_TRY(hipMemcpyAsync(h_lo.data(), lo.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipMemcpyAsync(h_hi.data(), hi.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    int64_t n_tb = 0;
    int max_nbt = 0;
    for (int t = 0; t < n_tiles; ++t) {
        h_tile_blk[t] = (int)n_tb;
        const int nbt = h_hi[t] < 0 ? 0 : h_hi[t] - h_lo[t] + 1;
        if (h_hi[t] < 0) h_lo[t] = 0;
        n_tb += nbt;
        max_nbt = std::max(max_nbt, nbt);
        if (n_tb * R >= INT32_MAX - 1) return fail(MLLP_ERANGE, "tiled copy: (tile, block) x rows exceeds int32 indexing");
    }
    h_tile_blk[n_tiles] = (int)n_tb;
    if (max_nbt > tiled_max_blocks_per_tile())
        return fail(MLLP_ERANGE, "tiled copy: a row tile touches more column blocks than the kernel's table holds");
    std::vector<int> h_blk((size_t)n_tb);
    for (int t = 0; t < n_tiles; ++t)
        for (int b = h_tile_blk[t]; b < h_tile_blk[t + 1]; ++b) h_blk[b] = h_lo[t] + (b - h_tile_blk[t]);
    Tiled tl;
    tl.owned = true;
    tl.n_tiles = n_tiles;
    tl.n_tb = (int)n_tb;
    int *d_tile_blk = nullptr, *d_blk = nullptr, *d_ptr2 = nullptr, *d_perm = nullptr, *d_ent = nullptr;
    const size_t n_slots = (size_t)n_tb * R;
    auto cleanup = [&]() {
        (void)hipFree(d_tile_blk); (void)hipFree(d_blk); (void)hipFree(d_ptr2); (void)hipFree(d_perm); (void)hipFree(d_ent);
    };
    if (hipMalloc((void**)&d_tile_blk, ((size_t)n_tiles + 1) * 4) != hipSuccess || hipMalloc((void**)&d_blk, std::max<size_t>(n_tb, 1) * 4) != hipSuccess ||
        hipMalloc((void**)&d_ptr2, (n_slots + 1) * 4) != hipSuccess || hipMalloc((void**)&d_perm, std::max<size_t>(n_slots, 1) * 4) != hipSuccess ||
        hipMalloc((void**)&d_ent, ((size_t)nnz + 1) * 8) != hipSuccess || cnt.alloc(n_slots) || start.alloc(n_slots) || off.alloc(n_slots) ||
        seg.alloc((size_t)n_tb) || seg_start.alloc((size_t)n_tb) || longest.alloc(1)) {
        cleanup();
        return fail(MLLP_ENOMEM, "tiled copy: hipMalloc failed");
    }
    auto bail = [&](hipError_t e, const char* what) { cleanup(); return hip_fail(e, what); };
    hipError_t e;
    if ((e = hipMemcpyAsync(d_tile_blk, h_tile_blk.data(), ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemcpyAsync(d_blk, h_blk.data(), (size_t)n_tb * 4, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemcpyAsync(lo.p, h_lo.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice, s)) != hipSuccess ||
        (e = hipMemsetAsync(cnt.p, 0, std::max<size_t>(n_slots, 1) * 4, s)) != hipSuccess ||
        (e = hipMemsetAsync(start.p, 0, std::max<size_t>(n_slots, 1) * 4, s)) != hipSuccess ||
        (e = hipMemsetAsync(d_ent + (size_t)nnz * 2, 0, 8, s)) != hipSuccess || (e = hipMemsetAsync(longest.p, 0, 4, s)) != hipSuccess)
        return bail(e, "tiled copy: copy / memset");
    hipLaunchKernelGGL(tb_count, dim3(n_tiles), dim3(T), 0, s, o.ptr, o.idx, o.n_dst, R, CB, lo.p, d_tile_blk, cnt.p, start.p);
    hipLaunchKernelGGL(tb_sort, dim3((unsigned)n_tb), dim3(R), (size_t)2 * R * 4, s, cnt.p, R, variant == 4 ? 1 : 0, d_perm,
                       off.p, seg.p, longest.p);
    std::vector<int> h_seg((size_t)n_tb), h_seg_start((size_t)n_tb);
    int h_longest = 0;
    if ((e = hipMemcpyAsync(h_seg.data(), seg.p, (size_t)n_tb * 4, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipMemcpyAsync(&h_longest, longest.p, 4, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess)
        return bail(e, "tiled copy: segment lengths");
    int64_t run = 0;
    int max_run = 0;
    for (int64_t b = 0; b < n_tb; ++b) {
        h_seg_start[b] = (int)run;
        run += h_seg[b];
        max_run = std::max(max_run, h_seg[b]);
    }
    if (run != nnz) {
        cleanup();
        return fail(MLLP_EINVAL, "tiled copy: the segments do not add up to nnz (column ids not ascending inside a row?)");
    }
    if ((e = hipMemcpyAsync(seg_start.p, h_seg_start.data(), (size_t)n_tb * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
        return bail(e, "tiled copy: segment starts");
    const long long n_p = (long long)n_slots;
    hipLaunchKernelGGL(tb_ptr2, dim3((unsigned)((n_p + 256) / 256)), dim3(256), 0, s, off.p, seg_start.p, R, n_p, (int)nnz, d_ptr2);
    hipLaunchKernelGGL(tb_fill, dim3((unsigned)n_tb), dim3(R), (size_t)R * 4, s, o.idx, o.val, d_blk, cnt.p, start.p, d_perm,
                       d_ptr2, R, CB, item_bytes, ((variant == 0 || variant == 1 || variant == 4) && h_longest <= 512) ? 1 : 0, variant == 4 ? 1 : 0,
                       reinterpret_cast<int2*>(d_ent));
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) return bail(e, "tiled copy: kernels");
    tl.tile_blk = d_tile_blk; tl.blk_id = d_blk; tl.ptr2 = d_ptr2; tl.perm = d_perm; tl.ent = d_ent;
    tl.max_nbt = max_nbt;
    tl.max_run = max_run;
    out = tl;
    return MLLP_OK;
}

}  // namespace mllp
