// transpose.hip -- CSR(A) -> CSR(A^T) on the device (mllp_csr_transpose_device): the second orientation of a batch that
// was generated or loaded on the GPU.  Replaces the stable sort by column that mllp_amd/graph.py::from_device_csr did with
// torch (rocPRIM radix sort of 512 M keys + two gathers).  The reference builds both directions of its edge list on the
// host every step (build_graph_from_weights_sets, linear_program_methods.py:89-103); here it happens once per batch.
//   tr_count     entries per column (integer atomics: the counts do not depend on the order)
//   tr_scan_*    exclusive prefix sum of the counts -> row pointers of A^T (three small passes)
//   tr_scatter   every entry to a free slot of its column's segment (cursor by integer atomic: ANY order)
//   tr_sort_*    every column's segment ordered by row id -- the row ids of a column are distinct, so the result is the
//                one stable transposition whatever order the scatter produced: deterministic, bit-identical to the sort
// Columns of up to TR_WAVE_MAX entries are ordered by one wavefront in LDS (rank = number of smaller row ids), longer
// ones by a whole workgroup through global memory (dense columns of Netlib-like matrices: rare).
#include <algorithm>

#include "internal.h"

namespace mllp {
namespace {

constexpr int TR_T = 256;
constexpr int TR_WAVE_MAX = 1024;       // entries of a column that one wavefront orders in LDS (8 KB per wavefront)
constexpr int TR_SCAN = 1024;           // elements per scan block

// input check: the row pointers ascend from 0 to nnz, and inside a row the column ids are strictly ascending
// and below n_cols -- a duplicate (row, column) would give two entries of a column the same rank in tr_sort_*, an id out
// of range would send tr_count's atomic outside its array.  bad[0] != 0 afterwards: MLLP_EINVAL, nothing else is launched.
__global__ void tr_validate(const int* __restrict__ ptr, const int* __restrict__ idx, long long n_rows, long long n_cols,
                            long long nnz, int* __restrict__ bad) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (long long)gridDim.x * blockDim.x) {
        const long long b = ptr[r], e = ptr[r + 1];
        bool ok = b <= e && b >= 0 && e <= nnz && (r > 0 || b == 0) && (r + 1 < n_rows || e == nnz);
        if (ok) {
            long long last = -1;
            for (long long k = b; k < e; ++k) {
                const long long c = idx[k];
                ok = ok && c > last && c < n_cols;
                last = c;
            }
        }
        if (!ok) atomicOr(bad, 1);
    }
}

__global__ void tr_count(const int* __restrict__ idx, long long nnz, int* __restrict__ cnt) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (long long)gridDim.x * blockDim.x)
        atomicAdd(&cnt[idx[e]], 1);
}

// block sums of cnt[0 .. n) -> part[block]; then part scanned by one block; then ptr[i] = exclusive prefix
__global__ __launch_bounds__(TR_T) void tr_scan_sums(const int* __restrict__ cnt, long long n, int* __restrict__ part) {
    __shared__ int sh[TR_T / 64];
    const long long base = (long long)blockIdx.x * TR_SCAN;
    int v = 0;
    for (int k = threadIdx.x; k < TR_SCAN; k += TR_T)
        if (base + k < n) v += cnt[base + k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < TR_T / 64; ++w) t += sh[w];
        part[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void tr_scan_parts(int* __restrict__ part, int n_part) {     // one workgroup, in place -> exclusive
    __shared__ int sh[1024];
    int run = 0;
    for (int base = 0; base < n_part; base += 1024) {
        const int i = base + threadIdx.x;
        const int mine = i < n_part ? part[i] : 0;
        sh[threadIdx.x] = mine;
        __syncthreads();
        int v = mine;
        for (int d = 1; d < 1024; d <<= 1) {
            const int add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            v += add;
            sh[threadIdx.x] = v;
            __syncthreads();
        }
        if (i < n_part) part[i] = run + v - mine;
        run += sh[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(TR_T) void tr_scan_final(const int* __restrict__ cnt, long long n, const int* __restrict__ part,
                                                      int* __restrict__ ptr, int* __restrict__ cursor) {
    __shared__ int sh[TR_SCAN];
    const long long base = (long long)blockIdx.x * TR_SCAN;
    for (int k = threadIdx.x; k < TR_SCAN; k += TR_T) sh[k] = base + k < n ? cnt[base + k] : 0;
    __syncthreads();
    if (threadIdx.x < 64) {       // one wavefront: 16 elements per lane, sequential inside the lane
        const int l = threadIdx.x;
        int s = 0;
        for (int k = 0; k < TR_SCAN / 64; ++k) s += sh[l * (TR_SCAN / 64) + k];
        int inc = s;
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(inc, d, 64);
            if (l >= d) inc += up;
        }
        int run = part[blockIdx.x] + inc - s;
        for (int k = 0; k < TR_SCAN / 64; ++k) {
            const int c = sh[l * (TR_SCAN / 64) + k];
            sh[l * (TR_SCAN / 64) + k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TR_SCAN; k += TR_T)
        if (base + k < n) {
            ptr[base + k] = sh[k];
            cursor[base + k] = sh[k];
        }
}

// one wavefront per row: its entries to the next free slots of their columns
__global__ __launch_bounds__(TR_T) void tr_scatter(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                   const float* __restrict__ val, long long n_rows, int* __restrict__ cursor,
                                                   int* __restrict__ t_idx, float* __restrict__ t_val) {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * (TR_T / 64) + (threadIdx.x >> 6); r < n_rows; r += (long long)gridDim.x * (TR_T / 64)) {
        for (int e = ptr[r] + lane; e < ptr[r + 1]; e += 64) {
            const int pos = atomicAdd(&cursor[idx[e]], 1);
            t_idx[pos] = (int)r;
            t_val[pos] = val[e];
        }
    }
}

// one wavefront per column of at most TR_WAVE_MAX entries: rank sort by row id in LDS
__global__ __launch_bounds__(TR_T) void tr_sort_short(const int* __restrict__ t_ptr, long long n_cols, int* __restrict__ t_idx,
                                                      float* __restrict__ t_val) {
    __shared__ int s_row[TR_T / 64][TR_WAVE_MAX];
    __shared__ float s_val[TR_T / 64][TR_WAVE_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long c = (long long)blockIdx.x * (TR_T / 64) + w; c < n_cols; c += (long long)gridDim.x * (TR_T / 64)) {
        const int b = t_ptr[c], len = t_ptr[c + 1] - b;
        if (len < 2 || len > TR_WAVE_MAX) continue;
        bool sorted = true;
        for (int k = lane; k < len; k += 64) {
            s_row[w][k] = t_idx[b + k];
            s_val[w][k] = t_val[b + k];
        }
        // (wavefront-private LDS: program order of the lanes' accesses suffices once the stores have been issued)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < len; k += 64) sorted = sorted && (k == 0 || s_row[w][k - 1] < s_row[w][k]);
        if (__all(sorted)) continue;          // (already in order: most columns when few wavefronts raced for them)
        for (int k = lane; k < len; k += 64) {
            const int mine = s_row[w][k];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += s_row[w][j] < mine ? 1 : 0;
            t_idx[b + rank] = mine;
            t_val[b + rank] = s_val[w][k];
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// one workgroup per long column (listed in `cols`): rank sort through a scratch copy in global memory
__global__ __launch_bounds__(1024) void tr_sort_long(const int* __restrict__ t_ptr, const int* __restrict__ cols,
                                                     int* __restrict__ t_idx, float* __restrict__ t_val,
                                                     int* __restrict__ scratch_idx, float* __restrict__ scratch_val) {
    const int c = cols[blockIdx.x];
    const int b = t_ptr[c], len = t_ptr[c + 1] - b;
    for (int k = threadIdx.x; k < len; k += 1024) {
        scratch_idx[b + k] = t_idx[b + k];
        scratch_val[b + k] = t_val[b + k];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < len; k += 1024) {
        const int mine = scratch_idx[b + k];
        int rank = 0;
        for (int j = 0; j < len; ++j) rank += scratch_idx[b + j] < mine ? 1 : 0;
        t_idx[b + rank] = mine;
        t_val[b + rank] = scratch_val[b + k];
    }
}
__global__ void tr_list_long(const int* __restrict__ t_ptr, long long n_cols, int* __restrict__ cols, int* __restrict__ n_long,
                             int capacity) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (long long)gridDim.x * blockDim.x)
        if (t_ptr[c + 1] - t_ptr[c] > TR_WAVE_MAX) {
            const int k = atomicAdd(n_long, 1);
            if (k < capacity) cols[k] = (int)c;
        }
}

}  // namespace
}  // namespace mllp

using namespace mllp;

extern "C" int mllp_csr_transpose_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* d_ptr,
                                         const int32_t* d_idx, const float* d_val, int32_t* d_t_ptr, int32_t* d_t_idx,
                                         float* d_t_val, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || !d_ptr || !d_t_ptr || (nnz > 0 && (!d_idx || !d_val || !d_t_idx || !d_t_val)))
        return fail(MLLP_EINVAL, "mllp_csr_transpose_device: bad arguments");
    if (nnz >= INT32_MAX || n_rows >= INT32_MAX || n_cols >= INT32_MAX)
        return fail(MLLP_ERANGE, "mllp_csr_transpose_device: sizes exceed int32 indexing");
    if (n_rows == 0 && nnz != 0) return fail(MLLP_EINVAL, "mllp_csr_transpose_device: nonzeros without rows");
    hipStream_t s = (hipStream_t)stream;
    const long long n1 = n_cols + 1;
    const int n_part = (int)((n1 + TR_SCAN - 1) / TR_SCAN);
    int *cnt = nullptr, *part = nullptr, *cursor = nullptr, *cols = nullptr, *n_long = nullptr;
    auto cleanup = [&]() { (void)hipFree(cnt); (void)hipFree(part); (void)hipFree(cursor); (void)hipFree(cols); (void)hipFree(n_long); };
    const int long_cap = (int)std::min<int64_t>(std::max<int64_t>(nnz / TR_WAVE_MAX, 1), 1 << 22);
    if (hipMalloc((void**)&cnt, (size_t)n1 * 4) != hipSuccess || hipMalloc((void**)&part, (size_t)std::max(n_part, 1) * 4) != hipSuccess ||
        hipMalloc((void**)&cursor, (size_t)n1 * 4) != hipSuccess || hipMalloc((void**)&cols, (size_t)long_cap * 4) != hipSuccess ||
        hipMalloc((void**)&n_long, 4) != hipSuccess) {
        cleanup();
        return fail(MLLP_ENOMEM, "mllp_csr_transpose_device: hipMalloc failed");
    }
    hipError_t e;
    auto bail = [&](hipError_t err, const char* what) { cleanup(); return hip_fail(err, what); };
    if ((e = hipMemsetAsync(cnt, 0, (size_t)n1 * 4, s)) != hipSuccess || (e = hipMemsetAsync(n_long, 0, 4, s)) != hipSuccess)
        return bail(e, "transpose: memset");
    {   // (n_long doubles as the flag of the input check: read back before anything indexes with the column ids)
        hipLaunchKernelGGL(tr_validate, dim3(4096), dim3(TR_T), 0, s, d_ptr, d_idx, (long long)n_rows, (long long)n_cols,
                           (long long)nnz, n_long);
        int h_bad = 0;
        if ((e = hipMemcpyAsync(&h_bad, n_long, 4, hipMemcpyDeviceToHost, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess)
            return bail(e, "transpose: input check");
        if (h_bad) {
            cleanup();
            return fail(MLLP_EINVAL, "mllp_csr_transpose_device: the row pointers must ascend from 0 to nnz and the column ids of a "
                                     "row must be strictly ascending and smaller than n_cols (no duplicate entries)");
        }
    }
    if (nnz > 0) hipLaunchKernelGGL(tr_count, dim3(4096), dim3(TR_T), 0, s, d_idx, (long long)nnz, cnt);
    hipLaunchKernelGGL(tr_scan_sums, dim3(n_part), dim3(TR_T), 0, s, cnt, n1, part);
    hipLaunchKernelGGL(tr_scan_parts, dim3(1), dim3(1024), 0, s, part, n_part);
    hipLaunchKernelGGL(tr_scan_final, dim3(n_part), dim3(TR_T), 0, s, cnt, n1, part, d_t_ptr, cursor);
    if (nnz > 0) {
        hipLaunchKernelGGL(tr_scatter, dim3(8192), dim3(TR_T), 0, s, d_ptr, d_idx, d_val, (long long)n_rows, cursor, d_t_idx, d_t_val);
        hipLaunchKernelGGL(tr_sort_short, dim3(8192), dim3(TR_T), 0, s, d_t_ptr, (long long)n_cols, d_t_idx, d_t_val);
        hipLaunchKernelGGL(tr_list_long, dim3(1024), dim3(TR_T), 0, s, d_t_ptr, (long long)n_cols, cols, n_long, long_cap);
        int h_long = 0;
        if ((e = hipMemcpyAsync(&h_long, n_long, 4, hipMemcpyDeviceToHost, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess)
            return bail(e, "transpose: long columns");
        if (h_long > long_cap) {
            cleanup();
            return fail(MLLP_ERANGE, "mllp_csr_transpose_device: more long columns than the list holds");
        }
        if (h_long > 0) {
            int* sc_idx = nullptr;
            float* sc_val = nullptr;
            if (hipMalloc((void**)&sc_idx, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&sc_val, (size_t)nnz * 4) != hipSuccess) {
                (void)hipFree(sc_idx);
                cleanup();
                return fail(MLLP_ENOMEM, "mllp_csr_transpose_device: hipMalloc failed");
            }
            hipLaunchKernelGGL(tr_sort_long, dim3(h_long), dim3(1024), 0, s, d_t_ptr, cols, d_t_idx, d_t_val, sc_idx, sc_val);
            e = hipStreamSynchronize(s);
            (void)hipFree(sc_idx);
            (void)hipFree(sc_val);
            if (e != hipSuccess) return bail(e, "transpose: long columns");
        }
    }
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) return bail(e, "transpose: kernels");
    cleanup();
    return MLLP_OK;
}
