// stream_build.hip -- device builder of the streamed SpMM copy (layout: stream_layout.h; reference builder with the
// same rules and the same bytes: host_stream.cpp).  Replaces, once per batch and orientation, what the reference
// redoes every step (build_graph_from_weights_sets, linear_program_methods.py:89-103).  Counting passes + one placement
// pass, no sort of the nonzeros:
//   sb_tile_range     per tile: first / last column block its rows touch
//   sb_tile_bitmap    per tile: bitmap of the touched blocks (relative to the first), their number
//   sb_tile_blocks    per tile: the ascending block list (blk_id) + prefix popcounts of the bitmap words
//   sb_count_rows     per (tile, block, row): number of entries and CSR position of the first one
//   sb_sort_rows      per (tile, block): rows by entry count, descending, ties by row (stable rank, brute force in LDS);
//                     steps of both passes of every walker wavefront
//   sb_wave_totals / sb_step_starts   position of every (tile, block, wavefront) record in the entry stream
//   sb_fill_padding   the whole stream = padding entries
//   sb_fill           records + the real entries: one thread per TEAM (4 rows that are read in one LDS cycle), jointly
//                     ordered over (column mod 4) exactly as host_stream.cpp::fill_slot does
// Three small arrays (per-tile ranges, block counts, per-(tile, wavefront) group counts) are scanned on the host.
#include <algorithm>
#include <climits>
#include <vector>

#include "internal.h"
#include "stream_layout.h"

namespace mllp {

void stream_copy_free(StreamCopy& sc);

namespace {


template <class G>
__global__ __launch_bounds__(G::R) void sb_tile_range(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                      const int* __restrict__ tile_row, int* __restrict__ lo,
                                                      int* __restrict__ hi) {
    __shared__ int s_lo, s_hi;
    if (threadIdx.x == 0) { s_lo = INT_MAX; s_hi = -1; }
    __syncthreads();
    const int r = tile_row[blockIdx.x] + threadIdx.x;
    if (r < tile_row[blockIdx.x + 1]) {
        const int b = ptr[r], e = ptr[r + 1];
        if (e > b) {
            atomicMin(&s_lo, idx[b] / G::CB);
            atomicMax(&s_hi, idx[e - 1] / G::CB);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { lo[blockIdx.x] = s_lo; hi[blockIdx.x] = s_hi; }
}

// bm: bitmap words of every tile, tile t at bm_off[t] (ceil((hi - lo + 1) / 32) words)
template <class G>
__global__ __launch_bounds__(G::R) void sb_tile_bitmap(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                       const int* __restrict__ tile_row, const int* __restrict__ lo, const int* __restrict__ hi,
                                                       const int* __restrict__ bm_off, unsigned* __restrict__ bm,
                                                       int* __restrict__ nb) {
    extern __shared__ unsigned s_bm[];
    __shared__ int s_cnt;
    const int t = blockIdx.x, l = lo[t], h = hi[t];
    if (h < 0) {
        if (threadIdx.x == 0) nb[t] = 0;
        return;
    }
    constexpr int SB_T = G::R, S_CB = G::CB;
    const int words = (h - l + 32) >> 5;
    for (int i = threadIdx.x; i < words; i += SB_T) s_bm[i] = 0u;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int r = tile_row[t] + threadIdx.x;
    if (r < tile_row[t + 1]) {
        int last = -1;
        for (int e = ptr[r]; e < ptr[r + 1]; ++e) {
            const int b = idx[e] / S_CB;
            if (b != last) {
                last = b;
                atomicOr(&s_bm[(b - l) >> 5], 1u << ((b - l) & 31));
            }
        }
    }
    __syncthreads();
    int c = 0;
    for (int i = threadIdx.x; i < words; i += SB_T) {
        c += __popc(s_bm[i]);
        bm[bm_off[t] + i] = s_bm[i];
    }
    if (c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) nb[t] = s_cnt;
}

// one wavefront per tile: ascending block list and the exclusive prefix popcount of every bitmap word
__global__ __launch_bounds__(64) void sb_tile_blocks(const int* __restrict__ lo, const int* __restrict__ hi,
                                                     const int* __restrict__ bm_off, const unsigned* __restrict__ bm,
                                                     const int* __restrict__ tile_blk, int* __restrict__ blk_id,
                                                     int* __restrict__ bm_pref) {
    const int t = blockIdx.x, l = lo[t], h = hi[t], lane = threadIdx.x;
    if (h < 0) return;
    const int words = (h - l + 32) >> 5;
    int running = 0;
    for (int base = 0; base < words; base += 64) {
        const int i = base + lane;
        unsigned w = i < words ? bm[bm_off[t] + i] : 0u;
        const int c = __popc(w);
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        int pos = running + inc - c;
        if (i < words) bm_pref[bm_off[t] + i] = pos;
        while (w) {
            const int bit = __ffs(w) - 1;
            w &= w - 1;
            blk_id[tile_blk[t] + pos++] = l + i * 32 + bit;
        }
        running += __shfl(inc, 63, 64);
    }
}

template <class G>
__global__ __launch_bounds__(G::R) void sb_count_rows(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                      const int* __restrict__ tile_row, const int* __restrict__ lo, const int* __restrict__ bm_off,
                                                      const unsigned* __restrict__ bm, const int* __restrict__ bm_pref,
                                                      const int* __restrict__ tile_blk, int* __restrict__ cnt,
                                                      int* __restrict__ start) {
    constexpr int S_R = G::R, S_CB = G::CB;
    const int t = blockIdx.x, r = tile_row[t] + threadIdx.x;
    if (r >= tile_row[t + 1]) return;
    const int l = lo[t], off = bm_off[t], tb0 = tile_blk[t];
    int last = -1, c = 0;
    size_t slot = 0;
    for (int e = ptr[r]; e < ptr[r + 1]; ++e) {
        const int b = idx[e] / S_CB;
        if (b != last) {
            if (c) cnt[slot] = c;
            last = b;
            c = 0;
            const int rel = b - l, w = rel >> 5;
            const int bi = bm_pref[off + w] + __popc(bm[off + w] & ((1u << (rel & 31)) - 1u));
            slot = (size_t)(tb0 + bi) * S_R + threadIdx.x;
            start[slot] = e;
        }
        ++c;
    }
    if (c) cnt[slot] = c;
}

// order[tb][k] = row with the k-th most entries (ties by row); npass[tb][w][2] = {n0 | n1 << 16, n2 | n3 << 16}
template <class G>
__global__ __launch_bounds__(G::R) void sb_sort_rows(const int* __restrict__ cnt, int* __restrict__ order,
                                                     int* __restrict__ npass) {
    constexpr int S_R = G::R, S_NW = G::NW, S_P = G::P, S_BR = G::BR;
    __shared__ int c[S_R], ord[S_R];
    const size_t tb = blockIdx.x;
    const int r = threadIdx.x;
    const int mine = cnt[tb * S_R + r];
    c[r] = mine;
    __syncthreads();
    int rank = 0;
    for (int k = 0; k < S_R; ++k) {
        const int o = c[k];
        rank += (o > mine || (o == mine && k < r)) ? 1 : 0;
    }
    ord[rank] = r;
    order[tb * S_R + rank] = r;
    __syncthreads();
    if (r < S_NW) {
        int n[2] = {0, 0};
        for (int j = 0; j < S_P; ++j) n[j] = c[ord[S_BR * G::bundle(r, j)]];
        npass[(tb * S_NW + r) * 2] = n[0] | n[1] << 16;
        npass[(tb * S_NW + r) * 2 + 1] = 0;
    }
}

template <class G>
__global__ void sb_wave_totals(const int* __restrict__ tile_blk, const int* __restrict__ npass, int n_tiles,
                               int* __restrict__ groups) {
    constexpr int S_NW = G::NW, S_GS = G::GS;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * S_NW) return;
    const int t = i / S_NW, w = i % S_NW;
    long long steps = 0;
    for (int tb = tile_blk[t]; tb < tile_blk[t + 1]; ++tb) {
        const unsigned v = (unsigned)npass[((size_t)tb * S_NW + w) * 2], u = (unsigned)npass[((size_t)tb * S_NW + w) * 2 + 1];
        steps += (v & 0xffffu) + (v >> 16) + (u & 0xffffu) + (u >> 16);
    }
    groups[i] = (int)((steps + S_GS - 1) / S_GS);
}

template <class G>
__global__ void sb_step_starts(const int* __restrict__ tile_blk, const int* __restrict__ npass, int n_tiles,
                               const int* __restrict__ base_group, int* __restrict__ step_start) {
    constexpr int S_NW = G::NW, S_GS = G::GS;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * S_NW) return;
    const int t = i / S_NW, w = i % S_NW;
    int run = base_group[i] * S_GS;
    for (int tb = tile_blk[t]; tb < tile_blk[t + 1]; ++tb) {
        const unsigned v = (unsigned)npass[((size_t)tb * S_NW + w) * 2], u = (unsigned)npass[((size_t)tb * S_NW + w) * 2 + 1];
        step_start[(size_t)tb * S_NW + w] = run;
        run += (int)((v & 0xffffu) + (v >> 16) + (u & 0xffffu) + (u >> 16));
    }
}

// n = number of (group, lane) items of S_ENT ints
template <class G>
__global__ void sb_fill_padding(int* __restrict__ ent, long long n) {
    constexpr int S_ENT = G::ENT, S_PAD_WORD = G::PAD_WORD;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * S_ENT; i += (long long)gridDim.x * blockDim.x)
        ent[i] = (i % S_ENT) == 0 ? S_PAD_WORD : 0;
}

// thread = (wavefront w, pass j, row slot, team): the joint ordering of host_stream.cpp::fill_slot for its four rows;
// the first S_NW x 16 threads also write the row records, the first S_NW the headers
template <class G>
struct SbFill {
    // one thread per (wavefront, pass, row slot, team); at least NW x 16 threads for the row records
    static constexpr int TEAM_T = G::NW * G::P * G::RQ * 4;
    static constexpr int T = TEAM_T >= G::NW * 16 ? TEAM_T : G::NW * 16;
    static_assert(T <= 1024, "sb_fill threads");
};
template <class G>
__global__ __launch_bounds__(SbFill<G>::T) void sb_fill(const int* __restrict__ idx, const float* __restrict__ val,
                                                     const int* __restrict__ blk_id, const int* __restrict__ cnt,
                                                     const int* __restrict__ start, const int* __restrict__ order,
                                                     const int* __restrict__ npass, const int* __restrict__ step_start,
                                                     int4* __restrict__ rows_out, int4* __restrict__ hdr_out,
                                                     int* __restrict__ ent) {
    constexpr int S_R = G::R, S_NW = G::NW, S_P = G::P, S_RQ = G::RQ, S_BR = G::BR, S_CB = G::CB, S_ROW_BYTES = G::ITEM;
    const size_t tb = blockIdx.x;
    const int tid = threadIdx.x;
    const int* ord = order + tb * S_R;
    if (tid < S_NW * 16) {
        const int w = tid >> 4, q = tid & 15;
        int r4[4] = {0, 0, 0, 0};
        for (int j = 0; j < S_P; ++j)
            for (int r = 0; r < S_RQ; ++r) r4[2 * j + (r >> 1)] |= ord[S_BR * G::bundle(w, j) + 16 * r + q] << (16 * (r & 1));
        rows_out[(tb * S_NW + w) * 16 + q] = make_int4(r4[0], r4[1], r4[2], r4[3]);
    }
    if (tid < S_NW) hdr_out[tb * S_NW + tid] = make_int4(step_start[tb * S_NW + tid], npass[(tb * S_NW + tid) * 2], blk_id[tb], 0);
    if (tid >= SbFill<G>::TEAM_T) return;
    const int w = tid / (S_P * S_RQ * 4), pass = (tid / (S_RQ * 4)) % S_P, slot = (tid >> 2) % S_RQ, tm = tid & 3;
    const int bundle = G::bundle(w, pass);
    const unsigned np0 = (unsigned)npass[(tb * S_NW + w) * 2];
    const long long S = (long long)step_start[tb * S_NW + w] + (pass ? (int)(np0 & 0xffffu) : 0);
    const int blk = blk_id[tb];
    const int c0 = blk * S_CB;
    int beg[4], rem[4], nxt[4][4], cl[4][4], quad[4];
    int maxlen = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        quad[i] = S_TEAMS[tm][i];
        const int r = ord[S_BR * bundle + 16 * slot + quad[i]];
        beg[i] = start[tb * S_R + r];
        rem[i] = cnt[tb * S_R + r];
#pragma unroll
        for (int k = 0; k < 4; ++k) { nxt[i][k] = 0; cl[i][k] = 0; }
        for (int e = 0; e < rem[i]; ++e) {
            const int k = (idx[beg[i] + e] - c0) & 3;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) cl[i][kk] += (kk == k) ? 1 : 0;
        }
        maxlen = max(maxlen, rem[i]);
    }
    for (int p = 0; p < maxlen; ++p) {
        unsigned used = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i != ((j + p) & 3)) continue;        // (static register indexing: i is a compile-time constant)
                if (rem[i] == 0) continue;
                int pick = -1, pick_any = -1, best = 0, best_any = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ck = cl[i][k];
                    if (ck == 0) continue;
                    if (pick_any < 0 || ck > best_any) { pick_any = k; best_any = ck; }
                    if (!(used >> k & 1u) && (pick < 0 || ck > best)) { pick = k; best = ck; }
                }
                if (pick < 0) pick = pick_any;
                int e = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) e = (k == pick) ? nxt[i][k] : e;
                while (((idx[beg[i] + e] - c0) & 3) != pick) ++e;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k == pick) { nxt[i][k] = e + 1; cl[i][k] -= 1; }
                }
                rem[i] -= 1;
                used |= 1u << pick;
                const long long step = S + p;
                // (the two halves of the offset word may belong to different passes / blocks, i.e. to different
                // threads: two-byte stores, no read-modify-write)
                int* dst = ent + G::ent_index(step, quad[i], slot);
                reinterpret_cast<unsigned short*>(dst)[step & 1] = (unsigned short)((idx[beg[i] + e] - c0) * S_ROW_BYTES);
                dst[1 + (step & 1)] = __float_as_int(val[beg[i] + e]);
            }
        }
    }
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? 0 : 1; }
};

}  // namespace

template <class G>
static int build_stream_device_t(const Orient& o, int64_t nnz, const std::vector<int>& tile_row, StreamCopy& sc, hipStream_t s) {
    constexpr int SB_T = G::R, S_R = G::R, S_NW = G::NW, S_GS = G::GS, S_K0 = G::K0, S_ENT = G::ENT;
    (void)nnz;
    const int n_tiles = (int)tile_row.size() - 1;
    MLLP_HIP_TRY(hipMalloc((void**)&sc.tile_row, ((size_t)n_tiles + 1) * 4));
    MLLP_HIP_TRY(hipMemcpyAsync(sc.tile_row, tile_row.data(), ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s));
    DevBuf<int> lo, hi, bm_off, nb, bm_pref, cnt, start, order, npass, groups, base, sstart;
    DevBuf<unsigned> bm;
    if (lo.alloc(n_tiles) || hi.alloc(n_tiles) || bm_off.alloc(n_tiles) || nb.alloc(n_tiles))
        return fail(MLLP_ENOMEM, "streamed copy: hipMalloc failed");
    hipLaunchKernelGGL(sb_tile_range<G>, dim3(n_tiles), dim3(SB_T), 0, s, o.ptr, o.idx, sc.tile_row, lo.p, hi.p);
    std::vector<int> h_lo(n_tiles), h_hi(n_tiles), h_off(n_tiles), h_nb(n_tiles);
    MLLP_HIP_TRY(hipMemcpyAsync(h_lo.data(), lo.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipMemcpyAsync(h_hi.data(), hi.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    int64_t words_total = 0;
    int words_max = 1;
    for (int t = 0; t < n_tiles; ++t) {
        const int words = h_hi[t] < 0 ? 0 : (h_hi[t] - h_lo[t] + 32) >> 5;
        h_off[t] = (int)words_total;
        words_total += words;
        words_max = std::max(words_max, words);
        if (words_total >= INT32_MAX) return fail(MLLP_ERANGE, "streamed copy: block bitmaps exceed int32 indexing");
    }
    if (words_max > 32768)
        return fail(MLLP_ERANGE, "streamed copy: a row tile spans more than 2^20 column blocks");
    if (bm.alloc((size_t)words_total) || bm_pref.alloc((size_t)words_total))
        return fail(MLLP_ENOMEM, "streamed copy: hipMalloc failed");
    MLLP_HIP_TRY(hipMemcpyAsync(bm_off.p, h_off.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice, s));
    if (words_max * 4 > 48 * 1024)
        MLLP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sb_tile_bitmap<G>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, words_max * 4));
    hipLaunchKernelGGL(sb_tile_bitmap<G>, dim3(n_tiles), dim3(SB_T), (size_t)words_max * 4, s, o.ptr, o.idx, sc.tile_row,
                       lo.p, hi.p, bm_off.p, bm.p, nb.p);
    MLLP_HIP_TRY(hipMemcpyAsync(h_nb.data(), nb.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    std::vector<int> h_tile_blk((size_t)n_tiles + 1, 0);
    int64_t n_tb = 0;
    for (int t = 0; t < n_tiles; ++t) {
        h_tile_blk[t] = (int)n_tb;
        n_tb += h_nb[t];
        if (n_tb >= (1 << 24)) return fail(MLLP_ERANGE, "streamed copy: too many (tile, block) pairs");
    }
    h_tile_blk[n_tiles] = (int)n_tb;
    sc.n_tiles = n_tiles;
    sc.n_tb = (int)n_tb;
    MLLP_HIP_TRY(hipMalloc((void**)&sc.tile_blk, ((size_t)n_tiles + 1) * 4));
    MLLP_HIP_TRY(hipMalloc((void**)&sc.blk_id, std::max<size_t>(n_tb, 1) * 4));
    MLLP_HIP_TRY(hipMalloc((void**)&sc.rows, std::max<size_t>(n_tb, 1) * S_NW * 256));
    MLLP_HIP_TRY(hipMalloc((void**)&sc.hdr, std::max<size_t>(n_tb, 1) * S_NW * 16));
    MLLP_HIP_TRY(hipMemcpyAsync(sc.tile_blk, h_tile_blk.data(), ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s));
    const size_t n_slots = (size_t)n_tb * S_R;
    if (cnt.alloc(n_slots) || start.alloc(n_slots) || order.alloc(n_slots) || npass.alloc((size_t)n_tb * S_NW * 2) ||
        sstart.alloc((size_t)n_tb * S_NW) || groups.alloc((size_t)n_tiles * S_NW) || base.alloc((size_t)n_tiles * S_NW))
        return fail(MLLP_ENOMEM, "streamed copy: hipMalloc failed");
    MLLP_HIP_TRY(hipMemsetAsync(cnt.p, 0, std::max<size_t>(n_slots, 1) * 4, s));
    MLLP_HIP_TRY(hipMemsetAsync(start.p, 0, std::max<size_t>(n_slots, 1) * 4, s));
    hipLaunchKernelGGL(sb_tile_blocks, dim3(n_tiles), dim3(64), 0, s, lo.p, hi.p, bm_off.p, bm.p, sc.tile_blk, sc.blk_id,
                       bm_pref.p);
    hipLaunchKernelGGL(sb_count_rows<G>, dim3(n_tiles), dim3(SB_T), 0, s, o.ptr, o.idx, sc.tile_row, lo.p, bm_off.p, bm.p,
                       bm_pref.p, sc.tile_blk, cnt.p, start.p);
    if (n_tb > 0) hipLaunchKernelGGL(sb_sort_rows<G>, dim3((unsigned)n_tb), dim3(SB_T), 0, s, cnt.p, order.p, npass.p);
    const int nw = n_tiles * S_NW;
    hipLaunchKernelGGL(sb_wave_totals<G>, dim3((nw + 255) / 256), dim3(256), 0, s, sc.tile_blk, npass.p, n_tiles, groups.p);
    std::vector<int> h_groups(nw), h_base(nw);
    MLLP_HIP_TRY(hipMemcpyAsync(h_groups.data(), groups.p, (size_t)nw * 4, hipMemcpyDeviceToHost, s));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    int64_t n_groups = 0;
    for (int i = 0; i < nw; ++i) {
        h_base[i] = (int)n_groups;
        n_groups += h_groups[i];
        if (n_groups * S_GS >= ((int64_t)1 << 31) - S_GS * S_K0) return fail(MLLP_ERANGE, "streamed copy: more than 2^31 steps");
    }
    sc.n_groups = n_groups;
    MLLP_HIP_TRY(hipMemcpyAsync(base.p, h_base.data(), (size_t)nw * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(sb_step_starts<G>, dim3((nw + 255) / 256), dim3(256), 0, s, sc.tile_blk, npass.p, n_tiles, base.p,
                       sstart.p);
    const long long n_ent = (long long)(n_groups + S_K0) * 64;
    MLLP_HIP_TRY(hipMalloc((void**)&sc.ent, (size_t)n_ent * S_ENT * 4));
    hipLaunchKernelGGL(sb_fill_padding<G>, dim3(4096), dim3(256), 0, s, sc.ent, n_ent);
    if (n_tb > 0)
        hipLaunchKernelGGL(sb_fill<G>, dim3((unsigned)n_tb), dim3(SbFill<G>::T), 0, s, o.idx, o.val, sc.blk_id, cnt.p, start.p,
                           order.p, npass.p, sstart.p, reinterpret_cast<int4*>(sc.rows), reinterpret_cast<int4*>(sc.hdr),
                           sc.ent);
    MLLP_HIP_TRY(hipGetLastError());
    sc.step_slots = n_groups * 128;      // entry slots of the stream, padding included
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    return MLLP_OK;
}

int build_stream_device(const Orient& o, int64_t nnz, const std::vector<int>& tile_row, StreamCopy& sc, hipStream_t s, int geom) {
    switch (geom) {
        case STREAM_GEOM_SPMM: return build_stream_device_t<SpmmGeom>(o, nnz, tile_row, sc, s);
        case STREAM_GEOM_ATTN: return build_stream_device_t<AttnGeom>(o, nnz, tile_row, sc, s);
        case STREAM_GEOM_BSRC: return build_stream_device_t<BsrcGeom>(o, nnz, tile_row, sc, s);
        case STREAM_GEOM_BDST: return build_stream_device_t<BdstGeom>(o, nnz, tile_row, sc, s);
        default: return fail(MLLP_EINVAL, "streamed copy: unknown geometry");
    }
}

}  // namespace mllp
