// graph.cpp -- build the HBM-resident block-diagonal batch (CSR of A and of A^T + row tiers).
//
// Replaces reference linear_program_methods.py:89-103 (per-step Python edge-list build) and :60-72
// (BipartiteData.__inc__ batching offsets): done ONCE per batch, on the host, in parallel over
// instances (they are independent blocks of the block-diagonal matrix), then uploaded.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "internal.h"

namespace mllp {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? MLLP_ENOMEM : MLLP_EHIP;
}
const char* last_error_cstr() { return g_err.c_str(); }

template <class T>
static int upload(mllp_graph* g, const T* host, size_t count, T** dev) {
    *dev = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void* p = nullptr;
    MLLP_HIP_TRY(hipMalloc(&p, bytes));
    g->allocs.push_back(p);
    if (count) MLLP_HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T*>(p);
    return MLLP_OK;
}

static void choose_tiers(mllp_graph* g) {
    // Few rows (real Netlib, ~1M nonzeros): a sweep is latency bound, so long rows are spread over many
    // lanes early and very long rows over several workgroups; every work item then loops <= ~4 times.
    // Many rows (synthetic, 5e8 nonzeros): throughput bound, 16 lanes per row keep every lane busy and
    // need no cross-wave merge.
    const bool throughput = g->nnz >= (int64_t)32 << 20;
    if (g->tier_wave <= 0) g->tier_wave = throughput ? 1024 : 64;
    if (g->tier_block <= 0) g->tier_block = throughput ? 16384 : 256;
    if (g->tier_block < g->tier_wave) g->tier_block = g->tier_wave;
    g->chunk_nnz = 4 * g->tier_block;
}

static int build_tiers(mllp_graph* g, Orient& o, const int* h_ptr) {
    std::vector<int> rg, rw, ck, sp;
    int slots = 0;
    for (int r = 0; r < o.n_dst; ++r) {
        const int beg = h_ptr[r], end = h_ptr[r + 1], deg = end - beg;
        if (deg > g->tier_block) {
            const int nck = (deg + g->chunk_nnz - 1) / g->chunk_nnz;
            if (nck == 1) {
                ck.insert(ck.end(), {r, beg, end, -1});
            } else {
                sp.insert(sp.end(), {r, slots, nck, 0});
                const int per = ((deg + nck - 1) / nck + 255) & ~255;   // equal shares, whole 256-nonzero passes
                for (int c = 0; c < nck; ++c) {
                    const int cb = std::min(beg + c * per, end), ce = std::min(cb + per, end);
                    ck.insert(ck.end(), {r, cb, ce, slots++});   // an empty tail chunk merges as a neutral state
                }
            }
        } else if (deg > g->tier_wave) rw.push_back(r);
        else rg.push_back(r);
    }
    o.n_group = (int)rg.size();
    o.tier_wave = g->tier_wave;
    o.short_rows = o.n_dst > 0 && (double)h_ptr[o.n_dst] / o.n_dst <= 16.0;
    o.n_wave = (int)rw.size();
    o.n_chunk = (int)(ck.size() / 4);
    o.n_split = (int)(sp.size() / 4);
    o.n_slots = slots;
    int rc;
    o.rows_group = nullptr;  // the group tier visits every row and skips the long ones in-kernel
    if ((rc = upload(g, rw.data(), rw.size(), &o.rows_wave))) return rc;
    if ((rc = upload(g, ck.data(), ck.size(), &o.chunks))) return rc;
    if ((rc = upload(g, sp.data(), sp.size(), &o.split))) return rc;
    return MLLP_OK;
}

static int finish_common(mllp_graph* g) {
    // per-variable 1/n_k and device copies of the instance offsets
    std::vector<float> inv_n((size_t)g->N);
    std::vector<int> ipn(g->n_inst + 1), ipm(g->n_inst + 1);
    g->max_inst_n = 0;
    for (int64_t k = 0; k < g->n_inst; ++k) {
        int64_t a = g->h_inst_ptr_n[k], b = g->h_inst_ptr_n[k + 1];
        for (int64_t i = a; i < b; ++i) inv_n[i] = 1.0f / (float)(b - a);
        g->max_inst_n = std::max<int>(g->max_inst_n, (int)(b - a));
    }
    for (int64_t k = 0; k <= g->n_inst; ++k) {
        ipn[k] = (int)g->h_inst_ptr_n[k];
        ipm[k] = (int)g->h_inst_ptr_m[k];
    }
    int rc;
    for (Orient* o : {&g->A, &g->At}) {
        void* p = nullptr;
        MLLP_HIP_TRY(hipMalloc(&p, (size_t)std::max(o->n_slots, 1) * SCRATCH_NS * sizeof(float)));
        g->allocs.push_back(p);
        o->scratch = static_cast<float*>(p);
    }
    MLLP_HIP_TRY(hipStreamCreateWithFlags(&g->aux, hipStreamNonBlocking));
    for (auto& e : g->ev) MLLP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if ((rc = upload(g, inv_n.data(), inv_n.size(), &g->inv_n))) return rc;
    if ((rc = upload(g, ipn.data(), ipn.size(), &g->inst_ptr_n))) return rc;
    if ((rc = upload(g, ipm.data(), ipm.size(), &g->inst_ptr_m))) return rc;
    return MLLP_OK;
}

}  // namespace mllp

using namespace mllp;

extern "C" const char* mllp_last_error(void) { return mllp::last_error_cstr(); }
extern "C" int mllp_abi_version(void) { return MLLP_ABI_VERSION; }

extern "C" int mllp_graph_create_host(int64_t n_inst, const int64_t* inst_m, const int64_t* inst_n,
                                      const int64_t* indptr, const int32_t* indices, const double* values,
                                      int32_t tier_wave, int32_t tier_block, mllp_graph_t** out) {
    if (!out) return fail(MLLP_EINVAL, "mllp_graph_create_host: out is null");
    *out = nullptr;
    if (n_inst < 0 || (n_inst > 0 && (!inst_m || !inst_n || !indptr)))
        return fail(MLLP_EINVAL, "mllp_graph_create_host: null instance arrays");
    std::vector<int64_t> pm(n_inst + 1, 0), pn(n_inst + 1, 0), pe(n_inst + 1, 0), pp(n_inst + 1, 0);
    for (int64_t k = 0; k < n_inst; ++k) {
        if (inst_m[k] < 0 || inst_n[k] < 0) return fail(MLLP_EINVAL, "negative instance size");
        pm[k + 1] = pm[k] + inst_m[k];
        pn[k + 1] = pn[k] + inst_n[k];
        pp[k + 1] = pp[k] + inst_m[k] + 1;
        const int64_t* ip = indptr + pp[k];
        if (ip[0] != 0) return fail(MLLP_EINVAL, "indptr block does not start at 0");
        pe[k + 1] = pe[k] + ip[inst_m[k]];
    }
    int64_t M = pm[n_inst], N = pn[n_inst], nnz = pe[n_inst];
    if (nnz >= INT32_MAX - 1 || M >= INT32_MAX - 1 || N >= INT32_MAX - 1)
        return fail(MLLP_ERANGE, "batch exceeds int32 indexing");
    if (nnz > 0 && (!indices || !values)) return fail(MLLP_EINVAL, "null indices/values");

    std::vector<int> csr_ptr(M + 1), csr_idx(nnz), csc_ptr(N + 1), csc_idx(nnz);
    std::vector<float> csr_val(nnz), csc_val(nnz);
    csr_ptr[0] = 0;
    csc_ptr[0] = 0;
    std::atomic<int> bad{0};
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        std::vector<int> cnt;
        for (;;) {
            int64_t k = next.fetch_add(1);
            if (k >= n_inst) break;
            int64_t m = inst_m[k], n = inst_n[k], e0 = pe[k];
            const int64_t* ip = indptr + pp[k];
            const int32_t* ix = indices + e0;
            const double* va = values + e0;
            cnt.assign(n + 1, 0);
            for (int64_t r = 0; r < m; ++r) {
                if (ip[r + 1] < ip[r]) { bad = 1; break; }
                csr_ptr[pm[k] + r + 1] = (int)(e0 + ip[r + 1]);
                int prev = -1;
                for (int64_t e = ip[r]; e < ip[r + 1]; ++e) {
                    int c = ix[e];
                    if (c < 0 || c >= n || c <= prev) { bad = 2; break; }  // sorted, unique, in range
                    prev = c;
                    csr_idx[e0 + e] = (int)(pn[k] + c);
                    csr_val[e0 + e] = (float)va[e];
                    cnt[c + 1]++;
                }
            }
            if (bad) break;
            // counting sort by column: stable, so constraint ids ascend within each column
            for (int64_t c = 0; c < n; ++c) cnt[c + 1] += cnt[c];
            for (int64_t c = 0; c < n; ++c) csc_ptr[pn[k] + c + 1] = (int)(e0 + cnt[c + 1]);
            for (int64_t r = 0; r < m; ++r)
                for (int64_t e = ip[r]; e < ip[r + 1]; ++e) {
                    int c = ix[e];
                    int64_t pos = e0 + cnt[c]++;
                    csc_idx[pos] = (int)(pm[k] + r);
                    csc_val[pos] = (float)va[e];
                }
        }
    };
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    nt = (unsigned)std::min<int64_t>(nt, std::max<int64_t>(n_inst, 1));
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (bad == 1) return fail(MLLP_EINVAL, "indptr not monotone");
    if (bad == 2) return fail(MLLP_EINVAL, "column ids must be in range, sorted and unique within a row");

    mllp_graph* g = new mllp_graph();
    g->M = M; g->N = N; g->nnz = nnz; g->n_inst = n_inst;
    g->h_inst_ptr_m = pm;
    g->h_inst_ptr_n = pn;
    g->tier_wave = tier_wave;
    g->tier_block = tier_block;
    choose_tiers(g);
    g->A.n_dst = (int)M; g->A.n_src = (int)N;
    g->At.n_dst = (int)N; g->At.n_src = (int)M;
    int rc = MLLP_OK;
    do {
        if ((rc = upload(g, csr_ptr.data(), csr_ptr.size(), &g->A.ptr))) break;
        if ((rc = upload(g, csr_idx.data(), csr_idx.size(), &g->A.idx))) break;
        if ((rc = upload(g, csr_val.data(), csr_val.size(), &g->A.val))) break;
        if ((rc = upload(g, csc_ptr.data(), csc_ptr.size(), &g->At.ptr))) break;
        if ((rc = upload(g, csc_idx.data(), csc_idx.size(), &g->At.idx))) break;
        if ((rc = upload(g, csc_val.data(), csc_val.size(), &g->At.val))) break;
        if ((rc = build_tiers(g, g->A, csr_ptr.data()))) break;
        if ((rc = build_tiers(g, g->At, csc_ptr.data()))) break;
        if ((rc = finish_common(g))) break;
    } while (0);
    if (rc) {
        std::string keep = mllp_last_error();
        mllp_graph_destroy(g);
        set_error(keep);
        return rc;
    }
    *out = g;
    return MLLP_OK;
}

extern "C" int mllp_graph_create_device(int64_t n_inst, const int64_t* inst_ptr_m, const int64_t* inst_ptr_n,
                                        int64_t nnz, const int32_t* d_csr_ptr, const int32_t* d_csr_idx,
                                        const float* d_csr_val, const int32_t* d_csc_ptr, const int32_t* d_csc_idx,
                                        const float* d_csc_val, int32_t tier_wave, int32_t tier_block, void* stream,
                                        mllp_graph_t** out) {
    if (!out) return fail(MLLP_EINVAL, "mllp_graph_create_device: out is null");
    *out = nullptr;
    if (n_inst <= 0 || !inst_ptr_m || !inst_ptr_n || !d_csr_ptr || !d_csc_ptr || nnz < 0)
        return fail(MLLP_EINVAL, "mllp_graph_create_device: bad arguments");
    if (nnz > 0 && (!d_csr_idx || !d_csr_val || !d_csc_idx || !d_csc_val))
        return fail(MLLP_EINVAL, "mllp_graph_create_device: null index/value arrays");
    int64_t M = inst_ptr_m[n_inst], N = inst_ptr_n[n_inst];
    if (nnz >= INT32_MAX - 1 || M >= INT32_MAX - 1 || N >= INT32_MAX - 1)
        return fail(MLLP_ERANGE, "batch exceeds int32 indexing");
    hipStream_t s = (hipStream_t)stream;
    mllp_graph* g = new mllp_graph();
    g->M = M; g->N = N; g->nnz = nnz; g->n_inst = n_inst;
    g->h_inst_ptr_m.assign(inst_ptr_m, inst_ptr_m + n_inst + 1);
    g->h_inst_ptr_n.assign(inst_ptr_n, inst_ptr_n + n_inst + 1);
    g->tier_wave = tier_wave;
    g->tier_block = tier_block;
    choose_tiers(g);
    g->A.n_dst = (int)M; g->A.n_src = (int)N;
    g->At.n_dst = (int)N; g->At.n_src = (int)M;
    auto dcopy = [&](const void* src, size_t bytes, void** dst) -> int {
        void* p = nullptr;
        MLLP_HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 4)));
        g->allocs.push_back(p);
        if (bytes) MLLP_HIP_TRY(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToDevice, s));
        *dst = p;
        return MLLP_OK;
    };
    int rc = MLLP_OK;
    std::vector<int> h_rp(M + 1), h_cp(N + 1);
    do {
        if ((rc = dcopy(d_csr_ptr, (M + 1) * 4, (void**)&g->A.ptr))) break;
        if ((rc = dcopy(d_csr_idx, nnz * 4, (void**)&g->A.idx))) break;
        if ((rc = dcopy(d_csr_val, nnz * 4, (void**)&g->A.val))) break;
        if ((rc = dcopy(d_csc_ptr, (N + 1) * 4, (void**)&g->At.ptr))) break;
        if ((rc = dcopy(d_csc_idx, nnz * 4, (void**)&g->At.idx))) break;
        if ((rc = dcopy(d_csc_val, nnz * 4, (void**)&g->At.val))) break;
        hipError_t e = hipMemcpyAsync(h_rp.data(), d_csr_ptr, (M + 1) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_cp.data(), d_csc_ptr, (N + 1) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { rc = hip_fail(e, "copy row pointers to host"); break; }
        if (h_rp[0] != 0 || h_cp[0] != 0 || h_rp[M] != nnz || h_cp[N] != nnz) {
            rc = fail(MLLP_EINVAL, "row pointers inconsistent with nnz");
            break;
        }
        if ((rc = build_tiers(g, g->A, h_rp.data()))) break;
        if ((rc = build_tiers(g, g->At, h_cp.data()))) break;
        if ((rc = finish_common(g))) break;
    } while (0);
    if (rc) {
        std::string keep = mllp_last_error();
        mllp_graph_destroy(g);
        set_error(keep);
        return rc;
    }
    *out = g;
    return MLLP_OK;
}

extern "C" int mllp_graph_destroy(mllp_graph_t* g) {
    if (!g) return MLLP_OK;
    for (void* p : g->allocs) (void)hipFree(p);
    for (auto& e : g->ev)
        if (e) (void)hipEventDestroy(e);
    if (g->aux) (void)hipStreamDestroy(g->aux);
    delete g;
    return MLLP_OK;
}

extern "C" int mllp_graph_dims(const mllp_graph_t* g, int64_t dims[12]) {
    if (!g || !dims) return fail(MLLP_EINVAL, "mllp_graph_dims: null argument");
    dims[0] = g->M; dims[1] = g->N; dims[2] = g->nnz; dims[3] = g->n_inst;
    dims[4] = g->A.n_group; dims[5] = g->A.n_wave; dims[6] = g->A.n_chunk;
    dims[7] = g->At.n_group; dims[8] = g->At.n_wave; dims[9] = g->At.n_chunk;
    dims[10] = g->A.n_split; dims[11] = g->At.n_split;
    return MLLP_OK;
}

extern "C" int mllp_graph_export(const mllp_graph_t* g, int which, void* host_dst, int64_t capacity_bytes) {
    if (!g || !host_dst) return fail(MLLP_EINVAL, "mllp_graph_export: null argument");
    const void* src = nullptr;
    int64_t bytes = 0;
    switch (which) {
        case 0: src = g->A.ptr; bytes = (g->M + 1) * 4; break;
        case 1: src = g->A.idx; bytes = g->nnz * 4; break;
        case 2: src = g->A.val; bytes = g->nnz * 4; break;
        case 3: src = g->At.ptr; bytes = (g->N + 1) * 4; break;
        case 4: src = g->At.idx; bytes = g->nnz * 4; break;
        case 5: src = g->At.val; bytes = g->nnz * 4; break;
        case 6: src = g->inv_n; bytes = g->N * 4; break;
        default: return fail(MLLP_EINVAL, "mllp_graph_export: unknown array id");
    }
    if (capacity_bytes < bytes) return fail(MLLP_EINVAL, "mllp_graph_export: destination too small");
    MLLP_HIP_TRY(hipDeviceSynchronize());
    if (bytes) MLLP_HIP_TRY(hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return MLLP_OK;
}
