// graph.cpp -- build the HBM-resident block-diagonal batch (CSR of A and of A^T + row tiers).
//
// Replaces reference linear_program_methods.py:89-103 (per-step Python edge-list build) and :60-72
// (BipartiteData.__inc__ batching offsets): done ONCE per batch, on the host, in parallel over
// instances (they are independent blocks of the block-diagonal matrix), then uploaded.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "host_graph.h"
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
    const TierConfig c = host_choose_tiers(g->nnz, g->tier_wave, g->tier_block);
    g->tier_wave = c.tier_wave; g->tier_block = c.tier_block; g->chunk_nnz = c.chunk_nnz;
}

static int build_tiers(mllp_graph* g, Orient& o, const int* h_ptr) {
    HostTiers t;
    TierConfig c;
    c.tier_wave = g->tier_wave; c.tier_block = g->tier_block; c.chunk_nnz = g->chunk_nnz;
    host_build_tiers(h_ptr, o.n_dst, c, &t);
    o.n_group = t.n_group;
    o.tier_wave = g->tier_wave;
    o.short_rows = t.short_rows;
    o.n_wave = (int)t.rows_wave.size();
    o.n_chunk = (int)(t.chunks.size() / 4);
    o.n_split = (int)(t.split.size() / 4);
    o.n_slots = t.n_slots;
    int rc;
    o.rows_group = nullptr;  // the group tier visits every row and skips the long ones in-kernel
    if ((rc = upload(g, t.rows_wave.data(), t.rows_wave.size(), &o.rows_wave))) return rc;
    if ((rc = upload(g, t.chunks.data(), t.chunks.size(), &o.chunks))) return rc;
    if ((rc = upload(g, t.split.data(), t.split.size(), &o.split))) return rc;
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
    {
        int dev = 0, cus = 0;
        MLLP_HIP_TRY(hipGetDevice(&dev));
        MLLP_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        g->n_cu = cus > 0 ? cus : 256;
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
    HostBatch hb;
    {
        std::string err;
        const int hrc = host_build_batch(n_inst, inst_m, inst_n, indptr, indices, values, &hb, &err);
        if (hrc) return fail(hrc, err);
    }
    const int64_t M = hb.M, N = hb.N, nnz = hb.nnz;
    const std::vector<int>&csr_ptr = hb.csr_ptr, &csr_idx = hb.csr_idx, &csc_ptr = hb.csc_ptr, &csc_idx = hb.csc_idx;
    const std::vector<float>&csr_val = hb.csr_val, &csc_val = hb.csc_val;
    const std::vector<int64_t>&pm = hb.pm, &pn = hb.pn;

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
        g->h_csr_ptr = csr_ptr; g->h_csc_ptr = csc_ptr;
        if (g->nnz < ((int64_t)32 << 20) && (rc = fused_graph_build(g, csr_ptr.data(), csc_ptr.data()))) break;
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
        g->h_csr_ptr = h_rp; g->h_csc_ptr = h_cp;
        if (g->nnz < ((int64_t)32 << 20) && (rc = fused_graph_build(g, h_rp.data(), h_cp.data()))) break;
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
    for (mllp::Orient* o : {&g->A, &g->At})
        for (mllp::StreamCopy* sc : {&o->stream, &o->stream_attn, &o->stream_bsrc, &o->stream_bdst}) mllp::stream_copy_free(*sc);
    for (mllp::Orient* o : {&g->A, &g->At}) mllp::lane_copy_free(o->lane1);
    for (mllp::Orient* o : {&g->A, &g->At})
        for (mllp::Tiled* tl : {&o->tiled, &o->tiled_attn, &o->tiled_bsrc, &o->tiled_scalar, &o->tiled_bdst}) mllp::tiled_free(*tl);
    for (void* p : g->allocs) (void)hipFree(p);
    if (g->tail_err_host) (void)hipHostFree(g->tail_err_host);
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
