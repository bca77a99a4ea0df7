// host_graph.h -- the pure-host half of the batch build: validation of the per-instance CSR blocks, the
// block-diagonal CSR(A) + CSR(A^T) (parallel over instances) and the row tiers.  No HIP types: this file and
// host_graph.cpp also compile with plain g++, which is how the AddressSanitizer / ThreadSanitizer build of the
// C-ABI's host code is made (`make host-sanitize`, tests/test_host_sanitize.py).
// Replaces reference linear_program_methods.py:89-103 (edge-list build) and :60-72 (__inc__ batching offsets).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace mllp {

struct HostBatch {
    int64_t M = 0, N = 0, nnz = 0, n_inst = 0;
    std::vector<int64_t> pm, pn;                 // [n_inst + 1] row / column offsets of the instances
    std::vector<int> csr_ptr, csr_idx;           // rows = constraints, global variable ids
    std::vector<float> csr_val;
    std::vector<int> csc_ptr, csc_idx;           // rows = variables, global constraint ids ascending inside a row
    std::vector<float> csc_val;
};

// 0 on success; otherwise an MLLP_E* code with *err set.  max_threads = 0: hardware concurrency, at most 16.
int host_build_batch(int64_t n_inst, const int64_t* inst_m, const int64_t* inst_n, const int64_t* indptr,
                     const int32_t* indices, const double* values, HostBatch* out, std::string* err,
                     unsigned max_threads = 0);

struct TierConfig {
    int tier_wave = 0, tier_block = 0, chunk_nnz = 0;
};
// thresholds of the row tiers (0 = choose by the size of the batch: latency regime < 32 M nonzeros <= throughput regime)
TierConfig host_choose_tiers(int64_t nnz, int tier_wave, int tier_block);

struct HostTiers {
    std::vector<int> rows_wave;   // rows of the wave tier
    std::vector<int> chunks;      // int4 {row, beg, end, slot}; slot < 0: the row's only chunk
    std::vector<int> split;       // int4 {row, first_slot, n_chunks, 0}
    int n_group = 0, n_slots = 0;
    bool short_rows = false;
};
void host_build_tiers(const int* ptr, int n_dst, const TierConfig& cfg, HostTiers* out);

// Work list of the fused latency-regime kernels (fused_kernels.hip): the rows of one orientation ordered by their
// number of nonzeros, descending (ties in row order), cut in three tiers:
//   block tier  deg > wave_max_deg    one whole workgroup walks the row
//   wave tier   deg > quad_max_deg    one wavefront walks the row
//   quad tier   the rest (incl. empty rows): 16 rows per wavefront (16-channel sweeps: a quad of lanes per row) or
//               64 rows per wavefront (1-channel sweeps: a lane per row); neighbours in the order have equal or
//               nearly equal length, so the lanes of a wavefront finish together
struct HostItems {
    std::vector<int> rows;    // [n_dst]
    int n_block = 0, n_wave = 0, n_quad = 0;
};
constexpr int ITEM_QUAD_MAX_DEG = 32, ITEM_WAVE_MAX_DEG = 512;
void host_build_items(const int* ptr, int n_dst, int quad_max_deg, int wave_max_deg, HostItems* out);

}  // namespace mllp
