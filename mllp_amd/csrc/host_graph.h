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

// Node renumbering of the fused latency-regime kernels (fused_kernels.hip).  Constraints are renumbered by their
// row length in A, variables by their column length (= row length in A^T), both descending with ties in the original
// order.  In renumbered ids every sweep walks rows 0, 1, 2, ... whose lengths fall monotonically, so the rows a
// wavefront shares have (nearly) equal length, every per-node tensor of the model is read and written with unit
// stride, and the row tiers are contiguous id ranges:
//   block  deg > T[2]   the whole workgroup walks the row
//   wave   deg > T[1]   one wavefront
//   group  deg > T[0]   a quarter of a wavefront (4 rows per wavefront)
//   base   the rest (incl. empty rows): 16 rows per wavefront (a quad of lanes per row, 16-channel sweeps) or 64
//          (a lane per row, 1-channel sweeps)
// The block-diagonal structure is kept for locality: the instances are dealt to FUSED_PARTS partitions of (nearly)
// equal nonzero count (longest-processing-time rule) and the renumbering is by (partition, length descending, original
// id), the same partition for a constraint / variable and for every node it exchanges messages with.  Workgroup b of a
// fused kernel works on partition b mod 8 -- the dispatcher deals workgroups round-robin over the 8 XCDs, so a
// partition's gathers (an eighth of every node tensor) stay in one XCD's 4 MB L2.  Speed only: any placement is correct.
constexpr int FUSED_PARTS = 8;
struct FusedTiers {
    int n_block = 0, n_wave = 0, n_group = 0, n_base = 0;
};
struct HostFusedOrient {
    std::vector<int> perm;      // [n] renumbered id -> original id
    std::vector<int> inv;       // [n] original id -> renumbered id
    std::vector<int> sptr;      // [n + 1] entry offsets of the renumbered rows
    int row0[FUSED_PARTS + 1];  // first renumbered id of every partition
    FusedTiers t16[FUSED_PARTS], t1[FUSED_PARTS];   // tiers of the 16-channel / 1-channel sweeps inside each partition
};
// quad / group (4 quads) / wave (16 quads) / block.  Every step of a sweep (4 nonzeros per quad) is one round trip
// to L2, about 2 k cycles whatever its width (in-kernel stamps): a wave-tier row of 1024 nonzeros keeps its wavefront for
// ~40 k cycles, the whole share of an average wavefront in a Netlib launch, while a block-tier row costs all 12
// wavefronts of its workgroup ~25 k cycles each (LDS merges, barriers).  (Measured: block tier above 2048 -> the
// wavefront that drew the longest row ran 116 k cycles and set the launch; above 256 -> 247 block rows.)
constexpr int FUSED_T16[3] = {16, 64, 1024};
constexpr int FUSED_T1[3] = {16, 64, 4096};      // lane / group (4 lanes, 16 rows per item) / wave (64 lanes) / block
// Cost model of the sweeps, in cycles of one wavefront (stamps of fused_fwd16_kernel on the Netlib batch)
constexpr int64_t FUSED_COST_ITEM = 5300, FUSED_COST_STEP = 2000, FUSED_COST_BLOCK_ROW = 25000;
// Estimated cycles of one 16-channel sweep over every instance's rows of one orientation (ptr: row pointers, inst_off:
// [n_inst + 1] row offsets of the instances): items of 16 base rows, 4 group rows or one wave row; a block row charges
// every wavefront of a workgroup.
std::vector<int64_t> host_instance_cost(const int* ptr, const std::vector<int64_t>& inst_off);
// Partition of every instance, deterministic.  A sweep walks either the constraints or the variables, so the greedy
// rule (largest instance first, to the partition whose worse side stays lowest) balances the two estimated costs
// together.  (With nonzeros alone the slowest partition of the Netlib batch had 1.7 x the mean number of items; with
// 12 rows + nonzeros the tiers were ignored -- a row of 17..64 nonzeros costs four times a row of 16 -- and the
// slowest partition still ran 1.3 x the mean.)
std::vector<int> host_partition_instances(const std::vector<int64_t>& cost_a, const std::vector<int64_t>& cost_b, int n_parts);
// Round 4.  The five kernel families of the fused path (forward-16, backward-16, source-16, forward-1, backward-1) have
// different costs per item and per step, so ONE scalar cost per instance balanced the model and not the measurement: the
// stamps of round 3 showed whole partitions 20 % under and over the mean (profiles/r03_fused_stamps.txt).  The time of a
// partition in any of these kernels is a_k * items + b_k * steps + c_k * block rows of the orientation the kernel walks,
// with kernel-specific a, b, c: a partition that holds an eighth of EACH of these quantities, for both orientations and
// both geometries (16-channel: 16 base rows / 4 group rows / 1 wave row per item, 4 nonzeros per quad and step; 1-channel:
// 64 / 16 / 1, 8 nonzeros per lane and step), is balanced for all of them whatever the coefficients are.
// (a block row in items: the model of rounds 2-3 charged 25 k cycles to each of the 12 wavefronts = 43 items, and the partition
// that held the Netlib batch's block rows ran 25 % under the mean; the stamps of round 4 fit 10)
constexpr int FUSED_BLOCK_ROW_ITEMS = 10;
constexpr int FUSED_LOAD_DIMS = 6;       // {items16, steps16, items1, steps1, block rows, modelled cycles of a 16-channel sweep}
                                         // of one orientation, fixed point
struct InstLoad {
    int64_t d[FUSED_LOAD_DIMS] = {0, 0, 0, 0, 0, 0};
};
std::vector<InstLoad> host_instance_loads(const int* ptr, const std::vector<int64_t>& inst_off);
// Deterministic.  Largest instance first, each to the partition that keeps the largest normalised load over all 2 x 5
// quantities lowest; then pairwise moves / swaps while they lower that maximum.
std::vector<int> host_partition_instances_v(const std::vector<InstLoad>& a, const std::vector<InstLoad>& b, int n_parts);
// max over partitions and quantities of (load of the partition) / (an n_parts-th of the total): 1.0 = perfectly even
double host_partition_imbalance(const std::vector<InstLoad>& a, const std::vector<InstLoad>& b, const std::vector<int>& part,
                                int n_parts);
// inst_off: [n_inst + 1] offsets of the instances' nodes (rows of this orientation)
void host_build_fused_orient(const int* ptr, int n, const std::vector<int64_t>& inst_off, const std::vector<int>& inst_part,
                             HostFusedOrient* out);
// Which items every wavefront of a sweep works on.  The time of a launch is its slowest wavefront's, and dealing the
// items round-robin left that one at 1.5 x the mean (group items cost several base items, the workgroups that drew
// block rows ran 25 k cycles late).  Static longest-processing-time assignment instead (deterministic: the statistics
// a wavefront accumulates are summed in a fixed order): wavefronts are charged the block rows of their workgroup first
// (row k of a partition goes to workgroup k mod gp), then every item, heaviest first, goes to the least loaded
// wavefront of its partition.  order[(q * waves_per_part + w) * L + k] = k-th item of wavefront w of partition q, -1 =
// none.  `scalar`: the 1-channel geometry (64 base rows / 16 group rows per item) instead of 16 / 4.
struct HostWaveLists {
    std::vector<int> order;
    int L = 0, waves_per_part = 0;
    int64_t max_load = 0, sum_load = 0;     // of the model, over all wavefronts (informational; tests)
};
// `nnz_per_step`: nonzeros of a base-tier row that one round trip of the sweep covers (4: forward / destination-major
// backward, 2: source-major backward, 8: the 1-channel sweeps)
void host_build_wave_lists(const HostFusedOrient& o, bool scalar, int waves_per_part, int waves_per_wg, int nnz_per_step,
                           HostWaveLists* out);

}  // namespace mllp
