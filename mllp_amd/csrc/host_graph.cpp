// host_graph.cpp -- see host_graph.h.  Plain C++17, no HIP.
#include "host_graph.h"

#include <array>

#include <algorithm>
#include <atomic>
#include <thread>

#include "../../include/mllp_hip.h"

namespace mllp {

int host_build_batch(int64_t n_inst, const int64_t* inst_m, const int64_t* inst_n, const int64_t* indptr,
                     const int32_t* indices, const double* values, HostBatch* out, std::string* err,
                     unsigned max_threads) {
    auto bad_arg = [&](int code, const char* msg) {
        if (err) *err = msg;
        return code;
    };
    if (!out) return bad_arg(MLLP_EINVAL, "host_build_batch: out is null");
    if (n_inst < 0 || (n_inst > 0 && (!inst_m || !inst_n || !indptr)))
        return bad_arg(MLLP_EINVAL, "mllp_graph_create_host: null instance arrays");
    std::vector<int64_t> pm(n_inst + 1, 0), pn(n_inst + 1, 0), pe(n_inst + 1, 0), pp(n_inst + 1, 0);
    for (int64_t k = 0; k < n_inst; ++k) {
        if (inst_m[k] < 0 || inst_n[k] < 0) return bad_arg(MLLP_EINVAL, "negative instance size");
        pm[k + 1] = pm[k] + inst_m[k];
        pn[k + 1] = pn[k] + inst_n[k];
        pp[k + 1] = pp[k] + inst_m[k] + 1;
        const int64_t* ip = indptr + pp[k];
        if (ip[0] != 0) return bad_arg(MLLP_EINVAL, "indptr block does not start at 0");
        if (ip[inst_m[k]] < 0) return bad_arg(MLLP_EINVAL, "indptr not monotone");
        pe[k + 1] = pe[k] + ip[inst_m[k]];
    }
    const int64_t M = pm[n_inst], N = pn[n_inst], nnz = pe[n_inst];
    if (nnz >= INT32_MAX - 1 || M >= INT32_MAX - 1 || N >= INT32_MAX - 1)
        return bad_arg(MLLP_ERANGE, "batch exceeds int32 indexing");
    if (nnz > 0 && (!indices || !values)) return bad_arg(MLLP_EINVAL, "null indices/values");

    HostBatch& b = *out;
    b.M = M; b.N = N; b.nnz = nnz; b.n_inst = n_inst;
    b.csr_ptr.assign(M + 1, 0); b.csr_idx.assign(nnz, 0); b.csr_val.assign(nnz, 0.0f);
    b.csc_ptr.assign(N + 1, 0); b.csc_idx.assign(nnz, 0); b.csc_val.assign(nnz, 0.0f);
    // instances are independent blocks: every worker writes only the index ranges of the instances it draws
    std::atomic<int> bad{0};
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        std::vector<int> cnt;
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_inst || bad.load()) break;
            const int64_t m = inst_m[k], n = inst_n[k], e0 = pe[k], e1 = pe[k + 1];
            const int64_t* ip = indptr + pp[k];
            const int32_t* ix = indices + e0;
            const double* va = values + e0;
            cnt.assign(n + 1, 0);
            bool ok = true;
            for (int64_t r = 0; r < m && ok; ++r) {
                if (ip[r + 1] < ip[r] || ip[r + 1] > e1 - e0) { bad = 1; ok = false; break; }
                b.csr_ptr[pm[k] + r + 1] = (int)(e0 + ip[r + 1]);
                int prev = -1;
                for (int64_t e = ip[r]; e < ip[r + 1]; ++e) {
                    const int c = ix[e];
                    if (c < 0 || c >= n || c <= prev) { bad = 2; ok = false; break; }  // sorted, unique, in range
                    prev = c;
                    b.csr_idx[e0 + e] = (int)(pn[k] + c);
                    b.csr_val[e0 + e] = (float)va[e];
                    cnt[c + 1]++;
                }
            }
            if (!ok) break;
            // counting sort by column: stable, so constraint ids ascend within each column
            for (int64_t c = 0; c < n; ++c) cnt[c + 1] += cnt[c];
            for (int64_t c = 0; c < n; ++c) b.csc_ptr[pn[k] + c + 1] = (int)(e0 + cnt[c + 1]);
            for (int64_t r = 0; r < m; ++r)
                for (int64_t e = ip[r]; e < ip[r + 1]; ++e) {
                    const int c = ix[e];
                    const int64_t pos = e0 + cnt[c]++;
                    b.csc_idx[pos] = (int)(pm[k] + r);
                    b.csc_val[pos] = (float)va[e];
                }
        }
    };
    unsigned nt = max_threads ? max_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    nt = (unsigned)std::min<int64_t>(nt, std::max<int64_t>(n_inst, 1));
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (bad == 1) return bad_arg(MLLP_EINVAL, "indptr not monotone");
    if (bad == 2) return bad_arg(MLLP_EINVAL, "column ids must be in range, sorted and unique within a row");
    b.pm = std::move(pm);
    b.pn = std::move(pn);
    return MLLP_OK;
}

TierConfig host_choose_tiers(int64_t nnz, int tier_wave, int tier_block) {
    // Few rows (real Netlib, ~1M nonzeros): a sweep is latency bound, so long rows are spread over many
    // lanes early and very long rows over several workgroups; every work item then loops <= ~4 times.
    // Many rows (synthetic, 5e8 nonzeros): throughput bound, 16 lanes per row keep every lane busy and
    // need no cross-wave merge.
    const bool throughput = nnz >= (int64_t)32 << 20;
    TierConfig c;
    c.tier_wave = tier_wave > 0 ? tier_wave : (throughput ? 1024 : 64);
    c.tier_block = tier_block > 0 ? tier_block : (throughput ? 16384 : 256);
    if (c.tier_block < c.tier_wave) c.tier_block = c.tier_wave;
    c.chunk_nnz = 4 * c.tier_block;
    return c;
}

void host_build_tiers(const int* ptr, int n_dst, const TierConfig& cfg, HostTiers* out) {
    HostTiers& t = *out;
    t = HostTiers();
    int slots = 0;
    for (int r = 0; r < n_dst; ++r) {
        const int beg = ptr[r], end = ptr[r + 1], deg = end - beg;
        if (deg > cfg.tier_block) {
            const int nck = (deg + cfg.chunk_nnz - 1) / cfg.chunk_nnz;
            if (nck == 1) {
                t.chunks.insert(t.chunks.end(), {r, beg, end, -1});
            } else {
                t.split.insert(t.split.end(), {r, slots, nck, 0});
                const int per = ((deg + nck - 1) / nck + 255) & ~255;   // equal shares, whole 256-nonzero passes
                for (int c = 0; c < nck; ++c) {
                    const int cb = std::min(beg + c * per, end), ce = std::min(cb + per, end);
                    t.chunks.insert(t.chunks.end(), {r, cb, ce, slots++});   // an empty tail chunk merges as a neutral state
                }
            }
        } else if (deg > cfg.tier_wave) {
            t.rows_wave.push_back(r);
        } else {
            t.n_group++;
        }
    }
    t.n_slots = slots;
    t.short_rows = n_dst > 0 && (double)ptr[n_dst] / n_dst <= 16.0;
}

std::vector<int64_t> host_instance_cost(const int* ptr, const std::vector<int64_t>& inst_off) {
    constexpr int64_t ITEM = FUSED_COST_ITEM, STEP = FUSED_COST_STEP;
    const size_t n = inst_off.empty() ? 0 : inst_off.size() - 1;
    std::vector<int64_t> cost(n, 0);
    for (size_t k = 0; k < n; ++k) {
        int64_t c = 0;
        for (int64_t r = inst_off[k]; r < inst_off[k + 1]; ++r) {
            const int64_t deg = ptr[r + 1] - ptr[r];
            if (deg > FUSED_T16[2]) c += 12 * FUSED_COST_BLOCK_ROW;
            else if (deg > FUSED_T16[1]) c += ITEM + STEP * ((deg + 63) / 64);
            else if (deg > FUSED_T16[0]) c += (ITEM + STEP * ((deg + 15) / 16)) / 4;
            else c += (ITEM + STEP * ((deg + 3) / 4)) / 16;
        }
        cost[k] = c;
    }
    return cost;
}

std::vector<InstLoad> host_instance_loads(const int* ptr, const std::vector<int64_t>& inst_off) {
    const size_t n = inst_off.empty() ? 0 : inst_off.size() - 1;
    std::vector<InstLoad> out(n);
    for (size_t k = 0; k < n; ++k) {
        InstLoad& l = out[k];
        for (int64_t r = inst_off[k]; r < inst_off[k + 1]; ++r) {
            const int64_t deg = ptr[r + 1] - ptr[r];
            // 16-channel geometry, in 1/16 of an item / of an item's step.  A block row (walked by the 12 wavefronts of a
            // workgroup) is counted as FUSED_BLOCK_ROW_ITEMS items, not as a quantity of its own: normalised, the handful of
            // such rows -- one instance owns half of them -- outweighed everything else (partitions at 14 k and at 64 k cycles)
            if (deg > FUSED_T16[2]) { l.d[0] += 16 * FUSED_BLOCK_ROW_ITEMS; l.d[1] += 16 * 2 * ((deg + 767) / 768); l.d[4] += 1; }
            else if (deg > FUSED_T16[1]) { l.d[0] += 16; l.d[1] += 16 * ((deg + 63) / 64); }
            else if (deg > FUSED_T16[0]) { l.d[0] += 4; l.d[1] += 4 * ((deg + 15) / 16); }
            else { l.d[0] += 1; l.d[1] += std::max<int64_t>((deg + 3) / 4, 1); }
            // 1-channel geometry, in 1/64
            if (deg > FUSED_T1[2]) { l.d[2] += 64 * FUSED_BLOCK_ROW_ITEMS; l.d[3] += 64 * 2 * ((deg + 6143) / 6144); }
            else if (deg > FUSED_T1[1]) { l.d[2] += 64; l.d[3] += 64 * ((deg + 511) / 512); }
            else if (deg > FUSED_T1[0]) { l.d[2] += 4; l.d[3] += 4 * ((deg + 31) / 32); }
            else { l.d[2] += 1; l.d[3] += std::max<int64_t>((deg + 7) / 8, 1); }
            // the one-cost model of rounds 2-3 (host_instance_cost) as a quantity of its own: balanced by items and steps
            // alone, fused_fwd16 / fused_bwd16 ran 2 us slower per launch than under that model, the other three families
            // 1-2 us faster (same-box A/B, profiles/r04_fused_stamps.txt); with both, every family keeps its better figure
            if (deg > FUSED_T16[2]) l.d[5] += 12 * FUSED_COST_BLOCK_ROW;
            else if (deg > FUSED_T16[1]) l.d[5] += FUSED_COST_ITEM + FUSED_COST_STEP * ((deg + 63) / 64);
            else if (deg > FUSED_T16[0]) l.d[5] += (FUSED_COST_ITEM + FUSED_COST_STEP * ((deg + 15) / 16)) / 4;
            else l.d[5] += (FUSED_COST_ITEM + FUSED_COST_STEP * ((deg + 3) / 4)) / 16;
        }
    }
    return out;
}

namespace {
struct VecBalance {
    static constexpr int D = 2 * FUSED_LOAD_DIMS;
    int n_parts;
    std::vector<std::array<double, D>> x;       // normalised load of every instance: its share of the total, times n_parts
    std::vector<std::array<double, D>> load;    // of every partition
    double max_of(int q) const {
        double m = 0.0;
        for (int d = 0; d < D; ++d) m = std::max(m, load[(size_t)q][d]);
        return m;
    }
    double worst() const {
        double m = 0.0;
        for (int q = 0; q < n_parts; ++q) m = std::max(m, max_of(q));
        return m;
    }
};
}  // namespace

double host_partition_imbalance(const std::vector<InstLoad>& a, const std::vector<InstLoad>& b, const std::vector<int>& part,
                                int n_parts) {
    double worst = 0.0;
    for (int side = 0; side < 2; ++side)
        for (int d = 0; d < FUSED_LOAD_DIMS; ++d) {
            if (d == 4) continue;                    // (block rows: counted as items)
            std::vector<double> l((size_t)n_parts, 0.0);
            double tot = 0.0;
            for (size_t i = 0; i < a.size(); ++i) {
                const double v = (double)(side ? b[i].d[d] : a[i].d[d]);
                l[(size_t)part[i]] += v;
                tot += v;
            }
            if (tot <= 0.0) continue;
            for (int q = 0; q < n_parts; ++q) worst = std::max(worst, l[(size_t)q] * n_parts / tot);
        }
    return worst;
}

std::vector<int> host_partition_instances_v(const std::vector<InstLoad>& a, const std::vector<InstLoad>& b, int n_parts) {
    const int n = (int)a.size();
    constexpr int D = VecBalance::D;
    VecBalance vb;
    vb.n_parts = n_parts;
    vb.x.assign((size_t)n, std::array<double, D>{});
    vb.load.assign((size_t)n_parts, std::array<double, D>{});
    double tot[D] = {};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < FUSED_LOAD_DIMS; ++d) { tot[d] += (double)a[i].d[d]; tot[FUSED_LOAD_DIMS + d] += (double)b[i].d[d]; }
    // (d[4], the number of block rows, is informational: they are counted as items above; a quantity that is absent weighs
    // nothing)
    std::vector<double> size((size_t)n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < D; ++d) {
            const double v = (double)(d < FUSED_LOAD_DIMS ? a[i].d[d] : b[i].d[d - FUSED_LOAD_DIMS]);
            vb.x[(size_t)i][d] = tot[d] > 0.0 && d % FUSED_LOAD_DIMS != 4 ? v * n_parts / tot[d] : 0.0;
            size[(size_t)i] = std::max(size[(size_t)i], vb.x[(size_t)i][d]);
        }
    std::vector<int> order((size_t)n), part((size_t)n, 0);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return size[(size_t)p] > size[(size_t)q]; });
    std::vector<int> count((size_t)n_parts, 0);
    for (int i : order) {
        int best = 0;
        double best_v = 0.0;
        for (int q = 0; q < n_parts; ++q) {
            double v = 0.0;
            for (int d = 0; d < D; ++d) v = std::max(v, vb.load[(size_t)q][d] + vb.x[(size_t)i][d]);
            // (empty instances -- all quantities zero -- are spread by count)
            if (q == 0 || v < best_v - 1e-12 || (v < best_v + 1e-12 && count[(size_t)q] < count[(size_t)best])) { best = q; best_v = v; }
        }
        part[(size_t)i] = best;
        count[(size_t)best]++;
        for (int d = 0; d < D; ++d) vb.load[(size_t)best][d] += vb.x[(size_t)i][d];
    }
    // improvement: move an instance out of the worst partition, or swap it with one of another partition, while the
    // largest load over all partitions and quantities goes down (bounded: a few hundred instances at most matter)
    if (n <= 4096) {
        for (int round = 0; round < 200; ++round) {
            int wq = 0;
            for (int q = 1; q < n_parts; ++q)
                if (vb.max_of(q) > vb.max_of(wq)) wq = q;
            const double cur = vb.max_of(wq);
            double best_gain = 1e-9;
            int bi = -1, bj = -1, bq = -1;
            auto pair_max = [&](int q1, int q2, int i, int j) {      // max load of q1, q2 after i (of q1) <-> j (of q2, or -1)
                double m = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double xi = vb.x[(size_t)i][d], xj = j >= 0 ? vb.x[(size_t)j][d] : 0.0;
                    m = std::max(m, std::max(vb.load[(size_t)q1][d] - xi + xj, vb.load[(size_t)q2][d] + xi - xj));
                }
                return m;
            };
            for (int i = 0; i < n; ++i) {
                if (part[(size_t)i] != wq) continue;
                for (int q = 0; q < n_parts; ++q) {
                    if (q == wq) continue;
                    const double m0 = pair_max(wq, q, i, -1);
                    if (cur - m0 > best_gain) { best_gain = cur - m0; bi = i; bj = -1; bq = q; }
                    for (int j = 0; j < n; ++j) {
                        if (part[(size_t)j] != q) continue;
                        const double m1 = pair_max(wq, q, i, j);
                        if (cur - m1 > best_gain) { best_gain = cur - m1; bi = i; bj = j; bq = q; }
                    }
                }
            }
            if (bi < 0) break;
            for (int d = 0; d < D; ++d) {
                const double xi = vb.x[(size_t)bi][d], xj = bj >= 0 ? vb.x[(size_t)bj][d] : 0.0;
                vb.load[(size_t)wq][d] += xj - xi;
                vb.load[(size_t)bq][d] += xi - xj;
            }
            part[(size_t)bi] = bq;
            if (bj >= 0) part[(size_t)bj] = wq;
        }
    }
    // the one-cost rule's deal is a candidate too: whichever has the lower largest load over all quantities
    std::vector<int64_t> ca((size_t)n), cb((size_t)n);
    for (int i = 0; i < n; ++i) { ca[(size_t)i] = a[(size_t)i].d[5]; cb[(size_t)i] = b[(size_t)i].d[5]; }
    const std::vector<int> alt = host_partition_instances(ca, cb, n_parts);
    return host_partition_imbalance(a, b, alt, n_parts) < host_partition_imbalance(a, b, part, n_parts) - 1e-12 ? alt : part;
}

std::vector<int> host_partition_instances(const std::vector<int64_t>& cost_a, const std::vector<int64_t>& cost_b, int n_parts) {
    const int n = (int)cost_a.size();
    std::vector<int64_t> ca((size_t)n), cb((size_t)n);
    int64_t sa = 1, sb = 1;
    for (int i = 0; i < n; ++i) {
        ca[i] = cost_a[i] + 1;      // + 1: empty instances are spread too
        cb[i] = cost_b[i] + 1;
        sa += ca[i]; sb += cb[i];
    }
    std::vector<int> order((size_t)n), part((size_t)n, 0);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ca[a] + cb[a] > ca[b] + cb[b]; });
    std::vector<int64_t> la((size_t)n_parts, 0), lb((size_t)n_parts, 0);
    for (int i : order) {
        int best = 0;
        double best_v = 0.0;
        for (int q = 0; q < n_parts; ++q) {
            const double v = std::max((double)(la[q] + ca[i]) / (double)sa, (double)(lb[q] + cb[i]) / (double)sb);
            if (q == 0 || v < best_v) { best = q; best_v = v; }
        }
        part[i] = best;
        la[best] += ca[i];
        lb[best] += cb[i];
    }
    return part;
}

void host_build_fused_orient(const int* ptr, int n, const std::vector<int64_t>& inst_off, const std::vector<int>& inst_part,
                             HostFusedOrient* out) {
    HostFusedOrient& o = *out;
    o = HostFusedOrient();
    n = std::max(n, 0);
    o.perm.resize((size_t)n);
    o.inv.resize((size_t)n);
    o.sptr.assign((size_t)n + 1, 0);
    std::vector<int> node_part((size_t)n, 0);
    for (size_t k = 0; k + 1 < inst_off.size(); ++k)
        for (int64_t r = inst_off[k]; r < inst_off[k + 1] && r < n; ++r) node_part[(size_t)r] = inst_part[k];
    for (int r = 0; r < n; ++r) o.perm[r] = r;
    std::stable_sort(o.perm.begin(), o.perm.end(), [&](int a, int b) {
        if (node_part[a] != node_part[b]) return node_part[a] < node_part[b];
        return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b];
    });
    for (int q = 0; q <= FUSED_PARTS; ++q) o.row0[q] = 0;
    for (int k = 0; k < n; ++k) {
        const int r = o.perm[k], deg = ptr[r + 1] - ptr[r], q = node_part[r];
        o.inv[r] = k;
        o.sptr[k + 1] = o.sptr[k] + deg;
        o.row0[q + 1]++;
        for (int which = 0; which < 2; ++which) {
            const int(&T)[3] = which ? FUSED_T1 : FUSED_T16;
            FusedTiers& t = which ? o.t1[q] : o.t16[q];
            if (deg > T[2]) t.n_block++;
            else if (deg > T[1]) t.n_wave++;
            else if (deg > T[0]) t.n_group++;
            else t.n_base++;
        }
    }
    for (int q = 0; q < FUSED_PARTS; ++q) o.row0[q + 1] += o.row0[q];
}

void host_build_wave_lists(const HostFusedOrient& o, bool scalar, int waves_per_part, int waves_per_wg, int nnz_per_step,
                           HostWaveLists* out) {
    HostWaveLists& wl = *out;
    wl = HostWaveLists();
    const int nw = std::max(waves_per_part, 1), wpg = std::max(std::min(waves_per_wg, nw), 1), gp = std::max(nw / wpg, 1);
    wl.waves_per_part = nw;
    const int U = scalar ? 64 : 16, RG = U / 4;              // base rows / group rows per item (fused_kernels.hip::Geo)
    // nonzeros of a row per round trip: per unit (quad / lane), 4 units per group row, all units of the wavefront
    const int e_base = std::max(nnz_per_step, 1), e_group = 4 * e_base, e_wave = (scalar ? 64 : 16) * e_base;
    std::vector<std::vector<int>> lists((size_t)FUSED_PARTS * nw);
    for (int q = 0; q < FUSED_PARTS; ++q) {
        const FusedTiers& t = scalar ? o.t1[q] : o.t16[q];
        const int n_gitems = (t.n_group + RG - 1) / RG, n_bitems = (t.n_base + U - 1) / U;
        const int n_items = t.n_wave + n_gitems + n_bitems;
        auto deg_of = [&](int rl) { const int k = o.row0[q] + rl; return (int64_t)o.sptr[(size_t)k + 1] - o.sptr[(size_t)k]; };
        // cost of every item from its first (longest) row; rows of a tier are sorted by length inside the partition
        std::vector<std::pair<int64_t, int>> items((size_t)n_items);
        for (int it = 0; it < n_items; ++it) {
            int64_t steps;
            if (it < t.n_wave) steps = (deg_of(t.n_block + it) + e_wave - 1) / e_wave;
            else if (it < t.n_wave + n_gitems) steps = (deg_of(t.n_block + t.n_wave + (it - t.n_wave) * RG) + e_group - 1) / e_group;
            else steps = (deg_of(t.n_block + t.n_wave + t.n_group + (it - t.n_wave - n_gitems) * U) + e_base - 1) / e_base;
            items[(size_t)it] = {FUSED_COST_ITEM + FUSED_COST_STEP * std::max<int64_t>(steps, 1), it};
        }
        std::stable_sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        // Wavefronts of a workgroup do not run equally fast: the SIMD issues its oldest wavefront first, and the stamps
        // show wavefronts 4-7 of a workgroup taking 7 % and 8-11 taking 19 % longer than 0-3 for the same items.  Each
        // item goes to the wavefront that would FINISH it first (ties by wavefront id).
        std::vector<int64_t> load((size_t)nw, 0), slow((size_t)nw, 0);
        for (int w = 0; w < nw; ++w) {
            const int bi = w / wpg;
            const int n_rows = bi < gp && t.n_block > bi ? (t.n_block - bi + gp - 1) / gp : 0;
            slow[(size_t)w] = 1000 + 95 * ((w % wpg) / 4);          // per mille
            load[(size_t)w] = FUSED_COST_BLOCK_ROW * n_rows * slow[(size_t)w] / 1000;
        }
        for (const auto& itc : items) {
            int best = 0;
            int64_t best_t = 0;
            for (int w = 0; w < nw; ++w) {
                const int64_t tw = load[(size_t)w] + itc.first * slow[(size_t)w] / 1000;
                if (w == 0 || tw < best_t) { best = w; best_t = tw; }
            }
            lists[(size_t)q * nw + best].push_back(itc.second);
            load[(size_t)best] = best_t;
        }
        for (int w = 0; w < nw; ++w) {
            wl.max_load = std::max(wl.max_load, load[(size_t)w]);
            wl.sum_load += load[(size_t)w];
            wl.L = std::max(wl.L, (int)lists[(size_t)q * nw + w].size());
        }
    }
    wl.L = std::max(wl.L, 1);
    wl.order.assign((size_t)FUSED_PARTS * nw * wl.L, -1);
    for (size_t k = 0; k < lists.size(); ++k) std::copy(lists[k].begin(), lists[k].end(), wl.order.begin() + k * wl.L);
}

}  // namespace mllp
