// host_graph_test.cpp -- sanitizer driver for host_graph.cpp (built by `make host-sanitize` with
// -fsanitize=address,undefined and with -fsanitize=thread; run by tests/test_host_sanitize.py).
// Random ragged batches (empty rows, empty columns, empty instances, one very long row) through the parallel
// build with 8 threads; the result is checked against a serial transposition, the row tiers against the rows.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>

#include "../../include/mllp_hip.h"
#include "host_graph.h"
#include "host_stream.h"
#include "lane_layout.h"
#include "stream_layout.h"

using namespace mllp;

namespace mllp {
static std::string g_test_err;
int fail(int code, const std::string& msg) { g_test_err = msg; return code; }     // the product's is in graph.cpp
}

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                               \
        }                                                               \
    } while (0)

struct Raw {
    std::vector<int64_t> m, n, indptr;
    std::vector<int32_t> idx;
    std::vector<double> val;
};

static Raw random_batch(std::mt19937& rng, int n_inst, int long_row) {
    Raw r;
    for (int k = 0; k < n_inst; ++k) {
        const int m = (k % 7 == 3) ? 0 : 1 + (int)(rng() % 60);
        const int n = (k % 11 == 5) ? 0 : 1 + (int)(rng() % 90);
        r.m.push_back(m);
        r.n.push_back(n);
        int64_t e = 0;
        r.indptr.push_back(0);
        for (int row = 0; row < m; ++row) {
            int deg = n == 0 ? 0 : (int)(rng() % 5);
            if (rng() % 4 == 0) deg = 0;
            if (k == 1 && row == 0) deg = std::min(n, long_row);
            deg = std::min(deg, n);
            std::vector<int> cols(n);
            for (int c = 0; c < n; ++c) cols[c] = c;
            for (int c = 0; c < deg; ++c) std::swap(cols[c], cols[c + rng() % (n - c)]);
            std::sort(cols.begin(), cols.begin() + deg);
            for (int c = 0; c < deg; ++c) {
                r.idx.push_back(cols[c]);
                r.val.push_back((double)(int)(rng() % 2001 - 1000) / 1000.0);
            }
            e += deg;
            r.indptr.push_back(e);
        }
    }
    return r;
}

static void check_batch(const Raw& r, const HostBatch& b) {
    const int64_t n_inst = (int64_t)r.m.size();
    CHECK(b.n_inst == n_inst && (int64_t)b.pm.size() == n_inst + 1);
    // serial reference: dense-free transposition by (column, row) sort
    std::vector<std::vector<std::pair<int, float>>> cols(b.N);
    int64_t e = 0, pp = 0;
    for (int64_t k = 0; k < n_inst; ++k) {
        for (int64_t row = 0; row < r.m[k]; ++row) {
            const int64_t gr = b.pm[k] + row;
            CHECK(b.csr_ptr[gr] == e);
            for (int64_t q = r.indptr[pp + row]; q < r.indptr[pp + row + 1]; ++q, ++e) {
                const int64_t idx_pos = e;
                CHECK(b.csr_idx[idx_pos] == (int)(b.pn[k] + r.idx[idx_pos]));
                CHECK(b.csr_val[idx_pos] == (float)r.val[idx_pos]);
                cols[b.pn[k] + r.idx[idx_pos]].push_back({(int)gr, (float)r.val[idx_pos]});
            }
        }
        pp += r.m[k] + 1;
    }
    CHECK(e == b.nnz && b.csr_ptr[b.M] == b.nnz && b.csc_ptr[b.N] == b.nnz && b.csc_ptr[0] == 0);
    for (int64_t c = 0; c < b.N; ++c) {
        CHECK(b.csc_ptr[c + 1] - b.csc_ptr[c] == (int)cols[c].size());
        for (size_t j = 0; j < cols[c].size(); ++j) {
            CHECK(b.csc_idx[b.csc_ptr[c] + j] == cols[c][j].first);      // constraint ids ascending
            CHECK(b.csc_val[b.csc_ptr[c] + j] == cols[c][j].second);
        }
    }
}

static void check_tiers(const int* ptr, int n_dst, const TierConfig& c) {
    HostTiers t;
    host_build_tiers(ptr, n_dst, c, &t);
    std::vector<int> covered(n_dst, 0);
    std::vector<int> nnz_in_chunks(n_dst, 0);
    for (int r : t.rows_wave) {
        CHECK(r >= 0 && r < n_dst);
        const int deg = ptr[r + 1] - ptr[r];
        CHECK(deg > c.tier_wave && deg <= c.tier_block);
        covered[r]++;
    }
    CHECK(t.chunks.size() % 4 == 0 && t.split.size() % 4 == 0);
    int slots_seen = 0;
    for (size_t i = 0; i < t.chunks.size(); i += 4) {
        const int r = t.chunks[i], beg = t.chunks[i + 1], end = t.chunks[i + 2], slot = t.chunks[i + 3];
        CHECK(r >= 0 && r < n_dst && beg >= ptr[r] && end <= ptr[r + 1] && beg <= end);
        CHECK(end - beg <= c.chunk_nnz + 256);
        nnz_in_chunks[r] += end - beg;
        if (slot < 0) covered[r]++;
        else { CHECK(slot == slots_seen); slots_seen++; }
    }
    CHECK(slots_seen == t.n_slots);
    int slot0 = 0;
    for (size_t i = 0; i < t.split.size(); i += 4) {
        const int r = t.split[i];
        CHECK(t.split[i + 1] == slot0 && t.split[i + 2] >= 2);
        slot0 += t.split[i + 2];
        covered[r]++;
    }
    CHECK(slot0 == t.n_slots);
    int n_group = 0;
    for (int r = 0; r < n_dst; ++r) {
        const int deg = ptr[r + 1] - ptr[r];
        if (deg <= c.tier_wave) { CHECK(covered[r] == 0); n_group++; }
        else CHECK(covered[r] == 1);
        if (deg > c.tier_block) CHECK(nnz_in_chunks[r] == deg);
    }
    CHECK(n_group == t.n_group);
}

static void check_fused(const int* ptr, int n, const std::vector<int64_t>& off, const std::vector<int>& part) {
    HostFusedOrient o;
    host_build_fused_orient(ptr, n, off, part, &o);
    CHECK((int)o.perm.size() == n && (int)o.inv.size() == n && (int)o.sptr.size() == n + 1);
    CHECK(o.row0[0] == 0 && o.row0[FUSED_PARTS] == n);
    std::vector<int> node_part(n, -1);
    for (size_t k = 0; k + 1 < off.size(); ++k)
        for (int64_t r = off[k]; r < off[k + 1]; ++r) node_part[r] = part[k];
    for (int q = 0; q < FUSED_PARTS; ++q) {
        int prev = 1 << 30, prev_row = -1;
        CHECK(o.row0[q] <= o.row0[q + 1]);
        for (int k = o.row0[q]; k < o.row0[q + 1]; ++k) {
            const int r = o.perm[k];
            CHECK(r >= 0 && r < n && o.inv[r] == k && node_part[r] == q);
            const int deg = ptr[r + 1] - ptr[r];
            CHECK(deg <= prev);
            if (deg == prev) CHECK(r > prev_row);
            CHECK(o.sptr[k + 1] - o.sptr[k] == deg);
            prev = deg; prev_row = r;
        }
        for (int which = 0; which < 2; ++which) {
            const FusedTiers& t = which ? o.t1[q] : o.t16[q];
            const int (&T)[3] = which ? FUSED_T1 : FUSED_T16;
            CHECK(t.n_block + t.n_wave + t.n_group + t.n_base == o.row0[q + 1] - o.row0[q]);
            for (int k = o.row0[q]; k < o.row0[q + 1]; ++k) {
                const int deg = o.sptr[k + 1] - o.sptr[k], kk = k - o.row0[q];
                if (kk < t.n_block) CHECK(deg > T[2]);
                else if (kk < t.n_block + t.n_wave) CHECK(deg > T[1] && deg <= T[2]);
                else if (kk < t.n_block + t.n_wave + t.n_group) CHECK(deg > T[0] && deg <= T[1]);
                else CHECK(deg <= T[0]);
            }
        }
    }
    CHECK(o.sptr[n] == ptr[n]);
    // wave lists: every item of a partition exactly once, same lists when built again, padded with -1 behind the last
    for (int scalar = 0; scalar < 2; ++scalar)
        for (int nw : {1, 12, 36, 384}) {
            HostWaveLists a, b;
            host_build_wave_lists(o, scalar != 0, nw, 12, scalar ? 8 : (nw & 1 ? 2 : 4), &a);
            host_build_wave_lists(o, scalar != 0, nw, 12, scalar ? 8 : (nw & 1 ? 2 : 4), &b);
            CHECK(a.order == b.order && a.L == b.L && a.L >= 1 && a.waves_per_part == nw);
            CHECK(a.order.size() == (size_t)FUSED_PARTS * nw * a.L);
            CHECK(a.max_load * (int64_t)FUSED_PARTS * nw >= a.sum_load);
            for (int q = 0; q < FUSED_PARTS; ++q) {
                const FusedTiers& t = scalar ? o.t1[q] : o.t16[q];
                const int n_items = t.n_wave + (t.n_group + (scalar ? 15 : 3)) / (scalar ? 16 : 4) +
                                    (t.n_base + (scalar ? 63 : 15)) / (scalar ? 64 : 16);
                std::vector<int> seen(n_items, 0);
                size_t longest = 0;
                for (int w = 0; w < nw; ++w) {
                    bool ended = false;
                    size_t len = 0;
                    for (int k = 0; k < a.L; ++k) {
                        const int it = a.order[((size_t)q * nw + w) * a.L + k];
                        if (it < 0) { ended = true; continue; }
                        CHECK(!ended && it < n_items);
                        seen[it]++;
                        len++;
                    }
                    longest = std::max(longest, len);
                }
                for (int it = 0; it < n_items; ++it) CHECK(seen[it] == 1);
                CHECK(longest <= (size_t)a.L);
            }
        }
}

// MPS reader under the sanitizers: the committed fixtures (argv) in both stages, and a synthetic file with RANGES on
// L / G / E rows, a G row, negative right-hand sides, a blank RHS set name, an RHS on the objective and a MARKER line
static void check_mps(const char* path, bool synthetic) {
    for (int normalize = 0; normalize < 2; ++normalize) {
        mllp_lp_t* lp = nullptr;
        CHECK(mllp_mps_read(path, normalize, &lp) == MLLP_OK);
        int64_t d[6];
        CHECK(mllp_lp_dims(lp, d) == MLLP_OK);
        CHECK(d[0] > 0 && d[1] == d[3] + d[4] + d[5] && (normalize || d[5] == 0));
        std::vector<int64_t> ip(d[0] + 1);
        std::vector<int32_t> ix(d[2]), sr(d[5]);
        std::vector<double> va(d[2]), co(d[1]), rh(d[0]);
        CHECK(mllp_lp_export(lp, ip.data(), ix.data(), va.data(), co.data(), rh.data(), sr.data()) == MLLP_OK);
        CHECK(ip[0] == 0 && ip[d[0]] == d[2]);
        for (int64_t i = 0; i < d[0]; ++i) {
            double ss = 0.0;
            for (int64_t e = ip[i]; e < ip[i + 1]; ++e) {
                CHECK(ix[e] >= 0 && ix[e] < d[1] && (e == ip[i] || ix[e] > ix[e - 1]) && va[e] != 0.0);
                ss += va[e] * va[e];
            }
            if (normalize && ip[i + 1] > ip[i]) {
                CHECK(std::fabs(rh[i]) <= 5.0 + 1e-12);
                CHECK(std::fabs(std::sqrt(ss) - 1.0) < 1e-12 || std::fabs(rh[i] - 5.0) < 1e-12);   // unit row or capped rhs
            }
        }
        if (synthetic) {
            CHECK(d[0] == 5 && d[3] == 3 && d[4] == 3);
            if (!normalize) { CHECK(rh[0] == -4.0 && rh[1] == 7.0 && rh[3] == 12.0); }
            else { CHECK(d[5] == 2 && sr[0] == 3 && sr[1] == 4); }
        }
        CHECK(mllp_lp_free(lp) == MLLP_OK);
    }
}


static void random_csr(std::mt19937& rng, int n_dst, int n_src, double mean_deg, int long_row, std::vector<int>& ptr,
                       std::vector<int>& idx, std::vector<float>& val) {
    ptr.assign(1, 0);
    for (int r = 0; r < n_dst; ++r) {
        int deg = (int)(mean_deg * 2.0 * (rng() % 1000) / 1000.0);
        if (rng() % 5 == 0) deg = 0;
        if (r == 7) deg = long_row;
        deg = std::min(deg, n_src);
        std::vector<int> cols;
        if (deg * 3 > n_src) {
            std::vector<int> all(n_src);
            for (int c = 0; c < n_src; ++c) all[c] = c;
            for (int c = 0; c < deg; ++c) std::swap(all[c], all[c + rng() % (n_src - c)]);
            cols.assign(all.begin(), all.begin() + deg);
        } else {
            while ((int)cols.size() < deg) {
                const int c = (int)(rng() % n_src);
                if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
            }
        }
        std::sort(cols.begin(), cols.end());
        for (int c : cols) {
            idx.push_back(c);
            val.push_back((float)(int)(rng() % 2001 - 1000) / 1000.0f + 0.0005f);
        }
        ptr.push_back((int)idx.size());
    }
}

// Lane-per-row copy of the layer-1 sweeps (host_stream.cpp::host_build_lane, layout lane_layout.h): walked the way the
// lanes walk it, it must reproduce A x, visit every nonzero exactly once, keep the rows of a tile in descending length and
// its tiles inside the segments; the attention geometries of the streamed layout are walked the same way as geometry 0.
static void check_lane(std::mt19937& rng, int n_dst, int n_src, double mean_deg, int long_row, int n_seg = 1) {
    std::vector<int> ptr, idx;
    std::vector<float> val;
    random_csr(rng, n_dst, n_src, mean_deg, long_row, ptr, idx, val);
    std::vector<int64_t> seg(1, 0);
    for (int k = 1; k < n_seg; ++k) seg.push_back((int64_t)n_dst * k / n_seg + (k & 1));
    seg.push_back(n_dst);
    HostLane a;
    std::string err;
    CHECK(host_build_lane(ptr.data(), idx.data(), val.data(), n_dst, n_seg > 1 ? seg.data() : nullptr, n_seg, &a, &err) == MLLP_OK);
    CHECK((int)a.tile_row.size() == a.n_tiles + 1 && a.tile_row[0] == 0 && a.tile_row[a.n_tiles] == n_dst);
    CHECK(a.real_slots == (int64_t)idx.size() && a.tile_blk[a.n_tiles] == a.n_tb);
    CHECK(a.offs.size() == (size_t)(a.n_groups + L1_PADG) * 128 && a.vals.size() == (size_t)(a.n_groups + L1_PADG) * 256);
    for (int t = 0; t < a.n_tiles; ++t) {
        const int nr = a.tile_row[t + 1] - a.tile_row[t];
        CHECK(nr > 0 && nr <= L1_R);
        for (int64_t sgm : seg) CHECK(!(sgm > a.tile_row[t] && sgm < a.tile_row[t + 1]));
        std::vector<int> seen(nr, 0);
        int last = INT32_MAX;
        for (int p = 0; p < L1_R; ++p) {
            const int local = a.rows[(size_t)t * L1_R + p];
            CHECK(p < nr ? (local >= 0 && local < nr) : local == -1);
            if (local < 0) continue;
            seen[local]++;
            const int len = ptr[a.tile_row[t] + local + 1] - ptr[a.tile_row[t] + local];
            CHECK(len <= last);
            last = len;
        }
        for (int v : seen) CHECK(v == 1);
        const int nb = a.tile_blk[t + 1] - a.tile_blk[t];
        CHECK(nb == 0 ? a.tile_col[2 * t + 1] == -1 : (a.tile_col[2 * t] % 4 == 0 && (a.tile_col[2 * t + 1] - a.tile_col[2 * t]) / L1_CB + 1 == nb));
    }
    std::vector<float> x(n_src);
    for (auto& v : x) v = (float)(int)(rng() % 2001 - 1000) / 500.0f;
    std::vector<double> y(n_dst, 0.0), yref(n_dst, 0.0);
    CHECK(host_walk_lane(a, n_dst, n_src, x.data(), y.data()) == (int64_t)idx.size());
    for (int r = 0; r < n_dst; ++r)
        for (int e = ptr[r]; e < ptr[r + 1]; ++e) yref[r] += (double)val[e] * (double)x[idx[e]];
    for (int r = 0; r < n_dst; ++r) CHECK(std::fabs(y[r] - yref[r]) <= 1e-9 * (1.0 + std::fabs(yref[r])));
    // the attention geometries of the streamed layout (stream_layout.h ids 1-3) on the same matrix
    for (int geom = 1; geom <= 3; ++geom) {
        HostStream s;
        CHECK(host_build_stream(ptr.data(), idx.data(), val.data(), n_dst, n_src, n_seg > 1 ? seg.data() : nullptr, n_seg, &s, &err, 4, geom) == MLLP_OK);
        std::vector<float> H((size_t)n_src * 16);
        for (auto& v : H) v = (float)(int)(rng() % 2001 - 1000) / 500.0f;
        std::vector<double> Y((size_t)n_dst * 16, 0.0), Yref((size_t)n_dst * 16, 0.0);
        CHECK(host_walk_stream(s, n_dst, n_src, H.data(), Y.data(), geom) == (int64_t)idx.size());
        for (int r = 0; r < n_dst; ++r)
            for (int e = ptr[r]; e < ptr[r + 1]; ++e)
                for (int k = 0; k < 16; ++k) Yref[(size_t)r * 16 + k] += (double)val[e] * (double)H[(size_t)idx[e] * 16 + k];
        for (size_t i = 0; i < Y.size(); ++i) CHECK(std::fabs(Y[i] - Yref[i]) <= 1e-9 * (1.0 + std::fabs(Yref[i])));
    }
}

// Streamed SpMM copy (host_stream.cpp): random matrices that span several row tiles and column blocks (empty rows, one
// row denser than a block's window, a ragged last tile / block); the copy is walked the way the kernel's wavefronts
// walk it and must reproduce the CSR product, visit every nonzero exactly once, and be identical when built again
// with another thread count.
static void check_stream(std::mt19937& rng, int n_dst, int n_src, double mean_deg, int long_row, int n_seg = 1) {
    std::vector<int> ptr(1, 0), idx;
    std::vector<float> val;
    for (int r = 0; r < n_dst; ++r) {
        int deg = (int)(mean_deg * 2.0 * (rng() % 1000) / 1000.0);
        if (rng() % 5 == 0) deg = 0;
        if (r == 7) deg = long_row;
        deg = std::min(deg, n_src);
        std::vector<int> cols;
        if (deg * 3 > n_src) {
            std::vector<int> all(n_src);
            for (int c = 0; c < n_src; ++c) all[c] = c;
            for (int c = 0; c < deg; ++c) std::swap(all[c], all[c + rng() % (n_src - c)]);
            cols.assign(all.begin(), all.begin() + deg);
        } else {
            while ((int)cols.size() < deg) {
                const int c = (int)(rng() % n_src);
                if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
            }
        }
        std::sort(cols.begin(), cols.end());
        for (int c : cols) {
            idx.push_back(c);
            val.push_back((float)(int)(rng() % 2001 - 1000) / 1000.0f + 0.0005f);
        }
        ptr.push_back((int)idx.size());
    }
    std::vector<float> H((size_t)n_src * 16);
    for (auto& h : H) h = (float)(int)(rng() % 2001 - 1000) / 500.0f;
    HostStream a, b;
    std::string err;
    std::vector<int64_t> seg(1, 0);
    for (int k = 1; k < n_seg; ++k) seg.push_back((int64_t)n_dst * k / n_seg + (k & 1));
    seg.push_back(n_dst);
    const int64_t* sp = n_seg > 1 ? seg.data() : nullptr;
    CHECK(host_build_stream(ptr.data(), idx.data(), val.data(), n_dst, n_src, sp, n_seg, &a, &err, 8) == MLLP_OK);
    CHECK(host_build_stream(ptr.data(), idx.data(), val.data(), n_dst, n_src, sp, n_seg, &b, &err, 1) == MLLP_OK);
    CHECK(a.tile_blk == b.tile_blk && a.blk_id == b.blk_id && a.rows == b.rows && a.hdr == b.hdr && a.ent == b.ent &&
          a.n_groups == b.n_groups && a.tile_row == b.tile_row);
    CHECK((int)a.tile_row.size() == a.n_tiles + 1 && a.tile_row[0] == 0 && a.tile_row[a.n_tiles] == n_dst);
    for (int t = 0; t < a.n_tiles; ++t) {
        CHECK(a.tile_row[t + 1] > a.tile_row[t] && a.tile_row[t + 1] - a.tile_row[t] <= S_RR);
        for (int64_t sgm : seg) CHECK(!(sgm > a.tile_row[t] && sgm < a.tile_row[t + 1]));      // no tile crosses a segment
    }
    CHECK((int)a.tile_blk.size() == a.n_tiles + 1 && a.tile_blk[a.n_tiles] == a.n_tb);
    CHECK(a.ent.size() == (size_t)(a.n_groups + S_K0) * 64 * S_ENT && a.real_slots == (int64_t)idx.size());
    CHECK(a.step_slots >= a.real_slots);
    std::vector<double> Y((size_t)n_dst * 16, 0.0), Yref((size_t)n_dst * 16, 0.0);
    CHECK(host_walk_stream(a, n_dst, n_src, H.data(), Y.data()) == (int64_t)idx.size());
    for (int r = 0; r < n_dst; ++r)
        for (int e = ptr[r]; e < ptr[r + 1]; ++e)
            for (int c = 0; c < 16; ++c) Yref[(size_t)r * 16 + c] += (double)val[e] * (double)H[(size_t)idx[e] * 16 + c];
    for (size_t i = 0; i < Y.size(); ++i) CHECK(std::fabs(Y[i] - Yref[i]) <= 1e-9 * (1.0 + std::fabs(Yref[i])));
    // records: every row of a tile appears exactly once per (tile, block); steps of a wavefront are contiguous
    for (int t = 0; t < a.n_tiles; ++t) {
        std::vector<int64_t> next(S_NW, -1);
        for (int tb = a.tile_blk[t]; tb < a.tile_blk[t + 1]; ++tb) {
            CHECK(tb == a.tile_blk[t] || a.blk_id[tb] > a.blk_id[tb - 1]);
            std::vector<int> seen(S_R, 0);
            for (int w = 0; w < S_NW; ++w) {
                const int* hdr = &a.hdr[((size_t)tb * S_NW + w) * 4];
                const int* rows = &a.rows[(((size_t)tb * S_NW + w) * 16) * 4];
                if (next[w] >= 0) CHECK(hdr[0] == next[w]);
                else CHECK(hdr[0] % S_GS == 0);
                CHECK(hdr[2] == a.blk_id[tb] && hdr[3] == 0);
                next[w] = (int64_t)hdr[0] + (hdr[1] & 0xffff) + (int)((unsigned)hdr[1] >> 16);
                for (int q = 0; q < 16; ++q)
                    for (int j = 0; j < S_P; ++j)
                        for (int r = 0; r < S_RQ; ++r) seen[(rows[q * 4 + 2 * j + (r >> 1)] >> (16 * (r & 1))) & 0xffff]++;
            }
            for (int r = 0; r < S_R; ++r) CHECK(seen[r] == 1);
        }
    }
}

int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) check_mps(argv[a], false);
    {
        const char* path = "/tmp/mllp_sanitize_synth.mps";
        FILE* fh = std::fopen(path, "w");
        CHECK(fh != nullptr);
        std::fputs("NAME          SYNTH\nROWS\n N  COST\n L  R1\n G  R2\n E  R3\n G  R4\n L  R5\n N  FREE\nCOLUMNS\n"
                   "    MARKER                 'MARKER'                 'INTORG'\n"
                   "    X1        COST         1.0   R1           2.0\n    X1        R2          -1.5   R4           1.0\n"
                   "    X2        COST        -2.0   R3           4.0\n    X2        R5           0.5\n"
                   "    X3        R1           1.0D0 R4          -3.0\n    X3        R5           8.0\n"
                   "RHS\n    RHS       R1          -4.0   R2           7.0\n              R4          12.0   COST        -9.0\n"
                   "RANGES\n    RNG       R1           2.0   R2           3.0\n    RNG       R3          -1.0\n"
                   "BOUNDS\n UP BND       X1           4.0\n FR BND       X2\nENDATA\n", fh);
        std::fclose(fh);
        check_mps(path, true);
        mllp_lp_t* lp = nullptr;
        CHECK(mllp_mps_read("/nonexistent/file.mps", 1, &lp) == MLLP_EINVAL && lp == nullptr);
        CHECK(mllp_mps_read(nullptr, 1, &lp) == MLLP_EINVAL);
    }
    std::mt19937 rng(12345);
    for (int round = 0; round < 6; ++round) {
        const int n_inst = round == 0 ? 1 : 5 + 9 * round;
        Raw r = random_batch(rng, n_inst, round == 3 ? 90 : 70);
        HostBatch b;
        std::string err;
        const int rc = host_build_batch(n_inst, r.m.data(), r.n.data(), r.indptr.data(), r.idx.data(), r.val.data(),
                                        &b, &err, 8);
        CHECK(rc == MLLP_OK);
        check_batch(r, b);
        for (int tw : {1, 4, 64})
            for (int tb : {2, 16, 256}) {
                TierConfig c = host_choose_tiers(b.nnz, tw, tb);
                CHECK(c.tier_block >= c.tier_wave && c.chunk_nnz == 4 * c.tier_block);
                check_tiers(b.csr_ptr.data(), (int)b.M, c);
                check_tiers(b.csc_ptr.data(), (int)b.N, c);
            }
        {
            const std::vector<int64_t> ka = host_instance_cost(b.csr_ptr.data(), b.pm), kb = host_instance_cost(b.csc_ptr.data(), b.pn);
            CHECK((int)ka.size() == n_inst && (int)kb.size() == n_inst);
            const std::vector<int> part = host_partition_instances(ka, kb, FUSED_PARTS);
            CHECK(part == host_partition_instances(ka, kb, FUSED_PARTS));       // deterministic
            std::vector<int64_t> la(FUSED_PARTS, 0), lb(FUSED_PARTS, 0);
            int64_t ma = 0, mb = 0, sa = 0, sb = 0;
            for (int k = 0; k < n_inst; ++k) {
                CHECK(part[k] >= 0 && part[k] < FUSED_PARTS && ka[k] >= 0 && kb[k] >= 0);
                const int64_t ca = ka[k] + 1, cb = kb[k] + 1;
                la[part[k]] += ca; lb[part[k]] += cb; ma = std::max(ma, ca); mb = std::max(mb, cb); sa += ca; sb += cb;
            }
            for (int q = 0; q < FUSED_PARTS; ++q)      // greedy bound on both sides (loose: two criteria)
                CHECK(la[q] <= 2 * (sa / FUSED_PARTS + ma) && lb[q] <= 2 * (sb / FUSED_PARTS + mb));
            check_fused(b.csr_ptr.data(), (int)b.M, b.pm, part);
            check_fused(b.csc_ptr.data(), (int)b.N, b.pn, part);
            // round 4: the partition that the library uses balances items, steps and block rows of both orientations
            const std::vector<InstLoad> va = host_instance_loads(b.csr_ptr.data(), b.pm), vc = host_instance_loads(b.csc_ptr.data(), b.pn);
            CHECK((int)va.size() == n_inst && (int)vc.size() == n_inst);
            const std::vector<int> pv = host_partition_instances_v(va, vc, FUSED_PARTS);
            CHECK(pv == host_partition_instances_v(va, vc, FUSED_PARTS));
            for (int k = 0; k < n_inst; ++k) CHECK(pv[k] >= 0 && pv[k] < FUSED_PARTS);
            // never worse than the scalar rule of rounds 2-3 on the quantities it balances
            CHECK(host_partition_imbalance(va, vc, pv, FUSED_PARTS) <= host_partition_imbalance(va, vc, part, FUSED_PARTS) + 1e-9);
            check_fused(b.csr_ptr.data(), (int)b.M, b.pm, pv);
            check_fused(b.csc_ptr.data(), (int)b.N, b.pn, pv);
        }
    }
    check_lane(rng, 1300, 900, 14.0, 700);         // 3 tiles (the last ragged), one block, a 700-entry row
    check_lane(rng, 700, 45011, 30.0, 5000);       // columns over 3 blocks of 20 000
    check_lane(rng, L1_R, 40, 3.0, 40);            // exactly one tile
    check_lane(rng, 3000, 1500, 9.0, 30, 4);       // four segments: ragged tiles at every segment end
    check_lane(rng, 5, 7, 0.0, 0);                 // no entries at all
    check_stream(rng, 2600, 2500, 12.0, 1400);     // 3 row tiles (the last ragged) x 4 column blocks, a 1400-entry row
    check_stream(rng, S_RR, S_CB, 3.0, S_CB);      // exactly one tile, one block, one full row
    check_stream(rng, 40, 60, 2.0, 5);
    check_stream(rng, 1200, 4100, 40.0, 0);
    check_stream(rng, 3000, 1500, 9.0, 30, 4);     // four segments: ragged tiles at every segment end
    {   // no rows at all
        HostStream e;
        std::string err;
        const int p0[1] = {0};
        CHECK(host_build_stream(p0, nullptr, nullptr, 0, 10, nullptr, 0, &e, &err) == MLLP_OK && e.n_tiles == 0 && e.n_groups == 0);
        CHECK(e.ent.size() == (size_t)S_K0 * 64 * S_ENT);
    }
    {   // empty batch and argument errors
        HostBatch b;
        std::string err;
        CHECK(host_build_batch(0, nullptr, nullptr, nullptr, nullptr, nullptr, &b, &err) == MLLP_OK && b.nnz == 0);
        const int64_t m1[1] = {2}, n1[1] = {3};
        const int64_t ip_ok[3] = {0, 2, 3};
        const int32_t ix_unsorted[3] = {2, 1, 0}, ix_range[3] = {0, 3, 1}, ix_ok[3] = {0, 2, 1};
        const double v[3] = {1.0, 2.0, 3.0};
        CHECK(host_build_batch(1, m1, n1, ip_ok, ix_unsorted, v, &b, &err, 2) == MLLP_EINVAL);
        CHECK(host_build_batch(1, m1, n1, ip_ok, ix_range, v, &b, &err, 2) == MLLP_EINVAL);
        CHECK(host_build_batch(1, m1, n1, ip_ok, ix_ok, v, &b, &err, 2) == MLLP_OK);
        const int64_t ip_bad[3] = {0, 2, 1}, ip_off[3] = {1, 2, 3};
        CHECK(host_build_batch(1, m1, n1, ip_bad, ix_ok, v, &b, &err, 2) == MLLP_EINVAL);
        CHECK(host_build_batch(1, m1, n1, ip_off, ix_ok, v, &b, &err, 2) == MLLP_EINVAL);
        const int64_t neg[1] = {-1};
        CHECK(host_build_batch(1, neg, n1, ip_ok, ix_ok, v, &b, &err, 2) == MLLP_EINVAL);
        CHECK(host_choose_tiers(100, 0, 0).tier_wave == 64 && host_choose_tiers((int64_t)40 << 20, 0, 0).tier_wave == 1024);
    }
    std::puts("host_graph sanitizer driver: ok");
    return 0;
}
