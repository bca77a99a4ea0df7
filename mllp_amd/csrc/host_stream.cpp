// host_stream.cpp -- host builder of the streamed SpMM copy (layout and rationale: stream_layout.h).
// Plain C++ (also built by `make host-sanitize`).  The copy replaces, for the plain SpMM, the per-step edge-list
// build of the reference (linear_program_methods.py:89-103): built once per batch and orientation.
#include "host_stream.h"

#include <algorithm>
#include <cstring>
#include <thread>

#include "../../include/mllp_hip.h"
#include "stream_layout.h"

namespace mllp {

int fail(int code, const std::string& msg);

namespace {

// analysis of one row tile: the column blocks it touches and, per (block, row), the run of the row's entries
struct TileScan {
    std::vector<int> blocks;     // ascending global block ids
    std::vector<int> cnt;        // [nb * S_R] entries of the row in the block
    std::vector<int> start;      // [nb * S_R] CSR position of the first of them
    std::vector<int> order;      // [nb * S_R] sorted position -> row (entries descending, ties by row)
};

void scan_tile(const int* ptr, const int* idx, int64_t n_dst, int t, TileScan& s) {
    const int64_t r0 = (int64_t)t * S_R;
    const int rows = (int)std::min<int64_t>(S_R, n_dst - r0);
    s.blocks.clear();
    for (int r = 0; r < rows; ++r) {
        int last = -1;
        for (int e = ptr[r0 + r]; e < ptr[r0 + r + 1]; ++e) {
            const int b = idx[e] / S_CB;
            if (b != last) s.blocks.push_back(last = b);
        }
    }
    std::sort(s.blocks.begin(), s.blocks.end());
    s.blocks.erase(std::unique(s.blocks.begin(), s.blocks.end()), s.blocks.end());
    const size_t nb = s.blocks.size();
    s.cnt.assign(nb * S_R, 0);
    s.start.assign(nb * S_R, 0);
    s.order.resize(nb * S_R);
    for (int r = 0; r < rows; ++r) {
        size_t bi = 0;
        for (int e = ptr[r0 + r]; e < ptr[r0 + r + 1]; ++e) {
            const int b = idx[e] / S_CB;
            while (s.blocks[bi] != b) ++bi;      // the row's columns ascend, so do its blocks
            if (s.cnt[bi * S_R + r]++ == 0) s.start[bi * S_R + r] = e;
        }
    }
    for (size_t bi = 0; bi < nb; ++bi) {
        int* o = &s.order[bi * S_R];
        for (int k = 0; k < S_R; ++k) o[k] = k;
        const int* c = &s.cnt[bi * S_R];
        std::stable_sort(o, o + S_R, [c](int a, int b) { return c[a] > c[b]; });
    }
}

inline int pass_steps(const TileScan& s, size_t bi, int pair) {     // entries of the pair's longest row
    return s.cnt[bi * S_R + s.order[bi * S_R + 32 * pair]];
}

// one row of a team during the joint ordering
struct RowCur {
    int beg = 0, len = 0, rem = 0;
    int cnt[4] = {0, 0, 0, 0};
    int nxt[4] = {0, 0, 0, 0};    // scan position per class (relative to beg)
};

// Entries of the 16 rows `rows[q]` (one per quad) of a pass, written as the A (half = 0) or B (half = 1) entries of
// steps S .. S + n - 1.  The four rows of a team are ordered JOINTLY: at step p the row that chooses first rotates
// with p; a row takes its most numerous remaining class (column mod 4; lowest class on ties) that no team mate has
// taken in this step, or its most numerous class when all are taken.  Rows shorter than n keep the padding entries.
void fill_half(const int* idx, const float* val, int blk, const TileScan& s, size_t bi, const int* rows, int64_t S,
               int half, int* ent) {
    for (int tm = 0; tm < 4; ++tm) {
        RowCur rc[4];
        int maxlen = 0;
        for (int i = 0; i < 4; ++i) {
            const int r = rows[S_TEAMS[tm][i]];
            rc[i].beg = s.start[bi * S_R + r];
            rc[i].len = rc[i].rem = s.cnt[bi * S_R + r];
            for (int e = 0; e < rc[i].len; ++e) rc[i].cnt[(idx[rc[i].beg + e] - blk * S_CB) & 3]++;
            maxlen = std::max(maxlen, rc[i].len);
        }
        for (int p = 0; p < maxlen; ++p) {
            unsigned used = 0;
            for (int j = 0; j < 4; ++j) {
                const int i = (j + p) & 3;
                RowCur& c = rc[i];
                if (c.rem == 0) continue;
                int pick = -1, pick_any = -1;
                for (int k = 0; k < 4; ++k) {
                    if (c.cnt[k] == 0) continue;
                    if (pick_any < 0 || c.cnt[k] > c.cnt[pick_any]) pick_any = k;
                    if (!(used >> k & 1) && (pick < 0 || c.cnt[k] > c.cnt[pick])) pick = k;
                }
                if (pick < 0) pick = pick_any;
                int e = c.nxt[pick];
                while (((idx[c.beg + e] - blk * S_CB) & 3) != pick) ++e;
                c.nxt[pick] = e + 1;
                c.cnt[pick]--;
                c.rem--;
                used |= 1u << pick;
                const int64_t step = S + p;
                const int q = S_TEAMS[tm][i];
                int* slot = ent + (((step >> 2) * 64 + q * 4 + (step & 3)) * 4 + half * 2);
                slot[0] = (idx[c.beg + e] - blk * S_CB) * S_ROW_BYTES;
                std::memcpy(&slot[1], &val[c.beg + e], 4);
            }
        }
    }
}

template <class F>
void parallel_tiles(int n_tiles, unsigned nt, F f) {
    if (nt <= 1 || n_tiles < 4) {
        for (int t = 0; t < n_tiles; ++t) f(t, 0u);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
        th.emplace_back([=]() {
            for (int t = (int)k; t < n_tiles; t += (int)nt) f(t, k);
        });
    for (auto& x : th) x.join();
}

}  // namespace

int host_build_stream(const int* ptr, const int* idx, const float* val, int64_t n_dst, int64_t n_src, HostStream* out,
                      std::string* err, unsigned max_threads) {
    auto bad = [&](int code, const char* msg) {
        if (err) *err = msg;
        return code;
    };
    if (n_dst < 0 || n_src < 0 || (n_dst > 0 && !ptr)) return bad(MLLP_EINVAL, "host_build_stream: bad arguments");
    const int64_t n_tiles64 = (n_dst + S_R - 1) / S_R;
    if (n_tiles64 >= (1 << 28)) return bad(MLLP_ERANGE, "host_build_stream: too many row tiles");
    const int n_tiles = (int)n_tiles64;
    const unsigned nt = max_threads ? max_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    HostStream& o = *out;
    o = HostStream();
    o.n_tiles = n_tiles;
    o.tile_blk.assign(n_tiles + 1, 0);

    // pass 1: blocks per tile, steps per (tile, wavefront)
    std::vector<int64_t> steps((size_t)n_tiles * S_NW, 0);
    std::vector<TileScan> scans(nt);
    parallel_tiles(n_tiles, nt, [&](int t, unsigned k) {
        TileScan& s = scans[k];
        scan_tile(ptr, idx, n_dst, t, s);
        o.tile_blk[t + 1] = (int)s.blocks.size();
        for (size_t bi = 0; bi < s.blocks.size(); ++bi)
            for (int w = 0; w < S_NW; ++w)
                steps[(size_t)t * S_NW + w] += pass_steps(s, bi, w) + pass_steps(s, bi, S_PAIRS - 1 - w);
    });
    int64_t n_tb = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const int nb = o.tile_blk[t + 1];
        o.tile_blk[t] = (int)n_tb;
        n_tb += nb;
        if (n_tb >= (1 << 24)) return bad(MLLP_ERANGE, "host_build_stream: too many (tile, block) pairs");
    }
    if (n_tiles) o.tile_blk[n_tiles] = (int)n_tb;
    o.n_tb = (int)n_tb;
    std::vector<int64_t> base((size_t)n_tiles * S_NW + 1, 0);     // first group of every (tile, wavefront)
    for (size_t i = 0; i < steps.size(); ++i) {
        base[i + 1] = base[i] + (steps[i] + 3) / 4;
    }
    o.n_groups = base[steps.size()];
    o.step_slots = o.n_groups * 128;
    if (o.n_groups * 4 >= ((int64_t)1 << 31) - 4 * S_K) return bad(MLLP_ERANGE, "host_build_stream: more than 2^31 steps");
    o.blk_id.assign(n_tb, 0);
    o.rec.assign((size_t)n_tb * S_NW * 16 * 4, 0);
    o.ent.resize((size_t)(o.n_groups + S_K) * 64 * 4);
    for (size_t i = 0; i < o.ent.size(); i += 2) {
        o.ent[i] = S_ZERO_OFF;
        o.ent[i + 1] = 0;
    }
    o.real_slots = n_dst ? ptr[n_dst] : 0;

    // pass 2: records and entries
    parallel_tiles(n_tiles, nt, [&](int t, unsigned k) {
        TileScan& s = scans[k];
        scan_tile(ptr, idx, n_dst, t, s);
        const int tb0 = o.tile_blk[t];
        int64_t cur[S_NW];
        for (int w = 0; w < S_NW; ++w) cur[w] = base[(size_t)t * S_NW + w] * 4;
        for (size_t bi = 0; bi < s.blocks.size(); ++bi) {
            const int blk = s.blocks[bi];
            o.blk_id[tb0 + bi] = blk;
            const int* ord = &s.order[bi * S_R];
            for (int w = 0; w < S_NW; ++w) {
                const int p0 = w, p1 = S_PAIRS - 1 - w;
                const int n0 = pass_steps(s, bi, p0), n1 = pass_steps(s, bi, p1);
                const int64_t S = cur[w];
                int* rec = &o.rec[(((size_t)(tb0 + bi) * S_NW + w) * 16) * 4];
                for (int q = 0; q < 16; ++q) {
                    rec[q * 4 + 0] = ord[32 * p0 + q] | ord[32 * p0 + 16 + q] << 16;
                    rec[q * 4 + 1] = ord[32 * p1 + q] | ord[32 * p1 + 16 + q] << 16;
                    rec[q * 4 + 2] = (int)S;
                    rec[q * 4 + 3] = n0 | n1 << 16;
                }
                fill_half(idx, val, blk, s, bi, ord + 32 * p0, S, 0, o.ent.data());
                fill_half(idx, val, blk, s, bi, ord + 32 * p0 + 16, S, 1, o.ent.data());
                fill_half(idx, val, blk, s, bi, ord + 32 * p1, S + n0, 0, o.ent.data());
                fill_half(idx, val, blk, s, bi, ord + 32 * p1 + 16, S + n0, 1, o.ent.data());
                cur[w] += n0 + n1;
            }
        }
    });
    (void)n_src;
    return MLLP_OK;
}

int64_t host_walk_stream(const HostStream& s, int64_t n_dst, int64_t n_src, const float* H, double* Y) {
    int64_t real = 0;
    for (int t = 0; t < s.n_tiles; ++t) {
        for (int tb = s.tile_blk[t]; tb < s.tile_blk[t + 1]; ++tb) {
            const int64_t c0 = (int64_t)s.blk_id[tb] * S_CB;
            for (int w = 0; w < S_NW; ++w) {
                const int* rec = &s.rec[(((size_t)tb * S_NW + w) * 16) * 4];
                const int64_t S = rec[2];
                const int n0 = rec[3] & 0xffff, n1 = (int)((unsigned)rec[3] >> 16);
                for (int q = 0; q < 16; ++q) {
                    if (rec[q * 4 + 2] != rec[2] || rec[q * 4 + 3] != rec[3]) return -1;
                    for (int pass = 0; pass < 2; ++pass) {
                        const int rows = rec[q * 4 + pass];
                        const int64_t a = pass ? S + n0 : S, b = pass ? S + n0 + n1 : S + n0;
                        for (int64_t st = a; st < b; ++st) {
                            if ((st >> 2) >= s.n_groups) return -1;
                            const int* e = &s.ent[(size_t)(((st >> 2) * 64 + q * 4 + (st & 3)) * 4)];
                            for (int half = 0; half < 2; ++half) {
                                const int off = e[half * 2];
                                if (off == S_ZERO_OFF) {
                                    if (e[half * 2 + 1] != 0) return -1;
                                    continue;
                                }
                                if (off < 0 || off % S_ROW_BYTES || off >= S_ZERO_OFF) return -1;
                                const int64_t col = c0 + off / S_ROW_BYTES;
                                const int64_t row = (int64_t)t * S_R + (half ? (rows >> 16) & 0xffff : rows & 0xffff);
                                if (col >= n_src || row >= n_dst) return -1;
                                float v;
                                std::memcpy(&v, &e[half * 2 + 1], 4);
                                for (int c = 0; c < 16; ++c) Y[row * 16 + c] += (double)v * (double)H[col * 16 + c];
                                ++real;
                            }
                        }
                    }
                }
            }
        }
    }
    return real;
}

}  // namespace mllp
