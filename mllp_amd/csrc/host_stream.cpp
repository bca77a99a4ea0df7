// host_stream.cpp -- host builder of the streamed SpMM copy (layout and rationale: stream_layout.h).
// Plain C++ (also built by `make host-sanitize`).  The copy replaces, for the plain SpMM, the per-step edge-list
// build of the reference (linear_program_methods.py:89-103): built once per batch and orientation.
#include "host_stream.h"

#include <algorithm>
#include <cstring>
#include <thread>

#include "../../include/mllp_hip.h"
#include "lane_layout.h"
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

template <class G>
void scan_tile(const int* ptr, const int* idx, const std::vector<int>& tile_row, int t, TileScan& s) {
    constexpr int S_R = G::R, S_CB = G::CB;
    const int64_t r0 = tile_row[t];
    const int rows = tile_row[t + 1] - tile_row[t];
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

template <class G>
inline int pass_steps(const TileScan& s, size_t bi, int bundle) {     // entries of the bundle's longest row
    return s.cnt[bi * G::R + s.order[bi * G::R + G::BR * bundle]];
}

// one row of a team during the joint ordering
struct RowCur {
    int beg = 0, len = 0, rem = 0;
    int cnt[4] = {0, 0, 0, 0};
    int nxt[4] = {0, 0, 0, 0};    // scan position per class (relative to beg)
};

// Entries of the 16 rows `rows[q]` (one per quad) of a pass, written as the entries of row slot `slot` of steps
// S .. S + n - 1.  The four rows of a team are ordered JOINTLY: at step p the row that chooses first rotates
// with p; a row takes its most numerous remaining class (column mod 4; lowest class on ties) that no team mate has
// taken in this step, or its most numerous class when all are taken.  Rows shorter than n keep the padding entries.
template <class G>
void fill_slot(const int* idx, const float* val, int blk, const TileScan& s, size_t bi, const int* rows, int64_t S,
               int slot, int* ent) {
    constexpr int S_R = G::R, S_CB = G::CB, S_ROW_BYTES = G::ITEM;
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
                int* dst = ent + G::ent_index(step, q, slot);
                const unsigned o16 = (unsigned)(idx[c.beg + e] - blk * S_CB) * S_ROW_BYTES;
                dst[0] = (step & 1) ? (int)(((unsigned)dst[0] & 0xffffu) | o16 << 16) : (int)(((unsigned)dst[0] & 0xffff0000u) | o16);
                std::memcpy(&dst[1 + (step & 1)], &val[c.beg + e], 4);
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

template <class G>
static std::vector<int> stream_tiles_t(const int64_t* seg_ptr, int64_t n_seg, int64_t n_dst) {
    constexpr int S_RR = G::RR;
    std::vector<int> tr(1, 0);
    const int64_t one[2] = {0, n_dst};
    if (!seg_ptr) { seg_ptr = one; n_seg = 1; }
    for (int64_t k = 0; k < n_seg; ++k) {
        // equal tiles inside a segment (10 000 rows = 11 x 909-910, not 10 x 960 + 400): workgroups of equal duration
        const int64_t len = seg_ptr[k + 1] - seg_ptr[k], nt = (len + S_RR - 1) / S_RR;
        for (int64_t i = 1; i <= nt; ++i) tr.push_back((int)(seg_ptr[k] + len * i / nt));
    }
    return tr;
}

template <class G>
static int build_stream_t(const int* ptr, const int* idx, const float* val, int64_t n_dst, int64_t n_src,
                          const int64_t* seg_ptr, int64_t n_seg, HostStream* out, std::string* err, unsigned max_threads) {
    constexpr int S_R = G::R, S_NW = G::NW, S_P = G::P, S_GS = G::GS, S_K0 = G::K0, S_ENT = G::ENT, S_PAD_WORD = G::PAD_WORD,
                  S_RQ = G::RQ, S_BR = G::BR;
    auto bad = [&](int code, const char* msg) {
        if (err) *err = msg;
        return code;
    };
    if (n_dst < 0 || n_src < 0 || (n_dst > 0 && !ptr)) return bad(MLLP_EINVAL, "host_build_stream: bad arguments");
    if (seg_ptr && (n_seg < 1 || seg_ptr[0] != 0 || seg_ptr[n_seg] != n_dst))
        return bad(MLLP_EINVAL, "host_build_stream: segments must cover the rows");
    if (n_dst >= ((int64_t)1 << 31) - 1) return bad(MLLP_ERANGE, "host_build_stream: too many rows");
    HostStream& o = *out;
    o = HostStream();
    o.tile_row = stream_tiles_t<G>(seg_ptr, n_seg, n_dst);
    const int n_tiles = (int)o.tile_row.size() - 1;
    const unsigned nt = max_threads ? max_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    o.n_tiles = n_tiles;
    o.tile_blk.assign(n_tiles + 1, 0);

    // pass 1: blocks per tile, steps per (tile, wavefront)
    std::vector<int64_t> steps((size_t)n_tiles * S_NW, 0);
    std::vector<TileScan> scans(nt);
    parallel_tiles(n_tiles, nt, [&](int t, unsigned k) {
        TileScan& s = scans[k];
        scan_tile<G>(ptr, idx, o.tile_row, t, s);
        o.tile_blk[t + 1] = (int)s.blocks.size();
        for (size_t bi = 0; bi < s.blocks.size(); ++bi)
            for (int w = 0; w < S_NW; ++w)
                for (int j = 0; j < S_P; ++j) steps[(size_t)t * S_NW + w] += pass_steps<G>(s, bi, G::bundle(w, j));
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
        base[i + 1] = base[i] + (steps[i] + S_GS - 1) / S_GS;
    }
    o.n_groups = base[steps.size()];
    o.step_slots = o.n_groups * 128;     // 64 lanes x 2 entries per group
    if (o.n_groups * S_GS >= ((int64_t)1 << 31) - S_GS * S_K0) return bad(MLLP_ERANGE, "host_build_stream: more than 2^31 steps");
    o.blk_id.assign(n_tb, 0);
    o.rows.assign((size_t)n_tb * S_NW * 16 * 4, 0);
    o.hdr.assign((size_t)n_tb * S_NW * 4, 0);
    o.ent.resize((size_t)(o.n_groups + S_K0) * 64 * S_ENT);
    for (size_t i = 0; i < o.ent.size(); i += S_ENT) {
        o.ent[i] = S_PAD_WORD;
        o.ent[i + 1] = o.ent[i + 2] = 0;
    }
    o.real_slots = n_dst ? ptr[n_dst] : 0;

    // pass 2: records and entries
    parallel_tiles(n_tiles, nt, [&](int t, unsigned k) {
        TileScan& s = scans[k];
        scan_tile<G>(ptr, idx, o.tile_row, t, s);
        const int tb0 = o.tile_blk[t];
        int64_t cur[S_NW];
        for (int w = 0; w < S_NW; ++w) cur[w] = base[(size_t)t * S_NW + w] * S_GS;
        for (size_t bi = 0; bi < s.blocks.size(); ++bi) {
            const int blk = s.blocks[bi];
            o.blk_id[tb0 + bi] = blk;
            const int* ord = &s.order[bi * S_R];
            for (int w = 0; w < S_NW; ++w) {
                int n[2] = {0, 0};
                int* rows = &o.rows[(((size_t)(tb0 + bi) * S_NW + w) * 16) * 4];
                int64_t S = cur[w];
                for (int j = 0; j < S_P; ++j) {
                    const int p = G::bundle(w, j);
                    n[j] = pass_steps<G>(s, bi, p);
                    for (int q = 0; q < 16; ++q)
                        for (int r = 0; r < S_RQ; ++r)
                            rows[q * 4 + 2 * j + (r >> 1)] |= ord[S_BR * p + 16 * r + q] << (16 * (r & 1));
                    for (int slot = 0; slot < S_RQ; ++slot)
                        fill_slot<G>(idx, val, blk, s, bi, ord + S_BR * p + 16 * slot, S, slot, o.ent.data());
                    S += n[j];
                }
                int* hdr = &o.hdr[((size_t)(tb0 + bi) * S_NW + w) * 4];
                hdr[0] = (int)cur[w];
                hdr[1] = n[0] | n[1] << 16;
                hdr[2] = blk;
                hdr[3] = 0;
                cur[w] = S;
            }
        }
    });
    (void)n_src;
    return MLLP_OK;
}

template <class G>
static int64_t walk_stream_t(const HostStream& s, int64_t n_dst, int64_t n_src, const float* H, double* Y) {
    constexpr int S_NW = G::NW, S_P = G::P, S_GS = G::GS, S_RQ = G::RQ, S_CB = G::CB, S_ZERO_OFF = G::ZERO_OFF, S_ROW_BYTES = G::ITEM;
    int64_t real = 0;
    for (int t = 0; t < s.n_tiles; ++t) {
        for (int tb = s.tile_blk[t]; tb < s.tile_blk[t + 1]; ++tb) {
            const int64_t c0 = (int64_t)s.blk_id[tb] * S_CB;
            for (int w = 0; w < S_NW; ++w) {
                const int* hdr = &s.hdr[((size_t)tb * S_NW + w) * 4];
                const int n[2] = {hdr[1] & 0xffff, (int)((unsigned)hdr[1] >> 16)};
                for (int q = 0; q < 16; ++q) {
                    const int* rows = &s.rows[(((size_t)tb * S_NW + w) * 16 + q) * 4];
                    int64_t a = hdr[0];
                    for (int pass = 0; pass < S_P; ++pass) {
                        const int64_t b = a + n[pass];
                        for (int64_t st = a; st < b; ++st) {
                            if (st / S_GS >= s.n_groups) return -1;
                            for (int slot = 0; slot < S_RQ; ++slot) {
                                const int* g = &s.ent[(size_t)G::ent_index(st, q, slot)];
                                const int off = (int)(((unsigned)g[0] >> ((st & 1) * 16)) & 0xffffu);
                                const int* e = g + (st & 1);         // e[1] = value bits
                                if (off == S_ZERO_OFF) {
                                    if (e[1] != 0) return -1;
                                    continue;
                                }
                                if (off < 0 || off % S_ROW_BYTES || off >= S_ZERO_OFF) return -1;
                                const int64_t col = c0 + off / S_ROW_BYTES;
                                const int rr = rows[2 * pass + (slot >> 1)];
                                const int64_t row = (int64_t)s.tile_row[t] + ((slot & 1) ? (rr >> 16) & 0xffff : rr & 0xffff);
                                if (col >= n_src || row >= s.tile_row[t + 1] || row >= n_dst) return -1;
                                float v;
                                std::memcpy(&v, &e[1], 4);
                                for (int c = 0; c < 16; ++c) Y[row * 16 + c] += (double)v * (double)H[col * 16 + c];
                                ++real;
                            }
                        }
                        a = b;
                    }
                }
            }
        }
    }
    return real;
}

// ---- public entry points: geometry by id (stream_layout.h::STREAM_GEOM_*) ---------------------------------------
#define MLLP_GEOM_DISPATCH(geom, CALL)                       \
    switch (geom) {                                          \
        case STREAM_GEOM_SPMM: { using G = SpmmGeom; CALL; } \
        case STREAM_GEOM_ATTN: { using G = AttnGeom; CALL; } \
        case STREAM_GEOM_BSRC: { using G = BsrcGeom; CALL; } \
        case STREAM_GEOM_BDST: { using G = BdstGeom; CALL; } \
        default: break;                                      \
    }

std::vector<int> host_stream_tiles(const int64_t* seg_ptr, int64_t n_seg, int64_t n_dst, int geom) {
    MLLP_GEOM_DISPATCH(geom, return stream_tiles_t<G>(seg_ptr, n_seg, n_dst))
    return std::vector<int>();
}

int host_build_stream(const int* ptr, const int* idx, const float* val, int64_t n_dst, int64_t n_src,
                      const int64_t* seg_ptr, int64_t n_seg, HostStream* out, std::string* err, unsigned max_threads, int geom) {
    MLLP_GEOM_DISPATCH(geom, return build_stream_t<G>(ptr, idx, val, n_dst, n_src, seg_ptr, n_seg, out, err, max_threads))
    if (err) *err = "host_build_stream: unknown geometry";
    return MLLP_EINVAL;
}

int64_t host_walk_stream(const HostStream& s, int64_t n_dst, int64_t n_src, const float* H, double* Y, int geom) {
    MLLP_GEOM_DISPATCH(geom, return walk_stream_t<G>(s, n_dst, n_src, H, Y))
    return -1;
}

// ------------------------------------------------------------------------------------------------ lane-per-row copy
int host_build_lane(const int* ptr, const int* idx, const float* val, int64_t n_dst, const int64_t* seg_ptr, int64_t n_seg,
                    HostLane* out, std::string* err) {
    HostLane& h = *out;
    h = HostLane();
    std::vector<int64_t> seg;
    if (seg_ptr && n_seg > 0) seg.assign(seg_ptr, seg_ptr + n_seg + 1);
    else { seg.push_back(0); seg.push_back(n_dst); }
    int64_t covered = 0;
    for (size_t i = 0; i + 1 < seg.size(); ++i) {
        for (int64_t r = seg[i]; r < seg[i + 1]; r += L1_R) h.tile_row.push_back((int)r);
        covered = seg[i + 1];
    }
    for (int64_t r = covered; r < n_dst; r += L1_R) h.tile_row.push_back((int)r);
    h.tile_row.push_back((int)n_dst);
    h.n_tiles = (int)h.tile_row.size() - 1;
    h.real_slots = n_dst > 0 ? ptr[n_dst] : 0;
    if (n_dst <= 0) { h.n_tiles = 0; h.tile_row.assign(1, 0); }
    h.tile_blk.assign((size_t)h.n_tiles + 1, 0);
    h.tile_col.assign((size_t)h.n_tiles * 2, 0);
    h.rows.assign((size_t)h.n_tiles * L1_R, -1);
    // pass 1: the tiles' column ranges, the order of their rows, blocks per tile
    for (int t = 0; t < h.n_tiles; ++t) {
        const int r0 = h.tile_row[t], nr = h.tile_row[t + 1] - r0;
        int cmin = INT32_MAX, cmax = -1;
        std::vector<int> order(nr);
        for (int i = 0; i < nr; ++i) {
            order[i] = i;
            const int b = ptr[r0 + i], e = ptr[r0 + i + 1];
            if (e > b) { cmin = std::min(cmin, idx[b]); cmax = std::max(cmax, idx[e - 1]); }
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {      // descending length, ties by row id
            return ptr[r0 + a + 1] - ptr[r0 + a] > ptr[r0 + b + 1] - ptr[r0 + b];
        });
        for (int p = 0; p < nr; ++p) h.rows[(size_t)t * L1_R + p] = order[p];
        const int c0 = cmax < 0 ? 0 : cmin & ~3;
        h.tile_col[2 * t] = c0;
        h.tile_col[2 * t + 1] = cmax;
        const int nb = cmax < 0 ? 0 : (cmax - c0) / L1_CB + 1;
        h.tile_blk[t + 1] = h.tile_blk[t] + nb;
        if (h.tile_blk[t + 1] >= (1 << 26)) {
            if (err) *err = "lane copy: too many (tile, block) pairs";
            return MLLP_ERANGE;
        }
    }
    h.n_tb = h.tile_blk[h.n_tiles];
    h.whdr.assign((size_t)h.n_tb * L1_NW * 2, 0);
    // pass 2: groups of every (tile, wavefront, block); a row's entries of block b are those below the block's last column
    auto row_stop = [&](int pos, int end, int nb, int b, int cb) {
        return b + 1 == nb ? end : (int)(std::lower_bound(idx + pos, idx + end, cb + L1_CB) - idx);
    };
    for (int t = 0; t < h.n_tiles; ++t) {
        const int nb = h.tile_blk[t + 1] - h.tile_blk[t], c0 = h.tile_col[2 * t];
        for (int w = 0; w < L1_NW; ++w) {
            std::vector<int> most(nb, 0);
            for (int l = 0; l < 64; ++l) {
                const int local = h.rows[(size_t)t * L1_R + 64 * w + l];
                if (local < 0) continue;
                int pos = ptr[h.tile_row[t] + local];
                const int end = ptr[h.tile_row[t] + local + 1];
                for (int b = 0; b < nb; ++b) {
                    const int stop = row_stop(pos, end, nb, b, c0 + b * L1_CB);
                    most[b] = std::max(most[b], stop - pos);
                    pos = stop;
                }
            }
            for (int b = 0; b < nb; ++b) h.whdr[((size_t)h.tile_blk[t] * L1_NW + (size_t)w * nb + b) * 2 + 1] = (most[b] + L1_GS - 1) / L1_GS;
        }
    }
    int64_t groups = 0;
    for (size_t i = 0; i < (size_t)h.n_tb * L1_NW; ++i) {
        h.whdr[2 * i] = (int)groups;
        groups += h.whdr[2 * i + 1];
        if (groups >= (1ll << 31) / 64) {
            if (err) *err = "lane copy: the stream exceeds int32 indexing";
            return MLLP_ERANGE;
        }
    }
    h.n_groups = groups;
    const uint32_t padw = (uint32_t)L1_PAD | (uint32_t)L1_PAD << 16;
    h.offs.assign((size_t)(groups + L1_PADG) * 64 * 2, padw);
    h.vals.assign((size_t)(groups + L1_PADG) * 64 * 4, 0.0f);
    // pass 3: the entries
    for (int t = 0; t < h.n_tiles; ++t) {
        const int nb = h.tile_blk[t + 1] - h.tile_blk[t], c0 = h.tile_col[2 * t];
        for (int p = 0; p < L1_R; ++p) {
            const int local = h.rows[(size_t)t * L1_R + p];
            if (local < 0) continue;
            const int w = p >> 6, l = p & 63;
            int pos = ptr[h.tile_row[t] + local];
            const int end = ptr[h.tile_row[t] + local + 1];
            for (int b = 0; b < nb; ++b) {
                const int cb = c0 + b * L1_CB, stop = row_stop(pos, end, nb, b, cb);
                const int* hd = &h.whdr[((size_t)h.tile_blk[t] * L1_NW + (size_t)w * nb + b) * 2];
                for (int e = pos; e < stop; ++e) {
                    const int s = e - pos;
                    const size_t slot = ((size_t)hd[0] + s / 4) * 64 + l;
                    uint32_t& word = h.offs[slot * 2 + ((s & 3) >> 1)];
                    const uint32_t o = (uint32_t)(idx[e] - cb);
                    word = (s & 1) ? ((word & 0xffffu) | o << 16) : ((word & 0xffff0000u) | o);
                    h.vals[slot * 4 + (s & 3)] = val[e];
                }
                pos = stop;
            }
        }
    }
    return MLLP_OK;
}

int64_t host_walk_lane(const HostLane& s, int64_t n_dst, int64_t n_src, const float* x, double* y) {
    int64_t real = 0;
    if ((int)s.tile_row.size() != s.n_tiles + 1 || (int)s.tile_blk.size() != s.n_tiles + 1) return -1;
    for (int t = 0; t < s.n_tiles; ++t) {
        const int nb = s.tile_blk[t + 1] - s.tile_blk[t], c0 = s.tile_col[2 * t];
        for (int w = 0; w < L1_NW; ++w)
            for (int b = 0; b < nb; ++b) {
                const int* hd = &s.whdr[((size_t)s.tile_blk[t] * L1_NW + (size_t)w * nb + b) * 2];
                for (int g = 0; g < hd[1]; ++g)
                    for (int l = 0; l < 64; ++l) {
                        const size_t slot = ((size_t)hd[0] + g) * 64 + l;
                        if ((slot + 1) * 4 > s.vals.size()) return -1;
                        const int local = s.rows[(size_t)t * L1_R + 64 * w + l];
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t o = (s.offs[slot * 2 + (k >> 1)] >> (16 * (k & 1))) & 0xffffu;
                            if (o == (uint32_t)L1_PAD) {
                                if (s.vals[slot * 4 + k] != 0.0f) return -1;
                                continue;
                            }
                            const int64_t col = (int64_t)c0 + (int64_t)b * L1_CB + o, row = (int64_t)s.tile_row[t] + local;
                            if (local < 0 || o >= (uint32_t)L1_CB || col >= n_src || row >= s.tile_row[t + 1] || row >= n_dst) return -1;
                            y[row] += (double)s.vals[slot * 4 + k] * (double)x[col];
                            ++real;
                        }
                    }
            }
    }
    return real;
}

}  // namespace mllp
