// stream_layout.h -- layout of the "streamed" copy of one orientation, the input of spmm_stream_kernel
// (stream_spmm.hip).  Shared by the host builder (host_stream.cpp, plain C++), the device builder and the kernel.
//
// Why another copy: the wave-specialised tiled SpMM of rounds 1-2 (tiled_kernels.hip::spmm_tiled_ws_kernel) staged the
// nonzeros of a (512-row tile, 1024-column block) in a 48 KB LDS window next to ONE 64 KB image of H, so the walkers
// stood idle while the next image was written (24.5 % of the kernel, profiles/r02_spmm_ws_phase_cycles.txt) and every
// nonzero cost 16 bytes of LDS write + read traffic on top of its 64-byte H row.  Here the nonzeros never enter LDS:
//   * the builder stores them in the order the LANES consume them, so a wavefront reads its share of a
//     (tile, block) straight into registers with coalesced 1 KB loads, one block ahead;
//   * LDS then holds TWO images of H (filled by LDS-DMA, no registers) + the tile's accumulators: image b + 1
//     lands while image b is walked; one barrier per block.
//
// Geometry: row tiles of at most S_RR = 960 destination rows (S_R = 1024 row slots) x column blocks of S_CB = 750 source nodes.  The first
// version (512 x 1000) measured bound by the CU's vector-memory fill path, not by LDS (profiles/r03_stream_*): every
// tile stages every block of its instance, 6400 / S_R bytes of H per nonzero at 1 % density whatever S_CB is, most of
// it L2 hits but 40 % of them misses (the hot H of the 2-3 instances an XCD works on does not fit its 4 MB L2 next to
// the entry stream).  1024 row slots halve that traffic; 64 KB of accumulators + 2 x (48 000 B + one all-zero row) of
// images = 161 664 B of LDS.  Tiles never cross an instance boundary (tile_row): a tile that straddles two instances
// would stage the blocks of both for half the rows each.
//
// Work split inside a (tile, block): the rows are ordered by their number of entries in the block (descending, ties
// by row id) and cut into S_NB = 16 BUNDLES of S_BR = 64 positions.  Wavefront w (of S_NW = 8) walks S_P = 2
// passes, pass j over bundle 8 j + (j odd ? 7 - w : w) (long rows with short rows: similar step counts for all
// wavefronts).  In a pass, quad q of the wavefront owns S_RQ = 4 rows, slot r = position 64 b + 16 r + q; a STEP is one
// entry of each of the quad's rows for each of the 16 quads; the pass takes n = (entries of the bundle's longest
// row) steps, shorter rows are padded with entries {zero row, 0.0f} that read the all-zero row behind the image
// (10 % of the slots on the synthetic batch).
// Tiles hold at most S_RR = 960 rows: the missing rows sort last and form the empty bundle 15, so wavefront 0, whose
// bundle 0 holds the rows of the long tail, has no second pass.  With 1024 rows its two passes took 1.38 x the steps of
// the average wavefront and everybody waited for it at the barrier of every block; with 960 it is 1.16 x.
// (Why 8 wavefronts of 4 rows per quad rather than 16 of 2: a wavefront issues about one instruction per 4-5 cycles
// whatever the instruction is, and the address / broadcast / guard instructions of a step are per quad-row group, not per
// row -- profiles/r03_stream_experiments.txt.)
//
// Arrays:
//   tile_row [n_tiles + 1]   first destination row of each tile (rows of tile t: tile_row[t] .. tile_row[t + 1] - 1)
//   tile_blk [n_tiles + 1]   first (tile, block) index of each tile
//   blk_id   [n_tb]          global column-block id of each (tile, block), ascending inside a tile
//   rows     [n_tb][S_NW][16] int4 per (tile-block, wavefront, quad): component 2 j + (r >> 1) holds row slot r of
//                            pass j in its low (r even) or high (r odd) 16 bits (rows inside the tile)
//   hdr      [n_tb][S_NW]    int4 per (tile-block, wavefront): .x = index of the wavefront's first step of this block
//                            in `ent`, .y = n0 | n1 << 16 (steps of the two passes), .z = blk_id of the block, .w = 0
//   ent      [(n_groups + S_K0) * 64 * 3] int32: a GROUP is S_GS = 8 / S_RQ consecutive steps = 768 B; lane
//                            4 q + p of a wavefront loads the 12 bytes number group * 64 + 4 q + p: row slot
//                            r = p % S_RQ of quad q, steps S_GS group + 2 (p / S_RQ) and the one behind it:
//                              {o(first) | o(second) << 16, value bits of the first, of the second}
//                            with o = byte offset of the source row in the image (< 2^16): 6 bytes per nonzero instead
//                            of CSR's 8.
//                            The steps of a (tile, wavefront) are contiguous over its blocks (pass 0 then pass 1 of
//                            each block) and start at a group boundary; S_K0 padding groups at the very end let the
//                            wavefronts load their register sets unconditionally.
#pragma once
#include <stdint.h>

namespace mllp {

// A geometry of the streamed layout.  Round 4 carries the layout to the attention sweeps (stream_attn.hip), which keep
// more state per row in LDS than the plain SpMM and therefore take smaller tiles: same arrays, same builders
// (host_stream.cpp / stream_build.hip are templates over this struct), other constants.
//   R     row slots per tile            RR    rows per tile (at most; the slots above sort last and stay empty)
//   CB    source nodes per column block NW    walking wavefronts           RQ   rows per quad and pass (2 or 4)
//   ITEM  bytes of one staged item (a 64-byte feature row, or the 160-byte backward record of the source-major sweep)
//   K0/K1 groups of pass 0 / pass 1 that a wavefront holds in registers (K0 groups of padding end the stream)
template <int R_, int RR_, int CB_, int NW_, int RQ_, int ITEM_, int K0_, int K1_>
struct StreamGeomT {
    static constexpr int R = R_, RR = RR_, CB = CB_, NW = NW_, RQ = RQ_, ITEM = ITEM_, K0 = K0_, K1 = K1_;
    static constexpr int GS = 8 / RQ;             // steps per group
    static constexpr int BR = 16 * RQ;            // rows per bundle
    static constexpr int NB = R / BR;             // bundles of sorted positions
    static constexpr int P = NB / NW;             // passes per wavefront and block
    static constexpr int ENT = 3;                 // int32 per (group, lane)
    static constexpr int ZERO_OFF = CB * ITEM;    // byte offset of the all-zero item behind the image
    static constexpr int PAD_WORD = ZERO_OFF | ZERO_OFF << 16;   // offset word of two padding entries
    static_assert(ZERO_OFF + ITEM <= 65536, "byte offsets inside the image are stored in 16 bits");
    static_assert(P * NW == NB && P == 2 && R <= 1024 && (RQ == 2 || RQ == 4) && RR <= R && ITEM % 16 == 0, "two passes per wavefront");
    // bundle of sorted positions that wavefront w walks in its pass j
    static constexpr int bundle(int w, int j) { return NW * j + ((j & 1) ? NW - 1 - w : w); }
    // int32 index of the {offsets, value, value} triple that holds step `st` of row slot r of quad q
    static constexpr int64_t ent_index(int64_t st, int q, int r) {
        return ((st / GS) * 64 + q * 4 + ((st % GS) >> 1) * RQ + r) * 3;
    }
};

// geometry ids of the C ABI (mllp_graph_build_stream_copy / _info / _export)
constexpr int STREAM_GEOM_SPMM = 0, STREAM_GEOM_ATTN = 1, STREAM_GEOM_BSRC = 2, STREAM_GEOM_BDST = 3, STREAM_GEOMS = 4;
using SpmmGeom = StreamGeomT<1024, 960, 750, 8, 4, 64, 12, 4>;     // plain SpMM (stream_spmm.hip), rationale above
// The attention sweeps (stream_attn.hip): 16 bundles of 32 rows (two rows per quad), so that eight wavefronts still walk a
// long and a short bundle each, two wavefronts per SIMD.  A row keeps state in LDS from block to block (480 rows + one
// dummy row for the slots above a tile's last row), and what is left of the 160 KB goes to the two images:
//   forward                      q', Z, {L, u, m, t}: 144 B per row      -> 720-column blocks
//   destination-major backward   q', gv, dq', 8 scalars: 224 B per row   -> 432-column blocks
//   source-major backward        x_j, dX_j: 128 B per row; the staged items are the 160-byte destination records -> 312
using AttnGeom = StreamGeomT<512, 480, 720, 8, 2, 64, 6, 3>;
using BdstGeom = StreamGeomT<512, 480, 432, 8, 2, 64, 4, 2>;
using BsrcGeom = StreamGeomT<512, 480, 312, 8, 2, 160, 4, 2>;

constexpr int S_R = SpmmGeom::R;                   // row slots per tile
constexpr int S_RR = SpmmGeom::RR;                 // rows per tile (at most): the last bundle stays empty, see above
constexpr int S_CB = SpmmGeom::CB;                 // source nodes per column block
constexpr int S_NW = SpmmGeom::NW;                 // walking wavefronts per workgroup
constexpr int S_RQ = SpmmGeom::RQ;                 // rows per quad and pass
constexpr int S_GS = SpmmGeom::GS;                 // steps per group
constexpr int S_BR = SpmmGeom::BR;                 // rows per bundle
constexpr int S_NB = SpmmGeom::NB;                 // bundles of sorted positions
constexpr int S_P = SpmmGeom::P;                   // passes per wavefront and block
constexpr int S_K0 = SpmmGeom::K0, S_K1 = SpmmGeom::K1;   // groups of pass 0 / pass 1 that a wavefront holds in registers
constexpr int S_ENT = SpmmGeom::ENT;               // int32 per (group, lane)
constexpr int S_ROW_BYTES = SpmmGeom::ITEM;        // one fp32 feature row
constexpr int S_ZERO_OFF = SpmmGeom::ZERO_OFF;     // byte offset of the all-zero row behind the image
constexpr int S_PAD_WORD = SpmmGeom::PAD_WORD;     // offset word of two padding entries

constexpr int s_bundle(int w, int j) { return SpmmGeom::bundle(w, j); }
constexpr int64_t s_ent_index(int64_t st, int q, int r) { return SpmmGeom::ent_index(st, q, r); }

// the four quads whose rows are read in the same LDS cycle of a ds_read_b128 (lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS): their source rows should sit in four different
// quarters (column mod 4) of the 256-byte bank row
constexpr int S_TEAMS[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

}  // namespace mllp
