// stream_layout.h -- layout of the "streamed" copy of one orientation, the input of spmm_stream_kernel
// (stream_spmm.hip).  Shared by the host builder (host_stream.cpp, plain C++), the device builder and the kernel.
//
// Why another copy: the wave-specialised tiled SpMM of rounds 1-2 (tiled_kernels.hip::spmm_tiled_ws_kernel) staged the
// nonzeros of a (512-row tile, 1024-column block) in a 48 KB LDS window next to ONE 64 KB image of H, so the walkers
// stood idle while the next image was written (24.5 % of the kernel, profiles/r02_spmm_ws_phase_cycles.txt) and every
// nonzero cost 16 bytes of LDS write + read traffic on top of its 64-byte H row.  Here the nonzeros never enter LDS:
//   * the builder stores them in the order the walker LANES consume them, so a walker wavefront reads its share
//     of a (tile, block) straight into registers with coalesced 1 KB loads, one block ahead;
//   * LDS then holds TWO images of H (filled by LDS-DMA, no registers) + the tile's accumulators: image b + 1
//     lands while image b is walked; one barrier per block.
//
// Geometry: row tiles of S_R = 512 destination rows x column blocks of S_CB = 1000 source nodes (64 000 B of fp32
// rows; 2 x (64 000 + one all-zero row) + 32 KB of accumulators = 160 896 B of the CU's 163 840 B of LDS).
//
// Work split inside a (tile, block): the rows are ordered by their number of entries in the block (descending, ties
// by row id) and cut into 16 PAIRS of 32 positions.  Walker wavefront w (of S_NW = 8) walks pair w in its pass 0 and
// pair 15 - w in its pass 1 (long rows with short rows: equal step counts for all wavefronts).  In a pass, quad q of
// the wavefront owns two rows, A = position 32 p + q and B = position 32 p + 16 + q; a STEP is one entry of A and one
// of B for each of the 16 quads; the pass takes n = (entries of the pair's longest row) steps, shorter rows are
// padded with entries {S_ZERO_OFF, 0.0f} that read the all-zero row behind the image.
//
// Arrays:
//   tile_blk [n_tiles + 1]   first (tile, block) index of each tile
//   blk_id   [n_tb]          global column-block id of each (tile, block), ascending inside a tile
//   rec      [n_tb][8][16]   int4 per (tile-block, wavefront, quad):
//                              .x = row A | row B << 16 of pass 0 (rows inside the tile),  .y = the same for pass 1,
//                              .z = index of the wavefront's first step of this block in `ent`,
//                              .w = n0 | n1 << 16 (steps of pass 0 / pass 1; the same in all 16 quads)
//   ent      [(n_groups + S_K) * 64] int4: a GROUP is four consecutive steps = 1 KB; lane 4 q + p of a wavefront
//                              loads int4 number group * 64 + 4 q + p = step 4 group + p of quad q:
//                              {byte offset of A's source row in the image, value bits, the same for B}.
//                            The steps of a (tile, wavefront) are contiguous over its blocks (pass 0 then pass 1 of
//                            each block) and start at a group boundary; S_K padding groups at the very end let the
//                            walkers load S_K groups unconditionally.
#pragma once
#include <stdint.h>

namespace mllp {

constexpr int S_R = 512;                    // rows per tile
constexpr int S_CB = 1000;                  // source nodes per column block
constexpr int S_NW = 8;                     // walker wavefronts per workgroup
constexpr int S_PAIRS = S_R / 32;           // 16 pairs of 32 sorted positions
constexpr int S_K = 8;                      // groups (of 4 steps) of a block that a walker holds in registers
constexpr int S_ROW_BYTES = 64;             // one fp32 feature row
constexpr int S_ZERO_OFF = S_CB * S_ROW_BYTES;   // byte offset of the all-zero row behind the image
static_assert(S_PAIRS == 2 * S_NW, "two passes per walker");

// the four quads whose rows are read in the same LDS cycle of a ds_read_b128 (lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS): their source rows should sit in four different
// quarters (column mod 4) of the 256-byte bank row
constexpr int S_TEAMS[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

}  // namespace mllp
