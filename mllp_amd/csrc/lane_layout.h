// lane_layout.h -- layout of the "lane-per-row" streamed copy of one orientation, the input of the layer-1 (one input
// channel) attention sweeps of lane_stream.hip.
//
// Why another copy: with ONE channel the staged item of a source node is 4 bytes, so a whole instance's x (20 000 columns
// = 80 KB on the synthetic batch) fits in LDS next to nothing else -- a row tile then has ONE column block instead of the
// 20 (1024 columns) of the LDS-tiled layer-1 kernel (tiled_kernels.hip::scalar_tiled_kernel: 2.0 ms forward / 1.75 ms
// backward at 512 M nonzeros, 25 % of the HBM time of its 8 bytes per nonzero), a row needs no state outside the
// registers of the lane that walks it, and nothing is synchronised between the staging of the image and the epilogue.
//
// Geometry: row tiles of at most L1_R = 512 rows that never cross an instance boundary (tile_row); the columns a tile
// touches, from col0 = (its smallest column id) & ~3, are cut into blocks of L1_CB = 20 000 columns (80 000 B of fp32 +
// one all-zero slot: two workgroups per CU).  The tile's rows are ordered by their number of entries (descending, ties by
// row id): sorted position p belongs to lane p % 64 of wavefront p / 64 for the whole tile, so the row's softmax state
// never leaves that lane's registers.
// A STEP of a wavefront is one entry of each of its 64 rows; in block b the wavefront takes (entries of its longest row
// inside the block) steps, shorter rows are padded with {L1_PAD, 0.0f}; a GROUP is L1_GS = 4 steps = 1536 bytes: lane l
// loads the 8 bytes number group * 64 + l of `offs` (four 16-bit column offsets inside the block, step 4 g + k in bits
// 16 k of the pair of words) and the 16 bytes number group * 64 + l of `vals` (the four values): 6 bytes per nonzero,
// every load of a wavefront one contiguous 512 B / 1 KB.  The groups of a (tile, wavefront) are contiguous over its
// blocks; L1_PADG padding groups end the stream so that the read-ahead needs no guard.
//
// Arrays:
//   tile_row [n_tiles + 1]        first destination row of each tile
//   tile_blk [n_tiles + 1]        first (tile, block) index of each tile
//   tile_col [n_tiles][2]         {col0, last column id the tile touches} ({0, -1} for a tile without entries)
//   rows     [n_tiles][L1_R]      row (inside the tile) of each sorted position, -1 above the tile's rows
//   whdr     [n_tb][L1_NW][2]     {first group, groups} of wavefront w in block b of tile t at index
//                                 (tile_blk[t] * L1_NW + w * nblk(t) + b)
//   offs     [(n_groups + L1_PADG) * 64 * 2] uint32,  vals [(n_groups + L1_PADG) * 64 * 4] float
#pragma once
#include <stdint.h>

namespace mllp {

#ifndef MLLP_L1_R                          // (experiments only: tools/lane_variants.sh)
#define MLLP_L1_R 512
#define MLLP_L1_CB 20000
#endif
constexpr int L1_R = MLLP_L1_R;            // rows per tile = threads per workgroup
constexpr int L1_NW = L1_R / 64;           // wavefronts
constexpr int L1_CB = MLLP_L1_CB;          // columns per block (a multiple of 4)
constexpr int L1_GS = 4;                   // steps per group
constexpr int L1_PAD = L1_CB;              // column offset of a padding entry: the all-zero slot behind the image
constexpr int L1_PADG = 8;                 // padding groups behind the stream (>= the read-ahead of the kernels)
constexpr int L1_LDS = (L1_CB + 4) * 4;    // bytes of LDS: the image + the all-zero slot
static_assert(L1_CB % 4 == 0 && L1_CB <= 65535 && L1_R % 64 == 0, "16-bit column offsets, float4 staging");

constexpr int STREAM_GEOM_LANE1 = 4;       // geometry id of the C ABI (mllp_graph_build_stream_copy / _info / _export)

}  // namespace mllp
