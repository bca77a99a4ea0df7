// host_stream.h -- host (plain C++) builder of the streamed SpMM copy (stream_layout.h) and a host walk of it.
// The device builder (stream_spmm.hip) must produce the same bytes; the GPU tests compare the two, and the sanitizer
// driver (host_graph_test.cpp) checks this one against the CSR it came from by walking the copy exactly as the
// kernel's wavefronts do.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace mllp {

struct HostStream {
    int n_tiles = 0, n_tb = 0;
    int64_t n_groups = 0;            // groups of 2 steps (without the padding groups at the end)
    int64_t real_slots = 0;          // = nnz
    int64_t step_slots = 0;          // 128 x groups: entry slots of the stream (padding included)
    std::vector<int> tile_row;       // [n_tiles + 1]
    std::vector<int> tile_blk;       // [n_tiles + 1]
    std::vector<int> blk_id;         // [n_tb]
    std::vector<int> rows;           // [n_tb * 8 * 16 * 4]
    std::vector<int> hdr;            // [n_tb * 8 * 4]
    std::vector<int> ent;            // [(n_groups + S_K0) * 64 * S_ENT]
};

// Row tiles of at most S_RR rows that never cross a segment boundary (seg_ptr: [n_seg + 1] ascending row offsets of the
// LP instances, first 0, last n_dst; null = one segment).
std::vector<int> host_stream_tiles(const int64_t* seg_ptr, int64_t n_seg, int64_t n_dst, int geom = 0);

// geom: stream_layout.h::STREAM_GEOM_* (0 = the plain SpMM's).  The walk decodes an entry's item index from its byte
// offset (offset / ITEM) and multiplies with row `item` of H [n_src, 16] whatever the geometry's item size is.
// 0 on success, an MLLP_E* code with *err set otherwise (sizes beyond int32 steps).  max_threads = 0: hardware
// concurrency, at most 16.
int host_build_stream(const int* ptr, const int* idx, const float* val, int64_t n_dst, int64_t n_src,
                      const int64_t* seg_ptr, int64_t n_seg, HostStream* out, std::string* err, unsigned max_threads = 0, int geom = 0);

// Y[n_dst,16] (double) += the product, computed by walking the copy the way spmm_stream_kernel does (records, passes,
// steps, groups); returns the number of real (non-padding) entries visited, -1 on a malformed copy.
int64_t host_walk_stream(const HostStream& s, int64_t n_dst, int64_t n_src, const float* H, double* Y, int geom = 0);

// ---- the lane-per-row copy of the layer-1 sweeps (lane_layout.h; device builder and kernels in lane_stream.hip) ----------
struct HostLane {
    int n_tiles = 0, n_tb = 0;
    int64_t n_groups = 0;            // groups of 4 steps (without the padding groups at the end)
    int64_t real_slots = 0;          // = nnz
    std::vector<int> tile_row;       // [n_tiles + 1]
    std::vector<int> tile_blk;       // [n_tiles + 1]
    std::vector<int> tile_col;       // [n_tiles][2]
    std::vector<int> rows;           // [n_tiles][L1_R]
    std::vector<int> whdr;           // [n_tb][L1_NW][2]
    std::vector<uint32_t> offs;      // [(n_groups + L1_PADG) * 64 * 2]
    std::vector<float> vals;         // [(n_groups + L1_PADG) * 64 * 4]
};
// Host reference builder: the device builder (lane_stream.hip::build_lane_copy) must produce the same bytes (GPU test).
int host_build_lane(const int* ptr, const int* idx, const float* val, int64_t n_dst, const int64_t* seg_ptr, int64_t n_seg,
                    HostLane* out, std::string* err);
// y[n_dst] (double) += A x computed by walking the copy the way the lanes of lane1_kernel do (tile, block, wavefront, group,
// lane, step); returns the number of real entries visited, -1 on a malformed copy.
int64_t host_walk_lane(const HostLane& s, int64_t n_dst, int64_t n_src, const float* x, double* y);

}  // namespace mllp
