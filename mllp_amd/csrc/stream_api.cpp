// stream_api.cpp -- C ABI of the streamed SpMM copy (stream_layout.h): build (device kernels or the host reference
// builder), info, export, drop.  The copy is library-owned device memory, freed on rebuild / drop / graph destroy.
#include <chrono>
#include <vector>

#include "host_stream.h"
#include "internal.h"
#include "lane_layout.h"
#include "stream_layout.h"

namespace mllp {

void stream_copy_free(StreamCopy& sc) {
    if (sc.tile_row) (void)hipFree(sc.tile_row);
    if (sc.tile_blk) (void)hipFree(sc.tile_blk);
    if (sc.blk_id) (void)hipFree(sc.blk_id);
    if (sc.rows) (void)hipFree(sc.rows);
    if (sc.hdr) (void)hipFree(sc.hdr);
    if (sc.ent) (void)hipFree(sc.ent);
    sc = StreamCopy();
}

static int build_on_host(const Orient& o, int64_t nnz, const std::vector<int64_t>& seg, StreamCopy& sc, hipStream_t s, int geom) {
    std::vector<int> ptr((size_t)o.n_dst + 1, 0), idx((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> val((size_t)std::max<int64_t>(nnz, 1));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    if (o.n_dst > 0) MLLP_HIP_TRY(hipMemcpy(ptr.data(), o.ptr, ((size_t)o.n_dst + 1) * 4, hipMemcpyDeviceToHost));
    if (nnz > 0) {
        MLLP_HIP_TRY(hipMemcpy(idx.data(), o.idx, (size_t)nnz * 4, hipMemcpyDeviceToHost));
        MLLP_HIP_TRY(hipMemcpy(val.data(), o.val, (size_t)nnz * 4, hipMemcpyDeviceToHost));
    }
    HostStream h;
    std::string err;
    const int rc = host_build_stream(ptr.data(), idx.data(), val.data(), o.n_dst, o.n_src, seg.data(), (int64_t)seg.size() - 1, &h, &err, 0, geom);
    if (rc) return fail(rc, err);
    sc.n_tiles = h.n_tiles;
    sc.n_tb = h.n_tb;
    sc.n_groups = h.n_groups;
    sc.step_slots = h.step_slots;
    auto up = [&](int*& d, const std::vector<int>& v) -> int {
        MLLP_HIP_TRY(hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * 4));
        if (!v.empty()) MLLP_HIP_TRY(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        return MLLP_OK;
    };
    int r;
    if ((r = up(sc.tile_row, h.tile_row)) || (r = up(sc.tile_blk, h.tile_blk)) || (r = up(sc.blk_id, h.blk_id)) ||
        (r = up(sc.rows, h.rows)) || (r = up(sc.hdr, h.hdr)) || (r = up(sc.ent, h.ent)))
        return r;
    return MLLP_OK;
}

static int build_lane_on_host(const Orient& o, int64_t nnz, const std::vector<int64_t>& seg, LaneCopy& lc, hipStream_t s) {
    std::vector<int> ptr((size_t)o.n_dst + 1, 0), idx((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> val((size_t)std::max<int64_t>(nnz, 1));
    MLLP_HIP_TRY(hipStreamSynchronize(s));
    MLLP_HIP_TRY(hipMemcpy(ptr.data(), o.ptr, ((size_t)o.n_dst + 1) * 4, hipMemcpyDeviceToHost));
    if (nnz > 0) {
        MLLP_HIP_TRY(hipMemcpy(idx.data(), o.idx, (size_t)nnz * 4, hipMemcpyDeviceToHost));
        MLLP_HIP_TRY(hipMemcpy(val.data(), o.val, (size_t)nnz * 4, hipMemcpyDeviceToHost));
    }
    HostLane h;
    std::string err;
    const int rc = host_build_lane(ptr.data(), idx.data(), val.data(), o.n_dst, seg.data(), (int64_t)seg.size() - 1, &h, &err);
    if (rc) return fail(rc, err);
    auto up = [&](void** d, const void* src, size_t bytes) -> int {
        MLLP_HIP_TRY(hipMalloc(d, std::max<size_t>(bytes, 4)));
        if (bytes) MLLP_HIP_TRY(hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice));
        return MLLP_OK;
    };
    int r;
    if ((r = up((void**)&lc.tile_row, h.tile_row.data(), h.tile_row.size() * 4)) || (r = up((void**)&lc.tile_blk, h.tile_blk.data(), h.tile_blk.size() * 4)) ||
        (r = up((void**)&lc.tile_col, h.tile_col.data(), h.tile_col.size() * 4)) || (r = up((void**)&lc.rows, h.rows.data(), h.rows.size() * 4)) ||
        (r = up((void**)&lc.whdr, h.whdr.data(), h.whdr.size() * 4)) || (r = up((void**)&lc.offs, h.offs.data(), h.offs.size() * 4)) ||
        (r = up((void**)&lc.vals, h.vals.data(), h.vals.size() * 4)))
        return r;
    lc.n_tiles = h.n_tiles; lc.n_tb = h.n_tb; lc.n_groups = h.n_groups; lc.nnz = nnz;
    return MLLP_OK;
}

}  // namespace mllp

using namespace mllp;

#define REQUIRE(cond, msg) \
    if (!(cond)) return fail(MLLP_EINVAL, std::string(__func__) + ": " + (msg))

namespace {
struct GeomInfo { int R, CB, NW, K0, RQ, ITEM; };
bool geom_info(int geom, GeomInfo* gi) {
    switch (geom) {
        case STREAM_GEOM_SPMM: *gi = {SpmmGeom::R, SpmmGeom::CB, SpmmGeom::NW, SpmmGeom::K0, SpmmGeom::RQ, SpmmGeom::ITEM}; return true;
        case STREAM_GEOM_ATTN: *gi = {AttnGeom::R, AttnGeom::CB, AttnGeom::NW, AttnGeom::K0, AttnGeom::RQ, AttnGeom::ITEM}; return true;
        case STREAM_GEOM_BSRC: *gi = {BsrcGeom::R, BsrcGeom::CB, BsrcGeom::NW, BsrcGeom::K0, BsrcGeom::RQ, BsrcGeom::ITEM}; return true;
        case STREAM_GEOM_BDST: *gi = {BdstGeom::R, BdstGeom::CB, BdstGeom::NW, BdstGeom::K0, BdstGeom::RQ, BdstGeom::ITEM}; return true;
        default: return false;
    }
}
StreamCopy* copy_slot(mllp_graph_t* g, int transpose, int geom) {
    Orient& o = transpose ? g->At : g->A;
    return geom == STREAM_GEOM_SPMM ? &o.stream : geom == STREAM_GEOM_ATTN ? &o.stream_attn : geom == STREAM_GEOM_BSRC ? &o.stream_bsrc :
           geom == STREAM_GEOM_BDST ? &o.stream_bdst : nullptr;
}
}  // namespace

extern "C" int mllp_graph_build_stream_copy(mllp_graph_t* g, int transpose, int geom, int where, void* stream) {
    REQUIRE(g, "null graph");
    REQUIRE(where == 0 || where == 1, "where must be 0 (device builder) or 1 (host reference builder)");
    if (geom == STREAM_GEOM_LANE1) {         // the lane-per-row copy of the layer-1 sweeps (lane_layout.h)
        Orient& o = transpose ? g->At : g->A;
        lane_copy_free(o.lane1);
        if (o.n_dst == 0) return MLLP_OK;
        const auto t0 = std::chrono::steady_clock::now();
        LaneCopy lc;
        const std::vector<int64_t>& lseg = transpose ? g->h_inst_ptr_n : g->h_inst_ptr_m;
        const int rc = where == 1 ? build_lane_on_host(o, g->nnz, lseg, lc, (hipStream_t)stream) : build_lane_copy(o, g->nnz, lseg, lc, (hipStream_t)stream);
        if (rc) {
            lane_copy_free(lc);
            return rc;
        }
        lc.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        o.lane1 = lc;
        return MLLP_OK;
    }
    GeomInfo gi;
    REQUIRE(geom_info(geom, &gi), "geom must be 0 (plain SpMM), 1 (attention forward), 2 (source-major backward) or 3 (destination-major backward)");
    Orient& o = transpose ? g->At : g->A;
    StreamCopy& slot = *copy_slot(g, transpose, geom);
    stream_copy_free(slot);
    if (o.n_dst == 0) return MLLP_OK;
    const auto t0 = std::chrono::steady_clock::now();
    StreamCopy sc;
    const std::vector<int64_t>& seg = transpose ? g->h_inst_ptr_n : g->h_inst_ptr_m;     // tiles stay inside an instance
    const int rc = where == 1 ? build_on_host(o, g->nnz, seg, sc, (hipStream_t)stream, geom)
                              : build_stream_device(o, g->nnz, host_stream_tiles(seg.data(), (int64_t)seg.size() - 1, o.n_dst, geom),
                                                    sc, (hipStream_t)stream, geom);
    if (rc) {
        stream_copy_free(sc);
        return rc;
    }
    sc.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    slot = sc;
    return MLLP_OK;
}

extern "C" int mllp_graph_drop_stream_copy(mllp_graph_t* g, int transpose, int geom) {
    REQUIRE(g, "null graph");
    if (geom == STREAM_GEOM_LANE1) {
        lane_copy_free((transpose ? g->At : g->A).lane1);
        return MLLP_OK;
    }
    StreamCopy* sc = copy_slot(g, transpose, geom);
    REQUIRE(sc, "unknown geometry");
    stream_copy_free(*sc);
    return MLLP_OK;
}

extern "C" int mllp_graph_stream_copy_info(const mllp_graph_t* g, int transpose, int geom, int64_t info[8]) {
    REQUIRE(g && info, "null argument");
    if (geom == STREAM_GEOM_LANE1) {
        const LaneCopy& lc = (transpose ? g->At : g->A).lane1;
        info[0] = lc.n_tiles;
        info[1] = lc.n_tb;
        info[2] = lc.n_groups;
        info[3] = lc.n_groups * 64 * L1_GS;                     // entry slots of the stream, padding included
        info[4] = lc.n_tiles ? ((int64_t)lc.n_tiles + 1) * 8 + (int64_t)lc.n_tiles * (8 + L1_R * 4) + (int64_t)lc.n_tb * L1_NW * 8 +
                                   (lc.n_groups + L1_PADG) * 64 * 24
                             : 0;
        info[5] = (int64_t)(lc.build_seconds * 1e6);
        info[6] = L1_R | (int64_t)1 << 16 | (int64_t)4 << 24;   // rows per tile, one row per lane, 4-byte items
        info[7] = (int64_t)L1_CB | (int64_t)L1_NW << 16 | (int64_t)L1_PADG << 24;
        return MLLP_OK;
    }
    GeomInfo gi;
    REQUIRE(geom_info(geom, &gi), "unknown geometry");
    const StreamCopy& sc = *copy_slot(const_cast<mllp_graph_t*>(g), transpose, geom);
    info[0] = sc.n_tiles;
    info[1] = sc.n_tb;
    info[2] = sc.n_groups;
    info[3] = sc.step_slots;
    info[4] = sc.n_tiles ? ((int64_t)sc.n_tiles + 1) * 8 + (int64_t)sc.n_tb * 4 + (int64_t)sc.n_tb * gi.NW * (256 + 16) +
                               (sc.n_groups + gi.K0) * 64 * 3 * 4
                         : 0;                                   // bytes of the copy
    info[5] = (int64_t)(sc.build_seconds * 1e6);                // microseconds the build took (host clock, synchronised)
    info[6] = gi.R | (int64_t)gi.RQ << 16 | (int64_t)gi.ITEM << 24;
    info[7] = gi.CB | gi.NW << 16 | (int64_t)gi.K0 << 24;
    return MLLP_OK;
}

extern "C" int mllp_graph_export_stream_copy(const mllp_graph_t* g, int transpose, int geom, int which, void* host_dst,
                                             int64_t capacity_bytes) {
    REQUIRE(g && host_dst, "null argument");
    if (geom == STREAM_GEOM_LANE1) {
        const LaneCopy& lc = (transpose ? g->At : g->A).lane1;
        REQUIRE(lc.n_tiles > 0, "no lane-per-row copy of this orientation (mllp_graph_build_stream_copy, geometry 4)");
        const void* src = nullptr;
        int64_t bytes = 0;
        switch (which) {
            case 0: src = lc.tile_blk; bytes = ((int64_t)lc.n_tiles + 1) * 4; break;
            case 1: src = lc.tile_col; bytes = (int64_t)lc.n_tiles * 8; break;
            case 2: src = lc.rows; bytes = (int64_t)lc.n_tiles * L1_R * 4; break;
            case 3: src = lc.offs; bytes = (lc.n_groups + L1_PADG) * 64 * 8; break;
            case 4: src = lc.tile_row; bytes = ((int64_t)lc.n_tiles + 1) * 4; break;
            case 5: src = lc.whdr; bytes = (int64_t)lc.n_tb * L1_NW * 8; break;
            case 6: src = lc.vals; bytes = (lc.n_groups + L1_PADG) * 64 * 16; break;
            default: return fail(MLLP_EINVAL, "mllp_graph_export_stream_copy: geometry 4 has arrays 0 (tile_blk), 1 (tile_col), 2 (rows), 3 (offs), 4 (tile_row), 5 (whdr), 6 (vals)");
        }
        REQUIRE(capacity_bytes >= bytes, "destination too small");
        MLLP_HIP_TRY(hipDeviceSynchronize());
        if (bytes > 0) MLLP_HIP_TRY(hipMemcpy(host_dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
        return MLLP_OK;
    }
    GeomInfo gi;
    REQUIRE(geom_info(geom, &gi), "unknown geometry");
    const StreamCopy& sc = *copy_slot(const_cast<mllp_graph_t*>(g), transpose, geom);
    REQUIRE(sc.n_tiles > 0, "no streamed copy of this orientation and geometry (mllp_graph_build_stream_copy)");
    const void* src = nullptr;
    int64_t bytes = 0;
    switch (which) {
        case 0: src = sc.tile_blk; bytes = ((int64_t)sc.n_tiles + 1) * 4; break;
        case 1: src = sc.blk_id; bytes = (int64_t)sc.n_tb * 4; break;
        case 2: src = sc.rows; bytes = (int64_t)sc.n_tb * gi.NW * 256; break;
        case 3: src = sc.ent; bytes = (sc.n_groups + gi.K0) * 64 * 3 * 4; break;
        case 4: src = sc.tile_row; bytes = ((int64_t)sc.n_tiles + 1) * 4; break;
        case 5: src = sc.hdr; bytes = (int64_t)sc.n_tb * gi.NW * 16; break;
        default: return fail(MLLP_EINVAL, "mllp_graph_export_stream_copy: which must be 0 (tile_blk), 1 (blk_id), 2 (rows), 3 (ent), 4 (tile_row) or 5 (hdr)");
    }
    REQUIRE(capacity_bytes >= bytes, "destination too small");
    MLLP_HIP_TRY(hipDeviceSynchronize());
    if (bytes > 0) MLLP_HIP_TRY(hipMemcpy(host_dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return MLLP_OK;
}

// the round-3 entry points: geometry 0
extern "C" int mllp_graph_build_spmm_copy(mllp_graph_t* g, int transpose, int where, void* stream) {
    return mllp_graph_build_stream_copy(g, transpose, STREAM_GEOM_SPMM, where, stream);
}
extern "C" int mllp_graph_drop_spmm_copy(mllp_graph_t* g, int transpose) {
    return mllp_graph_drop_stream_copy(g, transpose, STREAM_GEOM_SPMM);
}
extern "C" int mllp_graph_spmm_copy_info(const mllp_graph_t* g, int transpose, int64_t info[8]) {
    const int rc = mllp_graph_stream_copy_info(g, transpose, STREAM_GEOM_SPMM, info);
    if (rc == MLLP_OK) {      // (the round-3 meaning of the last two words)
        info[6] = S_R;
        info[7] = S_CB | S_NW << 16;
    }
    return rc;
}
extern "C" int mllp_graph_export_spmm_copy(const mllp_graph_t* g, int transpose, int which, void* host_dst,
                                           int64_t capacity_bytes) {
    return mllp_graph_export_stream_copy(g, transpose, STREAM_GEOM_SPMM, which, host_dst, capacity_bytes);
}
