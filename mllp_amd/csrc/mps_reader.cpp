// mps_reader.cpp -- SURVEY.md section 8f-2: MPS file -> the tensors the reference's loader reads
// (reference linear_program_data.py:58-80: <name>_constrs.npz (CSR), _coefs.npy, _rhs.npy; input format e.g.
// /root/reference/netlib_mps/afiro.mps:1-83).  The reference does not contain the script that made its
// dataset/netlib_mps_norm tensors; the rules below were recovered from those tensors and reproduce all 97 of them
// (raw stage exactly, normalized stage to 1e-12): oracle/mps_norm.py states them and is pinned by the reference's
// files; tests/test_mps.py compares this file with the oracle and with the packed reference tensors.
// Plain C++17 host code (no HIP): also built by `make host-sanitize`.
//
//   parse   whitespace tokens; sections ROWS / COLUMNS / RHS / RANGES / BOUNDS / OBJSENSE / ENDATA; the first N row is
//           the objective; rows keep the ROWS order, columns the order of first appearance; an RHS / RANGES line with an
//           even token count has a blank set name; BOUNDS and OBJSENSE are read and ignored (the reference's tensors
//           carry no bounds); integrality MARKER lines are skipped.
//   raw     A (m x n), c, b; every RANGES entry of a constraint row appends a column with one entry in that row:
//           +1 (L row), -1 (G row), sign(R) (E row).
//   norm    every L / G row without a range gets a slack column (+1 / -1) behind all others, in row order; each row is
//           scaled by s = 1 / ||row incl. slack||_2, and where |b s| > 5 by s = 5 / b instead; c / ||c||_2, zero-padded.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mllp_hip.h"

namespace mllp {
int fail(int code, const std::string& msg);   // graph.cpp (product) / host_graph_test.cpp (sanitizer driver)
}

struct mllp_lp {
    int64_t m = 0, n = 0, n_struct = 0, n_range = 0, n_slack = 0;
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> values, coefs, rhs;
    std::vector<int32_t> slack_rows;     // rows that received a slack column (normalized stage), in row order
};

namespace {

void split(const char* line, std::vector<std::string>& out) {
    out.clear();
    const char* p = line;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
        if (!*p) break;
        const char* q = p;
        while (*q && *q != ' ' && *q != '\t' && *q != '\r' && *q != '\n') ++q;
        out.emplace_back(p, q - p);
        p = q;
    }
}

bool to_double(const std::string& s, double* v) {
    char* end = nullptr;
    std::string t = s;
    for (char& ch : t)
        if (ch == 'D' || ch == 'd') ch = 'E';      // Fortran exponents
    *v = std::strtod(t.c_str(), &end);
    return end && *end == 0 && end != t.c_str();
}

}  // namespace

static int mps_read_impl(const char* path, int normalize, mllp_lp_t** out);

// (no C++ exception may cross the C ABI: a malformed or huge file must come back as an error code)
extern "C" int mllp_mps_read(const char* path, int normalize, mllp_lp_t** out) {
    using mllp::fail;
    if (!path || !out) return fail(MLLP_EINVAL, "mllp_mps_read: null argument");
    *out = nullptr;
    try {
        return mps_read_impl(path, normalize, out);
    } catch (const std::bad_alloc&) {
        return fail(MLLP_ENOMEM, std::string("mllp_mps_read: out of memory reading ") + path);
    } catch (const std::exception& e) {
        return fail(MLLP_EINVAL, std::string("mllp_mps_read: ") + e.what() + " (" + path + ")");
    }
}

static int mps_read_impl(const char* path, int normalize, mllp_lp_t** out) {
    using mllp::fail;
    // (the file handle, the getline buffer and the result are owned by RAII holders: an exception thrown while parsing --
    // bad_alloc on a huge file -- unwinds through them)
    struct FileCloser { void operator()(FILE* f) const { if (f) std::fclose(f); } };
    struct LineBuf { char* p = nullptr; ~LineBuf() { std::free(p); } };
    std::unique_ptr<FILE, FileCloser> fh_owner(std::fopen(path, "r"));
    FILE* fh = fh_owner.get();
    if (!fh) return fail(MLLP_EINVAL, std::string("mllp_mps_read: cannot open ") + path);
    std::vector<std::string> rows, cols;
    std::unordered_map<std::string, int> ridx, cidx;
    std::unordered_map<std::string, char> rtype;
    std::string obj;
    bool have_obj = false;
    std::vector<std::map<int, double>> arow;          // per constraint row: column -> value (last assignment wins)
    std::vector<double> c;
    std::unordered_map<std::string, double> rhs, ranges;
    std::vector<std::string> range_order;             // first appearance
    std::string sec;
    std::vector<std::string> f;
    LineBuf lb;
    char*& line = lb.p;
    size_t cap = 0;
    int lineno = 0, rc = MLLP_OK;
    std::string err;
    while (getline(&line, &cap, fh) >= 0) {
        ++lineno;
        if (line[0] == '*') continue;
        split(line, f);
        if (f.empty()) continue;
        if (line[0] != ' ' && line[0] != '\t') {
            sec = f[0];
            if (sec == "ENDATA") break;
            continue;
        }
        if (sec == "ROWS") {
            if (f.size() < 2) { rc = MLLP_EINVAL; err = "ROWS line needs a type and a name"; break; }
            const char t = f[0][0];
            rtype[f[1]] = t;
            if (t == 'N') {
                if (!have_obj) { obj = f[1]; have_obj = true; }
            } else if (t == 'E' || t == 'L' || t == 'G') {
                ridx[f[1]] = (int)rows.size();
                rows.push_back(f[1]);
                arow.emplace_back();
            } else { rc = MLLP_EINVAL; err = "unknown row type"; break; }
        } else if (sec == "COLUMNS") {
            bool marker = false;
            for (const auto& t : f) marker |= t == "'MARKER'";
            if (marker) continue;
            auto it = cidx.find(f[0]);
            int j;
            if (it == cidx.end()) {
                j = (int)cols.size();
                cidx[f[0]] = j;
                cols.push_back(f[0]);
                c.push_back(0.0);
            } else j = it->second;
            for (size_t k = 1; k + 1 < f.size(); k += 2) {
                double v;
                if (!to_double(f[k + 1], &v)) { rc = MLLP_EINVAL; err = "bad number in COLUMNS"; break; }
                if (have_obj && f[k] == obj) c[j] = v;
                else {
                    auto r = ridx.find(f[k]);
                    if (r != ridx.end()) {
                        if (v != 0.0) arow[r->second][j] = v; else arow[r->second].erase(j);
                    }
                }
            }
            if (rc) break;
        } else if (sec == "RHS" || sec == "RANGES") {
            const size_t o = f.size() % 2 == 1 ? 1 : 0;
            for (size_t k = o; k + 1 < f.size(); k += 2) {
                double v;
                if (!to_double(f[k + 1], &v)) { rc = MLLP_EINVAL; err = "bad number in " + sec; break; }
                if (sec == "RHS") rhs[f[k]] = v;
                else {
                    if (!ranges.count(f[k])) range_order.push_back(f[k]);
                    ranges[f[k]] = v;
                }
            }
            if (rc) break;
        }   // BOUNDS, OBJSENSE, NAME continuation: read and ignored
    }
    fh_owner.reset();
    if (rc) return fail(rc, std::string("mllp_mps_read: ") + path + ":" + std::to_string(lineno) + ": " + err);

    const int64_t m = (int64_t)rows.size();
    int64_t n = (int64_t)cols.size();
    std::unique_ptr<mllp_lp> lp_owner(new mllp_lp());
    mllp_lp* lp = lp_owner.get();
    lp->n_struct = n;
    std::vector<double> b((size_t)m, 0.0);
    for (const auto& kv : rhs) {
        auto r = ridx.find(kv.first);
        if (r != ridx.end()) b[r->second] = kv.second;
    }
    // range columns, in the order the ranged rows first appear in RANGES
    std::vector<char> ranged((size_t)m, 0);
    for (const auto& name : range_order) {
        auto r = ridx.find(name);
        if (r == ridx.end()) continue;
        const char t = rtype[name];
        const double sg = t == 'L' ? 1.0 : (t == 'G' ? -1.0 : (ranges[name] >= 0.0 ? 1.0 : -1.0));
        arow[r->second][(int)n] = sg;
        ranged[r->second] = 1;
        c.push_back(0.0);
        ++n;
    }
    lp->n_range = n - lp->n_struct;
    std::vector<double> scale((size_t)m, 1.0);
    if (normalize) {
        for (int64_t i = 0; i < m; ++i) {
            const char t = rtype[rows[i]];
            if ((t == 'L' || t == 'G') && !ranged[i]) {
                arow[i][(int)n] = t == 'L' ? 1.0 : -1.0;
                lp->slack_rows.push_back((int32_t)i);
                c.push_back(0.0);
                ++n;
            }
        }
        lp->n_slack = (int64_t)lp->slack_rows.size();
        for (int64_t i = 0; i < m; ++i) {
            double ss = 0.0;
            for (const auto& kv : arow[i]) ss += kv.second * kv.second;
            const double nrm = std::sqrt(ss);
            double s = nrm > 0.0 ? 1.0 / nrm : 1.0;
            if (std::fabs(b[i] * s) > 5.0) s = 5.0 / b[i];
            scale[i] = s;
        }
        double cs = 0.0;
        for (double v : c) cs += v * v;
        const double cn = std::sqrt(cs);
        if (cn > 0.0)
            for (double& v : c) v /= cn;
    }
    lp->m = m; lp->n = n;
    lp->indptr.assign((size_t)m + 1, 0);
    for (int64_t i = 0; i < m; ++i) {
        for (const auto& kv : arow[i]) {      // std::map: columns ascending
            lp->indices.push_back((int32_t)kv.first);
            lp->values.push_back(scale[i] * kv.second);
        }
        lp->indptr[i + 1] = (int64_t)lp->indices.size();
        b[i] *= scale[i];
    }
    lp->coefs = std::move(c);
    lp->rhs = std::move(b);
    *out = lp_owner.release();
    return MLLP_OK;
}

extern "C" int mllp_lp_dims(const mllp_lp_t* lp, int64_t dims[6]) {
    if (!lp || !dims) return mllp::fail(MLLP_EINVAL, "mllp_lp_dims: null argument");
    dims[0] = lp->m; dims[1] = lp->n; dims[2] = (int64_t)lp->indices.size();
    dims[3] = lp->n_struct; dims[4] = lp->n_range; dims[5] = lp->n_slack;
    return MLLP_OK;
}

extern "C" int mllp_lp_export(const mllp_lp_t* lp, int64_t* indptr, int32_t* indices, double* values, double* coefs,
                              double* rhs, int32_t* slack_rows) {
    if (!lp) return mllp::fail(MLLP_EINVAL, "mllp_lp_export: null argument");
    if (indptr) std::memcpy(indptr, lp->indptr.data(), lp->indptr.size() * sizeof(int64_t));
    if (indices && !lp->indices.empty()) std::memcpy(indices, lp->indices.data(), lp->indices.size() * sizeof(int32_t));
    if (values && !lp->values.empty()) std::memcpy(values, lp->values.data(), lp->values.size() * sizeof(double));
    if (coefs && !lp->coefs.empty()) std::memcpy(coefs, lp->coefs.data(), lp->coefs.size() * sizeof(double));
    if (rhs && !lp->rhs.empty()) std::memcpy(rhs, lp->rhs.data(), lp->rhs.size() * sizeof(double));
    if (slack_rows && !lp->slack_rows.empty())
        std::memcpy(slack_rows, lp->slack_rows.data(), lp->slack_rows.size() * sizeof(int32_t));
    return MLLP_OK;
}

extern "C" int mllp_lp_free(mllp_lp_t* lp) {
    delete lp;
    return MLLP_OK;
}
