// Host-side bulk reader for chunked HDF5 datasets.  No device code.
//
// The reference fills a batch with one h5py read per residue (design_utils/utils.py:514-529: every frame is its
// own gzip-chunked dataset `pdb/chain/residue`, 32 chunks of (6,11,11,3) float64 with h5py's defaults).  Here the
// Python side (timed_hip/h5lite.py) only resolves each residue's object header; this function then walks the
// chunk B-trees (v1), inflates the chunks and scatters them into the caller's batch array for MANY datasets at
// once on host threads — the per-chunk work never touches the interpreter.
//
// Supported, i.e. what h5py/HDF5 1.10 writes for aposteriori frame datasets: layout v3 chunked storage indexed by
// a version-1 B-tree, filter pipeline of deflate (1), shuffle (2), fletcher32 (3).  Anything else returns
// TH_EUNSUP and the caller falls back to the pure-Python path.
#include "common.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kUndef = 0xFFFFFFFFFFFFFFFFull;

struct Geometry {
    const uint8_t* file; int64_t file_len, base;
    int rank;                       // real dimensions (the B-tree keys carry rank + 1 offsets)
    int64_t shape[8]; int64_t chunk[8]; int esz;
    int n_filters; int filters[8];
    int64_t chunk_bytes;
};

template <typename T> inline T rd(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof v); return v; }

struct Scratch { std::vector<uint8_t> a, b; };

// undo the filter pipeline of one chunk; returns a pointer to chunk_bytes of data or nullptr
const uint8_t* unfilter(const Geometry& g, const uint8_t* src, size_t len, uint32_t mask, Scratch& s, std::string* err) {
    const uint8_t* cur = src;
    size_t cur_len = len;
    bool use_a = true;
    for (int i = g.n_filters - 1; i >= 0; --i) {
        if (mask & (1u << i)) continue;
        std::vector<uint8_t>& dst = use_a ? s.a : s.b;
        switch (g.filters[i]) {
            case 3:   // fletcher32: checksum appended
                if (cur_len < 4) { *err = "fletcher32 chunk shorter than its checksum"; return nullptr; }
                cur_len -= 4;
                break;
            case 1: {  // deflate
                dst.resize((size_t)g.chunk_bytes + 8);
                uLongf out_len = (uLongf)dst.size();
                const int rc = uncompress(dst.data(), &out_len, cur, (uLong)cur_len);
                if (rc != Z_OK) { *err = "zlib uncompress failed (" + std::to_string(rc) + ")"; return nullptr; }
                cur = dst.data(); cur_len = out_len; use_a = !use_a;
                break;
            }
            case 2: {  // shuffle: byte planes -> elements
                const size_t n = cur_len / (size_t)g.esz;
                dst.resize(cur_len);
                for (int b = 0; b < g.esz; ++b) {
                    const uint8_t* plane = cur + (size_t)b * n;
                    uint8_t* o = dst.data() + b;
                    for (size_t e = 0; e < n; ++e) o[e * g.esz] = plane[e];
                }
                std::memcpy(dst.data() + n * g.esz, cur + n * g.esz, cur_len - n * g.esz);
                cur = dst.data(); use_a = !use_a;
                break;
            }
            default:
                *err = "unsupported HDF5 filter id " + std::to_string(g.filters[i]);
                return nullptr;
        }
    }
    if ((int64_t)cur_len < g.chunk_bytes) { *err = "chunk holds fewer bytes than its dimensions"; return nullptr; }
    return cur;
}

// copy the part of a chunk that lies inside the dataset into dest (C order)
void place(const Geometry& g, const uint8_t* block, const int64_t* off, uint8_t* dest) {
    const int r = g.rank;
    int64_t ext[8];   // extent of the chunk inside the dataset per dimension
    for (int d = 0; d < r; ++d) {
        ext[d] = std::min<int64_t>(g.chunk[d], g.shape[d] - off[d]);
        if (ext[d] <= 0) return;
    }
    int64_t dstride[8], cstride[8];
    dstride[r - 1] = cstride[r - 1] = g.esz;
    for (int d = r - 2; d >= 0; --d) { dstride[d] = dstride[d + 1] * g.shape[d + 1]; cstride[d] = cstride[d + 1] * g.chunk[d + 1]; }
    const size_t run = (size_t)ext[r - 1] * g.esz;
    int64_t idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (true) {
        int64_t doff = 0, coff = 0;
        for (int d = 0; d < r - 1; ++d) { doff += (off[d] + idx[d]) * dstride[d]; coff += idx[d] * cstride[d]; }
        doff += off[r - 1] * dstride[r - 1];
        std::memcpy(dest + doff, block + coff, run);
        int d = r - 2;
        for (; d >= 0; --d) { if (++idx[d] < ext[d]) break; idx[d] = 0; }
        if (d < 0) break;
    }
}

bool walk(const Geometry& g, uint64_t addr, uint8_t* dest, Scratch& s, int64_t* chunks_done, std::string* err, int depth = 0) {
    if (depth > 16) { *err = "chunk B-tree too deep"; return false; }
    const int64_t a = g.base + (int64_t)addr;
    const int ksz = 8 + 8 * (g.rank + 1);
    if (a < 0 || a + 24 > g.file_len || std::memcmp(g.file + a, "TREE", 4) != 0) { *err = "bad chunk B-tree node"; return false; }
    const int ntype = g.file[a + 4], level = g.file[a + 5];
    const int used = rd<uint16_t>(g.file + a + 6);
    if (ntype != 1) { *err = "expected a raw-data chunk B-tree"; return false; }
    int64_t p = a + 24;
    if (p + (int64_t)used * (ksz + 8) + ksz > g.file_len) { *err = "chunk B-tree node runs past the end of the file"; return false; }
    for (int e = 0; e < used; ++e) {
        const uint32_t csize = rd<uint32_t>(g.file + p), mask = rd<uint32_t>(g.file + p + 4);
        int64_t off[8];
        for (int d = 0; d < g.rank; ++d) off[d] = (int64_t)rd<uint64_t>(g.file + p + 8 + 8 * d);
        const uint64_t child = rd<uint64_t>(g.file + p + ksz);
        p += ksz + 8;
        if (level > 0) {
            if (!walk(g, child, dest, s, chunks_done, err, depth + 1)) return false;
            continue;
        }
        const int64_t ca = g.base + (int64_t)child;
        if (ca < 0 || ca + (int64_t)csize > g.file_len) { *err = "chunk lies outside the file"; return false; }
        const uint8_t* block = unfilter(g, g.file + ca, csize, mask, s, err);
        if (!block) return false;
        for (int d = 0; d < g.rank; ++d)
            if (off[d] < 0 || off[d] >= g.shape[d]) { *err = "chunk offset outside the dataset"; return false; }
        place(g, block, off, dest);
        ++*chunks_done;
    }
    return true;
}

}  // namespace

extern "C" int th_h5_read_chunked(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs,
                                  void* const* dests, int rank, const int64_t* shape, const int64_t* chunk, int esz, int n_filters,
                                  const int* filter_ids, int nthreads) {
    if (!file || file_len <= 0 || n_datasets < 0 || (n_datasets && (!btree_addrs || !dests)) || !shape || !chunk)
        TH_FAIL(TH_EINVAL, "th_h5_read_chunked: null argument");
    if (rank < 1 || rank > 7 || esz < 1 || n_filters < 0 || n_filters > 8 || (n_filters && !filter_ids))
        TH_FAIL(TH_EUNSUP, "th_h5_read_chunked: rank %d / element size %d / %d filters not supported", rank, esz, n_filters);
    Geometry g;
    g.file = (const uint8_t*)file; g.file_len = file_len; g.base = base; g.rank = rank; g.esz = esz; g.n_filters = n_filters;
    g.chunk_bytes = esz;
    int64_t total_bytes = esz, n_chunks = 1;
    for (int d = 0; d < rank; ++d) {
        if (shape[d] <= 0 || chunk[d] <= 0) TH_FAIL(TH_EINVAL, "th_h5_read_chunked: bad dimensions");
        g.shape[d] = shape[d]; g.chunk[d] = chunk[d];
        g.chunk_bytes *= chunk[d]; total_bytes *= shape[d];
        n_chunks *= (shape[d] + chunk[d] - 1) / chunk[d];
    }
    for (int i = 0; i < n_filters; ++i) {
        g.filters[i] = filter_ids[i];
        if (filter_ids[i] < 1 || filter_ids[i] > 3) TH_FAIL(TH_EUNSUP, "th_h5_read_chunked: HDF5 filter %d not supported", filter_ids[i]);
    }
    if (n_datasets == 0) return TH_OK;
    unsigned hw = std::thread::hardware_concurrency();
    int nt = nthreads > 0 ? nthreads : (int)std::min<unsigned>(hw ? hw : 4, 128);
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_datasets));
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    std::vector<std::string> errs(nt);
    auto work = [&](int t) {
        Scratch s;
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_datasets || failed.load()) break;
            uint8_t* dest = (uint8_t*)dests[i];
            int64_t done = 0;
            bool ok = true;
            if ((uint64_t)btree_addrs[i] == kUndef) {
                std::memset(dest, 0, (size_t)total_bytes);   // nothing allocated: fill value
                continue;
            }
            ok = walk(g, (uint64_t)btree_addrs[i], dest, s, &done, &errs[t]);
            if (ok && done != n_chunks) {                      // unallocated chunks hold the fill value (0): redo over zeros
                std::memset(dest, 0, (size_t)total_bytes);
                done = 0;
                ok = walk(g, (uint64_t)btree_addrs[i], dest, s, &done, &errs[t]);
            }
            if (!ok) { failed.store(1); break; }
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    if (failed.load()) {
        for (auto& e : errs) if (!e.empty()) TH_FAIL(TH_EIO, "th_h5_read_chunked: %s", e.c_str());
        TH_FAIL(TH_EIO, "th_h5_read_chunked: failed");
    }
    return TH_OK;
}
