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
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <cstdio>

int th_usable_cpus() {
    static const int cached = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 4;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, a); }
        // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us
        long long quota = -1, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = "";
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else {
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        }
        if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        return std::max(1, n);
    }();
    return cached;
}

extern "C" int th_host_cpus(void) { return th_usable_cpus(); }

namespace {

constexpr uint64_t kUndef = 0xFFFFFFFFFFFFFFFFull;

struct Geometry {
    const uint8_t* file; int64_t file_len, base;
    int rank;                       // real dimensions (the B-tree keys carry rank + 1 offsets)
    int64_t shape[8]; int64_t chunk[8]; int esz;
    int n_filters; int filters[8];
    int64_t chunk_bytes;
    int conv = 0;                   // 0: bytes as stored; 1: float64 -> float32 while placing (what Keras' own cast does)
    int out_esz = 0;                // element size in the destination
};

template <typename T> inline T rd(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof v); return v; }

struct Scratch { std::vector<uint8_t> a, b; };

// The checksum of HDF5's fletcher32 filter: two sums over big-endian 16-bit words, folded with end-around carry every 360 words
// (so that 32-bit accumulators cannot overflow), an odd last byte counted as the high byte of a word.  `swapped`: the variant over
// byte-swapped words that libraries before 1.6.3 wrote — the library accepts either when it reads.
uint32_t h5_fletcher32(const uint8_t* data, size_t nbytes, bool swapped) {
    size_t len = nbytes / 2;
    uint32_t sum1 = 0, sum2 = 0;
    const int hi = swapped ? 1 : 0, lo = swapped ? 0 : 1;
    while (len) {
        size_t tlen = len > 360 ? 360 : len;
        len -= tlen;
        do {
            sum1 += ((uint32_t)data[hi] << 8) | (uint32_t)data[lo];
            data += 2;
            sum2 += sum1;
        } while (--tlen);
        sum1 = (sum1 & 0xffff) + (sum1 >> 16);
        sum2 = (sum2 & 0xffff) + (sum2 >> 16);
    }
    if (nbytes % 2) {
        sum1 += (uint32_t)*data << 8;
        sum2 += sum1;
        sum1 = (sum1 & 0xffff) + (sum1 >> 16);
        sum2 = (sum2 & 0xffff) + (sum2 >> 16);
    }
    sum1 = (sum1 & 0xffff) + (sum1 >> 16);
    sum2 = (sum2 & 0xffff) + (sum2 >> 16);
    return (sum2 << 16) | sum1;
}

// undo the filter pipeline of one chunk; returns a pointer to chunk_bytes of data or nullptr
const uint8_t* unfilter(const Geometry& g, const uint8_t* src, size_t len, uint32_t mask, Scratch& s, std::string* err) {
    const uint8_t* cur = src;
    size_t cur_len = len;
    bool use_a = true;
    for (int i = g.n_filters - 1; i >= 0; --i) {
        if (mask & (1u << i)) continue;
        std::vector<uint8_t>& dst = use_a ? s.a : s.b;
        switch (g.filters[i]) {
            case 3: {  // fletcher32: checksum appended (little-endian); HDF5 fails the read when it does not match
                if (cur_len < 4) { *err = "fletcher32 chunk shorter than its checksum"; return nullptr; }
                cur_len -= 4;
                const uint32_t stored = rd<uint32_t>(cur + cur_len);
                if (stored != h5_fletcher32(cur, cur_len, false) && stored != h5_fletcher32(cur, cur_len, true)) {
                    *err = "fletcher32 checksum of a chunk does not match its data";
                    return nullptr;
                }
                break;
            }
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

// copy the part of a chunk that lies inside the dataset into dest (C order), converting on the way when asked
void place(const Geometry& g, const uint8_t* block, const int64_t* off, uint8_t* dest) {
    const int r = g.rank;
    int64_t ext[8];   // extent of the chunk inside the dataset per dimension
    for (int d = 0; d < r; ++d) {
        ext[d] = std::min<int64_t>(g.chunk[d], g.shape[d] - off[d]);
        if (ext[d] <= 0) return;
    }
    int64_t dstride[8], cstride[8];
    dstride[r - 1] = g.out_esz;
    cstride[r - 1] = g.esz;
    for (int d = r - 2; d >= 0; --d) { dstride[d] = dstride[d + 1] * g.shape[d + 1]; cstride[d] = cstride[d + 1] * g.chunk[d + 1]; }
    const size_t run = (size_t)ext[r - 1];
    int64_t idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (true) {
        int64_t doff = 0, coff = 0;
        for (int d = 0; d < r - 1; ++d) { doff += (off[d] + idx[d]) * dstride[d]; coff += idx[d] * cstride[d]; }
        doff += off[r - 1] * dstride[r - 1];
        if (g.conv == 1) {
            float* o = reinterpret_cast<float*>(dest + doff);
            const uint8_t* in = block + coff;
            for (size_t e = 0; e < run; ++e) { double v; std::memcpy(&v, in + 8 * e, 8); o[e] = (float)v; }   // round to nearest even
        } else {
            std::memcpy(dest + doff, block + coff, run * g.esz);
        }
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

extern "C" int th_h5_read_chunked_as(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs,
                                     void* const* dests, int rank, const int64_t* shape, const int64_t* chunk, int esz, int n_filters,
                                     const int* filter_ids, int nthreads, int conv) {
    if (!file || file_len <= 0 || n_datasets < 0 || (n_datasets && (!btree_addrs || !dests)) || !shape || !chunk)
        TH_FAIL(TH_EINVAL, "th_h5_read_chunked: null argument");
    if (rank < 1 || rank > 7 || esz < 1 || n_filters < 0 || n_filters > 8 || (n_filters && !filter_ids))
        TH_FAIL(TH_EUNSUP, "th_h5_read_chunked: rank %d / element size %d / %d filters not supported", rank, esz, n_filters);
    if (conv != 0 && !(conv == 1 && esz == 8)) TH_FAIL(TH_EINVAL, "th_h5_read_chunked_as: conversion %d needs 8-byte (float64) elements", conv);
    Geometry g;
    g.file = (const uint8_t*)file; g.file_len = file_len; g.base = base; g.rank = rank; g.esz = esz; g.n_filters = n_filters;
    g.conv = conv; g.out_esz = conv == 1 ? 4 : esz;
    g.chunk_bytes = esz;
    int64_t total_bytes = g.out_esz, n_chunks = 1;
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
    const int hw = th_usable_cpus();
    int nt = nthreads > 0 ? nthreads : std::min(hw, 128);
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

// ---- the same datasets decoded ON THE DEVICE --------------------------------------------------------------------------
// Only the chunk B-trees are walked on the host (a few hundred bytes per dataset); the compressed chunk bytes go to the GPU as
// they lie in the file and are inflated there, one lane per chunk (csrc/inflate.hip), then placed into d_out
// [n_datasets][shape...] — float32 when conv = 1 (float64 data), the stored type otherwise.  Supported: the filter pipeline
// aposteriori writes (deflate alone), every chunk compressed (filter mask 0).  Anything else -> TH_EUNSUP and the caller uses
// the host reader.  Chunks that were never allocated read as zeros.
namespace {
struct ChunkRec { int ds; int64_t off; uint32_t csize; int coff[8]; };
bool collect(const Geometry& g, uint64_t addr, int ds, std::vector<ChunkRec>* out, std::string* err, bool* unsup, int depth = 0) {
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
        ChunkRec r;
        r.ds = ds;
        r.csize = rd<uint32_t>(g.file + p);
        const uint32_t mask = rd<uint32_t>(g.file + p + 4);
        for (int d = 0; d < 8; ++d) r.coff[d] = 0;
        for (int d = 0; d < g.rank; ++d) {
            const int64_t o = (int64_t)rd<uint64_t>(g.file + p + 8 + 8 * d);
            if (o < 0 || o >= g.shape[d]) { *err = "chunk offset outside the dataset"; return false; }
            if (o % g.chunk[d] != 0) { *err = "chunk offset is not a multiple of the chunk size"; return false; }
            r.coff[d] = (int)o;
        }
        const uint64_t child = rd<uint64_t>(g.file + p + ksz);
        p += ksz + 8;
        if (level > 0) {
            if (!collect(g, child, ds, out, err, unsup, depth + 1)) return false;
            continue;
        }
        if (mask != 0) { *unsup = true; return true; }          // a chunk stored without (some of) its filters
        r.off = g.base + (int64_t)child;
        if (r.off < 0 || r.off > g.file_len - (int64_t)r.csize) { *err = "chunk lies outside the file"; return false; }
        out->push_back(r);
    }
    return true;
}
}  // namespace

int inflate_place_device(int device, hipStream_t stream, const void* span, int64_t span_len, int64_t n_chunks, const int64_t* src_off,
                         const int64_t* csize, const int* ds, const int* coff8, int rank, const int64_t* shape, const int64_t* chunk, int esz,
                         int conv, void* d_out, int64_t* n_bad, int shuffle);

extern "C" int th_h5_decode_device(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs, int rank,
                                   const int64_t* shape, const int64_t* chunk, int esz, int n_filters, const int* filter_ids, int conv,
                                   int device, void* d_out) {
    if (!file || file_len <= 0 || n_datasets < 0 || (n_datasets && (!btree_addrs || !d_out)) || !shape || !chunk)
        TH_FAIL(TH_EINVAL, "th_h5_decode_device: null argument");
    if (rank < 1 || rank > 7 || esz < 1) TH_FAIL(TH_EUNSUP, "th_h5_decode_device: rank %d / element size %d not supported", rank, esz);
    // pipelines decoded on the device: deflate, or shuffle + deflate (what h5py writes for compression="gzip"[, shuffle=True])
    const bool plain = n_filters == 1 && filter_ids && filter_ids[0] == 1;
    const bool shuffled = n_filters == 2 && filter_ids && filter_ids[0] == 2 && filter_ids[1] == 1;
    if (!plain && !shuffled) TH_FAIL(TH_EUNSUP, "th_h5_decode_device: only deflate and shuffle + deflate pipelines are decoded on the device");
    if (conv != 0 && !(conv == 1 && esz == 8)) TH_FAIL(TH_EINVAL, "th_h5_decode_device: conversion %d needs float64 elements", conv);
    if (n_datasets == 0) return TH_OK;
    static const bool trace = getenv("TH_H5_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    Geometry g;
    g.file = (const uint8_t*)file; g.file_len = file_len; g.base = base; g.rank = rank; g.esz = esz; g.n_filters = n_filters;
    int64_t elems = 1;
    for (int d = 0; d < rank; ++d) {
        if (shape[d] <= 0 || chunk[d] <= 0 || shape[d] > 0x7fffffff || chunk[d] > 0x7fffffff) TH_FAIL(TH_EINVAL, "th_h5_decode_device: bad dimensions");
        g.shape[d] = shape[d]; g.chunk[d] = chunk[d];
        elems *= shape[d];
    }
    // B-trees on host threads
    const int hw = th_usable_cpus();
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(hw, 32), n_datasets / 64 + 1));
    std::vector<std::vector<ChunkRec>> parts(nt);
    std::vector<std::string> errs(nt);
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0}, unsupported{0};
    auto work = [&](int t) {
        for (;;) {
            const int64_t lo = next.fetch_add(64);
            if (lo >= n_datasets || failed.load() || unsupported.load()) break;
            for (int64_t i = lo; i < std::min<int64_t>(lo + 64, n_datasets); ++i) {
                if ((uint64_t)btree_addrs[i] == kUndef) continue;     // never written: zeros
                bool unsup = false;
                if (!collect(g, (uint64_t)btree_addrs[i], (int)i, &parts[t], &errs[t], &unsup)) { failed.store(1); return; }
                if (unsup) { unsupported.store(1); return; }
            }
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    if (failed.load()) {
        for (auto& e : errs) if (!e.empty()) TH_FAIL(TH_EIO, "th_h5_decode_device: %s", e.c_str());
        TH_FAIL(TH_EIO, "th_h5_decode_device: failed");
    }
    if (unsupported.load()) TH_FAIL(TH_EUNSUP, "th_h5_decode_device: a chunk is stored without its filters");
    size_t nch = 0;
    for (auto& v : parts) nch += v.size();
    const double t_walk = since();
    HIP_TRY(hipSetDevice(device));
    const int out_esz = conv == 1 ? 4 : esz;
    // chunks that were never written read as zeros; when every chunk of every dataset is there, the placement writes every
    // element and the 0.9 GB memset of a 4 096-frame batch is skipped
    int64_t chunks_per_dataset = 1;
    for (int d = 0; d < rank; ++d) chunks_per_dataset *= (shape[d] + chunk[d] - 1) / chunk[d];
    // (a malformed tree may list one chunk twice and another not at all: the count alone does not prove coverage — every
    // (dataset, chunk index) must be there exactly once; offsets are chunk-aligned, collect() checked that)
    bool covered = (int64_t)nch == n_datasets * chunks_per_dataset;
    if (covered) {
        std::vector<bool> seen(nch, false);
        for (auto& v : parts)
            for (const ChunkRec& r : v) {
                int64_t lin = 0;
                for (int d = 0; d < rank; ++d) lin = lin * ((shape[d] + chunk[d] - 1) / chunk[d]) + r.coff[d] / chunk[d];
                const size_t at = (size_t)((int64_t)r.ds * chunks_per_dataset + lin);
                if (seen[at]) covered = false;
                seen[at] = true;
            }
    }
    if (!covered) {
        HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)n_datasets * elems * out_esz, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    if (nch == 0) return TH_OK;
    std::vector<int64_t> src_off(nch), csize(nch);
    std::vector<int> ds(nch), coff(nch * 8);
    int64_t lo = INT64_MAX, hi = 0, total = 0;
    size_t k = 0;
    for (auto& v : parts)
        for (const ChunkRec& r : v) {
            src_off[k] = r.off; csize[k] = r.csize; ds[k] = r.ds;
            for (int d = 0; d < 8; ++d) coff[k * 8 + d] = r.coff[d];
            lo = std::min(lo, r.off); hi = std::max(hi, r.off + (int64_t)r.csize); total += r.csize;
            ++k;
        }
    // the chunks of consecutive residues lie (almost) back to back in the file: ship the byte span as it is; a scattered
    // selection is gathered first
    const uint8_t* span = g.file + lo;
    int64_t span_len = hi - lo;
    std::vector<uint8_t> gathered;
    if (span_len > 2 * total + (1 << 20)) {
        gathered.resize((size_t)total + 8);
        int64_t pos = 0;
        for (size_t i = 0; i < nch; ++i) {
            std::memcpy(gathered.data() + pos, g.file + src_off[i], (size_t)csize[i]);
            src_off[i] = pos;
            pos += csize[i];
        }
        span = gathered.data(); span_len = total;
    } else {
        for (size_t i = 0; i < nch; ++i) src_off[i] -= lo;
    }
    const double t_prep = since();
    int64_t bad = 0;
    int rc = inflate_place_device(device, nullptr, span, span_len, (int64_t)nch, src_off.data(), csize.data(), ds.data(), coff.data(), rank,
                                  shape, chunk, esz, conv, d_out, &bad, shuffled ? 1 : 0);
    if (trace)
        fprintf(stderr, "[h5 decode] %lld datasets, %zu chunks, span %.1f MB (%s): B-trees %.2f ms, memset + descriptors %.2f ms, copy + kernels %.2f ms\n",
                (long long)n_datasets, nch, span_len / 1e6, gathered.empty() ? "direct" : "gathered", t_walk, t_prep - t_walk, since() - t_prep);
    if (rc) return rc;
    if (bad) TH_FAIL(TH_EIO, "th_h5_decode_device: %lld of %zu chunks did not inflate", (long long)bad, nch);
    parts.clear(); parts.shrink_to_fit();
    if (trace) fprintf(stderr, "[h5 decode] returning after %.2f ms\n", since());
    return TH_OK;
}

extern "C" int th_h5_read_chunked(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs,
                                  void* const* dests, int rank, const int64_t* shape, const int64_t* chunk, int esz, int n_filters,
                                  const int* filter_ids, int nthreads) {
    return th_h5_read_chunked_as(file, file_len, base, n_datasets, btree_addrs, dests, rank, shape, chunk, esz, n_filters, filter_ids,
                                 nthreads, 0);
}

// ---- object-header resolution for MANY datasets ------------------------------------------------------------------
// What the per-residue Python work of load_batch / create_flat_dataset_map was: parse the dataset's object header
// (version 1 or 2), take dataspace / datatype / layout / filter pipeline, and read two small attributes —
// `encoded_residue` (the one-hot label row, reference utils.py:529) and `label` (the three-letter residue name,
// reference utils.py:375).  Everything unusual (shared messages, dense attribute storage, compound types ...) just
// leaves the corresponding status bit clear and the caller uses its general (Python) reader for that one dataset.
namespace {

struct Msg { int type; int flags; const uint8_t* data; int64_t size; };

bool header_messages(const uint8_t* f, int64_t flen, int64_t base, uint64_t addr, std::vector<Msg>* out) {
    out->clear();
    const int64_t a = base + (int64_t)addr;
    if (a < 0 || a + 16 > flen) return false;
    struct Block { int64_t p, len; };
    std::vector<Block> blocks;
    if (std::memcmp(f + a, "OHDR", 4) == 0) {           // version 2
        if (f[a + 4] != 2) return false;
        const int flags = f[a + 5];
        int64_t p = a + 6;
        if (flags & 0x20) p += 16;
        if (flags & 0x10) p += 4;
        const int szbytes = 1 << (flags & 3);
        if (p + szbytes > flen) return false;          // prefix (times, phase-change values, chunk-0 size) inside the file
        uint64_t chunk0 = 0;
        std::memcpy(&chunk0, f + p, szbytes);
        if (chunk0 > (uint64_t)flen) return false;
        p += szbytes;
        const bool track = flags & 4;
        blocks.push_back({p, (int64_t)chunk0});
        for (size_t b = 0; b < blocks.size() && b < 64; ++b) {
            int64_t q = blocks[b].p;
            if (q < 0 || blocks[b].len < 0 || blocks[b].len > flen || q > flen - blocks[b].len) return false;
            const int64_t end = q + blocks[b].len;
            while (q + 4 + (track ? 2 : 0) <= end) {
                const int mtype = f[q], msize = rd<uint16_t>(f + q + 1), mflags = f[q + 3];
                q += 4 + (track ? 2 : 0);
                if (q + msize > end) break;
                if (mtype == 0x10) {
                    if (msize < 16) return false;
                    const uint64_t co = rd<uint64_t>(f + q), cl = rd<uint64_t>(f + q + 8);
                    if (co > (uint64_t)flen || cl > (uint64_t)flen || cl < 8) return false;
                    blocks.push_back({base + (int64_t)co + 4, (int64_t)cl - 8});
                } else if (mtype != 0) {
                    out->push_back({mtype, mflags, f + q, msize});
                }
                q += msize;
            }
        }
        return true;
    }
    if (f[a] != 1) return false;
    const uint32_t hsize = rd<uint32_t>(f + a + 8);
    blocks.push_back({a + 16, (int64_t)hsize});
    for (size_t b = 0; b < blocks.size() && b < 64; ++b) {
        int64_t q = blocks[b].p;
        if (q < 0 || blocks[b].len < 0 || blocks[b].len > flen || q > flen - blocks[b].len) return false;
        const int64_t end = q + blocks[b].len;
        while (q + 8 <= end) {
            const int mtype = rd<uint16_t>(f + q), msize = rd<uint16_t>(f + q + 2), mflags = f[q + 4];
            q += 8;
            if (q + msize > end) break;
            if (mtype == 0x10) {
                if (msize < 16) return false;
                const uint64_t co = rd<uint64_t>(f + q), cl = rd<uint64_t>(f + q + 8);
                if (co > (uint64_t)flen || cl > (uint64_t)flen) return false;
                blocks.push_back({base + (int64_t)co, (int64_t)cl});
            } else if (mtype != 0) {
                out->push_back({mtype, mflags, f + q, msize});
            }
            q += msize;
        }
    }
    return true;
}

// number of elements of a (simple) dataspace message, or -1
int64_t dataspace_count(const uint8_t* d, int64_t n, int64_t* dims, int* rank_out) {
    if (n < 4) return -1;
    const int ver = d[0], rank = d[1];
    int64_t p;
    if (ver == 1) p = 8;
    else if (ver == 2) { if (d[3] == 2) return -1; p = 4; }
    else return -1;
    if (rank > 7 || p + 8 * rank > n) return -1;
    int64_t cnt = 1;
    for (int i = 0; i < rank; ++i) {
        const uint64_t v = rd<uint64_t>(d + p + 8 * i);
        if (v > (uint64_t)1 << 40) return -1;                          // hostile / unlimited dimension
        if (dims) dims[i] = (int64_t)v;
        if (v && cnt > ((int64_t)1 << 56) / (int64_t)v) return -1;     // product would overflow
        cnt *= (int64_t)v;
    }
    if (rank_out) *rank_out = rank;
    return cnt;
}

struct SimpleType { int cls = -1; int size = 0; bool is_signed = false; bool big_endian = false; bool vlen_str = false; };
bool simple_type(const uint8_t* d, int64_t n, SimpleType* t) {
    if (n < 8) return false;
    t->cls = d[0] & 0x0F;
    t->size = (int)rd<uint32_t>(d + 4);
    t->big_endian = d[1] & 1;
    if (t->cls == 0) { t->is_signed = d[1] & 8; return true; }
    if (t->cls == 1 || t->cls == 3) return true;
    if (t->cls == 9) { t->vlen_str = (d[1] & 0x0F) == 1; return t->vlen_str; }
    return false;
}

// an attribute message -> (name, type, element count, raw data)
bool parse_attribute(const Msg& m, std::string* name, SimpleType* t, int64_t* count, const uint8_t** raw, int64_t* raw_len) {
    const uint8_t* d = m.data;
    if (m.size < 8 || (m.flags & 2)) return false;       // shared message: not handled here
    const int ver = d[0];
    const int nsz = rd<uint16_t>(d + 2), dsz = rd<uint16_t>(d + 4), ssz = rd<uint16_t>(d + 6);
    int64_t p;
    auto pad = [&](int x) { return ver == 1 ? (x + 7) / 8 * 8 : x; };
    if (ver == 1) p = 8;
    else if (ver == 2) { if (d[1] & 3) return false; p = 8; }
    else if (ver == 3) { if (d[1] & 3) return false; p = 9; }
    else return false;
    if (p + pad(nsz) + pad(dsz) + pad(ssz) > m.size) return false;
    name->assign((const char*)d + p, strnlen((const char*)d + p, nsz));
    p += pad(nsz);
    if (!simple_type(d + p, dsz, t)) return false;
    p += pad(dsz);
    *count = dataspace_count(d + p, ssz, nullptr, nullptr);
    p += pad(ssz);
    if (*count < 0) return false;
    *raw = d + p;
    *raw_len = m.size - p;
    return true;
}

bool global_heap_object(const uint8_t* f, int64_t flen, int64_t base, uint64_t addr, uint32_t index, const uint8_t** obj, uint64_t* len) {
    const int64_t a = base + (int64_t)addr;
    if (a < 0 || a + 16 > flen || std::memcmp(f + a, "GCOL", 4) != 0) return false;
    const int64_t size = (int64_t)rd<uint64_t>(f + a + 8);
    int64_t p = a + 16;
    const int64_t end = std::min<int64_t>(a + size, flen);
    while (p + 16 <= end) {
        const int idx = rd<uint16_t>(f + p);
        const uint64_t osz = rd<uint64_t>(f + p + 8);
        if (idx == 0) break;
        if ((uint32_t)idx == index) { if (p + 16 + (int64_t)osz > end) return false; *obj = f + p + 16; *len = osz; return true; }
        p += 16 + (int64_t)((osz + 7) / 8 * 8);
    }
    return false;
}

}  // namespace

// geom_out (int64[40]): [0] rank, [1..7] shape, [8..14] chunk, [15] element size, [16] datatype class (0 int, 1 float, 8 enum),
//   [17] signed, [18] n_filters, [19..26] filter ids, [27] layout class (1 contiguous: btree_out is the data address;
//   2 chunked) — of the FIRST dataset; status bit 0 of dataset i says "chunked v3
//   storage with exactly this geometry" (so one th_h5_read_chunked call serves them all).
// status bits: 1 header parsed & geometry as geom_out, 2 numeric attribute filled, 4 string attribute filled.
extern "C" int th_h5_resolve(const void* file, int64_t file_len, int64_t base, int64_t n, const int64_t* ohdr_addrs,
                             const char* num_attr, double* num_out, int num_len, const char* str_attr, char* str_out, int str_len,
                             int64_t* btree_out, int64_t* geom_out, int* status_out, int nthreads) {
    if (!file || file_len <= 0 || n < 0 || (n && (!ohdr_addrs || !btree_out || !geom_out || !status_out)))
        TH_FAIL(TH_EINVAL, "th_h5_resolve: null argument");
    if ((num_attr && (!num_out || num_len <= 0)) || (str_attr && (!str_out || str_len <= 1))) TH_FAIL(TH_EINVAL, "th_h5_resolve: attribute buffers");
    const uint8_t* f = (const uint8_t*)file;
    for (int i = 0; i < 40; ++i) geom_out[i] = 0;
    struct Geo { int64_t v[40]; bool ok = false; uint64_t btree = kUndef; };
    auto one = [&](int64_t i, Geo* g, std::vector<Msg>& msgs) -> int {
        int status = 0;
        *g = Geo();
        if (!header_messages(f, file_len, base, (uint64_t)ohdr_addrs[i], &msgs)) return 0;
        const Msg *space = nullptr, *type = nullptr, *layout = nullptr, *filt = nullptr;
        bool shared = false;
        for (const Msg& m : msgs) {
            if (m.type == 0x01) space = &m; else if (m.type == 0x03) type = &m; else if (m.type == 0x08) layout = &m;
            else if (m.type == 0x0B) filt = &m;
            if ((m.type == 0x01 || m.type == 0x03 || m.type == 0x08 || m.type == 0x0B) && (m.flags & 2)) shared = true;
        }
        // a user-defined NON-ZERO fill value (message 0x05, or the old 0x04) changes what never-written elements read as;
        // this reader zero-fills, so such a dataset is left to the general reader
        bool nonzero_fill = false;
        for (const Msg& m : msgs) {
            const uint8_t* d = m.data;
            const uint8_t* v = nullptr;
            uint32_t sz = 0;
            if (m.type == 0x04 && m.size >= 4) { sz = rd<uint32_t>(d); v = d + 4; if (4 + (int64_t)sz > m.size) sz = 0; }
            else if (m.type == 0x05 && m.size >= 2) {
                if ((d[0] == 1 || (d[0] == 2 && m.size >= 4 && d[3])) && m.size >= 8) { sz = rd<uint32_t>(d + 4); v = d + 8; if (8 + (int64_t)sz > m.size) sz = 0; }
                else if (d[0] == 3 && (d[1] & 0x20) && m.size >= 6) { sz = rd<uint32_t>(d + 2); v = d + 6; if (6 + (int64_t)sz > m.size) sz = 0; }
            }
            for (uint32_t k = 0; v && k < sz; ++k) if (v[k]) nonzero_fill = true;
        }
        if (space && type && layout && !shared && !nonzero_fill && type->size >= 8 && layout->size >= 2 &&
            (!filt || filt->size >= 2)) {
            int rank = 0;
            int64_t dims[8];
            const int64_t cnt = dataspace_count(space->data, space->size, dims, &rank);
            const uint8_t* t = type->data;
            const uint8_t* l = layout->data;
            int cls = t[0] & 0x0F;
            int esz = (int)rd<uint32_t>(t + 4);
            if (esz < 1 || esz > 16) cls = -1;                          // numeric element sizes only
            bool big = t[1] & 1, sgn = (cls == 0) && (t[1] & 8);
            if (cls == 8 && type->size >= 16) { const uint8_t* b = t + 8; big = b[1] & 1; if ((b[0] & 0x0F) != 0) cls = -1; }   // enum over an integer (h5py bool)
            if (cnt > 0 && rank >= 1 && (cls == 0 || cls == 1 || cls == 8) && !big && layout->size >= 18 && l[0] == 3 && l[1] == 1) {
                // contiguous storage: "btree" carries the data address, geometry slot 27 says so
                g->v[0] = rank;
                for (int d = 0; d < rank; ++d) g->v[1 + d] = dims[d];
                g->v[15] = esz; g->v[16] = cls; g->v[17] = sgn; g->v[27] = 1;
                const uint64_t daddr = rd<uint64_t>(l + 2), dsize = rd<uint64_t>(l + 10);
                // cnt <= file_len / esz first: cnt * esz and the end address cannot overflow after that
                const bool fits = cnt <= file_len / esz && daddr <= (uint64_t)file_len && base >= 0 && base <= file_len &&
                                  (int64_t)daddr <= file_len - base - cnt * esz && dsize >= (uint64_t)(cnt * esz);
                if (!filt && (daddr == kUndef || fits)) {
                    g->ok = true; g->btree = daddr;
                }
            } else if (cnt > 0 && rank >= 1 && (cls == 0 || cls == 1 || cls == 8) && !big && layout->size >= 11 && l[0] == 3 && l[1] == 2 &&
                l[2] == rank + 1 && layout->size >= 11 + 4 * (rank + 1)) {
                g->v[27] = 2;
                g->v[0] = rank;
                for (int d = 0; d < rank; ++d) { g->v[1 + d] = dims[d]; g->v[8 + d] = rd<uint32_t>(l + 11 + 4 * d); }
                g->v[15] = esz; g->v[16] = cls; g->v[17] = sgn;
                bool fok = true;
                int nf = 0;
                if (filt) {
                    const uint8_t* d = filt->data;
                    const int ver = d[0];
                    nf = d[1];
                    int64_t p = ver == 1 ? 8 : 2;
                    if (nf > 8) fok = false;
                    for (int k = 0; k < nf && fok; ++k) {
                        if (p + 8 > filt->size) { fok = false; break; }
                        const int fid = rd<uint16_t>(d + p);
                        int ncd;
                        if (ver != 1 && ver != 2) { fok = false; break; }
                        if (ver == 1 || fid >= 256) {
                            const int nlen = rd<uint16_t>(d + p + 2);
                            ncd = rd<uint16_t>(d + p + 6);
                            p += 8 + (ver == 1 ? (nlen + 7) / 8 * 8 : nlen);
                        } else {
                            ncd = rd<uint16_t>(d + p + 4);
                            p += 6;
                        }
                        p += 4 * ncd + ((ver == 1 && (ncd & 1)) ? 4 : 0);
                        g->v[19 + k] = fid;
                    }
                }
                g->v[18] = nf;
                if (fok && (int)rd<uint32_t>(l + 11 + 4 * rank) == esz) { g->ok = true; g->btree = rd<uint64_t>(l + 3); }
            }
        }
        for (const Msg& m : msgs) {
            if (m.type != 0x0C) continue;
            std::string name; SimpleType t; int64_t cnt; const uint8_t* raw; int64_t raw_len;
            if (!parse_attribute(m, &name, &t, &cnt, &raw, &raw_len)) continue;
            if (num_attr && name == num_attr && cnt == num_len && !t.big_endian && (t.cls == 0 || t.cls == 1) &&
                raw_len >= cnt * t.size) {
                double* o = num_out + i * num_len;
                bool ok = true;
                for (int64_t e = 0; e < cnt && ok; ++e) {
                    const uint8_t* q = raw + e * t.size;
                    if (t.cls == 1 && t.size == 8) o[e] = rd<double>(q);
                    else if (t.cls == 1 && t.size == 4) o[e] = rd<float>(q);
                    else if (t.cls == 0 && t.size == 1) o[e] = t.is_signed ? (double)rd<int8_t>(q) : (double)rd<uint8_t>(q);
                    else if (t.cls == 0 && t.size == 2) o[e] = t.is_signed ? (double)rd<int16_t>(q) : (double)rd<uint16_t>(q);
                    else if (t.cls == 0 && t.size == 4) o[e] = t.is_signed ? (double)rd<int32_t>(q) : (double)rd<uint32_t>(q);
                    else if (t.cls == 0 && t.size == 8) o[e] = t.is_signed ? (double)rd<int64_t>(q) : (double)rd<uint64_t>(q);
                    else ok = false;
                }
                if (ok) status |= 2;
            } else if (str_attr && name == str_attr && cnt == 1) {
                char* o = str_out + i * str_len;
                std::memset(o, 0, str_len);
                if (t.cls == 3 && raw_len >= t.size) {
                    const size_t ln = std::min<size_t>(strnlen((const char*)raw, t.size), str_len - 1);
                    std::memcpy(o, raw, ln);
                    // space padding (str_pad 2) is stripped by the caller together with NULs
                    status |= 4;
                } else if (t.cls == 9 && t.vlen_str && raw_len >= 16) {
                    const uint32_t ln = rd<uint32_t>(raw);
                    const uint64_t haddr = rd<uint64_t>(raw + 4);
                    const uint32_t hidx = rd<uint32_t>(raw + 12);
                    const uint8_t* obj; uint64_t olen;
                    if (ln == 0 && haddr == 0) status |= 4;
                    else if (global_heap_object(f, file_len, base, haddr, hidx, &obj, &olen)) {
                        const size_t c = std::min<size_t>(std::min<uint64_t>(ln, olen), str_len - 1);
                        std::memcpy(o, obj, c);
                        status |= 4;
                    }
                }
            }
        }
        return status;
    };
    if (n == 0) return TH_OK;
    // dataset 0 defines the geometry the rest is compared with
    Geo g0;
    {
        std::vector<Msg> msgs;
        status_out[0] = one(0, &g0, msgs);
        if (g0.ok) { for (int k = 0; k < 40; ++k) geom_out[k] = g0.v[k]; status_out[0] |= 1; }
        btree_out[0] = g0.ok ? (g0.btree == kUndef ? -1 : (int64_t)g0.btree) : -1;
    }
    const int hw = th_usable_cpus();
    int nt = nthreads > 0 ? nthreads : std::min(hw, 16);
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n - 1) / 256 + 1));
    std::atomic<int64_t> next{1};
    auto work = [&]() {
        std::vector<Msg> msgs;
        Geo g;
        for (;;) {
            const int64_t lo = next.fetch_add(64);
            if (lo >= n) break;
            for (int64_t i = lo; i < std::min<int64_t>(lo + 64, n); ++i) {
                int st = one(i, &g, msgs);
                const bool same = g0.ok && g.ok && std::memcmp(g.v, g0.v, sizeof g.v) == 0;
                if (same) st |= 1;
                btree_out[i] = same ? (g.btree == kUndef ? -1 : (int64_t)g.btree) : -1;
                status_out[i] = st;
            }
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    return TH_OK;
}

// contiguous datasets (layout class 1): data_addrs[i] is the address of count elements of esz bytes (or -1: never
// written, reads as zeros); same conversion option as th_h5_read_chunked_as
extern "C" int th_h5_read_contiguous_as(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* data_addrs,
                                        void* const* dests, int64_t count, int esz, int conv) {
    if (!file || file_len <= 0 || n_datasets < 0 || (n_datasets && (!data_addrs || !dests)) || count < 0 || esz < 1)
        TH_FAIL(TH_EINVAL, "th_h5_read_contiguous_as: bad argument");
    if (conv != 0 && !(conv == 1 && esz == 8)) TH_FAIL(TH_EINVAL, "th_h5_read_contiguous_as: conversion %d needs float64 elements", conv);
    const uint8_t* f = (const uint8_t*)file;
    const int out_esz = conv == 1 ? 4 : esz;
    for (int64_t i = 0; i < n_datasets; ++i) {
        uint8_t* dest = (uint8_t*)dests[i];
        if (data_addrs[i] < 0) { std::memset(dest, 0, (size_t)(count * out_esz)); continue; }
        const int64_t a = base + data_addrs[i];
        if (a < 0 || a > file_len || count > (file_len - a) / esz) TH_FAIL(TH_EIO, "th_h5_read_contiguous_as: dataset %lld lies outside the file", (long long)i);
        if (conv == 1) {
            float* o = reinterpret_cast<float*>(dest);
            for (int64_t e = 0; e < count; ++e) { double v; std::memcpy(&v, f + a + 8 * e, 8); o[e] = (float)v; }
        } else {
            std::memcpy(dest, f + a, (size_t)(count * esz));
        }
    }
    return TH_OK;
}

// ---- group listing ---------------------------------------------------------------------------------------------------
// Every link of one old-style group (symbol-table message: a version-1 group B-tree over SNOD nodes, names in a local heap)
// in one call: create_flat_dataset_map (reference utils.py:357-375) lists a pdb group, its chain groups and ~100 residue
// names per chain; walking 40 k symbol-table entries in the interpreter costs 1.4 us each.  `heap_data` / `heap_size` are the
// ABSOLUTE offset and length of the local heap's data segment.  names: the link names, each NUL-terminated, in B-tree order;
// addrs[i]: the object-header address of link i.  TH_EINVAL for anything that is not a well-formed group B-tree inside the
// file (the caller's Python walk then raises its own format error), TH_ENOMEM when a capacity is too small.
namespace {
int walk_group_node(const uint8_t* f, int64_t len, int64_t base, uint64_t addr, int64_t heap_data, int64_t heap_size, char* names,
                    int64_t names_cap, int64_t* addrs, int64_t addrs_cap, int64_t* n, int64_t* used_names, int depth, int64_t* budget) {
    if (depth > 32 || --*budget < 0) return TH_EINVAL;
    if (addr == kUndef || addr > (uint64_t)len || base < 0 || base > len) return TH_EINVAL;
    const int64_t a = base + (int64_t)addr;
    if (a < 0 || a > len - 24 || std::memcmp(f + a, "TREE", 4) != 0 || f[a + 4] != 0) return TH_EINVAL;
    const int level = f[a + 5];
    const int entries = rd<uint16_t>(f + a + 6);
    if (a + 24 + (int64_t)entries * 16 + 8 > len) return TH_EINVAL;
    for (int i = 0; i < entries; ++i) {
        const uint64_t child = rd<uint64_t>(f + a + 24 + (int64_t)i * 16 + 8);
        if (level > 0) {
            const int rc = walk_group_node(f, len, base, child, heap_data, heap_size, names, names_cap, addrs, addrs_cap, n, used_names,
                                           depth + 1, budget);
            if (rc != TH_OK) return rc;
            continue;
        }
        if (child == kUndef || child > (uint64_t)len) return TH_EINVAL;
        const int64_t s = base + (int64_t)child;
        if (s < 0 || s > len - 8 || std::memcmp(f + s, "SNOD", 4) != 0) return TH_EINVAL;
        const int nsym = rd<uint16_t>(f + s + 6);
        if (s + 8 + (int64_t)nsym * 40 > len) return TH_EINVAL;
        for (int k = 0; k < nsym; ++k) {
            const uint8_t* e = f + s + 8 + (int64_t)k * 40;
            const uint64_t noff = rd<uint64_t>(e), ohdr = rd<uint64_t>(e + 8);
            if (noff >= (uint64_t)heap_size) return TH_EINVAL;
            const uint8_t* name = f + heap_data + noff;
            const void* z = std::memchr(name, 0, (size_t)(heap_size - (int64_t)noff));
            if (!z) return TH_EINVAL;
            const int64_t nl = (const uint8_t*)z - name + 1;
            if (*n >= addrs_cap || *used_names + nl > names_cap) return TH_ENOMEM;
            std::memcpy(names + *used_names, name, (size_t)nl);
            *used_names += nl;
            addrs[(*n)++] = (int64_t)ohdr;
        }
    }
    return TH_OK;
}
}  // namespace

extern "C" int th_h5_group_links(const void* file, int64_t file_len, int64_t base, int64_t btree_addr, int64_t heap_data, int64_t heap_size,
                                 char* names, int64_t names_cap, int64_t* addrs, int64_t addrs_cap, int64_t* n_out, int64_t* names_len) {
    if (!file || !names || !addrs || !n_out || !names_len || file_len < 0 || names_cap < 0 || addrs_cap < 0) return TH_EINVAL;
    if (heap_data < 0 || heap_size < 0 || heap_data > file_len || heap_size > file_len - heap_data) return TH_EINVAL;
    int64_t n = 0, used = 0, budget = 1 << 20;
    const int rc = walk_group_node((const uint8_t*)file, file_len, base, (uint64_t)btree_addr, heap_data, heap_size, names, names_cap, addrs,
                                   addrs_cap, &n, &used, 0, &budget);
    *n_out = n;
    *names_len = used;
    return rc;
}
