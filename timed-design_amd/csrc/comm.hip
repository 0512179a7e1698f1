// Multi-GPU reassembly of per-chain probability maps: one process per GPU, contiguous frame shards,
// one exchange step — an RCCL gather of [n_i, n_classes] fp32 row blocks to the root over xGMI
// (SURVEY.md §8e).  The reference is single-process (it has no collective at all); this is the
// piece that lets rank 0 write the .csv/.fasta exactly as reference predict.py:161-185 does.
//
// xGMI is point-to-point (7 direct links per GPU), so the gather is issued as grouped
// ncclSend/ncclRecv pairs: every peer -> root transfer rides its own direct link concurrently;
// a ring all-gather would serialise on per-link bandwidth for no benefit here.
//
// librccl.so is opened lazily with dlopen so that the single-GPU product (and CPU-only symbol
// checks) never depend on it.
#include "common.h"

#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

typedef struct ncclComm* ncclComm_t;
struct NcclUniqueId { char internal[128]; };
enum { kNcclFloat32 = 7 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.h ? &r : nullptr;
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.h) break;
    }
    if (!r.h) { th_set_error("cannot dlopen librccl.so: %s", dlerror()); return nullptr; }
#define SYM(field, name)                                                       \
    *(void**)(&r.field) = dlsym(r.h, name);                                    \
    if (!r.field) { th_set_error("librccl.so lacks %s", name); r.h = nullptr; return nullptr; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return &r;
}

#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int _r = (expr);                                                                           \
        if (_r != 0) { th_set_error("%s failed: %s", #expr, R->GetErrorString(_r)); return TH_ECOMM; } \
    } while (0)

}  // namespace

struct th_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    float* d_token = nullptr;
    // TH_COMM_SELF_RCCL=1 (read once by th_comm_init): the root's own block travels through a grouped ncclSend/ncclRecv
    // pair to itself instead of a device copy, so that a 1-rank communicator on a 1-GPU box drives the very transfer calls
    // an N-rank gather issues (tests/test_distributed.py); the product default stays the copy.
    bool self_rccl = false;
    int64_t n_send = 0, n_recv = 0, bytes_send = 0, bytes_recv = 0, n_copy = 0;   // th_comm_stats
};

extern "C" {

int th_comm_unique_id(char id[TH_COMM_ID_BYTES]) {
    Rccl* R = rccl();
    if (!R) return TH_ECOMM;
    NcclUniqueId u;
    NCCL_TRY(R->GetUniqueId(&u));
    static_assert(sizeof u.internal == TH_COMM_ID_BYTES, "id size");
    std::memcpy(id, u.internal, TH_COMM_ID_BYTES);
    return TH_OK;
}

int th_comm_init(const char id[TH_COMM_ID_BYTES], int n_ranks, int rank, int device, th_comm** out) {
    if (!id || !out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) TH_FAIL(TH_EINVAL, "th_comm_init: bad argument");
    Rccl* R = rccl();
    if (!R) return TH_ECOMM;
    HIP_TRY(hipSetDevice(device));
    th_comm* c = new th_comm;
    c->n_ranks = n_ranks; c->rank = rank; c->device = device;
    {
        const char* e = getenv("TH_COMM_SELF_RCCL");
        c->self_rccl = e && atoi(e) != 0;
    }
    NcclUniqueId u;
    std::memcpy(u.internal, id, TH_COMM_ID_BYTES);
    int r = R->CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != 0) { th_set_error("ncclCommInitRank failed: %s", R->GetErrorString(r)); delete c; return TH_ECOMM; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = th_malloc_retry(&c->d_token, sizeof(float));
    if (e == hipSuccess) e = hipMemset(c->d_token, 0, sizeof(float));   // the barrier all-reduces it: keep it finite
    if (e != hipSuccess) { th_set_error("th_comm_init: %s", hipGetErrorString(e)); th_comm_free(c); return TH_EHIP; }
    *out = c;
    return TH_OK;
}

void th_comm_free(th_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    if (c->d_token) (void)hipFree(c->d_token);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int th_comm_gather_rows(th_comm* c, const float* d_local, const int64_t* counts, int width, int root, float* d_out) {
    if (!c || !counts || width <= 0 || root < 0 || root >= c->n_ranks) TH_FAIL(TH_EINVAL, "th_comm_gather_rows: bad argument");
    Rccl* R = rccl();
    if (!R) return TH_ECOMM;
    HIP_TRY(hipSetDevice(c->device));
    if (c->rank == root && !d_out) TH_FAIL(TH_EINVAL, "root needs an output buffer");
    if (counts[c->rank] > 0 && !d_local) TH_FAIL(TH_EINVAL, "rank %d has %lld rows but no local buffer", c->rank, (long long)counts[c->rank]);
    int64_t my_row = 0;
    for (int r = 0; r < c->rank; ++r) my_row += counts[r];
    // the root's own block is a plain device copy: issued before (outside) the RCCL group — so a 1-rank gather issues NO
    // RCCL transfer at all unless TH_COMM_SELF_RCCL routes that block through a send/recv pair to self inside the group
    const bool self_pair = c->self_rccl && c->rank == root && counts[root] > 0;
    if (c->rank == root && counts[root] > 0 && !self_pair) {
        HIP_TRY(hipMemcpyAsync(d_out + (size_t)my_row * width, d_local, (size_t)counts[root] * width * sizeof(float),
                               hipMemcpyDeviceToDevice, c->stream));
        ++c->n_copy;
    }
    // Inside the group nothing may return early: an error is remembered, the group is always closed, and the first
    // failure is reported afterwards (an open group would poison every later collective on this thread).
    int first_err = 0;
    const char* what = "";
    int rc = R->GroupStart();
    if (rc != 0) { th_set_error("ncclGroupStart failed: %s", R->GetErrorString(rc)); return TH_ECOMM; }
    if (c->rank == root) {
        int64_t row = 0;
        for (int r = 0; r < c->n_ranks; ++r) {
            if ((r != root || self_pair) && counts[r] > 0 && !first_err) {
                rc = R->Recv(d_out + (size_t)row * width, (size_t)counts[r] * width, kNcclFloat32, r, c->comm, c->stream);
                if (rc != 0) { first_err = rc; what = "ncclRecv"; }
                else { ++c->n_recv; c->bytes_recv += counts[r] * (int64_t)width * 4; }
            }
            row += counts[r];
        }
    }
    if ((c->rank != root || self_pair) && counts[c->rank] > 0 && !first_err) {
        rc = R->Send(d_local, (size_t)counts[c->rank] * width, kNcclFloat32, root, c->comm, c->stream);
        if (rc != 0) { first_err = rc; what = "ncclSend"; }
        else { ++c->n_send; c->bytes_send += counts[c->rank] * (int64_t)width * 4; }
    }
    rc = R->GroupEnd();
    if (first_err) { th_set_error("%s failed: %s", what, R->GetErrorString(first_err)); return TH_ECOMM; }
    if (rc != 0) { th_set_error("ncclGroupEnd failed: %s", R->GetErrorString(rc)); return TH_ECOMM; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TH_OK;
}

int th_comm_stats(th_comm* c, int64_t out[5]) {
    if (!c || !out) TH_FAIL(TH_EINVAL, "th_comm_stats: bad argument");
    out[0] = c->n_send; out[1] = c->n_recv; out[2] = c->bytes_send; out[3] = c->bytes_recv; out[4] = c->n_copy;
    return TH_OK;
}

int th_comm_barrier(th_comm* c) {
    if (!c) TH_FAIL(TH_EINVAL, "null comm");
    Rccl* R = rccl();
    if (!R) return TH_ECOMM;
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(R->AllReduce(c->d_token, c->d_token, 1, kNcclFloat32, /*ncclSum*/ 0, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TH_OK;
}

}  // extern "C"
