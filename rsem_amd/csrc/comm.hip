// comm.hip -- rsem_comm: the collectives of the multi-GPU paths (C ABI: include/rsem_hip.h).
//
// What they replace in the reference: the serial reduction of the per-thread count vectors after every E step
// (EM.cpp:385-389) and the sum of the per-chain Gibbs accumulators in release() (Gibbs.cpp:372-388).  Here the
// shards / chains live on different GPUs, so the sums are RCCL collectives over xGMI, enqueued on the stream the
// kernels run on (no host round trip inside the EM loop).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "comm_internal.hpp"

namespace {

// librccl.so is half a gigabyte of code objects: it is loaded when the first communicator is asked for, not with
// librsem_hip.so (single-GPU runs never pay for it).  Inside a process that already holds RCCL (bench.py: torch) the
// same library instance is found by its soname.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static std::mutex mu;
    static Rccl R;
    std::lock_guard<std::mutex> lk(mu);
    if (R.lib || !R.error.empty()) return &R;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (R.lib) break;
    }
    if (!R.lib) { R.error = std::string("cannot load librccl.so: ") + dlerror(); return &R; }
#define SYM(field, sym)                                                        \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, sym));            \
    if (!R.field) { R.error = std::string("librccl.so lacks ") + sym; R.lib = nullptr; return &R; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(Reduce, "ncclReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return &R;
}

struct LocalGroup {  // ranks of one process, possibly on the same device
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<double*> bufs;
    std::vector<int> devices;
    int refs = 0;
    bool failed = false;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};

__global__ void k_sum_bufs(size_t n, int world, double* const* bufs, double* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int r = 0; r < world; r++) s += bufs[r][i];  // rank order: every rank computes the same bits
    out[i] = s;
}

}  // namespace

struct rsem_comm {
    int kind = 0;  // 0 RCCL, 1 LOCAL
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;
    LocalGroup* grp = nullptr;
    double* d_scratch = nullptr;  // LOCAL: result staging
    size_t scratch_n = 0;
    double** d_ptrs = nullptr;    // LOCAL: device copy of the peers' buffer pointers
};

#define RSEM_NCCL_TRY(expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) {                                                                             \
            rsem::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl()->GetErrorString(_r));  \
            return RSEM_ERR_HIP;                                                                             \
        }                                                                                                    \
    } while (0)
#define RSEM_NEED_RCCL()                                                       \
    do {                                                                       \
        if (!rccl()->lib) {                                                    \
            rsem::set_last_error("%s", rccl()->error.c_str());                 \
            return RSEM_ERR_HIP;                                               \
        }                                                                      \
    } while (0)

namespace rsem {

int comm_rank(const rsem_comm* c) { return c ? c->rank : 0; }
int comm_world(const rsem_comm* c) { return c ? c->world : 1; }

static int local_exchange(rsem_comm* c, double* d_buf, size_t n, hipStream_t st, bool all, int root) {
    LocalGroup* g = c->grp;
    // A failure on one rank must not leave the others waiting at a barrier: every rank arrives at BOTH barriers whatever
    // happened to it, the failure is recorded in the group, and every rank returns an error after the second barrier.
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && c->scratch_n < n) {
        (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_n = 0;
        e = hipMalloc((void**)&c->d_scratch, sizeof(double) * n);
        if (e == hipSuccess) c->scratch_n = n;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // this rank's contribution is complete
    g->bufs[c->rank] = d_buf;
    if (e != hipSuccess) { std::lock_guard<std::mutex> lk(g->mu); g->failed = true; }
    g->barrier();                            // ... and so is everybody else's
    bool group_failed;
    { std::lock_guard<std::mutex> lk(g->mu); group_failed = g->failed; }
    if (!group_failed && (all || c->rank == root)) {
        e = hipMemcpyAsync(c->d_ptrs, g->bufs.data(), sizeof(double*) * g->world, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_sum_bufs, dim3(ceil_div(n, 256)), dim3(256), 0, st, n, g->world, (double* const*)c->d_ptrs, c->d_scratch);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { std::lock_guard<std::mutex> lk(g->mu); g->failed = true; }
    }
    g->barrier();                            // every reader is done with the peers' buffers
    { std::lock_guard<std::mutex> lk(g->mu); group_failed = g->failed; }
    if (!group_failed && (all || c->rank == root))
        e = hipMemcpyAsync(d_buf, c->d_scratch, sizeof(double) * n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_last_error("local exchange: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RSEM_ERR_NOMEM : RSEM_ERR_HIP;
    }
    if (group_failed) {
        set_last_error("local exchange: another rank of the group failed");
        return RSEM_ERR_HIP;
    }
    return RSEM_OK;
}

// a one-rank communicator normally skips the collective; RSEM_COMM_FORCE=1 issues it anyway (single-GPU test of the RCCL calls)
static bool skip_single(const rsem_comm* c) {
    return !c || (c->world == 1 && getenv("RSEM_COMM_FORCE") == nullptr);
}

bool comm_active(const rsem_comm* c) { return !skip_single(c); }

int comm_allreduce_sum_f64(rsem_comm* c, double* d_buf, size_t n, hipStream_t st) {
    if (skip_single(c)) return RSEM_OK;
    if (c->kind == 1) return local_exchange(c, d_buf, n, st, true, 0);
    RSEM_NCCL_TRY(rccl()->AllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, c->nccl, st));
    return RSEM_OK;
}

int comm_reduce_sum_f64(rsem_comm* c, double* d_buf, size_t n, int root, hipStream_t st) {
    if (skip_single(c)) return RSEM_OK;
    if (c->kind == 1) return local_exchange(c, d_buf, n, st, false, root);
    RSEM_NCCL_TRY(rccl()->Reduce(d_buf, d_buf, n, ncclDouble, ncclSum, root, c->nccl, st));
    return RSEM_OK;
}

}  // namespace rsem

extern "C" {

int rsem_comm_unique_id(char* id) {
    RSEM_REQUIRE(id != nullptr, "id is NULL");
    static_assert(sizeof(ncclUniqueId) <= RSEM_COMM_ID_BYTES, "RSEM_COMM_ID_BYTES too small for ncclUniqueId");
    RSEM_NEED_RCCL();
    ncclUniqueId u;
    RSEM_NCCL_TRY(rccl()->GetUniqueId(&u));
    memset(id, 0, RSEM_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return RSEM_OK;
}

int rsem_comm_create(rsem_comm** out, int device, int rank, int world, const char* id) {
    RSEM_REQUIRE(out && id, "NULL argument");
    *out = nullptr;
    RSEM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        rsem::set_last_error("no HIP device %d (have %d)", device, ndev);
        return RSEM_ERR_NODEVICE;
    }
    RSEM_NEED_RCCL();
    RSEM_HIP_TRY(hipSetDevice(device));
    rsem_comm* c = new (std::nothrow) rsem_comm();
    if (!c) return RSEM_ERR_NOMEM;
    c->kind = 0;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = rccl()->CommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) {
        rsem::set_last_error("ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, rccl()->GetErrorString(r));
        delete c;
        return RSEM_ERR_HIP;
    }
    *out = c;
    return RSEM_OK;
}

int rsem_comm_create_local(rsem_comm** out, int world, const int* devices) {
    RSEM_REQUIRE(out && devices && world >= 1, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        rsem::set_last_error("no HIP device");
        return RSEM_ERR_NODEVICE;
    }
    for (int r = 0; r < world; r++) RSEM_REQUIRE(devices[r] >= 0 && devices[r] < ndev, "device index out of range");
    LocalGroup* g = new (std::nothrow) LocalGroup();
    if (!g) return RSEM_ERR_NOMEM;
    g->world = world;
    g->bufs.assign(world, nullptr);
    g->devices.assign(devices, devices + world);
    g->refs = world;
    for (int r = 0; r < world; r++) out[r] = nullptr;
    for (int r = 0; r < world; r++) {
        rsem_comm* c = new (std::nothrow) rsem_comm();
        hipError_t e = c ? hipSetDevice(devices[r]) : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_ptrs, sizeof(double*) * world);
        // ranks on different devices read each other's buffers directly
        for (int q = 0; e == hipSuccess && q < world; q++)
            if (devices[q] != devices[r]) {
                hipError_t pe = hipDeviceEnablePeerAccess(devices[q], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
                (void)hipGetLastError();
            }
        if (e != hipSuccess) {
            rsem::set_last_error("rsem_comm_create_local: %s", hipGetErrorString(e));
            if (c) { (void)hipFree(c->d_ptrs); delete c; }
            for (int q = 0; q < r; q++) { (void)hipSetDevice(devices[q]); (void)hipFree(out[q]->d_ptrs); delete out[q]; out[q] = nullptr; }
            delete g;
            return RSEM_ERR_HIP;
        }
        c->kind = 1;
        c->rank = r;
        c->world = world;
        c->device = devices[r];
        c->grp = g;
        out[r] = c;
    }
    return RSEM_OK;
}

int rsem_comm_rank(const rsem_comm* c) { return rsem::comm_rank(c); }
int rsem_comm_world(const rsem_comm* c) { return rsem::comm_world(c); }

int rsem_comm_allreduce_f64(rsem_comm* c, void* d_buf, uint64_t n, void* stream) {
    RSEM_REQUIRE(c && d_buf, "NULL argument");
    return rsem::comm_allreduce_sum_f64(c, (double*)d_buf, (size_t)n, (hipStream_t)stream);
}

int rsem_comm_destroy(rsem_comm* c) {
    if (!c) return RSEM_OK;
    (void)hipSetDevice(c->device);
    if (c->kind == 0) {
        if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    } else {
        (void)hipFree(c->d_scratch);
        (void)hipFree(c->d_ptrs);
        bool last = false;
        {
            std::lock_guard<std::mutex> lk(c->grp->mu);
            last = (--c->grp->refs == 0);
        }
        if (last) delete c->grp;
    }
    delete c;
    return RSEM_OK;
}

}  // extern "C"
