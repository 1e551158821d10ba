// The one exchange step of a sharded evaluation, for callers of the C ABI that have no collective of
// their own: an all-reduce (sum) of the error counters over RCCL, one process per GPU.
// (The reference sums per-device berStatistics on the host after its process pool returns,
// ldpcCUDA.py:897-912.)  RCCL is opened at run time, when the first communicator is made, so the
// library builds and loads on machines without it and single-GPU users never pay for it.
#include <hip/hip_runtime.h>
// Types and constants only: every call goes through dlsym.  Where the RCCL headers are not installed the few
// declarations this file uses are stated here (they are part of RCCL's stable C ABI, the same as NCCL's), so that
// the library builds -- and not only loads -- without RCCL.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif
enum { LDPC_COMM_ID_BYTES_EXPECTED = 128 };      // = LDPC_COMM_ID_BYTES of include/ldpc_hip.h (checked below)
static_assert(sizeof(ncclUniqueId) == LDPC_COMM_ID_BYTES_EXPECTED, "the identifier the C ABI hands around is RCCL's");

#include <dlfcn.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <new>

#include "../../include/ldpc_hip.h"

extern "C" __attribute__((visibility("hidden"))) void ldpc_set_error_(const char *text);      // ldpc_hip.hip: sets this thread's ldpc_last_error() text
static_assert(LDPC_COMM_ID_BYTES == LDPC_COMM_ID_BYTES_EXPECTED, "identifier size");

namespace {

int comm_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ldpc_set_error_(buf);
    return code;
}

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;

// The library a framework in this process already loaded (PyTorch bundles its own copy under the same
// SONAME) before any other: two copies of a collective library in one process do not share devices well.
int open_rccl(const RcclApi **out)
{
    std::lock_guard<std::mutex> g(g_rccl_mutex);
    if (!g_rccl.lib) {
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return comm_fail(LDPC_E_UNSUPPORTED, "RCCL is not available: %s", dlerror());
        RcclApi r;
        r.lib = lib;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
            dlclose(lib);
            return comm_fail(LDPC_E_UNSUPPORTED, "the RCCL library found lacks a needed entry point");
        }
        g_rccl = r;
    }
    *out = &g_rccl;
    return 0;
}

}  // namespace

struct ldpc_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    const RcclApi *api = nullptr;      // the entry points this communicator was made with (published under the mutex once)
    hipStream_t stream = nullptr;
    unsigned long long *d_buf = nullptr;      // device staging of the counters
    size_t cap = 0;                           // in counters
};

#define COMM_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return comm_fail(LDPC_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int ldpc_comm_unique_id(void *id_out)
{
    if (!id_out) return comm_fail(LDPC_E_ARG, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == LDPC_COMM_ID_BYTES, "LDPC_COMM_ID_BYTES must be RCCL's id size");
    const RcclApi *r;
    int rc = open_rccl(&r);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return comm_fail(LDPC_E_DEVICE, "ncclGetUniqueId failed: %s", r->GetErrorString(e));
    __builtin_memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" void ldpc_comm_destroy(ldpc_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->api) (void)c->api->CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ldpc_comm_create(int device, int rank, int world, const void *id, ldpc_comm **out)
{
    if (!out) return comm_fail(LDPC_E_ARG, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return comm_fail(LDPC_E_ARG, "need 0 <= rank < world (got %d of %d)", rank, world);
    if (!id) return comm_fail(LDPC_E_ARG, "id is NULL (rank 0 makes one with ldpc_comm_unique_id and hands it to every rank)");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return comm_fail(LDPC_E_DEVICE, "no HIP device available");
    if (device < 0 || device >= cnt) return comm_fail(LDPC_E_ARG, "device %d out of range (%d visible)", device, cnt);
    const RcclApi *r;
    int rc = open_rccl(&r);
    if (rc) return rc;
    COMM_HIP(hipSetDevice(device));
    ldpc_comm *c = new (std::nothrow) ldpc_comm();
    if (!c) return comm_fail(LDPC_E_NOMEM, "out of host memory");
    c->device = device; c->rank = rank; c->world = world; c->api = r;
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete c; return comm_fail(LDPC_E_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(he)); }
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof(uid));
    ncclResult_t e = r->CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
        c->comm = nullptr;
        ldpc_comm_destroy(c);
        return comm_fail(LDPC_E_DEVICE, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, r->GetErrorString(e));
    }
    *out = c;
    return 0;
}

extern "C" int ldpc_allreduce_counters(ldpc_comm *c, uint64_t *counters, int count)
{
    if (!c) return comm_fail(LDPC_E_ARG, "communicator is NULL");
    if (count < 0 || (count > 0 && !counters)) return comm_fail(LDPC_E_ARG, "bad counter array");
    if (count == 0) return 0;
    COMM_HIP(hipSetDevice(c->device));
    if ((size_t)count > c->cap) {
        if (c->d_buf) { (void)hipFree(c->d_buf); c->d_buf = nullptr; c->cap = 0; }
        const size_t cap = ((size_t)count + 63) & ~(size_t)63;
        COMM_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_buf), cap * sizeof(uint64_t)));
        c->cap = cap;
    }
    const size_t bytes = (size_t)count * sizeof(uint64_t);
    COMM_HIP(hipMemcpyAsync(c->d_buf, counters, bytes, hipMemcpyHostToDevice, c->stream));
    ncclResult_t e = c->api->AllReduce(c->d_buf, c->d_buf, (size_t)count, ncclUint64, ncclSum, c->comm, c->stream);
    if (e != ncclSuccess) return comm_fail(LDPC_E_DEVICE, "ncclAllReduce failed: %s", c->api->GetErrorString(e));
    COMM_HIP(hipMemcpyAsync(counters, c->d_buf, bytes, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
