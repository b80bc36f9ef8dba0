// Gradient exchange through the C ABI: theia_comm_* over RCCL (xGMI), one communicator per process / GPU.
//
// RCCL is bound at run time (dlopen "librccl.so.1"), not linked: a process that already carries an RCCL -- PyTorch loads its own copy
// under the same soname for torch.distributed -- must end up with ONE instance (two would each open their own xGMI rings and IPC
// handles), and a single-GPU user of libtheia_hip.so should not need the library at all.  The collectives are stream-ordered: enqueued
// on the caller's stream, never synchronised here.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers still builds the library (RCCL is bound by dlopen at run time; without it the
// theia_comm_* entries return THEIA_ERR_UNSUPPORTED).  The six entry points and the few types they use, as NCCL 2.x / RCCL define them:
#define NCCL_UNIQUE_ID_BYTES 128
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <mutex>

#include "../../include/theia_hip.h"
#include "common.h"

namespace {

struct rccl_api_t {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    char why[256] = {0};
};

rccl_api_t& rccl() {
    static rccl_api_t api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle != nullptr) break;
        }
        if (api.handle == nullptr) {
            const char* e = dlerror();
            snprintf(api.why, sizeof(api.why), "RCCL not found (%s)", e != nullptr ? e : "dlopen failed");
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(api.handle, s);
            if (p == nullptr && api.why[0] == 0) snprintf(api.why, sizeof(api.why), "RCCL symbol %s missing", s);
            return p;
        };
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(sym("ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(sym("ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(sym("ncclAllReduce"));
        api.broadcast = reinterpret_cast<decltype(api.broadcast)>(sym("ncclBroadcast"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
        api.ok = api.why[0] == 0;
    });
    return api;
}

struct comm_t {
    ncclComm_t comm;
    int world, rank;
};

#define COMM_API_OR_FAIL()                                            \
    rccl_api_t& api = rccl();                                         \
    if (!api.ok) {                                                    \
        theia_set_error("theia_comm: %s", api.why);                   \
        return THEIA_ERR_UNSUPPORTED;                                 \
    }
#define COMM_CHECK(call, what)                                                        \
    do {                                                                              \
        ncclResult_t r__ = (call);                                                    \
        if (r__ != ncclSuccess) {                                                     \
            theia_set_error("%s: RCCL error %d (%s)", what, (int)r__, api.error_string(r__)); \
            return THEIA_ERR_LAUNCH;                                                  \
        }                                                                             \
    } while (0)

int nccl_type(int dtype, ncclDataType_t* t) {
    if (dtype == THEIA_F32) *t = ncclFloat32;
    else if (dtype == THEIA_BF16) *t = ncclBfloat16;
    else return THEIA_ERR_INVALID;
    return THEIA_OK;
}

}  // namespace

static_assert(THEIA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "theia_comm id size");

extern "C" int theia_comm_unique_id(void* id_host) {
    THEIA_CHECK_ARG(id_host != nullptr, "theia_comm_unique_id: null id");
    COMM_API_OR_FAIL();
    ncclUniqueId id;
    COMM_CHECK(api.get_unique_id(&id), "theia_comm_unique_id");
    memcpy(id_host, id.internal, NCCL_UNIQUE_ID_BYTES);
    return THEIA_OK;
}

extern "C" int theia_comm_init(void** comm_out, const void* id_host, int world, int rank) {
    THEIA_CHECK_ARG(comm_out != nullptr && id_host != nullptr, "theia_comm_init: null pointer");
    THEIA_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "theia_comm_init: rank %d of %d", rank, world);
    COMM_API_OR_FAIL();
    ncclUniqueId id;
    memcpy(id.internal, id_host, NCCL_UNIQUE_ID_BYTES);
    comm_t* c = new comm_t{nullptr, world, rank};
    ncclResult_t r = api.comm_init_rank(&c->comm, world, id, rank);  // (on the calling thread's current device)
    if (r != ncclSuccess) {
        theia_set_error("theia_comm_init: RCCL error %d (%s)", (int)r, api.error_string(r));
        delete c;
        return THEIA_ERR_LAUNCH;
    }
    *comm_out = c;
    return THEIA_OK;
}

extern "C" int theia_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream) {
    THEIA_CHECK_ARG(comm != nullptr && buf != nullptr && count > 0, "theia_comm_allreduce: bad arguments");
    ncclDataType_t t;
    THEIA_CHECK_ARG(nccl_type(dtype, &t) == THEIA_OK, "theia_comm_allreduce: dtype %d (THEIA_F32 or THEIA_BF16)", dtype);
    COMM_API_OR_FAIL();
    comm_t* c = static_cast<comm_t*>(comm);
    COMM_CHECK(api.all_reduce(buf, buf, (size_t)count, t, average ? ncclAvg : ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream)),
               "theia_comm_allreduce");
    return THEIA_OK;
}

extern "C" int theia_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream) {
    THEIA_CHECK_ARG(comm != nullptr && buf != nullptr && count > 0, "theia_comm_broadcast: bad arguments");
    ncclDataType_t t;
    THEIA_CHECK_ARG(nccl_type(dtype, &t) == THEIA_OK, "theia_comm_broadcast: dtype %d (THEIA_F32 or THEIA_BF16)", dtype);
    COMM_API_OR_FAIL();
    comm_t* c = static_cast<comm_t*>(comm);
    THEIA_CHECK_ARG(root >= 0 && root < c->world, "theia_comm_broadcast: root %d of %d", root, c->world);
    COMM_CHECK(api.broadcast(buf, buf, (size_t)count, t, root, c->comm, reinterpret_cast<hipStream_t>(stream)), "theia_comm_broadcast");
    return THEIA_OK;
}

extern "C" int theia_comm_size(void* comm, int* world, int* rank) {
    THEIA_CHECK_ARG(comm != nullptr, "theia_comm_size: null communicator");
    comm_t* c = static_cast<comm_t*>(comm);
    if (world != nullptr) *world = c->world;
    if (rank != nullptr) *rank = c->rank;
    return THEIA_OK;
}

extern "C" int theia_comm_destroy(void* comm) {
    if (comm == nullptr) return THEIA_OK;
    COMM_API_OR_FAIL();
    comm_t* c = static_cast<comm_t*>(comm);
    ncclResult_t r = api.comm_destroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        theia_set_error("theia_comm_destroy: RCCL error %d (%s)", (int)r, api.error_string(r));
        return THEIA_ERR_LAUNCH;
    }
    return THEIA_OK;
}
