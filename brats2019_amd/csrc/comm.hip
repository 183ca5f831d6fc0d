// comm.hip -- RCCL in the boundary: ru_comm_* + ru_allreduce put the two collectives of a data-parallel step (the
// [2C+1] criterion sums, the live runs of the flat gradient buffer) on the SAME HIP stream as the kernels, with no Python or
// torch.distributed call in the data path.  librccl is bound at run time (dlopen; a process that already loaded an RCCL -- PyTorch
// ships one -- gets that instance through the SONAME), so the library still loads on machines without RCCL and a single-GPU user
// never touches it.  Replaces what nn.DataParallel does with NCCL broadcast / reduce_add inside the reference (main.py:61).
#include "ru_common.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

namespace ru {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

static RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString && api.GroupStart && api.GroupEnd;
    });
    return api;
}

static int rccl_fail(ncclResult_t r, const char* what) {
    set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
    return RU_EHIP;
}

}  // namespace ru

using namespace ru;

struct ru_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int ru_comm_unique_id(void* id_out) {
    RU_REQUIRE(id_out, "ru_comm_unique_id: null argument");
    if (!rccl().ok) { set_error("ru_comm_unique_id: librccl could not be loaded (%s)", dlerror() ? dlerror() : "missing symbols"); return RU_EHIP; }
    static_assert(sizeof(ncclUniqueId) == RU_COMM_ID_BYTES, "RU_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return RU_OK;
}

extern "C" int ru_comm_init(ru_comm_t* out, const void* id, int rank, int world) {
    RU_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "ru_comm_init: bad argument");
    if (!rccl().ok) { set_error("ru_comm_init: librccl could not be loaded"); return RU_EHIP; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ru_comm* c = new ru_comm();
    c->rank = rank; c->world = world;
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);     // binds the communicator to the CURRENT HIP device
    if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    *out = c;
    return RU_OK;
}

extern "C" int ru_comm_destroy(ru_comm_t c) {
    if (!c) return RU_OK;
    ncclResult_t r = ncclSuccess;
    if (c->comm && rccl().ok) r = rccl().CommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? RU_OK : rccl_fail(r, "ncclCommDestroy");
}

extern "C" int ru_comm_rank(ru_comm_t c) { return c ? c->rank : -1; }
extern "C" int ru_comm_world(ru_comm_t c) { return c ? c->world : 0; }

extern "C" int ru_allreduce(ru_comm_t c, void* buf, size_t count, int dtype, ru_stream_t stream) {
    RU_REQUIRE(c && c->comm && buf && (dtype == RU_DT_F32 || dtype == RU_DT_F64), "ru_allreduce: bad argument");
    if (count == 0) return RU_OK;
    const ncclResult_t r = rccl().AllReduce(buf, buf, count, dtype == RU_DT_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? RU_OK : rccl_fail(r, "ncclAllReduce");
}

// ncclGroupStart / ncclGroupEnd around several ru_allreduce calls: RCCL fuses them into ONE launch (the gradient runs of a step travel
// together instead of as back-to-back collectives, each with its own launch and ring set-up)
extern "C" int ru_comm_group_begin(void) {
    if (!rccl().ok) { set_error("ru_comm_group_begin: librccl could not be loaded"); return RU_EHIP; }
    const ncclResult_t r = rccl().GroupStart();
    return r == ncclSuccess ? RU_OK : rccl_fail(r, "ncclGroupStart");
}
extern "C" int ru_comm_group_end(void) {
    if (!rccl().ok) { set_error("ru_comm_group_end: librccl could not be loaded"); return RU_EHIP; }
    const ncclResult_t r = rccl().GroupEnd();
    return r == ncclSuccess ? RU_OK : rccl_fail(r, "ncclGroupEnd");
}
