// comm.hip -- RCCL binding of libl3hip.so.
//
// Replaces the gradient exchange of the reference's in-graph data parallelism (l3embedding/
// training_utils.py:141-170: every replica's gradient flows into one AddN on the parameter device) by
// ncclAllReduce over xGMI between one-process-per-GPU ranks.
//
// librccl is bound with dlopen at the first l3_comm_* call instead of at link time: a single-GPU user
// never maps the 570 MB library, and a process that already carries an RCCL (PyTorch bundles its own copy
// under the same SONAME) keeps exactly one -- dlopen("librccl.so.1") resolves to the copy that is already
// loaded, which is also the one bound to the process's HIP runtime.
#include "comm.h"
#include "knobs.h"

#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>

#include <mutex>

namespace l3 {

namespace {

struct Api {
    void* handle = nullptr;
    std::string path, error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

Api g_api;
std::once_flag g_api_once;

void bind_api() {
    const char* override_path = getenv("L3_RCCL_LIB");
    const char* candidates[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : candidates) {
        if (!name || !*name) continue;
        g_api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_api.handle) break;
        g_api.error = dlerror();
    }
    if (!g_api.handle) {
        g_api.error = "cannot load librccl: " + g_api.error;
        return;
    }
    auto sym = [&](const char* n) {
        void* p = dlsym(g_api.handle, n);
        if (!p) g_api.error = std::string("librccl lacks ") + n;
        return p;
    };
    g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(sym("ncclCommInitRank"));
    g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(sym("ncclCommDestroy"));
    g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(sym("ncclAllReduce"));
    g_api.AllGather = reinterpret_cast<decltype(g_api.AllGather)>(sym("ncclAllGather"));
    g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(sym("ncclGetErrorString"));
    g_api.GetVersion = reinterpret_cast<decltype(g_api.GetVersion)>(sym("ncclGetVersion"));
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommDestroy || !g_api.AllReduce || !g_api.AllGather || !g_api.GetErrorString) {
        dlclose(g_api.handle);
        g_api.handle = nullptr;
        return;
    }
    g_api.error.clear();
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(g_api.AllReduce), &info) && info.dli_fname) g_api.path = info.dli_fname;
}

const Api* api(std::string* err) {
    std::call_once(g_api_once, bind_api);
    if (!g_api.handle) {
        if (err) *err = g_api.error;
        return nullptr;
    }
    return &g_api;
}

bool ok(const Api* a, ncclResult_t r, const char* what, std::string* err) {
    if (r == ncclSuccess) return true;
    if (err) *err = std::string(what) + ": " + a->GetErrorString(r);
    return false;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int world = 0, rank = 0, device = 0;
};

const char* comm_library_path() { return g_api.path.c_str(); }

int comm_unique_id(void* id128, std::string* err) {
    const Api* a = api(err);
    if (!a) return -1;
    static_assert(sizeof(ncclUniqueId) == COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    if (!ok(a, a->GetUniqueId(&id), "ncclGetUniqueId", err)) return -1;
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int comm_create(const void* id128, int world, int rank, int device, Comm** out, std::string* err) {
    const Api* a = api(err);
    if (!a) return -1;
    if (world < 1 || rank < 0 || rank >= world) {
        if (err) *err = "l3_comm_init: rank outside [0, world)";
        return -1;
    }
    if (hipSetDevice(device) != hipSuccess) {
        if (err) *err = "l3_comm_init: hipSetDevice failed";
        return -1;
    }
    Comm* c = new Comm();
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (!ok(a, a->CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank", err)) {
        delete c;
        return -1;
    }
    // HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  When this stream ends up
    // sharing a queue with one of the engine's two tower streams, its event waits become false dependencies for that
    // tower: measured at world size 1, the audio tower's backward queued behind the waits for the vision buckets
    // (+3.6 % per step; +13 % with a high-priority stream here).  With GPU_MAX_HW_QUEUES=8 -- set in the environment
    // before the HIP runtime initialises; l3embedding_amd._lib.load() and bench.py default it -- the data-parallel step
    // costs +0.6 % over the single-GPU step (profiles/r02_dp_overhead.txt).
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        if (err) *err = "l3_comm_init: hipStreamCreate failed";
        a->CommDestroy(c->comm);
        delete c;
        return -1;
    }
    *out = c;
    return 0;
}

void comm_destroy(Comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)l3::stream_wait(c->stream);
    if (c->comm && g_api.handle) g_api.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int comm_world(const Comm* c) { return c->world; }
int comm_rank(const Comm* c) { return c->rank; }
hipStream_t comm_stream(const Comm* c) { return c->stream; }

int comm_allreduce_f32(Comm* c, float* buf, size_t n, int op, std::string* err) {
    if (n == 0) return 0;
    const Api* a = api(err);
    if (!a) return -1;
    return ok(a, a->AllReduce(buf, buf, n, ncclFloat, op == 1 ? ncclMax : ncclSum, c->comm, c->stream), "ncclAllReduce", err) ? 0 : -1;
}

int comm_allgather_f32(Comm* c, const float* send, float* recv, size_t n_per_rank, std::string* err) {
    if (n_per_rank == 0) return 0;
    const Api* a = api(err);
    if (!a) return -1;
    return ok(a, a->AllGather(send, recv, n_per_rank, ncclFloat, c->comm, c->stream), "ncclAllGather", err) ? 0 : -1;
}

int comm_version() {
    int v = 0;
    if (g_api.handle && g_api.GetVersion) (void)g_api.GetVersion(&v);
    return v;
}

int comm_allreduce_f64(Comm* c, double* buf, size_t n, int op, std::string* err) {
    if (n == 0) return 0;
    const Api* a = api(err);
    if (!a) return -1;
    return ok(a, a->AllReduce(buf, buf, n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, c->stream), "ncclAllReduce", err) ? 0 : -1;
}

}  // namespace l3
