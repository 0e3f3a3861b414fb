// fake_rccl.hip -- TEST DOUBLE of librccl for the ordering test of the data-parallel step (tests/test_parity_gpu.py::
// test_dp_event_ordering_with_a_fake_collective).  Not part of the product; built by __graft_entry__.build() into
// tests/fake_rccl/libfake_rccl.so and selected with L3_RCCL_LIB (csrc/comm.hip binds whatever library that names).
//
// Two ranks cannot share one GPU under the real RCCL, and at world size 1 an in-place all-reduce is an identity, so a
// collective launched BEFORE its bucket's backward finished -- or an Adam step launched before the last collective --
// would go unnoticed.  This double makes the collective visible and slow:
//   ncclAllReduce(sum)  = "world" ranks that all hold this rank's data: out = world * in, on the caller's stream,
//                         behind a kernel that spins for FAKE_RCCL_DELAY_US (default 300) microseconds;
//   ncclAllReduce(max)  = identity;
//   ncclAllGather       = every rank's slot holds this rank's data, behind the same delay.
// FAKE_RCCL_BLOCKS / FAKE_RCCL_THREADS / FAKE_RCCL_LDS_KB give the delay kernel the CU footprint of a real ring kernel.
// With an engine whose global batch is world x its own batch (loss gradients scaled by 1 / (world * B)) the reduced
// gradients equal, bit for bit (powers of two), those of the plain single-GPU step at global batch B -- if and only if
// every bucket is reduced after its last writer and before Adam reads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

struct FakeComm {
    int world, rank;
    long launches;
};

namespace {

// One workgroup of the collective's stand-in: holds its threads' registers and `lds_bytes` of LDS for `cycles` of the 100-MHz
// wall clock, counted from the moment the workgroup itself starts (a workgroup that has to wait for a CU holds it that much later).
__global__ void spin_kernel(long long cycles, int lds_bytes, int* sink) {
    extern __shared__ int lds[];
    if (lds_bytes > 0) lds[threadIdx.x] = (int)threadIdx.x;         // the allocation is what matters; keep it live
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (lds_bytes > 0 && lds[(threadIdx.x + 1) % blockDim.x] < 0) *sink = 1;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v != nullptr && *v ? atoi(v) : dflt;
}

// The delay in front of every collective.  Default: one 64-thread workgroup, no LDS (rounds 3-5: visible and slow, but with no
// footprint).  A real RCCL ring kernel is tens of 256-512-thread workgroups with LDS and many registers that cannot share a CU
// with a 160-KiB-LDS convolution workgroup: FAKE_RCCL_BLOCKS x FAKE_RCCL_THREADS threads x FAKE_RCCL_LDS_KB KiB of LDS, each
// workgroup holding its CU share for FAKE_RCCL_DELAY_US.
void fake_delay(hipStream_t stream) {
    const long long us = env_int("FAKE_RCCL_DELAY_US", 300);
    if (us <= 0) return;
    const int blocks = env_int("FAKE_RCCL_BLOCKS", 1), threads = env_int("FAKE_RCCL_THREADS", 64), lds = env_int("FAKE_RCCL_LDS_KB", 0) * 1024;
    static int* sink = nullptr;
    if (sink == nullptr && hipMalloc((void**)&sink, 64) != hipSuccess) return;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(spin_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(threads < 64 ? 64 : threads > 1024 ? 1024 : threads), (size_t)lds, stream,
                       us * 100, lds, sink);      // wall_clock64: 100 MHz
}

template <typename T>
__global__ void scale_kernel(const T* in, T* out, size_t n, T f) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * f;
}

// ncclMax with a peer that holds `peer` in element `at` (FAKE_RCCL_MAX_PEER="<index>:<value>"): a rank that was configured differently
__global__ void max_peer_kernel(const double* in, double* out, size_t n, int at, double peer) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)i == at && peer > in[i] ? peer : in[i];
}

long g_total_launches = 0;

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
    *version = 99999;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    memcpy(id->internal, "fake-rccl", 9);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    *comm = reinterpret_cast<ncclComm_t>(new FakeComm{nranks, rank, 0});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<FakeComm*>(comm);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    fake_delay(stream);
    const float f = op == ncclSum ? (float)c->world : 1.f;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (const char* mp = getenv("FAKE_RCCL_MAX_PEER"); mp != nullptr && op == ncclMax && dt == ncclDouble) {
        const char* colon = strchr(mp, ':');
        hipLaunchKernelGGL(max_peer_kernel, dim3(blocks), dim3(256), 0, stream, (const double*)send, (double*)recv, count, atoi(mp),
                           colon ? atof(colon + 1) : 0.0);
        ++c->launches;
        ++g_total_launches;
        return ncclSuccess;
    }
    if (dt == ncclFloat)
        hipLaunchKernelGGL(scale_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)send, (float*)recv, count, f);
    else if (dt == ncclDouble)
        hipLaunchKernelGGL(scale_kernel<double>, dim3(blocks), dim3(256), 0, stream, (const double*)send, (double*)recv, count,
                           (double)f);
    else
        return ncclInvalidArgument;
    ++c->launches;
    ++g_total_launches;
    return ncclSuccess;
}

// "world" ranks that all hold this rank's data: every slot of recv = send (behind the same delay as an all-reduce)
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (dt != ncclFloat) return ncclInvalidArgument;
    fake_delay(stream);
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpyAsync((float*)recv + (size_t)r * count, send, count * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    ++c->launches;
    ++g_total_launches;
    return ncclSuccess;
}

// test hook: collectives launched so far in this process
long fake_rccl_launches() { return g_total_launches; }

}  // extern "C"
