// fake_rccl.hip -- TEST DOUBLE of librccl for the ordering test of the data-parallel step (tests/test_parity_gpu.py::
// test_dp_event_ordering_with_a_fake_collective).  Not part of the product; built by __graft_entry__.build() into
// tests/fake_rccl/libfake_rccl.so and selected with L3_RCCL_LIB (csrc/comm.hip binds whatever library that names).
//
// Two ranks cannot share one GPU under the real RCCL, and at world size 1 an in-place all-reduce is an identity, so a
// collective launched BEFORE its bucket's backward finished -- or an Adam step launched before the last collective --
// would go unnoticed.  This double makes the collective visible and slow:
//   ncclAllReduce(sum)  = "world" ranks that all hold this rank's data: out = world * in, on the caller's stream,
//                         behind a kernel that spins for FAKE_RCCL_DELAY_US (default 300) microseconds;
//   ncclAllReduce(max)  = identity.
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

__global__ void spin_kernel(long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

template <typename T>
__global__ void scale_kernel(const T* in, T* out, size_t n, T f) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * f;
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
    const char* d = getenv("FAKE_RCCL_DELAY_US");
    const long long us = d ? atoll(d) : 300;
    if (us > 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, us * 100);      // wall_clock64: 100 MHz
    const float f = op == ncclSum ? (float)c->world : 1.f;
    const unsigned blocks = (unsigned)((count + 255) / 256);
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

// test hook: collectives launched so far in this process
long fake_rccl_launches() { return g_total_launches; }

}  // extern "C"
