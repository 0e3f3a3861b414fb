// device_common.h -- vector types and small device helpers shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>

#include "knobs.h"

namespace l3 {

// per-device one-time launch set-up (hipFuncSetAttribute) is keyed by the device ordinal modulo this power of two
constexpr int L3_MAX_DEVICES = 64;

// clang native vectors: stay in VGPRs (HIP's float4 is a struct; assigning one from a dereference
// made hipcc go through a scratch memcpy)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Bijective XCD-aware remap: the hardware places workgroup b on XCD b % 8; give every XCD a contiguous
// range of logical tiles so neighbouring tiles (shared halo rows, shared filter slice) hit the same
// private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---- work counters of a persistent grid ------------------------------------------------------------------------------------
// A persistent grid (one workgroup per CU walking the tile blocks of a launch) that gives workgroup b the blocks b, b + grid,
// b + 2 grid, ... assumes that every workgroup starts at once.  Beside a collective that holds CUs (an RCCL ring kernel: tens of
// 256-512-thread workgroups with LDS that cannot share a CU with a 150-KiB convolution workgroup) k workgroups start only when
// others have finished their WHOLE list, and the launch takes twice as long.  Dynamic assignment instead: the
// tile blocks of a launch are eight queues, one per XCD (queue x = the physical ids lt with lt % 8 == x, which xcd_remap maps to
// a contiguous range of logical blocks: neighbours in time share that XCD's L2, as under the static stride); a workgroup on XCD
// x = blockIdx.x % 8 takes the next id of queue x with one atomicAdd and, when its own queue is empty, of the others in turn.  A
// workgroup that starts late or not at all costs only what it would have done itself.  Which workgroup computes a tile block
// does not enter its arithmetic: results are bit-identical.
//   ctr[0..7] = next index of queue x, ctr[8] = workgroups that have left; all zero between launches: the last workgroup to
//   leave resets them (persistent_work_counters, kernels.h: one set per stream, and launches on a stream do not overlap).
__device__ __forceinline__ int wq_base(int first, int x) { return first + ((x - first) & 7); }
// next tile block in [first, end) for a workgroup of XCD x0, trying the queues x0 + d0, x0 + d0 + 1, ...; -1: none left
__device__ __forceinline__ int wq_claim(int* ctr, int x0, int d0, int first, int end) {
    for (int d = d0; d < 8; ++d) {
        const int y = (x0 + d) & 7;
        const int lt = wq_base(first, y) + 8 * atomicAdd(ctr + y, 1);
        if (lt < end) return lt;
    }
    return -1;
}
// the first TWO tile blocks of a workgroup with one round trip (its own queue's next two; each falls back to wq_claim when the queue
// is empty); c[1] = -1 if nothing is left
__device__ __forceinline__ void wq_claim2(int* ctr, int x0, int first, int end, int& c0, int& c1) {
    const int k = atomicAdd(ctr + x0, 2);
    c0 = wq_base(first, x0) + 8 * k;
    c1 = c0 + 8;
    if (c0 >= end) c0 = wq_claim(ctr, x0, 1, first, end);
    if (c0 < 0) {
        c1 = -1;
        return;
    }
    if (c1 >= end) c1 = wq_claim(ctr, x0, 1, first, end);
}
__device__ __forceinline__ void wq_leave(int* ctr, int grid) {
    if (atomicAdd(ctr + 8, 1) == grid - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicExch(ctr + i, 0);
    }
}

}  // namespace l3
