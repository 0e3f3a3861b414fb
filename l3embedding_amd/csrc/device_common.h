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

}  // namespace l3
