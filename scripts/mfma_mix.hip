// How much MFMA issue bandwidth do interleaved non-MFMA instructions cost on gfx950?
// Each wave runs groups of 4 independent v_mfma_f32_32x32x2_f32 plus NV VALU / NS SALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    float v0 = a, v1 = b, v2 = a * b, v3 = a - b;
    int s0 = iters, s1 = 3;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            asm volatile("s_add_i32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
        }
    }
    float s = v0 + v3 + (float)s0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NS>
void run(float* d, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NS>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 4.0 * 4096.0;
    printf("waves/SIMD %d  per 4 MFMA: %2d VALU %2d SALU  -> %.1f TFLOP/s\n", blocks_per_cu, NV, NS, flops / (best * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0, 0>(d, w); run<4, 0>(d, w); run<8, 0>(d, w); run<16, 0>(d, w); run<0, 8>(d, w); run<0, 16>(d, w); run<8, 8>(d, w); run<16, 16>(d, w);
    }
    return 0;
}
