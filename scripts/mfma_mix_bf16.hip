// The bf16 counterpart of mfma_mix.hip: how much do interleaved fp32 VALU instructions cost v_mfma_f32_32x32x16_bf16 on gfx950?
// (For the fp32 MFMA every VALU instruction takes ~4.3 cycles of matrix time; this asks whether the bf16 matrix pipe runs beside them.)
// Each wave runs groups of 4 independent MFMAs (8 passes each) plus NV VALU; `pk` = v_pk_fma_f32 instead of v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + threadIdx.x * 1e-3f); bv[i] = (__bf16)b; }
    float v0 = a, v1 = b, v2 = a * b, v3 = a - b;
    f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a * b, a};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));
            else    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));
        }
    }
    float s = v0 + v3 + p0[0] + p0[1];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the fp32 MFMA with packed VALU beside it (mfma_mix.hip measured the plain one)
template <int NV, bool PK>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    float v0 = a, v1 = b, v2 = a * b;
    f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a * b, a};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));
            else    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));
        }
    }
    float s = v0 + p0[0] + p0[1];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
float best_ms(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <int NV, bool PK>
void run(float* d, int w) {
    const int blocks = 256 * w, iters = 4000;
    float ms = best_ms([&] { hipLaunchKernelGGL((k<NV, PK>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f); });
    double flops = (double)blocks * 4 * iters * 4.0 * 32768.0;
    // cycles per group of 4 MFMAs per SIMD at the clock the fp32 study found (2.1 GHz under load is typical): report time per group
    printf("bf16 MFMA  waves/SIMD %d  per 4 MFMA: %2d %s VALU -> %7.1f TFLOP/s  (%.1f ns per group per wave-slot)\n", w, NV, PK ? "packed" : "plain ",
           flops / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * (double)w));
}
template <int NV, bool PK>
void run32(float* d, int w) {
    const int blocks = 256 * w, iters = 4000;
    float ms = best_ms([&] { hipLaunchKernelGGL((k32<NV, PK>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f); });
    double flops = (double)blocks * 4 * iters * 4.0 * 4096.0;
    printf("fp32 MFMA  waves/SIMD %d  per 4 MFMA: %2d %s VALU -> %7.1f TFLOP/s\n", w, NV, PK ? "packed" : "plain ", flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0, false>(d, w); run<4, false>(d, w); run<8, false>(d, w); run<16, false>(d, w); run<32, false>(d, w);
        run<4, true>(d, w); run<8, true>(d, w); run<16, true>(d, w);
    }
    for (int w : {1, 2, 4}) {
        run32<0, false>(d, w); run32<8, false>(d, w); run32<16, false>(d, w); run32<8, true>(d, w); run32<16, true>(d, w);
    }
    return 0;
}
