// Do LDS reads cost the fp32 MFMA issue time on gfx950?  Groups of 4 independent v_mfma_f32_32x32x2_f32 plus NL LDS reads
// (ds_read_b128 or ds_read_b64, conflict free) whose results are consumed by one v_add per group (kept alive), and, for scale,
// the same number of plain VALU instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NL, int WIDE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) float sm[256 * 4 * 4];
    for (int i = threadIdx.x; i < 256 * 16; i += 256) sm[i] = a * i;
    __syncthreads();
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    const float* p = sm + threadIdx.x * 4;
    float keep = 0.f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if (WIDE == 4) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"((j & 3) * 4096));
                asm volatile("" :: "v"(v));
            } else {
                f32x2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"((j & 3) * 4096));
                asm volatile("" :: "v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = keep;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NL, int WIDE>
void run(float* d, int w) {
    const int blocks = 256 * w, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NL, WIDE>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 4.0 * 4096.0;
    printf("waves/SIMD %d  per 4 MFMA: %2d ds_read_b%d -> %.1f TFLOP/s\n", w, NL, WIDE == 4 ? 128 : 64, flops / (best * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4}) {
        run<0, 4>(d, w); run<2, 4>(d, w); run<4, 4>(d, w); run<8, 4>(d, w); run<16, 4>(d, w);
        run<4, 2>(d, w); run<8, 2>(d, w); run<16, 2>(d, w);
    }
    return 0;
}
