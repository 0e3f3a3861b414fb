// stream_shapes -- what shape of streaming kernel reaches which HBM rate on this chip: y = relu(x * s[c] + t[c]) over a bf16 (or fp32)
// NHWC tensor of the size of the towers' first-block activations, with 8- or 16-byte accesses per lane, 1 / 2 / 4 / 8 of them in
// flight per thread, and grids from 1024 to 16384 blocks (the shapes bn_fused.hip's kernels could take).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/stream_shapes.hip -o scripts/probes/stream_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 h; h[0] = (__bf16)a; h[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float lo16(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi16(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// W = 32-bit words per lane and access (2 = 8 B = 4 bf16, 4 = 16 B = 8 bf16), U = accesses in flight per thread, NT = non-temporal
template <int W, int U, bool NT>
__global__ __launch_bounds__(256) void bf16_apply(const unsigned* __restrict__ x, unsigned* __restrict__ y, const float* __restrict__ sc,
                                                  const float* __restrict__ sh, int64_t n, int CW) {     // n = accesses, CW = accesses per pixel row
    typedef unsigned vec __attribute__((ext_vector_type(W)));
    const int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)(q0 % CW) * (2 * W);
    float s[2 * W], t[2 * W];
#pragma unroll
    for (int e = 0; e < 2 * W; ++e) { s[e] = sc[c + e]; t[e] = sh[c + e]; }
    const vec* xv = reinterpret_cast<const vec*>(x);
    vec* yv = reinterpret_cast<vec*>(y);
    int64_t q = q0;
    for (; q + (U - 1) * stride < n; q += U * stride) {
        vec r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(xv + q + u * stride) : xv[q + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vec o;
#pragma unroll
            for (int w = 0; w < W; ++w) o[w] = pack2(fmaxf(fmaf(lo16(r[u][w]), s[2 * w], t[2 * w]), 0.f), fmaxf(fmaf(hi16(r[u][w]), s[2 * w + 1], t[2 * w + 1]), 0.f));
            if (NT) __builtin_nontemporal_store(o, yv + q + u * stride); else yv[q + u * stride] = o;
        }
    }
    for (; q < n; q += stride) {
        vec r = xv[q], o;
#pragma unroll
        for (int w = 0; w < W; ++w) o[w] = pack2(fmaxf(fmaf(lo16(r[w]), s[2 * w], t[2 * w]), 0.f), fmaxf(fmaf(hi16(r[w]), s[2 * w + 1], t[2 * w + 1]), 0.f));
        yv[q] = o;
    }
}
// blocked: the U accesses of a thread are 256 apart (a block covers U * 256 consecutive accesses), grid-stride by gridDim * U * 256
template <int W, int U, bool NT>
__global__ __launch_bounds__(256) void bf16_apply_blk(const unsigned* __restrict__ x, unsigned* __restrict__ y, const float* __restrict__ sc,
                                                      const float* __restrict__ sh, int64_t n, int CW) {
    typedef unsigned vec __attribute__((ext_vector_type(W)));
    const int c = (int)(threadIdx.x % CW) * (2 * W);
    float s[2 * W], t[2 * W];
#pragma unroll
    for (int e = 0; e < 2 * W; ++e) { s[e] = sc[c + e]; t[e] = sh[c + e]; }
    const vec* xv = reinterpret_cast<const vec*>(x);
    vec* yv = reinterpret_cast<vec*>(y);
    const int64_t stride = (int64_t)gridDim.x * U * 256;
    for (int64_t q = (int64_t)blockIdx.x * U * 256 + threadIdx.x; q < n; q += stride) {
        vec r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (q + u * 256 < n) r[u] = NT ? __builtin_nontemporal_load(xv + q + u * 256) : xv[q + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vec o;
#pragma unroll
            for (int w = 0; w < W; ++w) o[w] = pack2(fmaxf(fmaf(lo16(r[u][w]), s[2 * w], t[2 * w]), 0.f), fmaxf(fmaf(hi16(r[u][w]), s[2 * w + 1], t[2 * w + 1]), 0.f));
            if (q + u * 256 < n) { if (NT) __builtin_nontemporal_store(o, yv + q + u * 256); else yv[q + u * 256] = o; }
        }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void f32_apply_blk(const f32x4* __restrict__ x, f32x4* __restrict__ y, const f32x4* __restrict__ sc,
                                                     const f32x4* __restrict__ sh, int64_t n, int C4) {
    const f32x4 s = sc[threadIdx.x % C4], t = sh[threadIdx.x % C4];
    const int64_t stride = (int64_t)gridDim.x * U * 256;
    for (int64_t q = (int64_t)blockIdx.x * U * 256 + threadIdx.x; q < n; q += stride) {
        f32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (q + u * 256 < n) r[u] = NT ? __builtin_nontemporal_load(x + q + u * 256) : x[q + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(r[u][e], s[e], t[e]), 0.f);
            if (q + u * 256 < n) { if (NT) __builtin_nontemporal_store(o, y + q + u * 256); else y[q + u * 256] = o; }
        }
    }
}
// fp32 in / fp32 out, 16 B per access
template <int U, bool NT>
__global__ __launch_bounds__(256) void f32_apply(const f32x4* __restrict__ x, f32x4* __restrict__ y, const f32x4* __restrict__ sc,
                                                 const f32x4* __restrict__ sh, int64_t n, int C4) {
    const int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const f32x4 s = sc[q0 % C4], t = sh[q0 % C4];
    int64_t q = q0;
    for (; q + (U - 1) * stride < n; q += U * stride) {
        f32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(x + q + u * stride) : x[q + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(r[u][e], s[e], t[e]), 0.f);
            if (NT) __builtin_nontemporal_store(o, y + q + u * stride); else y[q + u * stride] = o;
        }
    }
    for (; q < n; q += stride) {
        f32x4 r = x[q], o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(r[e], s[e], t[e]), 0.f);
        y[q] = o;
    }
}

template <typename F>
static double time_us(F f, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128, C = 64;
    const int64_t elems = (int64_t)N * 224 * 224 * C;
    void *x, *y; float *sc, *sh;
    CK(hipMalloc(&x, elems * 4)); CK(hipMalloc(&y, elems * 4)); CK(hipMalloc(&sc, 4096)); CK(hipMalloc(&sh, 4096));
    CK(hipMemset(x, 0x3c, elems * 4)); CK(hipMemset(sc, 0, 4096)); CK(hipMemset(sh, 0, 4096));
    printf("# %d x 224 x 224 x %d: bf16 tensor %.0f MB (read + write %.2f GB), fp32 tensor %.0f MB\n", N, C, elems * 2e-6, elems * 4e-9, elems * 4e-6);
    const int grids[] = {1024, 2048, 4096, 8192, 16384, 3840, 15360, 2000, 100000};
#define BF(W, U, NT) for (int g : grids) { const int64_t n = elems / (2 * W); const double us = time_us([&] { hipLaunchKernelGGL((bf16_apply<W, U, NT>), dim3(g), dim3(256), 0, 0, (const unsigned*)x, (unsigned*)y, sc, sh, n, C / (2 * W)); }); \
        printf("bf16 %2d B/lane x %d in flight %s grid %5d: %7.1f us  %.2f TB/s\n", 4 * W, U, NT ? "nt" : "  ", g, us, elems * 4e-6 / us); }
    BF(2, 1, true) BF(2, 4, true) BF(2, 8, true)
#define BFB(W, U, NT) for (int g : grids) { const int64_t n = elems / (2 * W); const double us = time_us([&] { hipLaunchKernelGGL((bf16_apply_blk<W, U, NT>), dim3(g), dim3(256), 0, 0, (const unsigned*)x, (unsigned*)y, sc, sh, n, C / (2 * W)); }); \
        printf("bf16 %2d B/lane x %d blocked   %s grid %6d: %7.1f us  %.2f TB/s\n", 4 * W, U, NT ? "nt" : "  ", g, us, elems * 4e-6 / us); }
    BFB(2, 2, true) BFB(2, 4, true) BFB(2, 8, true) BFB(4, 2, true) BFB(4, 4, true)
#define F32B(U, NT) for (int g : grids) { const int64_t n = elems / 4; const double us = time_us([&] { hipLaunchKernelGGL((f32_apply_blk<U, NT>), dim3(g), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, (const f32x4*)sc, (const f32x4*)sh, n, C / 4); }); \
        printf("f32  16 B/lane x %d blocked   %s grid %6d: %7.1f us  %.2f TB/s\n", U, NT ? "nt" : "  ", g, us, elems * 8e-6 / us); }
    F32B(2, true) F32B(4, true)
#define F32(U, NT) for (int g : grids) { const int64_t n = elems / 4; const double us = time_us([&] { hipLaunchKernelGGL((f32_apply<U, NT>), dim3(g), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, (const f32x4*)sc, (const f32x4*)sh, n, C / 4); }); \
        printf("f32  16 B/lane x %d in flight %s grid %5d: %7.1f us  %.2f TB/s\n", U, NT ? "nt" : "  ", g, us, elems * 8e-6 / us); }
    F32(1, true) F32(2, true) F32(4, true)
    return 0;
}
