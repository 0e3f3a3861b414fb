// Developer probe: what does an LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB) cost a SIMD that is busy with fp32 MFMAs?
// 12 waves per CU (3 per SIMD), every wave: P pieces then 24 independent-accumulator v_mfma_f32_32x32x2_f32 per iteration,
// one barrier per iteration (the stage shape of conv_wino4).  Reported: cycles per iteration against the 4608 the MFMAs need.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_vs_mfma scripts/probes/dma_vs_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WAVES = 12, THREADS = WAVES * 64;
// MODE 0: every wave issues P pieces.  MODE 1: only waves 0..3 issue 3 P pieces each (same total).  MODE 2: pieces replaced by
// P plain buffer_load_dwordx4 into registers (per wave).  BAR: barrier + vmcnt(0) per iteration or only vmcnt(0)
template <int P, int MODE, int BAR>
__global__ __launch_bounds__(THREADS) void k(const float* src, size_t span_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)span_bytes, 0x00020000);
    unsigned off = (unsigned)(((size_t)blockIdx.x * 9973 * 1024 + (size_t)wave * 1024) % span_bytes);
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = (float)lane, bv = 1.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int np = MODE == 1 ? (wave < 4 ? 3 * P : 0) : (MODE == 3 ? 0 : P);
#pragma unroll
        for (int q = 0; q < (MODE == 1 ? 3 * P : P); ++q) {
            if (q < np) {
                if (MODE == 2) {
                    keep += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(off + lane * 16), 0, 0));
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + (wave * 9 + (q % 9)) * 256),
                                                             16, (int)(off + lane * 16), 0, 0, 0);
                }
                off += WAVES * 1024;
                if (off >= span_bytes) off -= (unsigned)span_bytes;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MODE == 3 && P > 0 && m % (24 / P) == 0 && m / (24 / P) < P) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + (wave * 9 + ((m / (24 / P)) % 9)) * 256),
                                                         16, (int)(off + lane * 16), 0, 0, 0);
                off += WAVES * 1024;
                if (off >= span_bytes) off -= (unsigned)span_bytes;
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[m % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m % 6], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 2) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = keep[0] + keep[1] + keep[2] + keep[3];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = smem[t];
}
template <int P, int MODE, int BAR>
void run(const float* src, size_t span, float* sink, const char* what) {
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void*)k<P, MODE, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 108 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        k<P, MODE, BAR><<<256, THREADS, 108 * 1024>>>(src, span, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s P=%2d  %.3f us/iter (MFMA alone would be %.3f us at 2.4 GHz)\n", what, P, ms * 1e3 / iters, 4608 / 2400.0);
}
int main() {
    const size_t span = (size_t)576 << 10;
    float *src, *sink;
    (void)hipMalloc(&src, span);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(src, 0, span);
    run<0, 0, 1>(src, span, sink, "no loads, barrier");
    run<0, 0, 0>(src, span, sink, "no loads, no barrier");
    run<3, 0, 1>(src, span, sink, "LDS-DMA, every wave, barrier");
    run<6, 0, 1>(src, span, sink, "LDS-DMA, every wave, barrier");
    run<9, 0, 1>(src, span, sink, "LDS-DMA, every wave, barrier");
    run<18, 0, 1>(src, span, sink, "LDS-DMA, every wave, barrier");
    run<9, 0, 0>(src, span, sink, "LDS-DMA, every wave, no barrier");
    run<9, 1, 1>(src, span, sink, "LDS-DMA, waves 0-3 issue all (27 each), barrier");
    run<3, 1, 1>(src, span, sink, "LDS-DMA, waves 0-3 issue all (9 each), barrier");
    run<6, 3, 1>(src, span, sink, "LDS-DMA spread between MFMAs, barrier");
    run<8, 3, 1>(src, span, sink, "LDS-DMA spread between MFMAs, barrier");
    run<12, 3, 1>(src, span, sink, "LDS-DMA spread between MFMAs, barrier");
    run<8, 3, 0>(src, span, sink, "LDS-DMA spread between MFMAs, no barrier");
    run<9, 2, 1>(src, span, sink, "register loads, every wave, barrier");
    run<18, 2, 1>(src, span, sink, "register loads, every wave, barrier");
    return 0;
}
