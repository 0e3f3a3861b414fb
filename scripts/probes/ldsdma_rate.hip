// Developer probe: what rate does one CU sustain for buffer_load_dwordx4 ... lds (1 KiB per wave-instruction) out of an
// L2-resident buffer, with every CU doing the same?  Compared with the same bytes loaded into registers.
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate scripts/probes/ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 12, THREADS = WAVES * 64;
template <int MODE>   // 0: LDS-DMA, 1: into registers
__global__ __launch_bounds__(THREADS) void k(const float* src, size_t span_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)span_bytes, 0x00020000);
    // every block sweeps the same span (as every tile block re-reads the same filter slices), from a different start
    unsigned off = (unsigned)(((size_t)blockIdx.x * 9973 * 1024 + (size_t)wave * 1024) % span_bytes);
    f32x4 accv = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + (wave * 9 + q) * 256),
                                                         16, (int)(off + lane * 16), 0, 0, 0);
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(off + lane * 16), 0, 0));
                accv += v;
            }
            off += WAVES * 1024;
            if (off >= span_bytes) off -= (unsigned)span_bytes;
        }
        if (MODE == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | 0x0);   // vmcnt(0)
        }
        __builtin_amdgcn_s_barrier();
    }
    if (MODE == 0) {
        __syncthreads();
        accv[0] = smem[t];
    }
    if (accv[0] + accv[1] + accv[2] + accv[3] == 12345.678f) sink[0] = 1.f;
}
int main() {
    const int CUS = 256;
    for (size_t span : {(size_t)576 << 10, (size_t)4 << 20, (size_t)36 << 20}) {
        float *src, *sink;
        hipMalloc(&src, span);
        hipMalloc(&sink, 4);
        hipMemset(src, 0, span);
        hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 108 * 1024);
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = 2000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0)
                    k<0><<<CUS, THREADS, 108 * 1024>>>(src, span, iters, sink);
                else
                    k<1><<<CUS, THREADS, 108 * 1024>>>(src, span, iters, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)CUS * iters * 9 * WAVES * 1024;
            printf("span %6zu KiB  %-9s  %.3f ms  %.2f TB/s aggregate  %.1f GB/s per CU  (%.1f B/clk/CU at 2.4 GHz); per 108-KiB stage %.2f us\n",
                   span >> 10, mode == 0 ? "LDS-DMA" : "registers", ms, bytes / ms / 1e9, bytes / ms / 1e6 / CUS,
                   bytes / (ms * 1e-3) / CUS / 2.4e9, ms * 1e3 / iters);
        }
        hipFree(src);
        hipFree(sink);
    }
    return 0;
}
