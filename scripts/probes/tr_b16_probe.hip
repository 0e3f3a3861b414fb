// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds 16-bit counters (value = element index), every
// lane passes its own 8-byte-aligned address, and the four 16-bit values each lane receives are printed.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe scripts/probes/tr_b16_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int stride, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int elem;
    if (mode == 0) elem = (i >> 2) * stride + (i & 3) * 4 + g * 4 * stride;   // lanes 4r..4r+3 of a group -> row r, 4 cols each
    else if (mode == 1) elem = i * stride + g * 16 * stride;                  // lane i -> row i, cols 0..3
    else elem = (i & 3) * stride + (i >> 2) * 4 + g * 4 * stride;             // lanes i%4 -> row, i/4 -> col quad
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + elem));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        const int stride = 64;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (row stride %d elements; value = row*%d + col):\n", mode, stride, stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
