// What do the split's VALU instructions cost beside v_mfma_f32_32x32x16_bf16?  (round 5: conv_wino_bx6.hip runs ~8 VALU per MFMA and takes
// ~95 SIMD cycles per MFMA.)  Each wave: groups of 4 independent MFMAs + NV VALU instructions of one kind on 8 independent registers
// (no dependent chain shorter than 8 instructions).  W waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/vb valu_beside_bf16_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { SUB, AND_LIT, AND_SGPR, PERM, CVTPK, LSHL, MIX };
template <int KIND, int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, unsigned msk, unsigned sel) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + threadIdx.x * 1e-3f); bv[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a * (i + 1) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MF) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c3, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float& x = v[j & 7];
            float& y = v[(j + 3) & 7];
            const int kind = KIND == MIX ? (j % 11 < 4 ? SUB : j % 11 < 8 ? AND_LIT : PERM) : KIND;
            if (kind == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (kind == AND_LIT) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
            if (kind == AND_SGPR) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(msk));
            if (kind == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(sel));
            if (kind == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            if (kind == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
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
template <int KIND, int NV, bool MF>
void run(float* d, int w, const char* name) {
    const int blocks = 256 * w, iters = 4000;
    float ms = best_ms([&] { hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f, 0xffff0000u, 0x07060302u); });
    // ns per group (4 MFMA + NV VALU) per SIMD: every SIMD runs w waves x iters groups
    printf("%-9s waves/SIMD %d  %s4 MFMA + %2d VALU: %7.1f ns per group and SIMD  (%.2f ns per VALU and SIMD if alone)\n", name, w, MF ? "" : "(no MFMA) ", NV,
           ms * 1e6 / (iters * (double)w), NV ? ms * 1e6 / (iters * (double)w * NV) : 0.0);
}
#define ALL(KIND, NAME)                                                                                                     \
    for (int w : {2, 4}) {                                                                                                  \
        run<KIND, 32, false>(d, w, NAME); run<KIND, 8, true>(d, w, NAME); run<KIND, 16, true>(d, w, NAME); run<KIND, 32, true>(d, w, NAME); \
    }
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 4}) run<SUB, 0, true>(d, w, "none");
    ALL(SUB, "v_sub_f32") ALL(AND_LIT, "and lit") ALL(AND_SGPR, "and sgpr") ALL(PERM, "v_perm") ALL(CVTPK, "cvt_pk") ALL(LSHL, "lshl") ALL(MIX, "mix")
    return 0;
}
