// conv_first.hip -- forward of the FIRST convolution of a tower: 3x3 'same', 1 or 3 input channels, 64 filters
// (l3embedding/audio_model.py:376-378 on the (256,199,1) mel spectrogram, vision_model.py:130-132 on the
// (224,224,3) frame).
//
// K = 9 or 27: as an implicit GEMM this layer pads K to 16 / 32 and spends its time writing the 64-channel
// full-resolution output (the largest activation of the network) -- it is HBM-bound, not MFMA-bound.  So it is a
// plain fp32 FMA kernel shaped for the store stream:
//   lane   = output channel (64), the 9 x Cin filter taps of that channel live in registers;
//   wave   = runs of 32 output pixels along an image row; the 3 x 34 x Cin input window of a run goes to a
//            wave-private LDS slab once, and the pixel loop keeps a sliding 3 x 3 x Cin window in registers: one new
//            column (3 x Cin broadcast ds_read_b32) + 9 x Cin FMAs per pixel;
//   store  = one 256-B (fp32) or 128-B (bf16) row of 64 channels per pixel, fully coalesced;
//   fused  = the BatchNorm statistic partials of the output (sum, sum of squares about the pivot bias[c]; of
//            relu(output) for the ReLU->BN order) in the bn_fused.hip partial layout, one block per workgroup --
//            the separate statistics pass over this tensor disappears;
//   bf16   = in L3_DTYPE_BF16 engines the output is stored as bfloat16 (oracle mixed-precision rule (2)) and the
//            statistics are those of the rounded values.
#include "kernels.h"
#include "device_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace l3 {

namespace {

struct FirstArgs {
    const float* x;       // (N, H, W, CIN) fp32
    const float* w;       // (3, 3, CIN, 64) keras HWIO
    const float* bias;
    void* y;              // (N, H, W, 64) fp32 or bf16
    int N, H, W;
    int segs;             // 32-pixel runs per image row
    int units;            // N * H * segs
    int per_wave;         // runs per wave
    float* stat_part;
    int stat_mode;
};

constexpr int RUN = 32;

template <int CIN, bool STATS, bool OBF>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(FirstArgs a) {
    constexpr int ROWF = (RUN + 2) * CIN;                      // floats per window row
    __shared__ float slab[4][3 * ROWF];
    __shared__ float red[2][4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* S = slab[wave];

    float wt[9][CIN];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wt[t9][ci] = a.w[(t9 * CIN + ci) * 64 + lane];
    const float bz = a.bias != nullptr ? a.bias[lane] : 0.f;
    const bool srelu = STATS && a.stat_mode == 2;
    const float pivot = srelu ? fmaxf(bz, 0.f) : bz;
    float s0 = 0.f, s1 = 0.f;

    const int gw = blockIdx.x * 4 + wave;
    const int u_begin = gw * a.per_wave;
    const int u_end = min(a.units, u_begin + a.per_wave);
    for (int u = u_begin; u < u_end; ++u) {
        const int seg = u % a.segs, row = u / a.segs;           // row = n * H + y
        const int yy = row % a.H;
        const int x0 = seg * RUN;
        // ---- window rows y-1..y+1, pixels x0-1..x0+32, into the slab (zeros outside the image) ----
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int gy = yy - 1 + r;
            const bool rok = (unsigned)gy < (unsigned)a.H;
            const float* src = a.x + ((size_t)(row - 1 + r) * a.W) * CIN;        // row (n, gy), valid when rok
            for (int f = lane; f < ROWF; f += 64) {
                const int gx = x0 - 1 + f / CIN;
                S[r * ROWF + f] = (rok && (unsigned)gx < (unsigned)a.W) ? src[(size_t)(x0 - 1) * CIN + f] : 0.f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the slab is wave-private
        __builtin_amdgcn_wave_barrier();
        // ---- sliding window: win[r][c][ci], c = 0..2 <-> pixels px-1, px, px+1 ----
        float win[3][3][CIN];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) win[r][c + 1][ci] = S[r * ROWF + c * CIN + ci];
        const size_t out_row = ((size_t)row * a.W + x0) * 64 + lane;
        const int npx = min(RUN, a.W - x0);
        // fully unrolled: the window rotation becomes register renaming instead of 6 x CIN moves per pixel
#pragma unroll
        for (int px = 0; px < RUN; ++px) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    win[r][0][ci] = win[r][1][ci];
                    win[r][1][ci] = win[r][2][ci];
                    win[r][2][ci] = S[r * ROWF + (px + 2) * CIN + ci];
                }
            float acc = bz;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc = fmaf(win[r][c][ci], wt[r * 3 + c][ci], acc);
            if (px < npx) {                    // wave-uniform: the last run of a row may be short
                float yv = acc;
                if constexpr (OBF) {
                    const __bf16 h = (__bf16)acc;
                    reinterpret_cast<__bf16*>(a.y)[out_row + (size_t)px * 64] = h;
                    yv = (float)h;
                } else {
                    __builtin_nontemporal_store(acc, reinterpret_cast<float*>(a.y) + out_row + (size_t)px * 64);
                }
                if constexpr (STATS) {
                    const float d = (srelu ? fmaxf(yv, 0.f) : yv) - pivot;
                    s0 += d;
                    s1 = fmaf(d, d, s1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();       // the next run overwrites the slab
    }
    if constexpr (STATS) {
        red[0][wave][lane] = s0;
        red[1][wave][lane] = s1;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
            a.stat_part[((size_t)blockIdx.x * 2 + which) * 64 + ch] =
                (red[which][0][ch] + red[which][1][ch]) + (red[which][2][ch] + red[which][3][ch]);
        }
    }
}

#ifdef L3_EXPERIMENTS
// x = h1 + h2 + h3 exactly, three bfloat16 terms (round to nearest even, the remainder is exact in fp32), two values per
// v_cvt_pk_bf16_f32: the packed words are the MFMA operand registers as they are
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float xa, float xb, unsigned& h1, unsigned& h2, unsigned& h3) {
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{xa, xb}, bf16x2s));
    const float ra = xa - __uint_as_float(h1 << 16), rb = xb - __uint_as_float(h1 & 0xffff0000u);
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{ra, rb}, bf16x2s));
    const float sa = ra - __uint_as_float(h2 << 16), sb = rb - __uint_as_float(h2 & 0xffff0000u);
    h3 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{sa, sb}, bf16x2s));
}
__device__ __forceinline__ void split3_oct(const float (&x)[8], bf16x8& t1, bf16x8& t2, bf16x8& t3) {
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pair(x[2 * q], x[2 * q + 1], a[q], b[q], c[q]);
    t1 = __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], a[2], a[3]});
    t2 = __builtin_bit_cast(bf16x8, u32x4{b[0], b[1], b[2], b[3]});
    t3 = __builtin_bit_cast(bf16x8, u32x4{c[0], c[1], c[2], c[3]});
}

// Matrix-core form (round 6, second session; an experiment, see the end of this comment): the FMA kernel above is VALU-bound (27 v_fmac per output; 442 us where the 822-MB bf16 store of
// the vision tower's first layer takes 140 at 128 pairs).  An earlier attempt on the fp32 MFMA was latency-bound at two waves per SIMD and an
// fp32 MFMA shares the vector ALUs anyway (profiles/r06_first_conv_mfma.txt).  Here the GEMM runs on the bf16 matrix pipe -- beside
// which VALU work is free -- with fp32-grade products: x and w are split EXACTLY into three bfloat16 terms each (x = x1 + x2 + x3, 8 + 8 + 8
// mantissa bits) and the six products x_i w_j with i + j <= 4 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first (the dropped
// terms are below 2^-24 of |x||w|: the error model of round 5's split-operand convolutions, scripts/wino_split_error_model.py).
//   wave   = runs of 32 pixels (M) x 64 filters (two N tiles), K = 9 Cin padded to 16 / 32: k = tap * Cin + ci, so the 9 (Cin = 1) or 27
//            (Cin = 3) operands of pixel p are, per window row dy, the Cin * 3 CONSECUTIVE floats of the slab from 3 p on: a lane's sixteen
//            operand addresses are fixed for the whole kernel (the padding k read a zero word);
//   B      = the three terms of the filter in registers for the whole kernel;
//   out    = the 32 x 64 tile through a wave-private LDS transpose as 8-byte (bf16) / 16-byte (fp32) stores, bias, BatchNorm partials of
//            the stored values per lane (four filters) and one cross-lane fold at the end;
//   x      = no slab: a lane gathers its sixteen operands with buffer loads (out-of-image and padding k = out-of-range offset = 0), and the
//            loads of run u + 1 are issued before run u is multiplied -- with the slab of the FMA kernel in front of every run (fill, wait,
//            read back) this form was SLOWER than the FMA kernel (547 against 436 us), and so was the fp32-MFMA attempt before it.
// Measured (rocprofv3 stats of the serialised step, us per launch, same box): 3 channels 227-260 -> 179 (fp32, 64 pairs), 436-449 -> 331
// (bf16 output, 128 pairs); 1 channel 156-168 -> 151 / 225-229 -> 236.  In the two-tower step that is +0.4 % at 128 pairs bf16 and nothing
// measurable at 64 pairs fp32 (31.15-31.19 against 31.16-31.39 ms), and the batch-1 golden of cnn_L3_orig -- where every BatchNorm
// normalises over ONE sample and a last-bit difference in the first layer moves ReLU masks downstream -- lands 0.39 of a tensor's RMS from
// float64 where the FMA chain lands 0.08 (bound 0.25; the fp32 NumPy oracle itself: 0.31): NOT the product path, built only with
// L3_BUILD_EXPERIMENTS=1 (L3_FIRST_FWD_X6=1 selects it; tests/test_parity_gpu.py test_first_convolution_forward[x6]).  168 registers = three waves per SIMD; a run is still ~3 300 cycles of a wave (split 100 VALU, 24 MFMAs on two chains, tile
// write / read back, ~200 VALU of bias / rounding / partials / stores) -- the layer's 822-MB store would take 140 us.  Against float64 the
// split products are closer than the FMA chain: 1.1e-7 against 2.1-3.8e-7 of the output range (scripts: tests' first-convolution cases).
template <int CIN, bool STATS, bool OBF>
__global__ __launch_bounds__(256, 3) void conv_first_fwd_x6_kernel(FirstArgs a) {
    constexpr int KS = CIN == 3 ? 2 : 1;                       // k-steps of 16
    __shared__ __attribute__((aligned(16))) float tile[4][32 * 64];
    __shared__ float red[2][4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, hi32 = lane >> 5;
    float* Es = tile[wave];

    // operand k = 16 s4 + 8 hi32 + j of pixel l31 = tap (dy, dx), channel ci with k = (3 dy + dx) Cin + ci: the lane gathers its
    // sixteen operands straight from the image (out-of-image = out-of-range buffer offset = 0; the padding k too) -- the loads of run
    // u + 1 are issued before run u is multiplied, so that no phase of a run waits for memory
    bf16x8 w1[KS][2], w2[KS][2], w3[KS][2];
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s4 + 8 * hi32 + j;
                wv[j] = k < 9 * CIN ? a.w[k * 64 + jn * 32 + l31] : 0.f;
            }
            split3_oct(wv, w1[s4][jn], w2[s4][jn], w3[s4][jn]);
        }
    }
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * CIN * 4), 0x00020000);
    const int c4 = (lane & 15) * 4;
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bz = *reinterpret_cast<const f32x4*>(a.bias + c4);
    const bool srelu = STATS && a.stat_mode == 2;
    const float lo = srelu ? 0.f : -__builtin_inff();
    f32x4 pivot;
#pragma unroll
    for (int e = 0; e < 4; ++e) pivot[e] = fmaxf(bz[e], lo);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};

    const int gw = blockIdx.x * 4 + wave;
    const int u_begin = gw * a.per_wave;
    const int u_end = min(a.units, u_begin + a.per_wave);
    // the sixteen operands of run u (a run past the wave's range loads zeros)
    auto gather = [&](int u, float (&xv)[KS][8]) {
        const bool live = u < u_end;
        const int seg = u % a.segs, row = u / a.segs;           // row = n * H + y
        const int yy = row % a.H, x0 = seg * RUN;
        const int gx = x0 + l31;
        const int base = (row * a.W + gx) * CIN * 4;
        const int wc4 = a.W * CIN * 4;
        // window row dy is inside the image (wave-uniform), window column dx is (per lane)
        const bool rowok[3] = {live && yy >= 1, live, live && yy + 1 < a.H};
        const bool colok[3] = {gx >= 1 && gx - 1 < a.W, gx < a.W, gx + 1 < a.W};
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // the slot's two candidates (k of the lower / upper half of the wave) are compile-time: tap, channel, offsets
                const int kA = 16 * s4 + j, kB = kA + 8;
                const int tA = kA / CIN, tB = kB / CIN;
                const int dyA = tA / 3, dxA = tA - 3 * dyA, dyB = tB / 3, dxB = tB - 3 * dyB;
                const bool okA = kA < 9 * CIN && rowok[dyA < 3 ? dyA : 0] && colok[dxA];
                const bool okB = kB < 9 * CIN && rowok[dyB < 3 ? dyB : 0] && colok[dxB];
                const int oA = (dyA - 1) * wc4 + ((dxA - 1) * CIN + (kA - tA * CIN)) * 4;
                const int oB = (dyB - 1) * wc4 + ((dxB - 1) * CIN + (kB - tB * CIN)) * 4;
                const bool ok = hi32 ? okB : okA;
                const unsigned vo = ok ? (unsigned)(base + (hi32 ? oB : oA)) : 0x80000000u;
                xv[s4][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrd, (int)vo, 0, 0));
            }
    };
    float xn[KS][8];
    gather(u_begin, xn);
    for (int u = u_begin; u < u_end; ++u) {
        const int seg = u % a.segs, row = u / a.segs;
        const int x0 = seg * RUN;
        // ---- this run's operands, split into three bfloat16 terms; then the next run's loads ----
        bf16x8 x1[KS], x2[KS], x3[KS];
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) split3_oct(xn[s4], x1[s4], x2[s4], x3[s4]);
        gather(u + 1, xn);
        // smallest products first; the two N tiles are two independent MFMA chains
        f32x16 acc[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jn][r] = 0.f;
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const bf16x8 xa = term == 0 ? x3[s4] : (term == 1 || term == 3) ? x2[s4] : x1[s4];
                    const bf16x8 wb = (term == 0 || term == 3 || term == 5) ? w1[s4][jn] : (term == 1 || term == 4) ? w2[s4][jn] : w3[s4][jn];
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wb, acc[jn], 0, 0, 0);      // x3 w1, x2 w2, x1 w3, x2 w1, x1 w2, x1 w1
                }
        // ---- 32 pixels x 64 filters leave through the wave's LDS tile: lane = pixel row 4 q + (lane >> 4), filters c4 .. c4 + 3 ----
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * hi32) * 64 + jn * 32 + l31] = acc[jn][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int npx = min(RUN, a.W - x0);
        const size_t out_row = ((size_t)row * a.W + x0) * 64 + c4;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int px = 4 * q + (lane >> 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(Es + px * 64 + c4);
            v += bz;
            const bool in = px < npx;
            f32x4 yv = v;
            if constexpr (OBF) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (__bf16)v[e];
                    yv[e] = (float)h[e];
                }
                if (in) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.y) + out_row + (size_t)px * 64) = h;
            } else {
                if (in) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + out_row + (size_t)px * 64));
            }
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = in ? fmaxf(yv[e], lo) - pivot[e] : 0.f;
                    s0[e] += d;
                    s1[e] = fmaf(d, d, s1[e]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();       // the next run overwrites the tile
    }
    if constexpr (STATS) {
#pragma unroll
        for (int off = 16; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0[e] += __shfl_xor(s0[e], off, 64);
                s1[e] += __shfl_xor(s1[e], off, 64);
            }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][wave][c4 + e] = s0[e];
                red[1][wave][c4 + e] = s1[e];
            }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
            a.stat_part[((size_t)blockIdx.x * 2 + which) * 64 + ch] =
                (red[which][0][ch] + red[which][1][ch]) + (red[which][2][ch] + red[which][3][ch]);
        }
    }
}

#endif  // L3_EXPERIMENTS

int first_blocks(const ConvGeom& g) {
    const int segs = (g.W + RUN - 1) / RUN;
    const long units = (long)g.N * g.H * segs;
    long blocks = (units + 4 * 16 - 1) / (4 * 16);      // >= 16 runs per wave: the filter taps are loaded once per wave
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int CIN>
void launch_first(const FirstArgs& a, int blocks, hipStream_t s, bool out_bf16) {
#ifdef L3_EXPERIMENTS
    // the split-bf16 matrix-core form (measured, not the product path: see its header): L3_FIRST_FWD_X6=1 (read per call) selects it
    const char* x6 = l3_knob("L3_FIRST_FWD_X6");
    if (x6 != nullptr && atoi(x6) != 0) {
        if (a.stat_part != nullptr) {
            if (out_bf16) hipLaunchKernelGGL((conv_first_fwd_x6_kernel<CIN, true, true>), dim3(blocks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv_first_fwd_x6_kernel<CIN, true, false>), dim3(blocks), dim3(256), 0, s, a);
        } else {
            if (out_bf16) hipLaunchKernelGGL((conv_first_fwd_x6_kernel<CIN, false, true>), dim3(blocks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv_first_fwd_x6_kernel<CIN, false, false>), dim3(blocks), dim3(256), 0, s, a);
        }
        return;
    }
#endif
    if (a.stat_part != nullptr) {
        if (out_bf16) hipLaunchKernelGGL((conv_first_fwd_kernel<CIN, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_first_fwd_kernel<CIN, true, false>), dim3(blocks), dim3(256), 0, s, a);
    } else {
        if (out_bf16) hipLaunchKernelGGL((conv_first_fwd_kernel<CIN, false, true>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_first_fwd_kernel<CIN, false, false>), dim3(blocks), dim3(256), 0, s, a);
    }
}

}  // namespace

bool conv_first_ok(const ConvGeom& g) {
    const char* env = l3_knob("L3_CONV_FIRST");            // read per call: the tests switch it inside one process
    return (env ? atoi(env) : 1) && g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W &&
           (g.Cin == 1 || g.Cin == 3) && g.Cout == 64 && (size_t)g.N * g.H * ((g.W + RUN - 1) / RUN) < (1ull << 31);
}

int conv_first_stat_blocks(const ConvGeom& g) { return conv_first_ok(g) ? first_blocks(g) : 0; }

void conv_first_fwd(const float* x, const float* w, const float* bias, void* y, const ConvGeom& g, hipStream_t s,
                    float* stat_part, int stat_mode, bool out_bf16) {
    FirstArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.N = g.N; a.H = g.H; a.W = g.W;
    a.segs = (g.W + RUN - 1) / RUN;
    a.units = g.N * g.H * a.segs;
    const int blocks = first_blocks(g);
    a.per_wave = (a.units + blocks * 4 - 1) / (blocks * 4);
    a.stat_part = stat_part;
    a.stat_mode = stat_mode;
    if (g.Cin == 1)
        launch_first<1>(a, blocks, s, out_bf16);
    else
        launch_first<3>(a, blocks, s, out_bf16);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of the first convolution (input = the 1 or 3 normalised channels + the ones channel that carries
// the input BatchNorm's beta: Cin = 2 or 4, engine.hip `xaug`), 64 filters.
//
// dW[tap][ci][co] = sum_p x[p + tap][ci] dY[p][co]: 18 / 36 rows x 64 columns, K = every pixel of the batch -- the
// work is reading dY (the largest tensor of the network) once.  As a GEMM on v_mfma_f32_16x16x4_f32:
//   wave   = a contiguous range of 4-pixel units (4 consecutive pixels of an image row) = the k-steps;
//   B      = dY: lane (n, k) loads the 16 bytes dY[pixel k][4n .. 4n+3] -- one fully coalesced 1-KiB buffer load per
//            unit -- and feeds component j to the MFMA of column tile j (column n of tile j = filter 4n + j);
//   A      = lane (m, k) gathers x[pixel k + tap(m)][ci(m)] for row m = tap * Cin + ci (2 or 3 row tiles of 16;
//            out-of-image = out-of-range buffer offset = 0);
//   output = one partial [rows][64] per workgroup (its four waves summed in order through LDS), summed in order by conv.hip's
//            split-K reduce.
//   FUSE   = the BatchNorm that follows the convolution hands over its backward COEFFICIENTS instead of dY (kernels.h
//            FirstWgFuse): dY = cA * mask(dA) + (cB * y + cC) per element, what bn_fused.hip's bn_bwd_apply_fast_kernel would have
//            written -- the same fp32 expression, so the same values -- is formed from the convolution's stored output y and the
//            gradient dA behind the BatchNorm (FUSE 1: fp32 tensors, 2: bfloat16-stored) as the unit is consumed.  dY is the
//            largest tensor of the network and nothing else reads it: its pass (12 / 8 bytes per element) disappears, this
//            kernel reads 8 / 4 bytes per element instead of 4.
struct FirstWgArgs {
    const float* x;     // (N, H, W, CA)
    const float* dy;    // (N, H, W, 64); FUSE: unused
    float* part;        // [workgroups][9 * CA][64]
    int N, H, W;
    int segs;           // 4-pixel units per image row
    int units;          // N * H * segs
    int per_wave;
    FirstWgFuse f;      // FUSE != 0
};

// RELU (FUSE only): 1 = the BatchNorm is followed by a ReLU (the gradient passes where its output was positive), 0 = none.  A
// template parameter: as a run-time value the compiler evaluated every form and selected (59 VALU instructions per unit beside the
// 12 fp32 MFMAs, each worth ~4 cycles of matrix time on this chip: the fp32 kernel took 495 us for 245 + the pass it replaced).
template <int CA, int FUSE, int RELU>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(FirstWgArgs a) {
    constexpr int ROWS = 9 * CA, MT = (ROWS + 15) / 16;
    // (readfirstlane: the unit range of a wave, and everything counted from it below -- image row, segment, the loop -- is wave-uniform
    // and must be KNOWN to be: with a vector-valued range every `unit < end` test became an exec-masked branch around the loads, and
    // the compiler, unable to count loads across those branches, put a full s_waitcnt vmcnt(0) in front of every unit -- the four
    // units "in flight" were one; round 6)
    const int lane = threadIdx.x & 63, wave_g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int l15 = lane & 15, k = lane >> 4;
    // row m of tile mt: source displacement and validity
    int rel[MT], dyy[MT], dxx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l15;
        const int tap = m / CA, ci = m - tap * CA;
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        rel[mt] = ((dh * a.W + dw) * CA + ci) * 4;
        dyy[mt] = m < ROWS ? dh : -100000;          // rows beyond 9 * CA never load
        dxx[mt] = dw;
    }
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * CA * 4), 0x00020000);
    constexpr int EB = FUSE == 2 ? 2 : 4;                 // bytes per element of the streamed tensor(s)
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(FUSE ? a.f.dA : (const void*)a.dy), 0, (int)((size_t)a.N * a.H * a.W * 64 * EB), 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(FUSE ? a.f.bnx : (const void*)a.dy), 0, (int)((size_t)a.N * a.H * a.W * 64 * EB), 0x00020000);
    // this lane's four filters 4 l15 .. 4 l15 + 3 for the whole kernel: the BatchNorm's forward scale / shift (ReLU mask) and
    // its backward coefficients
    f32x4 fsc = {0.f, 0.f, 0.f, 0.f}, fsh = fsc, fA = fsc, fB = fsc, fC = fsc;
    if constexpr (FUSE != 0) {
        fsc = reinterpret_cast<const f32x4*>(a.f.scale)[l15];
        fsh = reinterpret_cast<const f32x4*>(a.f.shift)[l15];
        fA = reinterpret_cast<const f32x4*>(a.f.cA)[l15];
        fB = reinterpret_cast<const f32x4*>(a.f.cB)[l15];
        fC = reinterpret_cast<const f32x4*>(a.f.cC)[l15];
    }

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int u0 = wave_g * a.per_wave, u1 = min(a.units, u0 + a.per_wave);
    int row = u0 / a.segs, seg = u0 - row * a.segs;          // row = n * H + y
    int yrow = row % a.H;                                    // y, kept beside row (a modulo per unit was 20 instructions)
    // FUSE: bv / xv carry the raw loads (dA, y) of the unit; dy_of() turns them into dY when the unit is consumed
    // `live` (wave-uniform): the unit belongs to this wave's range; a dead unit loads nothing (out-of-range offsets = zeros) and adds
    // exact zeros (its A operand is 0) -- every unit of a DEPTH group runs the same straight-line code
    auto load = [&](float (&av)[MT], f32x4& bv, f32x4& xv, bool live) {
        const int y = yrow, px = seg * 4 + k;
        const bool pok = live & (px < a.W);
        const unsigned pix = (unsigned)(row * a.W + px);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // (& not &&: a short-circuit here compiles to an exec-masked branch around the load)
            const bool ok = pok & ((unsigned)(y + dyy[mt]) < (unsigned)a.H) & ((unsigned)(px + dxx[mt]) < (unsigned)a.W);
            av[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   xsrd, ok ? (int)(pix * (CA * 4) + rel[mt]) : (int)0x80000000, 0, 0));
        }
        if constexpr (FUSE == 2) {
            const u32x2 d2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(ysrd, pok ? (int)(pix * 128 + l15 * 8) : (int)0x80000000, 0, 2));
            const u32x2 x2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bsrd, pok ? (int)(pix * 128 + l15 * 8) : (int)0x80000000, 0, 2));
            bv = f32x4{__uint_as_float(d2.x), __uint_as_float(d2.y), 0.f, 0.f};         // (widened in dy_of)
            xv = f32x4{__uint_as_float(x2.x), __uint_as_float(x2.y), 0.f, 0.f};
        } else {
            bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               ysrd, pok ? (int)(pix * 256 + l15 * 16) : (int)0x80000000, 0, 2));      // aux 2 = nt: streamed once
            if constexpr (FUSE == 1)
                xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   bsrd, pok ? (int)(pix * 256 + l15 * 16) : (int)0x80000000, 0, 2));
        }
        if (++seg == a.segs) {
            seg = 0;
            ++row;
            if (++yrow == a.H) yrow = 0;
        }
    };
    // dY is streamed once, 1 KiB per unit and wave: DEPTH units in flight per wave (with one, sixteen waves per CU kept 16 KiB on
    // the wire -- 2.9 TB/s; profiles/r05m_bf16_b128_kernels_in_order.txt)
    // bn_bwd_apply_fast_kernel's expression (bn_fused.hip), non-pooled form, BatchNorm [then ReLU].  A lane outside the row (zero
    // loads) yields cC -- times its zero A operand.
    auto dy_of = [&](f32x4 dr, f32x4 xr) {
        if constexpr (FUSE == 0) return dr;
        if constexpr (FUSE == 2) {
            const unsigned d0 = __float_as_uint(dr.x), d1 = __float_as_uint(dr.y), x0 = __float_as_uint(xr.x), x1 = __float_as_uint(xr.y);
            dr = f32x4{__uint_as_float(d0 << 16), __uint_as_float(d0 & 0xffff0000u), __uint_as_float(d1 << 16), __uint_as_float(d1 & 0xffff0000u)};
            xr = f32x4{__uint_as_float(x0 << 16), __uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 << 16), __uint_as_float(x1 & 0xffff0000u)};
        }
        f32x4 d = dr, o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (RELU == 1) d[e] = fmaf(xr[e], fsc[e], fsh[e]) > 0.f ? d[e] : 0.f;
            o[e] = fA[e] * d[e] + (fB[e] * xr[e] + fC[e]);
        }
        return o;
    };
    // (the fp32 vision form would take 134 registers with four units: three waves per SIMD, and the launch's 1024 workgroups -- four per
    // CU -- would run as a full round and a quarter-full one; with three units it fits 128 and the same 96 KiB per CU stay in flight)
    constexpr int DEPTH = (CA == 4 && FUSE == 1) ? 3 : 4;
    float av[DEPTH][MT];
    f32x4 bv[DEPTH], xq[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        load(av[d], bv[d], xq[d], u0 + d < u1);
        __builtin_amdgcn_sched_barrier(0);      // in slot order, as the loop issues them: its first wait is then for slot 0 alone
    }
    for (int u = u0; u < u1; u += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // a REAL copy of the unit's A operands: the loads below then land in av[d] itself.  Left to the register allocator the new
            // values arrived in fresh registers and were copied into place at the loop's end -- behind a wait for every load in flight
            float ac[MT];
#pragma unroll
            // (s_nop 1: the two wait states a VALU result needs before an MFMA reads it -- nothing pads the inside of an asm statement;
            // as scheduled today some twenty instructions lie between, but that is the scheduler's choice)
            for (int mt = 0; mt < MT; ++mt) asm volatile("v_mov_b32 %0, %1\n\ts_nop 1" : "=v"(ac[mt]) : "v"(av[d][mt]));
            if constexpr (FUSE != 2) {
                asm volatile("" : "+v"(bv[d]));
                if constexpr (FUSE == 1) asm volatile("" : "+v"(xq[d]));
            }
            const f32x4 bc = dy_of(bv[d], xq[d]);
            load(av[d], bv[d], xq[d], u + d + DEPTH < u1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mt], bc[j], acc[mt][j], 0, 0, 0);
            // units stay in program order: scheduled as one block of four the loads of slot 0 ended up LAST and the loop waited for all
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D tile (mt, j): lane (n = l15, k) holds rows 4k .. 4k+3 of column n  ->  filter 4n + j.  The four waves of the workgroup meet
    // in LDS and leave ONE partial (summed in wave order): a quarter of the split-K partials for conv.hip's reduce.
    __shared__ f32x4 slab[4][ROWS * 16];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + 4 * k + r;
            if (m < ROWS) slab[wave][m * 16 + l15] = f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
        }
    __syncthreads();
    f32x4* out = reinterpret_cast<f32x4*>(a.part + (size_t)blockIdx.x * ROWS * 64);
    for (int i = threadIdx.x; i < ROWS * 16; i += 256) out[i] = ((slab[0][i] + slab[1][i]) + slab[2][i]) + slab[3][i];
}

constexpr int FIRST_WG_WAVES = 4096;

bool conv_first_wgrad_ok(const ConvGeom& g) {
    const char* env = l3_knob("L3_FIRST_WGRAD");          // read per call: the tests switch it inside one process
    return (env ? atoi(env) : 1) && g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W &&
           (g.Cin == 2 || g.Cin == 4) && g.Cout == 64 && (size_t)g.N * g.H * g.W * 64 * 4 < (1ull << 31);
}

size_t conv_first_wgrad_scratch_floats(const ConvGeom& g) {
    return conv_first_wgrad_ok(g) ? (size_t)FIRST_WG_WAVES * 9 * g.Cin * 64 : 0;
}

// partials: returns the number of [9 * Cin][64] slices written to `part`
int conv_first_wgrad(const float* x, const float* dy, float* part, const ConvGeom& g, hipStream_t s, const FirstWgFuse* fuse) {
    FirstWgArgs a;
    a.x = x; a.dy = dy; a.part = part;
    a.f = fuse != nullptr ? *fuse : FirstWgFuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    a.N = g.N; a.H = g.H; a.W = g.W;
    a.segs = (g.W + 3) / 4;
    a.units = g.N * g.H * a.segs;
    int waves = FIRST_WG_WAVES;
    if (waves > (a.units + 15) / 16) waves = ((a.units + 15) / 16 + 3) / 4 * 4;      // >= 16 units per wave
    a.per_wave = (a.units + waves - 1) / waves;
    // (fuse->relu is 0 or 1: the engine does not defer the ReLU -> BatchNorm order; anything else has no kernel -- refuse loudly)
    if (fuse != nullptr && (fuse->relu < 0 || fuse->relu > 1)) {
        fprintf(stderr, "libl3hip: conv_first_wgrad: FirstWgFuse::relu = %d has no kernel\n", fuse->relu);
        abort();
    }
    const int fm = fuse == nullptr ? 0 : (fuse->bf16 ? 3 : 1) + (fuse->relu == 1 ? 1 : 0);
    using Fn = void (*)(FirstWgArgs);
    static const Fn fns[2][5] = {{conv_first_wgrad_kernel<2, 0, 0>, conv_first_wgrad_kernel<2, 1, 0>, conv_first_wgrad_kernel<2, 1, 1>,
                                  conv_first_wgrad_kernel<2, 2, 0>, conv_first_wgrad_kernel<2, 2, 1>},
                                 {conv_first_wgrad_kernel<4, 0, 0>, conv_first_wgrad_kernel<4, 1, 0>, conv_first_wgrad_kernel<4, 1, 1>,
                                  conv_first_wgrad_kernel<4, 2, 0>, conv_first_wgrad_kernel<4, 2, 1>}};
    hipLaunchKernelGGL(fns[g.Cin == 2 ? 0 : 1][fm], dim3(waves / 4), dim3(256), 0, s, a);
    return waves / 4;
}

}  // namespace l3
