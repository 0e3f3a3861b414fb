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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
                    reinterpret_cast<float*>(a.y)[out_row + (size_t)px * 64] = acc;
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
    const char* env = getenv("L3_CONV_FIRST");            // read per call: the tests switch it inside one process
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

}  // namespace l3
