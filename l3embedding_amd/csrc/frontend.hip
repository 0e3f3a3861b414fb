// frontend.hip -- kapre-equivalent audio front-end on the GPU.
//
// Replaces (reference call sites, relative to the reference tree):
//   kapre Spectrogram / Melspectrogram layers   l3embedding/audio_model.py:39-40,149-151,257-260,367-369
//   the L3-paper log normalisation Lambda       l3embedding/audio_model.py:43
// [3P] semantics restated from kapre 0.1.3.1/0.1.4 (time_frequency.py, backend.py,
// backend_keras.py): STFT as a strided conv with Hann-windowed cos / -sin kernels,
// power = re^2 + im^2, optional mel projection (librosa filters.mel, htk), optional
// sqrt (power != 2.0), amplitude_to_decibel = 10*log10(max(x,1e-10)) - max, clamp -80.
//
// The DFT itself runs on the matrix cores: frame_audio() materialises the strided
// frames (B*n_frames, n_dft) and conv_fwd() multiplies them with the (n_dft, re|im)
// basis as a 1x1 convolution.  The mel projection uses the band structure of the
// filterbank (<= 28 taps per filter for 256 mels) instead of a dense 1025x256 GEMM.
#include "kernels.h"
#include "device_common.h"

namespace l3 {

__global__ __launch_bounds__(256) void frame_audio_kernel(const float* audio, float* frames, int B, int T,
                                                          int n_dft, int n_hop, int pad_left, int n_frames) {
    const int64_t total = (int64_t)B * n_frames * n_dft;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % n_dft);
        const int64_t r = i / n_dft;
        const int f = (int)(r % n_frames);
        const int b = (int)(r / n_frames);
        const int src = f * n_hop - pad_left + k;
        frames[i] = (src >= 0 && src < T) ? audio[(size_t)b * T + src] : 0.f;
    }
}
void frame_audio(const float* audio, float* frames, int B, int T, const FrontendCfg& c, hipStream_t s) {
    const int64_t total = (int64_t)B * c.n_frames * c.n_dft;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(frame_audio_kernel, dim3((int)blocks), dim3(256), 0, s, audio, frames, B, T, c.n_dft,
                       c.n_hop, c.pad_left, c.n_frames);
}

// folded frames: fe[m][j] = x[j] + x[N-j] (j = 1..N/2-1), x[0], x[N/2];  fo[m][j] = x[j+1] - x[N-j-1]
__global__ __launch_bounds__(256) void frame_audio_folded_kernel(const float* __restrict__ audio, float* __restrict__ fe,
                                                                 float* __restrict__ fo, int B, int T, FrontendCfg c) {
    const int N = c.n_dft, H = N / 2, W2 = c.ke + c.ko;
    const int64_t total = (int64_t)B * c.n_frames * W2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j2 = (int)(i % W2);
        const int64_t r = i / W2;
        const int f = (int)(r % c.n_frames);
        const int b = (int)(r / c.n_frames);
        const int base = f * c.n_hop - c.pad_left;
        const float* a = audio + (size_t)b * T;
        auto at = [&](int n) {
            const int src = base + n;
            return (src >= 0 && src < T) ? a[src] : 0.f;
        };
        if (j2 < c.ke) {
            const int n = j2;
            float v = 0.f;
            if (n == 0 || n == H) v = at(n);
            else if (n < H) v = at(n) + at(N - n);
            fe[r * c.ke + j2] = v;
        } else {
            const int j = j2 - c.ke, n = j + 1;
            fo[r * c.ko + j] = n < H ? at(n) - at(N - n) : 0.f;
        }
    }
}
void frame_audio_folded(const float* audio, float* fe, float* fo, int B, int T, const FrontendCfg& c, hipStream_t s) {
    const int64_t total = (int64_t)B * c.n_frames * (c.ke + c.ko);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(frame_audio_folded_kernel, dim3((int)blocks), dim3(256), 0, s, audio, fe, fo, B, T, c);
}

// ---- factored DFT (stock kapre kernels): n_dft = N1 * N2 ------------------------------------------------------------------------
// X[k1 + N1 k2] = sum_n2 W_N^(n2 k1) W_N2^(n2 k2) sum_n1 xw[N2 n1 + n2] W_N1^(n1 k1)      (W_M = exp(-2 pi i / M))
// One workgroup per frame in both elementwise passes: the frame goes through LDS once so that global reads AND writes are coalesced
// (the GEMM rows are the frame read with stride N2).  LDS rows are padded by one float: the transposed accesses are conflict free.
__global__ __launch_bounds__(256) void dft_pack_frames_kernel(const float* __restrict__ audio, const float* __restrict__ win,
                                                              float* __restrict__ a1, int T, FrontendCfg c) {
    extern __shared__ float sm[];                   // [N1][N2 + 1]
    const int row = blockIdx.x, b = row / c.n_frames, f = row - b * c.n_frames;
    const int base = f * c.n_hop - c.pad_left;
    const float* a = audio + (size_t)b * T;
    for (int n = threadIdx.x; n < c.n_dft; n += 256) {
        const int src = base + n;
        const float v = (src >= 0 && src < T) ? a[src] : 0.f;
        sm[(n / c.N2) * (c.N2 + 1) + n % c.N2] = v * win[n];
    }
    __syncthreads();
    float* out = a1 + (size_t)row * c.n_dft;
    for (int i = threadIdx.x; i < c.n_dft; i += 256) {          // i = n2 * N1 + n1
        const int n2 = i / c.N1, n1 = i - n2 * c.N1;
        out[i] = sm[n1 * (c.N2 + 1) + n2];
    }
}
void dft_pack_frames(const float* audio, const float* win, float* a1, int B, int T, const FrontendCfg& c, hipStream_t s) {
    hipLaunchKernelGGL(dft_pack_frames_kernel, dim3(B * c.n_frames), dim3(256), (size_t)c.N1 * (c.N2 + 1) * sizeof(float), s, audio, win,
                       a1, T, c);
}

__global__ __launch_bounds__(256) void dft_twiddle_kernel(const float* __restrict__ y, const float* __restrict__ tw,
                                                          float* __restrict__ a2, float* __restrict__ nyq, FrontendCfg c) {
    extern __shared__ float sm[];                   // [N2][2 N1 + 1]: row n2 = re k1 (N1) | im k1 (N1)
    const int row = blockIdx.x, W1 = 2 * c.N1, tot = c.N2 * W1;
    const float* yin = y + (size_t)row * tot;
    for (int i = threadIdx.x; i < tot; i += 256) sm[(i / W1) * (W1 + 1) + i % W1] = yin[i];
    __syncthreads();
    float* out = a2 + (size_t)row * tot;            // [N1][re n2 (N2) | im n2 (N2)]
    const int W2 = 2 * c.N2;
    for (int i = threadIdx.x; i < tot; i += 256) {
        const int k1 = i / W2, cc = i - k1 * W2, n2 = cc % c.N2;
        const float yr = sm[n2 * (W1 + 1) + k1], yi = sm[n2 * (W1 + 1) + c.N1 + k1];
        const float co = tw[(n2 * c.N1 + k1) * 2], si = tw[(n2 * c.N1 + k1) * 2 + 1];     // exp(-i t) = (co, -si)
        out[i] = cc < c.N2 ? fmaf(yr, co, yi * si) : fmaf(yi, co, -(yr * si));
    }
    if (threadIdx.x < 64) {                         // Nyquist bin (k1 = 0, k2 = N2 / 2): sum_n2 (-1)^n2 Y[n2][0]   (Y[.][0] is real)
        float v = 0.f;
        for (int n2 = threadIdx.x; n2 < c.N2; n2 += 64) v += (n2 & 1) ? -sm[n2 * (W1 + 1)] : sm[n2 * (W1 + 1)];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (threadIdx.x == 0) nyq[row] = v;
    }
}
void dft_twiddle(const float* y, const float* tw, float* a2, float* nyq, int frames, const FrontendCfg& c, hipStream_t s) {
    hipLaunchKernelGGL(dft_twiddle_kernel, dim3(frames), dim3(256), (size_t)c.N2 * (2 * c.N1 + 1) * sizeof(float), s, y, tw, a2, nyq, c);
}

// ---- the whole factored front-end of one frame in ONE wave (round 6) ---------------------------------------------------------------
// window -> 64 length-32 DFTs (MFMA) -> twiddles -> 32 length-64 complex DFTs (MFMA) -> power -> mel -> sqrt / dB, nothing but the
// audio read and the (B, mels, frames) features written: the two-GEMM form above moves 2 x n_dft floats per frame through HBM four
// times.  Persistent workgroups of eight waves, one frame per wave at a time; LDS (145 KiB) = the twiddle table and one 64-entry cosine
// table once per workgroup + per wave ONE region that holds the frame as [n1 32][65], then the twiddled first stage as [k1 32][129], then
// the power spectrum (constants DF_* below).
//   v_mfma_f32_32x32x2_f32: lane l supplies A[row l % 32][k l / 32] and B[k l / 32][col l % 32]; it holds column l % 32 of the result,
//   register r = row 8 (r / 4) + 4 (l / 32) + r % 4.
// Stage 1: C1[n2][j] = sum_n1 xw[64 n1 + n2] B1[n1][j]   (j < 32: re k1 = j; j >= 32: im k1 = j - 32)        2 x 2 tiles, 16 k-steps
// Stage 2: C2[k1][j] = sum_c  Z[k1][c] B2[c][j]          (c < 64: re n2 = c, else im n2 = c - 64;  j: re / im k2)  1 x 2 tiles, 64 k-steps
// A lane ends up with re and im of the SAME bin (k1 = its rows, k2 = its column) in its two column tiles: the power spectrum needs
// no exchange.  Only for n_dft = 2048 = 32 x 64 and a mel front-end (what FrontendCfg::factored is set for).
constexpr int DF_N1 = 32, DF_N2 = 64, DF_XW = DF_N1 * (DF_N2 + 1), DF_Z = DF_N1 * (2 * DF_N2 + 1);
constexpr int DF_B1 = DF_N1 * 2 * DF_N1, DF_B2 = 2 * DF_N2 * DF_N2, DF_TW = DF_N2 * DF_N1 * 2;
// LDS: the twiddle table [n2][k1](cos, sin) and cos(2 pi m / 64), m < 64 -- every entry of the two small DFT matrices is one of
// those 64 values (a 64-entry table is one entry per bank: any index pattern is conflict free) -- once per workgroup; per wave ONE
// region of 32 x 129 floats that holds, in turn, the windowed frame [n1][65], the twiddled first stage [k1][129] and the power
// spectrum: 16.1 KiB per wave, EIGHT waves per workgroup = two per SIMD (with the matrices in LDS there was room for four: one wave per
// SIMD hides nothing, 244 us per launch at 64 pairs for 73 us of matrix work).
constexpr int DF_WAVES = 8;
constexpr int DF_WAVE = DF_Z;
constexpr size_t DF_LDS_BYTES = (size_t)(DF_TW + 64 + DF_WAVES * DF_WAVE) * sizeof(float);

__global__ __launch_bounds__(64 * DF_WAVES) void dft_fused_kernel(const float* __restrict__ audio, const float* __restrict__ win,
                                                        const float* __restrict__ b1g, const float* __restrict__ b2g,
                                                        const float* __restrict__ twg, const float* __restrict__ melw,
                                                        const int* __restrict__ mel_start, const int* __restrict__ mel_len,
                                                        const int* __restrict__ mel_off, float* __restrict__ out, int B, int T,
                                                        FrontendCfg c) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const TW = sm;
    float* const C64 = TW + DF_TW;                      // cos(2 pi m / 64) = the first column pair of the length-64 DFT matrix: b2g[m * 64 + 1]
    const int t = threadIdx.x, wave = t >> 6, l = t & 63, col = l & 31, hf = l >> 5;
    float* const Z = C64 + 64 + wave * DF_WAVE;         // [k1][129]
    float* const XW = Z;                                // [n1][65] before stage 1 is done; after stage 2: the power spectrum
    (void)b1g;
    for (int i = t; i < DF_TW; i += 64 * DF_WAVES) TW[i] = twg[i];
    if (t < 64) C64[t] = b2g[t * DF_N2 + 1];            // B2[n2 = t][re k2 = 1] = cos(2 pi t / 64)
    __syncthreads();
    const int M = B * c.n_frames;
    for (int row = ((int)blockIdx.x * DF_WAVES + wave); row < M; row += (int)gridDim.x * DF_WAVES) {
        const int b = row / c.n_frames, f = row - b * c.n_frames;
        // ---- frame x window -> XW[n1][n2]   (n = 64 n1 + n2: iteration = n1, lane = n2)
        {
            const int base = f * c.n_hop - c.pad_left;
            const float* a = audio + (size_t)b * T;
            // unconditional loads from a clamped index, selected afterwards: a predicated load compiles to one exec-masked branch and
            // one full wait PER sample -- 32 serial round trips per frame were three quarters of this kernel's time
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float xv[DF_N1 / 2], wv[DF_N1 / 2];
#pragma unroll
                for (int i = 0; i < DF_N1 / 2; ++i) {
                    const int n = DF_N2 * (h * (DF_N1 / 2) + i) + l, src = base + n;
                    xv[i] = a[src < 0 ? 0 : src >= T ? T - 1 : src];
                    wv[i] = win[n];
                }
#pragma unroll
                for (int i = 0; i < DF_N1 / 2; ++i) {
                    const int n1 = h * (DF_N1 / 2) + i, src = base + DF_N2 * n1 + l;
                    XW[n1 * (DF_N2 + 1) + l] = (src >= 0 && src < T) ? xv[i] * wv[i] : 0.f;
                }
            }
        }
        // ---- stage 1
        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        {   // operands of k-step ks + 1 are requested before the four MFMAs of k-step ks are issued (one wave per SIMD: nobody else
            // hides the LDS latency)
            // B1[n1][re k1] = cos(2 pi n1 k1 / 32) = C64[(2 n1 k1) % 64], B1[n1][im k1] = -sin = C64[(2 n1 k1 + 16) % 64];  n1 = 2 ks + hf
            const float* xa = XW + hf * (DF_N2 + 1) + col;
            int qi = 2 * col * hf;
            float a0 = xa[0], a1 = xa[32], q0 = C64[qi & 63], q1 = C64[(qi + 16) & 63];
#pragma unroll
            for (int ks = 0; ks < DF_N1 / 2; ++ks) {
                float na0 = 0.f, na1 = 0.f, nq0 = 0.f, nq1 = 0.f;
                if (ks + 1 < DF_N1 / 2) {
                    na0 = xa[2 * (ks + 1) * (DF_N2 + 1)];
                    na1 = xa[2 * (ks + 1) * (DF_N2 + 1) + 32];
                    qi += 4 * col;
                    nq0 = C64[qi & 63];
                    nq1 = C64[(qi + 16) & 63];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q1, acc[1][1], 0, 0, 0);
                a0 = na0; a1 = na1; q0 = nq0; q1 = nq1;
            }
        }
        // ---- twiddles: lane = k1 (its column), rows = n2; Z[k1][n2] / Z[k1][64 + n2]; the Nyquist bin from the k1 = 0 column
        float nyq = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n2 = 32 * mt + 8 * (r >> 2) + 4 * hf + (r & 3);
                const float yr = acc[mt][0][r], yi = acc[mt][1][r];
                const f32x2 w = *reinterpret_cast<const f32x2*>(TW + (n2 * DF_N1 + col) * 2);        // exp(-i t) = (w.x, -w.y)
                Z[col * (2 * DF_N2 + 1) + n2] = fmaf(yr, w.x, yi * w.y);
                Z[col * (2 * DF_N2 + 1) + DF_N2 + n2] = fmaf(yi, w.x, -(yr * w.y));
                nyq += (n2 & 1) ? -yr : yr;
            }
        nyq += __shfl_xor(nyq, 32, 64);              // lanes 0 and 32 hold the two row halves of column k1 = 0
        nyq = __shfl(nyq, 0, 64);
        // ---- stage 2 (two k halves per column tile: four independent accumulator chains)
        f32x16 ac2[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ac2[kh][nt][r] = 0.f;
        {
            // rows c < 64 (re n2 = c):  B2[c][re k2] = cos p = C64[n2 k2 % 64],  B2[c][im k2] = -sin p = C64[(n2 k2 + 16) % 64]
            // rows c >= 64 (im n2):     B2[c][re k2] = sin p = C64[(n2 k2 - 16) % 64],  B2[c][im k2] = cos p;   n2 = 2 ks + hf, k2 = col
            const float* za = Z + col * (2 * DF_N2 + 1) + hf;
            int qi = col * hf;
            float av[2], q0[2], q1[2];
            av[0] = za[0];
            av[1] = za[DF_N2];
            q0[0] = C64[qi & 63];
            q1[0] = C64[(qi + 16) & 63];
            q0[1] = C64[(qi + 48) & 63];
            q1[1] = q0[0];
#pragma unroll 8
            for (int ks = 0; ks < DF_N2 / 2; ++ks) {
                float nav[2] = {0.f, 0.f}, nq0[2] = {0.f, 0.f}, nq1[2] = {0.f, 0.f};
                if (ks + 1 < DF_N2 / 2) {
                    nav[0] = za[2 * (ks + 1)];
                    nav[1] = za[2 * (ks + 1) + DF_N2];
                    qi += 2 * col;
                    nq0[0] = C64[qi & 63];
                    nq1[0] = C64[(qi + 16) & 63];
                    nq0[1] = C64[(qi + 48) & 63];
                    nq1[1] = nq0[0];
                }
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    ac2[kh][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kh], q0[kh], ac2[kh][0], 0, 0, 0);
                    ac2[kh][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kh], q1[kh], ac2[kh][1], 0, 0, 0);
                    av[kh] = nav[kh]; q0[kh] = nq0[kh]; q1[kh] = nq1[kh];
                }
            }
        }
        // ---- power spectrum: this lane's column = k2, rows = k1; bin k = k1 + 32 k2 at P[k1 * 33 + k2]   (the region is free again)
        float* const P = Z;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k1 = 8 * (r >> 2) + 4 * hf + (r & 3);
            const float re = ac2[0][0][r] + ac2[1][0][r], im = ac2[0][1][r] + ac2[1][1][r];
            P[k1 * 33 + col] = re * re + im * im;
        }
        if (l == 0) P[DF_N1 * 33] = nyq * nyq;
        // ---- mel projection + sqrt / dB (spec_to_features_kernel's arithmetic on the same P)
        const int F = c.n_mels;
        for (int q = l; q < F; q += 64) {
            const int st = mel_start[q], ln = mel_len[q], of = mel_off[q];
            float v = 0.f;
            for (int i = 0; i < ln; ++i) {
                const int k = st + i;
                v = fmaf(k == c.n_dft / 2 ? P[DF_N1 * 33] : P[(k & 31) * 33 + (k >> 5)], melw[of + i], v);
            }
            if (c.sqrt_out) v = sqrtf(v);
            if (c.db) v = 10.f * logf(fmaxf(v, 1e-10f)) / 2.302585092994046f;
            if (c.loglambda) v = logf(fmaxf(v, 1e-12f)) / 5.0f;
            out[((size_t)b * F + q) * c.n_frames + f] = v;
        }
    }
}
void dft_fused(const float* audio, const float* win, const float* b1, const float* b2, const float* tw, const float* melw,
               const int* mel_start, const int* mel_len, const int* mel_off, float* out, int B, int T, const FrontendCfg& c,
               hipStream_t s) {
    // per device, once: the LDS attribute and the CU count (hipGetDeviceProperties is not a call to make every step)
    static int cus[L3_MAX_DEVICES] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& ncu = cus[dev & (L3_MAX_DEVICES - 1)];
    if (ncu == 0) {
        (void)hipFuncSetAttribute((const void*)dft_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DF_LDS_BYTES);
        hipDeviceProp_t prop;
        ncu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int M = B * c.n_frames, groups = (M + DF_WAVES - 1) / DF_WAVES;
    hipLaunchKernelGGL(dft_fused_kernel, dim3(groups < ncu ? groups : ncu), dim3(64 * DF_WAVES), DF_LDS_BYTES, s, audio, win, b1, b2, tw, melw,
                       mel_start, mel_len, mel_off, out, B, T, c);
}

__global__ __launch_bounds__(256) void spec_to_features_kernel(const float* spec, const float* melw,
                                                               const int* mel_start, const int* mel_len,
                                                               const int* mel_off, float* out, FrontendCfg c, const float* nyq) {
    extern __shared__ float pw[];
    const int row = blockIdx.x;               // b * n_frames + f
    const int b = row / c.n_frames, f = row - b * c.n_frames;
    if (c.factored) {
        // x2[(row * N1 + k1)][re k2 (N2 / 2) | im k2 (N2 / 2)]: bin k = k1 + N1 k2; the frame's N1 * N2 floats are contiguous
        const int h2 = c.N2 / 2;
        const float* xr = spec + (size_t)row * c.N1 * c.N2;
        for (int i = threadIdx.x; i < c.N1 * h2; i += 256) {
            const int k1 = i / h2, k2 = i - k1 * h2;
            const float re = xr[k1 * c.N2 + k2], im = xr[k1 * c.N2 + h2 + k2];
            pw[k1 + c.N1 * k2] = re * re + im * im;
        }
        if (threadIdx.x == 0) pw[c.n_dft / 2] = nyq[row] * nyq[row];
    } else {
    // unfolded: one row = [re | im | pad];  folded: all re rows (nc wide), then all im rows
    const float* spr = c.folded ? spec + (size_t)row * c.nc : spec + (size_t)row * c.ncols_pad;
    const float* spi = c.folded ? spec + ((size_t)gridDim.x + row) * c.nc : spr + c.n_freq;
    for (int j = threadIdx.x; j < c.n_freq; j += 256) {
        const float re = spr[j], im = spi[j];
        pw[j] = re * re + im * im;
    }
    }
    __syncthreads();
    const int F = c.n_mels ? c.n_mels : c.n_freq;
    for (int q = threadIdx.x; q < F; q += 256) {
        float v;
        if (c.n_mels) {
            const int st = mel_start[q], ln = mel_len[q], of = mel_off[q];
            v = 0.f;
            for (int i = 0; i < ln; ++i) v = fmaf(pw[st + i], melw[of + i], v);
        } else {
            v = pw[q];
        }
        if (c.sqrt_out) v = sqrtf(v);
        if (c.db) v = 10.f * logf(fmaxf(v, 1e-10f)) / 2.302585092994046f;
        if (c.loglambda) v = logf(fmaxf(v, 1e-12f)) / 5.0f;
        out[((size_t)b * F + q) * c.n_frames + f] = v;
    }
}
void spec_to_features(const float* spec, const float* melw, const int* mel_start, const int* mel_len,
                      const int* mel_off, float* out, int B, const FrontendCfg& c, hipStream_t s, const float* nyq) {
    hipLaunchKernelGGL(spec_to_features_kernel, dim3(B * c.n_frames), dim3(256), c.n_freq * sizeof(float), s,
                       spec, melw, mel_start, mel_len, mel_off, out, c, nyq);
}

__global__ __launch_bounds__(256) void sample_max_kernel(const float* x, float* smax, int64_t per_sample) {
    __shared__ float sm[256];
    const float* p = x + (size_t)blockIdx.x * per_sample;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < per_sample; i += 256) m = fmaxf(m, p[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) smax[blockIdx.x] = sm[0];
}
__global__ void batch_max_kernel(float* smax, int B) {
    __shared__ float sm[256];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < B; i += 256) m = fmaxf(m, smax[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    m = sm[0];
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) smax[i] = m;
}
__global__ __launch_bounds__(256) void db_sub_clamp_kernel(float* x, const float* smax, int64_t per_sample,
                                                           int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / per_sample);
        x[i] = fmaxf(x[i] - smax[b], -80.f);
    }
}
void db_normalize(float* x, float* smax, int B, int64_t per_sample, int batch_scope, hipStream_t s) {
    hipLaunchKernelGGL(sample_max_kernel, dim3(B), dim3(256), 0, s, x, smax, per_sample);
    if (batch_scope) hipLaunchKernelGGL(batch_max_kernel, dim3(1), dim3(256), 0, s, smax, B);
    const int64_t total = (int64_t)B * per_sample;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(db_sub_clamp_kernel, dim3((int)blocks), dim3(256), 0, s, x, smax, per_sample, total);
}

}  // namespace l3
