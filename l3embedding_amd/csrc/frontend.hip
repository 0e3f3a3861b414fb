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
