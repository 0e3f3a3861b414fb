// elementwise.hip -- HBM-bound kernels of the L3 training step: BatchNorm
// (statistics, apply+ReLU, backward), ReLU, max-pool, input preprocessing, the
// dense/softmax/cross-entropy head and Adam.
//
// Reference semantics restated (paths relative to the reference tree):
//   BatchNormalization  l3embedding/audio_model.py:370,379,...; vision_model.py:124,133,...
//                       keras defaults axis=-1, eps=1e-3, momentum=0.99, biased batch variance
//   Activation('relu')  ibid.   MaxPooling2D  audio_model.py:386,402,418,435; vision_model.py:140,...
//   Dense/softmax       l3embedding/model.py:25-31
//   loss / optimizer    l3embedding/train.py:270-284 (categorical_crossentropy, Adam)
//   preprocessing       l3embedding/train.py:186,189; l3embedding/audio.py:4-31
//
// Channel-wise reductions are two-stage and deterministic: stage 1 writes fp32
// per-block partials, stage 2 combines them in fp64.
#include "kernels.h"

namespace l3 {

static constexpr int RED_BLOCK = 256;
static constexpr int RED_MAX_BLOCKS = 1024;
static constexpr int SMALLC_MAX = 16;

static inline int red_blocks(int64_t rows, int C) {
    // each block should see at least ~64 rows-passes of work
    int rows_per_pass = (C % 4 == 0 && C >= 4 && C <= 1024) ? (RED_BLOCK / (C / 4) > 0 ? RED_BLOCK / (C / 4) : 1) : RED_BLOCK;
    int64_t b = (rows + (int64_t)rows_per_pass * 16 - 1) / ((int64_t)rows_per_pass * 16);
    if (b < 1) b = 1;
    if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
    return (int)b;
}

size_t colreduce_scratch_floats(int64_t rows, int C) {
    const size_t a = (size_t)red_blocks(rows, C) * 2 * (size_t)(C < 4 ? 4 : C) + 64;
    const size_t b = bn_fast_ok(C) ? bn_fast_scratch_floats(C) : 0;
    return a > b ? a : b;
}

// ---- stage 1: generic per-channel accumulation -------------------------------
// Functor F provides: static constexpr int Q (1 or 2 quantities);
//   __device__ void operator()(int64_t idx /*element index*/, int c, float& q0, float& q1)
template <class F>
__global__ __launch_bounds__(RED_BLOCK) void colreduce_vec_kernel(F f, float* part, int64_t rows, int C) {
    // C % 4 == 0, C/4 divides 256
    __shared__ float sm[2][RED_BLOCK * 4];
    const int tpr = C >> 2;
    const int rpp = RED_BLOCK / tpr;
    const int t = threadIdx.x;
    const int c4 = t % tpr, rl = t / tpr;
    const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0 + rl; r < r1; r += rpp) f.vec(r * C + c4 * 4, c4 * 4, a0, a1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sm[0][(rl * tpr + c4) * 4 + e] = a0[e];
        sm[1][(rl * tpr + c4) * 4 + e] = a1[e];
    }
    __syncthreads();
    // threads 0..C-1 reduce over row lanes
    for (int c = t; c < C; c += RED_BLOCK) {
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < rpp; ++k) {
            s0 += sm[0][k * C + c];
            s1 += sm[1][k * C + c];
        }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s0;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}

template <class F>
__global__ __launch_bounds__(RED_BLOCK) void colreduce_small_kernel(F f, float* part, int64_t rows, int C) {
    // C <= SMALLC_MAX: one thread per row, all channels in registers
    __shared__ float sm[2][RED_BLOCK];
    const int t = threadIdx.x;
    const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float a0[SMALLC_MAX], a1[SMALLC_MAX];
#pragma unroll
    for (int c = 0; c < SMALLC_MAX; ++c) a0[c] = a1[c] = 0.f;
    for (int64_t r = r0 + t; r < r1; r += RED_BLOCK) {
#pragma unroll
        for (int c = 0; c < SMALLC_MAX; ++c)
            if (c < C) f.one(r * C + c, c, a0[c], a1[c]);
    }
#pragma unroll
    for (int c = 0; c < SMALLC_MAX; ++c) {
        if (c < C) {   // C is uniform
            sm[0][t] = a0[c];
            sm[1][t] = a1[c];
            __syncthreads();
            for (int s = RED_BLOCK / 2; s > 0; s >>= 1) {
                if (t < s) {
                    sm[0][t] += sm[0][t + s];
                    sm[1][t] += sm[1][t + s];
                }
                __syncthreads();
            }
            if (t == 0) {
                part[((size_t)blockIdx.x * 2 + 0) * C + c] = sm[0][0];
                part[((size_t)blockIdx.x * 2 + 1) * C + c] = sm[1][0];
            }
            __syncthreads();
        }
    }
}

template <class F>
__global__ __launch_bounds__(RED_BLOCK) void colreduce_any_kernel(F f, float* part, int64_t rows, int C) {
    // any C: one thread per column (strided), rows of the block's slab in sequence
    const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int c = threadIdx.x; c < C; c += RED_BLOCK) {
        float q0 = 0.f, q1 = 0.f;
        for (int64_t r = r0; r < r1; ++r) f.one(r * C + c, c, q0, q1);
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = q0;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = q1;
    }
}

template <class F>
static int launch_colreduce(F f, float* part, int64_t rows, int C, hipStream_t s) {
    const int blocks = red_blocks(rows, C);
    if (C % 4 == 0 && C >= 4 && C <= 1024 && RED_BLOCK % (C / 4) == 0) {
        hipLaunchKernelGGL((colreduce_vec_kernel<F>), dim3(blocks), dim3(RED_BLOCK), 0, s, f, part, rows, C);
    } else if (C <= SMALLC_MAX) {
        hipLaunchKernelGGL((colreduce_small_kernel<F>), dim3(blocks), dim3(RED_BLOCK), 0, s, f, part, rows, C);
    } else {
        hipLaunchKernelGGL((colreduce_any_kernel<F>), dim3(blocks), dim3(RED_BLOCK), 0, s, f, part, rows, C);
    }
    return blocks;
}

// ---- stage 2: combine partials in fp64 ----------------------------------------
// G: functor  __device__ void operator()(int c, double s0, double s1)
// CW = channels per workgroup (a power of two <= 64): the 256 threads are 256 / CW segments over the partial blocks -- the input
// BatchNorms have 1 or 3 channels and thousands of partial blocks (four segments of one thread each took 40-56 us there).
template <class G>
__global__ __launch_bounds__(256) void colfinal_kernel(G g, const float* part, int nblk, int C, int CW) {
    __shared__ double sm[2][256];
    const int t = threadIdx.x;
    const int ch = t & (CW - 1), seg = t / CW, nseg = 256 / CW;
    const int c = blockIdx.x * CW + ch;
    double s0 = 0.0, s1 = 0.0;
    if (c < C)
        for (int b = seg; b < nblk; b += nseg) {
            s0 += (double)part[((size_t)b * 2 + 0) * C + c];
            s1 += (double)part[((size_t)b * 2 + 1) * C + c];
        }
    sm[0][t] = s0;
    sm[1][t] = s1;
    __syncthreads();
    if (seg == 0 && c < C) {
        for (int k = 1; k < nseg; ++k) {                 // segments in order
            s0 += sm[0][k * CW + ch];
            s1 += sm[1][k * CW + ch];
        }
        g(c, s0, s1);
    }
}

template <class G>
static void launch_colfinal(G g, const float* part, int nblk, int C, hipStream_t s) {
    int cw = 64;
    while (cw > 1 && cw / 2 >= C) cw /= 2;
    hipLaunchKernelGGL((colfinal_kernel<G>), dim3((C + cw - 1) / cw), dim3(256), 0, s, g, part, nblk, C, cw);
}

// ---- column sum -----------------------------------------------------------------
struct SumF {
    const float* x;
    __device__ void vec(int64_t idx, int, float* a0, float*) const {
        const float4 v = *reinterpret_cast<const float4*>(x + idx);
        a0[0] += v.x; a0[1] += v.y; a0[2] += v.z; a0[3] += v.w;
    }
    __device__ void one(int64_t idx, int, float& q0, float&) const { q0 += x[idx]; }
};
struct SumG {
    float* out;
    __device__ void operator()(int c, double s0, double) const { out[c] = (float)s0; }
};
void colsum(const float* x, float* out, float* scratch, int64_t rows, int C, hipStream_t s) {
    const int nb = launch_colreduce(SumF{x}, scratch, rows, C, s);
    launch_colfinal(SumG{out}, scratch, nb, C, s);
}

// ---- batch-norm statistics ------------------------------------------------------
// sums are taken about the pivot x[0][c] to avoid E[x^2]-E[x]^2 cancellation
struct StatF {
    const float* x;
    __device__ void vec(int64_t idx, int c, float* a0, float* a1) const {
        const float4 v = *reinterpret_cast<const float4*>(x + idx);
        const float4 p = *reinterpret_cast<const float4*>(x + c);
        const float d0 = v.x - p.x, d1 = v.y - p.y, d2 = v.z - p.z, d3 = v.w - p.w;
        a0[0] += d0; a0[1] += d1; a0[2] += d2; a0[3] += d3;
        a1[0] += d0 * d0; a1[1] += d1 * d1; a1[2] += d2 * d2; a1[3] += d3 * d3;
    }
    __device__ void one(int64_t idx, int c, float& q0, float& q1) const {
        const float d = x[idx] - x[c];
        q0 += d;
        q1 += d * d;
    }
};
struct StatG {
    const float* x;
    const float* gamma;
    const float* beta;
    float* mean;
    float* var;
    float* scale;
    float* shift;
    double inv_n;
    float eps;
    __device__ void operator()(int c, double s0, double s1) const {
        const double dm = s0 * inv_n;
        double v = s1 * inv_n - dm * dm;
        if (v < 0.0) v = 0.0;
        const double m = (double)x[c] + dm;
        mean[c] = (float)m;
        var[c] = (float)v;
        const double sc = (double)gamma[c] / sqrt(v + (double)eps);
        scale[c] = (float)sc;
        shift[c] = (float)((double)beta[c] - m * sc);
    }
};
void bn_stats(const float* x, const float* gamma, const float* beta, float* mean, float* var,
              float* scale, float* shift, float* scratch, int64_t rows, int C, float eps,
              hipStream_t s) {
    if (bn_fast_ok(C)) {
        bn_stats_fast(x, gamma, beta, mean, var, scale, shift, scratch, rows, C, eps, 0, s);
        return;
    }
    const int nb = launch_colreduce(StatF{x}, scratch, rows, C, s);
    launch_colfinal(StatG{x, gamma, beta, mean, var, scale, shift, 1.0 / (double)rows, eps}, scratch, nb, C, s);
}

__global__ void bn_scale_shift_kernel(const float* gamma, const float* beta, const float* mean,
                                      const float* var, float* scale, float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const double sc = (double)gamma[c] / sqrt((double)var[c] + (double)eps);
        scale[c] = (float)sc;
        shift[c] = (float)((double)beta[c] - (double)mean[c] * sc);
    }
}
void bn_scale_shift(const float* gamma, const float* beta, const float* mean, const float* var,
                    float* scale, float* shift, int C, float eps, hipStream_t s) {
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, mean,
                       var, scale, shift, C, eps);
}

static inline int ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const float4* x, const float4* scale,
                                                           const float4* shift, float4* y,
                                                           int64_t n4, int c4n, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n);
        const float4 v = x[i], sc = scale[c], sh = shift[c];
        float4 o;
        o.x = v.x * sc.x + sh.x; o.y = v.y * sc.y + sh.y; o.z = v.z * sc.z + sh.z; o.w = v.w * sc.w + sh.w;
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        y[i] = o;
    }
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, const float* scale, const float* shift,
                                                       float* y, int64_t n, int C, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float o = x[i] * scale[c] + shift[c];
        if (relu) o = fmaxf(o, 0.f);
        y[i] = o;
    }
}
void bn_apply(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
              int relu, hipStream_t s) {
    if (bn_fast_ok(C)) {
        bn_apply_fast(x, scale, shift, y, rows, C, relu, s);
        return;
    }
    const int64_t n = rows * C;
    if (C % 4 == 0) {
        hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(scale),
                           reinterpret_cast<const float4*>(shift), reinterpret_cast<float4*>(y), n / 4,
                           C / 4, relu);
    } else {
        hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, scale, shift, y, n, C, relu);
    }
}

// ---- batch-norm backward ----------------------------------------------------------
struct BnBwdF {
    const float* x;
    const float* y;
    const float* dy;
    const float* mean;
    const float* var;
    float eps;
    int relu;
    __device__ void vec(int64_t idx, int c, float* a0, float* a1) const {
        const float4 xv = *reinterpret_cast<const float4*>(x + idx);
        float4 d = *reinterpret_cast<const float4*>(dy + idx);
        if (relu) {
            const float4 yv = *reinterpret_cast<const float4*>(y + idx);
            d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
            d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
        }
        const float4 m = *reinterpret_cast<const float4*>(mean + c);
        const float4 v = *reinterpret_cast<const float4*>(var + c);
        a0[0] += d.x; a0[1] += d.y; a0[2] += d.z; a0[3] += d.w;
        a1[0] += d.x * (xv.x - m.x) * rsqrtf(v.x + eps);
        a1[1] += d.y * (xv.y - m.y) * rsqrtf(v.y + eps);
        a1[2] += d.z * (xv.z - m.z) * rsqrtf(v.z + eps);
        a1[3] += d.w * (xv.w - m.w) * rsqrtf(v.w + eps);
    }
    __device__ void one(int64_t idx, int c, float& q0, float& q1) const {
        float d = dy[idx];
        if (relu && !(y[idx] > 0.f)) d = 0.f;
        q0 += d;
        q1 += d * (x[idx] - mean[c]) * rsqrtf(var[c] + eps);
    }
};
struct BnBwdG {
    float* dgamma;
    float* dbeta;
    __device__ void operator()(int c, double s0, double s1) const {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
    }
};
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* x, const float* y, const float* dy,
                                                           const float* gamma, const float* mean,
                                                           const float* var, const float* dgamma,
                                                           const float* dbeta, float* dx, int64_t n, int C,
                                                           float eps, float inv_rows, int relu, int training) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float d = dy[i];
        if (relu && !(y[i] > 0.f)) d = 0.f;
        const float rstd = rsqrtf(var[c] + eps);
        const float xh = (x[i] - mean[c]) * rstd;
        float o = d;
        if (training) o = d - dbeta[c] * inv_rows - xh * dgamma[c] * inv_rows;
        dx[i] = gamma[c] * rstd * o;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float4* x, const float4* y, const float4* dy,
                                                               const float* gamma, const float* mean,
                                                               const float* var, const float* dgamma,
                                                               const float* dbeta, float4* dx, int64_t n4,
                                                               int c4n, float eps, float inv_rows, int relu,
                                                               int training) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const float4 xv = x[i];
        float4 d = dy[i];
        if (relu) {
            const float4 yv = y[i];
            d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
            d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
        }
        const float dd[4] = {d.x, d.y, d.z, d.w};
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float oo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rstd = rsqrtf(var[c + e] + eps);
            const float xh = (xx[e] - mean[c + e]) * rstd;
            float o = dd[e];
            if (training) o = dd[e] - dbeta[c + e] * inv_rows - xh * dgamma[c + e] * inv_rows;
            oo[e] = gamma[c + e] * rstd * o;
        }
        dx[i] = make_float4(oo[0], oo[1], oo[2], oo[3]);
    }
}
void bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean,
            const float* var, float* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows,
            int C, float eps, int relu, int training, hipStream_t s) {
    const int nb = launch_colreduce(BnBwdF{x, y, dy, mean, var, eps, relu}, scratch, rows, C, s);
    launch_colfinal(BnBwdG{dgamma, dbeta}, scratch, nb, C, s);
    if (dx == nullptr) return;   // input BN of a tower: only dgamma/dbeta are needed
    const int64_t n = rows * C;
    const float inv_rows = (float)(1.0 / (double)rows);
    if (C % 4 == 0)
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
                           reinterpret_cast<const float4*>(dy), gamma, mean, var, dgamma, dbeta,
                           reinterpret_cast<float4*>(dx), n / 4, C / 4, eps, inv_rows, relu, training);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, dy, gamma, mean,
                           var, dgamma, dbeta, dx, n, C, eps, inv_rows, relu, training);
}

__global__ void bn_moving_update_kernel(float* moving, float* biased, const float* batch, int C,
                                        float momentum, int zero_debias, double corr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        if (zero_debias) {
            // tf assign_moving_average(zero_debias=True): biased -= (biased-value)*(1-m);
            // variable = biased / (1 - m^step)
            const float b = biased[c] - (biased[c] - batch[c]) * (1.f - momentum);
            biased[c] = b;
            moving[c] = (float)((double)b / corr);
        } else {
            moving[c] = moving[c] * momentum + batch[c] * (1.f - momentum);
        }
    }
}
void bn_moving_update(float* moving, float* biased, const float* batch, int C, float momentum,
                      int zero_debias, int step, hipStream_t s) {
    const double corr = 1.0 - pow((double)momentum, (double)step);
    hipLaunchKernelGGL(bn_moving_update_kernel, dim3((C + 255) / 256), dim3(256), 0, s, moving, biased,
                       batch, C, momentum, zero_debias, corr);
}

// every BatchNormalization of an engine in one launch: entry b = one (moving, biased, batch) triple, blockIdx.y = b
// Data-parallel step (l3_config.dp_moving = L3_DP_MOVING_REPLICAS): multi_gpu_model calls the template model once per replica
// (training_utils.py:141-157), so each BatchNormalization issues one moving-average update PER REPLICA on its one shared variable;
// `gathered` holds every replica's batch statistic and the updates are applied in replica order (the order the loop builds them).
__global__ void bn_moving_update_all_kernel(const BnMovingEntry* tab, float momentum, int zero_debias, double corr, const float* gathered,
                                            int replicas, int64_t stride) {
    const BnMovingEntry en = tab[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < en.C) {
        if (zero_debias) {
            float b = en.biased[c];
            for (int r = 0; r < replicas; ++r) {
                const float v = gathered != nullptr ? gathered[(int64_t)r * stride + en.off + c] : en.batch[c];
                b = b - (b - v) * (1.f - momentum);
            }
            en.biased[c] = b;
            en.moving[c] = (float)((double)b / corr);
        } else {
            float m = en.moving[c];
            for (int r = 0; r < replicas; ++r) {
                const float v = gathered != nullptr ? gathered[(int64_t)r * stride + en.off + c] : en.batch[c];
                m = m * momentum + v * (1.f - momentum);
            }
            en.moving[c] = m;
        }
    }
}
void bn_moving_update_all(const BnMovingEntry* tab_dev, int entries, int max_c, float momentum, int zero_debias, int64_t step,
                          hipStream_t s, const float* gathered, int replicas, int64_t stride) {
    const double corr = 1.0 - pow((double)momentum, (double)step);
    hipLaunchKernelGGL(bn_moving_update_all_kernel, dim3((max_c + 255) / 256, entries), dim3(256), 0, s, tab_dev, momentum,
                       zero_debias, corr, gathered, gathered != nullptr ? replicas : 1, stride);
}
__global__ void bn_moving_pack_kernel(const BnMovingEntry* tab, float* packed) {
    const BnMovingEntry en = tab[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < en.C) packed[en.off + c] = en.batch[c];
}
void bn_moving_pack(const BnMovingEntry* tab_dev, int entries, int max_c, float* packed, hipStream_t s) {
    hipLaunchKernelGGL(bn_moving_pack_kernel, dim3((max_c + 255) / 256, entries), dim3(256), 0, s, tab_dev, packed);
}

// ---- ReLU -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* x, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* y, const float* dy, float* dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
void relu_fwd(const float* x, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n);
}
void relu_bwd(const float* y, const float* dy, float* dx, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, y, dy, dx, n);
}

// ---- max pooling ------------------------------------------------------------------
// TF semantics: 'same' windows are clipped to the input (padding behaves as -inf);
// ties go to the first element in row-major window scan order.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, float* y, PoolGeom g) {
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        int64_t r = i / g.C;
        const int wo = (int)(r % g.Wo);
        r /= g.Wo;
        const int ho = (int)(r % g.Ho);
        const int n = (int)(r / g.Ho);
        const int h0 = ho * g.sh - g.padT, w0 = wo * g.sw - g.padL;
        float best = -INFINITY;
        for (int a = 0; a < g.ph; ++a) {
            const int h = h0 + a;
            if ((unsigned)h >= (unsigned)g.H) continue;
            for (int b = 0; b < g.pw; ++b) {
                const int w = w0 + b;
                if ((unsigned)w >= (unsigned)g.W) continue;
                best = fmaxf(best, x[((size_t)(n * g.H + h) * g.W + w) * g.C + c]);
            }
        }
        y[(size_t)n * g.out_batch_stride + ((size_t)ho * g.Wo + wo) * g.C + c] = best;
    }
}
// ---- global max pool (window == whole image: the (28,28) / (32,24) pools that end the towers,
// vision_model.py:203, audio_model.py:444) -------------------------------------------------
// One block per (sample, 64 channels): 16 pixel groups x 16 channel quads scan the image with
// coalesced float4 rows, keeping (max, first index); the 16 groups are merged through LDS with
// the same "first maximum in row-major order" rule as the generic kernel.
typedef float gp4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gp_scan(const float* __restrict__ xb, int P, int C, int pg, float best[4], int arg[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        best[e] = -INFINITY;
        arg[e] = 0x7fffffff;
    }
    for (int p = pg; p < P; p += 16) {
        const gp4 v = *reinterpret_cast<const gp4*>(xb + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] > best[e] || arg[e] == 0x7fffffff) {   // strictly greater: earlier pixel wins ties
                best[e] = v[e];
                arg[e] = p;
            }
    }
}
__device__ __forceinline__ void gp_merge(float (*sb)[64], int (*sa)[64], int pg, int cq, float best[4], int arg[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sb[pg][cq * 4 + e] = best[e];
        sa[pg][cq * 4 + e] = arg[e];
    }
    __syncthreads();
    if (pg == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float b = best[e];
            int a = arg[e];
            for (int k = 1; k < 16; ++k) {
                const float vb = sb[k][cq * 4 + e];
                const int va = sa[k][cq * 4 + e];
                if (vb > b || (vb == b && va < a)) {
                    b = vb;
                    a = va;
                }
            }
            sb[0][cq * 4 + e] = b;
            sa[0][cq * 4 + e] = a;
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void global_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int P,
                                                                 int C, int64_t out_batch_stride) {
    __shared__ float sb[16][64];
    __shared__ int sa[16][64];
    const int cb = blockIdx.x % (C / 64), n = blockIdx.x / (C / 64);
    const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
    float best[4];
    int arg[4];
    gp_scan(x + (size_t)n * P * C + cb * 64 + cq * 4, P, C, pg, best, arg);
    gp_merge(sb, sa, pg, cq, best, arg);
    if (threadIdx.x < 64) y[(size_t)n * out_batch_stride + cb * 64 + threadIdx.x] = sb[0][threadIdx.x];
}
__global__ __launch_bounds__(256) void global_maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, int P, int C,
                                                                 int64_t out_batch_stride) {
    __shared__ float sb[16][64];
    __shared__ int sa[16][64];
    const int cb = blockIdx.x % (C / 64), n = blockIdx.x / (C / 64);
    const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
    float best[4];
    int arg[4];
    const size_t base = (size_t)n * P * C + cb * 64 + cq * 4;
    gp_scan(x + base, P, C, pg, best, arg);
    gp_merge(sb, sa, pg, cq, best, arg);
    const gp4 d = *reinterpret_cast<const gp4*>(dy + (size_t)n * out_batch_stride + cb * 64 + cq * 4);
    int am[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) am[e] = sa[0][cq * 4 + e];
    for (int p = pg; p < P; p += 16) {
        gp4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = p == am[e] ? d[e] : 0.f;
        *reinterpret_cast<gp4*>(dx + base + (size_t)p * C) = o;
    }
}
static bool is_global_pool(const PoolGeom& g) {
    return g.Ho == 1 && g.Wo == 1 && g.padT == 0 && g.padL == 0 && g.ph == g.H && g.pw == g.W && g.C % 64 == 0 &&
           g.out_batch_stride % 4 == 0;
}

void maxpool_fwd(const float* x, float* y, const PoolGeom& g, hipStream_t s) {
    if (is_global_pool(g)) {
        hipLaunchKernelGGL(global_maxpool_fwd_kernel, dim3(g.N * (g.C / 64)), dim3(256), 0, s, x, y, g.H * g.W, g.C,
                           (int64_t)g.out_batch_stride);
        return;
    }
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, g);
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* x, const float* dy, float* dx, PoolGeom g) {
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        int64_t r = i / g.C;
        const int wo = (int)(r % g.Wo);
        r /= g.Wo;
        const int ho = (int)(r % g.Ho);
        const int n = (int)(r / g.Ho);
        const int h0 = ho * g.sh - g.padT, w0 = wo * g.sw - g.padL;
        float best = -INFINITY;
        int ba = -1, bb = -1;
        for (int a = 0; a < g.ph; ++a) {
            const int h = h0 + a;
            if ((unsigned)h >= (unsigned)g.H) continue;
            for (int b = 0; b < g.pw; ++b) {
                const int w = w0 + b;
                if ((unsigned)w >= (unsigned)g.W) continue;
                const float v = x[((size_t)(n * g.H + h) * g.W + w) * g.C + c];
                if (ba < 0 || v > best) { best = v; ba = a; bb = b; }
            }
        }
        const float d = dy[(size_t)n * g.out_batch_stride + ((size_t)ho * g.Wo + wo) * g.C + c];
        for (int a = 0; a < g.ph; ++a) {
            const int h = h0 + a;
            if ((unsigned)h >= (unsigned)g.H) continue;
            for (int b = 0; b < g.pw; ++b) {
                const int w = w0 + b;
                if ((unsigned)w >= (unsigned)g.W) continue;
                dx[((size_t)(n * g.H + h) * g.W + w) * g.C + c] = (a == ba && b == bb) ? d : 0.f;
            }
        }
    }
}
void maxpool_bwd(const float* x, const float* dy, float* dx, const PoolGeom& g, hipStream_t s) {
    if (is_global_pool(g)) {
        hipLaunchKernelGGL(global_maxpool_bwd_kernel, dim3(g.N * (g.C / 64)), dim3(256), 0, s, x, dy, dx, g.H * g.W,
                           g.C, (int64_t)g.out_batch_stride);
        return;
    }
    // windows do not overlap (stride >= pool); pixels outside every window get zero
    const bool covers = (g.Ho - 1) * g.sh - g.padT + g.ph >= g.H && (g.Wo - 1) * g.sw - g.padL + g.pw >= g.W &&
                        g.sh == g.ph && g.sw == g.pw;
    if (!covers) (void)hipMemsetAsync(dx, 0, (size_t)g.N * g.H * g.W * g.C * sizeof(float), s);
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, dy, dx, g);
}

// ---- first conv of a tower behind a trainable input BatchNorm --------------------------------------
// The conv's input is u = gamma * xhat + beta (audio_model.py:370, vision_model.py:124), zero padded.
// Its weight gradient and the BatchNorm's parameter gradients all follow from ONE weight-gradient
// pass over the augmented input [xhat, 1]:
//     G[tap][ci][co] = sum_p xhat[p + tap][ci] dy[p][co]      S[tap][co] = sum_{p: p + tap inside} dy[p][co]
//     dW = gamma G + beta S      dgamma[ci] = sum_{tap,co} W G      dbeta[ci] = sum_{tap,co} W S
// (dbeta = sum of the conv's input gradient, dgamma = sum of it times xhat, with the data-gradient
// convolution expanded) -- so the 3-channel data gradient of the first layer, which exists only to
// feed these two reductions, is never computed.
__global__ __launch_bounds__(256) void bn_xhat_ones_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float eps, float* __restrict__ xa,
                                                           int64_t rows, int C) {
    const int C1 = C + 1;
    const int64_t total = rows * C1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C1);
        const int64_t r = i / C1;
        xa[i] = c == C ? 1.f : (x[r * C + c] - mean[c]) * (1.f / sqrtf(var[c] + eps));
    }
}
void bn_xhat_ones(const float* x, const float* mean, const float* var, float eps, float* xa, int64_t rows, int C,
                  hipStream_t s) {
    hipLaunchKernelGGL(bn_xhat_ones_kernel, dim3(ew_blocks(rows * (C + 1))), dim3(256), 0, s, x, mean, var, eps, xa, rows, C);
}

// gaug: [taps][C + 1][Co] (channel C = the ones channel).  One block; fixed summation order.
__global__ __launch_bounds__(256) void first_conv_grads_kernel(const float* __restrict__ gaug, const float* __restrict__ w,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ dw, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dbias, int taps,
                                                               int C, int Co) {
    __shared__ double red[2][256];
    const int t = threadIdx.x;
    const int C1 = C + 1;
    for (int i = t; i < taps * C * Co; i += 256) {
        const int co = i % Co, ci = (i / Co) % C, tap = i / (Co * C);
        dw[i] = gamma[ci] * gaug[(tap * C1 + ci) * Co + co] + beta[ci] * gaug[(tap * C1 + C) * Co + co];
    }
    if (dbias != nullptr)
        for (int co = t; co < Co; co += 256) dbias[co] = gaug[((taps / 2) * C1 + C) * Co + co];   // centre tap: every pixel
    for (int ci = 0; ci < C; ++ci) {
        double sg = 0.0, sb = 0.0;
        for (int i = t; i < taps * Co; i += 256) {
            const int co = i % Co, tap = i / Co;
            const double wv = (double)w[(tap * C + ci) * Co + co];
            sg += wv * (double)gaug[(tap * C1 + ci) * Co + co];
            sb += wv * (double)gaug[(tap * C1 + C) * Co + co];
        }
        red[0][t] = sg;
        red[1][t] = sb;
        __syncthreads();
        if (t == 0) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 256; ++k) {
                a += red[0][k];
                b += red[1][k];
            }
            dgamma[ci] = (float)a;
            dbeta[ci] = (float)b;
        }
        __syncthreads();
    }
}
void first_conv_grads(const float* gaug, const float* w, const float* gamma, const float* beta, float* dw, float* dgamma,
                      float* dbeta, float* dbias, int taps, int C, int Co, hipStream_t s) {
    hipLaunchKernelGGL(first_conv_grads_kernel, dim3(1), dim3(256), 0, s, gaug, w, gamma, beta, dw, dgamma, dbeta, dbias, taps,
                       C, Co);
}

// ---- preprocessing ------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_video_kernel(const uint8_t* u8, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // train.py:186: 2 * img_as_float(u8).astype('float32') - 1 ; img_as_float = x/255 in float64
        const float f = (float)((double)u8[i] / 255.0);
        out[i] = 2.f * f - 1.f;
    }
}
__global__ __launch_bounds__(256) void preprocess_audio_kernel(const int16_t* pcm, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (float)pcm[i] / 32768.f;   // audio.py:28-31
}
__global__ void labels_onehot_kernel(const int32_t* lab, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)lab[i];
}
void preprocess_video(const uint8_t* u8, float* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_video_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, u8, out, n);
}
void preprocess_audio(const int16_t* pcm, float* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_audio_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, pcm, out, n);
}
void labels_onehot(const int32_t* lab, float* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(labels_onehot_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, lab, out, n);
}

// ---- dense head -------------------------------------------------------------------------
// One block per batch row; the K range is cut into DENSE_KS slices per output (a thread owns (slice, n)), combined in slice
// order through LDS: the 1024-long dependent fma chain of one thread per output was 107 us of an otherwise idle GPU
// between the towers and the loss.
constexpr int DENSE_KS = 8;
__global__ void dense_fwd_kernel(const float* x, const float* w, const float* b, float* y, int K, int N, int relu) {
    extern __shared__ float xs[];            // [K] the row, then [DENSE_KS][N] partial sums
    float* ps = xs + K;
    const int bi = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) xs[k] = x[(size_t)bi * K + k];
    __syncthreads();
    const int klen = (K + DENSE_KS - 1) / DENSE_KS;
    for (int id = threadIdx.x; id < DENSE_KS * N; id += blockDim.x) {
        const int ks = id / N, n = id - ks * N;
        const int k0 = ks * klen, k1 = min(K, k0 + klen);
        float acc = 0.f;
#pragma unroll 16
        for (int k = k0; k < k1; ++k) acc = fmaf(xs[k], w[(size_t)k * N + n], acc);      // unrolled: sixteen weight loads in flight
        ps[id] = acc;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float acc = ps[n];
#pragma unroll
        for (int ks = 1; ks < DENSE_KS; ++ks) acc += ps[ks * N + n];
        acc += b[n];
        if (relu) acc = fmaxf(acc, 0.f);
        y[(size_t)bi * N + n] = acc;
    }
}
void dense_fwd(const float* x, const float* w, const float* b, float* y, int B, int K, int N, int relu,
               hipStream_t s) {
    int threads = DENSE_KS * N;
    threads = threads >= 1024 ? 1024 : (threads < 64 ? 64 : ((threads + 63) / 64) * 64);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(B), dim3(threads), (K + DENSE_KS * N) * sizeof(float), s, x, w, b, y, K, N, relu);
}
__global__ void dense_bwd_w_kernel(const float* x, const float* dy, float* dw, float* db, int B, int K, int N) {
    const int k = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        // (sits between the loss and the towers' backward, where nothing else runs: eight samples' loads in flight instead of one
        // L2 round trip per sample of the sequential sum -- same order of additions, 16.4 -> 5.3 us per launch at 64 pairs)
        if (k < K) {
            float acc = 0.f;
#pragma unroll 8
            for (int b = 0; b < B; ++b) acc = fmaf(x[(size_t)b * K + k], dy[(size_t)b * N + n], acc);
            dw[(size_t)k * N + n] = acc;
        } else {
            float acc = 0.f;
#pragma unroll 8
            for (int b = 0; b < B; ++b) acc += dy[(size_t)b * N + n];
            db[n] = acc;
        }
    }
}
void dense_bwd_w(const float* x, const float* dy, float* dw, float* db, int B, int K, int N, hipStream_t s) {
    const int threads = N >= 256 ? 256 : (N < 64 ? 64 : ((N + 63) / 64) * 64);
    hipLaunchKernelGGL(dense_bwd_w_kernel, dim3(K + 1), dim3(threads), 0, s, x, dy, dw, db, B, K, N);
}
__global__ void dense_bwd_x_kernel(const float* dy, const float* w, float* dx, int K, int N) {
    extern __shared__ float ds[];
    const int bi = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) ds[n] = dy[(size_t)bi * N + n];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(ds[n], w[(size_t)k * N + n], acc);
        dx[(size_t)bi * K + k] = acc;
    }
}
void dense_bwd_x(const float* dy, const float* w, float* dx, int B, int K, int N, hipStream_t s) {
    hipLaunchKernelGGL(dense_bwd_x_kernel, dim3(B), dim3(256), N * sizeof(float), s, dy, w, dx, K, N);
}

// softmax + keras categorical_crossentropy(prob-space, clip 1e-7) for 2 classes, and
// its exact gradient w.r.t. the logits (through normalise -> clip -> log).
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const float* labels, float* probs,
                                                         float* dlogits, float* stats, int B, float gscale) {
    __shared__ float sl[256], sc[256];
    const float eps = 1e-7f;
    float lsum = 0.f, csum = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float z0 = logits[2 * b], z1 = logits[2 * b + 1];
        const float t0 = labels[2 * b], t1 = labels[2 * b + 1];
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        const float inv = 1.f / (e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        probs[2 * b] = p0;
        probs[2 * b + 1] = p1;
        const float sm = p0 + p1;
        const float q0 = p0 / sm, q1 = p1 / sm;
        const float c0 = fminf(fmaxf(q0, eps), 1.f - eps), c1 = fminf(fmaxf(q1, eps), 1.f - eps);
        lsum += -(t0 * logf(c0) + t1 * logf(c1));
        const int ap = p1 > p0 ? 1 : 0, at = t1 > t0 ? 1 : 0;
        csum += ap == at ? 1.f : 0.f;
        // backward
        float dq0 = (q0 >= eps && q0 <= 1.f - eps) ? -(t0 / c0) * gscale : 0.f;
        float dq1 = (q1 >= eps && q1 <= 1.f - eps) ? -(t1 / c1) * gscale : 0.f;
        const float dot = (dq0 * p0 + dq1 * p1) / (sm * sm);
        const float dp0 = dq0 / sm - dot, dp1 = dq1 / sm - dot;
        const float pd = dp0 * p0 + dp1 * p1;
        dlogits[2 * b] = p0 * (dp0 - pd);
        dlogits[2 * b + 1] = p1 * (dp1 - pd);
    }
    sl[threadIdx.x] = lsum;
    sc[threadIdx.x] = csum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sl[threadIdx.x] += sl[threadIdx.x + s];
            sc[threadIdx.x] += sc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = sl[0];
        stats[1] = sc[0];
    }
}
void softmax_ce(const float* logits, const float* labels, float* probs, float* dlogits, float* stats,
                int B, float gscale, hipStream_t s) {
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, s, logits, labels, probs, dlogits, stats, B, gscale);
}

// ---- sum of squares (L2 penalty) -----------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, int64_t n, float* part) {
    __shared__ float sm[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        acc = fmaf(v, v, acc);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ void sumsq_final_kernel(const float* part, int nb, float* out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)part[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sm[0];
}
size_t sumsq_scratch_floats(int64_t) { return 1024; }
void sumsq(const float* x, int64_t n, float* out, float* scratch, hipStream_t s) {
    int nb = (int)((n + 256 * 32 - 1) / (256 * 32));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, s, x, n, scratch);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, scratch, nb, out);
}
// All the regularised tensors of a model in two launches: block (b, seg) sums its slice of segment seg, block seg of the second
// kernel combines the segment's SUMSQ_BLOCKS partials in double.  Fixed grids, fixed order: deterministic.
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const float* base, SumsqSegs segs, float* part) {
    __shared__ float sm[4];
    const int seg = blockIdx.y;
    const float* x = base + segs.off[seg];
    const int64_t n = segs.n[seg];
    float acc = 0.f;
    // 16-byte loads, four of them in flight per thread, over the part of the range that is 16-byte aligned; the few floats in
    // front of and behind it go to the first threads of block 0
    const int64_t head = min(n, (int64_t)((4 - ((reinterpret_cast<uintptr_t>(x) >> 2) & 3)) & 3));
    const int64_t n4 = (n - head) >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)SUMSQ_BLOCKS * 256) {
        const float4 v = x4[i];
        acc = fmaf(v.x, v.x, acc);
        acc = fmaf(v.y, v.y, acc);
        acc = fmaf(v.z, v.z, acc);
        acc = fmaf(v.w, v.w, acc);
    }
    if (blockIdx.x == 0) {
        const int64_t rest = n - head - 4 * n4;
        if (threadIdx.x < head) acc = fmaf(x[threadIdx.x], x[threadIdx.x], acc);
        if (threadIdx.x < rest) {
            const float v = x[head + 4 * n4 + threadIdx.x];
            acc = fmaf(v, v, acc);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[seg * SUMSQ_BLOCKS + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ __launch_bounds__(64) void sumsq_multi_final_kernel(const float* part, float* out) {
    double acc = (double)part[blockIdx.x * SUMSQ_BLOCKS + threadIdx.x];
    static_assert(SUMSQ_BLOCKS == 64, "one partial per lane");
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)acc;
}
void sumsq_multi(const float* base, const SumsqSegs& segs, float* out, float* scratch, hipStream_t s) {
    if (segs.count <= 0) return;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(SUMSQ_BLOCKS, segs.count), dim3(256), 0, s, base, segs, scratch);
    hipLaunchKernelGGL(sumsq_multi_final_kernel, dim3(segs.count), dim3(64), 0, s, scratch, out);
}

// ---- Adam (keras 2.0.9) + L2 gradient --------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                   int64_t n_l2, float l2x2, float lr_t, float b1, float b2,
                                                   float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float w = p[i];
        float gi = g[i] * gscale;
        if (i < n_l2) gi = fmaf(l2x2, w, gi);       // d/dw [1e-5 * sum w^2] = 2e-5 * w
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = w - lr_t * mi / (sqrtf(vi) + eps);
    }
}
void adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_l2, float l2x2,
               float lr_t, float b1, float b2, float eps, float gscale, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, g, m, v, n, n_l2, l2x2, lr_t, b1,
                       b2, eps, gscale);
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
void fill(float* p, float v, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, v, n);
}

// fp32 -> bfloat16 storage (round to nearest even, as v_cvt_pk_bf16_f32 and the BatchNorm/pool writers do)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* x, __bf16* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (__bf16)x[i];
}
void cast_bf16(const float* x, void* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, (__bf16*)y, n);
}

}  // namespace l3
