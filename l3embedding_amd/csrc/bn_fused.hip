// bn_fused.hip -- fast-path BatchNorm kernels for power-of-two channel counts (all the
// 64/128/256/512-channel layers of the L3 towers).
//
// Same arithmetic as the generic kernels in elementwise.hip (keras BatchNormalization,
// Activation('relu'), MaxPooling2D((2,2), strides=2): l3embedding/audio_model.py:376-418,
// vision_model.py:130-172), restructured for HBM bandwidth:
//   * every thread owns one fixed channel quad (grid stride is a multiple of C/4), so the
//     per-channel coefficients live in registers: no modulo, no table loads in the loop;
//   * the ReLU mask is recomputed from the conv output (fma(x, scale, shift) > 0, the
//     very expression the forward used), so backward never reads the activation y;
//   * for Conv-BN-ReLU-MaxPool2x2 the BN apply, ReLU and pool are one kernel that writes
//     only the pooled tensor, and backward routes the pooled gradient through the
//     recomputed first-max of each window (TF tie rule) -- the full-resolution y and dy
//     tensors are never materialised;
//   * backward apply also emits the column sums of dx (the preceding conv's bias grad).
// Reductions stay two-stage and deterministic (fp32 per-block partials, fp64 combine).
#include "kernels.h"
#include "device_common.h"

namespace l3 {



// Every tensor these kernels touch is streamed -- read once, written once, hundreds of MB -- while the convolution kernels of the
// other tower, running beside them, live on what the L2 and the Infinity Cache hold of their filter slices and patches: the
// accesses are marked non-temporal (`nt`).  Same-box A/B of the two-tower step: 33.41 -> 32.87 ms (loads -0.33, stores a
// further -0.2; stores alone nothing).
// Element quad q of an output tensor that is either fp32 or (mixed-precision mode: tensors consumed only
// as bf16 convolution operands) bfloat16, rounded to nearest even exactly like the conv kernels'
// v_cvt_pk_bf16_f32 -- storing the rounded value is bit-identical to rounding it at operand fetch.
__device__ __forceinline__ void store_quad(void* base, int64_t q, f32x4 v, int obf) {
    if (obf) {
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        __builtin_nontemporal_store(h, reinterpret_cast<bf16x4*>(base) + q);
    } else {
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base) + q);
    }
}

// Element quad q of an INPUT tensor that is fp32 or (mixed-precision mode: the conv outputs the bf16 kernels
// write) bfloat16 -- widening is exact.
template <bool XBF>
__device__ __forceinline__ f32x4 load_quad(const void* base, int64_t q) {
    if constexpr (XBF) {
        const u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(base) + q);
        return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u)};
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + q);
    }
}

static constexpr int FB = 256;          // threads per block
static constexpr int FAST_MAX_BLOCKS = 1024;

bool bn_fast_ok(int C) { return C >= 4 && C <= 1024 && (C & (C - 1)) == 0; }

static inline int fast_blocks(int64_t work_items) {
    int64_t b = (work_items + FB * 8 - 1) / (FB * 8);
    if (b < 1) b = 1;
    if (b > FAST_MAX_BLOCKS) b = FAST_MAX_BLOCKS;
    return (int)b;
}

size_t bn_fast_scratch_floats(int C) { return (size_t)FAST_MAX_BLOCKS * 2 * C + 8 * (size_t)C + 64; }
const float* bn_bwd_fast_coeffs(const float* scratch, int C) { return scratch + (size_t)FAST_MAX_BLOCKS * 2 * C; }      // cA; cB = + C, cC = + 2 C

// block-level combine of per-thread channel-quad sums -> part[blk][0..1][C]
__device__ __forceinline__ void block_write_partials(f32x4 a0, f32x4 a1, float* part, int C4) {
    __shared__ f32x4 sm0[FB];
    __shared__ f32x4 sm1[FB];
    const int t = threadIdx.x;
    sm0[t] = a0;
    sm1[t] = a1;
    __syncthreads();
    if (t < C4) {
        f32x4 s0 = sm0[t], s1 = sm1[t];
        for (int k = t + C4; k < FB; k += C4) {
            s0 += sm0[k];
            s1 += sm1[k];
        }
        const int C = C4 * 4;
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + t * 4) = s0;
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + t * 4) = s1;
    }
}

__device__ __forceinline__ f32x4 bn_pre(f32x4 x, f32x4 sc, f32x4 sh) {
    return f32x4{fmaf(x.x, sc.x, sh.x), fmaf(x.y, sc.y, sh.y), fmaf(x.z, sc.z, sh.z), fmaf(x.w, sc.w, sh.w)};
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// ReLU placement modes: 0 none, 1 BN then ReLU (every layer but one), 2 ReLU then BN (the
// vision_model.py:138-139 quirk: Activation before BatchNormalization)

// ---- statistics ---------------------------------------------------------------------------
template <bool XBF>
__global__ __launch_bounds__(FB) void bn_stats_fast_kernel(const void* x, float* part, float* pivot_out, int64_t n4, int C4,
                                                          int prerelu) {
    const int64_t q0 = (int64_t)blockIdx.x * FB + threadIdx.x;
    const int c4 = (int)(q0 % C4);
    const f32x4 row0 = load_quad<XBF>(x, c4);
    const f32x4 pivot = prerelu ? relu4(row0) : row0;   // row 0: sums about a pivot avoid cancellation
    if (pivot_out != nullptr && q0 < C4) reinterpret_cast<f32x4*>(pivot_out)[c4] = row0;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t q = q0; q < n4; q += stride) {
        const f32x4 xv = load_quad<XBF>(x, q);
        const f32x4 d = (prerelu ? relu4(xv) : xv) - pivot;
        a0 += d;
        a1 += d * d;
    }
    block_write_partials(a0, a1, part, C4);
}

// generic stage 2: G(c, sum0, sum1).  A wave per channel; lanes stride over the partials, fp64 butterfly per wave.
// PRE: the partials are the fp64 rows fast_prefinal_kernel left in place, one per chunk of `rpb` rows.
template <class G, bool PRE>
__global__ __launch_bounds__(256) void fast_final_kernel(G g, const float* part, int nblk, int C, int rpb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        if constexpr (PRE) {
            const double* row = reinterpret_cast<const double*>(part + (size_t)b * rpb * 2 * C);
            s0 += row[c];
            s1 += row[C + c];
        } else {
            s0 += (double)part[((size_t)b * 2 + 0) * C + c];
            s1 += (double)part[((size_t)b * 2 + 1) * C + c];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
    }
    if (lane == 0) g(c, s0, s1);
}

// Stage 1.5, when a convolution epilogue left thousands of partial rows (one per tile block / patch: 25 088 rows of [2][64] = 12.8
// MB behind the first block of a 128-pair mixed-precision tower -- a wave per channel striding through them took 70 us): PRE_BLOCKS
// workgroups each sum a contiguous chunk of `rpb` rows in fp64 -- coalesced 16-byte reads, a thread adds its rows in order, the
// row groups of a workgroup are combined in order through LDS -- and leave [2][C] doubles IN PLACE, over the first two rows of
// their own chunk (which nobody else reads; the partials are consumed exactly once).  C / 2 must divide 256.
constexpr int PRE_BLOCKS = 256;
__global__ __launch_bounds__(256) void fast_prefinal_kernel(float* part, int nblk, int C, int rpb) {
    __shared__ double sm[256][4];
    const int t = threadIdx.x;
    const int qpr = C / 2, rpi = 256 / qpr;            // 16-byte pieces per row of [2][C]; rows per iteration
    const int q = t % qpr, rr = t / qpr;
    // the last chunk takes every remaining row: the host folds a one-row tail into it (a chunk's result is TWO float rows wide)
    const int r0 = blockIdx.x * rpb, r1 = blockIdx.x == gridDim.x - 1 ? nblk : r0 + rpb;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = r0 + rr; r < r1; r += rpi) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)r * 2 * C + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += (double)v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[t][e] = a[e];
    __syncthreads();                                    // ... and every read of this chunk is done
    if (t < qpr && r0 < r1) {
        for (int k = 1; k < rpi; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += sm[k * qpr + t][e];
        double* out = reinterpret_cast<double*>(part + (size_t)r0 * 2 * C) + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = a[e];
    }
}

template <class G>
static void launch_fast_final(G g, const float* part, int nblk, int C, hipStream_t s) {
    if (nblk > 2048 && C >= 8 && C <= 512 && 256 % (C / 2) == 0 && C % 2 == 0) {
        const int rpb = (nblk + PRE_BLOCKS - 1) / PRE_BLOCKS;
        // [2][C] doubles = two rows of [2][C] floats: a tail of one row (nblk % rpb == 1, e.g. nblk = 3841) joins the chunk before it
        // rather than have its result written one row past the partials
        const int chunks = nblk % rpb == 1 ? nblk / rpb : (nblk + rpb - 1) / rpb;
        hipLaunchKernelGGL(fast_prefinal_kernel, dim3(chunks), dim3(256), 0, s, const_cast<float*>(part), nblk, C, rpb);
        hipLaunchKernelGGL((fast_final_kernel<G, true>), dim3((C + 3) / 4), dim3(256), 0, s, g, part, chunks, C, rpb);
        return;
    }
    hipLaunchKernelGGL((fast_final_kernel<G, false>), dim3((C + 3) / 4), dim3(256), 0, s, g, part, nblk, C, 1);
}

struct StatFinal {
    const float* x;
    const float* gamma;
    const float* beta;
    float *mean, *var, *scale, *shift;
    double inv_n;
    float eps;
    int prerelu;
    __device__ void operator()(int c, double s0, double s1) const {
        const double dm = s0 * inv_n;
        double v = s1 * inv_n - dm * dm;
        if (v < 0.0) v = 0.0;
        const double m = (double)(prerelu ? fmaxf(x[c], 0.f) : x[c]) + dm;
        mean[c] = (float)m;
        var[c] = (float)v;
        const double sc = (double)gamma[c] / sqrt(v + (double)eps);
        scale[c] = (float)sc;
        shift[c] = (float)((double)beta[c] - m * sc);
    }
};

// Statistics from partials another kernel already produced (the Winograd conv epilogue): `part` holds
// nblk blocks of [sum, sum of squares][C] about the per-channel pivot `pivot[c]` (relu(pivot[c]) when
// prerelu), i.e. exactly what bn_stats_fast_kernel writes with pivot = row 0.
void bn_stats_from_partials(const float* part, int nblk, const float* pivot, const float* gamma, const float* beta,
                            float* mean, float* var, float* scale, float* shift, int64_t rows, int C, float eps,
                            int prerelu, hipStream_t s) {
    launch_fast_final(StatFinal{pivot, gamma, beta, mean, var, scale, shift, 1.0 / (double)rows, eps, prerelu}, part, nblk,
                      C, s);
}

void bn_stats_fast(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* scale,
                   float* shift, float* scratch, int64_t rows, int C, float eps, int prerelu, hipStream_t s, int x_bf16) {
    const int64_t n4 = rows * (C / 4);
    const int nb = fast_blocks(n4);
    if (x_bf16) {
        // row 0 (the pivot) is needed as floats by the finalize step: the kernel leaves it behind the partials
        float* pivot = scratch + (size_t)FAST_MAX_BLOCKS * 2 * C;
        hipLaunchKernelGGL(bn_stats_fast_kernel<true>, dim3(nb), dim3(FB), 0, s, (const void*)x, scratch, pivot, n4, C / 4, prerelu);
        launch_fast_final(StatFinal{pivot, gamma, beta, mean, var, scale, shift, 1.0 / (double)rows, eps, prerelu}, scratch, nb, C, s);
        return;
    }
    hipLaunchKernelGGL(bn_stats_fast_kernel<false>, dim3(nb), dim3(FB), 0, s, (const void*)x, scratch, (float*)nullptr, n4,
                       C / 4, prerelu);
    launch_fast_final(StatFinal{x, gamma, beta, mean, var, scale, shift, 1.0 / (double)rows, eps, prerelu}, scratch, nb, C, s);
}

// ---- apply (+ReLU) --------------------------------------------------------------------------

// Shape of this two-stream kernel (round 6, scripts/probes/stream_shapes.hip: y = relu(x s + t) over the first block's 822-MB bf16 /
// 1644-MB fp32 tensor): ONE access in flight per thread and a grid-stride loop -- the form it had -- reaches 5.0-5.2 TB/s (bf16, 8 B
// per lane) / 4.5-5.2 TB/s (fp32, 16 B per lane); a block that covers U * FB CONSECUTIVE quads with its U loads issued together reaches
// 6.0-6.35 / 5.7-6.3 TB/s whatever the grid (U = 8 / 4), and best with one tile per block; U accesses a grid stride apart (a power of
// two: the same HBM channel) gain half of that.  In the serialised fp32 step: 112.6 -> 99.5 us per launch.  The three-stream kernels
// below (x, dy -> dx: two loads in flight as they are) and the pooled ones (four) are at 5.6-6.1 TB/s already and gained nothing
// from the same treatment (profiles/r06_stream_shapes.txt).  FB % C4 == 0: a thread keeps its channel quad from tile to tile.
template <bool XBF>
__global__ __launch_bounds__(FB) void bn_apply_fast_kernel(const void* x, const f32x4* scale, const f32x4* shift,
                                                          void* y, int64_t n4, int C4, int relu, int obf) {
    constexpr int U = XBF ? 8 : 4;
    const int c4 = (int)threadIdx.x % C4;
    const f32x4 sc = scale[c4], sh = shift[c4];
    const int64_t stride = (int64_t)gridDim.x * (FB * U);
    for (int64_t q = (int64_t)blockIdx.x * (FB * U) + threadIdx.x; q < n4; q += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q + u * FB < n4) v[u] = load_quad<XBF>(x, q + u * FB);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q + u * FB < n4) {
                f32x4 o = bn_pre(v[u], sc, sh);
                if (relu) o = relu4(o);
                store_quad(y, q + u * FB, o, obf);
            }
    }
}
void bn_apply_fast(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C, int relu,
                   hipStream_t s, int out_bf16, int x_bf16) {
    const int64_t n4 = rows * (C / 4);
    const int tile = FB * (x_bf16 ? 8 : 4);
    int64_t nb = (n4 + tile - 1) / tile;        // one tile per block up to 2^16 blocks
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    auto k = x_bf16 ? bn_apply_fast_kernel<true> : bn_apply_fast_kernel<false>;
    hipLaunchKernelGGL(k, dim3((int)nb), dim3(FB), 0, s, (const void*)x, reinterpret_cast<const f32x4*>(scale),
                       reinterpret_cast<const f32x4*>(shift), (void*)y, n4, C / 4, relu, out_bf16);
}

// ---- fused BN + ReLU + MaxPool 2x2/2 forward -------------------------------------------------
// Window (i,j) covers rows 2i,2i+1 and cols 2j,2j+1 clipped to the input ('same' with an odd
// size clips the last window; 'valid' simply has fewer windows): TF pads 0 before for 2x2/2.
struct Pool2Geom {
    int N, H, W, C4, Ho, Wo, Hc, Wc;     // Hc/Wc = ceil(H/2), ceil(W/2): coverage grid for backward
    int64_t out_bs4;                     // pooled batch stride in quads
    int lgC4;                            // C4 is a power of two (bn_fast_ok)
    int small;                           // N * Hc * Wc < 2^31: window indices decompose in 32-bit arithmetic
};

// window index q = ((n * Hd + h) * Wd + w) * C4 + c4 -> (n, h, w); the quotient chain in 32 bits where it fits (three 64-bit
// divisions per window were ~250 VALU instructions of every iteration)
__device__ __forceinline__ void pool_index(const Pool2Geom& g, int64_t q, int Hd, int Wd, int& n, int& h, int& w) {
    if (g.small) {
        unsigned r = (unsigned)(q >> g.lgC4);
        const unsigned r1 = r / (unsigned)Wd;
        w = (int)(r - r1 * (unsigned)Wd);
        const unsigned r2 = r1 / (unsigned)Hd;
        h = (int)(r1 - r2 * (unsigned)Hd);
        n = (int)r2;
    } else {
        int64_t r = q >> g.lgC4;
        w = (int)(r % Wd);
        r /= Wd;
        h = (int)(r % Hd);
        n = (int)(r / Hd);
    }
}

// first-max-wins selection among 4 candidates with validity flags, per component
__device__ __forceinline__ int argmax4(float a, float b, float c, float d, bool bok, bool cok, bool dok) {
    int k = 0;
    float best = a;
    if (bok && b > best) { best = b; k = 1; }
    if (cok && c > best) { best = c; k = 2; }
    if (dok && d > best) { best = d; k = 3; }
    return k;
}

template <bool XBF, bool WIN>
__global__ __launch_bounds__(FB) void bn_relu_pool2_fwd_kernel(const void* __restrict__ x, const f32x4* scale, const f32x4* shift,
                                                              void* __restrict__ p, Pool2Geom g, int mode, int obf, void* __restrict__ xwin) {
    // a block covers U * FB consecutive windows; a thread's U windows are FB apart and their 4 U loads are issued together
    // (bn_apply_fast_kernel; profiles/r06_stream_shapes.txt): 120.2 -> 110.0 us per launch (fp32, 64 pairs), 127.1 -> 113.3 (bf16, 128)
    constexpr int U = 2;
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C4;
    const int c4 = (int)threadIdx.x % g.C4;
    const f32x4 sc = scale[c4], sh = shift[c4];
    const int64_t stride = (int64_t)gridDim.x * (FB * U);
    for (int64_t q = (int64_t)blockIdx.x * (FB * U) + threadIdx.x; q < total; q += stride) {
        f32x4 xs[U][4];
        int64_t po[U], wo_[U];
        bool w1[U], h1[U], live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = q + u * FB < total;
            int n = 0, ho = 0, wo = 0;
            pool_index(g, live[u] ? q + u * FB : q, g.Ho, g.Wo, n, ho, wo);
            const int h0 = 2 * ho, w0 = 2 * wo;
            h1[u] = h0 + 1 < g.H; w1[u] = w0 + 1 < g.W;
            const int64_t base = ((int64_t)(n * g.H + h0) * g.W + w0) * g.C4 + c4;
            // clamp out-of-range taps onto the (0,0) tap: max() is unaffected by duplicates
            const int64_t dw = w1[u] ? g.C4 : 0, dh = h1[u] ? (int64_t)g.W * g.C4 : 0;
            xs[u][0] = load_quad<XBF>(x, base); xs[u][1] = load_quad<XBF>(x, base + dw);
            xs[u][2] = load_quad<XBF>(x, base + dh); xs[u][3] = load_quad<XBF>(x, base + dh + dw);
            po[u] = (int64_t)n * g.out_bs4 + ((int64_t)ho * g.Wo + wo) * g.C4 + c4;
            wo_[u] = ((int64_t)(n * g.Ho + ho) * g.Wo + wo) * g.C4 + c4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            f32x4 x00 = xs[u][0], x01 = xs[u][1], x10 = xs[u][2], x11 = xs[u][3];
            const bool h1ok = h1[u], w1ok = w1[u];
            if (mode == 2) { x00 = relu4(x00); x01 = relu4(x01); x10 = relu4(x10); x11 = relu4(x11); }
            const f32x4 v00 = bn_pre(x00, sc, sh), v01 = bn_pre(x01, sc, sh);
            const f32x4 v10 = bn_pre(x10, sc, sh), v11 = bn_pre(x11, sc, sh);
            f32x4 m;
            m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
            m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
            m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
            m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
            if (mode == 1) m = relu4(m);
            store_quad(p, po[u], m, obf);
            if constexpr (WIN) {
                // the winner of the window as the backward kernels choose it (bn_bwd_*_fast_kernel: first maximum in row-major
                // order -- BN -> ReLU: of the post-ReLU values, and where every value is <= 0 nothing passes and any element
                // will do; ReLU -> BN (mode 2): of the BatchNorm outputs, and the element kept is the rectified input)
                const f32x4 r00 = mode == 1 ? relu4(v00) : v00, r01 = mode == 1 ? relu4(v01) : v01;
                const f32x4 r10 = mode == 1 ? relu4(v10) : v10, r11 = mode == 1 ? relu4(v11) : v11;
                const bool ok11 = h1ok && w1ok;
                f32x4 xw;
#define L3_WIN(comp)                                                                                  \
    {                                                                                                 \
        const int k = argmax4(r00.comp, r01.comp, r10.comp, r11.comp, w1ok, h1ok, ok11);              \
        xw.comp = k == 0 ? x00.comp : k == 1 ? x01.comp : k == 2 ? x10.comp : x11.comp;               \
    }
                L3_WIN(x) L3_WIN(y) L3_WIN(z) L3_WIN(w)
#undef L3_WIN
                store_quad(xwin, wo_[u], xw, XBF ? 1 : 0);
            }
        }
    }
}

static Pool2Geom make_pool2(int N, int H, int W, int C, int Ho, int Wo, int64_t out_batch_stride) {
    Pool2Geom g;
    g.N = N; g.H = H; g.W = W; g.C4 = C / 4; g.Ho = Ho; g.Wo = Wo;
    g.Hc = (H + 1) / 2; g.Wc = (W + 1) / 2;
    g.out_bs4 = out_batch_stride / 4;
    g.lgC4 = 0;
    while ((1 << g.lgC4) < g.C4) ++g.lgC4;
    g.small = (int64_t)N * g.Hc * g.Wc < ((int64_t)1 << 31) && (int64_t)N * H * W < ((int64_t)1 << 31);
    return g;
}

void bn_relu_pool2_fwd(const float* x, const float* scale, const float* shift, float* p, int N, int H, int W, int C,
                       int Ho, int Wo, int64_t out_batch_stride, int mode, hipStream_t s, int out_bf16, int x_bf16, void* xwin) {
    const Pool2Geom g = make_pool2(N, H, W, C, Ho, Wo, out_batch_stride);
    const int64_t total = (int64_t)N * Ho * Wo * g.C4;
    int64_t nb = (total + FB * 2 - 1) / (FB * 2);       // one tile of two windows per thread per block up to 2^16 blocks
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    auto k = xwin != nullptr ? (x_bf16 ? bn_relu_pool2_fwd_kernel<true, true> : bn_relu_pool2_fwd_kernel<false, true>)
                             : (x_bf16 ? bn_relu_pool2_fwd_kernel<true, false> : bn_relu_pool2_fwd_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3((int)nb), dim3(FB), 0, s, (const void*)x, reinterpret_cast<const f32x4*>(scale),
                       reinterpret_cast<const f32x4*>(shift), (void*)p, g, mode, out_bf16, xwin);
}

// ---- backward ---------------------------------------------------------------------------------
// dz = upstream gradient at the BN(+ReLU) output.  Plain mode: dz = relu ? dy*(pre>0) : dy.
// Pooled mode: dz = dP at the first max of the window (row-major scan), times (pre>0).
struct BwdCoef {          // per channel, written by the finalize step
    float *A, *B, *Cc;    // dx = A*dz + B*x + Cc
};

template <bool POOL, bool XBF, bool DYBF>
__global__ __launch_bounds__(FB) void bn_bwd_reduce_fast_kernel(const void* x, const void* dy, const f32x4* scale,
                                                               const f32x4* shift, const f32x4* mean,
                                                               const f32x4* var, float* part, int64_t n4,
                                                               Pool2Geom g, float eps, int relu) {
    const int64_t q0 = (int64_t)blockIdx.x * FB + threadIdx.x;
    const int c4 = (int)(q0 % g.C4);
    const f32x4 sc = scale[c4], sh = shift[c4], mu = mean[c4];
    const f32x4 vv = var[c4];
    const f32x4 rs = {rsqrtf(vv.x + eps), rsqrtf(vv.y + eps), rsqrtf(vv.z + eps), rsqrtf(vv.w + eps)};
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * FB;
    if constexpr (!POOL) {
        for (int64_t q = q0; q < n4; q += stride) {
            f32x4 xv = load_quad<XBF>(x, q);
            f32x4 d = load_quad<DYBF>(dy, q);
            if (relu == 1) {
                const f32x4 pre = bn_pre(xv, sc, sh);
                d.x = pre.x > 0.f ? d.x : 0.f; d.y = pre.y > 0.f ? d.y : 0.f;
                d.z = pre.z > 0.f ? d.z : 0.f; d.w = pre.w > 0.f ? d.w : 0.f;
            } else if (relu == 2) {
                xv = relu4(xv);
            }
            a0 += d;
            a1 += d * ((xv - mu) * rs);
        }
    } else {
        const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.C4;      // only real windows carry gradient
        for (int64_t q = q0; q < total; q += stride) {
            int n, ho, wo;
            pool_index(g, q, g.Ho, g.Wo, n, ho, wo);
            const int h0 = 2 * ho, w0 = 2 * wo;
            const bool h1ok = h0 + 1 < g.H, w1ok = w0 + 1 < g.W;
            const int64_t base = ((int64_t)(n * g.H + h0) * g.W + w0) * g.C4 + c4;
            const int64_t dw = w1ok ? g.C4 : 0, dh = h1ok ? (int64_t)g.W * g.C4 : 0;
            f32x4 x00 = load_quad<XBF>(x, base), x01 = load_quad<XBF>(x, base + dw), x10 = load_quad<XBF>(x, base + dh),
                  x11 = load_quad<XBF>(x, base + dh + dw);
            if (relu == 2) { x00 = relu4(x00); x01 = relu4(x01); x10 = relu4(x10); x11 = relu4(x11); }
            f32x4 p00 = bn_pre(x00, sc, sh), p01 = bn_pre(x01, sc, sh), p10 = bn_pre(x10, sc, sh),
                  p11 = bn_pre(x11, sc, sh);
            const float gate = relu == 2 ? -INFINITY : 0.f;     // post-ReLU kills the gradient where pre <= 0
            if (relu != 2) { p00 = relu4(p00); p01 = relu4(p01); p10 = relu4(p10); p11 = relu4(p11); }
            const f32x4 dp = load_quad<DYBF>(dy, (int64_t)n * g.out_bs4 + ((int64_t)ho * g.Wo + wo) * g.C4 + c4);
            const bool ok11 = h1ok && w1ok;
#define L3_RED(comp)                                                                                      \
    {                                                                                                     \
        const int k = argmax4(p00.comp, p01.comp, p10.comp, p11.comp, w1ok, h1ok, ok11);                  \
        const float pk = k == 0 ? p00.comp : k == 1 ? p01.comp : k == 2 ? p10.comp : p11.comp;            \
        const float xk = k == 0 ? x00.comp : k == 1 ? x01.comp : k == 2 ? x10.comp : x11.comp;            \
        const float d = pk > gate ? dp.comp : 0.f;                                                        \
        a0.comp += d;                                                                                     \
        a1.comp += d * ((xk - mu.comp) * rs.comp);                                                        \
    }
            L3_RED(x) L3_RED(y) L3_RED(z) L3_RED(w)
#undef L3_RED
        }
    }
    block_write_partials(a0, a1, part, g.C4);
}

struct BwdFinal {
    const float* gamma;
    const float* mean;
    const float* var;
    float *dgamma, *dbeta, *A, *B, *Cc;
    double inv_n;
    float eps;
    int training;
    __device__ void operator()(int c, double s0, double s1) const {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        const double gr = (double)gamma[c] * rstd;
        A[c] = (float)gr;
        if (training) {
            // dx = g*rstd*(dz - dbeta/n - xhat*dgamma/n),  xhat = (x-mean)*rstd
            B[c] = (float)(-gr * rstd * s1 * inv_n);
            Cc[c] = (float)(gr * (-s0 * inv_n + (double)mean[c] * rstd * s1 * inv_n));
        } else {
            B[c] = 0.f;
            Cc[c] = 0.f;
        }
    }
};

template <bool POOL, bool XBF, bool DYBF>
__global__ __launch_bounds__(FB) void bn_bwd_apply_fast_kernel(const void* x, const void* dy, const f32x4* scale,
                                                              const f32x4* shift, const f32x4* cA, const f32x4* cB,
                                                              const f32x4* cC, void* dx, float* part, int64_t n4,
                                                              Pool2Geom g, int relu, int obf) {
    const int64_t q0 = (int64_t)blockIdx.x * FB + threadIdx.x;
    const int c4 = (int)(q0 % g.C4);
    const f32x4 sc = scale[c4], sh = shift[c4], A = cA[c4], B = cB[c4], Cc = cC[c4];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * FB;
    if constexpr (!POOL) {
        for (int64_t q = q0; q < n4; q += stride) {
            const f32x4 xr = load_quad<XBF>(x, q);
            f32x4 xv = xr;
            f32x4 d = load_quad<DYBF>(dy, q);
            if (relu == 1) {
                const f32x4 pre = bn_pre(xv, sc, sh);
                d.x = pre.x > 0.f ? d.x : 0.f; d.y = pre.y > 0.f ? d.y : 0.f;
                d.z = pre.z > 0.f ? d.z : 0.f; d.w = pre.w > 0.f ? d.w : 0.f;
            } else if (relu == 2) {
                xv = relu4(xv);
            }
            f32x4 o = A * d + (B * xv + Cc);
            if (relu == 2) {
                o.x = xr.x > 0.f ? o.x : 0.f; o.y = xr.y > 0.f ? o.y : 0.f;
                o.z = xr.z > 0.f ? o.z : 0.f; o.w = xr.w > 0.f ? o.w : 0.f;
            }
            store_quad(dx, q, o, obf);
            a0 += o;
        }
    } else {
        // coverage grid: every input pixel belongs to exactly one (possibly gradient-free) cell.  A block covers U * FB consecutive
        // cells; a thread's U cells are FB apart and their 5 U loads are issued together (bn_apply_fast_kernel): two cells on
        // bf16-stored tensors (205.9 -> 195.4 us per launch at 128 pairs), one on fp32 ones (two: 184.0 -> 188.9 us at 64 pairs)
        constexpr int U = XBF ? 2 : 1;
        const int64_t total = (int64_t)g.N * g.Hc * g.Wc * g.C4;
        for (int64_t q = (int64_t)blockIdx.x * (FB * U) + threadIdx.x; q < total; q += stride * U) {
            f32x4 rs[U][4], dps[U];
            int64_t bases[U];
            bool w1[U], h1[U], lives[U], on[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                on[u] = q + u * FB < total;
                int n, hc, wc;
                pool_index(g, on[u] ? q + u * FB : q, g.Hc, g.Wc, n, hc, wc);
                const int h0 = 2 * hc, w0 = 2 * wc;
                h1[u] = h0 + 1 < g.H; w1[u] = w0 + 1 < g.W;
                lives[u] = hc < g.Ho && wc < g.Wo;          // 'valid' leftovers receive no gradient
                const int64_t base = ((int64_t)(n * g.H + h0) * g.W + w0) * g.C4 + c4;
                const int64_t dw = w1[u] ? g.C4 : 0, dh = h1[u] ? (int64_t)g.W * g.C4 : 0;
                bases[u] = base;
                rs[u][0] = load_quad<XBF>(x, base); rs[u][1] = load_quad<XBF>(x, base + dw);
                rs[u][2] = load_quad<XBF>(x, base + dh); rs[u][3] = load_quad<XBF>(x, base + dh + dw);
                dps[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (lives[u]) dps[u] = load_quad<DYBF>(dy, (int64_t)n * g.out_bs4 + ((int64_t)hc * g.Wo + wc) * g.C4 + c4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!on[u]) continue;
                const bool h1ok = h1[u], w1ok = w1[u], live = lives[u];
                const int64_t base = bases[u];
                const int64_t dw = w1ok ? g.C4 : 0, dh = h1ok ? (int64_t)g.W * g.C4 : 0;
                const f32x4 r00 = rs[u][0], r01 = rs[u][1], r10 = rs[u][2], r11 = rs[u][3];
                f32x4 x00 = r00, x01 = r01, x10 = r10, x11 = r11;
                if (relu == 2) { x00 = relu4(x00); x01 = relu4(x01); x10 = relu4(x10); x11 = relu4(x11); }
                f32x4 d00 = {0.f, 0.f, 0.f, 0.f}, d01 = d00, d10 = d00, d11 = d00;
                if (live) {
                    f32x4 p00 = bn_pre(x00, sc, sh), p01 = bn_pre(x01, sc, sh), p10 = bn_pre(x10, sc, sh),
                          p11 = bn_pre(x11, sc, sh);
                    const float gate = relu == 2 ? -INFINITY : 0.f;
                    if (relu != 2) { p00 = relu4(p00); p01 = relu4(p01); p10 = relu4(p10); p11 = relu4(p11); }
                    const f32x4 dp = dps[u];
                    const bool ok11 = h1ok && w1ok;
#define L3_APP(comp)                                                                                      \
    {                                                                                                     \
        const int k = argmax4(p00.comp, p01.comp, p10.comp, p11.comp, w1ok, h1ok, ok11);                  \
        const float pk = k == 0 ? p00.comp : k == 1 ? p01.comp : k == 2 ? p10.comp : p11.comp;            \
        const float d = pk > gate ? dp.comp : 0.f;                                                        \
        d00.comp = k == 0 ? d : 0.f;                                                                      \
        d01.comp = k == 1 ? d : 0.f;                                                                      \
        d10.comp = k == 2 ? d : 0.f;                                                                      \
        d11.comp = k == 3 ? d : 0.f;                                                                      \
    }
                    L3_APP(x) L3_APP(y) L3_APP(z) L3_APP(w)
#undef L3_APP
                }
                auto gate2 = [&](f32x4 o, f32x4 r) {
                    if (relu == 2) {
                        o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f;
                        o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
                    }
                    return o;
                };
                const f32x4 o00 = gate2(A * d00 + (B * x00 + Cc), r00);
                store_quad(dx, base, o00, obf);
                a0 += o00;
                if (w1ok) {
                    const f32x4 o = gate2(A * d01 + (B * x01 + Cc), r01);
                    store_quad(dx, base + dw, o, obf);
                    a0 += o;
                }
                if (h1ok) {
                    const f32x4 o = gate2(A * d10 + (B * x10 + Cc), r10);
                    store_quad(dx, base + dh, o, obf);
                    a0 += o;
                }
                if (h1ok && w1ok) {
                    const f32x4 o = gate2(A * d11 + (B * x11 + Cc), r11);
                    store_quad(dx, base + dh + dw, o, obf);
                    a0 += o;
                }
            }
        }
    }
    if (part != nullptr) block_write_partials(a0, a0, part, g.C4);
}

struct SumFinal {
    float* out;
    __device__ void operator()(int c, double s0, double) const { out[c] = (float)s0; }
};

// scratch: bn_fast_scratch_floats(C) floats
void bn_bwd_fast(const float* x, const float* scale, const float* shift, const float* mean, const float* var,
                 const float* gamma, const float* dy, int pooled, int N, int H, int W, int C, int Ho, int Wo,
                 int64_t dy_batch_stride, float* dx, float* dgamma, float* dbeta, float* dbias, float* scratch,
                 float eps, int relu, int training, hipStream_t s, int dx_bf16, int x_bf16, int dy_bf16,
                 const float* ready_part, int ready_blocks) {
    const int64_t rows = (int64_t)N * H * W;
    const int64_t n4 = rows * (C / 4);
    Pool2Geom g = make_pool2(N, H, W, C, pooled ? Ho : H, pooled ? Wo : W, pooled ? dy_batch_stride : (int64_t)H * W * C);
    float* part = scratch;
    float* cA = scratch + (size_t)FAST_MAX_BLOCKS * 2 * C;
    float* cB = cA + C;
    float* cC = cB + C;
    const void* xv = (const void*)x;
    const void* dyv = (const void*)dy;
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(scale);
    const f32x4* sh4 = reinterpret_cast<const f32x4*>(shift);
    const int64_t work_r = pooled ? (int64_t)N * Ho * Wo * (C / 4) : n4;
    const int nb_r = fast_blocks(work_r);
    const int variant = (pooled ? 4 : 0) | (x_bf16 ? 2 : 0) | (dy_bf16 ? 1 : 0);
    using ReduceFn = void (*)(const void*, const void*, const f32x4*, const f32x4*, const f32x4*, const f32x4*, float*, int64_t,
                              Pool2Geom, float, int);
    static const ReduceFn reduce_fns[8] = {
        bn_bwd_reduce_fast_kernel<false, false, false>, bn_bwd_reduce_fast_kernel<false, false, true>,
        bn_bwd_reduce_fast_kernel<false, true, false>,  bn_bwd_reduce_fast_kernel<false, true, true>,
        bn_bwd_reduce_fast_kernel<true, false, false>,  bn_bwd_reduce_fast_kernel<true, false, true>,
        bn_bwd_reduce_fast_kernel<true, true, false>,   bn_bwd_reduce_fast_kernel<true, true, true>};
    if (ready_part != nullptr) {
        // the data-gradient kernel that wrote dy left the reduction partials in its epilogue (kernels.h BnBwdFuse)
        launch_fast_final(BwdFinal{gamma, mean, var, dgamma, dbeta, cA, cB, cC, 1.0 / (double)rows, eps, training}, ready_part,
                          ready_blocks, C, s);
    } else {
        hipLaunchKernelGGL(reduce_fns[variant], dim3(nb_r), dim3(FB), 0, s, xv, dyv, sc4, sh4, reinterpret_cast<const f32x4*>(mean),
                           reinterpret_cast<const f32x4*>(var), part, n4, g, eps, relu);
        launch_fast_final(BwdFinal{gamma, mean, var, dgamma, dbeta, cA, cB, cC, 1.0 / (double)rows, eps, training}, part,
                          nb_r, C, s);
    }
    if (dx == nullptr) return;
    const int64_t work_a = pooled ? (int64_t)N * g.Hc * g.Wc * (C / 4) : n4;
    const int nb_a = fast_blocks(work_a);
    float* part2 = dbias ? part : nullptr;
    using ApplyFn = void (*)(const void*, const void*, const f32x4*, const f32x4*, const f32x4*, const f32x4*, const f32x4*, void*,
                             float*, int64_t, Pool2Geom, int, int);
    static const ApplyFn apply_fns[8] = {
        bn_bwd_apply_fast_kernel<false, false, false>, bn_bwd_apply_fast_kernel<false, false, true>,
        bn_bwd_apply_fast_kernel<false, true, false>,  bn_bwd_apply_fast_kernel<false, true, true>,
        bn_bwd_apply_fast_kernel<true, false, false>,  bn_bwd_apply_fast_kernel<true, false, true>,
        bn_bwd_apply_fast_kernel<true, true, false>,   bn_bwd_apply_fast_kernel<true, true, true>};
    hipLaunchKernelGGL(apply_fns[variant], dim3(nb_a), dim3(FB), 0, s, xv, dyv, sc4, sh4, reinterpret_cast<const f32x4*>(cA),
                       reinterpret_cast<const f32x4*>(cB), reinterpret_cast<const f32x4*>(cC), (void*)dx, part2, n4, g, relu,
                       dx_bf16);
    if (dbias) launch_fast_final(SumFinal{dbias}, part, nb_a, C, s);
}

}  // namespace l3
