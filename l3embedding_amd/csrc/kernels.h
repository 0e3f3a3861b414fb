// kernels.h -- launch-function declarations shared by the HIP sources of libl3hip.so.
// All tensors are fp32, NHWC; all launches go to the given stream and never sync.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l3 {

struct ConvGeom {
    int N, H, W, Cin;      // input tensor
    int Ho, Wo, Cout;      // output tensor
    int KH, KW, padT, padL;
    int f2x2 = 0;          // 1: the fp32 Winograd path takes F(2x2,3x3) whatever the channel count (l3_config.fp32_conv);
                           // 2: F(2x2,3x3) with split-bf16 operands on the bf16 matrix pipe (conv_wino_bx6.hip) where it applies
    int solo = 0;          // 1: nothing else is queued beside this launch (a tower on its own: l3_tower_step, l3_embed_*, the operator
                           // entry points): the F(4x4,3x3) kernel then splits its last, partial round of tile blocks over channel slices
                           // (conv_wino4_launch).  In the two-tower training step the other tower's kernels fill that tail for free and
                           // the split costs more CU time than it saves (measured: +1 % per step), so the engine leaves it 0 there.
    int dynamic = 0;       // 1: a persistent grid hands its tile blocks out through work counters (device_common.h wq_*) instead of the
                           // static stride: set by the engine while a communicator exists -- a collective that holds CUs makes workgroups
                           // start late, and under the static stride every late workgroup still owes its whole list.  Without collectives
                           // the static stride is 0.25 ms per step faster (profiles/r06_dp_footprint.txt), so it stays the plain step's.
    float* tail_scratch = nullptr;      // (solo launches) scratch of that channel-slice tail, owned by the caller: an engine hands its
    size_t tail_scratch_bytes = 0;      // own buffer of conv_wino4_tail_scratch_bytes(); null = the library's per-(device, stream) pool
};
size_t conv_wino4_tail_scratch_bytes();     // enough for the tail of any layer

// y = conv(x, w) + bias   (implicit GEMM on v_mfma_f32_32x32x2_f32).
// w is (KH*KW*Cin, Cout) row-major == keras HWIO.  bias may be null.
// wino_u (optional): conv_wino_transform_weights() of the filter; used when conv_wino_ok(g).
// bn_part (optional, Winograd path only): the kernel also leaves per-block sum / sum-of-squares
// partials of its output (bn_mode 1) or of relu(output) (bn_mode 2) about the pivot bias[c], in
// conv_wino_stat_blocks(g) blocks of the bn_fused.hip partial layout, for bn_stats_from_partials().
// bn_bwd (optional, Winograd path, with bn_part): the launch is the DATA GRADIENT that produces dL/dy of a BatchNorm(+ReLU)
// whose input is bn_bwd->x; the epilogue then leaves that BatchNorm's backward reduction partials (sum of the masked
// gradient, sum of masked gradient * x_hat: what bn_bwd_fast's reduce pass computes from x and dy) in bn_part instead --
// the separate pass over x and dy disappears (bn_bwd_fast(..., ready_part, ready_blocks)).
struct BnBwdFuse {
    const float* x;                       // the BatchNorm's input (N, H, W, C) = geometry of the launch's output
    const float *scale, *shift;           // forward scale / shift (the ReLU mask is recomputed from them)
    const float *mean, *var;
    float eps;
    int relu;                             // 0 none, 1 BN then ReLU
};
void conv_fwd(const float* x, const float* w, const float* bias, float* y, const ConvGeom& g,
              hipStream_t s, const float* wino_u = nullptr, float* bn_part = nullptr, int bn_mode = 0,
              const BnBwdFuse* bn_bwd = nullptr);
// Winograd path for 3x3 / pad 1 convs with Cin % 8 == 0, Cout % 64 == 0: F(4x4,3x3) (conv_wino4.hip) for Cin >= 64 -- all
// 14 such layers of the VGG-style towers --, F(2x2,3x3) (conv_wino.hip) below that or when g.f2x2 is set; chosen per launch
// inside these entry points.  conv_wino_floats() sizes U for the larger form (36 positions per filter) either way.
bool conv_wino_ok(const ConvGeom& g);
size_t conv_wino_floats(const ConvGeom& g);          // floats of U, 0 if not eligible
double conv_wino_executed_flops(const ConvGeom& g);  // MFMA flops the Winograd kernel issues (16 per tile, c, k)
// U = G g G^T in [pos][Cin/4][Cout][4] order.  from_fwd_for_dgrad: g describes the DATA-GRADIENT conv
// (Cin = forward Cout, Cout = forward Cin) and w is the forward filter (flip + transpose folded in).
void conv_wino_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s);
void conv_wino_fwd(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, hipStream_t s,
                   float* stat_part = nullptr, int stat_mode = 0, const BnBwdFuse* bn_bwd = nullptr);
// The Winograd kernel runs as a persistent grid (one block per CU).  Round 2 switched the whole PROCESS to one block per
// tile block whenever a multi-GPU communicator came up, on the guess that RCCL's kernels need CUs; that was a process-global
// side effect of l3_comm_init and was never measured (no N > 1 box), so it is gone: either launch shape keeps every CU
// busy for the length of a convolution (<= 1 ms) and a collective's workgroups start as blocks retire.  L3_WINO_PERSIST=0
// (debug knob) is the A/B switch for an 8-GPU node.
int conv_wino_stat_blocks(const ConvGeom& g);          // partial blocks the Winograd kernel writes (0: not eligible)
int conv_wino_stat_blocks_max(const ConvGeom& g);      // over F(2x2,3x3) / F(4x4,3x3): for sizing the partial scratch
// data gradient of a first-layer conv (Cin in {1,3}, 64 filters, 3x3 'same'); g is the FORWARD
// geometry, w the forward filter.  Returns false (nothing launched) for other shapes.
bool conv_dgrad_small(const float* dy, const float* w, float* dx, const ConvGeom& g, hipStream_t s);
// wt[kh][kw][co][ci] = w[KH-1-kh][KW-1-kw][ci][co]  (dgrad filter)
void conv_flip_weights(const float* w, float* wt, int KH, int KW, int Cin, int Cout, hipStream_t s);
// dw[k][co] = sum_m im2col(x)[m][k] * dy[m][co].  `part` is scratch of
// conv_wgrad_scratch_floats() floats.  g describes the FORWARD conv.
size_t conv_wgrad_scratch_floats(const ConvGeom& g);
// bf16 = true (only honoured when conv_wgrad_bf16_ok(g)): operands rounded to bfloat16, fp32 accumulate.
bool conv_wgrad_bf16_ok(const ConvGeom& g);
// in_bf16 (with bf16): x and dy are bfloat16 tensors (mixed-precision storage).
struct FirstWgFuse;          // (first-layer kernel, below)
void conv_wgrad(const float* x, const float* dy, float* dw, float* part, const ConvGeom& g,
                hipStream_t s, bool bf16 = false, bool in_bf16 = false, const FirstWgFuse* first_fuse = nullptr);

// First convolution of a tower (conv_first.hip): 3x3 'same', Cin in {1, 3}, 64 filters, as an fp32 FMA kernel with
// the BatchNorm statistic partials fused (stat_part: conv_first_stat_blocks(g) blocks of [2][64] about the pivot
// bias[c]; stat_mode as conv_fwd's bn_mode) and an optional bfloat16 output.
bool conv_first_ok(const ConvGeom& g);
// its weight gradient (Cin = 2 or 4: the normalised input channels + the ones channel, engine.hip xaug): one [9 * Cin][64]
// partial per wave into `part` (conv_first_wgrad_scratch_floats(g) floats), returns the number of partials
bool conv_first_wgrad_ok(const ConvGeom& g);
size_t conv_first_wgrad_scratch_floats(const ConvGeom& g);
// fuse (optional): the BatchNorm(+ReLU) that follows the convolution did not write dY (bn_bwd_fast with dx = null); the kernel
// forms it from the convolution's stored output `bnx`, the gradient `dA` behind the BatchNorm (both bfloat16-stored when bf16
// is set; never one of each) and the backward coefficients bn_bwd_fast left (bn_bwd_fast_coeffs): dY = cA mask(dA) + cB y + cC,
// bn_bwd_apply_fast_kernel's expression.  The bias gradient (the column sums of dY) is the ones-channel row of the centre tap.
struct FirstWgFuse {
    const void *bnx, *dA;
    const float *scale, *shift;           // the BatchNorm's forward scale / shift (the ReLU mask is recomputed from them)
    const float *cA, *cB, *cC;
    int relu;                             // 0 none, 1 BN then ReLU (the ReLU -> BN order is not deferred)
    int bf16;
};
int conv_first_wgrad(const float* x, const float* dy, float* part, const ConvGeom& g, hipStream_t s, const FirstWgFuse* fuse = nullptr);
int conv_first_stat_blocks(const ConvGeom& g);
void conv_first_fwd(const float* x, const float* w, const float* bias, void* y, const ConvGeom& g, hipStream_t s,
                    float* stat_part = nullptr, int stat_mode = 0, bool out_bf16 = false);

// fp32 Winograd F(3x3, 2x2) weight gradient (conv_wgrad_wino.hip): `launch` writes conv_wgrad_wino_splits(g, n) partial
// slices of 16 * Cin * Cout floats for n samples, `finish` turns (summed) slices into dw (3, 3, Cin, Cout)
bool conv_wgrad_wino_ok(const ConvGeom& g);
int conv_wgrad_wino_splits(const ConvGeom& g, int n);
int conv_wgrad_wino_max_splits(const ConvGeom& g, int n);      // over the unit shapes a later launch may pick: for sizing scratch
double conv_wgrad_wino_executed_flops(const ConvGeom& g);
void conv_wgrad_wino_launch(const float* x, const float* dy, float* part, const ConvGeom& g, int n, hipStream_t s);
void conv_wgrad_wino_finish(const float* part, float* dw, const ConvGeom& g, int splits, hipStream_t s);
// MFMA flops conv_wgrad() issues for g (-1: the direct count)
double conv_wgrad_executed_flops(const ConvGeom& g, bool bf16);
// bf16-stored weight gradient on gfx950 transpose reads (conv_wgrad_bf16.hip); `splits` split-K slices of `part`
bool conv_wgrad_bf16_tr_enabled();
void conv_wgrad_bf16_tr_launch(const void* x, const void* dy, float* part, const ConvGeom& g, int n, int splits, hipStream_t s);

// Mixed-precision direct convolution (conv_bf16.hip): y = conv(bf16(x), bf16(w)) + bias, fp32 accumulate.
// wn is the filter as [flipped tap][Cout][Cin] (conv_flip_weights(w); for a data gradient: the forward filter).
bool conv_bf16_ok(const ConvGeom& g);
// operands_bf16: x and wn point to bfloat16 data (x written as bf16 by the BatchNorm/pool kernels, wn from
// conv_weights_bf16) instead of fp32 data rounded on the fly: same products, different fp32 summation order.
// stat_part (bf16-operand kernel only): BatchNorm partials of the output (mode 1) / relu(output) (mode 2)
// in conv_bf16_stat_blocks(g) blocks, as conv_fwd's bn_part.
// out_bf16 (bf16-operand kernel only): y is written as bfloat16 (round to nearest even of accumulator + bias),
// and the statistics partials are those of the ROUNDED values -- the tensor the following BatchNorm reads.
void conv_bf16_fwd(const float* x, const float* wn, const float* bias, float* y, const ConvGeom& g, hipStream_t s,
                   bool operands_bf16 = false, float* stat_part = nullptr, int stat_mode = 0, bool out_bf16 = false,
                   const BnBwdFuse* bn_bwd = nullptr);    // halo kernel, bf16 output: as conv_fwd's bn_bwd; bn_bwd->x is bf16-stored
int conv_bf16_stat_blocks(const ConvGeom& g);
// LDS-resident-halo variant (conv_bf16_halo.hip) of the stored-operand kernel.  `n` samples of
// geometry g starting at x / y; one statistics block per 256-pixel patch.
bool conv_bf16_halo_ok(const ConvGeom& g);
int conv_bf16_halo_patches(const ConvGeom& g, int n);
void conv_bf16_halo_launch(const void* x, const void* wn, const float* bias, void* y, const ConvGeom& g, int n,
                           hipStream_t s, float* stat_part, int stat_mode, bool out_bf16, const BnBwdFuse* bn_bwd = nullptr);
// `out` holds TWO bfloat16 copies of the filter (2 * KH * KW * Cin * Cout elements): [tap'][N][K] and, for K % 32 == 0, the
// chunk-major [tap'][K / 32][N][32] (conv_bf16.hip weights_bf16_kernel)
void conv_weights_bf16(const float* w, void* out, int KH, int KW, int Cin, int Cout, bool flip, hipStream_t s);

// first conv of a tower behind a trainable input BatchNorm (elementwise.hip): augmented input
// [xhat, 1] and the closed-form parameter gradients from its weight gradient
void bn_xhat_ones(const float* x, const float* mean, const float* var, float eps, float* xa, int64_t rows, int C,
                  hipStream_t s);
void first_conv_grads(const float* gaug, const float* w, const float* gamma, const float* beta, float* dw, float* dgamma,
                      float* dbeta, float* dbias, int taps, int C, int Co, hipStream_t s);

// column sums / batch-norm
// partial scratch for reductions over `rows` rows of C channels
size_t colreduce_scratch_floats(int64_t rows, int C);
void colsum(const float* x, float* out, float* scratch, int64_t rows, int C, hipStream_t s);
// batch moments (biased var) -> mean,var,scale,shift where y = x*scale + shift
void bn_stats(const float* x, const float* gamma, const float* beta, float* mean, float* var,
              float* scale, float* shift, float* scratch, int64_t rows, int C, float eps,
              hipStream_t s);
// scale/shift from given (moving) statistics
void bn_scale_shift(const float* gamma, const float* beta, const float* mean, const float* var,
                    float* scale, float* shift, int C, float eps, hipStream_t s);
void bn_apply(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
              int relu, hipStream_t s);
// training-mode backward.  dz = relu ? dy*(y>0) : dy.
//  dgamma = sum dz*xhat, dbeta = sum dz, dx = gamma*rstd*(dz - dbeta/n - xhat*dgamma/n)
void bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean,
            const float* var, float* dx, float* dgamma, float* dbeta, float* scratch, int64_t rows,
            int C, float eps, int relu, int training, hipStream_t s);
// moving <- update(moving, batch) ; zero_debias keeps `biased` and uses step
void bn_moving_update(float* moving, float* biased, const float* batch, int C, float momentum,
                      int zero_debias, int step, hipStream_t s);
// the same update (same arithmetic per element) for a table of (moving, biased, batch, C) entries in device memory: one launch
// 16 zero-initialised ints of device memory per (device, stream): the work counters of a persistent grid launched on that stream
// (device_common.h wq_*; the kernel leaves them zero).  nullptr when the table is full or the allocation failed: static stride then.
int* persistent_work_counters(hipStream_t s);

struct BnMovingEntry {
    float* moving;
    float* biased;
    const float* batch;
    int C;
    int off;        // position of this statistic in the packed vector of all of them (the data-parallel exchange)
};
// `step` = moving-average updates applied once this call is done.  gathered == nullptr: one update from entry.batch.  Else the
// `replicas` updates of a data-parallel step, replica r's statistics at gathered[r * stride + entry.off ...], applied in order.
void bn_moving_update_all(const BnMovingEntry* tab_dev, int entries, int max_c, float momentum, int zero_debias, int64_t step,
                          hipStream_t s, const float* gathered = nullptr, int replicas = 1, int64_t stride = 0);
// packed[entry.off + c] = entry.batch[c] for every entry
void bn_moving_pack(const BnMovingEntry* tab_dev, int entries, int max_c, float* packed, hipStream_t s);

// ---- fast path for power-of-two channel counts (bn_fused.hip) ----------------------------
bool bn_fast_ok(int C);
size_t bn_fast_scratch_floats(int C);
// relu placement `mode`: 0 none, 1 BN->ReLU, 2 ReLU->BN (vision_model.py:138-139); `prerelu` = (mode == 2)
void bn_stats_fast(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* scale,
                   float* shift, float* scratch, int64_t rows, int C, float eps, int prerelu, hipStream_t s, int x_bf16 = 0);
void bn_stats_from_partials(const float* part, int nblk, const float* pivot, const float* gamma, const float* beta,
                            float* mean, float* var, float* scale, float* shift, int64_t rows, int C, float eps,
                            int prerelu, hipStream_t s);
// out_bf16 / dx_bf16 (mixed-precision mode): the output tensor is consumed only as a bf16 convolution
// operand and is stored as bfloat16 (same element indexing, half the bytes; bit-identical to rounding later).
// x_bf16 (mixed-precision mode): the input x is a conv output that the bf16 conv kernel stored as bfloat16.
void bn_apply_fast(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C, int relu,
                   hipStream_t s, int out_bf16 = 0, int x_bf16 = 0);
// p = maxpool2x2/2(relu(x*scale+shift)); the full-resolution activation is not stored.
// xwin (nullable; contiguous pooled layout, x's storage type): the input element (mode 2: the rectified one) that WON each window
// (first maximum in row-major order, the rule of the backward kernels) -- with it the BatchNorm's backward reduction needs nothing at full
// resolution and runs in the epilogue of the data gradient that produces the pooled gradient (BnBwdFuse with x = xwin).
void bn_relu_pool2_fwd(const float* x, const float* scale, const float* shift, float* p, int N, int H, int W, int C,
                       int Ho, int Wo, int64_t out_batch_stride, int mode, hipStream_t s, int out_bf16 = 0, int x_bf16 = 0,
                       void* xwin = nullptr);
// backward of BN(+ReLU)(+MaxPool2x2) with the ReLU mask / pool arg-max recomputed from x.
// dy is the gradient at the BN(+ReLU) output (pooled=0) or at the pooled output (pooled=1).
// dbias (nullable) receives the column sums of dx (bias gradient of the preceding conv).
// dy_bf16 (mixed-precision mode): dy was stored as bfloat16 by the mixed-precision data-gradient kernel.
void bn_bwd_fast(const float* x, const float* scale, const float* shift, const float* mean, const float* var,
                 const float* gamma, const float* dy, int pooled, int N, int H, int W, int C, int Ho, int Wo,
                 int64_t dy_batch_stride, float* dx, float* dgamma, float* dbeta, float* dbias, float* scratch,
                 float eps, int relu, int training, hipStream_t s, int dx_bf16 = 0, int x_bf16 = 0, int dy_bf16 = 0,
                 const float* ready_part = nullptr, int ready_blocks = 0);   // reduction partials left by the producer of dy (BnBwdFuse)
// dx = null: reduction and parameter gradients only; the coefficients of dx = cA mask(dy) + cB x + cC stay in `scratch` -- C floats
// each at bn_bwd_fast_coeffs(scratch, C), + C, + 2 C -- until the next BatchNorm kernel that is given the same scratch
// (conv_first_wgrad's FirstWgFuse applies them itself)
const float* bn_bwd_fast_coeffs(const float* scratch, int C);

void relu_fwd(const float* x, float* y, int64_t n, hipStream_t s);
void relu_bwd(const float* y, const float* dy, float* dx, int64_t n, hipStream_t s);

struct PoolGeom {
    int N, H, W, C, Ho, Wo, ph, pw, sh, sw, padT, padL;
    int64_t out_batch_stride;   // elements between samples in y / dy (>= Ho*Wo*C)
};
void maxpool_fwd(const float* x, float* y, const PoolGeom& g, hipStream_t s);
// requires sh>=ph && sw>=pw (non-overlapping windows; all reference pools)
void maxpool_bwd(const float* x, const float* dy, float* dx, const PoolGeom& g, hipStream_t s);

// preprocessing (train.py:186,189)
void preprocess_video(const uint8_t* u8, float* out, int64_t n, hipStream_t s);
void preprocess_audio(const int16_t* pcm, float* out, int64_t n, hipStream_t s);
void labels_onehot(const int32_t* lab, float* out, int64_t n, hipStream_t s);

// audio front-end (kapre Spectrogram / Melspectrogram)
struct FrontendCfg {
    int n_dft, n_hop, pad_left, n_frames, n_freq, n_mels;   // n_freq = n_dft/2+1
    int sqrt_out;       // power != 2.0 -> sqrt
    int db;             // amplitude_to_decibel
    int loglambda;      // log(max(x,1e-12))/5
    int ncols_pad;      // padded column count of the DFT matrix (re | im | zero pad)
    // folded DFT (symmetric kernels: real[n] == real[N-n], imag[n] == -imag[N-n], true for kapre's Hann-windowed
    // cos / sin kernels): re = [x0, x1+x_{N-1}, ..., x_{N/2}] . R[0..N/2], im = [x1-x_{N-1}, ...] . I[1..N/2-1]
    // -- two GEMMs of half the depth.  ke / ko = padded widths of the two folded frame matrices,
    // nc = ncols_pad / 2 = padded column count of each half-spectrum.
    int folded, ke, ko, nc;
    // factored DFT (round 6): when the kernels are kapre's STOCK ones (periodic Hann window x DFT basis, checked bit for bit against
    // host_dft_kernels), n_dft = N1 * N2 = 32 * 64 and the transform runs as two small GEMMs around a twiddle pass (Cooley-Tukey:
    // n = N2 n1 + n2, k = k1 + N1 k2) -- 5x fewer multiplies than the folded full-depth GEMMs.  frontend.hip dft_*.
    int factored, N1, N2;
};
// xw[(frame * N2 + n2) * N1 + n1] = win[N2 n1 + n2] * audio[frame's sample N2 n1 + n2]: rows of the first GEMM (K = N1)
void dft_pack_frames(const float* audio, const float* win, float* a1, int B, int T, const FrontendCfg& c, hipStream_t s);
// y[(frame * N2 + n2)][re k1 (N1) | im k1 (N1)] -> a2[(frame * N1 + k1)][re n2 (N2) | im n2 (N2)] = y * exp(-2 pi i n2 k1 / n_dft)
// (tw: [N2][N1] pairs (cos, sin)); nyq[frame] = the Nyquist bin, sum_n (-1)^n xw[n]
void dft_twiddle(const float* y, const float* tw, float* a2, float* nyq, int frames, const FrontendCfg& c, hipStream_t s);
void frame_audio_folded(const float* audio, float* fe, float* fo, int B, int T, const FrontendCfg& c, hipStream_t s);
void frame_audio(const float* audio, float* frames, int B, int T, const FrontendCfg& c, hipStream_t s);
// spec (B*n_frames, ncols_pad) -> out (B, F, n_frames) with F = n_mels or n_freq
// ... and all of it, mel projection and sqrt / dB included, in one kernel (n_dft = 2048 = 32 x 64, n_mels > 0): audio -> out (B, mels, frames)
void dft_fused(const float* audio, const float* win, const float* b1, const float* b2, const float* tw, const float* melw,
               const int* mel_start, const int* mel_len, const int* mel_off, float* out, int B, int T, const FrontendCfg& c,
               hipStream_t s);
// (factored: spec = x2[(frame * N1 + k1)][re k2 (N2 / 2) | im k2 (N2 / 2)], bin k = k1 + N1 k2, and nyq[frame])
void spec_to_features(const float* spec, const float* melw, const int* mel_start, const int* mel_len,
                      const int* mel_off, float* out, int B, const FrontendCfg& c, hipStream_t s, const float* nyq = nullptr);
void db_normalize(float* x, float* smax, int B, int64_t per_sample, int batch_scope, hipStream_t s);

// head: dense + softmax + categorical cross-entropy
void dense_fwd(const float* x, const float* w, const float* b, float* y, int B, int K, int N, int relu,
               hipStream_t s);
void dense_bwd_w(const float* x, const float* dy, float* dw, float* db, int B, int K, int N, hipStream_t s);
void dense_bwd_x(const float* dy, const float* w, float* dx, int B, int K, int N, hipStream_t s);
// probs, per-batch loss/acc sums -> stats[0]=sum loss_i, stats[1]=#correct ; dlogits scaled by gscale
void softmax_ce(const float* logits, const float* labels, float* probs, float* dlogits, float* stats,
                int B, float gscale, hipStream_t s);
void sumsq(const float* x, int64_t n, float* out /*1 float*/, float* scratch, hipStream_t s);
size_t sumsq_scratch_floats(int64_t n);
// the sums of squares of up to SUMSQ_MAX_SEGS ranges [off, off + n) of `base` in two launches: out[i] = sum of range i;
// scratch: SUMSQ_MAX_SEGS * SUMSQ_BLOCKS floats
constexpr int SUMSQ_MAX_SEGS = 24, SUMSQ_BLOCKS = 64;
struct SumsqSegs { int count; int64_t off[SUMSQ_MAX_SEGS], n[SUMSQ_MAX_SEGS]; };
void sumsq_multi(const float* base, const SumsqSegs& segs, float* out, float* scratch, hipStream_t s);

void adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_l2, float l2x2,
               float lr_t, float b1, float b2, float eps, float gscale, hipStream_t s);
void fill(float* p, float v, int64_t n, hipStream_t s);
// fp32 -> bfloat16 storage, round to nearest even (op-level entry points: emulates a mixed-precision writer)
void cast_bf16(const float* x, void* y, int64_t n, hipStream_t s);

}  // namespace l3
