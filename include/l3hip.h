/* l3hip.h -- C ABI of libl3hip.so: the MI355X-native L3-Net AVC training path.
 *
 * The reference (marl/l3embedding) has no FFI seam; its boundary is the Python
 * object protocol between l3embedding/train.py + model.py and Keras (SURVEY.md
 * section 8b).  Each entry point below names the reference call it stands in for
 * (paths relative to the reference tree).  The Python mirror of the reference's
 * interface (l3embedding_amd/model.py, train.py) binds these with ctypes; the
 * stub a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain C, host pointers unless the name ends in _dev, float32 data,
 * NHWC activations, HWIO conv kernels, (in,out) dense kernels.  Every function
 * returns 0 on success and a negative L3_E* code on failure; l3_last_error() gives
 * the message.  One engine per process/GPU; calls on one engine are not
 * re-entrant.  Inputs/outputs stay caller-owned; weights are copied in/out.
 */
#ifndef L3HIP_H
#define L3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L3_OK 0
#define L3_EINVAL (-1)   /* bad argument (mirrors ValueError: model.py:113-114,175-176) */
#define L3_EHIP (-2)     /* HIP runtime error */
#define L3_ENOMEM (-3)
#define L3_ESTATE (-4)   /* call out of order */
#define L3_ECOMM (-5)    /* RCCL error (or librccl could not be loaded) */

/* MODELS registry keys, model.py:307-313 */
#define L3_MODEL_CNN_L3_ORIG 0
#define L3_MODEL_TINY_L3 1
#define L3_MODEL_CNN_L3_KAPREDBINPUTBN 2
#define L3_MODEL_CNN_L3_MELSPEC1 3
#define L3_MODEL_CNN_L3_MELSPEC2 4

typedef struct l3_engine l3_engine;

#define L3_DTYPE_F32 0
#define L3_DTYPE_BF16 1

/* fp32 convolution algorithm of the 14 3x3 'same' layers (forward and data gradient).  Both are plain fp32 arithmetic on
 * the fp32 matrix cores; they differ in rounding error and speed:
 *   L3_FP32_CONV_F4X4 (default)  Winograd F(4x4,3x3): layer outputs within ~8e-6 of the output range of a direct fp32
 *                                convolution (float64 oracle; tests/test_layer_parity_gpu.py bound 3e-5); fastest
 *   L3_FP32_CONV_F2X2            Winograd F(2x2,3x3): within ~1e-6 (bound 3e-6), the step ~17 % slower -- for a caller
 *                                that wants the tightest parity with the reference's direct convolution */
#define L3_FP32_CONV_F4X4 0
#define L3_FP32_CONV_F2X2 1
/* Value 2 (L3_FP32_CONV_F2X2_BF16X6: F(2x2,3x3) on exact bfloat16 triples, bf16 matrix cores) is a measured experiment that is
 * slower than the default at the accuracy class of L3_FP32_CONV_F2X2 (profiles/r05_bx6_ablations.txt).  It is not part of the
 * product library: l3_create accepts it only from a library built with L3_BUILD_EXPERIMENTS=1 (l3_build_experiments() == 1). */
#define L3_FP32_CONV_F2X2_BF16X6 2

/* BatchNormalization moving statistics of a data-parallel step (l3_step_dp, world > 1).  The reference's multi_gpu_model CALLS
 * the one template model once per replica (training_utils.py:155 `outputs = model(inputs)`), so every BatchNormalization
 * (vision_model.py:124-187, audio_model.py:370-433) adds one moving-average update PER REPLICA to ONE shared variable.
 *   L3_DP_MOVING_REPLICAS (default)  every rank gathers all ranks' batch means / variances (one small all-gather per step on
 *                                    the communicator stream) and applies the `world` updates in replica order 0..world-1 to
 *                                    its moving variables: identical on every rank, momentum 0.99^world per step, every
 *                                    shard's statistics in the checkpoint whichever rank writes it
 *   L3_DP_MOVING_RANK_LOCAL          one update per step from the rank's own shard (rounds 1-5); ranks then validate with
 *                                    different statistics */
#define L3_DP_MOVING_REPLICAS 0
#define L3_DP_MOVING_RANK_LOCAL 1

typedef struct l3_config {
    int32_t struct_size;     /* sizeof(l3_config) */
    int32_t model_type;      /* L3_MODEL_* */
    int32_t batch;           /* per-device batch (fixed for the engine's life) */
    int32_t global_batch;    /* batch over all ranks (0 => batch); loss gradient is
                                scaled 1/global_batch so a SUM all-reduce gives the
                                gradient of the mean loss over the concatenated batch
                                (training_utils.py:165-170 + keras mean loss) */
    int32_t device;          /* HIP device ordinal */
    int32_t db_max_scope;    /* 0: per-sample max (kapre 0.1.4), 1: batch max (0.1.3.1) */
    int32_t bn_zero_debias;  /* 1: keras-2.0.9/TF-1.4 assign_moving_average(zero_debias=True) */
    int32_t dtype;           /* L3_DTYPE_F32 (0): fp32 everywhere (reference: Input(dtype='float32'),
                                audio_model.py:363, vision_model.py:123).  L3_DTYPE_BF16 (1): mixed precision
                                of BASELINE configs[4] -- the 14 3x3 'same' convolutions with Cin, Cout % 64 == 0
                                round both operands to bfloat16 and accumulate in fp32 (forward, data gradient,
                                weight gradient); every tower convolution that feeds a BatchNormalization STORES
                                its output as bfloat16 (and the data gradient it hands to the preceding BatchNorm);
                                the BatchNorm arithmetic on those tensors, weights, BN parameters and statistics,
                                the two embedding-layer outputs, the head, the loss and Adam stay fp32
                                (oracle.mixed_precision('bf16') restates the three rules) */
    void *stream;            /* hipStream_t to launch on, NULL => engine-owned stream */
    int32_t fp32_conv;       /* L3_FP32_CONV_*: see above (ignored by the bf16 engine's 14 layers) */
    int32_t dp_moving;       /* L3_DP_MOVING_*: see above (was reserved0; 0 keeps the struct compatible) */
} l3_config;

/* MODELS[model_type](num_gpus=...) -- model.py:184-195,307-313; train.py:267.
 * Weights are initialised like the reference (he_normal kernels, zero biases, BN
 * gamma=1 beta=0 mean=0 var=1; kapre DFT / mel constants) from `seed`.
 * (Host waits of the library -- l3_sync, the result readers -- sleep between hipStreamQuery calls instead of spinning in
 * hipStreamSynchronize: one host core per rank less; L3_HOST_WAIT=spin restores the spin.) */
int l3_create(const l3_config *cfg, uint64_t seed, l3_engine **out);
void l3_destroy(l3_engine *e);
const char *l3_last_error(const l3_engine *e);   /* e may be NULL (create errors) */
int l3_model_type_from_name(const char *name);   /* <0 if not in MODELS (model.py:113-114) */
int l3_build_experiments(void);                  /* 1: built with L3_BUILD_EXPERIMENTS=1 (measured-and-rejected kernel variants compiled in) */
/* AMD GPUs visible to this process (0 without one) -- _get_available_devices(), training_utils.py:12-18,
 * behind multi_gpu_model's "we expect the following devices to be available" check (:107-119). */
int l3_device_count(void);

/* model.get_weights()/set_weights()/load_weights() -- model.py:77,119.
 * Parameters are enumerated in keras get_weights() order (layer by layer; kapre
 * tensors first in the audio tower); names look like
 * "vision_model/conv2d_1/kernel", "audio_model/batch_normalization_10/moving_mean". */
int l3_param_count(const l3_engine *e);
int l3_param_info(const l3_engine *e, int index, char *name, int name_cap,
                  int32_t *ndim, int64_t shape[4], int32_t *trainable, int64_t *numel);
int l3_set_param(l3_engine *e, const char *name, const float *src, int64_t numel);
int l3_get_param(l3_engine *e, const char *name, float *dst, int64_t numel);
int l3_get_grad(l3_engine *e, const char *name, float *dst, int64_t numel);
/* Adam moments + iteration counter reset (keras save_weights does not store the
 * optimizer: a resumed run restarts them -- train.py:263-265,316-355). */
int l3_reset_optimizer(l3_engine *e);
/* A Keras model keeps ONE set of variables whatever batch size is fed (fit_generator alternates
 * train_batch_size / validation_batch_size steps on the same model, train.py:408-414).  An engine's
 * activation buffers are sized for one batch, so the host keeps one engine per fed batch size and moves
 * the model state between them, device to device: every parameter (trainable and not), the Adam
 * moments and step count, and the BatchNorm zero-debias accumulators + update count. */
int l3_copy_state(l3_engine *dst, l3_engine *src);
int l3_optimizer_steps(const l3_engine *e, int64_t *adam_t, int64_t *bn_steps);

/* model.predict / test-mode forward (BN moving statistics) or training-mode
 * forward (batch statistics).  video (B,224,224,3) in [-1,1], audio (B,1,48000);
 * probs/logits (B,2), either may be NULL.  train.py:382-384 feed order. */
int l3_forward(l3_engine *e, const float *video, const float *audio, int training,
               float *probs, float *logits);

/* One fit_generator step -- train.py:282-284,408-414: forward(training) ->
 * categorical_crossentropy + L2 -> backward -> Adam(lr) -> BN moving update.
 * labels (B,2) one-hot.  loss includes the L2 penalty like keras' logged loss. */
int l3_train_step(l3_engine *e, const float *video, const float *audio,
                  const float *labels, float lr, float *loss, float *acc);
/* test_on_batch (validation, train.py:408-414 validation_data): inference-mode BN. */
int l3_eval_step(l3_engine *e, const float *video, const float *audio,
                 const float *labels, float *loss, float *acc);

/* Device-resident input path (the measured one: inputs already in HBM).
 * l3_upload_batch copies caller host buffers into the engine's input tensors;
 * l3_upload_batch_raw takes the HDF5 blob dtypes (uint8 frames, int16 PCM, int
 * labels; data/avc/sample.py:371-377) and applies train.py:186,189 on the GPU. */
int l3_upload_batch(l3_engine *e, const float *video, const float *audio, const float *labels);
int l3_upload_batch_raw(l3_engine *e, const uint8_t *video_u8, const int16_t *audio_i16,
                        const int32_t *labels_i32);
/* Pipelined variant for a training loop: copies the NEXT batch (stored dtypes) to the device over the
 * engine's own copy stream and returns; the copy overlaps the step that is still running, and the
 * following l3_step_forward() adopts the staged batch (scaling kernels in stream order) before it
 * starts.  Keras' fit_generator keeps batches queued ahead of the device the same way
 * (train.py:408-414, max_queue_size=10). */
int l3_stage_batch_raw(l3_engine *e, const uint8_t *video_u8, const int16_t *audio_i16, const int32_t *labels_i32);
/* Staged step on the resident batch, so the host can overlap the gradient
 * all-reduce with backward (buckets complete head -> block4 -> ... -> block1):
 *   l3_step_forward          forward + loss + head backward          (bucket 0 ready)
 *   l3_step_backward_bucket  backward of tower block k = 1..n-1      (bucket k ready)
 *   l3_step_update           Adam on (all-reduced) grads * grad_scale + BN moving update */
int l3_step_forward(l3_engine *e, int training);
/* One sub-network alone on the resident batch (SURVEY 8(d) config "audio tower only"): forward in
 * training mode (batch-norm batch statistics; tower 1 includes the kapre front-end of
 * audio_model.py:367-369), and with backward != 0 the backward pass from the stand-in loss
 * mean(tower output).  tower: 0 = vision_model, 1 = audio_model.  No optimizer step.  Asynchronous:
 * l3_sync() to wait. */
int l3_tower_step(l3_engine *e, int tower, int backward);
int l3_step_bucket_count(const l3_engine *e);
int l3_step_backward_bucket(l3_engine *e, int bucket);
int l3_step_update(l3_engine *e, float lr, float grad_scale);
int l3_step_resident(l3_engine *e, float lr);   /* all of the above, world size 1 */
int l3_step_results(l3_engine *e, float *loss, float *acc, float *probs, float *logits); /* syncs */
/* The same loss / accuracy without stalling the pipeline: _enqueue copies the sums of the step just enqueued to pinned slot 0 or 1
 * behind it (call it right after l3_step_resident), _wait waits for that copy only -- the caller may enqueue the next step in
 * between, which is how fit_generator (train.py:408-414) reads every step's loss while the GPU never waits for the host. */
int l3_step_results_enqueue(l3_engine *e, int slot, int reduce);   /* reduce = 1 (after l3_comm_init): the sums are added up over the
                                                                       ranks first -- loss / acc of the concatenated batch,
                                                                       training_utils.py:165-170; every rank must call it */
int l3_step_results_wait(l3_engine *e, int slot, float *loss, float *acc);

/* Data parallelism -- multi_gpu_model, training_utils.py:21-170, reached through gpu_wrapper
 * (model.py:184-195).  One process per GPU; every rank owns an engine created with batch = its shard
 * (training_utils.py:121-133) and global_batch = the batch over all ranks.  The reference's implicit
 * gradient AddN over replicas (training_utils.py:141-170) becomes an RCCL SUM all-reduce of the flat
 * gradient arena, issued by the library itself:
 *   l3_comm_unique_id   rank 0 obtains the 128-byte ncclUniqueId and ships it to the other ranks by any
 *                       means the host has (file, TCP store, MPI); no engine needed
 *   l3_comm_init        ncclCommInitRank for this engine's GPU (collective over all ranks)
 *   l3_step_dp          one training step on the resident batch: as each gradient bucket completes
 *                       (head, vision block 4..1, audio block 4..1) its ncclAllReduce is enqueued on the
 *                       communicator's own HIP stream behind an event, while backward continues on the
 *                       engine's streams; Adam waits for the last bucket.  No host code in the overlap path.
 *   l3_comm_allreduce_host  sum (op 0) / max (op 1) of up to 64 host doubles over the ranks (logged
 *                       loss/accuracy over the concatenated batch, timing, and -- it synchronises -- a barrier)
 * librccl is bound at the first call (dlopen), see csrc/comm.hip. */
#define L3_COMM_ID_BYTES 128
int l3_comm_unique_id(void *id128);
int l3_comm_init(l3_engine *e, const void *id128, int world, int rank);
int l3_comm_destroy(l3_engine *e);
int l3_comm_info(const l3_engine *e, int *world, int *rank, char *library_path, int path_cap);
int l3_comm_version(void);   /* ncclGetVersion() code of the bound librccl (0 before the first l3_comm_* call) */
int l3_comm_allreduce_host(l3_engine *e, double *vals, int n, int op);
int l3_step_dp(l3_engine *e, float lr);
/* Measurement mode of l3_step_dp (never on in a timed region: every step is waited for): hipEvents around each bucket's
 * ncclAllReduce on the communicator's stream, at "backward done" on the engine's stream and behind the last collective.
 * _read returns per-step averages since the last l3_comm_timing(e, 1):
 *   exposed_ms  how long the optimizer step had to wait for the wire after backward was done (0 when the collectives
 *               are hidden behind backward -- what the overlap of training_utils.py:141-170's gradient sum is for)
 *   span_ms     first collective started -> last collective done
 *   bucket_ms   duration of each bucket's all-reduce (l3_step_bucket_count() entries: head, vision 4..1, audio 4..1) */
int l3_comm_timing(l3_engine *e, int on);
int l3_comm_timing_read(l3_engine *e, double *exposed_ms, double *span_ms, double *bucket_ms, int cap, int *steps);

/* Flat fp32 gradient arena (device) and its buckets, for RCCL all-reduce
 * (replaces the implicit gradient AddN of training_utils.py:141-170). */
int l3_grad_arena_dev(l3_engine *e, void **dev_ptr, int64_t *numel);
int l3_bucket_range(const l3_engine *e, int bucket, int64_t *offset, int64_t *numel);
/* BatchNormalization moving statistics under data parallelism for a caller that runs its own collectives (l3_step_dp does this
 * itself; l3_config.dp_moving, training_utils.py:141-157): after l3_step_forward(e, 1), l3_bn_stats_pack_dev packs this rank's
 * batch means / variances (`numel` floats, engine stream) -> the caller all-gathers them into the buffer
 * l3_bn_stats_replicas_dev returns (world x numel floats, rank-major) -> the next l3_step_update applies the `world` replica
 * updates in rank order instead of the rank's own single one. */
int l3_bn_stats_pack_dev(l3_engine *e, void **send_dev, int64_t *numel);
int l3_bn_stats_replicas_dev(l3_engine *e, int world, void **gathered_dev);

/* load_embedding(...).predict -- model.py:131-181; audio_model.py:445-487;
 * vision_model.py:198-218: MaxPooling2D(pool, padding='same') on the conv output
 * of '<audio|vision>_embedding_layer' (before its BN/ReLU), inference-mode BN,
 * Flatten.  n may exceed the engine batch (processed in chunks).  out (n, D). */
int l3_embed_audio(l3_engine *e, const float *audio, int64_t n, int pool_h, int pool_w, float *out);
int l3_embed_vision(l3_engine *e, const float *video, int64_t n, int pool_h, int pool_w, float *out);
int64_t l3_embed_dim(const l3_engine *e, int vision, int pool_h, int pool_w);

/* Diagnostics / parity taps. */
int l3_get_activation(l3_engine *e, const char *name, float *dst, int64_t numel); /* e.g. "audio_model/frontend", "vision_model/conv2d_1" */
int l3_activation_numel(l3_engine *e, const char *name, int64_t *numel);
int l3_sync(l3_engine *e);
/* Per-kernel-family device time (ms) accumulated with hipEvents on the engine's
 * stream while profiling is enabled; families: 0 conv_fwd, 1 conv_dgrad, 2 conv_wgrad,
 * 3 elementwise/bn/pool, 4 frontend, 5 head+loss, 6 adam. */
int l3_profile_enable(l3_engine *e, int on);
/* The engine runs the audio tower (front-end included) on an internal second stream beside
 * the vision tower, forked from and joined back into the engine's stream by events, so one
 * tower's HBM-bound BatchNorm/pool kernels overlap the other's MFMA-bound convolutions
 * (the two sub-networks are independent until model.py:218's concatenate).  on=0 serialises
 * both towers on the engine's stream -- per-kernel durations are only meaningful that way.
 * Results are identical either way.  Default on (environment L3_TWO_STREAMS=0 disables). */
int l3_set_tower_overlap(l3_engine *e, int on);
int l3_profile_read(l3_engine *e, int family, double *ms, int64_t *launches, double *flops);
/* `flops` above are ALGORITHMIC (direct convolution, 2*M*K*N).  This returns the flops the
 * family's kernels actually issued: the 3x3 forward / data-gradient launches run Winograd
 * F(4x4,3x3) (36 multiplies per 4x4 output tile and channel pair instead of 144: 9/36 of direct, plus tile
 * padding; F(2x2,3x3) with L3_FP32_CONV_F2X2: 16/36), the weight gradient F(3x3,2x2) (16/36). */
int l3_profile_read_executed(l3_engine *e, int family, double *flops);
/* ... and its ALGORITHMIC HBM bytes: the sum, over the family's launches, of every tensor a launch must read once and write
 * once (convolutions: input + output + filter; BatchNorm / pool kernels: the tensors of each pass) -- what the launch durations
 * and the rocprofv3 FETCH_SIZE / WRITE_SIZE counters are compared with (bench.py hbm_tbs, traffic_over_algorithmic). */
int l3_profile_read_bytes(l3_engine *e, int family, double *bytes);

/* Stand-alone operator entry points (host buffers) used by the op-level parity
 * tests; each replaces the TF op a Keras/kapre layer instantiates. */
/* dtype variants of the two conv operators: same arguments plus L3_DTYPE_*; BF16 falls back to the
 * fp32 kernels for geometries the mixed-precision kernels do not take (first layers).
 * L3_OP_BF16_STORED (these two entry points only): the operands are first written to HBM as bfloat16
 * and the stored-operand kernels run -- the path an L3_DTYPE_BF16 engine takes for its mixed-precision
 * layers (same products as L3_DTYPE_BF16, different fp32 summation order). */
#define L3_OP_BF16_STORED 2
/* L3_OP_BF16_STORED_OUT: as above, and the output tensor is stored as bfloat16 too -- y of l3_op_conv2d_fwd_dt (what
 * the engine does for a tower conv that feeds a BatchNormalization) and dx of l3_op_conv2d_bwd_dt (the data gradient
 * a mixed-precision conv hands to the preceding BatchNorm's backward); the stored values are returned widened. */
#define L3_OP_BF16_STORED_OUT 3
int l3_op_conv2d_fwd_dt(int device, int dtype, const float *x, const float *w, const float *b, float *y,
                        int n, int h, int wd, int cin, int cout, int kh, int kw, int same);
int l3_op_conv2d_bwd_dt(int device, int dtype, const float *x, const float *w, const float *dy, float *dx,
                        float *dw, float *db, int n, int h, int wd, int cin, int cout, int kh, int kw, int same);
int l3_op_conv2d_fwd(int device, const float *x, const float *w, const float *b, float *y,
                     int n, int h, int wd, int cin, int cout, int kh, int kw, int same);
int l3_op_conv2d_bwd(int device, const float *x, const float *w, const float *dy,
                     float *dx, float *dw, float *db,
                     int n, int h, int wd, int cin, int cout, int kh, int kw, int same);
/* x_bf16 (these four BatchNorm operators; c a power of two >= 4), a bit mask: bit 0 -- x is first written to HBM as
 * bfloat16 and the kernels that widen it on load run (how an L3_DTYPE_BF16 engine reads the output of a
 * mixed-precision conv); bit 1 (the two backward operators) -- the incoming gradient dy / dp likewise (how it
 * reads the data gradient a mixed-precision conv stored). */
int l3_op_bn_relu_fwd(int device, const float *x, const float *gamma, const float *beta,
                      float *y, float *mean, float *var, int64_t rows, int c, int relu, int x_bf16);
/* beta non-NULL and c a power of two >= 4: the engine's fast kernels (ReLU mask recomputed from
 * x*scale+shift, y unused); beta NULL: the generic kernels (mask from y). */
int l3_op_bn_relu_bwd(int device, const float *x, const float *y, const float *dy,
                      const float *gamma, const float *beta, const float *mean, const float *var,
                      float *dx, float *dgamma, float *dbeta, int64_t rows, int c, int relu, int x_bf16);
/* Conv-BN-ReLU-MaxPool2D((2,2), strides=2) tail as the engine fuses it (c must be a power of
 * two >= 4), batch statistics; backward from the pooled gradient.  relu_mode 1: p = pool(relu(bn(x)))
 * (vision_model.py:130-134 and every other block); relu_mode 2: p = pool(bn(relu(x))), the
 * Activation-before-BatchNormalization order of vision_model.py:137-139. */
int l3_op_bn_relu_pool2_fwd(int device, const float *x, const float *gamma, const float *beta, float *p,
                            float *mean, float *var, int n, int h, int wd, int c, int same, int relu_mode, int x_bf16);
int l3_op_bn_relu_pool2_bwd(int device, const float *x, const float *gamma, const float *beta, const float *dp,
                            float *dx, float *dgamma, float *dbeta, float *dbias, int n, int h, int wd, int c,
                            int same, int relu_mode, int x_bf16);
int l3_op_maxpool_fwd(int device, const float *x, float *y, int n, int h, int wd, int c,
                      int ph, int pw, int sh, int sw, int same);
int l3_op_maxpool_bwd(int device, const float *x, const float *dy, float *dx, int n, int h,
                      int wd, int c, int ph, int pw, int sh, int sw, int same);
int l3_op_frontend(int device, int model_type, const float *audio, int n, int db_max_scope, float *out);
/* keras BatchNormalization batch moments (vision_model.py:124-187, audio_model.py:370-433) from the partial sums the
 * convolution epilogues leave: nblk rows of [sum, sum of squares][c] about pivot[c] over `rows` elements per channel.
 * Runs the engine's second reduction stage on a buffer of exactly nblk rows; L3_EINVAL if it wrote past them. */
int l3_op_bn_stats_from_partials(int device, const float *part, int nblk, int c, const float *pivot, int64_t rows,
                                 float eps, float *mean, float *var);
int l3_op_preprocess(int device, const uint8_t *video_u8, int64_t nv, float *video,
                     const int16_t *audio_i16, int64_t na, float *audio);

#ifdef __cplusplus
}
#endif
#endif /* L3HIP_H */
