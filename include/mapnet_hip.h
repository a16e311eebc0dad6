/*
 * mapnet_hip.h -- C ABI of libmapnet_hip.so, the gfx950 (MI355X) implementation of the MapNet
 * training hot path of NVlabs/geomapnet.
 *
 * The reference has no FFI for this path (it is plain PyTorch 0.4.1 Python); the
 * boundary a drop-in replaces is the Python surface that scripts/train.py, scripts/eval.py and
 * common/train.py consume.  Each entry point below names the reference interface it stands
 * behind (paths relative to /root/reference).  The Python mirror of that surface lives in
 * geomapnet_amd/ and binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE pointers
 *     (HBM) owned by the caller; the library never allocates device memory: the caller hands
 *     it arenas sized by the mn_model_* queries and mn_plan_bytes.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); every call only
 *     enqueues work on that stream and returns (asynchronous, like the reference until its
 *     loss.item() at common/train.py:361).
 *   - return value: 0 on success, non-zero on error; mn_last_error() gives the message.
 *     No exceptions cross the ABI.
 *   - one mn_handle per device per (mode, batch, H, W, dtype); not thread-safe per handle.
 */
#ifndef MAPNET_HIP_H
#define MAPNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MN_DTYPE_F32:   fp32 tensors, contractions on v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain; 157 TF peak).
 * MN_DTYPE_F16:   fp16 activations / gradients / conv operands, fp32 accumulate, fp32 master weights (the benchmark mode).
 * MN_DTYPE_F32X3: fp32 tensors exactly as MN_DTYPE_F32, but every convolution contracts on the f16 / bf16 matrix pipe:
 *                 each fp32 operand is split in registers into hi + lo halves and a product costs three
 *                 v_mfma_f32_32x32x16_{f16,bf16} (fp16 halves in the forward pass, bf16 halves -- fp32's exponent range --
 *                 in the backward pass).  fp32-class results (the reference computes in fp32, common/train.py:322-363)
 *                 at several times the fp32 pipe's rate: the mode the parity bar is met in.  Operator entry points accept
 *                 it where their dtype argument selects the arithmetic (mn_op_igemm, mn_op_wgrad, mn_op_conv_dgrad: 2 =
 *                 f16x3 for igemm, bf16x3 for wgrad / dgrad; tensors are fp32).
 * MN_DTYPE_F16X2: (round 4) fp32-class values stored as fp16 PAIRS ("h2": hi + lo halves, 4 bytes per element like fp32, x to
 *                 2^-22; layout in geomapnet_amd/csrc/common.h: per row and 32-channel group 64 bytes of hi halves, then 64
 *                 bytes of lo halves).  Every tensor a convolution CONSUMES (activations, weight copies, d(conv output)) is
 *                 h2 and is split once, by the kernel that produces it; conv outputs, data gradients and everything the
 *                 BatchNorm / pooling / head / optimiser kernels compute on stay fp32.  The convolutions then run three
 *                 v_mfma_f32_32x32x16_f16 per product on operands that reach LDS by DMA, with no conversion in their K loops.
 *                 Gradients are kept inside fp16's range by the loss scale + overflow guard of MN_DTYPE_F16.  Operator entry
 *                 points: dtype 3 = A / Bw / dY / X / gates h2, out / res / dW fp32.
 * MN_DTYPE_F16X2M: (round 5, plans only) the forward pass of MN_DTYPE_F16X2 bit for bit -- the loss and the predicted poses of a
 *                 step ARE that mode's, i.e. inside the north-star tolerance -- and a backward pass on the MN_DTYPE_F16 kernels:
 *                 one MFMA per product on single fp16 operands (d(conv output), data gradients and the data-gradient weight copy
 *                 are plain fp16; the weight gradients' X operand and the ReLU gates read plain fp16 copies of the activations
 *                 that the h2 producers write beside the h2 tensor -- round 6: the hi halves of the h2 tensors in place).  Every
 *                 ReLU gate is the exact outcome of the forward pass and BatchNorm's backward sums are taken in fp64 over the forward
 *                 pass's xhat rounded once to fp16 (round 6: a 2-byte record written by the forward apply, csrc/elementwise_h2.h
 *                 rec_pack; round 5 re-read the fp32 conv output; the stem reads an fp16 copy of its conv output unless
 *                 MN_DETERMINISTIC / MN_STEM_BWD=0), so the gradients differ from MN_DTYPE_F16X2's by operand rounding
 *                 only: 1.1e-3 relative L2 overall (tools/mixed_budget.py), below the 4.9e-3 by which two fp32 evaluations of
 *                 the reference's step differ through ReLU gate flips.  Loss scale + overflow guard as MN_DTYPE_F16.
 * MN_DTYPE_F16X2Q: (round 5, plans only; experimental) MN_DTYPE_F16X2M whose forward convolutions take BOTH cross terms of a
 *                 split-operand product from fp8 (e4m3) copies with fixed exponents on gfx950's block-scaled MFMA
 *                 (v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate): conv-consumed tensors are "h2q" -- fp16 hi halves where
 *                 h2 has them, then an fp8 lo plane and an fp8 copy of the hi plane (geomapnet_amd/csrc/common.h) -- still 4 bytes
 *                 per element.  2 instead of 3 MFMA-equivalents per forward product; poses ~3e-4 from the fp32 reference (bar
 *                 1e-3) instead of 1.6e-5 (tools/fp8_cross_budget.py). */
enum { MN_DTYPE_F32 = 0, MN_DTYPE_F16 = 1, MN_DTYPE_F32X3 = 2, MN_DTYPE_F16X2 = 3, MN_DTYPE_F16X2M = 4, MN_DTYPE_F16X2Q = 5 };
/* criterion / batch-layout modes */
enum {
  MN_MODE_POSENET = 0,      /* PoseNetCriterion,        common/criterion.py:33-52   */
  MN_MODE_MAPNET = 1,       /* MapNetCriterion,         common/criterion.py:54-109  */
  MN_MODE_MAPNET_ONLINE = 2,/* MapNetOnlineCriterion,   common/criterion.py:111-184 */
  MN_MODE_MAPNET_GPS = 3    /* ... with gps_mode=True,  common/criterion.py:166,173-180 */
};

const char* mn_last_error(void);
/* "hip" for the product library; the test-only SIMT-emulator build reports "emu". */
const char* mn_backend(void);

/* ------------------------------------------------------------------------------------------
 * Model parameter layout.  Replaces: torchvision resnet34 + PoseNet.__init__ parameter
 * registration (models/posenet.py:37-63) and the state_dict key set.
 * Parameters live in ONE flat fp32 arena (conv weights stored OHWI, everything else in the
 * reference's own layout) followed by the four criterion scalars sax,saq,srx,srq; buffers
 * (BatchNorm running stats, num_batches_tracked as int64) live in a second arena.
 * ------------------------------------------------------------------------------------------ */
typedef struct mn_entry {
  char name[96];      /* state_dict key, e.g. "feature_extractor.layer1.0.conv1.weight" */
  int64_t offset;     /* element offset (fp32 elements; bytes/8 for int64 buffers) in its arena */
  int64_t numel;
  int32_t ndim;
  int32_t shape[4];   /* reference (torch) shape, e.g. OIHW for conv weights */
  int32_t is_buffer;  /* 0: parameter arena, 1: buffer arena */
  int32_t is_int64;   /* num_batches_tracked */
  int32_t ohwi;       /* 1: stored OHWI; the torch view is arena.view(O,H,W,I).permute(0,3,1,2) */
  int32_t stage;      /* data-parallel gradient bucket: 0 stem+layer1, 1 layer2, 2 layer3, 3 layer4+head */
} mn_entry;

int mn_model_entries(int feat_dim);                       /* number of entries              */
int mn_model_entry(int feat_dim, int idx, mn_entry* out); /* idx-th entry in state_dict order */
int64_t mn_model_param_floats(int feat_dim);              /* parameter arena size incl. 4 criterion scalars */
int64_t mn_model_buffer_bytes(int feat_dim);              /* buffer arena size in bytes       */

/* ------------------------------------------------------------------------------------------
 * Training / inference plan.
 * ------------------------------------------------------------------------------------------ */
typedef struct mn_config {
  int32_t mode;        /* MN_MODE_* */
  int32_t dtype;       /* MN_DTYPE_*: storage type of activations and conv operands */
  int32_t windows;     /* N: windows per step on this device (PoseNet: images) */
  int32_t T;           /* frames per window (PoseNet: 1; MapNet++: T, images per window = 2T) */
  int32_t H, W;        /* image size */
  int32_t feat_dim;    /* 2048 */
  int32_t filter_nans; /* models/posenet.py:50-51 */
  float loss_scale;    /* static loss scale for fp16 gradients (1 for fp32) */
  int32_t eps_mode;    /* Adam epsilon placement: 0 torch>=1.0, 1 torch 0.4.1 */
} mn_config;

typedef struct mn_handle mn_handle;

int64_t mn_plan_bytes(const mn_config* cfg); /* work arena bytes (activations, gradients, scratch) */

/* params: fp32 [param_floats]; opt_state: fp32 [3*param_floats] = grads | exp_avg | exp_avg_sq;
 * buffers: [buffer_bytes]; work: [plan_bytes].  Arenas must outlive the handle. */
mn_handle* mn_create(const mn_config* cfg, float* params, float* opt_state, void* buffers, void* work,
                     void* stream);
void mn_destroy(mn_handle* h);

/* Criterion log-weights live at params[param_floats-4 ..]; which of them Adam may update:
 * replaces `learn_beta` / `learn_gamma` (common/criterion.py:39-40,71-74; scripts/train.py:104-112) */
int mn_set_learn_flags(mn_handle* h, int learn_beta, int learn_gamma);

/* replaces Optimizer(params, 'adam', base_lr, weight_decay, **kw) (common/optimizer.py:12-23)
 * and max_grad_norm (common/train.py:357-358) */
int mn_set_optim(mn_handle* h, float lr, float weight_decay, float beta1, float beta2, float eps,
                 float max_grad_norm);
/* The other two methods of the reference's Optimizer (common/optimizer.py:16-26; every shipped config uses 'adam'):
 * method 0 = Adam (default); 1 = torch.optim.SGD -- mn_set_optim's beta1 is `momentum`, beta2 is `dampening`, eps unused;
 * 2 = torch.optim.RMSprop (centered = False) -- beta1 is `momentum`, beta2 is `alpha`.  The momentum buffer lives in the
 * first moment arena (opt_state + param_floats), RMSprop's square average in the second. */
int mn_set_optim_method(mn_handle* h, int method, int nesterov);
int mn_set_step_count(mn_handle* h, int64_t step); /* Adam step counter (checkpoint resume): writes the device counter */
/* Early read-back of the training loss (the reference's `loss.item()`, common/train.py:361): with a pinned host float
 * registered, every training step copies the loss there as soon as the criterion has run and records an event;
 * mn_wait_loss blocks on that event only (0 = *pinned_host holds this step's loss, 1 = nothing was posted -- read the
 * device scalar passed as loss_out instead), while backward and the optimiser of the same step are still running. */
int mn_set_loss_host(mn_handle* h, float* pinned_host);
int mn_wait_loss(mn_handle* h);
/* optimiser steps APPLIED so far = the device-resident counter Adam's bias corrections use (steps skipped on an fp16
 * overflow are not steps); synchronises the device */
int64_t mn_get_step_count(mn_handle* h);

/* replaces model(data_var) in step_feedfwd (common/train.py:343) = MapNet.forward / PoseNet.forward
 * (models/posenet.py:65-73,87-97).  images: fp32 NCHW [windows*frames][3][H][W] on device (or uint8 NHWC
 * after mn_set_input_u8);
 * poses_out: fp32 [windows*frames][6].  training!=0 uses batch statistics and updates running
 * stats (model.train()), else running statistics (model.eval()). */
int mn_forward(mn_handle* h, const void* images, float* poses_out, int training, void* stream);

/* Dropout on the feature vector, between its ReLU and the two pose heads: replaces `F.dropout(x, p=self.droprate)`
 * (models/posenet.py:68-69).  The reference calls it WITHOUT `training=`: under its pinned PyTorch 0.4.1 the default is
 * training=False, i.e. an identity in train() and eval() alike; under PyTorch >= 1.0 the default is True, i.e. dropout is always
 * on.  The library implements the operator; which reading a model uses is the host's choice: p = 0 (default) is the pinned
 * identity, p > 0 drops in TRAINING forward passes (mn_train_step / mn_train_forward_loss / mn_forward with training != 0) and
 * never at inference.  The mask is inverted dropout (kept values scaled by 1 / (1 - p)) drawn from Philox4x32-10 with key `seed`
 * and counter (element / 4, number of training forward passes since this call): reproducible, independent of launch shapes, and
 * readable after a step as mn_debug_tensor "dropmask" ([images][feat_dim] floats, 0 or 1 / (1 - p)). */
int mn_set_dropout(mn_handle* h, float p, uint64_t seed);
/* The Philox counter's "training forward passes so far" of this plan (mn_set_dropout resets it to 0).  A host that runs several
 * plans of one model (another batch size for the last partial batch of an epoch) or resumes from a checkpoint sets it to its own
 * count of steps when a plan takes over, so that no plan replays the mask sequence from 0; data-parallel ranks pass different
 * seeds (scripts/train.py: seed ^ rank << 32) so that replicas do not drop the same features. */
int mn_set_dropout_calls(mn_handle* h, uint32_t calls);

/* Device-side input pipeline (replaces torchvision's ToTensor + Normalize of the reference's transforms,
 * scripts/train.py:120-128): with enable != 0 the `images` argument of mn_forward / mn_train_step /
 * mn_train_forward_loss is uint8 NHWC [windows*frames][H][W][3] (decoded image bytes) and
 * (x/255 - mean[c]) / std[c] is applied on the device; mean, std: 3 floats each in host memory.
 * Cuts the host-to-device copy of a 192-image step from 201 MB to 50 MB. */
int mn_set_input_u8(mn_handle* h, int enable, const float* mean, const float* std);

/* replaces criterion(output, target) (common/train.py:351): loss only, on given predictions */
int mn_loss(mn_handle* h, const float* pred, const float* targ, float* loss_out, void* stream);

/* replaces the train branch of step_feedfwd (common/train.py:343-361): forward, criterion,
 * zero_grad, backward, clip, Adam step.  loss_out: device fp32[1]; poses_out: device fp32. */
int mn_train_step(mn_handle* h, const void* images, const float* targets, float* loss_out, float* poses_out,
                  void* stream);

/* The same step in pieces, so a data-parallel host can all-reduce gradient bucket `stage`
 * (RCCL) while the remaining backward stages run.  Order: forward_loss, backward stage 3,2,1,0,
 * then optim_step(grad_mul = 1/world). */
int mn_train_forward_loss(mn_handle* h, const void* images, const float* targets, float* loss_out,
                          float* poses_out, void* stream);
int mn_train_backward_stage(mn_handle* h, int stage, void* stream);
int mn_grad_bucket(mn_handle* h, int stage, int64_t* offset, int64_t* count); /* range in the grads arena */
int mn_optim_step(mn_handle* h, float grad_mul, void* stream);
/* Optional bf16 transport of a bucket: pack = out_bf16[i] = bf16(grads[offset + i]), `count` values (mn_grad_bucket); the host
 * all-reduces them; unpack = grads[offset + i] = in_bf16[i].  Half the bytes on xGMI and half the time the collective's workgroups
 * share the CUs with the backward convolutions, at 8 bits of mantissa per rank's contribution (bf16 because weight gradients span
 * fp32's range: times the loss scale they overflow fp16).  geomapnet_amd/dp.py, MN_DP_GRAD_DTYPE=bf16. */
int mn_grad_bucket_pack_bf16(mn_handle* h, int stage, void* out_bf16, void* stream);
int mn_grad_bucket_unpack_bf16(mn_handle* h, int stage, const void* in_bf16, void* stream);

/* fp16 loss scaling.  The reference trains in fp32 (common/train.py:351-359) and cannot overflow; the fp16 plan scales
 * d(pred) by `scale` (mn_config.loss_scale initially) and divides it out where gradients enter the fp32 arena.  A step
 * whose gradients contain inf/NaN is SKIPPED on the device (parameters, Adam moments and the Adam step counter stay
 * untouched) and counted; the host halves the scale for each skipped step it has seen when the next step is enqueued
 * (no synchronisation: it reacts one or two steps late, the device keeps skipping meanwhile) and doubles it after
 * `growth_interval` clean steps (0 = never; default 2000, up to 65536).  mn_get_loss_scale reports the scale the next
 * step will use and the number of skipped steps that have reached the host (synchronise the stream first for an exact
 * count). */
int mn_set_loss_scale(mn_handle* h, float scale, int growth_interval);
int mn_get_loss_scale(mn_handle* h, float* scale, int64_t* skipped_steps);
/* CONSECUTIVE steps that were skipped although the loss scale already was 1 (as seen by the host so far): non-finite values no
 * loss scale can remove -- a NaN input or a forward pass that overflowed.  Any step that is applied, a scale growth or
 * mn_set_loss_scale resets the count to 0 (isolated bad batches do not accumulate); training makes no progress while it grows
 * and the Python mirror raises after a few of them (the reference, in fp32, would print NaN losses). */
int64_t mn_stuck_overflow_steps(mn_handle* h);

/* Inspection for parity tooling (tests/, tools/layer_error.py): device pointer, element count and MN_DTYPE_* of a named
 * tensor of the work arena as the last step left it.  Names: "xpad", "stem.y", "stem.gy", "p0", "gp0", "pooled", "feat",
 * "poses", "dposes", "dz", "dpooled", "dropmask", and per residual block i = 0..15 "b<i>.y1 | a1 | y2 | out | gy1 | ga1 | gy2 | gout"
 * (+ "yd", "zd", "gyd" for blocks with a projection); NHWC.  In the fp16x2 mode the tensors the convolutions consume ("p0",
 * "a1", "out", "zd", "gy1", "gy2", "gyd") are h2 tensors and report dtype MN_DTYPE_F16X2; in the fp16x2m mode the forward ones
 * of these are h2 and every gradient tensor of the blocks ("gp0", "gy1", "ga1", "gy2", "gout", "gyd") is plain fp16 (MN_DTYPE_F16),
 * (round 5's plain fp16 copies "p0.f16" / "b<i>.a1.f16" / "b<i>.out.f16" are gone: the backward kernels read the hi halves of the h2
 * tensors in place); in the fp16x2q mode the forward ones are
 * h2q tensors (geomapnet_amd/csrc/common.h) and report MN_DTYPE_F16X2Q.  Read-only for the caller. */
int mn_debug_tensor(mn_handle* h, const char* name, void** ptr, int64_t* numel, int32_t* dtype);

/* parameters changed behind the library's back (load_state_dict): refresh compute copies */
int mn_params_changed(mn_handle* h);

/* timing of the dominant kernel class inside the last mn_train_step (HIP events on `stream`).
 * which: 0 = all conv MFMA kernels (forward + data-gradient + weight-gradient). */
int mn_set_profiling(mn_handle* h, int enable);
int mn_last_kernel_ms(mn_handle* h, int which, float* ms, int* launches);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the parity tests and usable on their own).
 * ------------------------------------------------------------------------------------------ */
typedef struct mn_gather_geom {
  int32_t B, Hi, Wi, C, P, Q, R, S, mul_p, mul_q, rsign, ssign, off_h, off_w, div, M, N, K;
} mn_gather_geom;

/* out[m][n] = alpha * sum_k gather(A)[m][k] * Bw[n][k] (+bias, relu, residual): conv forward /
 * data-gradient / linear.  stats: [mn_op_igemm_grid_m(M)][2][N] fp32 partial column sums or NULL.
 * A and Bw are read through 32-bit-offset buffer resources (each must be smaller than 4 GiB); taps outside
 * the image are zero-filled by the hardware bounds check.  zero_page: >= 16 zero bytes in device memory,
 * 16-byte aligned; required (the weight-gradient kernels that gather strided convolutions read their
 * out-of-image taps from it; mn_op_igemm accepts it for symmetry). */
int mn_op_igemm(int dtype, const mn_gather_geom* g, const void* A, const void* Bw, void* out, int ldc, float* stats,
                const float* bias, int relu, const void* res, const void* res_gate, float alpha, const void* zero_page,
                void* stream);
int mn_op_igemm_grid_m(int M);
/* dW[n][colmap(k)] += alpha * sum_m dY[m][n] * gather(X)[m][k]  (fp32 atomics into dW) */
int mn_op_wgrad(int dtype, const mn_gather_geom* g, const void* dY, int ldy, const void* X, float* dW, int ldw,
                const int32_t* colmap, float alpha, int target_blocks, const void* zero_page, void* stream);
/* The same operator with scratch for partial results: the 3x3 stride-1 fp16 layers (csrc/wgrad_fused.h) then store one
 * partial tile per pixel range into `ws` (plain stores) and add the ranges up in index order in a second launch, instead
 * of fp32 atomics into dW.  ws: device fp32 [ws_floats], ws_floats >= mn_op_wgrad_ws_floats(); launches sharing one
 * workspace must be ordered on one stream.  Other layer shapes ignore the workspace. */
int64_t mn_op_wgrad_ws_floats(void);
int mn_op_wgrad_ws(int dtype, const mn_gather_geom* g, const void* dY, int ldy, const void* X, float* dW, int ldw, float alpha,
                   float* ws, int64_t ws_floats, const void* zero_page, void* stream);
/* fp16 3x3 stride-1 same-size convolution of 64 -> 64 channels (ResNet layer1 forward, and -- with the mirrored geometry
 * rsign = ssign = -1 -- its data gradient) from an LDS-resident 18x18-pixel input halo per 16x16-pixel output tile, as a
 * persistent kernel (csrc/halo_pp.h): one 8-wave workgroup per CU, two wave groups alternating between the MFMA loop of one
 * tile and the epilogue + next halo fetch of another, all nine weight slices LDS-resident.  Operands and epilogue as
 * mn_op_igemm plus out_gate (result zeroed where out_gate <= 0; at most one of res_gate / out_gate).  stats_accum:
 * [stats_rows][2][64] fp64 column sums (sum, sum of squares), ADDED to atomically, or NULL (only without residual / gates).
 * wgs: number of persistent workgroups, 0 = one per CU. */
int mn_op_conv_halo_pp(const mn_gather_geom* g, const void* A, const void* Bw, void* out, int ldc, double* stats_accum,
                       int stats_rows, int relu, const void* res, const void* res_gate, const void* out_gate, float alpha,
                       int wgs, void* stream);
/* 3x3 stride-1 same-size FORWARD convolution of 64 -> 64 channel h2 tensors (ResNet layer1 in the fp16x2 / fp16x2m modes) with the
 * weights held in registers (csrc/halo_h2.h): one persistent 4-wave workgroup per CU, a wave keeps the hi and lo halves of its 32
 * output channels' weights as MFMA operands for the whole launch and only the input halo of an 8 x 16-pixel tile passes through LDS.
 * A: h2 activation [B][H][W][64], Bw: h2 weights [64][9*64] (mn_op_igemm's dtype-3 operands), out: fp32 [B][H][W][64];
 * stats_accum: optional [stats_rows][2][64] fp64 column sums (sum, sum of squares), added to atomically. */
int mn_op_conv_halo_h2(const mn_gather_geom* g, const void* A, const void* Bw, float* out, int ldc, double* stats_accum,
                       int stats_rows, void* stream);
/* Data gradient of a convolution (what autograd computes for conv2d's input, torch 0.4.1 `loss.backward()` under
 * common/train.py:351): gx[B][Hin][Win][Cin] = conv_transpose(gy[B][Hout][Wout][Cout], W) (+ res, res only where
 * res_gate > 0), zeroed where out_gate <= 0.  wd: weights in the data-gradient layout [Cin][k][k][Cout].  stride 1 or 2;
 * parity != 0: a stride-2 gradient is computed per parity class of the input pixel (1+2+2+4 taps of a 3x3 kernel
 * instead of 9 taps of which three quarters are structural zeros); both forms give the same result. */
int mn_op_conv_dgrad(int dtype, int B, int Hin, int Win, int Cin, int Cout, int k, int stride, int pad, const void* gy,
                     const void* wd, void* gx, const void* res, const void* res_gate, const void* out_gate, int parity,
                     const void* zero_page, void* stream);
/* Stem convolution forward (torchvision ResNet conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels) for fp16 tensors
 * (csrc/stem.h).  xpad: [B][H+6][Wp][4] zero-padded NHWC4 input (3 rows / columns of padding at the top / left, channel 3
 * zero), Wp even and >= W + 7; wf: [64][224] weights in the pixel-pair layout [n][r][s4][2 pixels x 4 channels] (tap
 * column 7 and channel 3 zero); y: [B][(H-1)/2+1][(W-1)/2+1][64]; stats_accum: optional [stats_rows][2][64] fp64 column
 * sums (sum, sum of squares), added to atomically. */
int mn_op_stem_conv(const void* xpad, const void* wf, void* y, double* stats_accum, int stats_rows, int B, int H, int W, int Wp,
                    void* stream);
/* the same kernel shape for fp32 tensors (fp32x3 / fp16x2 modes): fp32 padded input and pair-layout weights in, operands split
 * into fp16 hi + lo halves on the way into LDS / registers, three MFMAs per product, fp32 output */
int mn_op_stem_conv_x3(const float* xpad, const float* wf, float* y, double* stats_accum, int stats_rows, int B, int H, int W,
                       int Wp, void* stream);
/* The dense layer of the pose head in fp32 (csrc/dense.h; /root/reference/models/posenet.py:46,65-66: self.feature_extractor.fc and
 * the ReLU behind it): C[M][N] = act(A[M][K] . W[N][K]^T + bias[N]); K a multiple of 128, bias optional, relu 0 / 1.  With W = the
 * transposed weight the same call is the layer's data gradient. */
int mn_op_dense(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int relu, void* stream);
/* ... its weight and bias gradient: dW[F][Cin] += alpha * dY[B][F]^T . X[B][Cin], db[F] += alpha * column sums of dY (db optional) */
int mn_op_dense_wgrad(const float* dY, const float* X, float* dW, float* db, int B, int F, int Cin, float alpha, void* stream);
/* weight and bias gradients of the two pose regressors (posenet.py:48-49,71-73): dWx[3][K] / dbx[3] from dposes[:, 0:3], dWq[3][K] /
 * dbq[3] from dposes[:, 3:6], all += scale * sum over the B rows; filter_nans zeroes NaN entries of the rotation gradients
 * (filter_hook, posenet.py:28-34) */
int mn_op_head_wgrad(const float* dposes, const float* feat, float* dWx, float* dbx, float* dWq, float* dbq, int B, int K, float scale,
                     int filter_nans, void* stream);
/* Stem backward (fp16) below maxpool(relu(bn1(conv1(x)))) in two launches (csrc/stem_bwd.h): BatchNorm sums with the
 * max-pool's input gradient gathered on the fly from (idx, gp), then conv1's weight gradient with d(conv output) computed
 * tile by tile in LDS.  y: raw conv output [B][H0][W0][64]; idx / gp: argmax bytes and gradient of the pooled activation
 * [B][Po][Qo][64]; xpad / Wp as mn_op_stem_conv; dW: [64][ldw] fp32 (+=, atomics), column k of the 224-column pair layout
 * goes to colmap[k] (or is dropped where colmap[k] < 0); dgamma / dbeta: += ; coef_scratch: 4*64 floats, accum_scratch:
 * 2*64 doubles; alpha multiplies every gradient (1 / loss scale). */
int mn_op_stem_bwd(const void* y, const unsigned char* idx, const void* gp, const float* gamma, const float* beta,
                   const float* mean, const float* invstd, const void* xpad, float* dW, int ldw, const int32_t* colmap,
                   float* dgamma, float* dbeta, float* coef_scratch, double* accum_scratch, int B, int H, int W, int Wp,
                   float alpha, void* stream);
/* conv weight layout helpers: OIHW fp32 <-> OHWI fp32 */
int mn_op_oihw_to_ohwi(const float* src, float* dst, int O, int I, int H, int W, int to_ohwi, void* stream);

/* criteria: replaces PoseNetCriterion/MapNetCriterion/MapNetOnlineCriterion.forward + autograd
 * backward (common/criterion.py).  s: device fp32[4].  dpred/ds/vos_out may be NULL. */
int mn_op_criterion(int mode, int N, int T, const float* pred, const float* targ, const float* s, float* loss,
                    float* dpred, float* ds, float* vos_out, float grad_scale, void* stream);
/* replaces pose_utils.calc_vos (common/pose_utils.py:248-260) and its autograd VJP */
int mn_op_calc_vos(const float* poses, int N, int T, float* vos, const float* cot, float* dposes, void* stream);

/* Pose-graph optimisation of W windows in one launch (one wavefront per window, fp64).  Replaces
 * PoseGraph.optimize / PoseGraphFC.optimize behind optimize_poses (common/pose_utils.py:458-804), which
 * scripts/eval.py:177-182 calls per window on the host.  poses [W][N][7] (t, unit quaternion w x y z), vos [W][P][7]
 * with P = N-1 (consecutive pairs) or N(N-1)/2 (fc_vos: all pairs i<j in lexicographic order), out [W][N][7],
 * status [W] (1 = the normal matrix was not positive definite, where scipy raises LinAlgError).  2 <= N <= 12.
 * sax..srq are the covariances of optimize_poses; n_iters = 10 in the reference.  All pointers are device fp64. */
int mn_pgo_optimize(const double* poses, const double* vos, double* out, int32_t* status, int W, int N, int fc_vos,
                    double sax, double saq, double srx, double srq, int n_iters, void* stream);

/* fused Adam over a flat range; replaces torch.optim.Adam.step + clip_grad_norm */
int mn_op_adam(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_clip, float lr, float wd,
               float beta1, float beta2, float eps, int64_t step, float grad_mul, float max_norm, double* sqnorm_scratch,
               int eps_mode, void* stream);
/* the same launch for any method of mn_set_optim_method (method, nesterov as there; beta1 / beta2 carry momentum and
 * dampening | alpha); step = 1 is SGD's first step */
int mn_op_optim(int method, int nesterov, float* p, const float* g, float* m, float* v, int64_t n, int64_t n_clip, float lr,
                float wd, float beta1, float beta2, float eps, int64_t step, float grad_mul, float max_norm,
                double* sqnorm_scratch, int eps_mode, void* stream);

/* elementwise / reduction operators (NHWC, C multiple of 16 bytes).
 * Scratch: mn_op_bn_train_fwd's accum_scratch = 2*C doubles followed by 2*C floats; mn_op_bn_bwd's accum_scratch = 2*C
 * doubles (handed back zeroed), coef_scratch = 3*C floats. */
int mn_op_bn_train_fwd(int dtype, const void* y, int64_t M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* mean, float* invstd, const void* res,
                       int relu, void* out, float eps, float momentum, double* accum_scratch, void* stream);
int mn_op_bn_bwd(int dtype, const void* g, const void* gate, const void* y, int64_t M, int C, const float* gamma,
                 const float* mean, const float* invstd, float* dgamma, float* dbeta, void* gy, float* coef_scratch,
                 double* accum_scratch, float grad_unscale, void* stream);
/* 3x3/2 pad 1 max-pool; idx (optional, uint8 per output element) records the winning tap for the backward */
int mn_op_maxpool_fwd(int dtype, const void* in, void* out, unsigned char* idx, int B, int H, int W, int C,
                      void* stream);
int mn_op_maxpool_bwd(int dtype, const unsigned char* idx, const void* gout, void* gin, int B, int H, int W, int C,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference, which is single-device: common/train.py:91-92).
 * Stand-in for the reduction workgroups of an RCCL ring all-reduce on a one-GPU box: `workgroups` x `threads` stay resident
 * for `microseconds`, streaming `bytes` of dst += src (fp32, 16-byte pieces) at an even pace over that time.  geomapnet_amd/dp.py
 * launches it where a gradient bucket's all-reduce is issued (MN_DP_STANDIN) so that the CU contention between the collective and
 * the backward convolutions -- one workgroup per CU, launches sized as one round of the chip -- can be measured without peers
 * (tools/rccl_rehearsal.py, profiles/r06/rccl_rehearsal.txt).  src / dst may be NULL with bytes = 0 (residency only); lds_kb = 0, 32
 * or 64: LDS each stand-in workgroup holds (a CU hosting one with tens of KB cannot also host a 150 KB convolution workgroup).
 * ------------------------------------------------------------------------------------------ */
int mn_op_occupy(int workgroups, int threads, float microseconds, const void* src, void* dst, int64_t bytes, int lds_kb,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPNET_HIP_H */
