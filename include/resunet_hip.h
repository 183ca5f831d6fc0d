/* resunet_hip.h -- C-ABI of libresunet_hip.so, the MI355X (gfx950) implementation of the ResUNet hot path
 * of lachinov/brats2019.
 *
 * The reference has no FFI of its own: the path is stock torch.nn calls inside model.py / loss.py.  This
 * header is therefore the boundary a maintainer binds with ctypes; each entry point
 * names the reference call site (file:line into the reference repo) whose arithmetic it replaces.
 *
 * Conventions:
 *   - all tensors are contiguous NCDHW float32 DEVICE pointers owned by the caller (PyTorch-ROCm
 *     `tensor.data_ptr()`); the library never allocates user-visible memory -- scratch comes from the
 *     caller-provided workspace (`ws`, `ws_bytes`; sizes from the *_workspace_bytes queries);
 *   - every function enqueues on `stream` (a hipStream_t passed as void*; 0 = default stream), performs no
 *     device synchronisation and no hipMalloc: calls are graph-capturable;
 *   - return 0 on success, a negative RU_E* code on error; `ru_last_error()` returns the thread-local
 *     message; nothing throws across the ABI.
 */
#ifndef RESUNET_HIP_H_
#define RESUNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RU_OK 0
#define RU_EINVAL (-1)    /* bad argument / unsupported shape */
#define RU_ENOMEM (-2)    /* workspace too small */
#define RU_EHIP (-3)      /* HIP runtime error (launch failed ...) */
#define RU_ESTATE (-4)    /* call order violated (backward without forward ...) */

/* arithmetic of the 3x3x3 convolutions (everything else is always float32):
 *   RU_PREC_F32    exact float32 products on v_mfma_f32_16x16x4_f32 (bit-for-bit an fmaf chain)
 *   RU_PREC_BF16X3 split-bf16: v = hi + lo (2 x bf16), products hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_bf16 with
 *                  float32 accumulation; ~2^-16 relative per product, |dp| ~ 5e-5 on the whole network (bar: 1e-3).
 *                  Needs W % 4 == 0, otherwise the f32 kernel runs.  HBM tensors stay float32 in both modes. */
#define RU_PREC_F32 0
#define RU_PREC_BF16X3 1
#define RU_PREC_BF16 2        /* gradient precision only (ru_unet_set_grad_precision): plain bf16 operands, one MFMA product, fp32 accumulate */

typedef void* ru_stream_t;                /* hipStream_t */
typedef struct ru_unet* ru_unet_t;        /* opaque engine handle */

const char* ru_last_error(void);
int ru_version(void);                     /* 100*major + minor */
/* 1 if a HIP device is usable by this process, 0 otherwise (never throws) */
int ru_device_ok(void);

/* ---------------------------------------------------------------- convolutions
 * nn.Conv3d as the reference constructs it: (k=3,stride=1,pad=1) model.py:72-73,336,348;
 * (k=2,stride=2,pad=0) model.py:361-363; (k=1,stride=1,pad=0) model.py:393,401.  Cross-correlation,
 * zero padding, weight layout [Cout][Cin][k][k][k].  D,H,W are the INPUT extents; for k=2 they must be even.
 * `bias` may be NULL.  Exact-f32 arithmetic (v_mfma_f32_16x16x4_f32 / v_fmac_f32).  */
size_t ru_conv3d_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int k);
int ru_conv3d_fwd(const float* x, const float* w, const float* bias, float* y,
                  int N, int Cin, int Cout, int D, int H, int W, int k,
                  void* ws, size_t ws_bytes, ru_stream_t stream);
/* same as ru_conv3d_fwd / ru_conv3d_bwd_data with an explicit RU_PREC_* for k=3 (k=1,2 are always float32) */
int ru_conv3d_fwd_p(const float* x, const float* w, const float* bias, float* y,
                    int N, int Cin, int Cout, int D, int H, int W, int k, int precision,
                    void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_conv3d_bwd_data_p(const float* dy, const float* w, float* dx,
                         int N, int Cin, int Cout, int D, int H, int W, int k, int precision,
                         void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_conv3d_bwd_weight_p(const float* x, const float* dy, float* dw, float* db,
                           int N, int Cin, int Cout, int D, int H, int W, int k, int precision,
                           void* ws, size_t ws_bytes, ru_stream_t stream);
/* dx = d(loss)/dx given dy (autograd of the calls above) */
int ru_conv3d_bwd_data(const float* dy, const float* w, float* dx,
                       int N, int Cin, int Cout, int D, int H, int W, int k,
                       void* ws, size_t ws_bytes, ru_stream_t stream);
/* dw [Cout][Cin][k][k][k] and (if db != NULL) db[Cout] */
int ru_conv3d_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                         int N, int Cin, int Cout, int D, int H, int W, int k,
                         void* ws, size_t ws_bytes, ru_stream_t stream);

/* ---------------------------------------------------------------- GroupNorm (+ fused LeakyReLU / residual)
 * nn.GroupNorm(G, C), eps, affine (model.py:95-96,338) followed by LeakyReLU(slope) (model.py:93-94; pass
 * slope = 1 for "no activation", as after norm_input, model.py:413) and, if residual != NULL, the Residual
 * skip add `x + out` (model.py:115).  V = D*H*W.  mean/rstd: [N*G] outputs (saved for backward).  */
size_t ru_groupnorm_workspace_bytes(int N, int C, size_t V);
int ru_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                     float* mean, float* rstd, int N, int C, size_t V, int G, float eps, float slope,
                     void* ws, size_t ws_bytes, ru_stream_t stream);
/* backward of y = lrelu(GN(x)): dy is d/d(activated output); the LeakyReLU mask is recomputed from the sign
 * of the normalised value (== sign of the in-place output the reference keeps).
 * dgamma/dbeta are OVERWRITTEN.  */
int ru_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                     const float* dy, float* dx, float* dgamma, float* dbeta,
                     int N, int C, size_t V, int G, float slope,
                     void* ws, size_t ws_bytes, ru_stream_t stream);

/* ---------------------------------------------------------------- LeakyReLU (model.py:352,422) */
int ru_leaky_relu_fwd(const float* x, float* y, size_t n, float slope, ru_stream_t stream);
int ru_leaky_relu_bwd(const float* y, const float* dy, float* dx, size_t n, float slope, ru_stream_t stream);

/* ---------------------------------------------------------------- trilinear x2 (model.py:12-14)
 * F.interpolate(scale_factor=2, mode='trilinear'), align_corners=False.  D,H,W = INPUT extents.  */
int ru_upsample2x_trilinear_fwd(const float* x, float* y, int N, int C, int D, int H, int W, ru_stream_t stream);
int ru_upsample2x_trilinear_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W, ru_stream_t stream);

/* ---------------------------------------------------------------- sigmoid (model.py:351,431) */
int ru_sigmoid_fwd(const float* x, float* y, size_t n, ru_stream_t stream);

/* ---------------------------------------------------------------- criterion (loss.py:64-79,98-122; train.py:203-205)
 * Phase 1: per-class partial sums over THIS rank's shard, float64:
 *     sums[0..C)  = sum_{n,v} p*g          (Dice intersection, no epsilon)
 *     sums[C..2C) = sum_{n,v} (p*p + g)    (Dice union, no epsilon)
 *     sums[2C]    = sum g*log(p+1e-6) + bg_weight*(1-g)*log((1+1e-6)-p)
 * (all-reduce `sums` across data-parallel ranks between the phases).
 * Phase 2: dp = w_dice * dDice/dp + w_bce * dBCE/dp with the GLOBAL sums and GLOBAL element count
 * (`count` = N_global*C*V).  The reference's training criterion is w_dice = w_bce = 0.5, bg_weight 1e-2,
 * priority 1 (main.py:126-128).  loss value: see ru_criterion_value().  */
size_t ru_criterion_workspace_bytes(int N, int C, size_t V);
int ru_criterion_sums(const float* p, const float* g, double* sums, int N, int C, size_t V, float bg_weight,
                      void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_criterion_grad(const float* p, const float* g, const double* sums, double count,
                      float w_dice, float w_bce, float bg_weight, float priority,
                      float* dp, int N, int C, size_t V, ru_stream_t stream);
/* host helper: (dice, bce) from global sums (host pointer) */
int ru_criterion_value(const double* sums_host, int C, double count, double priority, double* dice, double* bce);
/* the same on the device, one launch, no host sync: out[0] = w_dice*dice + w_bce*bce (train.py:203-205 with w = 1/2), out[1] = dice
 * (loss.py:114-122), out[2] = bce (loss.py:79) as float64, from the (all-reduced) device sums */
int ru_criterion_value_device(const double* sums, int C, double count, double priority, double w_dice, double w_bce,
                              double* out3, ru_stream_t stream);

/* ---------------------------------------------------------------- optimizer (main.py:133-142)
 * torch.optim.Adam(amsgrad=True) with L2 weight decay added to the gradient; `step` is 1-based.  */
int ru_adam_amsgrad_step(float* w, const float* g, float* m, float* v, float* vmax, size_t n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         ru_stream_t stream);
/* the same with amsgrad optional: vmax_or_null == NULL is torch.optim.Adam(amsgrad=False).  What brats2019_amd.optim.Adam.step() -- the
 * optimizer class Trainer.train instantiates in place of torch.optim.Adam (main.py:134, train.py:82-83) -- calls once per contiguous run
 * of parameters of the flat buffer.  */
int ru_adam_step(float* w, const float* g, float* m, float* v, float* vmax_or_null, size_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                 ru_stream_t stream);

/* ---------------------------------------------------------------- whole-network engine
 * model.UNet(depth, encoder_layers, decoder_layers, number_of_channels, number_of_outputs) (model.py:309)
 * with block=Residual and 4 input channels (model.py:336).  Parameters live in ONE flat float buffer in
 * reference state_dict() order (ru_unet_param_* describe it); gradients are written to a flat buffer of
 * the same layout (dead parameters -- decoder_convs.{depth-1}.*, decoder_convs1x1.{depth-1} -- get zeros,
 * model.py:420: they are constructed but never executed).  */
ru_unet_t ru_unet_create(int depth, const int* encoder_layers, const int* decoder_layers,
                         const int* number_of_channels, int number_of_outputs);
void ru_unet_destroy(ru_unet_t h);
/* RU_PREC_* used by the 3x3x3 convolutions (forward, data gradient, weight gradient) of subsequent forward/backward calls;
 * default RU_PREC_F32.  Change it only between steps (not between a forward and its backward).  */
int ru_unet_set_precision(ru_unet_t h, int precision);
/* Inference with constant weights: frozen != 0 promises that the values behind `params` and the workspace `ws` are not modified between
 * ru_unet_forward calls; the weight packs the forward builds at the head of `ws` are then built once and reused while the same
 * (params pointer, ws pointer, precision) come back (they are rebuilt otherwise).  Training must leave it off (0, the default): the
 * reference's optimizer changes the weights every step (train.py:220).  */
int ru_unet_freeze_params(ru_unet_t h, int frozen);
int ru_unet_get_precision(ru_unet_t h);
/* Arithmetic of the 3x3x3 DATA and WEIGHT gradients of subsequent ru_unet_backward calls when the forward precision is RU_PREC_BF16X3
 * (ignored otherwise).  RU_PREC_BF16X3 (default): three split-bf16 products like the forward, gradients within ~1e-5 relative of float32.
 * RU_PREC_BF16: the operands of the gradient convolutions are rounded to bf16 (hi*hi only, fp32 accumulate) -- BASELINE configs[2]
 * ("bf16 forward+backward") taken literally for the backward; the forward, and with it the probabilities (bar 1e-3, train.py:201-205 /
 * model.py:407-433), is untouched.  One third of the matrix work and half of the operand staging in those kernels; parameter gradients
 * then carry bf16 rounding noise (relative L2 error ~3e-3 per tensor against the float32 reference, tests/test_hip_unet.py).  The
 * persistent voxel-major kernels honour it (every level of the shipped configuration at training sizes); the small-shape fallbacks and
 * the 4-channel stem / head data paths keep three products. */
int ru_unet_set_grad_precision(ru_unet_t h, int precision);
int ru_unet_get_grad_precision(ru_unet_t h);
/* Backward-pass fusions of the voxel-major split-bf16 engine (on by default unless noted; same arithmetic either way up to summation order --
 * the separate passes stay available so that tests can hold the fused kernels to them):
 *   RU_FUSE_GN_BWD_STATS  the GroupNorm-backward sums are taken in the epilogue of the data-gradient conv that produces the incoming
 *                         gradient (no reduce pass over (y, d));
 *   RU_FUSE_GN_BWD_APPLY  16-channel level: the GroupNorm-backward apply is computed in the weight gradient's dy staging (no apply pass). */
#define RU_FUSE_GN_BWD_STATS 1
#define RU_FUSE_GN_BWD_APPLY 2
/*   RU_FUSE_SIDE_STREAM   the 3x3x3 weight gradients below the 16-channel level run on a second (library-owned, lower-priority) HIP stream,
 *                         event-ordered behind the kernel that produces their dy and joined before ru_unet_backward returns: same kernels,
 *                         same arithmetic, bit-identical gradients; the memory-bound passes of the chain run in their shadow.  */
#define RU_FUSE_SIDE_STREAM 4
/*   RU_FUSE_BATCH_WREDUCE the per-workgroup partials of every weight gradient of a backward pass are summed by ONE launch behind the last
 *                         weight-gradient kernel (nobody reads a weight gradient before the optimizer) instead of one small launch each:
 *                         same summation order, bit-identical gradients.
 *   RU_FUSE_TAIL_FINALIZE the GroupNorm statistics (forward) and the GroupNorm-backward coefficients are finalized by the LAST workgroup of
 *                         the kernel that produces their partial sums (one integer ticket per launch, reset by the finisher; partials
 *                         published with agent-scope stores and read back in the same fixed order: deterministic, no float atomics)
 *                         instead of by a finalize launch behind it.  OFF by default: built, verified and measured (round 4) -- the
 *                         serial tail (ticket + reading the partials back through the fabric) costs what the finalize launch and its
 *                         boundary cost.  */
#define RU_FUSE_BATCH_WREDUCE 8
#define RU_FUSE_TAIL_FINALIZE 16
/*   RU_FUSE_PW_DGRAD      decoder 1x1x1 conv over the (never materialised) concat, Cout <= 32: its weight-gradient kernel also forms the data
 *                         gradient of both halves from the dy tile it has staged (LeakyReLU backward of the up-sampled half included) --
 *                         one pass over (dy, skip, up) instead of two.  */
#define RU_FUSE_PW_DGRAD 32
int ru_unet_set_fusion(ru_unet_t h, unsigned mask);
/* In-situ timing of the dominant kernel (bench.py's roofline line): while enabled, every forward brackets its launches of
 * the 3x3x3 convolution 16 -> 16 at the input resolution (voxel-major split-bf16 engine; the forward of the shipped net has four) with a
 * HIP event pair on the stream the kernels run on.  ru_unet_probe_read waits for the recorded events, returns the summed duration and
 * the number of launches since the last read, and clears the record.  Off by default; costs two event records per probed launch. */
int ru_unet_probe(ru_unet_t h, int enable);
int ru_unet_probe_read(ru_unet_t h, double* total_ms, int* launches);
/* ru_unet_probe(h, 2): EVERY launch of ru_unet_forward / ru_unet_backward is bracketed and booked to one of RU_PROBE_FAMILIES kernel
 * families (bench.py's `roofline_families`): 0 3x3x3 conv fwd + data gradient at the 16-channel level (stem and head included), 1 the
 * same at the deeper levels, 2 / 3 3x3x3 weight gradients likewise, 4 GroupNorm passes (apply, backward reduce / apply, finalizes),
 * 5 1x1 / 2x2x2 convolutions, their weight gradients and the trilinear kernels, 6 everything else (weight packing, fills, head
 * gradient, partial-sum reductions of the weight gradients).  ms[f] = summed duration, launches[f] = count since the last read.
 * The event records cost ~1 us per launch: use it in probe steps, not in a timed region.
 * nfam = RU_PROBE_FAMILIES + RU_PROBE_INSTANCES additionally returns, behind the families, the launches that are ONE kernel instantiation worth naming
 * (bench.py's `roofline_top`; each is booked to its family as well): 0 / 1 the 16 -> 16 voxel-major 3x3x3 convolution forward / data gradient, 2 / 3 the
 * 32..128-channel ones, 4 the 16-channel level's weight gradient with the GroupNorm-backward apply fused into its staging, 5 the one without, 6 the
 * 32..128-channel weight gradients. */
#define RU_PROBE_FAMILIES 7
#define RU_PROBE_INSTANCES 7
int ru_unet_probe_read_families(ru_unet_t h, double* ms, int* launches, int nfam);
int ru_unet_param_count(ru_unet_t h);
const char* ru_unet_param_name(ru_unet_t h, int i);          /* state_dict key */
int ru_unet_param_ndim(ru_unet_t h, int i);
int ru_unet_param_dim(ru_unet_t h, int i, int d);
size_t ru_unet_param_offset(ru_unet_t h, int i);             /* in floats, into the flat buffer */
size_t ru_unet_param_total(ru_unet_t h);                     /* floats */
int ru_unet_param_is_dead(ru_unet_t h, int i);
/* Bytes of device workspace `ws` a forward (training == 0: inference, block temporaries are recycled) or a forward + backward pair
 * (training != 0: every activation is kept) needs at this shape; 0 for extents the network cannot take.  `ws` must be 256-byte aligned
 * (any hipMalloc pointer is) and may be larger than asked for. */
size_t ru_unet_workspace_bytes(ru_unet_t h, int N, int D, int H, int W, int training);
/* UNet.forward (model.py:407-433): x [N,4,D,H,W] -> probs [N,n_out,D,H,W] (sigmoid).  D,H,W divisible by
 * 2^(depth-1).  training != 0 keeps activations in `ws` for ru_unet_backward; `x` and `probs` are then read again by
 * ru_unet_backward (weight gradient of conv_input; sigmoid backward) and must stay valid and unmodified until it has run.  */
int ru_unet_forward(ru_unet_t h, const float* params, const float* x, float* probs,
                    int N, int D, int H, int W, int training,
                    void* ws, size_t ws_bytes, ru_stream_t stream);
/* loss.backward() through the network (train.py:210): dprobs = d(loss)/d(probs) -> grads (flat, overwritten).
 * Must follow a training-mode ru_unet_forward with the same params/ws.  dx (may be NULL) = d/d(input).  */
int ru_unet_backward(ru_unet_t h, const float* params, const float* dprobs, float* grads, float* dx,
                     ru_stream_t stream);
/* The same with the criterion's gradient formed on the way (train.py:203-210: `loss = (Dice + BCE) / 2; loss.backward()`): dprobs is not an
 * operand but the second phase of the criterion -- ru_criterion_grad(probs, target, sums, count, w_dice, w_bce, bg_weight, priority) --
 * evaluated inside the head's sigmoid-backward pass from the probabilities the forward wrote to the caller's buffer, `target` and the
 * (all-reduced) `sums` of ru_criterion_sums: the gradient w.r.t. the probabilities is never written or read back (one pass instead of
 * three over the class tensors; the same float expressions, so the gradients agree with the two-call sequence to float rounding --
 * 1e-7 relative L2 measured, tests/test_hip_unet.py).  */
int ru_unet_backward_criterion(ru_unet_t h, const float* params, const float* target, const double* sums, double count,
                               float w_dice, float w_bce, float bg_weight, float priority, float* grads, float* dx,
                               ru_stream_t stream);
/* per-layer GroupNorm statistics of the last forward, for parity checks: copies mean/rstd [N*8] of the
 * idx-th executed GroupNorm (execution order) into DEVICE buffers.  Returns number of GN layers if idx<0. */
int ru_unet_gn_stats(ru_unet_t h, int idx, float* mean, float* rstd, ru_stream_t stream);

/* ---------------------------------------------------------------- data-parallel collectives (replaces nn.DataParallel, main.py:61)
 * One process per GPU, full replica each, minibatch sharded over the ranks.  Per step two SUM all-reduces keep the result
 * identical to the reference's global-batch step: the [2C+1] float64 criterion sums between ru_criterion_sums and ru_criterion_grad
 * (Dice_loss_joint sums over the GLOBAL batch, loss.py:114-115), and the live runs of the flat float32 gradient buffer after
 * ru_unet_backward (SUM, not mean).  ru_allreduce enqueues ncclAllReduce (RCCL over xGMI) on `stream` -- the stream of the kernels --
 * in place; no host synchronisation.  librccl is bound at run time: the library loads without it, these calls then fail with RU_EHIP.
 * Set-up: rank 0 calls ru_comm_unique_id and hands the RU_COMM_ID_BYTES bytes to every rank out of band (the host mirror broadcasts
 * them through torch.distributed or a file); every rank then calls ru_comm_init with its HIP device current.  */
typedef struct ru_comm* ru_comm_t;
#define RU_COMM_ID_BYTES 128
#define RU_DT_F32 0
#define RU_DT_F64 1
int ru_comm_unique_id(void* id_out /* RU_COMM_ID_BYTES host bytes */);
int ru_comm_init(ru_comm_t* out, const void* id, int rank, int world);
int ru_comm_destroy(ru_comm_t c);
int ru_comm_rank(ru_comm_t c);
int ru_comm_world(ru_comm_t c);
int ru_allreduce(ru_comm_t c, void* buf, size_t count, int dtype, ru_stream_t stream);
/* ncclGroupStart / ncclGroupEnd: the ru_allreduce calls between them are issued as ONE RCCL launch (the runs of the gradient bucket of
 * a step; replaces the per-parameter reduce_add loop inside nn.DataParallel's backward, main.py:61) */
int ru_comm_group_begin(void);
int ru_comm_group_end(void);

/* ---------------------------------------------------------------- inference post-processing (test.py:115-159)
 * ru_tta_merge: `probs` holds K predictions [K][C][D][H][W] of flipped copies of one volume; bits 3k..3k+2 of `flips` say
 * which axes (bit0 D, bit1 H, bit2 W) copy k was reversed along (test.py:117-120 uses {none, D, H, D+H}).  Each prediction
 * is un-flipped (test.py:134-136) and they are averaged as sum(outputs)/K in list order (test.py:138, float32).  Outputs:
 * mean_out [C][D][H][W] (may be NULL), mask [C][V] uint8 = mean > 0.5 (test.py:144), counts[C] = voxels set per channel.
 * ru_compose_labels: labels[v] = 2 where mask[0], then 1 where mask[1], then 4 where mask[2] if counts[2] > et_min
 * (test.py:153-159, et_min = 32).  counts is read on the device: no host synchronisation.  */
int ru_tta_merge(const float* probs, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts,
                 int C, int D, int H, int W, ru_stream_t stream);
int ru_compose_labels(const unsigned char* mask, const unsigned long long* counts, unsigned long long et_min, unsigned char* labels,
                      size_t V, ru_stream_t stream);

/* ---------------------------------------------------------------- inference driver on the device (csrc/inference.hip)
 * Sliding window (loader_helper.py:34-97, train.py:158-174).  `origins` = T x 3 HOST ints, index_min of each tile (get_indices: centre
 * block position * centre - border; may be negative).  ru_tile_gather is loader_helper.copy for T tiles in one launch: tiles
 * [(t*N + n), C, td, th, tw] = data[n, :, origin + (z, y, x)], zero outside the volume (tw % 4 == 0).  ru_tile_scatter is
 * loader_helper.copy_back: the centre block [border, border + center) of every tile is pasted at origin + border, clipped at the end
 * of the volume.  Centre blocks of different tiles do not overlap, so the order of the tiles does not matter.  */
int ru_tile_gather(const float* data, float* tiles, int N, int C, int D, int H, int W, int T, const int* origins,
                   int td, int th, int tw, ru_stream_t stream);
int ru_tile_scatter(const float* tiles, float* out, int N, int C, int D, int H, int W, int T, const int* origins,
                    int td, int th, int tw, const int* border, const int* center, ru_stream_t stream);
/* Case preparation (test.py:47-49,85-120).  ru_case_bbox: box[c*6 .. c*6+5] (DEVICE ints) = {min z, y, x, max z, y, x} of the non-zero
 * voxels of modality c of image [C][D][H][W]; {INT_MAX x3, -1 x3} for an all-zero modality (the host applies test.py:47-49's union and
 * its rule for empty modalities).  ru_case_stats: per channel over the crop box [lo, lo + size): stats[c*3..] = count(x > 0), sum x,
 * sum x^2 (float64, DEVICE).  ru_case_prepare: batch [K][C][padded] = the K test-time flips (3 bits per copy, bit0 D, bit1 H, bit2 W,
 * as ru_tta_merge) of the crop zero-padded by pad_left to `padded` and z-scored with the moments in `stats` -- every voxel,
 * padding included, (x - mean) / std in float64 as the reference's numpy does (test.py:103-113).  lo / size / pad_left / padded: HOST.  */
int ru_case_bbox(const float* image, int* box, int C, int D, int H, int W, ru_stream_t stream);
size_t ru_case_workspace_bytes(int C, int D, int H, int W);
int ru_case_stats(const float* image, double* stats, int C, int D, int H, int W, const int* lo, const int* size,
                  void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_case_prepare(const float* image, const double* stats, float* batch, int C, int D, int H, int W, const int* lo, const int* size,
                    const int* pad_left, const int* padded, int K, unsigned flips, ru_stream_t stream);
/* ru_tta_merge restricted to the box [lo, lo + size) of the padded prediction (test.py:140-144 removes the padding before the masks are
 * counted): mean_out / mask are [C][size], counts[C] the voxels set inside the box.  */
int ru_tta_merge_box(const float* probs, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts,
                     int C, int D, int H, int W, const int* lo, const int* size, ru_stream_t stream);
/* Post-processing (test.py:51-62,162-164): 26-connected components of labels > 0 by union-find on the device, every component smaller
 * than ratio * (V - size of the most frequent label, background included) is zeroed in place.  Only component sizes enter the rule, so
 * the result equals skimage.morphology.label + reject_small_regions whatever the numbering.  ru_paste_labels (test.py:167-168):
 * full [D][H][W] = 0 outside the box, lab [size] inside.  */
size_t ru_cc_workspace_bytes(int D, int H, int W);
int ru_cc_reject(unsigned char* labels, int D, int H, int W, double ratio, void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_paste_labels(const unsigned char* lab, unsigned char* full, int D, int H, int W, const int* lo, const int* size, ru_stream_t stream);

/* ---------------------------------------------------------------- voxel-major working layout ("C16")
 * Between the first and the last convolution the split-bf16 engine keeps activations as [N][C/16][D][H][W][16]
 * (16 channels of a voxel contiguous; C % 16 == 0): a halo tile of a 3x3x3 convolution is then a few long contiguous
 * runs instead of 16 channel planes x short rows.  Nothing in that layout crosses the boundary of ru_unet_*; the
 * entry points below exist so the tests and probes can drive the layout-aware kernels one at a time.
 * flags: bit 0 = x (input) is C16, bit 1 = y (output) is C16, bit 2 = x is NCDHW with Cin <= 4 and goes through the 4-channel
 * tap-pair kernel (needs extra workspace: 16 bytes per input voxel); bit 3 = x is voxel-major in SPLIT form (hi / lo bf16 packets); bit 4 = exact-f32
 * arithmetic (v_mfma_f32_16x16x4_f32 on voxel-major tensors: the exact-f32 inference forward of the engine; not with bits 2 / 3); bit 5 = x is an
 * ACTIVATION tensor (a forward convolution, model.py:72-73): shapes that have the kernel (16 input channels, voxel-major both sides, a grid that fills the
 * chip) take the fp16 + MX-fp8 product scheme the engine's forward convolutions take (f16*f16 + two e4m3 cross terms, RU_MX=0: off); never set for gradients;
 * bit 6 = x (float32, voxel-major both sides, 16 -> 16 channels) is a GRADIENT: where the kernel takes the shape it is converted to the gradient-operand form (bf16 + two
 * e4m3 planes + one exponent byte per voxel) and convolved with bf16 main + MX-fp8 cross products, as the engine's 16-channel data-gradient convolutions are (RU_MXG=0:
 * off; needs 64 more workspace bytes per input voxel); ignored elsewhere.  k = 3. */
int ru_layout_convert(const float* src, float* dst, int N, int C, size_t V, int to_c16, ru_stream_t stream);
int ru_conv3d_fwd_l(const float* x, const float* w, const float* bias, float* y,
                    int N, int Cin, int Cout, int D, int H, int W, int flags,
                    void* ws, size_t ws_bytes, ru_stream_t stream);
/* weight gradient; flags: bit 0 = x is C16, bit 1 = dy is C16 */
int ru_conv3d_bwd_weight_l(const float* x, const float* dy, float* dw,
                           int N, int Cin, int Cout, int D, int H, int W, int flags,
                           void* ws, size_t ws_bytes, ru_stream_t stream);

/* trilinear x2 and its transpose on C16 tensors (C % 16 == 0; D,H,W = extents of the COARSE side, as in
 * ru_upsample2x_trilinear_*).  out_slope: LeakyReLU slope applied to the interpolated value (the decoder fuses model.py:401-402
 * into the up-sampling; 1 = none). */
int ru_upsample2x_trilinear_fwd_l(const float* x, float* y, int N, int C, int D, int H, int W, float out_slope, ru_stream_t stream);
int ru_upsample2x_trilinear_bwd_l(const float* dy, float* dx, int N, int C, int D, int H, int W, ru_stream_t stream);

/* ---------------------------------------------------------------- evaluation metric (metrics.py:108-133, `Dice.update`)
 * counts[(n*C + c)*2 + {0,1}] = { #(p > 0.5 and g > 0.5), #(p > 0.5) + #(g > 0.5) } over the V voxels of sample n, channel c.
 * The metric is 2*counts[0]/counts[1] per (n, c) (NaN -> 1), averaged over the batch (host side: brats2019_amd/metrics.py). */
int ru_dice_counts(const float* p, const float* g, unsigned long long* counts, int N, int C, size_t V, ru_stream_t stream);
/* The rest of `Dice.update` (metrics.py:124-130) on the device: acc[c] += mean over the N samples of r(n, c), r = 2*counts[0]/counts[1] formed
 * in float32 like the reference's numpy line (0/0 -> NaN -> 1), the mean and the accumulator in float64; c < nacc (= classes - 1 <= C). */
int ru_dice_accumulate(const unsigned long long* counts, double* acc, int N, int C, int nacc, ru_stream_t stream);

/* ---------------------------------------------------------------- training input pipeline (dataloader.py:100-216, SimpleReader)
 * ru_zscore_stats: per channel stats[c] = { #(x > 0), sum x, sum x^2 } over all V voxels in float64 -- the three numbers the
 *   reference's normalisation is made of (dataloader.py:124-130: the count is over positive voxels, the sums over all).
 * ru_augment_patch: ONE pass from the resident raw case to a training patch (dataloader.py:147-205): crop [lo, lo + patch),
 *   z-score ((x - mean) * inv_std), zoom by `scale` per axis (scipy.ndimage.affine_transform with a diagonal matrix, order 1,
 *   mode 'reflect', applied to the modalities and to the one-hot label), flips (flags bit 0/1/2 = axes D/H/W), transpose of D
 *   and H (bit 3), per-channel gain and bias, WT/TC/ET soft targets.  data_out [C][Q0][Q1][P2], target_out [3][Q0][Q1][P2] with
 *   (Q0, Q1) = (P1, P0) when transposed.  The small parameter arrays are HOST pointers, read at launch.  C <= 8. */
size_t ru_zscore_workspace_bytes(int C, size_t V);
int ru_zscore_stats(const float* image, double* stats, int C, size_t V, void* ws, size_t ws_bytes, ru_stream_t stream);
int ru_augment_patch(const float* image, const unsigned char* label, const float* mean, const float* inv_std,
                     int C, int D, int H, int W, const int* crop_lo, const int* patch, const double* scale, int flags,
                     const float* gain, const float* bias, float* data_out, float* target_out, ru_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RESUNET_HIP_H_ */
