/*
 * dynmm_hip.h — C ABI of libdynmm_hip.so: the MI355X (gfx950) kernels behind the fusion-level DynMM
 * hot path (dual ResNet-34 RGB+depth encoders, gated SE fusion, global gate, PPM, ESANet decoder).
 *
 * The reference (zihuixue/DynMM, FusionDynMM/src) has no FFI: its device work is issued through
 * ATen.  Each entry point below therefore cites the *reference operation* (file:line under
 * /root/reference/FusionDynMM) whose ATen call sequence it replaces; INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - all tensors are fp32, NCHW, contiguous, device pointers owned by the caller (PyTorch's
 *     caching allocator); the library allocates nothing and keeps no state;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising, so calls are capturable in a hipGraph;
 *   - return value: 0 on success, a negative DYNMM_E* code on bad arguments, or -(1000+hipError_t)
 *     if a launch failed.  Nothing throws across the boundary;
 *   - reduction outputs ("double* sums", dbias, dw of the depthwise conv) are zeroed by the callee
 *     (hipMemsetAsync on `stream`) before accumulation; SE / gate parameter gradients are plain stores.
 */
#ifndef DYNMM_HIP_H
#define DYNMM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYNMM_OK 0
#define DYNMM_EINVAL (-1)      /* inconsistent dimensions / null pointer            */
#define DYNMM_EUNSUPPORTED (-2) /* shape outside what the kernels implement        */
#define DYNMM_EWORKSPACE (-3)   /* workspace too small (see *_workspace_bytes)      */

#define DYNMM_ACT_NONE 0
#define DYNMM_ACT_RELU 1
#define DYNMM_ACT_TANH 2

/* ABI version + build info (smoke / loader check). */
int dynmm_abi_version(void);
const char* dynmm_build_info(void);

/* Test hook (tests/test_skip_esanet.py measures the fp32 conditioning of a gate gradient by running one pass under two correct
 * summation orders): which implicit-GEMM generation serves the stride-1 same-padded 1x1 / 3x1 / 1x3 / 3x3 convolutions that
 * do not take the Winograd kernels (forward + input gradient).  mode 0 = the register-staged kernels (csrc/conv_igemm.hip),
 * 1 or -1 (default) = the operand-ring kernels (csrc/conv_igemm_v5.hip).  Process-wide; not meant to be flipped while
 * launches are being issued from other threads. */
int dynmm_debug_set_igemm_v5(int mode);

/* Geometry of one convolution, shared by fwd / dgrad / wgrad.
 * x:[N,Ci,H,W]  w:[Co,Ci,KH,KW]  y:[N,Co,Ho,Wo], Ho = (H+2PH-KH)/SH+1 (same for W). groups = 1.
 * If x2 != NULL the logical input is cat([x, x2], dim=1) with x holding the first `c_split`
 * channels (GlobalGate's torch.concat, src/models/model_skip_mod_globalgate.py:389). */
typedef struct {
    int N, Ci, H, W;
    int Co, Ho, Wo;
    int KH, KW, SH, SW, PH, PW;
    int c_split; /* == Ci when x2 is NULL */
} dynmm_conv_geom;

/* 1 if dynmm_conv2d_fwd (dgrad = 0) / dynmm_conv2d_dgrad (dgrad = 1) serve this geometry with the operand-ring kernels
 * (csrc/conv_igemm_v5.hip; 16-byte aligned operands assumed) — for tools that label launches (bench.py's roofline leg). */
int dynmm_conv2d_uses_operand_ring(const dynmm_conv_geom* g, int dgrad);

/* Re-layout of a conv weight for the implicit-GEMM kernels (done per step; weights are small).
 *   wp_fwd  [(tap*CiR+ci)][CoP]  (tap = r*KW+s)   — operand of dynmm_conv2d_fwd
 *   wp_dgrad[(tap*CoR+co)][CiP]                    — operand of dynmm_conv2d_dgrad
 * CoP / CiP = Co / Ci rounded up to a multiple of 4 (rows stay 16-byte aligned for dwordx4 loads); CiR / CoR =
 * rows per filter tap: the channel count, rounded up to a multiple of 16 when it is >= 8 and not one already
 * (zero rows: e.g. the input gradient of the 40-class convs then runs the one-tap-per-K-step fast path).
 * All padding is written as zeros.  Buffer sizes in floats: dynmm_packed_weight_floats(..., dgrad = 0 / 1).
 * Either output may be NULL. */
size_t dynmm_packed_weight_floats(int Co, int Ci, int KH, int KW, int dgrad);
int dynmm_pack_weight(const float* w, float* wp_fwd, float* wp_dgrad,
                      int Co, int Ci, int KH, int KW, void* stream);

/* dynmm_pack_weight for many weights in one launch.  desc (device memory) = ndesc records of 5 int64 words:
 *   { src, dst_fwd, dst_dgrad (-1: none) : float offsets from src_base / dst_base ;
 *     Co | Ci << 32 ;  KH*KW | first_workgroup << 32 }
 * record d owns the workgroups [first_workgroup_d, first_workgroup_{d+1}), dynmm_pack_weight_multi_blocks(..) of them
 * (32 x 32-channel tiles transposed through LDS for filters of up to 9 taps, 256 output elements per workgroup
 * otherwise); total_blocks = their sum. */
int dynmm_pack_weight_multi_blocks(int Co, int Ci, int KH, int KW, int dgrad);
int dynmm_pack_weight_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                            void* stream);

/* y = act( conv(x|x2, w) * scale[co] + shift[co] + residual )      (scale/shift/residual optional)
 * Replaces nn.Conv2d (+ folded eval BatchNorm2d + ReLU + residual add) of
 *   ResNet stem src/models/resnet.py:352-358, BasicBlock :66-84, NonBottleneck1D :124-147,
 *   ConvBNAct src/models/model_utils.py:11-23, Decoder.conv_out / side_output src/models/model.py:286,339,
 *   GlobalGate convs src/models/model_skip_mod_globalgate.py:380-386.
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM; bias is passed as shift with scale = NULL. */
int dynmm_conv2d_fwd(const float* x, const float* x2, const float* wp_fwd,
                     const float* scale, const float* shift, const float* residual,
                     float* y, const dynmm_conv_geom* g, int act, void* stream);

/* The ResNet stem convolution (7x7, stride 2, padding 3, Ci 1 | 3 -> 64: src/models/resnet.py:216-217 conv1) feeding a
 * training-mode BatchNorm (resnet.py:229 bn1): y = conv(x) + bias as dynmm_conv2d_fwd, and the per-channel sums of y and y^2 over
 * (N, H, W) ADDED to stats [2][Co] (fp64, zeroed by the caller) from the kernel's epilogue — the `sums` operand of
 * dynmm_bn_finalize / dynmm_bn_apply without a dynmm_bn_stats launch and its pass over y (629 MB per stem at batch 32). */
int dynmm_conv2d_stem_fwd_stats_supported(const dynmm_conv_geom* g);
int dynmm_conv2d_stem_fwd_stats(const float* x, const float* wp_fwd, const float* bias, float* y, double* stats,
                                const dynmm_conv_geom* g, void* stream);

/* dx = conv_transpose(dy, w) * [mask > 0] + accum   (mask, accum optional, shaped like x):
 * the autograd "input gradient" of the conv above, with the ReLU backward of the producer of x and
 * the gradient arriving over a residual branch fused into the epilogue (saves the separate
 * threshold_backward and add passes of the reference's autograd).
 * dx covers cat([x,x2]) when g->c_split < g->Ci: dx2 receives channels [c_split, Ci). */
int dynmm_conv2d_dgrad(const float* dy, const float* wp_dgrad, const float* mask, const float* accum,
                       float* dx, float* dx2, const dynmm_conv_geom* g, void* stream);

/* ---- three-tap convolutions by 1-D Winograd F(2,3) on the fp32 matrix cores (csrc/conv_wino.hip) ----
 * Stride-1, same-padded 1x3 / 3x1 convolutions with W % 4 == 0 whose GEMM has rows % 64 == 0 (forward: % 8, >= 24) and a
 * reduction of a multiple of 8 (>= 24) channels — (rows, reduction) = (Co, Ci) forward, (Ci, Co) input gradient — (the factorised
 * convolutions of resnet.py:124-147; 3x3 filters: the 2-D form below): four channel contractions per output PAIR
 * along the tap axis instead of six, i.e. 2/3 of the direct convolution's matrix work, in fp32 (error vs fp64 of the same
 * class as a direct fp32 sum: 2e-7 .. 4e-7 rms).  dynmm_conv2d_wino_supported(g, dgrad) = 1 when that pass qualifies;
 * = 2 (dgrad only) for the STRIDE-2 three-tap convolutions — 3x1 stride (2,1) / 1x3 stride (1,2), resnet.py:104-107 in the first block
 * of stages 2-4 — whose input gradient runs on the same pair kernel in polyphase form (dx[2j] = W1^T dy[j], dx[2j+1] = W2^T dy[j] +
 * W0^T dy[j+1]: the direct operation count, 8-byte stores); their operand is dynmm_wino_pack(..., dgrad = 2) = (W1, W2, W0, 0).
 * Operand: the filter transforms ut[K][C rounded up to 64][4] ((K, C) = (Ci, Co) forward, (Co, Ci) input gradient),
 * dynmm_wino_packed_floats floats (0 for any other filter shape), 16-byte aligned, written by dynmm_wino_pack or — many filters in ONE launch —
 * by dynmm_wino_pack_multi: desc (device memory) = ndesc records of 4 int64 words { src, dst : float offsets from
 * src_base / dst_base (dst % 4 == 0) ; Co | Ci << 32 ; KH | KW << 8 | dgrad (0, 1, 2) << 16 | first_workgroup << 32 }, first
 * workgroups being the running sum of dynmm_wino_pack_multi_blocks.
 *   fwd  : y = act(conv(x, w) + bias + residual)                  (bias, residual optional; an eval-mode BatchNorm folds
 *          into `scale` at pack time and `bias`: model_utils.py:11-23 conv -> BN -> act as one kernel)
 *   dgrad: dx = conv_transpose(dy, w) * [mask > 0] + accum        (mask, accum optional; the epilogue of dynmm_conv2d_dgrad)
 * x / dy / ut 16-byte aligned, y / dx / residual / mask / accum 8-byte aligned, else DYNMM_EUNSUPPORTED. */
int dynmm_conv2d_wino_supported(const dynmm_conv_geom* g, int dgrad);
size_t dynmm_wino_packed_floats(int Co, int Ci, int KH, int KW);
int dynmm_wino_pack(const float* w, float* ut, const float* scale /* [Co] or NULL, forward only: ut = transform(scale[co] * w) */,
                    int Co, int Ci, int KH, int KW, int dgrad, void* stream);
int dynmm_wino_pack_multi_blocks(int Co, int Ci, int KH, int KW);
int dynmm_wino_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                          void* stream);
int dynmm_conv2d_wino_fwd(const float* x, const float* ut, const float* bias, const float* residual, float* y,
                          const dynmm_conv_geom* g, int act, void* stream);
/* The input gradient of a 3x1 convolution whose input is z = relu(BN(c)) (resnet.py:131-135 conv3x1_2 after bn1 + ReLU), together
 * with that BatchNorm's backward reductions: dx = the convolution's input gradient masked by [BN(c) > 0] (re-derived from c = bn_x
 * with bn_apply's own fma), and sums[s][0][ch] += sum dx, sums[s][1][ch] += sum dx * xhat over (N, H, W) — the `sums` operand of
 * dynmm_bn_bwd_apply (fp64 [slots][2][Ci], zeroed by the caller; pixel tile p adds into slab p % slots, slots =
 * dynmm_conv2d_wino_dgrad_bnred_slots(g) = dynmm_bn_bwd_apply's `training`) without a dynmm_bn_bwd_reduce launch. */
int dynmm_conv2d_wino_dgrad_bnred_supported(const dynmm_conv_geom* g);
int dynmm_conv2d_wino_dgrad_bnred_slots(const dynmm_conv_geom* g);
int dynmm_conv2d_wino_dgrad_bnred(const float* dy, const float* ut, const float* bn_x, const float* bn_mean,
                                  const float* bn_invstd, const float* bn_gamma, const float* bn_beta, double* sums, float* dx,
                                  const dynmm_conv_geom* g, void* stream);
/* The same for a BatchNorm + identity + ReLU (resnet.py:136-147: out = relu(bn2(c) + identity)) whose output feeds the next block and
 * nothing else: this launch is that block's first 3x1 convolution, `accum` the gradient of its identity branch (or NULL), so the
 * epilogue holds the complete gradient of `out`.  dx = (conv_transpose(dy, w) + accum) * [out > 0] with the decisions read from
 * relu_bits (the record dynmm_bn_apply left: dynmm_bn_relu_bits_words), sums as above with xhat = (bn_x - mean) * invstd.  The
 * BatchNorm's backward continues with dynmm_bn_bwd_apply(act = NONE) on dx, which is also the identity branch's gradient.
 * Geometry: dynmm_conv2d_wino_dgrad_bnred_supported and H * W % 4 == 0. */
int dynmm_conv2d_wino_dgrad_bnred2(const float* dy, const float* ut, const float* accum, const float* bn_x,
                                   const unsigned long long* relu_bits, const float* bn_mean, const float* bn_invstd,
                                   double* sums, float* dx, const dynmm_conv_geom* g, void* stream);
/* The forward of a convolution that feeds a training-mode BatchNorm (resnet.py:110,118 `bn1` / `bn2` after conv1x3_*;
 * model_utils.py:11-23 ConvBNAct), 1x3 taps (3x3: dynmm_conv2d_wino2d_fwd's stats), Co % 64 == 0, no activation: y as dynmm_conv2d_wino_fwd, and the
 * per-channel sums of y and y^2 over (N, H, W) ADDED to stats [nslots][2][Co] (fp64, zeroed by the caller; pixel tile p adds into
 * slab p % nslots, nslots = dynmm_conv2d_wino_fwd_stats_slots(g): thousands of tiles on one address serialise) from the kernel's
 * epilogue — the `sums` operand of dynmm_bn_apply (training = nslots) without a dynmm_bn_stats launch. */
int dynmm_conv2d_wino_fwd_stats_supported(const dynmm_conv_geom* g);
int dynmm_conv2d_wino_fwd_stats_slots(const dynmm_conv_geom* g);
int dynmm_conv2d_wino_fwd_stats(const float* x, const float* ut, const float* bias, float* y, double* stats, int nslots,
                                const dynmm_conv_geom* g, void* stream);
int dynmm_conv2d_wino_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                            const dynmm_conv_geom* g, void* stream);

/* ---- 3x3 convolutions by 2-D Winograd F(2x2, 3x3) (csrc/conv_wino2d.hip, v_mfma_f32_16x16x4_f32): sixteen channel contractions per
 * 2x2 output tile — 4/9 of the direct convolution's matrix work, plain fp32, 1e-7-level error (src/models/resnet.py:66-84 BasicBlock,
 * model.py:286,343-357 conv_out / decoder conv3x3).  Stride 1, padding 1, W % 4 == 0, N*H*W >= 256; forward: Ci % 4 == 0 >= 12,
 * Co % 8 == 0 >= 24; input gradient: Ci % 64 == 0, Co % 4 == 0 >= 12 (dynmm_conv2d_wino2d_supported).  Operand ut
 * (dynmm_wino2d_packed_floats floats, 16-byte aligned): U = G g G^T as [K][4][rows rounded up to 64][4], written by dynmm_wino2d_pack
 * (scale: an eval-mode BatchNorm's per-channel factor folded into the forward operand; dgrad = 1: the flipped filter with the channel
 * roles swapped) or, for many filters at once, dynmm_wino2d_pack_multi (descriptors {src, dst: float offsets; Co | Ci << 32;
 * dgrad << 16 | first workgroup << 32}).
 * fwd:   y = act(conv(x) + bias + residual); stats != NULL (act none, no residual, Co % 64 == 0): the per-channel sums of y and y^2 are
 *        ADDED to stats [nslots][2][Co] (fp64, zeroed by the caller; nslots = dynmm_conv2d_wino2d_stats_slots) — the `sums` operand of
 *        dynmm_bn_apply without a dynmm_bn_stats launch.
 * dgrad: dx = [mask > 0] . conv_transpose(dy) + accum   (mask / accum like dx, optional). */
int dynmm_conv2d_wino2d_supported(const dynmm_conv_geom* g, int dgrad);
size_t dynmm_wino2d_packed_floats(int Co, int Ci);
int dynmm_wino2d_pack(const float* w, float* ut, const float* scale, int Co, int Ci, int dgrad, void* stream);
int dynmm_wino2d_pack_multi_blocks(int Co, int Ci, int dgrad);
int dynmm_wino2d_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks, void* stream);
int dynmm_conv2d_wino2d_stats_slots(const dynmm_conv_geom* g);
int dynmm_conv2d_wino2d_fwd(const float* x, const float* ut, const float* bias, const float* residual, float* y, double* stats,
                            int nslots, const dynmm_conv_geom* g, int act, void* stream);
int dynmm_conv2d_wino2d_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                              const dynmm_conv_geom* g, void* stream);

/* ---- input gradients by 1-D Winograd F(4,3) (csrc/conv_wino43.hip): four neighbouring outputs of a three-tap filter from six
 * multiplications — HALF of the direct convolution's matrix work (F(2,3) above: 2/3).  The transforms carry factors up to 8 and
 * 1/24: 1.7e-6 .. 2.8e-6 from fp64 in max-norm (direct fp32: 2e-7 .. 3e-7), which is why only the BACKWARD uses it.
 * The stride-1 1x3 convolutions with Ci % 64 == 0, Co % 8 == 0 >= 24, N*H*W >= 256 (dynmm_conv2d_wino43_supported); every tensor
 * 16-byte aligned.  Operand ut (dynmm_wino43_packed_floats floats): U0..U3 [Co][Ci][4] followed by U4, U5 [Co][Ci][2],
 * written by dynmm_wino43_pack or, for many filters in ONE launch, dynmm_wino43_pack_multi (descriptor as dynmm_wino_pack_multi's,
 * without the dgrad bit).
 *   dx = conv_transpose(dy, w) * [mask > 0] + accum */
int dynmm_conv2d_wino43_supported(const dynmm_conv_geom* g);
size_t dynmm_wino43_packed_floats(int Co, int Ci, int KH, int KW);
int dynmm_wino43_pack(const float* w, float* ut, int Co, int Ci, int KH, int KW, void* stream);
int dynmm_wino43_pack_multi_blocks(int Co, int Ci, int KH, int KW);
int dynmm_wino43_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                            void* stream);
int dynmm_conv2d_wino43_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                              const dynmm_conv_geom* g, void* stream);

/* dw[Co,Ci,KH,KW] = sum_{n,oh,ow} dy * x(window).  Split over the pixel range into partial slabs in
 * `workspace` (>= dynmm_conv2d_wgrad_workspace_bytes), reduced deterministically.
 * dbias (optional, [Co]) = sum_{n,oh,ow} dy: the bias gradient of the same conv, produced from the dy tiles
 * the kernel stages anyway (no separate pass over dy). */
size_t dynmm_conv2d_wgrad_workspace_bytes(const dynmm_conv_geom* g);
int dynmm_conv2d_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias,
                       void* workspace, size_t workspace_bytes,
                       const dynmm_conv_geom* g, void* stream);

/* Weight gradients of n <= 8 convolutions of IDENTICAL geometry (e.g. the factorised convs of consecutive residual
 * blocks) in one launch: one residency round whatever n is, so every workgroup walks an n-times longer pixel range and
 * the per-launch fixed costs and the split-K slab traffic are paid once per group.  xs / dys / dws / dbiases: host arrays
 * of n device pointers (dbiases NULL, or all entries set / all NULL).  Falls back to n ordinary launches when the
 * geometry is not groupable (dynmm_conv2d_wgrad_groupable: the vectorised 128x128 kernel, single input). */
int dynmm_conv2d_wgrad_groupable(const dynmm_conv_geom* g);
/* Which weight-gradient kernel serves this geometry when the tensors are 16-byte aligned (launch labels of bench.py):
 * 6 = the three-tap kernel (conv_wgrad_v6.hip), 4 = the vectorised 128x128 kernel, 0 = the generic tiles / stem kernel. */
int dynmm_conv2d_wgrad_variant(const dynmm_conv_geom* g);
size_t dynmm_conv2d_wgrad_group_workspace_bytes(const dynmm_conv_geom* g, int n);
int dynmm_conv2d_wgrad_group(int n, const float* const* xs, const float* const* dys, float* const* dws,
                             float* const* dbiases, void* workspace, size_t workspace_bytes,
                             const dynmm_conv_geom* g, void* stream);

/* g_out = g * act'(y) ; dbias[c] = sum_{n,hw} g_out   (either output may be NULL).
 * ReLU/tanh backward + bias gradient of a conv+bias+act (autograd of resnet.py:125-126 etc.).
 * dbias is summed over sample splits through `workspace` (>= *_workspace_bytes) in a fixed order. */
size_t dynmm_act_bwd_bias_workspace_bytes(int N, int C);
int dynmm_act_bwd_bias(const float* g, const float* y, float* g_out, float* dbias, float* workspace,
                       int N, int C, int HW, int act, void* stream);

/* ---- BatchNorm2d (src/models/resnet.py:59,110; model_utils.py:22; …globalgate.py:381,384) ---- */
/* per-channel sum / sum of squares over (N,HW) into sums[2*C] (double).  sums_are_zero != 0: the caller
 * hands in a buffer that is already zero (e.g. a slice of an arena cleared once per step) and the memset
 * launch is skipped. */
int dynmm_bn_stats(const float* x, double* sums, int N, int C, int HW, int sums_are_zero, void* stream);
/* y = act( (x-mean)*invstd*gamma + beta + residual ).
 * training=k>=1: mean/var from `sums` [k][2][C] — k slabs added in slab order (1: dynmm_bn_stats; the slots of
 *             dynmm_conv2d_wino_fwd_stats) — (biased var for normalisation); writes save_mean/save_invstd[C],
 *             updates running_mean/var with `momentum` (unbiased var), as F.batch_norm does, and
 *             increments *num_batches_tracked (int64, optional) as nn.BatchNorm2d does.
 * training=0: uses running_mean/var; sums / save_* may be NULL.
 * relu_bits (optional; act = ReLU, HW % 4 == 0, 16-byte aligned tensors): the ReLU decisions [y > 0], ONE BIT per element
 *             (dynmm_bn_relu_bits_words(N, C, HW) 64-bit words: per plane and group of 256 consecutive elements four ballot
 *             words, word j bit l = element 4 l + j).  The two backward passes read them in place of y — with a residual
 *             (resnet.py:136-147: bn2 + identity + ReLU) the mask cannot be re-derived from x, and y is 4 bytes per element
 *             in each pass. */
size_t dynmm_bn_relu_bits_words(int N, int C, int HW);
int dynmm_bn_apply(const float* x, const double* sums, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                   const float* residual, float* y, long long* num_batches_tracked, int N, int C, int HW,
                   float eps, float momentum, int training, int act, unsigned long long* relu_bits, void* stream);
/* The per-channel part of dynmm_bn_apply alone (training mode): mean / invstd, running statistics (+ step counter),
 * and scale = gamma*invstd, shift = beta - mean*scale for consumers that normalise on load. */
int dynmm_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* save_mean, float* save_invstd, long long* num_batches_tracked,
                      float* scale, float* shift, int N, int C, int HW, float eps, float momentum, void* stream);
/* backward: sums[2*C] <- (sum g_eff, sum g_eff*xhat), g_eff = g*act'(y).
 * y may be NULL for act = ReLU when the forward had no residual: the mask [y > 0] is then re-derived from
 * x as fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0 — bit-identical to the forward's own evaluation
 * (gamma, beta required in that case) — which saves one tensor read in each of the two backward passes.
 * relu_bits (optional): the forward's ReLU decisions as written by dynmm_bn_apply; y is then not read (may be NULL). */
int dynmm_bn_bwd_reduce(const float* g, const float* y, const float* x,
                        const float* mean, const float* invstd, const float* gamma, const float* beta,
                        double* sums, int N, int C, int HW, int act, int sums_are_zero,
                        const unsigned long long* relu_bits, void* stream);
/* dx = gamma*invstd*(g_eff - sum_g/M - xhat*sum_gx/M) (training) or gamma*invstd*g_eff (eval);
 * d_residual = g_eff (optional); dgamma/dbeta from sums.  training = 0 (eval) or the number n >= 1 of slabs the sums arrive in,
 * sums[n][2][C] (1 from dynmm_bn_bwd_reduce; dynmm_conv2d_wino_dgrad_bnred_slots from a convolution's epilogue) — the same
 * convention as dynmm_bn_apply's. */
int dynmm_bn_bwd_apply(const float* g, const float* y, const float* x,
                       const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const double* sums, float* dx, float* d_residual,
                       float* dgamma, float* dbeta,
                       int N, int C, int HW, int training, int act, const unsigned long long* relu_bits, void* stream);
/* eval-mode folding for the fused conv epilogue: scale = gamma*rsqrt(var+eps),
 * shift = beta + (conv_bias - mean)*scale   (conv_bias optional). */
int dynmm_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, const float* conv_bias, float* scale, float* shift,
                  int C, float eps, void* stream);

/* ---- pooling / resampling ---- */
/* F.max_pool2d(k=3,s=2,p=1) (…globalgate.py:260-261). idx (int8, optional) = argmax tap 0..8. */
int dynmm_maxpool3x3s2_fwd(const float* x, float* y, signed char* idx,
                           int N, int C, int H, int W, int Ho, int Wo, void* stream);
int dynmm_maxpool3x3s2_bwd(const float* g, const signed char* idx, float* dx,
                           int N, int C, int H, int W, int Ho, int Wo, void* stream);
/* F.adaptive_avg_pool2d (context_modules.py:56; also GAP when OH=OW=1). */
int dynmm_adaptive_avgpool_fwd(const float* x, float* y, int NC, int H, int W, int OH, int OW, void* stream);
int dynmm_adaptive_avgpool_bwd(const float* g, float* dx, int NC, int H, int W, int OH, int OW, void* stream);
/* out[:, c_off:c_off+C] = nearest_resize(y, (H,W))  — the cat+interpolate of PyramidPoolingModule
 * (context_modules.py:72-86).  out has Ctot channels. bwd sums the gradient back. */
int dynmm_nearest_into_fwd(const float* y, float* out, int N, int C, int h, int w,
                           int Ctot, int c_off, int H, int W, void* stream);
int dynmm_nearest_into_bwd(const float* g_out, float* dy, int N, int C, int h, int w,
                           int Ctot, int c_off, int H, int W, void* stream);
/* Upsample 'learned-3x3-zeropad' (src/models/model.py:404-410) = nearest x2 + depthwise 3x3 + bias,
 * plus the decoder's `out += encoder_features` (model.py:354-355) when skip != NULL. */
int dynmm_upsample2x_dw3x3_fwd(const float* x, const float* w, const float* b, const float* skip,
                               float* y, int N, int C, int H, int W, void* stream);
size_t dynmm_upsample2x_dw3x3_bwd_workspace_bytes(int N, int C);
int dynmm_upsample2x_dw3x3_bwd(const float* g, const float* x, const float* w,
                               float* dx, float* dw, float* db, float* workspace,
                               int N, int C, int H, int W, void* stream);

/* ---- SE fusion + gated blend (rgb_depth_fusion.py:22-26, model_utils.py:47-51, …globalgate.py:282-310) ---- */
/* s[n,c] = mean_hw x[n,c,:]  for two tensors at once. */
int dynmm_gap2_fwd(const float* xr, const float* xd, float* sr, float* sd, int NC, int HW, void* stream);
/* Per-sample SE MLPs and blend coefficients:
 *   g_m = sigmoid(W2_m relu(W1_m s_m + b1_m) + b2_m)  (m in {rgb, depth});  use_se=0 -> g = 1
 *   a = wc + (1-wc) g_r ;  b = (1-wc) g_d           (wc[n] optional, NULL -> 0)
 * so that  out = a*rgb + b*depth  ==  wc*rgb + (1-wc)*(rgb*g_r + depth*g_d).
 * params: 8 pointers {W1r,b1r,W2r,b2r,W1d,b1d,W2d,b2d}; hidden = C/16.
 * saves h_r,h_d [N,C/16] and g_r,g_d [N,C] for the backward. */
int dynmm_se_coeff_fwd(const float* sr, const float* sd, const float* const* params,
                       const float* wc, int wc_stride, float* a, float* b,
                       float* hr, float* hd, float* gr, float* gd,
                       int N, int C, int use_se, void* stream);
/* backward: per-sample pass (pre-activation gradients of both MLPs into `workspace`,
 * >= dynmm_se_coeff_bwd_workspace_bytes(N, C); pooled gradients dsr/dsd; dwc) followed by the parameter
 * gradients summed over the samples in a fixed order — no float atomics, bit-reproducible. */
size_t dynmm_se_coeff_bwd_workspace_bytes(int N, int C);
int dynmm_se_coeff_bwd(const float* da, const float* db, const float* sr, const float* sd,
                       const float* const* params, const float* wc, int wc_stride,
                       const float* hr, const float* hd, const float* gr, const float* gd,
                       float* const* dparams, float* dsr, float* dsd, float* dwc, int dwc_stride,
                       float* workspace, int N, int C, int use_se, void* stream);
/* out = a[n,c]*xr + b[n,c]*xd */
int dynmm_axpby_fwd(const float* xr, const float* xd, const float* a, const float* b, float* out,
                    int NC, int HW, void* stream);
/* da[n,c] = sum_hw g*xr ; db[n,c] = sum_hw g*xd */
int dynmm_axpby_bwd_reduce(const float* g, const float* xr, const float* xd, float* da, float* db,
                           int NC, int HW, void* stream);
/* dxr = a*g + cscale*ca ; dxd = b*g + cscale*cb   (ca,cb [N,C] optional: the GAP backward terms
 * ds of the SE squeeze, cscale = 1/HW) */
int dynmm_axpby_bwd_apply(const float* g, const float* a, const float* b,
                          const float* ca, const float* cb, float cscale, float* dxr, float* dxd,
                          int NC, int HW, void* stream);

/* Stem fusion + both 3x3/s2/p1 max-pools in one forward and two backward passes (…globalgate.py:258-261):
 *   fuse = a*rgb + b*depth (never written);  y_out = max_pool(fuse);  y_depth = max_pool(depth);  idx_* = arg-max codes.
 *   bwd_reduce: da/db [NC] = sum d(fuse)*rgb / *depth with d(fuse) = max_pool_backward(g_out, idx_out) on the fly;
 *   bwd_apply : dxr = a*d(fuse) + ca*cscale;  dxd = b*d(fuse) + cb*cscale + max_pool_backward(g_depth, idx_depth).
 * Even H and W % 8 == 0 only (dynmm_axpby_pool_supported); DYNMM_EUNSUPPORTED otherwise (callers keep the unfused ops). */
int dynmm_axpby_pool_supported(int H, int W);
int dynmm_axpby_pool_fwd(const float* xr, const float* xd, const float* a, const float* b, float* y_out,
                         signed char* idx_out, float* y_depth, signed char* idx_depth, const float* bn_tr, int C,
                         int NC, int H, int W, void* stream);
int dynmm_axpby_pool_bwd_reduce(const float* g_out, const signed char* idx_out, const float* xr, const float* xd,
                                float* da, float* db, const float* bn_tr, int C, int NC, int H, int W, void* stream);
/* bn_tr (optional, [4][C] = scale_r, shift_r, scale_d, shift_d from dynmm_bn_finalize): xr / xd are the stem conv
 * outputs BEFORE their BatchNorm + ReLU (resnet.py:229-231), which the kernels apply on load — the normalised
 * 629 MB tensors are then never written either.  dynmm_gap2_bnrelu_fwd: the SE squeeze of the same virtual tensors. */
int dynmm_gap2_bnrelu_fwd(const float* xr, const float* xd, const float* bn_tr, int C, float* sr, float* sd, int NC,
                          int HW, void* stream);
/* BatchNorm(+ReLU) backward of a stem whose output gradient is never written (the training backward of the chain above):
 *   gy = coef[n,c] * max_pool_backward(g_out, idx_out) + off[n,c]*cscale (+ max_pool_backward(g_depth, idx_depth)),
 * i.e. the rows of dynmm_axpby_pool_bwd_apply, fed block by block into dynmm_bn_bwd_reduce / _apply's arithmetic
 * (ReLU mask re-derived from x).  g_depth / idx_depth / off may be NULL. */
int dynmm_stem_bn_bwd_reduce(const float* g_out, const signed char* idx_out, const float* g_depth,
                             const signed char* idx_depth, const float* coef, const float* off, float cscale,
                             const float* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, double* sums, int N, int C, int H, int W, int sums_are_zero, void* stream);
int dynmm_stem_bn_bwd_apply(const float* g_out, const signed char* idx_out, const float* g_depth,
                            const signed char* idx_depth, const float* coef, const float* off, float cscale,
                            const float* x, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const double* sums, float* dx, float* dgamma, float* dbeta,
                            int N, int C, int H, int W, void* stream);
int dynmm_axpby_pool_bwd_apply(const float* g_out, const signed char* idx_out, const float* g_depth,
                               const signed char* idx_depth, const float* a, const float* b, const float* ca,
                               const float* cb, float cscale, float* dxr, float* dxd, int NC, int H, int W, void* stream);

/* ---- SkipESANet per-stage gate + 2-way blend (rgb_depth_fusion.py:29-65, model_utils.py:54-70,
 *      model_skip_mod.py:235-311) ----
 * Gate (evaluated when wnext != NULL) from the pooled maps sr,sd [N,C] of the stage's rgb/depth features:
 *   p = [sr; sd];  g = sigmoid(W2 relu(W1 p + b1) + b2);  s = sum_c g[c] p[c] / 2C  (== mean(x*g));
 *   w = sigmoid(s);  y = gumbel_softmax([w, 1-w]/temp, tau=1, hard)  with Gumbel noise G = -log E:
 *   E[n,2] ~ Exp(1) taken from `noise` when given, else drawn by Philox4x32-10 keyed (seed; n, offset);
 *   prev (stride prev_stride, optional): b1 = y1*prev, b0 = 1-b1.   wnext[N,2] = (b0,b1) or y.
 *   params: 4 pointers {W1[2C/16,2C], b1, W2[2C,2C/16], b2}.  Saves h[N,2C/16], g[N,2C],
 *   aux[N,6] = {w, ysoft0, ysoft1, y1, E0, E1}.
 * Blend coefficients (written when a != NULL) for out = a*rgb + b*depth:
 *   blend_mode 0: rgb only (1,0);  1: rgb+depth (1,1);  2: w0*rgb + w1*(rgb+depth) -> (w0+w1, w1)
 *   with wblend[N,2].  The backward returns d_wblend from da,db of dynmm_axpby_bwd_reduce, the pooled
 *   gradients dsr,dsd (to be fed to dynmm_axpby_bwd_apply), the 4 parameter gradients and d_prev[N]. */
int dynmm_reweigh_fwd(const float* sr, const float* sd, const float* const* params,
                      const float* wblend, int blend_mode, const float* prev, int prev_stride,
                      const float* noise, unsigned long long seed, unsigned long long offset,
                      float temp, int hard, float* a, float* b, float* wnext, float* h, float* g,
                      float* aux, int N, int C, void* stream);
size_t dynmm_reweigh_bwd_workspace_bytes(int N, int C);
int dynmm_reweigh_bwd(const float* d_wnext, const float* da, const float* db, const float* sr,
                      const float* sd, const float* const* params, const float* prev, int prev_stride,
                      const float* h, const float* g, const float* aux, float* const* dparams,
                      float* dsr, float* dsd, float* d_wblend, float* d_prev, float* workspace, float temp,
                      int N, int C, void* stream);

/* ---- global gate head (…globalgate.py:20-30, 263-272, 314-315, 391-394) ----
 * mode 0: logits = fc[5,J] . pooled[n,J];  weight = DiffSoftmax(logits, temp, hard)
 * mode 1: weight given (baseline / ini_stage one-hots), pooled/fc ignored
 * outputs: weight[N,5]; wcum[N,4] = {w0, w0+w1, w0+w1+w2, 1-w4}; soft[N,5] (saved);
 *          flop_loss = mean_k( mean_n(weight)[k] * flop_table[k] ).
 * force_branch (optional, device int[N]; benchmark / test knob): with hard != 0 the straight-through one-hot is
 * taken at force_branch[n] instead of the arg-max — a FIXED synthetic branch distribution with the gate
 * network, its soft output and its gradient still evaluated (BASELINE configs[3]). */
int dynmm_gate_head_fwd(const float* pooled, const float* fc, float* weight, float* wcum,
                        float* soft, float* flop_loss, const float* flop_table, const int* force_branch,
                        int N, int J, float temp, int hard, int mode, void* stream);
int dynmm_gate_head_bwd(const float* d_weight, const float* d_wcum, const float* d_loss,
                        const float* pooled, const float* fc, const float* soft,
                        const float* flop_table, float* d_pooled, float* d_fc,
                        int N, int J, float temp, void* stream);

/* ---- weighted multi-scale cross entropy (src/utils.py:34-50), SURVEY §8f-1 ----
 * loss_sum += sum_px w[t]*(-log softmax(x)[t]);  wsum += sum_px w[t]   (t = target-1, void skipped)
 * dx = (softmax - onehot) * w[t] * gscale[0]   (gscale = upstream_grad / wsum, device scalar). */
int dynmm_ce2d_fwd(const float* x, const unsigned char* target, const float* cw,
                   double* loss_sum_wsum /*[2]*/, int N, int C, int HW, int acc_is_zero, void* stream);
/* validate()'s losses (train.py:104-115; src/utils.py:53-74 CrossEntropyLoss2dForValidData, :77-97 ...Unweighted):
 * acc4 += (sum_px w[t]*CE, sum_px w[t], sum_px CE, #non-void px) in ONE pass over the logits; the caller zeroes acc4
 * at the start of a validation run and divides at its end (acc4[0]/weighted_pixel_sum, acc4[2]/acc4[3]). */
int dynmm_ce2d_valid(const float* x, const unsigned char* target, const float* cw,
                     double* acc4, int N, int C, int HW, void* stream);
/* train.py:313-321 on the device: losses[s] = acc[2s]/acc[2s+1] for the S scales,
 * total = sum_s losses[s] + ratio*max(0, flop_loss - budget), and the backward seeds gscale[s] = 1/acc[2s+1]
 * (for dynmm_ce2d_bwd) and d_flop = ratio*[flop_loss > budget].  flop_loss / d_flop optional. */
int dynmm_loss_head(const double* acc, int S, const float* flop_loss, float ratio, float budget,
                    float* losses, float* total, float* gscale, float* d_flop, void* stream);
int dynmm_ce2d_bwd(const float* x, const unsigned char* target, const float* cw,
                   const float* gscale, float* dx, int N, int C, int HW, void* stream);

/* ---- training tail: last learned 2x up-sampling (model.py:404-410) FUSED with the full-resolution weighted CE
 * (src/utils.py:34-50).  x [N,C,H,W] is the input of the up-sampling (C <= 64), w [C,9] / b [C] its depthwise conv,
 * target uint8 [N,2H,2W] (0 = void), cw [C].  The logits are never written: fwd adds this scale's (sum, wsum) to
 * loss_sum_wsum and stores lse[N,2H,2W] (log-sum-exp per output pixel); bwd re-derives the logits and returns
 * dx (gradient of x), dw [C,9], db [C] for d loss = gscale[0] * d(sum).  workspace >= *_workspace_bytes. */
int dynmm_up2ce_fwd(const float* x, const float* w, const float* b, const unsigned char* target, const float* cw,
                    float* lse, double* loss_sum_wsum, int N, int C, int H, int W, int acc_is_zero, void* stream);
size_t dynmm_up2ce_bwd_workspace_bytes(int N, int C, int H, int W);
int dynmm_up2ce_bwd(const float* x, const float* w, const float* b, const unsigned char* target, const float* cw,
                    const float* lse, const float* gscale, float* dx, float* dw, float* db, float* workspace,
                    int N, int C, int H, int W, void* stream);

/* ---- eval post-processing + confusion matrix on device (eval.py:117-141, src/confusion_matrix.py:118-130;
 * SURVEY §8f-2): cm[(label-1)*C + argmax_c bilinear(logits)(label pixel)] += 1 for label > 0.
 * logits [N,C,H,W]; label uint8 [N,Ho,Wo] (0 = void); cm int64 [C*C], accumulated (caller zeroes). */
int dynmm_eval_confusion(const float* logits, const unsigned char* label, long long* cm,
                         int N, int C, int H, int W, int Ho, int Wo, void* stream);

/* ---- gate-decision stream compaction (new: SURVEY.md K16; the reference never skips compute,
 * src/models/model_skip_mod_globalgate.py:276-310) ----
 * gather: dst[i] = src[idx[i]]  (i < n_out);  merge: out[n] = map[n] >= 0 ? sub[map[n]] : base[n].
 * Rows are whole samples of `row` floats; idx/map are device int32. */
int dynmm_batch_gather(const float* src, const int* idx, float* dst, int n_out, size_t row, void* stream);
/* (idx == NULL: plain copy of n_out rows.)
 * Gate decision -> compaction plan, on the device: branch[n] = arg-max of the one-hot gate weights [N,5];
 * order = samples sorted by branch, descending, stable; inv = inverse permutation;
 * counts[j-1] = #{n : branch[n] >= j} (j = 1..4).  In the sorted batch the samples that run depth stage j are the
 * contiguous prefix of length counts[j-1]: stages run on prefix views, and the only device->host traffic of a
 * compacted forward is these 4 ints. */
int dynmm_gate_decide(const float* weight, int* branch, int* order, int* inv, int* counts, int N, void* stream);
int dynmm_batch_merge(const float* base, const float* sub, const int* map, float* out, int N, size_t row,
                      void* stream);

/* ---- helpers ---- */
/* out = srcs[0] + ... + srcs[n-1] (2 <= n <= 4, left to right): the gradient of an activation that fans out to
 * several consumers (block input -> first conv + down-sample conv, stage output -> skip connection + next stage,
 * ...) in one pass instead of autograd's pairwise accumulation. */
int dynmm_add_n(const float* const* srcs, int n, float* out, size_t numel, void* stream);
/* out[i] = sum_s slabs[s][i] */
int dynmm_reduce_slabs(const float* slabs, float* out, int n, int nslabs, void* stream);
/* ---- fused flat optimizers (train.py:554-579), SURVEY §8f-1 ----
 * p / g / state are the 16-byte aligned bases of flat fp32 buffers; one call updates the element range
 * [lo, hi) (parameters that took no gradient in a step are skipped by the caller, as torch.optim does).
 * hyper: DEVICE float array — SGD {lr, momentum}, Adam {lr, beta1, beta2, eps} — so values the driver changes
 * between steps (OneCycleLR: lr and momentum/beta1, train.py:119-128) reach a captured hipGraph.
 * step: DEVICE int step counter of the range's parameter group, incremented by dynmm_opt_tick before the call
 * (Adam's bias correction; torch keeps one counter per parameter).
 * loss / nan_flag (optional): if *loss is not finite the update is skipped and *nan_flag (int, zeroed by the
 * caller) latches 1 + *step — the device-side form of train.py:334-335.
 *   SGD : d = g*grad_scale + wd*p;  buf = mom*buf + d;  p -= lr*(d + mom*buf)            (nesterov)
 *   Adam: d as above; m = b1*m + (1-b1)*d; v = b2*v + (1-b2)*d*d;
 *         p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)                                        */
int dynmm_opt_tick(int* step, void* stream);
int dynmm_sgd_nesterov(float* p, const float* g, float* buf, size_t lo, size_t hi, const float* hyper,
                       float weight_decay, float grad_scale, const float* loss, int* nan_flag,
                       const int* step, void* stream);
int dynmm_adam(float* p, const float* g, float* m, float* v, size_t lo, size_t hi, const float* hyper,
               const int* step, float weight_decay, float grad_scale, const float* loss, int* nan_flag,
               int decoupled /* != 0: AdamW, p *= 1 - lr*wd */, const float* grad_scale_dev /* optional */,
               void* stream);

/* ================= modality-level DynMM (ModalityDynMM/affect/affect_dyn.py, BASELINE configs[4]) =================
 * Activations are [B, D, T] (the layout the reference feeds its Conv1d after x.permute([0,2,1])); every Linear /
 * Conv1d(k=1) is dynmm_conv2d_* with H = 1, W = T.  PARITY UNPINNED: the experts are MultiBench modules
 * (unimodals.common_models.Transformer / MLP, fusions.common_fusions.Concat), which /root/reference does not
 * contain and does not pin; these entry points follow torch.nn.TransformerEncoderLayer (post-norm, ReLU), which
 * MultiBench's Transformer wraps. */
/* LayerNorm over D per token (b,t) of s = x + res (res optional: the encoder layer's residual connection, fused):
 * y = (s-mean)*rstd*gamma + beta; mean/rstd [B*T] saved for the backward; dx is also the gradient of res. */
int dynmm_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, int B, int D, int T, float eps, void* stream);
int dynmm_layernorm_bwd(const float* g, const float* x, const float* res, const float* gamma, const float* mean,
                        const float* rstd, float* dx, float* dgamma, float* dbeta, int B, int D, int T, void* stream);
/* Self-attention core of nn.MultiheadAttention for T <= 64, D/heads <= 32: qkv [B,3D,T] -> out [B,D,T];
 * probs [B*heads,T,T] saved for the backward (no dropout, no mask: the reference ignores the padding lengths). */
int dynmm_mha_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads, void* stream);
int dynmm_mha_bwd(const float* g, const float* qkv, const float* probs, float* dqkv, int B, int D, int T, int heads,
                  void* stream);
/* Dropout as nn.TransformerEncoderLayer applies it while training (p = 0.1: attention probabilities, attention-block output,
 * feed-forward hidden layer, feed-forward output): an element survives with probability 1-p and is scaled by 1/(1-p).
 * The decision for element i is Philox-4x32-10(counter = i, offset + *step; key = seed) >= p, a pure function of its
 * arguments: the backward pass regenerates it (no stored masks) and a captured hipGraph draws new masks at every replay
 * because `step` (optional) is a DEVICE counter the training step advances.  `mask` (optional, tests): explicit keep
 * flags, one byte per element, instead of the generator.  p == 0 or a NULL descriptor: no dropout. */
typedef struct dynmm_dropout {
    const unsigned char* mask;
    const unsigned long long* step;
    unsigned long long seed, offset;
    float p;
} dynmm_dropout;
/* y[i] = x[i] * keep_i / (1-p)  (forward and, with the same descriptor, backward); in place allowed. */
int dynmm_dropout_apply(const float* x, float* y, size_t n, const dynmm_dropout* drop, void* stream);
/* LayerNorm(dropout(x) + res): indices of the dropout site run over x's [B, D, T] layout.  Backward: dres = gradient of the
 * normalised sum (what the residual branch receives), dx = dres * keep/(1-p); either may be NULL. */
int dynmm_layernorm_drop_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                             float* mean, float* rstd, int B, int D, int T, float eps, const dynmm_dropout* drop,
                             void* stream);
int dynmm_layernorm_drop_bwd(const float* g, const float* x, const float* res, const float* gamma, const float* mean,
                             const float* rstd, float* dx, float* dres, float* dgamma, float* dbeta, int B, int D, int T,
                             const dynmm_dropout* drop, void* stream);
/* The same backward with a workspace (dynmm_layernorm_bwd_workspace_bytes): dgamma / dbeta come out of the input-gradient
 * pass as per-workgroup sums plus one ordered reduction, instead of a second pass over g, x and res.  workspace NULL = the
 * entry above. */
size_t dynmm_layernorm_bwd_workspace_bytes(int B, int D, int T);
int dynmm_layernorm_drop_bwd_ws(const float* g, const float* x, const float* res, const float* gamma, const float* mean,
                                const float* rstd, float* dx, float* dres, float* dgamma, float* dbeta, int B, int D, int T,
                                const dynmm_dropout* drop, float* workspace, size_t workspace_bytes, void* stream);
/* The same forward for an x that arrives as `nparts` partial sums ([nparts][B, D, T], added in slab order) plus an optional
 * per-channel bias `xbias` — what dynmm_ffn_fwd leaves behind; `xsum` [B, D, T] receives the assembled x (the `x` of
 * dynmm_layernorm_drop_bwd). */
int dynmm_layernorm_parts_fwd(const float* parts, int nparts, const float* xbias, float* xsum, const float* res,
                              const float* gamma, const float* beta, float* y, float* mean, float* rstd, int B, int D, int T,
                              float eps, const dynmm_dropout* drop, void* stream);
/* ---- the feed-forward block of nn.TransformerEncoderLayer (torch/nn/modules/transformer.py `_ff_block`:
 * linear2(dropout(relu(linear1(x)))), the experts of ModalityDynMM/affect/affect_dyn.py:107-175) on [B, D, T], one launch ----
 * supported: D <= 128, F % 32 == 0, 16-byte aligned weights.  nsplit = dynmm_ffn_nsplit(...) ways of splitting the
 * hidden units over workgroups (a divisor of F / 32).
 * fwd: hidden [B, F, T] = dropout(relu(w1 x + b1)) (kept for the backward), out_parts [nsplit][B, D, T] = partial sums of
 *      w2 . hidden WITHOUT b2 (dynmm_layernorm_parts_fwd adds them and b2).  drop: indices over hidden's layout; with the
 *      generator one Philox call serves eight hidden units of a token (16 bits each: P(keep) = 1 - round(65536 p) / 65536).
 * bwd_data: dhidden = (w2^T dout) * (hidden > 0) / (1 - p), dx_parts [nsplit][B, D, T] = partial sums of w1^T dhidden.  The
 *      weight / bias gradients are 1x1-convolution weight gradients of (hidden, dout) and (x, dhidden): dynmm_conv2d_wgrad. */
int dynmm_ffn_supported(int B, int D, int T, int F);
int dynmm_ffn_nsplit(int B, int D, int T, int F);
int dynmm_ffn_fwd(const float* x, const float* w1, const float* b1, const float* w2, float* hidden, float* out_parts, int B,
                  int D, int T, int F, int nsplit, const dynmm_dropout* drop, void* stream);
int dynmm_ffn_bwd_data(const float* dout, const float* hidden, const float* w1, const float* w2, float* dhidden,
                       float* dx_parts, int B, int D, int T, int F, int nsplit, float p, void* stream);
/* attention with dropout on the probabilities (indices over probs' [B*heads, T, T] layout); probs holds the
 * probabilities BEFORE dropout, which is what the backward needs together with the regenerated keep flags. */
int dynmm_mha_drop_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads,
                       const dynmm_dropout* drop, void* stream);
int dynmm_mha_drop_bwd(const float* g, const float* qkv, const float* probs, float* dqkv, int B, int D, int T, int heads,
                       const dynmm_dropout* drop, void* stream);
/* Mixture head (affect_dyn.py:152-165) + loss (Supervised_Learning.py:135-136): w = DiffSoftmax(logits[B,K]/temp),
 * out = sum_k w_k pred_k, aux = mean w[:,K-1], loss1 = mean |out - target|, scalars = {loss1, aux, loss1 + reg*aux};
 * with target != NULL also the backward seeds d_preds[k][B] (optional per k) and d_logits[B,K]. K <= 4. */
int dynmm_moe_head(const float* logits, const float* const* preds, int K, const float* target, float temp, int hard,
                   float reg, float* out, float* weight, float* scalars, float* const* d_preds, float* d_logits, int B,
                   void* stream);
/* backward of the blend alone, for arbitrary upstream gradients d_out[B] / d_aux[1] (either may be NULL). */
int dynmm_moe_blend_bwd(const float* d_out, const float* d_aux, const float* logits, const float* const* preds, int K,
                        const float* weight, float temp, float* const* d_preds, float* d_logits, int B, void* stream);
/* torch.nn.utils.clip_grad_norm_ (Supervised_Learning.py:143) on a flat gradient buffer: norm_and_coef[0] = total
 * L2 norm, [1] = min(1, max_norm / (norm + 1e-6)), to be passed to the optimizer as grad_scale_dev. */
size_t dynmm_clip_grad_norm_workspace_bytes(void);
int dynmm_clip_grad_norm(const float* flat_grad, size_t n, float max_norm, double* workspace, float* norm_and_coef,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNMM_HIP_H */
