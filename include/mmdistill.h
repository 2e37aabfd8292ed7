/* mmdistill.h — C ABI of libmmdistill_hip.so (MI355X / gfx950 kernels for the MM-DistillNet distillation step).
 *
 * Drop-in boundary: the reference has no FFI; its hot path is reached through Python
 * call signatures.  Each entry point below replaces the torch/torchvision operator(s) cited above it
 * (file:line relative to the reference tree).  Conventions: extern "C", raw DEVICE pointers + sizes,
 * fp32 NHWC activations ("rows" = B*H*W pixels x C channels), last argument is the hipStream_t to
 * launch on, return 0 on success / negative errno-style code on bad arguments or launch failure,
 * the caller owns every buffer including workspaces, no hidden allocation, no host synchronisation
 * (every entry point is hipGraph-capturable).
 */
#ifndef MMDISTILL_H
#define MMDISTILL_H
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMD_ACT_NONE 0
#define MMD_ACT_SWISH 1
#define MMD_ACT_SIGMOID 2

// BiFPN fast-attention fusion node: swish(sum_i w_i * operand_i), nearest-x2 upsample and SAME max-pool fused
// (src/YetAnotherEfficientDet.py:338-390).
int mmd_bifpn_fuse_fwd(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, float* out, int B, int H, int W, int C, hipStream_t stream);

// Fusion node + its depthwise 3x3 (SeparableConvBlock.depthwise_conv) in one launch; f_out (nullable) keeps the fused
// activation for the backward.
int mmd_bifpn_node_dw_fwd(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, float* f_out, float* zd, int B, int H, int W, int C, hipStream_t stream);

// Whole BiFPN node of a frozen net in one kernel: fusion + swish + depthwise 3x3 + 1x1 conv (w_pw [C, C] as stored upstream) + bias + folded
// BatchNorm (SeparableConvBlock(norm=True) after BiFPN._forward_fast_attention's weighted sum, src/YetAnotherEfficientDet.py:150-185,338-390,
// eval mode).  -22 for a width without a kernel: ask mmd_bifpn_node_fused_supported (C in {64, 112, 160, 224}: the fpn widths of D0 / D2 / D3 / D4) and keep mmd_bifpn_node_dw_fwd + mmd_pwconv_fwd.
int mmd_bifpn_node_fused_supported(int C);
int mmd_bifpn_node_fwd_fused(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* w_pw, const float* bias, const float* scale, const float* shift, float* y, int B, int H, int W, int C, hipStream_t stream);

// The same node for the TRAINABLE net in train mode: z = the raw 1x1-conv output (+ bias), stats [2C] (+)= [sum z, sum z^2] (the batch
// statistics of the node's BatchNorm, src/YetAnotherEfficientDet.py:171-176), zd = the depthwise output (read by the 1x1 conv's weight
// gradient in the backward).  C as mmd_bifpn_node_fused_supported.
int mmd_bifpn_node_fwd_fused_train(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats, int B, int H, int W, int C, hipStream_t stream);

// Backward of the fusion node, part 1: dx = df*swish'(x), wdot[i] += <dx, operand_i>.
int mmd_bifpn_fuse_bwd(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* df, float* dx, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, hipStream_t stream);

// Backward of w = relu(theta)/(sum+1e-4).
// Node backward in one launch: depthwise 3x3 input gradient (from an LDS tile of dzd, the gradient w.r.t. the depthwise output)
// + fusion backward (same outputs as mmd_bifpn_fuse_bwd fed with df = dwconv^T(dzd, w_dw)); dup (nullable, needs `up`): the gradient of
// the nearest-upsampled operand [B, H/2, W/2, C] (=|+=) w_up * 2x2 block sums, written here instead of by mmd_upsample2_bwd_acc;
// dw_grad (nullable): the node's depthwise weight gradient [9, C] += , from the recomputed fused activation and the dzd tile this launch
// already holds (instead of a mmd_dwconv_bwd_weight launch over a materialised f).
int mmd_bifpn_node_dw_bwd(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, hipStream_t stream);

// Round 3: "the writer of the last contribution to a gradient computes the BatchNorm-backward sums of the total".  Same launch as
// mmd_bifpn_node_dw_bwd; for each operand gradient it writes (d0 / d1 / dup) an optional (z, mean, invstd, sums): when this launch
// completes that gradient and the operand is the output y = BN(z) of a BatchNorm (a BiFPN node / down-channel output), sums [2C] (+)=
// [sum g, sum g*xhat], xhat = (z - mean)*invstd, over the completed gradient g - what mmd_bn_bwd_reduce(act = NONE) would compute in a
// launch of its own on the backward's serial chain (autograd of the BatchNorm in SeparableConvBlock, src/YetAnotherEfficientDet.py:171-176).
int mmd_bifpn_node_dw_bwd2(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0, const float* z1, const float* mean1, const float* invstd1, double* sums1, const float* zu, const float* meanu, const float* invstdu, double* sumsu, hipStream_t stream);

// + the POOLED operand's gradient out of the same launch (round 3): dpool [B, 2H, 2W, C] += w_pool * g at the arg-max element of every output
// pixel's 3x3 / stride-2 SAME window (first maximum in scan order; a winning zero-padding element swallows the gradient - torch's max-pool
// backward, src/YetAnotherEfficientNet.py:68-104), fp32 atomics on a buffer holding zeros or the earlier contributions - instead of dx + a
// mmd_maxpool_same_bwd_acc launch.  A scattered gradient has no last writer, so the BatchNorm-backward sums of a pooled tensor are kept
// linearly: (zp, meanp, invstdp, sumsp) receive the sums of this launch's share; own bit 0 / 1 / 2: the sums of d0 / d1 / dup likewise
// cover this launch's share instead of the accumulated total.
int mmd_bifpn_node_dw_bwd3(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0, const float* z1, const float* mean1, const float* invstd1, double* sums1, const float* zu, const float* meanu, const float* invstdu, double* sumsu, float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own, hipStream_t stream);

// Round 4, "lazy" BiFPN operands of the trainable net: a node's / down-channel conv's train-mode BatchNorm (no activation) is applied by its
// CONSUMERS while they load the operand, instead of by an mmd_affine_act launch behind every producer (40 launches on the student's forward
// chain; BiFPN._forward_fast_attention, src/YetAnotherEfficientDet.py:320-392).  Operand i (order in0, in1, up, pool) then holds the RAW 1x1
// output z_i; host arrays of 4 entries describe the transforms, a null entry = plain tensor.
//   forward:  y_i = z_i * scale_i + shift_i with (scale, shift) derived in-kernel from the live batch sums op_stats4[i] [2C] (double),
//             op_gamma4[i], op_beta4[i], op_count4[i] (long long rows) - bit-identical to mmd_bn_finalize's coefficients.  C <= 160.
//   backward: op_scale4[i] / op_shift4[i] = the finalized coefficients; applied wherever the launch needs the operand's value (fused
//             activation, fusion-weight dot products, pool arg-max); gradients / BatchNorm sums are w.r.t. the BatchNorm output as before.
int mmd_bifpn_node_fwd_fused_train_lz(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats, int B, int H, int W, int C, const void* op_stats4, const void* op_gamma4, const void* op_beta4, const void* op_count4, hipStream_t stream);
int mmd_bifpn_node_dw_bwd3_lz(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0, const float* z1, const float* mean1, const float* invstd1, double* sums1, const float* zu, const float* meanu, const float* invstdu, double* sumsu, float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own, const void* op_scale4, const void* op_shift4, hipStream_t stream);
// Whole-node BiFPN backward (round 4): mmd_bifpn_node_dw_bwd3_lz with the node's 1x1 conv's input gradient inside the launch (replaces
// mmd_pwconv_bwd_data_bn in front of it): dzd = BnBwd(g, z) . w_pw is computed per tile - g the gradient w.r.t. the node's BatchNorm output,
// z its raw 1x1 output [B*H*W, C], (bn_scale, bn_mean, bn_invstd) of that BatchNorm, bn_sums = [sum g, sum g xhat] over `count` rows,
// w_pw_t [C in, C out] = the conv's weight TRANSPOSED (the operand mmd_pwconv_bwd_data takes); dz_out receives the evaluated BatchNorm backward (the weight-gradient GEMM's operand), dgamma / dbeta (+)= the sums.
// C % 16 == 0, C <= 224; operand sets (in, up), (in, td, pool), (in, pool).  SeparableConvBlock backward, src/YetAnotherEfficientDet.py:150-185.
int mmd_bifpn_node_bwd_full(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0, const float* z1, const float* mean1, const float* invstd1, double* sums1, const float* zu, const float* meanu, const float* invstdu, double* sumsu, float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own, const void* op_scale4, const void* op_shift4, const float* g, const float* z, const float* bn_scale, const float* bn_mean, const float* bn_invstd, const double* bn_sums, long long count, const float* w_pw_t, float* dz_out, float* dgamma, float* dbeta, hipStream_t stream);
// mmd_bifpn_node_bwd_full picks its block shape from the launch size: with fewer than 128 64-channel blocks (the 4x4 .. 16x16 levels at B = 8)
// it runs 16-channel blocks - same results up to the order of the per-channel sums - and on launches of >= 128 blocks a pooled operand's
// scattered gradient is collected in an LDS tile of the fine map.  EXCLUSIVITY (mmd_bifpn_node_bwd_full, mmd_bifpn_node_dw_bwd3[_lz]): with the
// LDS tile the interior of `dpool` is read at block start and plainly stored at block end, so no other launch may write `dpool` while this
// one runs (every contributor to a scattered gradient on ONE stream; the engine asserts it).
// _form: the same launch with both choices made by the caller (tests, timing; per call - no process-wide state): small_below = launches with
// fewer 64-channel blocks than this take the 16-channel form (0 never, 1 << 30 always, < 0 default); pool_lds = 1 / 0 the LDS tile on / off, < 0 default.
int mmd_bifpn_node_bwd_full_form(const float* in0, const float* in1, const float* up, const float* pool, const float* theta, const float* w_dw, float* wdot, int B, int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up, float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0, const float* z1, const float* mean1, const float* invstd1, double* sums1, const float* zu, const float* meanu, const float* invstdu, double* sumsu, float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own, const void* op_scale4, const void* op_shift4, const float* g, const float* z, const float* bn_scale, const float* bn_mean, const float* bn_invstd, const double* bn_sums, long long count, const float* w_pw_t, float* dz_out, float* dgamma, float* dbeta, int small_below, int pool_lds, hipStream_t stream);

int mmd_bifpn_theta_bwd(const float* theta, const float* wdot, float* dtheta, int n, hipStream_t stream);

// d theta of every fusion node of a net in ONE launch: desc [nodes][2] = (offset of the node's theta in theta_base / dtheta_base, its
// operand count); wdot_all [nodes][4] = the dot products the node backward launches accumulated.
int mmd_bifpn_theta_bwd_batched(const float* theta_base, float* dtheta_base, const float* wdot_all, const long long* desc, int nodes, hipStream_t stream);

// dst (+)= w_idx(theta) * src (same-resolution operand gradient).
int mmd_scale_acc(const float* src, float* dst, const float* theta, int ntheta, int widx, int accumulate, long long numel, hipStream_t stream);

// Backward of nn.Upsample(scale_factor=2, nearest) (src/YetAnotherEfficientDet.py:223-226).
int mmd_upsample2_bwd_acc(const float* dx, float* dst, const float* theta, int ntheta, int widx, int accumulate, int B, int H, int W, int C, hipStream_t stream);

// MaxPool2dStaticSamePadding(3,2) forward: zero padding takes part in the max (src/YetAnotherEfficientNet.py:68-104).
int mmd_maxpool_same_fwd(const float* src, float* out, int B, int PH, int PW, int C, hipStream_t stream);

// Backward of MaxPool2dStaticSamePadding(3,2) in gather form (first maximum in scan order wins).
int mmd_maxpool_same_bwd_acc(const float* src, const float* dout, float* dst, const float* theta, int ntheta, int widx, int accumulate, int B, int PH, int PW, int C, hipStream_t stream);

// + (z, mean, invstd, sums): the BatchNorm-backward sums of the completed gradient dst when this launch is its last contribution and
// src = BN(z) (see mmd_bifpn_node_dw_bwd2).  C <= 512.
int mmd_maxpool_bwd_sums_ok(int B, int PH, int PW, int C);

int mmd_maxpool_same_bwd_acc2(const float* src, const float* dout, float* dst, const float* theta, int ntheta, int widx, int accumulate, int B, int PH, int PW, int C, const float* z, const float* mean, const float* invstd, double* sums, hipStream_t stream);

// Depthwise kxk TF-SAME conv, NHWC, fused producer BN+swish prologue, stats / eval-BN+swish / SE-pool epilogue.
// pool [B, C] (nullable, frozen nets): the squeeze-excite average pool (src/YetAnotherEfficientNet.py:470) accumulated as 64-bit fixed-point
// integers, pool[b, c] += round(2^36 * mean_hw y): integer atomics commute exactly, so a frozen net's outputs are bit-identical from run to
// run (zero the array first; mmd_se_fc_fwd_q reads it back).
// stats_ws/ws_slots (nullable/0): zeroed workspace of ws_slots*2C doubles; launches that would send > 128 blocks to one
// BatchNorm-sum address spread their f64 atomics over the slots and fold them into `stats` (workspace left zero).
// Replaces Conv2dStaticSamePadding(groups=C) (src/YetAnotherEfficientNet.py:433-435, src/YetAnotherEfficientDet.py:169-170) incl. F.pad (:51-65).
int mmd_dwconv_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int k, int stride, const float* in_scale, const float* in_shift, int in_act, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, const float* out_scale, const float* out_shift, int out_act, double* stats, long long* pool, double* stats_ws, int ws_slots, hipStream_t stream);

// Frozen-net MBConv front half in one kernel: expand 1x1 conv + folded BN0 + swish -> depthwise kxk/stride (TF-SAME) + folded BN1 + swish
// + squeeze-excite average pool (pool[B,Cmid] +=, nullable; Q36 fixed-point integers as for mmd_dwconv_fwd).  The 6x expanded tensor stays in LDS (MFMA -> LDS -> depthwise).
// w_expand [Cmid, Cin] as stored by the reference, w_dw tap-major [k*k, Cmid].  -22 for a geometry without a kernel: ask
// mmd_mbconv_expand_dw_supported (Cin in {16,24,32,40,48,56}, Cmid % 48 == 0, k in {3,5}, stride in {1,2}) and keep mmd_pwconv_fwd +
// mmd_dwconv_fwd otherwise.  Replaces MBConvBlock.forward's `_expand_conv`/`_bn0`/swish/`_depthwise_conv`/`_bn1`/swish/avg-pool
// in eval mode (src/YetAnotherEfficientNet.py:450-470).
int mmd_mbconv_expand_dw_supported(int Cin, int Cmid, int k, int stride);
int mmd_mbconv_expand_dw_fwd(const float* x, const float* w_expand, const float* scale0, const float* shift0, const float* w_dw, const float* scale1, const float* shift1, float* y, long long* pool, int B, int H, int W, int Cin, int Cmid, int k, int stride, hipStream_t stream);

// Input gradient of the depthwise conv.  With bn_sums (stride 1 only) the launch also accumulates the sums of the BatchNorm(+swish)
// backward that consumes dx: bn_sums[c] += sum dx*swish'(u), bn_sums[C+c] += sum dx*swish'(u)*xhat, u = bn_z*bn_scale+bn_shift,
// xhat = (bn_z-bn_mean)*bn_invstd (bn_z = that BN's forward input, same shape as dx); stats_ws/ws_slots as in mmd_dwconv_fwd.
// dw_grad (nullable; needs bn_sums): the conv's weight gradient [k*k, C] += out of the same launch, with the forward input taken as
// swish(bn_z*bn_scale+bn_shift) - what mmd_dwconv_bwd_weight computes from x = bn_z with that producer transform.
int mmd_dwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int k, int stride, const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_sums, double* stats_ws, int ws_slots, float* dw_grad, hipStream_t stream);

// Round 3: the same launch for an MBConv block's stride-1 depthwise conv (C >= 64) with the BatchNorm-1 (+swish) backward - including the
// squeeze-excite terms: g' = (g1 * q_gate[img,c] + q_add[img,c]) * swish'(z1*scale+shift) - evaluated on the dY operand while its tile is
// staged, instead of by mmd_bn_bwd_apply(mul_bc = gate, add_bc = dpooled) writing dz1 to HBM first; q_dgamma / q_dbeta (+)= from q_sums.
// bn_* / dw_grad: as mmd_dwconv_bwd_data (all required).  Autograd of src/YetAnotherEfficientNet.py:462-474.
int mmd_dwconv_bwd_data_bn1(const float* g1, const float* z1, const float* w, float* dx, int B, int H, int W, int C, int k, const float* q_scale, const float* q_shift, const float* q_mean, const float* q_invstd, const double* q_sums, long long q_count, const float* q_gate, const float* q_add, float* q_dgamma, float* q_dbeta, const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_sums, double* stats_ws, int ws_slots, float* dw_grad, hipStream_t stream);

// Weight gradient of the depthwise conv, tap-major dw[k*k, C] (+=).
int mmd_dwconv_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int k, int stride, const float* in_scale, const float* in_shift, int in_act, hipStream_t stream);

// Train-mode BatchNorm2d: batch statistics -> (scale, shift, mean, invstd) + running-stat update (momentum 0.01, eps 1e-3).
// Replaces nn.BatchNorm2d forward in training (call sites SURVEY 2.1).
int mmd_bn_finalize(const double* stats, long long count, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean_out, float* invstd_out, int C, hipStream_t stream);

// All train-mode BN layers of a net finalized in one launch (running stats + saved mean/invstd for the backward);
// forward consumers derive (scale, shift) on the fly from the raw sums (in_stats/in_gamma/in_beta/in_count arguments).
int mmd_bn_finalize_all(const double* stats_flat, const float* count, const int* layer_off, const int* layer_C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean_out, float* invstd_out, int total, long long* num_batches_tracked, int n_layers, hipStream_t stream);

// Eval-mode BatchNorm2d folded to per-channel (scale, shift).
int mmd_bn_fold(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift, int C, hipStream_t stream);

// y = act(z*scale+shift) * rowscale[image] + res : BN apply, drop-connect scaling and identity skip
// (src/YetAnotherEfficientNet.py:173-182,479-485).
int mmd_affine_act(const float* z, const float* scale, const float* shift, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, int act, const float* rowscale, int rows_per_image, const float* res, float* y, int M, int C, hipStream_t stream);

// out[b,c] += s * sum_hw (g? g*a : a), a = act(z*scale+shift): SE global average pool (src/YetAnotherEfficientNet.py:470) and d(gate).
int mmd_chan_pool(const float* z, const float* scale, const float* shift, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, int act, const float* g, float* out, float out_scale, int B, int rows_per_image, int C, hipStream_t stream);

// Squeeze-excite FCs: gate = sigmoid(We*swish(Wr*pooled+br)+be) (src/YetAnotherEfficientNet.py:471-474).
int mmd_se_fc_fwd(const float* pooled, const float* wr, const float* br, const float* we, const float* be, float* hpre, float* gate, int B, int C, int S, hipStream_t stream);
// The frozen nets' form: pooled_q [B, C] holds the Q36 fixed-point pool sums of mmd_dwconv_fwd / mmd_mbconv_expand_dw_fwd (value = q / 2^36).
int mmd_se_fc_fwd_q(const long long* pooled_q, const float* wr, const float* br, const float* we, const float* be, float* hpre, float* gate, int B, int C, int S, hipStream_t stream);

// Round 4, grouped frozen nets: several frozen nets of ONE architecture (the three teachers) evaluated as one batch of n_groups x
// images_per_group images - a launch covers the same layer of every net and each workgroup picks its net's parameters by the image it
// works on: group g = image / images_per_group reads its weights g * w_stride floats and its folded BatchNorm coefficients g * bn_stride
// floats behind the pointers passed in (the nets' flat parameter / coefficient buffers are laid out alike, a constant stride apart).
// mmd_set_group applies to the launches the CALLING HOST THREAD issues after it until cleared with n_groups <= 1 (thread-local descriptor:
// other threads' launches never see it); clearing returns -22 when no launch issued in between read the group, i.e. an entry point
// without a group mode ran under it (with the first net's parameters for every image).  Honoured by the frozen
// forward entry points mmd_pwconv_fwd / mmd_pwconv_fwd_pyr (LDS-tiled kernels; every group's rows are whole 128-row tiles),
// mmd_dwconv_fwd, mmd_dwconv3_pyr (forward), mmd_mbconv_expand_dw_fwd, mmd_se_fc_fwd(_q), mmd_bifpn_node_fwd_fused and mmd_bifpn_node_dw_fwd; they return -22
// for a launch form the mode does not cover (statistics, live BatchNorm prologues, bf16 storage).
int mmd_set_group(int n_groups, int images_per_group, long long w_stride, long long bn_stride);

// One pass over (z1, g1): out5[5][B][C] += (sum g1*swish(u), sum g1*swish'(u), sum g1*swish'(u)*xhat, sum swish'(u), sum swish'(u)*xhat)
// per (image, channel): d(gate) for the SE backward plus the partials of the BatchNorm-1 backward sums (autograd of
// src/YetAnotherEfficientNet.py:466-476), so the expanded tensor is not read again by a BN reduce pass.
int mmd_chan_pool_bwd(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* g1, float* out5, int B, int rows_per_image, int C, hipStream_t stream);

// Backward of the squeeze-excite FCs (weight grads +=, dpooled scaled by dpool_scale = 1/HW).  With pool5 (from
// mmd_chan_pool_bwd; dgate = pool5[0]) it also finishes the BN-1 sums: bn_sums[c] += sum_b gate*pool5[1] + dpooled*pool5[3], [C+c] likewise with [2],[4].
int mmd_se_fc_bwd(const float* dgate, const float* gate, const float* hpre, const float* pooled, const float* wr, const float* we, float* dpe_ws, float* dpr_ws, float* dh_zeroed, float* dpooled, float dpool_scale, float* dwr, float* dbr, float* dwe, float* dbe, int B, int C, int S, const float* pool5, double* bn_sums, hipStream_t stream);

// Weight/bias gradients of the SE FCs alone (mmd_se_fc_bwd with dwr == NULL skips them): dpe/dpr are mmd_se_fc_bwd's workspaces.
int mmd_se_fc_wgrad(const float* dpe, const float* dpr, const float* hpre, const float* pooled, float* dwr, float* dbr, float* dwe, float* dbe, int B, int C, int S, hipStream_t stream);

// mmd_se_fc_bwd's two data-gradient kernels as ONE launch (dh is not materialised; C <= 3072): dpe_ws [B,C], dpr_ws [B,S], dpooled [B,C],
// and with pool5 / bn_sums the BatchNorm-1 backward sums, exactly as mmd_se_fc_bwd(dwr = NULL) (src/YetAnotherEfficientNet.py:469-474).
int mmd_se_fc_bwd_fused(const float* dgate, const float* gate, const float* hpre, const float* wr, const float* we, float* dpe_ws, float* dpr_ws, float* dpooled, float dpool_scale, int B, int C, int S, const float* pool5, double* bn_sums, hipStream_t stream);

// mmd_se_fc_wgrad for every squeeze-excite block of a backward segment in one launch.  desc: device array of n records of ten 8-byte fields
// {dpe, dpr, hpre, pooled, dwr, dbr, dwe, dbe (pointers), C, S (64-bit ints)}; max_cs = the largest C*S among them.
int mmd_se_fc_wgrad_batched(const void* desc, int n, int max_cs, int B, hipStream_t stream);

// BN(+swish) backward pass 1: g = (g_in*mul+add)*act'(y); per-channel sum(g), sum(g*xhat). g_out may be NULL (g not stored).
int mmd_bn_bwd_reduce(const float* g_in, const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, int act, const float* mul_bc, const float* mul_b, const float* add_bc, int rows_per_image, float* g_out, double* sums, int M, int C, double* stats_ws, int ws_slots, hipStream_t stream);

// BN backward pass 2: dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dgamma/dbeta +=. With act/mul/add given, `g` is pass 1's g_in and g is recomputed.
int mmd_bn_bwd_apply(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma, const double* sums, long long count, float* dz, float* dgamma, float* dbeta, int M, int C, const float* scale, const float* shift, int act, const float* mul_bc, const float* mul_b, const float* add_bc, int rows_per_image, hipStream_t stream);

// out[c] += sum_rows a[row,c] (bias gradients).
int mmd_colsum(const float* a, float* out, int M, int C, hipStream_t stream);

// Stem 3x3/s2 TF-SAME conv lowered to rows [B*OH*OW, Kp] (src/YetAnotherEfficientNet.py:519,603); the GEMM is mmd_pwconv_fwd.
int mmd_stem_im2col(const float* x, float* col, int B, int Cin, int H, int W, int Kp, hipStream_t stream);
// Weight gradient of the stem conv without the im2col matrix (round 4; csrc/stem_wgrad.hip): dw [Cout, Kp] (+)= sum over the output pixels of
// dz[(b, oh, ow)][co] * x[b][ci][2 oh + i - pad_t][2 ow + j - pad_l] at column ci*9 + i*3 + j - autograd of Conv2dStaticSamePadding(Cin, Cout, 3, stride=2)
// (src/YetAnotherEfficientNet.py:519-523, 597-604; padding :27-65).  x [B, Cin, H, W] NCHW, dz [B*OH*OW, Cout] rows.  ws: mmd_stem_wgrad_ws_floats(Cout)
// floats of scratch (per-block partials, folded in a fixed order).  _supported -> 1 for Cin <= 8, Kp <= 80, Cout <= 48, ceil(W/2) % 64 == 0.
int mmd_stem_conv_bwd_weight_supported(int Cin, int H, int W, int Kp, int Cout);
int mmd_stem_wgrad_ws_floats(int Cout);
int mmd_stem_conv_bwd_weight(const float* x, const float* dz, float* dw, float* ws, int B, int Cin, int H, int W, int Kp, int Cout, hipStream_t stream);
// ... with the stem BatchNorm(+swish) backward in the prologue: dz = BnBwd(g, z) is evaluated while the tile is staged (never a tensor), dgamma /
// dbeta (+)= sums - autograd of `_bn0` + swish behind the stem conv (src/YetAnotherEfficientNet.py:523-524); arguments as mmd_pwconv_bwd_data_bn.
int mmd_stem_conv_bwd_weight_bn(const float* x, const float* g, const float* z, float* dw, float* ws, int B, int Cin, int H, int W, int Kp, int Cout, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, float* dgamma, float* dbeta, hipStream_t stream);

// dlogit = dprob * p * (1-p) (classifier sigmoid, src/YetAnotherEfficientDet.py:529).
int mmd_sigmoid_bwd(const float* dprob, const float* prob, float* dlogit, long long n, hipStream_t stream);

// Copy one pyramid level out of a concatenated [B, A_total, C] head tensor.
int mmd_slice_rows(const float* src, float* dst, int B, int rows, int N, long long batch_stride, long long offset, hipStream_t stream);
// ... every level in one launch, into the level-major pyramid matrix the head backward works on (the split the reference's per-level loop
// implies, src/YetAnotherEfficientDet.py:470-480 / 520-530 run backwards): dst[row0[l] + b * HW_l + r, :] = src[b * batch_stride + offsets[l] + r * N ...];
// pyr_desc as for mmd_dwconv3_pyr, offsets = one host value per level.
int mmd_slice_rows_pyr(const float* src, float* dst, const int* pyr_desc, int N, long long batch_stride, const long long* offsets, hipStream_t stream);

// MTALoss.at without the normalisation: a[b,j] = mean_c f^p (src/loss/MTALoss.py:76-77).
int mmd_mta_attention(const float* f, float* a, int rows, int C, float p, hipStream_t stream);

// Per-level MTA loss (pairwise or list mode) and d/d(student attention) (src/loss/MTALoss.py:36-74).
int mmd_mta_kl(const float* a_s, const float* a_t0, const float* a_t1, const float* a_t2, int nteachers, int B, int HW, float T, float* loss, float* da_s, float gscale, int accumulate, hipStream_t stream);

// df (+)= da * p * f^(p-1) / C.
// Every (pyramid level, teacher) pair of one step in a single launch (host arrays of device pointers: a_s[nlev], a_t[nteachers*nlev]
// teacher-major, da_s[nlev] nullable, HW[nlev]).  list_mode = 0: pairwise MTALoss calls (src/optimization/train_methods.py:341-346),
// loss[t*nlev + l], da_s[l] accumulated over teachers with atomics (zero on entry when nteachers > 1); list_mode = 1: the list form
// (src/loss/MTALoss.py:36-52), loss[l].
int mmd_mta_kl_multi(const float* const* a_s, const float* const* a_t, float* const* da_s, const int* HW, int nlev, int nteachers, int list_mode, int B, float T, float* loss, float gscale, hipStream_t stream);

int mmd_mta_attention_bwd(const float* f, const float* da, float* df, int rows, int C, float p, int accumulate, hipStream_t stream);

// YetAnotherFocalLoss forward + gradients (src/loss/YetAnotherFocalLoss.py:27-190).
int mmd_focal_loss(const float* cls, const float* reg, const float* anchors, const float* boxes, const int* nbox, int maxg, int B, int A, int NC, int* assign_ws, int* npos_ws, double* acc_ws, float* loss_out, float* dcls, float* dreg, float grad_scale, int to_logit, int* any_boxes, hipStream_t stream);

// torch.optim.Adam step on a flat segment (src/optimization/train_methods.py:825-833, traditional.py:190).
int mmd_adam_step(float* p, const float* g, float* m, float* v, float* state, const float* hyper, const int* active, float grad_scale, long long n, hipStream_t stream);

// Adam over the whole flat buffer with the head (regressor/classifier) ranges skipped until they first receive a gradient.
int mmd_adam_step_gated(float* p, const float* g, float* m, float* v, float* state_main, float* state_head, const float* hyper, const int* head_active, long long b0, long long e0, long long b1, long long e1, long long b2, long long e2, float grad_scale, long long n, hipStream_t stream);

// The reference's other optimizers on the same flat, head-gated pass (cfg `optimizer` = SGD | Adam | AdamW,
// src/optimization/train_methods.py:808-836): mode 0 Adam, 1 AdamW, 2 SGD(momentum, weight_decay; m = momentum buffer).
// hyper6: device floats [lr, beta1, beta2, eps, weight_decay, momentum].
int mmd_opt_step_gated(int mode, float* p, const float* g, float* m, float* v, float* state_main, float* state_head, const float* hyper6, const int* head_active, long long b0, long long e0, long long b1, long long e1, long long b2, long long e2, float grad_scale, long long n, hipStream_t stream);

// hipMemsetAsync wrapper (graph-capturable zeroing of stats / gradient buffers).
int mmd_memset_async(void* p, int value, long long bytes, hipStream_t stream);

// Drop-connect scales of one step, drawn on the device (drop_connect, src/YetAnotherEfficientNet.py:173-182): out [n_skip, batch] =
// floor(keep[s] + U[0,1)) / keep[s], U from Philox4x32-10 keyed by `seed` with counter (state[0] = draws so far, element / 4).  state [2]
// (device): state[0] is advanced by the launch, state[1] != 0 = the caller injected out[] itself and the launch leaves it alone.
// Capturable: the replay loop of a captured step then issues no RNG / elementwise launches of its own.
int mmd_drop_scale(float* out, const float* keep, int n_skip, int batch, unsigned long long seed, unsigned long long* state, hipStream_t stream);

// clip_grad_norm_ on the flat gradient buffer (src/optimization/traditional.py:184-188).
int mmd_clip_grad_norm(float* g, long long n, float max_norm, double* sumsq_ws, hipStream_t stream);

// Box decode + clip + conf threshold + class filter, ordered compaction
// (src/YetAnotherEfficientDet.py:574-602, src/utils/utils.py:123-204).
int mmd_decode_filter(const float* cls, const float* reg, const float* anchors, int B, int A, int NC, float conf_threshold, unsigned long long valid_class_mask, float image_size, float* score_ws, unsigned char* clsid_ws, unsigned char* flags_ws, float* over_scores, float* cand, int* n_over, int* n_keep, int* overflow, int cap, hipStream_t stream);

// Per-teacher batched_nms + int truncation + label remap (src/utils/utils.py:205-231,285-323).
// cand / out: [B, cap, 6] rows; mask_ws: B*1024*16 words; big_ws (nullable when cap <= 1024): B * mmd_nms_ws_floats(cap) floats.
// The reference runs torchvision's NMS over EVERY over-threshold anchor (src/utils/utils.py:179-205, no cap): lists longer than
// 1024 rows take a chunked (1024 rows at a time, exact greedy) path through big_ws.
int mmd_nms_teacher(const float* cand, const int* n_keep, const float* over_scores, const int* label_map, float nms_threshold, int inclusive, float image_size, int B, float* out, int* out_cnt, unsigned long long* mask_ws, int* overflow, int cap, float* big_ws, hipStream_t stream);

// Cross-teacher concat + nms(0.5) + drop score (src/optimization/train_methods.py:361-411).
// merge01 != 0: image 1 also takes image 0's rows, in front of its own, when both have rows (augment=True, :379-387).
int mmd_nms_merge(const float* t0, const int* c0, const float* t1, const int* c1, const float* t2, const int* c2, int nteachers, float iou_threshold, int inclusive, int B, float* boxes, int* nbox, int maxg, unsigned long long* mask_ws, int* overflow, int merge01, int cap, float* big_ws, hipStream_t stream);

// The same merge over 1..4 sources (host arrays of device pointers, teacher order); the 4th source is the "augmentation" pass of
// ModelWithNMSKDListLossAugmented (src/optimization/train_methods.py:73-110).  big_ws: B * mmd_nms_ws_floats(nsrc * cap * (merge01 ? 2 : 1)).
int mmd_nms_merge_n(const float* const* srcs, const int* const* cnts, int nsrc, float iou_threshold, int inclusive, int B, float* boxes, int* nbox, int maxg, unsigned long long* mask_ws, int* overflow, int merge01, int cap, float* big_ws, hipStream_t stream);

// Floats per image of `big_ws` for NMS lists of up to nmax rows (0 when nmax <= 1024).  mmd_nms_merge: nmax = nteachers * cap * (merge01 ? 2 : 1).
int mmd_nms_ws_floats(int nmax);

// Rows per image the single-pass (all-in-LDS) NMS handles; the arrays' capacity itself is the run-time `cap` argument.
int mmd_pp_cap(void);

// Profiling hooks for bench.py (hipEvents on the launch stream).
int mmd_prof_is_on(int family);

int mmd_prof_dump_to(const char* path);

int mmd_prof_enable(int family, int on);

int mmd_prof_collect(int family, double* out);

// 1x1 conv as fp32 MFMA GEMM with fused producer-BN/swish/SE-gate prologue and bias/BN/act/residual/stats epilogue.
// Replaces nn.Conv2d(k=1) in Conv2dStaticSamePadding (src/YetAnotherEfficientNet.py:27-65; call sites :427,446,
// src/YetAnotherEfficientDet.py:171,238-265) + BatchNorm2d/swish that follow (:428,447,126-143).
int mmd_pwconv_fwd(const float* x, const float* w, float* y, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, const float* gate, int rows_per_image, const float* bias, const float* out_scale, const float* out_shift, int out_act, const float* residual, double* stats, long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots, hipStream_t stream);

// --- feature-pyramid ("pyr") launches: the 5 levels of a shared-weight head layer in ONE launch.  pyr_desc is a HOST
// int array {n, B, H0, W0, H1, W1, ...}; level l occupies rows [row0_l, row0_l + B*H_l*W_l) of the row buffer and
// every level starts at a multiple of 128 rows; per-level BatchNorm data sit lev_stride channels apart.
// (Regressor/Classifier.forward, src/YetAnotherEfficientDet.py:463-532: conv_list shared across levels, bn_list per level.)
int mmd_pwconv_fwd_pyr(const float* x, const float* w, float* y, const int* pyr_desc, int K, int N, const float* bias, int out_act, double* stats, long long lev_stride, long long y_batch_stride, const long long* y_off_lev, hipStream_t stream);

// dw_grad (nullable; flip = 1 launches): the conv's weight gradient [9, C] += from the same launch, x operand = act(wg_x * wg_scale + wg_shift)
// with per-level coefficients lev_stride apart (what mmd_dwconv3_pyr_bwd_weight(wg_x, x, ...) computes); bn_sums (nullable, needs dw_grad
// and a swish producer): the sums of the BatchNorm(+swish) backward that consumes y - what mmd_bn_bwd_reduce_pyr(y, wg_x, ...) computes.
int mmd_dwconv3_pyr(const float* x, const float* w, float* y, const int* pyr_desc, int C, int flip, const float* in_scale, const float* in_shift, int in_act, const double* in_stats, const float* in_gamma, const float* in_beta, long long lev_stride, const float* wg_x, const float* wg_scale, const float* wg_shift, int wg_act, float* dw_grad, const float* wg_mean, const float* wg_invstd, double* bn_sums, hipStream_t stream);

int mmd_dwconv3_pyr_bwd_weight(const float* x, const float* dy, float* dw, const int* pyr_desc, int C, const float* in_scale, const float* in_shift, int in_act, long long lev_stride, hipStream_t stream);

int mmd_bn_bwd_reduce_pyr(const float* g_in, const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, int act, const int* pyr_desc, long long lev_stride, float* g_out, double* sums, int C, hipStream_t stream);

int mmd_bn_bwd_apply_pyr(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma, const double* sums, const int* pyr_desc, long long lev_stride, float* dz, float* dgamma, float* dbeta, int C, const float* scale, const float* shift, int act, hipStream_t stream);

// out = a + b (+ c, nullable) over a pyramid row buffer [rows, C] (the two heads' input gradients + the MTA loss' feature gradient =
// the gradient w.r.t. the last BiFPN cell's outputs), and per level l with z[l] != NULL (host arrays of n device pointers; that level's
// feature map is BN(z[l]), z[l] its own [B*H*W, C] tensor) sums[l] [2C] (+)= [sum out, sum out*xhat]: the BatchNorm-backward reduce
// passes of the five output nodes (src/YetAnotherEfficientDet.py:338-390) in the launch that completes their upstream gradient.
int mmd_pyr_add_bnsums(const float* a, const float* b, const float* c, float* out, const int* pyr_desc, int C, const float* const* z, const float* const* mean, const float* const* invstd, double* const* sums, hipStream_t stream);

// dW[N,K] += dY^T * pro(X) (autograd of the 1x1 conv weight; reference: loss.backward(), src/optimization/traditional.py:182).
int mmd_pwconv_bwd_weight(const float* dy, const float* x, float* dw, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image, hipStream_t stream);

// All 1x1-conv weight gradients of a backward segment in ONE persistent launch + one deterministic fold (csrc/pw_wgrad_grouped.hip;
// reference: autograd of every nn.Conv2d(k=1) weight of the student, src/YetAnotherEfficientNet.py:427,446, src/YetAnotherEfficientDet.py:171,238-265).
// The caller fills dy / x / dw / in_scale / in_shift / gate / M / K / N / in_act / rows_per_image of each layer (same operand meaning as
// mmd_pwconv_bwd_weight), mmd_wgrad_plan (host) fills the rest and returns the item / tile counts and the workspace size; the planned
// table is then copied to device memory once and mmd_wgrad_grouped launched every step.  dW is WRITTEN (not accumulated), sums are
// bit-reproducible run to run (no atomics).
typedef struct MmdWgradLayer {
  const float* dy; const float* x; float* dw;
  const float* in_scale; const float* in_shift; const float* gate;
  int M, K, N; int in_act; int rows_per_image;
  int mchunk; int nsplit; int ntn, ntk; int item0; int tile0; int pad_;
  long long ws_off;
} MmdWgradLayer;
int mmd_wgrad_plan(MmdWgradLayer* layers_host, int n, int rows_per_item, int* n_items, int* n_tiles, long long* ws_floats);
int mmd_wgrad_grouped(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks, double flops, double bytes, hipStream_t stream);
// precision "bf16" (BASELINE configs[4]): the same launch with both operands rounded to bf16 at the MFMA input, fp32 accumulate
// (mmd_pwconv_bwd_weight_bf16's arithmetic for every layer of the table)
int mmd_wgrad_grouped_bf16(const void* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks, double flops, double bytes, hipStream_t stream);
// mmd_wgrad_grouped / _bf16 with one more choice made by the caller: rows32 = 1 promises that every layer's M is a multiple of 32 (any net
// at B >= 2: the smallest map is 4 x 4) - the kernel then runs without row clamps / masks and with running operand pointers
// (57 v_cndmask + 43 address instructions per 32-row step less on a kernel whose VALU instructions are MFMA time).  rows32 = 0: as the plain entry points.
// bf16: 0 = fp32 products - with rows32 on the bf16 matrix pipe in the SPLIT form (see mmd_pwconv_fwd_form below; MMD_MFMA_F32=1 in the
// environment keeps v_mfma_f32), 1 = operands rounded to bf16 (mmd_wgrad_grouped_bf16), 2 = fp32 products on v_mfma_f32_32x32x2_f32.
int mmd_wgrad_grouped_form(const void* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks, double flops, double bytes, int bf16, int rows32, hipStream_t stream);

// dX[M,K] (=|+=) dY[M,N] * W[N,K] using the transposed weight copy Wt[K,N] (autograd of the 1x1 conv input).
int mmd_pwconv_bwd_data(const float* dy, const float* wt, float* dx, int M, int K, int N, int accumulate, hipStream_t stream);

// --- bf16 mixed-precision variants of the four 1x1-conv entry points (BASELINE config 5: "bf16 mixed precision, MFMA bf16").
// Same contracts and fp32 tensors; the two MFMA operands are rounded to bf16 (round-to-nearest-even) right before
// v_mfma_f32_32x32x16_bf16, accumulation / prologue / epilogue / BatchNorm statistics stay fp32.  The reference has no
// reduced-precision path ("no AMP anywhere"); this is what torch.autocast(bf16) would make of its nn.Conv2d(k=1).
int mmd_pwconv_fwd_bf16(const float* x, const float* w, float* y, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, const float* gate, int rows_per_image, const float* bias, const float* out_scale, const float* out_shift, int out_act, const float* residual, double* stats, long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots, hipStream_t stream);

// mmd_pwconv_fwd with the kernel family chosen PER CALL (round 6; replaces the process-wide mmd_pwconv_rows_mode / mmd_pwconv_longk_mode
// setters - the library keeps no global state besides the communicator): form 0 = the measured shape filters decide (what
// mmd_pwconv_fwd does), 1 = thin-K row-slab kernel (csrc/pw_rows.hip), 2 = LDS-tiled kernels only, 3 = long-K LDS-DMA kernel (csrc/pw_longk.hip),
// 4 = all-N K-sliced slab kernel (csrc/pw_slab.hip).  A family that does not support the launch falls through to the LDS-tiled kernels.
// ws / ws_floats (nullable): caller-owned workspace for the slab kernel's K slices (mmd_pwconv_slab_ws_floats floats; contents undefined
// before and after, no zeroing needed); without one a launch that would need slices keeps the LDS-tiled kernels (form 0) or is refused (form 4).
// The engine calls this with form 0; forms 1 - 4: tests and A/B timing.  Conv2dStaticSamePadding(k=1), src/YetAnotherEfficientNet.py:27-65.
// form | 16 (MMD_PW_FORM_NATIVE): fp32 products on v_mfma_f32_32x32x2_f32.  Without it the LDS-tiled kernels compute the SAME fp32 GEMM in the
// split form where K >= 64 and N > 48 (csrc/common.h): each fp32 operand value is split exactly into three bf16 pieces (round to nearest:
// x = h + m + l, both residuals exact), a * b is taken as the six largest of the nine partial products - each exact in fp32 - on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulate; the three dropped ones are at most 2^-23 |a * b| (2^-27 rms), the size of ONE 2^-24 rounding of an fp32
// accumulate.  192 matrix-pipe cycles per 16-deep k group instead of 512.  Error against float64: not above the v_mfma_f32 chain's
// (tests/test_gpu_kernels.py::test_split3_precision; measured lower on every shape).  Operands must be finite (Inf - Inf in a residual is NaN).
// MMD_MFMA_F32=1 in the environment: the whole library on v_mfma_f32 (A/B timing, bisecting).
int mmd_pwconv_fwd_form(const float* x, const float* w, float* y, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, const float* gate, int rows_per_image, const float* bias, const float* out_scale, const float* out_shift, int out_act, const float* residual, double* stats, long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots, float* ws, long long ws_floats, int form, hipStream_t stream);

int mmd_pwconv_fwd_pyr_bf16(const float* x, const float* w, float* y, const int* pyr_desc, int K, int N, const float* bias, int out_act, double* stats, long long lev_stride, long long y_batch_stride, const long long* y_off_lev, hipStream_t stream);

int mmd_pwconv_bwd_weight_bf16(const float* dy, const float* x, float* dw, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image, hipStream_t stream);

int mmd_pwconv_bwd_data_bf16(const float* dy, const float* wt, float* dx, int M, int K, int N, int accumulate, hipStream_t stream);

// --- BatchNorm backward as an operand prologue of the two gradient GEMMs of the conv in front of it (one launch and one
// [M, N] round trip less per BatchNorm on the backward's main chain).  (g, z): gradient w.r.t. the BatchNorm(+swish) output and the
// conv's raw output, both [M, N]; scale = gamma*invstd, shift = beta - mean*scale; sums[2N] = [sum g', sum g'*xhat] from
// mmd_bn_bwd_reduce (or the depthwise input-gradient epilogue); g' = g * mul_b[image] * swish'(z*scale+shift) (act = 1) - the same
// rule as mmd_bn_bwd_apply.  mmd_pwconv_bwd_data_bn side outputs (nullable): dz_out [M, N] = the evaluated BatchNorm backward, stored once
// so that the weight gradient can be the plain mmd_pwconv_bwd_weight(dz_out, x); dgamma / dbeta (+)= [sum g'*xhat, sum g'].  Autograd of nn.BatchNorm2d (train) behind nn.Conv2d(k=1): src/YetAnotherEfficientNet.py:427-428,446-447,477;
// src/YetAnotherEfficientDet.py:171-176.
int mmd_pwconv_bwd_data_bn(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int rows_per_image, float* dz_out, float* dgamma, float* dbeta, hipStream_t stream);

// mmd_pwconv_bwd_data_bn with two more (optional) jobs for its epilogue.  residual (may alias dx): dx = BnBwd(g, z) * W + residual - the
// skip branch's / earlier consumers' contributions already sit in the gradient buffer.  xs_*: dx is then the COMPLETE gradient w.r.t. a
// tensor y' = BN'(xs_z) * xs_mul_b[image] (+ skip) (an MBConv block output / a tap), and xs_sums [2K] (+)= [sum g', sum g'*xhat'],
// g' = dx * xs_mul_b[image], xhat' = (xs_z - xs_mean)*xs_invstd: the reduce pass of that upstream BatchNorm's backward, without a launch of
// its own and without the scale_acc launch that added the skip gradient (src/YetAnotherEfficientNet.py:477-485).  stats_ws / ws_slots
// (nullable / 0) as in mmd_pwconv_fwd: slotted sums for launches with more than 128 row tiles.  p5_* (an MBConv project conv; not with
// residual / xs): dx = g1 is the gradient w.r.t. the gated activation, and p5_out [5][p5_B][K] (+)= what mmd_chan_pool_bwd(p5_z = z1,
// p5_scale .. p5_invstd of BatchNorm-1, g1) computes - the pooled pass of the squeeze-excite / BatchNorm-1 backward - taken from the
// output tiles instead of by a launch re-reading g1 and z1.
int mmd_pwconv_bwd_data_bn2(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int rows_per_image, float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z, const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums, double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale, const float* p5_shift, const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B, hipStream_t stream);
// All-N, K-sliced slab GEMM (csrc/pw_slab.hip, round 6) for the small-M 1x1 convs whose A operand carries an arithmetic prologue (the
// student's expand-conv input gradients behind BatchNorm + swish, its project convs behind live BatchNorm + swish + gate; autograd of
// src/YetAnotherEfficientNet.py:427-447): a block owns a 32-row slab x ALL N x a slice of K, so the prologue is evaluated once per element
// instead of once per 64-wide column tile; launches with few row slabs are cut along K and the slices' partial slabs are added in slice
// order by a second launch that owns the epilogue (deterministic, no atomics).  mmd_pwconv_slab_ws_floats: workspace floats such a launch
// needs (0 = it runs unsliced, or the family does not take the shape); mmd_pwconv_bwd_data_bn2_form = mmd_pwconv_bwd_data_bn2 + workspace + form.
int mmd_pwconv_slab_ws_floats(int M, int K, int N, int bn_operand);
int mmd_pwconv_bwd_data_bn2_form(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int rows_per_image, float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z, const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums, double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale, const float* p5_shift, const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B, float* ws, long long ws_floats, int form, hipStream_t stream);

int mmd_pwconv_bwd_data_bn2_bf16(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int rows_per_image, float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z, const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums, double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale, const float* p5_shift, const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B, hipStream_t stream);

// Round 4: backward of an MBConv EXPAND conv with its train-mode BatchNorm-0 + swish in ONE pass over the 6x expanded gradient, for the
// thin-input high-resolution blocks (autograd of `_expand_conv` -> `_bn0` -> swish, src/YetAnotherEfficientNet.py:456-460):
//   dz0 = BnBwd0(g0, z0) evaluated once per element into LDS (never written to HBM);  dx[M, Cin] = dz0 . w[Cmid, Cin] (+ residual, may be dx);
//   dw[Cmid, Cin] += dz0^T . x[M, Cin];  dgamma / dbeta (+)= the reduce pass' sums;  optional xs_* as in mmd_pwconv_bwd_data_bn2 (dx completes
//   the gradient of an upstream BatchNorm output: xs_sums[2 Cin] (+)= its backward sums).
// Replaces mmd_pwconv_bwd_data_bn2 (which stores dz0 for the weight gradient) + the layer's entry in mmd_wgrad_grouped (which reads it back).
// mmd_mbconv_expand_bwd_supported(Cin, Cmid) -> 1 for (16, 96), (24, 144), (32, 192).
int mmd_mbconv_expand_bwd_supported(int Cin, int Cmid);
int mmd_mbconv_expand_bwd_fused(const float* g0, const float* z0, const float* x, const float* w, float* dx, const float* residual, float* dw, int M, int Cin, int Cmid, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, float* dgamma, float* dbeta, const float* xs_z, const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums, hipStream_t stream);

int mmd_pwconv_bwd_data_bn_bf16(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int rows_per_image, float* dz_out, float* dgamma, float* dbeta, hipStream_t stream);

int mmd_pwconv_bwd_weight_bn(const float* g, const float* z, const float* x, float* dw, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int bn_rows_per_image, float* dgamma, float* dbeta, hipStream_t stream);

int mmd_pwconv_bwd_weight_bn_bf16(const float* g, const float* z, const float* x, float* dw, int M, int K, int N, const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image, const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, int bn_rows_per_image, float* dgamma, float* dbeta, hipStream_t stream);

// Every 1x1 weight of the student transposed in one launch (desc rows: src_off, dst_off, R, C, first_tile).
int mmd_transpose_batched(const float* src_base, float* dst_base, const long long* desc, int n, int total_tiles, hipStream_t stream);

// dst[C,R] = src[R,C]^T (refreshes the Wt copies after an optimizer step).
int mmd_transpose2d(const float* src, float* dst, int R, int C, hipStream_t stream);

// Stem conv forward, direct: NCHW image -> NHWC rows [B*OH*OW, Cout] (Cout in {32,40,48,56,64}; 3x3, stride 2, TF-SAME; w = the [Cout, Kp] im2col weight
// layout, k = ci*9 + i*3 + j).  Epilogue: folded BN + swish (frozen nets) or raw output + BatchNorm sums (train; stats_ws/ws_slots
// as in mmd_dwconv_fwd).  src/YetAnotherEfficientNet.py:519-523,597-604.  mmd_stem_im2col + the GEMM entry points remain for the weight gradient.
int mmd_stem_conv_fwd(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Kp, int Cout, const float* out_scale, const float* out_shift, int out_act, double* stats, double* stats_ws, int ws_slots, hipStream_t stream);

// ModelWithNMSLossAugmented.merge_batch_0_1 (src/optimization/train_methods.py:291-308): out = in, except image 1 =
// log10(max(in[0]^10 + in[1]^10, 1e-7)) (the reference's literal 10th power).  per_image = C*H*W.
int mmd_audio_merge01(const float* in, float* out, long long per_image, int B, hipStream_t stream);

// ModelWithNMSLossAugmented.average_batch_0_1 (:279-289): f[image 1] = (f[image 0] + f[image 1]) / 2, in place.
int mmd_avg_image01(float* f, long long per_image, hipStream_t stream);

// ---- input preparation.  cv2 sampling rules restated (cv2 is not in the reference tree): parity with
// cv2 itself is unpinned, the kernels are pinned to oracle/input_ref.py.
// (min, max) of clamp(src, lo, hi) for the thermal min-max stretch (src/datasets/MultimodalDetection.py:196-211). dtype: 0 u8, 1 u16, 2 f32.
int mmd_image_minmax(const void* src, int dtype, long long n, float lo, float hi, float* mm, hipStream_t stream);

// Normalizer + Resizer + HWC->CHW for one image (src/datasets/transformations.py:315-330,407-433; MultimodalDetection.py:245-255):
// dst[C,S,S] = letterbox(bilinear((pre(src)*scale - mean)/std)); pre = clamp + round((v-min)*255/(max-min)) when minmax != 0.
int mmd_image_letterbox(const void* src, int dtype, int H, int W, int C, float scale, const float* mean, const float* stdv, int minmax, float lo, float hi, const float* mm, int common_size, float* dst, hipStream_t stream);

// Resizer's audio branch: cv2.resize(INTER_CUBIC) of an [h,w,C] spectrogram stack to [C,S,S] (transformations.py:435-441).
int mmd_resize_cubic(const float* src, int h, int w, int C, int common_size, float* dst, hipStream_t stream);


// ---- data-parallel exchange (RCCL over xGMI), SURVEY.md section 8b.  Replaces DistributedDataParallel's gradient reduction
// (src/optimization/train_methods.py:944-961): student gradients only, sum (the 1/N average rides in the optimizer pass), plus the
// MAX-reduce of head_active.  librccl is opened lazily: -38 when it is not installed.
// mmd_comm_unique_id: 128-byte rendezvous token made on rank 0 and handed to every rank by the host.
int mmd_comm_unique_id(void* out128);
int mmd_comm_init(void** comm_out, int rank, int world, const void* unique_id128);
// In place, asynchronous on `stream`; dtype 0 = float32, 1 = int32; op 0 = sum, 1 = max.
int mmd_comm_allreduce_bucket(void* comm, void* buf, long long count, int dtype, int op, hipStream_t stream);
// ranks RCCL reports for the communicator (ncclCommCount) -> *(int*)count_out
int mmd_comm_count(void* comm, void* count_out);
int mmd_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
