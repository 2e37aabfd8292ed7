// Argument block shared by the 1x1-conv GEMM kernels (pw_gemm.hip: LDS-tiled K loop; pw_rows.hip: thin-K row-slab kernel).
#pragma once
#include "common.h"

// BatchNorm(+swish, +drop-connect row scale) backward as a GEMM operand prologue: instead of a separate pass that writes
//   dz[m,c] = scale_c * ( g'[m,c] - m1_c - (z[m,c] - mean_c) * invstd_c * m2_c ),   g' = g * mul_b[image(m)] * swish'(z*scale_c + shift_c)
// (scale = gamma*invstd, shift = beta - mean*scale; m1 = sum(g')/count, m2 = sum(g'*xhat)/count from the reduce pass), the
// input-gradient and weight-gradient GEMMs of the conv in front of the BatchNorm read (g, z) and evaluate dz while staging
// their tile.  Removes one launch and one [M, C] write + two reads per BatchNorm from the backward's main chain.
// Reference: autograd of nn.BatchNorm2d in train mode, call sites src/YetAnotherEfficientNet.py:428,447,477.
struct BnBwdOp {
  const float* z; const float* scale; const float* shift; const float* mean; const float* invstd;
  const double* sums; double inv_count; int C; int act; const float* mul_b; int rows_per_image;
  // input-gradient launch only: the evaluated dz is also stored once ([M, C], by the blocks of the first column panel), so the
  // weight-gradient GEMM reads ONE plain tensor instead of re-evaluating the BatchNorm backward from (g, z) (+18-31 % on
  // that kernel); dgamma / dbeta (+)= the reduce pass' sums
  float* dz_out; float* dgamma; float* dbeta;
};
// per-channel coefficients: dz = a1*g' + a2*(z - mu) + a3
__device__ __forceinline__ void bn_bwd_coef(const BnBwdOp& b, int c, float& a1, float& a2, float& a3, float& mu, float& sh) {
  const float m1 = (float)(b.sums[c] * b.inv_count), m2 = (float)(b.sums[b.C + c] * b.inv_count);
  const float is = b.invstd[c];
  a1 = b.scale[c]; mu = b.mean[c]; sh = b.shift[c];
  a2 = -a1 * is * m2; a3 = -a1 * m1;
}
struct BnBwdCoef4 { float4 a1, a2, a3, mu, sh; };
// The coefficients of all K channels, once per block, in LDS: tab[5][K] (K % 4 == 0).  The GEMM kernels used to evaluate them per K step
// and thread from the batch sums (8 double loads + 12 float loads + ~45 VALU instructions per 4 channels, every row group of a step
// repeating the same channels): their BatchNorm-backward operand launches are instruction-bound (tools/dev/skinny_phases.py).
__device__ __forceinline__ void bn_bwd_tab_fill(const BnBwdOp& b, int K, float* tab, int tid, int nthreads) {
  for (int c = tid; c < K; c += nthreads) {
    float a1, a2, a3, mu, sh;
    bn_bwd_coef(b, c, a1, a2, a3, mu, sh);
    tab[c] = a1; tab[K + c] = a2; tab[2 * K + c] = a3; tab[3 * K + c] = mu; tab[4 * K + c] = sh;
  }
}
__device__ __forceinline__ void bn_bwd_tab4(const float* tab, int K, int c, BnBwdCoef4& o) {
  o.a1 = *reinterpret_cast<const float4*>(tab + c); o.a2 = *reinterpret_cast<const float4*>(tab + K + c);
  o.a3 = *reinterpret_cast<const float4*>(tab + 2 * K + c); o.mu = *reinterpret_cast<const float4*>(tab + 3 * K + c);
  o.sh = *reinterpret_cast<const float4*>(tab + 4 * K + c);
}
__device__ __forceinline__ void bn_bwd_coef4(const BnBwdOp& b, int c, BnBwdCoef4& o) {
  bn_bwd_coef(b, c, o.a1.x, o.a2.x, o.a3.x, o.mu.x, o.sh.x); bn_bwd_coef(b, c + 1, o.a1.y, o.a2.y, o.a3.y, o.mu.y, o.sh.y);
  bn_bwd_coef(b, c + 2, o.a1.z, o.a2.z, o.a3.z, o.mu.z, o.sh.z); bn_bwd_coef(b, c + 3, o.a1.w, o.a2.w, o.a3.w, o.mu.w, o.sh.w);
}
__device__ __forceinline__ float bn_bwd_eval(float g, float z, float rs, int act, float a1, float a2, float a3, float mu, float sh) {
  g *= rs;
  if (act == MMD_ACT_SWISH) g *= mmd_swish_grad(z * a1 + sh);
  return a1 * g + a2 * (z - mu) + a3;
}
// (ONE uniform branch on `act` per quad: per element hipcc emits a scalar branch each and, behind it, one element's exp / rcp chain alone -
// round 6, ISA of the GEMM epilogues' mmd_act, same pattern)
__device__ __forceinline__ float4 bn_bwd_eval4(float4 g, float4 z, float rs, int act, const BnBwdCoef4& q) {
  g.x *= rs; g.y *= rs; g.z *= rs; g.w *= rs;
  if (act == MMD_ACT_SWISH) {
    g.x *= mmd_swish_grad(z.x * q.a1.x + q.sh.x); g.y *= mmd_swish_grad(z.y * q.a1.y + q.sh.y);
    g.z *= mmd_swish_grad(z.z * q.a1.z + q.sh.z); g.w *= mmd_swish_grad(z.w * q.a1.w + q.sh.w);
  }
  return make_float4(q.a1.x * g.x + q.a2.x * (z.x - q.mu.x) + q.a3.x, q.a1.y * g.y + q.a2.y * (z.y - q.mu.y) + q.a3.y,
                     q.a1.z * g.z + q.a2.z * (z.z - q.mu.z) + q.a3.z, q.a1.w * g.w + q.a2.w * (z.w - q.mu.w) + q.a3.w);
}

// Stem 3x3 / stride-2 TF-SAME convolution as an implicit GEMM: row m = output pixel (b, oh, ow), k = ci*9 + i*3 + j, the A
// element is x[b, ci, 2*oh + i - pad_t, 2*ow + j - pad_l] (zero outside the image), gathered from the NCHW image while staging;
// the weight is the stem's native [Cout, Kp] matrix (mmd_stem_im2col's column order).  Replaces the direct VALU kernel
// (one thread per pixel x all output channels: 188 us for the 8-channel student stem).
struct StemOp { int Cin, H, W, OH, OW, pad_t, pad_l; };

// BatchNorm-backward sums of the launch's OUTPUT, taken in the epilogue: the output y (after the residual add) is the complete gradient
// w.r.t. a tensor BN(z)*rowscale (+ skip), so the reduce pass of THAT BatchNorm - sum g', sum g'*xhat with g' = y * mul_b[image],
// xhat = (z - mean)*invstd - rides here instead of a launch that re-reads y and z (the writer of the last contribution to a gradient
// computes the sums of the total).  `stats` (or its slotted workspace) receives [sum g', sum g'*xhat].
struct BnSumOp { const float* z; const float* mean; const float* invstd; const float* mul_b; int rows_per_image; };
// epilogue helper of the BnSumOp mode: v = the final output value at `off` (row `row`) -> the two BatchNorm-backward sums
// (_pre: the z quad and the row's scale already in registers - the epilogues issue every row's loads before the first use)
__device__ __forceinline__ void pw_xs_acc_pre(float4 v, const float4& zz, float rs, const float4& mu, const float4& is, float4& s4, float4& q4) {
  v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
  s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
  q4.x += v.x * (zz.x - mu.x) * is.x; q4.y += v.y * (zz.y - mu.y) * is.y;
  q4.z += v.z * (zz.z - mu.z) * is.z; q4.w += v.w * (zz.w - mu.w) * is.w;
}
__device__ __forceinline__ void pw_xs_acc(const BnSumOp& xs, float4 v, size_t off, int row, const float4& mu, const float4& is,
                                          float4& s4, float4& q4) {
  pw_xs_acc_pre(v, mmd_ld4(xs.z + off), xs.mul_b ? xs.mul_b[row / xs.rows_per_image] : 1.f, mu, is, s4, q4);
}

// Epilogue of an MBConv project conv's input-gradient GEMM: its output g1 [M, N] is the gradient w.r.t. the squeeze-excite-gated activation;
// the five per-(image, channel) sums the squeeze-excite backward and the BatchNorm-1 backward need (mmd_chan_pool_bwd's out5 [5][B][N]:
// sum g1*a, g1*s', g1*s'*xhat, s', s'*xhat with u = z*scale+shift, a = swish(u), s' = swish'(u), xhat = (z-mean)*invstd) are taken from
// the tile while it is written, instead of by a pass that re-reads g1 and z1.  A row tile must lie inside one image.
struct Pool5Op { const float* z; const float* scale; const float* shift; const float* mean; const float* invstd; float* out; int B; int rows_per_image; };

// Kernel family of a 1x1-conv launch, chosen per call (round 6: replaces the process-wide mmd_pwconv_rows_mode / _longk_mode setters - the
// library keeps no global state besides the communicator).  AUTO = the measured shape filters of pw_dispatch decide; the others force one
// family for every launch it supports (a launch it does not support falls through to the LDS-tiled kernels) - tests and A/B timing.
enum { MMD_PW_FORM_AUTO = 0, MMD_PW_FORM_ROWS = 1, MMD_PW_FORM_TILED = 2, MMD_PW_FORM_LONGK = 3, MMD_PW_FORM_SLAB = 4 };
// OR-ed into `form`: fp32 launches on v_mfma_f32_32x32x2_f32 instead of the split form (six bf16 MFMAs on a three-way exact split of both
// operands, common.h) - the per-call version of MMD_MFMA_F32=1; tests compare the two forms against float64 with it
enum { MMD_PW_FORM_NATIVE = 16 };

struct PwArgs {
  const float* x; const float* w; float* y;
  int M, K, N;
  const float* in_scale; const float* in_shift; int in_act; BnLive in_bn;
  const float* gate; int rows_per_image;
  const float* bias; const float* out_scale; const float* out_shift; int out_act;
  const float* residual; double* stats;
  double* stats_ws; int ws_slots;            // slotted sums (common.h), tiled kernel only
  long long y_batch_stride; long long y_offset;
  int ntn; int nblk;
  Pyr pyr; long long yoff_lev[MMD_MAX_LEV]; long long lev_stride;
  int bf16;                                  // host-side: operands rounded to bf16 at the MFMA input (mixed-precision mode)
  BnBwdOp bb;                                // PRO == 1: the A operand is a BatchNorm backward evaluated on the fly
  StemOp st;                                 // PRO == 2: the A operand is the im2col of an NCHW image, gathered on the fly
  BnSumOp xs;                                // epilogue: `stats` = BatchNorm-backward sums of the output instead of (sum y, sum y^2)
  Pool5Op p5;                                // epilogue: squeeze-excite / BatchNorm-1 backward partial sums of the output
  // bf16 storage ("w16", common.h): which tensors of the launch are bf16 arrays - the A operand x, the output y, and for the
  // BatchNorm-backward operand launches the second A tensor bb.z, the stored dz (bb.dz_out) and the pooled pass' z (p5.z)
  int x16, y16, z16, dz16, p5z16;
  // grouped frozen nets (common.h MmdGroup): g_images != 0 -> the row tile's group g = image / g_images reads w / bias g * g_w floats and
  // out_scale / out_shift g * g_bn floats behind the given pointers (LDS-tiled kernels only)
  int g_images; long long g_w, g_bn;
  float* slab_ws; long long slab_ws_floats;      // host-side: caller's workspace for the K slices' partial slabs of the slab kernel (pw_slab.hip)
  int native;      // host-side, per call: MMD_PW_FORM_NATIVE was set (a.form holds the family only)
  int form;        // host-side, per call (mmd_pwconv_fwd_form / _bwd_data_bn_form): which kernel family takes the launch, MMD_PW_FORM_*
  int bq_lds;      // BatchNorm-backward operand launches: the per-channel coefficients come from a per-block LDS table (5 x K floats of dynamic LDS)
};


// pw_rows.hip: thin-K (K <= 128) launches on the row-slab kernel; returns 1 when it took the launch, 0 when the shape / operand mode is
// not covered (the caller then uses the LDS-tiled kernels), < 0 on error.
int pw_rows_try(PwArgs& a, hipStream_t stream);
// pw_slab.hip: small-M launches with an arithmetic A prologue on the all-N, K-sliced slab kernel; same return convention (auto_ok: the shape
// filter of pw_dispatch - launches that would run the skinny kernel with PRO 0 / 1 - said yes; ignored when a.form forces the family)
int pw_slab_try(PwArgs& a, float* ws, long long ws_floats, bool auto_ok, hipStream_t stream);
// pw_longk.hip: long-K small-M launches (plain / gate-only A operand) on the LDS-DMA pipelined kernel; same return convention.
int pw_longk_try(PwArgs& a, hipStream_t stream);
