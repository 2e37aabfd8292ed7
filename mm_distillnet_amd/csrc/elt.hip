// Streaming (HBM-bound) kernels on NHWC fp32 rows: BatchNorm statistics/finalize/backward, fused
// affine+activation+residual, squeeze-excite pool and FCs, bias/column sums, stem im2col.
// Layout rule for all "row x channel" kernels: a thread owns 4 consecutive channels (float4) and walks
// rows; 16 lanes cover a 64-channel chunk (256 contiguous bytes), 16 row-groups per 256-thread block.
// Per-channel reductions: shuffle over the 4 row-groups of a wave (lanes l^16, l^32), then LDS over
// the 4 waves, then one double/float atomic per channel per block.
#include "common.h"
#include <cstdlib>

#define ROWS_PER_BLOCK 256
// Rows per block of the row-streaming elementwise / reduce kernels.  256 rows (16 loop trips per thread, four loads in flight at a time) suit
// the chip-filling tensors; on the small maps the launch is a handful of blocks whose time is those four dependent batches of loads, so
// they take 64 rows - one batch, every load of the thread in flight at once - on four times as many blocks.
// (reducing kernels end in same-address f64 atomics, ~17 ns apiece: they keep the row-block count per channel <= 64)
static inline int elt_rows_per_block(long long M, int C, bool reduces = false) {
  const long long blocks = (long long)cdiv(C, 64) * cdiv(M, ROWS_PER_BLOCK);
  int rpb = blocks >= 1024 ? ROWS_PER_BLOCK : (blocks >= 256 ? 128 : 64);
  while (reduces && rpb < ROWS_PER_BLOCK && cdiv(M, rpb) > 64) rpb *= 2;
  return rpb;
}

__device__ __forceinline__ float4 red_rowgroups(float4 v) {
  v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
  v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
  return v;
}
// block-level per-channel sum of a float4 held by every thread (fixed c4 = (tid&15)*4); result valid for tid<64
__device__ __forceinline__ float block_chan_sum(float4 v, float* sRed /*[256]*/, int tid) {
  v = red_rowgroups(v);
  __syncthreads();
  if ((tid & 63) < 16) *reinterpret_cast<float4*>(&sRed[(tid >> 6) * 64 + (tid & 15) * 4]) = v;
  __syncthreads();
  float r = 0.f;
  if (tid < 64) r = sRed[tid] + sRed[64 + tid] + sRed[128 + tid] + sRed[192 + tid];
  return r;
}

// ---------------------------------------------------------------- BN finalize (train forward)
// reference: nn.BatchNorm2d(momentum=0.01, eps=1e-3) in training mode
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* rmean, float* rvar, float momentum, float eps,
                                   float* scale, float* shift, float* mean_out, float* invstd_out, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean = stats[c] / count;
  double var = stats[C + c] / count - mean * mean;
  if (var < 0) var = 0;
  float invstd = mmd_bn_invstd(var, eps);
  float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = __fmaf_rn(-(float)mean, sc, beta[c]);      // explicit fma: must equal bn_live_coef (common.h) bit for bit
  if (mean_out) { mean_out[c] = (float)mean; invstd_out[c] = invstd; }
  if (rmean) {
    double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

extern "C" int mmd_bn_finalize(const double* stats, long long count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean_out, float* invstd_out, int C,
                               hipStream_t stream) {
  if (!stats || !gamma || !beta || !scale || !shift || C <= 0 || count <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, stats, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out, C);
  return mmd_check_launch();
}

// All BN layers of a net in ONE launch (end of the train-mode forward): stats_flat [2 x C_l per layer at 2*off_l],
// count[c] = rows that produced channel c's sums, layer_off[c] = channel offset of c's layer, layer_C[c] = its width.
__global__ void bn_finalize_all_kernel(const double* __restrict__ stats, const float* __restrict__ count,
                                       const int* __restrict__ layer_off, const int* __restrict__ layer_C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* rmean,
                                       float* rvar, float momentum, float eps, float* scale, float* shift, float* mean_out,
                                       float* invstd_out, int total, long long* nbt, int n_layers) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (nbt && c < n_layers) nbt[c] += 1;      // num_batches_tracked of every BatchNorm (n_layers <= total: a layer has at least one channel)
  if (c >= total) return;
  double n = (double)count[c];
  if (n <= 0.0) return;                     // padding channel or a layer that did not run this step
  int lo = layer_off[c], lc = layer_C[c], ci = c - lo;
  double mean = stats[2 * (size_t)lo + ci] / n;
  double var = stats[2 * (size_t)lo + lc + ci] / n - mean * mean;
  if (var < 0) var = 0;
  float invstd = mmd_bn_invstd(var, eps);
  float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = __fmaf_rn(-(float)mean, sc, beta[c]);      // explicit fma: must equal bn_live_coef (common.h) bit for bit
  mean_out[c] = (float)mean; invstd_out[c] = invstd;
  double unb = n > 1.0 ? var * n / (n - 1.0) : var;
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
}
extern "C" int mmd_bn_finalize_all(const double* stats_flat, const float* count, const int* layer_off, const int* layer_C,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* scale, float* shift, float* mean_out,
                                   float* invstd_out, int total, long long* num_batches_tracked, int n_layers, hipStream_t stream) {
  if (!stats_flat || !count || !layer_off || !layer_C || !gamma || !beta || !running_mean || !running_var || !scale ||
      !shift || !mean_out || !invstd_out || total <= 0 || (num_batches_tracked && (n_layers <= 0 || n_layers > total)))
    return MMD_EINVAL;
  hipLaunchKernelGGL(bn_finalize_all_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, stats_flat, count, layer_off,
                     layer_C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out, total,
                     num_batches_tracked, n_layers);
  return mmd_check_launch();
}

// eval-mode BN folded to scale/shift: scale = g/sqrt(rv+eps), shift = b - rm*scale
__global__ void bn_fold_kernel(const float* g, const float* b, const float* rm, const float* rv, float eps, float* scale,
                               float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc = g[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = b[c] - rm[c] * sc;
}
extern "C" int mmd_bn_fold(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                           float* scale, float* shift, int C, hipStream_t stream) {
  if (!gamma || !beta || !rmean || !rvar || !scale || !shift || C <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, rmean, rvar, eps, scale, shift, C);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- y = act(z*scale+shift)*rowscale[img] + res
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, BnLive bn, int act,
                                                         const float* __restrict__ rowscale, int rows_per_image,
                                                         const float* __restrict__ res, float* __restrict__ y, int M, int C,
                                                         int rpb) {
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  if (c >= C) return;
  float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
  if (bn.stats) bn_live_coef4(bn, c, sc, sh);
  else if (scale) { sc = mmd_ld4(scale + c); sh = mmd_ld4(shift + c); }
  const int r0 = blockIdx.y * rpb;
  const int r1 = min(M, r0 + rpb);
#pragma unroll 4
  for (int row = r0 + (tid >> 4); row < r1; row += 16) {
    size_t off = (size_t)row * C + c;
    float4 v = mmd_ld4(z + off);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
    else if (act == MMD_ACT_SIGMOID) { v.x = mmd_sigmoid(v.x); v.y = mmd_sigmoid(v.y); v.z = mmd_sigmoid(v.z); v.w = mmd_sigmoid(v.w); }
    if (rowscale) { float rs = rowscale[row / rows_per_image]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
    if (res) { float4 q = mmd_ld4(res + off); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    mmd_st4(y + off, v);
  }
}
extern "C" int mmd_affine_act(const float* z, const float* scale, const float* shift, const double* in_stats,
                              const float* in_gamma, const float* in_beta, long long in_count, int act,
                              const float* rowscale, int rows_per_image, const float* res, float* y, int M, int C,
                              hipStream_t stream) {
  if (!z || !y || M <= 0 || C <= 0 || (C & 3) || (rowscale && rows_per_image <= 0)) return MMD_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return MMD_EINVAL;
  if (in_stats && (scale || !in_gamma || !in_beta || in_count <= 0)) return MMD_EINVAL;
  mmd_prof_tag(MMD_FAM_ELT, "affine M%lld C%lld r%lld", M, C, (long long)(res?1:0), 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  const int rpb = elt_rows_per_block(M, C);
  hipLaunchKernelGGL(affine_act_kernel, dim3(cdiv(C, 64), cdiv(M, rpb)), dim3(256), 0, stream, z, scale, shift,
                     mmd_make_bn(in_stats, in_gamma, in_beta, in_count, C), act, rowscale, rows_per_image, res, y, M, C, rpb);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * M * (double)C * (res ? 3 : 2));
  return mmd_check_launch();
}

// ---------------------------------------------------------------- per-(image,channel) reductions
// out[b,c] += out_scale * sum_{rows of image b} (g ? g*a : a),  a = act(z*scale+shift)
// (squeeze-excite average pool forward; d(gate) in the backward)
__global__ __launch_bounds__(256) void chan_pool_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, BnLive bn, int act,
                                                        const float* __restrict__ g, float* __restrict__ out,
                                                        float out_scale, int rows_per_image, int C, int nsplit, int z16) {
  __shared__ float sRed[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  const int b = blockIdx.y;
  const bool cok = c < C;
  float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
  if (cok) {
    if (bn.stats) bn_live_coef4(bn, c, sc, sh);
    else if (scale) { sc = mmd_ld4(scale + c); sh = mmd_ld4(shift + c); }
  }
  float4 acc = make_float4(0, 0, 0, 0);
  if (cok) {
#pragma unroll 4
    for (int r = blockIdx.z * 16 + (tid >> 4); r < rows_per_image; r += 16 * nsplit) {
      size_t off = ((size_t)b * rows_per_image + r) * C + c;
      float4 v = mmd_ldw4(z, off, z16);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      if (act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
      if (g) { float4 q = mmd_ld4(g + off); v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w; }
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  float s = block_chan_sum(acc, sRed, tid);
  if (tid < 64 && blockIdx.x * 64 + tid < C) atomicAdd(&out[(size_t)b * C + blockIdx.x * 64 + tid], s * out_scale);
}
static int chan_pool_impl(const float* z, const float* scale, const float* shift, const double* in_stats,
                          const float* in_gamma, const float* in_beta, long long in_count, int act, const float* g,
                          float* out, float out_scale, int B, int rows_per_image, int C, int z16, hipStream_t stream) {
  if (!z || !out || B <= 0 || rows_per_image <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return MMD_EINVAL;
  if (in_stats && (scale || !in_gamma || !in_beta || in_count <= 0)) return MMD_EINVAL;
  int base = cdiv(C, 64) * B;
  int ns = cdiv(1024, base); int mx = cdiv(rows_per_image, 64); if (ns > mx) ns = mx; if (ns < 1) ns = 1;
  mmd_prof_tag(MMD_FAM_ELT, "cpool B%lld R%lld C%lld", B, rows_per_image, C, 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  hipLaunchKernelGGL(chan_pool_kernel, dim3(cdiv(C, 64), B, ns), dim3(256), 0, stream, z, scale, shift,
                     mmd_make_bn(in_stats, in_gamma, in_beta, in_count, C), act, g, out, out_scale, rows_per_image, C, ns, z16);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * B * (double)rows_per_image * C * (g ? 2 : 1));
  return mmd_check_launch();
}
extern "C" int mmd_chan_pool(const float* z, const float* scale, const float* shift, const double* in_stats,
                             const float* in_gamma, const float* in_beta, long long in_count, int act, const float* g,
                             float* out, float out_scale, int B, int rows_per_image, int C, hipStream_t stream) {
  return chan_pool_impl(z, scale, shift, in_stats, in_gamma, in_beta, in_count, act, g, out, out_scale, B, rows_per_image, C, 0, stream);
}
// z is a bf16 array (common.h w16)
static int mmd_chan_pool_w16(const float* z, const float* scale, const float* shift, const double* in_stats,
                                 const float* in_gamma, const float* in_beta, long long in_count, int act, const float* g,
                                 float* out, float out_scale, int B, int rows_per_image, int C, hipStream_t stream) {
  if (true && !MMD_W16_BUILD) return MMD_EINVAL;      // this build has the bf16-storage branches compiled out
  return chan_pool_impl(z, scale, shift, in_stats, in_gamma, in_beta, in_count, act, g, out, out_scale, B, rows_per_image, C, 1, stream);
}

// Backward companion of the squeeze-excite pool: ONE pass over (z1, g1) yields, per (image, channel), everything the
// SE backward and the BatchNorm-1 backward need, so no separate BN reduce pass reads the expanded tensor again:
//   out[0] = sum g1*a      (d gate)            a  = swish(u), u = z*scale+shift
//   out[1] = sum g1*s'     out[2] = sum g1*s'*xhat      s' = swish'(u), xhat = (z-mean)*invstd
//   out[3] = sum s'        out[4] = sum s'*xhat
// With g = (g1*gate + dpooled)*s' (the BN-1 upstream gradient):  sum g = gate*out[1] + dpooled*out[3] summed over images,
// sum g*xhat = gate*out[2] + dpooled*out[4]  (finished in se_bwd_b_kernel once dpooled is known).   out: [5][B][C], zeroed.
__global__ __launch_bounds__(256) void chan_pool_bwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ g1,
                                                            float* __restrict__ out, int B, int rows_per_image, int C,
                                                            int nsplit) {
  __shared__ float sRed[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  const int b = blockIdx.y;
  const bool cok = c < C;
  float acc[5][4];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
  if (cok) {
    const float4 sc = mmd_ld4(scale + c), sh = mmd_ld4(shift + c), mu = mmd_ld4(mean + c), is = mmd_ld4(invstd + c);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll 2
    for (int r = blockIdx.z * 16 + (tid >> 4); r < rows_per_image; r += 16 * nsplit) {
      size_t off = ((size_t)b * rows_per_image + r) * C + c;
      const float4 zz = mmd_ld4(z + off), gg = mmd_ld4(g1 + off);
      const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u = zv[i] * scv[i] + shv[i];
        float sg = mmd_sigmoid(u);
        float sp = sg * (1.0f + u * (1.0f - sg));
        float xh = (zv[i] - muv[i]) * isv[i];
        acc[0][i] += gv[i] * (u * sg);
        acc[1][i] += gv[i] * sp;
        acc[2][i] += gv[i] * sp * xh;
        acc[3][i] += sp;
        acc[4][i] += sp * xh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float s = block_chan_sum(make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]), sRed, tid);
    if (tid < 64 && blockIdx.x * 64 + tid < C) atomicAdd(&out[((size_t)k * B + b) * C + blockIdx.x * 64 + tid], s);
    __syncthreads();
  }
}
extern "C" int mmd_chan_pool_bwd(const float* z, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const float* g1, float* out5, int B, int rows_per_image, int C,
                                 hipStream_t stream) {
  if (!z || !scale || !shift || !mean || !invstd || !g1 || !out5 || B <= 0 || rows_per_image <= 0 || C <= 0 || (C & 3))
    return MMD_EINVAL;
  int base = cdiv(C, 64) * B;
  int ns = cdiv(1024, base); int mx = cdiv(rows_per_image, 64); if (ns > mx) ns = mx; if (ns < 1) ns = 1;
  mmd_prof_tag(MMD_FAM_ELT, "cpoolbwd B%lld R%lld C%lld", B, rows_per_image, C, 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  hipLaunchKernelGGL(chan_pool_bwd_kernel, dim3(cdiv(C, 64), B, ns), dim3(256), 0, stream, z, scale, shift, mean, invstd, g1,
                     out5, B, rows_per_image, C, ns);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * B * (double)rows_per_image * C * 2);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- squeeze-excite FCs
// reference: src/YetAnotherEfficientNet.py:469-474.  wr [S,C], br [S], be [C]; the expand weight is kept TRANSPOSED,
// wet [S,C] (native layout of the engine), so every access below is coalesced over channels.
// Two small launches with enough blocks to fill the chip (one block per image was 50 us per call):
//   hidden: one wave per (image, j): hpre[b,j] = wr[j,:].pooled[b,:] + br[j]        (coalesced over C)
//   gate  : one wave per 16 channels: gate[b,c] = sigmoid(we[c,:].swish(hpre[b,:]) + be[c])   (coalesced over S)
static thread_local MmdGroup g_group{1, 0, 0, 0};
static thread_local bool g_group_read = false;
const MmdGroup& mmd_group() { g_group_read = true; return g_group; }
// Grouped frozen nets (common.h MmdGroup): the launches the CALLING THREAD issues after this call cover n_groups nets x images_per_group
// images each; n_groups <= 1 switches the mode off.  Thread-local host-side state (captured into a graph by value): launches of other host
// threads never see it.  Switching a group off that no launch has read returns MMD_EINVAL: an entry point without a group mode ran under
// it, i.e. with the first net's parameters for every image - the caller must not use that result.
extern "C" int mmd_set_group(int n_groups, int images_per_group, long long w_stride, long long bn_stride) {
  if (n_groups <= 1) {
    const bool unread = g_group.n > 1 && !g_group_read;
    g_group = MmdGroup{1, 0, 0, 0};
    return unread ? MMD_EINVAL : MMD_OK;
  }
  if (images_per_group <= 0 || w_stride <= 0 || bn_stride <= 0 || (w_stride & 3) || (bn_stride & 3)) return MMD_EINVAL;
  g_group = MmdGroup{n_groups, images_per_group, w_stride, bn_stride};
  g_group_read = false;
  return MMD_OK;
}

// Q: `pooled` holds the frozen nets' Q36 fixed-point pool sums (common.h mmd_pool_add) instead of floats
template <bool Q>
__global__ __launch_bounds__(256) void se_hidden_kernel(const void* __restrict__ pooled_, const float* __restrict__ wr,
                                                        const float* __restrict__ br, float* __restrict__ hpre, int C, int S,
                                                        int g_images, long long g_w) {
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.y * 4 + wave;
  if (j >= S) return;
  if (g_images) { const size_t o = (size_t)(b / g_images) * g_w; wr += o; br += o; }
  const float* p = reinterpret_cast<const float*>(pooled_) + (size_t)b * C;
  const long long* pq = reinterpret_cast<const long long*>(pooled_) + (size_t)b * C;
  const float* w = wr + (size_t)j * C;
  float acc = 0.f;
#pragma unroll 4
  for (int c = lane * 4; c < C; c += 256) {
    float4 a;
    if (Q) {
      const longlong2 q0 = *reinterpret_cast<const longlong2*>(pq + c), q1 = *reinterpret_cast<const longlong2*>(pq + c + 2);
      a = make_float4(mmd_pool_get(q0.x), mmd_pool_get(q0.y), mmd_pool_get(q1.x), mmd_pool_get(q1.y));
    } else {
      a = mmd_ld4(p + c);
    }
    const float4 q = mmd_ld4(w + c);
    acc += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) hpre[(size_t)b * S + j] = acc + br[j];
}
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ hpre, const float* __restrict__ wet,
                                                      const float* __restrict__ be, float* __restrict__ gate, int C, int S,
                                                      int g_images, long long g_w) {
  __shared__ float sh[256];
  const int b = blockIdx.x;
  if (g_images) { const size_t o = (size_t)(b / g_images) * g_w; wet += o; be += o; }
  for (int j = threadIdx.x; j < S; j += 256) sh[j] = mmd_swish(hpre[(size_t)b * S + j]);
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  float acc = be[c];
#pragma unroll 8
  for (int j = 0; j < S; ++j) acc += wet[(size_t)j * C + c] * sh[j];      // wet [S][C]: coalesced over c
  gate[(size_t)b * C + c] = mmd_sigmoid(acc);
}
static int se_fc_fwd_impl(const void* pooled, bool q, const float* wr, const float* br, const float* we, const float* be,
                          float* hpre, float* gate, int B, int C, int S, hipStream_t stream) {
  if (!pooled || !wr || !br || !we || !be || !hpre || !gate || B <= 0 || C <= 0 || (C & 3) || S <= 0 || S > 256) return MMD_EINVAL;
  const MmdGroup& gr = mmd_group();
  const int gi = gr.n > 1 ? gr.images : 0;
  if (gi && B != gr.n * gr.images) return MMD_EINVAL;
  // algorithmic bytes: both FC matrices once per group, the pooled / hidden / gate vectors
  mmd_prof_tag(MMD_FAM_SE, "sefwd B%lld C%lld S%lld q%lld", B, C, S, q ? 1 : 0);
  mmd_prof_begin(MMD_FAM_SE, stream);
  if (q) hipLaunchKernelGGL(se_hidden_kernel<true>, dim3(B, cdiv(S, 4)), dim3(256), 0, stream, pooled, wr, br, hpre, C, S, gi, gr.w_stride);
  else hipLaunchKernelGGL(se_hidden_kernel<false>, dim3(B, cdiv(S, 4)), dim3(256), 0, stream, pooled, wr, br, hpre, C, S, gi, gr.w_stride);
  hipLaunchKernelGGL(se_gate_kernel, dim3(B, cdiv(C, 256)), dim3(256), 0, stream, hpre, we, be, gate, C, S, gi, gr.w_stride);
  mmd_prof_end(MMD_FAM_SE, stream, 4.0 * B * (double)C * S, 4.0 * ((gr.n > 1 ? gr.n : 1) * 2.0 * C * S + (double)B * ((q ? 3.0 : 2.0) * C + 2.0 * S)));
  return mmd_check_launch();
}
extern "C" int mmd_se_fc_fwd(const float* pooled, const float* wr, const float* br, const float* we, const float* be,
                             float* hpre, float* gate, int B, int C, int S, hipStream_t stream) {
  return se_fc_fwd_impl(pooled, false, wr, br, we, be, hpre, gate, B, C, S, stream);
}
// the frozen nets' form: pooled_q [B, C] = the Q36 fixed-point pool sums the depthwise / fused expand+depthwise epilogues accumulate
extern "C" int mmd_se_fc_fwd_q(const long long* pooled_q, const float* wr, const float* br, const float* we, const float* be,
                               float* hpre, float* gate, int B, int C, int S, hipStream_t stream) {
  return se_fc_fwd_impl(pooled_q, true, wr, br, we, be, hpre, gate, B, C, S, stream);
}

// backward, step 1a: dpe[b,c] = dgate*gate*(1-gate) (recomputed per wave); dh[b,j] = sum_c wet[j,c]*dpe[b,c], one wave per (b,j)
__global__ __launch_bounds__(256) void se_bwd_a_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                       const float* __restrict__ wet, float* __restrict__ dpe, float* dh,
                                                       int C, int S) {
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.y * 4 + wave;
  if (j >= S) return;
  // channel quads, four trips' loads in flight together: the launch is ~100 waves of a few KB each - its time is its dependent round trips
  const float* gp = gate + (size_t)b * C;
  const float* dp = dgate + (size_t)b * C;
  const float* wp = wet + (size_t)j * C;
  float acc = 0.f;
#pragma unroll 4
  for (int c = lane * 4; c < C; c += 256) {
    const float4 g = mmd_ld4(gp + c), dg = mmd_ld4(dp + c), w = mmd_ld4(wp + c);
    const float4 d = make_float4(dg.x * g.x * (1.f - g.x), dg.y * g.y * (1.f - g.y), dg.z * g.z * (1.f - g.z), dg.w * g.w * (1.f - g.w));
    if (j == 0) mmd_st4(dpe + (size_t)b * C + c, d);
    acc += w.x * d.x + w.y * d.y + w.z * d.z + w.w * d.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) dh[(size_t)b * S + j] = acc;
}
// step 1b: dpr[b,j] = dh[b,j]*swish'(hpre[b,j]); dpooled[b,c] = dpool_scale * sum_j wr[j,c]*dpr[b,j]
__global__ __launch_bounds__(256) void se_bwd_b_kernel(const float* __restrict__ dh, const float* __restrict__ hpre,
                                                       const float* __restrict__ wr, float* __restrict__ dpr,
                                                       float* __restrict__ dpooled, float dpool_scale, int C, int S,
                                                       const float* __restrict__ gate, const float* __restrict__ pool5,
                                                       double* bn_sums, int B) {
  __shared__ float sd[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < S; j += 256) {
    float d = dh[(size_t)b * S + j] * mmd_swish_grad(hpre[(size_t)b * S + j]);
    sd[j] = d;
    if (blockIdx.y == 0) dpr[(size_t)b * S + j] = d;
  }
  __syncthreads();
  const int c = blockIdx.y * 256 + tid;
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll 8
  for (int j = 0; j < S; ++j) acc += wr[(size_t)j * C + c] * sd[j];
  const float dp = acc * dpool_scale;
  dpooled[(size_t)b * C + c] = dp;
  if (bn_sums) {     // BatchNorm-1 sums from the pooled partials (chan_pool_bwd_kernel): B adds per address
    const size_t i = (size_t)b * C + c, n = (size_t)B * C;
    const float gt = gate[i];
    atomicAdd(&bn_sums[c], (double)(gt * pool5[n + i] + dp * pool5[3 * n + i]));
    atomicAdd(&bn_sums[C + c], (double)(gt * pool5[2 * n + i] + dp * pool5[4 * n + i]));
  }
}
// step 2 (grid over weights, sums over the batch, no atomics):
//   dwe[c,j] += sum_b dpe[b,c]*swish(hpre[b,j]); dbe[c] += sum_b dpe[b,c]
//   dwr[j,c] += sum_b dpr[b,j]*pooled[b,c];      dbr[j] += sum_b dpr[b,j]
__global__ void se_fc_wgrad_kernel(const float* __restrict__ dpe, const float* __restrict__ dpr,
                                   const float* __restrict__ hpre, const float* __restrict__ pooled, float* dwr, float* dbr,
                                   float* dwe, float* dbe, int B, int C, int S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * S) return;
  {  // dwe laid out [S][C] (transposed native layout)
    int j = i / C, c = i % C;
    float acc = 0.f, accb = 0.f;
    for (int b = 0; b < B; ++b) {
      float d = dpe[(size_t)b * C + c];
      acc += d * mmd_swish(hpre[(size_t)b * S + j]);
      accb += d;
    }
    dwe[i] += acc;
    if (j == 0) dbe[c] += accb;
  }
  {  // dwr laid out [S][C]
    int j = i / C, c = i % C;
    float acc = 0.f, accb = 0.f;
    for (int b = 0; b < B; ++b) {
      float d = dpr[(size_t)b * S + j];
      acc += d * pooled[(size_t)b * C + c];
      accb += d;
    }
    dwr[i] += acc;
    if (c == 0) dbr[j] += accb;
  }
}
extern "C" int mmd_se_fc_bwd(const float* dgate, const float* gate, const float* hpre, const float* pooled,
                             const float* wr, const float* we, float* dpe_ws, float* dpr_ws, float* dh_zeroed,
                             float* dpooled, float dpool_scale, float* dwr, float* dbr, float* dwe, float* dbe, int B,
                             int C, int S, const float* pool5, double* bn_sums, hipStream_t stream) {
  if (!dgate || !gate || !hpre || !pooled || !wr || !we || !dpe_ws || !dpr_ws || !dh_zeroed || !dpooled) return MMD_EINVAL;
  if (dwr && (!dbr || !dwe || !dbe)) return MMD_EINVAL;         // dwr == NULL: weight gradients left to mmd_se_fc_wgrad
  if (B <= 0 || C <= 0 || (C & 3) || S <= 0 || S > 256 || ((pool5 == nullptr) != (bn_sums == nullptr))) return MMD_EINVAL;
  mmd_prof_tag(MMD_FAM_SE, "sebwd B%lld C%lld S%lld", B, C, S, 0);
  mmd_prof_begin(MMD_FAM_SE, stream);
  hipLaunchKernelGGL(se_bwd_a_kernel, dim3(B, cdiv(S, 4)), dim3(256), 0, stream, dgate, gate, we, dpe_ws, dh_zeroed, C, S);
  hipLaunchKernelGGL(se_bwd_b_kernel, dim3(B, cdiv(C, 256)), dim3(256), 0, stream, dh_zeroed, hpre, wr, dpr_ws, dpooled,
                     dpool_scale, C, S, gate, pool5, bn_sums, B);
  if (dwr)
    hipLaunchKernelGGL(se_fc_wgrad_kernel, dim3(cdiv((long long)C * S, 256)), dim3(256), 0, stream, dpe_ws, dpr_ws, hpre,
                       pooled, dwr, dbr, dwe, dbe, B, C, S);
  mmd_prof_end(MMD_FAM_SE, stream, 4.0 * B * (double)C * S, 4.0 * (2.0 * C * S + (double)B * (4.0 * C + 3.0 * S)));
  return mmd_check_launch();
}
// Both FC layers' backward in ONE launch (round 3: se_bwd_a + se_bwd_b were two dependent ~10 us launches on the backward's serial chain per
// MBConv block).  Grid (B, ceil(C / 256)); every block first recomputes its image's hidden gradient - dpe[c] = dgate*gate*(1-gate) for all C
// into LDS, then dh[j] = sum_c wet[j,c]*dpe[c] one wave per j (coalesced over c, S*C MACs per block: at most 186 K) - and then runs step 1b
// for its 256 channels.  Same outputs as the two-kernel form (dpe, dpr, dpooled, BatchNorm-1 sums); dh is not materialised.
#define SE_MAXC 3072
__global__ __launch_bounds__(256) void se_bwd_ab_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                        const float* __restrict__ hpre, const float* __restrict__ wr,
                                                        const float* __restrict__ wet, float* __restrict__ dpe, float* __restrict__ dpr,
                                                        float* __restrict__ dpooled, float dpool_scale, int C, int S,
                                                        const float* __restrict__ pool5, double* bn_sums, int B) {
  __shared__ float sdpe[SE_MAXC];
  __shared__ float sd[256];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int c = tid; c < C; c += 256) {
    const float g = gate[(size_t)b * C + c];
    const float d = dgate[(size_t)b * C + c] * g * (1.f - g);
    sdpe[c] = d;
    if (blockIdx.y == 0) dpe[(size_t)b * C + c] = d;
  }
  __syncthreads();
  for (int j = wave; j < S; j += 4) {
    const float* w = wet + (size_t)j * C;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w[c] * sdpe[c];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float d = acc * mmd_swish_grad(hpre[(size_t)b * S + j]);
      sd[j] = d;
      if (blockIdx.y == 0) dpr[(size_t)b * S + j] = d;
    }
  }
  __syncthreads();
  const int c = blockIdx.y * 256 + tid;
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll 8
  for (int j = 0; j < S; ++j) acc += wr[(size_t)j * C + c] * sd[j];
  const float dp = acc * dpool_scale;
  dpooled[(size_t)b * C + c] = dp;
  if (bn_sums) {
    const size_t i = (size_t)b * C + c, n = (size_t)B * C;
    const float gt = gate[i];
    atomicAdd(&bn_sums[c], (double)(gt * pool5[n + i] + dp * pool5[3 * n + i]));
    atomicAdd(&bn_sums[C + c], (double)(gt * pool5[2 * n + i] + dp * pool5[4 * n + i]));
  }
}
extern "C" int mmd_se_fc_bwd_fused(const float* dgate, const float* gate, const float* hpre, const float* wr, const float* we,
                                   float* dpe_ws, float* dpr_ws, float* dpooled, float dpool_scale, int B, int C, int S,
                                   const float* pool5, double* bn_sums, hipStream_t stream) {
  if (!dgate || !gate || !hpre || !wr || !we || !dpe_ws || !dpr_ws || !dpooled) return MMD_EINVAL;
  if (B <= 0 || C <= 0 || C > SE_MAXC || S <= 0 || S > 256 || ((pool5 == nullptr) != (bn_sums == nullptr))) return MMD_EINVAL;
  mmd_prof_tag(MMD_FAM_SE, "sebwdf B%lld C%lld S%lld", B, C, S, 0);
  mmd_prof_begin(MMD_FAM_SE, stream);
  hipLaunchKernelGGL(se_bwd_ab_kernel, dim3(B, cdiv(C, 256)), dim3(256), 0, stream, dgate, gate, hpre, wr, we, dpe_ws, dpr_ws, dpooled,
                     dpool_scale, C, S, pool5, bn_sums, B);
  mmd_prof_end(MMD_FAM_SE, stream, 4.0 * B * (double)C * S, 4.0 * (2.0 * C * S + (double)B * (4.0 * C + 3.0 * S)));
  return mmd_check_launch();
}

// The FC weight gradients of EVERY squeeze-excite block of a backward segment in one launch (they are leaves: 23 launches of ~9 us, each
// behind its own cross-stream edge).  desc: n entries {dpe, dpr, hpre, pooled, dwr, dbr, dwe, dbe, C, S} (8 pointers + 2 ints as 10 x 8 bytes).
struct SeWgDesc { const float* dpe; const float* dpr; const float* hpre; const float* pooled; float* dwr; float* dbr; float* dwe; float* dbe;
                  long long C; long long S; };
__global__ void se_fc_wgrad_batched_kernel(const SeWgDesc* __restrict__ desc, int B) {
  const SeWgDesc d = desc[blockIdx.y];
  const int C = (int)d.C, S = (int)d.S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * S) return;
  const int j = i / C, c = i % C;
  float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;
  for (int b = 0; b < B; ++b) {
    // (operand pointers out of a device table: global loads / stores made explicit, common.h mmd_ldg)
    const float pe = mmd_ldg(d.dpe + (size_t)b * C + c), pr = mmd_ldg(d.dpr + (size_t)b * S + j);
    a1 += pe * mmd_swish(mmd_ldg(d.hpre + (size_t)b * S + j)); b1 += pe;
    a2 += pr * mmd_ldg(d.pooled + (size_t)b * C + c); b2 += pr;
  }
  mmd_stg(d.dwe + i, mmd_ldg(d.dwe + i) + a1);
  if (j == 0) mmd_stg(d.dbe + c, mmd_ldg(d.dbe + c) + b1);
  mmd_stg(d.dwr + i, mmd_ldg(d.dwr + i) + a2);
  if (c == 0) mmd_stg(d.dbr + j, mmd_ldg(d.dbr + j) + b2);
}
extern "C" int mmd_se_fc_wgrad_batched(const void* desc, int n, int max_cs, int B, hipStream_t stream) {
  if (!desc || n <= 0 || max_cs <= 0 || B <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(se_fc_wgrad_batched_kernel, dim3(cdiv(max_cs, 256), n), dim3(256), 0, stream, (const SeWgDesc*)desc, B);
  return mmd_check_launch();
}

// the weight-gradient half of mmd_se_fc_bwd on its own (a leaf of the backward graph: the engine issues it on the wgrad stream)
extern "C" int mmd_se_fc_wgrad(const float* dpe, const float* dpr, const float* hpre, const float* pooled, float* dwr, float* dbr,
                               float* dwe, float* dbe, int B, int C, int S, hipStream_t stream) {
  if (!dpe || !dpr || !hpre || !pooled || !dwr || !dbr || !dwe || !dbe || B <= 0 || C <= 0 || S <= 0 || S > 256) return MMD_EINVAL;
  hipLaunchKernelGGL(se_fc_wgrad_kernel, dim3(cdiv((long long)C * S, 256)), dim3(256), 0, stream, dpe, dpr, hpre, pooled, dwr,
                     dbr, dwe, dbe, B, C, S);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- slotted BatchNorm sums (see common.h)
// one wave per sum: lane = slot, so the `slots` loads of an address are in flight together (a per-thread loop over the
// slots serialises on load->store ordering: ~1 us per slot)
__global__ __launch_bounds__(256) void stats_fold_kernel(double* stats, double* ws, int slots, int n) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0;
  for (int k = lane; k < slots; k += 64) { s += ws[(size_t)k * n + i]; ws[(size_t)k * n + i] = 0; }
  s = wave_sum_d(s);
  if (lane == 0) stats[i] += s;
}
int mmd_stats_fold(double* stats, double* ws, int slots, int n, hipStream_t stream) {
  hipLaunchKernelGGL(stats_fold_kernel, dim3(cdiv(n, 4)), dim3(256), 0, stream, stats, ws, slots, n);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- BN (+activation) backward
// pass 1: g = (g_in * mul_bc[img,c] * mul_b[img] + add_bc[img,c]) * act'(z*scale+shift);
//         sums[c] += g, sums[C+c] += g * xhat,  xhat = (z-mean)*invstd          
//         g is written only when g_out != nullptr: pass 2 can recompute it from g_in (one HBM write less per layer)
__device__ __forceinline__ float4 bn_bwd_g(float4 g, const float4& zz, const float4& sc, const float4& sh, int act,
                                           const float* __restrict__ mul_bc, const float* __restrict__ mul_b,
                                           const float* __restrict__ add_bc, int img, int C, int c) {
  if (mul_bc) { float4 m = mmd_ld4(mul_bc + (size_t)img * C + c); g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w; }
  if (mul_b) { float m = mul_b[img]; g.x *= m; g.y *= m; g.z *= m; g.w *= m; }
  if (add_bc) { float4 m = mmd_ld4(add_bc + (size_t)img * C + c); g.x += m.x; g.y += m.y; g.z += m.z; g.w += m.w; }
  if (act == MMD_ACT_SWISH) {
    g.x *= mmd_swish_grad(zz.x * sc.x + sh.x); g.y *= mmd_swish_grad(zz.y * sc.y + sh.y);
    g.z *= mmd_swish_grad(zz.z * sc.z + sh.z); g.w *= mmd_swish_grad(zz.w * sc.w + sh.w);
  }
  return g;
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g_in, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            int act, const float* __restrict__ mul_bc,
                                                            const float* __restrict__ mul_b, const float* __restrict__ add_bc,
                                                            int rows_per_image, float* __restrict__ g_out, double* sums,
                                                            int M, int C, Pyr pyr, long long lev_stride, int rpb,
                                                            double* ws, int slots) {
  __shared__ float sRed[256];
  if (ws) sums = ws + (size_t)(blockIdx.y % slots) * 2 * C;
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  const bool cok = c < C;
  const int r0 = blockIdx.y * rpb;
  int r1 = min(M, r0 + rpb);
  if (pyr.n) {                       // per-level BN parameters / sums; rows past the level's valid count are padding
    const int lev = pyr_level_of_row(pyr, r0);
    scale += lev * lev_stride; shift += lev * lev_stride; mean += lev * lev_stride; invstd += lev * lev_stride;
    sums += 2 * lev * lev_stride;
    r1 = min(r1, pyr.row0[lev] + pyr.B * pyr.H[lev] * pyr.W[lev]);
  }
  float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0), mu = make_float4(0, 0, 0, 0), is = make_float4(1, 1, 1, 1);
  if (cok) { sc = mmd_ld4(scale + c); sh = mmd_ld4(shift + c); mu = mmd_ld4(mean + c); is = mmd_ld4(invstd + c); }
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  if (cok) {
  #pragma unroll 4
  for (int row = r0 + (tid >> 4); row < r1; row += 16) {
      size_t off = (size_t)row * C + c;
      float4 g = mmd_ld4(g_in + off), zz = mmd_ld4(z + off);
      int img = (mul_bc || mul_b || add_bc) ? row / rows_per_image : 0;
      g = bn_bwd_g(g, zz, sc, sh, act, mul_bc, mul_b, add_bc, img, C, c);
      if (g_out) mmd_st4(g_out + off, g);
      s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
      s2.x += g.x * (zz.x - mu.x) * is.x; s2.y += g.y * (zz.y - mu.y) * is.y;
      s2.z += g.z * (zz.z - mu.z) * is.z; s2.w += g.w * (zz.w - mu.w) * is.w;
    }
  }
  float a = block_chan_sum(s1, sRed, tid);
  float b2 = block_chan_sum(s2, sRed, tid);
  if (tid < 64 && blockIdx.x * 64 + tid < C) {
    atomicAdd(&sums[blockIdx.x * 64 + tid], (double)a);
    atomicAdd(&sums[C + blockIdx.x * 64 + tid], (double)b2);
  }
}
extern "C" int mmd_bn_bwd_reduce(const float* g_in, const float* z, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, int act, const float* mul_bc,
                                 const float* mul_b, const float* add_bc, int rows_per_image, float* g_out,
                                 double* sums, int M, int C, double* stats_ws, int ws_slots, hipStream_t stream) {
  if (!g_in || !z || !scale || !shift || !mean || !invstd || !sums || M <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((mul_bc || mul_b || add_bc) && rows_per_image <= 0) return MMD_EINVAL;
  mmd_prof_tag(MMD_FAM_ELT, "bnred M%lld C%lld o%lld", M, C, (long long)(g_out?1:0), 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  const bool slotted = stats_ws && ws_slots > 1 && cdiv(M, ROWS_PER_BLOCK) > MMD_STATS_DEPTH;
  const int rpb = elt_rows_per_block(M, C, true);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C, 64), cdiv(M, rpb)), dim3(256), 0, stream, g_in, z, scale,
                     shift, mean, invstd, act, mul_bc, mul_b, add_bc, rows_per_image, g_out, sums, M, C, Pyr{}, 0, rpb,
                     slotted ? stats_ws : nullptr, ws_slots);
  if (slotted) mmd_stats_fold(sums, stats_ws, ws_slots, 2 * C, stream);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * M * (double)C * (g_out ? 3 : 2));
  return mmd_check_launch();
}

// pass 2: dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); block row 0 also does dgamma += sum(g*xhat), dbeta += sum(g)
//         g is either given (act NONE, no modifiers) or recomputed from g_in exactly as pass 1 did
struct BnBwdMod { const float* scale; const float* shift; int act; const float* mul_bc; const float* mul_b; const float* add_bc;
                  int rows_per_image; };
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           double count, float* __restrict__ dz, float* dgamma, float* dbeta,
                                                           int M, int C, Pyr pyr, long long lev_stride, int rpb, BnBwdMod mod) {
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  if (c >= C) return;
  const int r0 = blockIdx.y * rpb;
  int r1 = min(M, r0 + rpb);
  bool first = blockIdx.y == 0;
  if (pyr.n) {
    const int lev = pyr_level_of_row(pyr, r0);
    mean += lev * lev_stride; invstd += lev * lev_stride; gamma += lev * lev_stride; sums += 2 * lev * lev_stride;
    if (mod.scale) { mod.scale += lev * lev_stride; mod.shift += lev * lev_stride; }
    if (dgamma) { dgamma += lev * lev_stride; dbeta += lev * lev_stride; }
    count = (double)pyr.B * pyr.H[lev] * pyr.W[lev];
    r1 = min(r1, pyr.row0[lev] + (int)count);
    first = r0 == pyr.row0[lev];
  }
  float4 mu = mmd_ld4(mean + c), is = mmd_ld4(invstd + c), ga = mmd_ld4(gamma + c);
  float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
  if (mod.act != MMD_ACT_NONE) { sc = mmd_ld4(mod.scale + c); sh = mmd_ld4(mod.shift + c); }
  const bool per_img = mod.mul_bc || mod.mul_b || mod.add_bc;
  float m1[4], m2[4];
  // (one f64 reciprocal instead of eight f64 divisions in every block's preamble: the other BatchNorm-backward consumers - BnBwdOp,
  // NodeGemm - multiply by 1 / count as well)
  const double inv_count = 1.0 / count;
#pragma unroll
  for (int i = 0; i < 4; ++i) { m1[i] = (float)(sums[c + i] * inv_count); m2[i] = (float)(sums[C + c + i] * inv_count); }
  if (first && (tid >> 4) == 0 && dgamma) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dgamma[c + i] += (float)sums[C + c + i]; dbeta[c + i] += (float)sums[c + i]; }
  }
#pragma unroll 4
  for (int row = r0 + (tid >> 4); row < r1; row += 16) {
    size_t off = (size_t)row * C + c;
    float4 gg = mmd_ld4(g + off), zz = mmd_ld4(z + off), o;
    gg = bn_bwd_g(gg, zz, sc, sh, mod.act, mod.mul_bc, mod.mul_b, mod.add_bc, per_img ? row / mod.rows_per_image : 0, C, c);
    o.x = ga.x * is.x * (gg.x - m1[0] - (zz.x - mu.x) * is.x * m2[0]);
    o.y = ga.y * is.y * (gg.y - m1[1] - (zz.y - mu.y) * is.y * m2[1]);
    o.z = ga.z * is.z * (gg.z - m1[2] - (zz.z - mu.z) * is.z * m2[2]);
    o.w = ga.w * is.w * (gg.w - m1[3] - (zz.w - mu.w) * is.w * m2[3]);
    mmd_st4(dz + off, o);
  }
}
extern "C" int mmd_bn_bwd_apply(const float* g, const float* z, const float* mean, const float* invstd,
                                const float* gamma, const double* sums, long long count, float* dz, float* dgamma,
                                float* dbeta, int M, int C, const float* scale, const float* shift, int act,
                                const float* mul_bc, const float* mul_b, const float* add_bc, int rows_per_image,
                                hipStream_t stream) {
  if (!g || !z || !mean || !invstd || !gamma || !sums || !dz || M <= 0 || C <= 0 || (C & 3) || count <= 0) return MMD_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  if (act != MMD_ACT_NONE && (act != MMD_ACT_SWISH || !scale || !shift)) return MMD_EINVAL;
  if ((mul_bc || mul_b || add_bc) && rows_per_image <= 0) return MMD_EINVAL;
  mmd_prof_tag(MMD_FAM_ELT, "bnapp M%lld C%lld", M, C, 0, 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  const int rpb = elt_rows_per_block(M, C);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(C, 64), cdiv(M, rpb)), dim3(256), 0, stream, g, z, mean,
                     invstd, gamma, sums, (double)count, dz, dgamma, dbeta, M, C, Pyr{}, 0, rpb,
                     BnBwdMod{scale, shift, act, mul_bc, mul_b, add_bc, rows_per_image});
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * M * (double)C * 3);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- column sums (bias gradients): out[c] += sum_rows a[row,c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* out, int M, int C) {
  __shared__ float sRed[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  const int r0 = blockIdx.y * ROWS_PER_BLOCK, r1 = min(M, r0 + ROWS_PER_BLOCK);
  if (c < C)
  #pragma unroll 4
  for (int row = r0 + (tid >> 4); row < r1; row += 16) {
      float4 v = mmd_ld4(a + (size_t)row * C + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  float s = block_chan_sum(acc, sRed, tid);
  if (tid < 64 && blockIdx.x * 64 + tid < C) atomicAdd(&out[blockIdx.x * 64 + tid], s);
}
extern "C" int mmd_colsum(const float* a, float* out, int M, int C, hipStream_t stream) {
  if (!a || !out || M <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), cdiv(M, ROWS_PER_BLOCK)), dim3(256), 0, stream, a, out, M, C);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- stem im2col: NCHW image -> [B*OH*OW, Kp] rows
// (3x3 stride-2 TF-SAME conv as a GEMM; k index = ci*9 + i*3 + j matches the [Cout,Cin,3,3] weight layout)
__global__ void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int Cin, int H, int W, int OH,
                                   int OW, int pad_t, int pad_l, int Kp) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)B * OH * OW * Kp;
  if (idx >= total) return;
  int k = (int)(idx % Kp); size_t m = idx / Kp;
  int ow = (int)(m % OW); m /= OW;
  int oh = (int)(m % OH); int b = (int)(m / OH);
  float v = 0.f;
  if (k < Cin * 9) {
    int ci = k / 9, t = k % 9, i = t / 3, j = t % 3;
    int ih = oh * 2 + i - pad_t, iw = ow * 2 + j - pad_l;
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((size_t)b * Cin + ci) * H + ih) * W + iw];
  }
  col[idx] = v;
}
extern "C" int mmd_stem_im2col(const float* x, float* col, int B, int Cin, int H, int W, int Kp, hipStream_t stream) {
  if (!x || !col || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Kp < Cin * 9 || (Kp & 3)) return MMD_EINVAL;
  int OH = (H + 1) / 2, OW = (W + 1) / 2;
  int eh = (OH - 1) * 2 - H + 3, ew = (OW - 1) * 2 - W + 3;
  if (eh < 0) eh = 0; if (ew < 0) ew = 0;
  size_t total = (size_t)B * OH * OW * Kp;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, col, B, Cin, H, W, OH, OW,
                     eh / 2, ew / 2, Kp);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- misc
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ dl, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float q = p[i]; dl[i] = dp[i] * q * (1.f - q); }
}
// dlogit = dprob * p * (1-p)
extern "C" int mmd_sigmoid_bwd(const float* dprob, const float* prob, float* dlogit, long long n, hipStream_t stream) {
  if (!dprob || !prob || !dlogit || n <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dprob, prob, dlogit, (size_t)n);
  return mmd_check_launch();
}

// gather [B, A_total, C] head-gradient rows of one pyramid level into a dense [B*HW*?, C'] matrix and back is
// avoided: the header GEMMs read/write the concatenated buffer through (batch_stride, offset) directly.
// This helper copies a level slice out of the concatenated tensor: dst[b*rows + r, :] = src[b*bs + off + r*N ...]
__global__ void slice_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int rows, int N,
                                  long long bstride, long long off) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t per = (size_t)rows * N;
  if (i >= per * B) return;
  int b = (int)(i / per); size_t r = i % per;
  dst[i] = src[(size_t)b * bstride + off + r];
}
extern "C" int mmd_slice_rows(const float* src, float* dst, int B, int rows, int N, long long batch_stride,
                              long long offset, hipStream_t stream) {
  if (!src || !dst || B <= 0 || rows <= 0 || N <= 0) return MMD_EINVAL;
  size_t n = (size_t)B * rows * N;
  hipLaunchKernelGGL(slice_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, src, dst, B, rows, N, batch_stride, offset);
  return mmd_check_launch();
}

// every level of the pyramid in one launch (five launches of 5 - 12 us at the head of the backward's serial chain otherwise):
// dst[row0[l] + b * HW_l + r, :] = src[b * bstride + off[l] + r * N ...]
struct SliceOff { long long v[MMD_MAX_LEV]; };
__global__ __launch_bounds__(256) void slice_rows_pyr_kernel(const float* __restrict__ src, float* __restrict__ dst, Pyr p, int N,
                                                             long long bstride, SliceOff off, int vec) {
  // one thread per `vec` consecutive floats of a destination row block (vec = 2 when every slice start and N are even)
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * vec;
  const size_t total = (size_t)p.row0[p.n] * N;
  if (i >= total) return;
  const int row = (int)(i / N);
  const int l = pyr_level_of_row(p, row);
  long long o = off.v[0]; int r0 = 0, hw = p.H[0] * p.W[0];
#pragma unroll
  for (int k = 1; k < MMD_MAX_LEV; ++k) if (k == l) { o = off.v[k]; r0 = p.row0[k]; hw = p.H[k] * p.W[k]; }
  const size_t rel = i - (size_t)r0 * N;                 // offset inside the level's [B][HW][N] block
  const size_t per = (size_t)hw * N;
  const int b = (int)(rel / per);
  if (b >= p.B) {                                        // the level's padding rows (levels start at multiples of 128 rows)
    if (vec == 2) *reinterpret_cast<float2*>(dst + i) = make_float2(0.f, 0.f); else dst[i] = 0.f;
    return;
  }
  const float* s = src + (size_t)b * bstride + o + (rel - (size_t)b * per);
  if (vec == 2) *reinterpret_cast<float2*>(dst + i) = *reinterpret_cast<const float2*>(s);
  else dst[i] = *s;
}
extern "C" int mmd_slice_rows_pyr(const float* src, float* dst, const int* pyr_desc, int N, long long batch_stride,
                                  const long long* offsets, hipStream_t stream) {
  if (!src || !dst || !pyr_desc || !offsets || N <= 0) return MMD_EINVAL;
  Pyr p;
  if (mmd_make_pyr(p, pyr_desc)) return MMD_EINVAL;
  SliceOff off{};
  bool even = !(N & 1) && !(batch_stride & 1);
  for (int l = 0; l < p.n; ++l) { off.v[l] = offsets[l]; even = even && !(offsets[l] & 1); }
  const int vec = even ? 2 : 1;
  const size_t n = (size_t)p.row0[p.n] * N / vec;
  hipLaunchKernelGGL(slice_rows_pyr_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, src, dst, p, N, batch_stride, off, vec);
  return mmd_check_launch();
}

// BN(+swish) backward over a whole pyramid in one launch each (per-level BN parameters lev_stride channels apart)
extern "C" int mmd_bn_bwd_reduce_pyr(const float* g_in, const float* z, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, int act, const int* pyr_desc,
                                     long long lev_stride, float* g_out, double* sums, int C, hipStream_t stream) {
  if (!g_in || !z || !scale || !shift || !mean || !invstd || !sums || !pyr_desc || C <= 0 || (C & 3)) return MMD_EINVAL;
  Pyr p;
  if (mmd_make_pyr(p, pyr_desc)) return MMD_EINVAL;
  int M = p.row0[p.n];
  mmd_prof_tag(MMD_FAM_ELT, "bnredpyr C%lld", C, 0, 0, 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C, 64), cdiv(M, 128)), dim3(256), 0, stream, g_in, z, scale, shift, mean,
                     invstd, act, nullptr, nullptr, nullptr, 1, g_out, sums, M, C, p, lev_stride, 128, nullptr, 0);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * M * (double)C * 3);
  return mmd_check_launch();
}
extern "C" int mmd_bn_bwd_apply_pyr(const float* g, const float* z, const float* mean, const float* invstd,
                                    const float* gamma, const double* sums, const int* pyr_desc, long long lev_stride,
                                    float* dz, float* dgamma, float* dbeta, int C, const float* scale, const float* shift,
                                    int act, hipStream_t stream) {
  if (!g || !z || !mean || !invstd || !gamma || !sums || !dz || !pyr_desc || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  if (act != MMD_ACT_NONE && (act != MMD_ACT_SWISH || !scale || !shift)) return MMD_EINVAL;
  Pyr p;
  if (mmd_make_pyr(p, pyr_desc)) return MMD_EINVAL;
  int M = p.row0[p.n];
  mmd_prof_tag(MMD_FAM_ELT, "bnapppyr C%lld", C, 0, 0, 0);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(C, 64), cdiv(M, 128)), dim3(256), 0, stream, g, z, mean, invstd, gamma, sums,
                     1.0, dz, dgamma, dbeta, M, C, p, lev_stride, 128, BnBwdMod{scale, shift, act, nullptr, nullptr, nullptr, 1});
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * M * (double)C * 3);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- head gradients -> BiFPN output gradients, with their BatchNorm sums
// out = a + b (+ c) over a pyramid row buffer [row0[n], C] (the two heads' input gradients and the MTA loss' feature gradient), and, for
// every level whose feature map is the output of a BatchNorm (z_l: that level's own [B*H*W, C] raw conv output), the sums of that
// BatchNorm's backward: sums_l [2C] (+)= [sum out, sum out*xhat].  Replaces two scale_acc launches per step + one per level and the five
// mmd_bn_bwd_reduce launches of the last BiFPN cell.  Rows are walked in 128-row blocks (levels start at multiples of 128).
struct PyrBnDst { const float* z[MMD_MAX_LEV]; const float* mean[MMD_MAX_LEV]; const float* invstd[MMD_MAX_LEV]; double* sums[MMD_MAX_LEV]; };
template <typename T> __device__ __forceinline__ T pyr_sel(const T* arr, int lev) {
  T v = arr[0];
#pragma unroll
  for (int i = 1; i < MMD_MAX_LEV; ++i) if (lev == i) v = arr[i];
  return v;
}
__global__ __launch_bounds__(256) void pyr_add_bnsums_kernel(const float* a, const float* __restrict__ b,
                                                            const float* __restrict__ c3, float* out, int C, Pyr pyr,   // out may alias a (in place)
                                                            PyrBnDst d) {
  __shared__ float sRed[256];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * 64 + (tid & 15) * 4;
  const bool cok = c < C;
  const int r0 = blockIdx.y * 128;
  const int lev = pyr_level_of_row(pyr, r0);
  const int lrow0 = pyr_sel(pyr.row0, lev);
  const int r1 = min(r0 + 128, lrow0 + pyr.B * pyr_sel(pyr.H, lev) * pyr_sel(pyr.W, lev));
  const float* z = pyr_sel(d.z, lev);
  float4 mu = make_float4(0, 0, 0, 0), is = make_float4(0, 0, 0, 0), s1 = mu, s2 = mu;
  if (z && cok) { mu = mmd_ld4(pyr_sel(d.mean, lev) + c); is = mmd_ld4(pyr_sel(d.invstd, lev) + c); }
  if (cok) {
#pragma unroll 4
    for (int row = r0 + (tid >> 4); row < r1; row += 16) {
      const size_t off = (size_t)row * C + c;
      float4 v = mmd_ld4(a + off);
      const float4 u = mmd_ld4(b + off);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      if (c3) { const float4 t = mmd_ld4(c3 + off); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      mmd_st4(out + off, v);
      if (z) {
        const float4 zz = mmd_ld4(z + (size_t)(row - lrow0) * C + c);
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * (zz.x - mu.x) * is.x; s2.y += v.y * (zz.y - mu.y) * is.y;
        s2.z += v.z * (zz.z - mu.z) * is.z; s2.w += v.w * (zz.w - mu.w) * is.w;
      }
    }
  }
  if (!z) return;                     // block-uniform
  const float sa = block_chan_sum(s1, sRed, tid);
  const float sb = block_chan_sum(s2, sRed, tid);
  if (tid < 64 && blockIdx.x * 64 + tid < C) {
    double* sums = pyr_sel(d.sums, lev);
    atomicAdd(&sums[blockIdx.x * 64 + tid], (double)sa);
    atomicAdd(&sums[C + blockIdx.x * 64 + tid], (double)sb);
  }
}
// z / mean / invstd / sums: arrays of n level pointers (host arrays of device pointers; z[l] == NULL: no sums for that level)
extern "C" int mmd_pyr_add_bnsums(const float* a, const float* b, const float* c, float* out, const int* pyr_desc, int C,
                                  const float* const* z, const float* const* mean, const float* const* invstd, double* const* sums,
                                  hipStream_t stream) {
  if (!a || !b || !out || !pyr_desc || C <= 0 || (C & 3)) return MMD_EINVAL;
  Pyr p;
  if (mmd_make_pyr(p, pyr_desc)) return MMD_EINVAL;
  PyrBnDst d{};
  for (int l = 0; l < p.n; ++l) {
    if (z && z[l]) {
      if (!mean || !invstd || !sums || !mean[l] || !invstd[l] || !sums[l]) return MMD_EINVAL;
      d.z[l] = z[l]; d.mean[l] = mean[l]; d.invstd[l] = invstd[l]; d.sums[l] = sums[l];
    }
  }
  const int M = p.row0[p.n];
  hipLaunchKernelGGL(pyr_add_bnsums_kernel, dim3(cdiv(C, 64), cdiv(M, 128)), dim3(256), 0, stream, a, b, c, out, C, p, d);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- "augmented" step variant (ModelWithNMSLossAugmented)
// merge_batch_0_1 (src/optimization/train_methods.py:291-308): image 1 of the audio batch becomes
// log10(max(a0^10 + a1^10, 1e-7)) - literally the 10th POWER of the dB-scale spectrogram, as the reference computes it;
// every other image is copied.  Out of place, so a graph replay does not compound the merge.
__global__ void audio_merge01_kernel(const float* __restrict__ in, float* __restrict__ out, size_t per, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = in[i];
  if (i >= per && i < 2 * per) {
    float a0 = in[i - per];
    float p0 = a0 * a0; p0 = p0 * p0 * p0 * p0 * (a0 * a0);          // torch.pow(x, 10)
    float p1 = v * v; p1 = p1 * p1 * p1 * p1 * (v * v);
    float m = p0 + p1;
    if (m < 1e-7f) m = 1e-7f;
    v = log10f(m);
  }
  out[i] = v;
}
extern "C" int mmd_audio_merge01(const float* in, float* out, long long per_image, int B, hipStream_t stream) {
  if (!in || !out || per_image <= 0 || B < 2) return MMD_EINVAL;
  size_t total = (size_t)per_image * B;
  hipLaunchKernelGGL(audio_merge01_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, in, out, (size_t)per_image, total);
  return mmd_check_launch();
}
// average_batch_0_1 (:279-289): teacher feature map of image 1 <- (image 0 + image 1) / 2, in place
__global__ void avg_image01_kernel(float* f, size_t per) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < per) f[per + i] = (f[i] + f[per + i]) / 2.f;
}
extern "C" int mmd_avg_image01(float* f, long long per_image, hipStream_t stream) {
  if (!f || per_image <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(avg_image01_kernel, dim3(cdiv(per_image, 256)), dim3(256), 0, stream, f, (size_t)per_image);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- stem: direct 3x3 / stride-2 TF-SAME conv, NCHW image -> NHWC rows
// Replaces im2col + GEMM in the forward (the im2col buffer, 58-150 MB per net, is only needed by the weight gradient
// and is built there): one thread = one output pixel x all 32 output channels, weights broadcast from LDS ([k][32]),
// input read straight from the NCHW image (neighbouring threads cover neighbouring column pairs).  Epilogue: either the
// frozen nets' folded BN + swish, or the raw output plus the slotted BatchNorm sums (train).
// Reference: Conv2dStaticSamePadding(in, 32, k=3, s=2, bias=False) + _bn0 + swish (src/YetAnotherEfficientNet.py:519-523,597-604).
template <int STEM_CO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                        int B, int Cin, int H, int W, int OH, int OW, int pad_t, int pad_l, int Kp,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                        double* stats, double* ws, int slots) {
  extern __shared__ float sW[];                 // [Cin*9][32]
  __shared__ float sRed[2 * 4 * STEM_CO];
  const int tid = threadIdx.x, K = Cin * 9;
  for (int i = tid; i < K * STEM_CO; i += 256) { int k = i / STEM_CO, c = i % STEM_CO; sW[i] = w[(size_t)c * Kp + k]; }
  __syncthreads();
  const size_t total = (size_t)B * OH * OW;
  const size_t m = (size_t)blockIdx.x * 256 + tid;
  const bool ok = m < total;
  float acc[STEM_CO];
#pragma unroll
  for (int c = 0; c < STEM_CO; ++c) acc[c] = 0.f;
  if (ok) {
    const int ow = (int)(m % OW); size_t t = m / OW;
    const int oh = (int)(t % OH), b = (int)(t / OH);
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int ih = oh * 2 + i - pad_t;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int iw = ow * 2 + j - pad_l;
          float v = 0.f;
          if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = xp[(size_t)ih * W + iw];
          const float4* wk = reinterpret_cast<const float4*>(&sW[(ci * 9 + i * 3 + j) * STEM_CO]);
#pragma unroll
          for (int q = 0; q < STEM_CO / 4; ++q) {
            const float4 wv = wk[q];
            acc[4 * q] += v * wv.x; acc[4 * q + 1] += v * wv.y; acc[4 * q + 2] += v * wv.z; acc[4 * q + 3] += v * wv.w;
          }
        }
      }
    }
  }
  if (stats) {        // per-channel sum / sum of squares of the raw output: wave reduce, 4 waves through LDS, slotted f64 atomics
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < STEM_CO; ++c) {
      float s = wave_sum(ok ? acc[c] : 0.f), q = wave_sum(ok ? acc[c] * acc[c] : 0.f);
      if (lane == 0) { sRed[wave * STEM_CO + c] = s; sRed[4 * STEM_CO + wave * STEM_CO + c] = q; }
    }
    __syncthreads();
    if (tid < STEM_CO) {
      double* st = ws ? ws + (size_t)(blockIdx.x % slots) * 2 * STEM_CO : stats;
      float s = sRed[tid] + sRed[STEM_CO + tid] + sRed[2 * STEM_CO + tid] + sRed[3 * STEM_CO + tid];
      float q = sRed[4 * STEM_CO + tid] + sRed[5 * STEM_CO + tid] + sRed[6 * STEM_CO + tid] + sRed[7 * STEM_CO + tid];
      atomicAdd(&st[tid], (double)s);
      atomicAdd(&st[STEM_CO + tid], (double)q);
    }
  }
  if (!ok) return;
  float* yo = y + m * STEM_CO;
#pragma unroll
  for (int q = 0; q < STEM_CO / 4; ++q) {
    float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (scale) {
      const float4 sc = mmd_ld4(scale + 4 * q), sh = mmd_ld4(shift + 4 * q);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    }
    if (act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
    mmd_st4(yo + 4 * q, v);
  }
}
int mmd_pw_stem_gemm(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int OH, int OW, int pad_t, int pad_l,
                     int Kp, int Cout, const float* out_scale, const float* out_shift, int out_act, double* stats,
                     double* stats_ws, int ws_slots, hipStream_t stream);      // pw_gemm.hip
extern "C" int mmd_stem_conv_fwd(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Kp, int Cout,
                                 const float* out_scale, const float* out_shift, int out_act, double* stats,
                                 double* stats_ws, int ws_slots, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cin > 64 || H <= 0 || W <= 0 || Kp < Cin * 9) return MMD_EINVAL;
  if (Cout != 32 && Cout != 40 && Cout != 48 && Cout != 56 && Cout != 64) return MMD_EINVAL;      // EfficientNet b0-b7 stems
  if ((out_scale == nullptr) != (out_shift == nullptr)) return MMD_EINVAL;
  int OH = (H + 1) / 2, OW = (W + 1) / 2;
  int eh = (OH - 1) * 2 - H + 3, ew = (OW - 1) * 2 - W + 3;
  if (eh < 0) eh = 0; if (ew < 0) ew = 0;
  size_t total = (size_t)B * OH * OW;
  int nb = cdiv(total, 256);
  const bool slotted = stats && stats_ws && ws_slots > 1 && nb > MMD_STATS_DEPTH;
  mmd_prof_tag(MMD_FAM_ELT, "stem B%lld Cin%lld H%lld Co%lld", B, Cin, H, Cout);
  mmd_prof_begin(MMD_FAM_ELT, stream);
  static const int direct = getenv("MMD_STEM_DIRECT") ? 1 : 0;
  if (!direct && !(Kp & 3) && !(Cout & 3)) {
    // implicit GEMM on the MFMA kernel (pw_gemm.hip, StemOp): the image is gathered while the A tile is staged
    int rc = mmd_pw_stem_gemm(x, w, y, B, Cin, H, W, OH, OW, eh / 2, ew / 2, Kp, Cout, out_scale, out_shift, out_act, stats,
                              stats_ws, ws_slots, stream);
    mmd_prof_end(MMD_FAM_ELT, stream, 2.0 * total * Kp * Cout, 4.0 * ((double)B * Cin * H * W + (double)total * Cout));
    return rc;
  }
  void (*kern)(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, const float*, const float*, int, double*,
               double*, int) = Cout == 32 ? stem_conv_kernel<32> : Cout == 40 ? stem_conv_kernel<40> : Cout == 48 ? stem_conv_kernel<48>
                             : Cout == 56 ? stem_conv_kernel<56> : stem_conv_kernel<64>;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), (size_t)Cin * 9 * Cout * sizeof(float), stream, x, w, y, B, Cin, H, W,
                     OH, OW, eh / 2, ew / 2, Kp, out_scale, out_shift, out_act, stats, slotted ? stats_ws : nullptr, ws_slots);
  if (slotted) mmd_stats_fold(stats, stats_ws, ws_slots, 2 * Cout, stream);
  mmd_prof_end(MMD_FAM_ELT, stream, 0.0, 4.0 * ((double)B * Cin * H * W + (double)total * Cout));
  return mmd_check_launch();
}
