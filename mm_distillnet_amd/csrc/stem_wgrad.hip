// Weight gradient of the stem conv (3x3, stride 2, TF-SAME, NCHW image -> NHWC rows) WITHOUT the im2col matrix.
//   dW[co][ci*9 + i*3 + j] += sum_{b, oh, ow} dz[(b*OH + oh)*OW + ow][co] * x[b][ci][2*oh + i - pad_t][2*ow + j - pad_l]
// Reference: autograd of `Conv2dStaticSamePadding(in_channels, 32, 3, stride=2)` (src/YetAnotherEfficientNet.py:519-523, 597-604; padding rule
// :27-65).  Until round 4 the step built the [B*OH*OW, Kp] patch matrix (mmd_stem_im2col: 168 MB written for the 8-channel student at 512^2,
// 139 us) and ran the weight gradient as a GEMM item of the last grouped flush, which read it back - both at the very end of the backward,
// where nothing of the main chain is left to hide them.
//
// Here a persistent block walks tiles of 64 consecutive output pixels of one output row.  Per tile it stages
//   sX  [Cin*3][132]  the 3 input rows x Cin channels x 130 consecutive input columns the tile's patches are made of (coalesced row reads;
//                     zero outside the image), so patch element (p, ci, i, j) = sX[ci*3 + i][2p + j]
//   sDz [64][48]      the dz tile (row stride 48: conflict-free per half-wave for the transposed read below)
// and accumulates dW[co][n] += sum_p dz[p][co] * patch[p][n] on v_mfma_f32_16x16x4_f32 with M = co, N = n (ci, i, j), K = p: lane (r, g) of
// MFMA step s supplies a = sDz[4s + g][co0 + r] and b = sX[(n0 + r) / 3][2 (4s + g) + (n0 + r) % 3].  Output tiles (Cout / 16) x 5 (Kp <= 80)
// are dealt to the four waves; the accumulators stay in registers across the block's tiles.  The next tile's global loads are in flight
// while the current one is multiplied.  Blocks leave their partial [Cout][80] in a workspace slot of their own (plain stores); a second,
// tiny launch adds the slots into dW in a fixed order (deterministic, no same-address atomics).
// HBM: x once (~1.5x through the shared rows of neighbouring output rows: L2) + dz once - 0.13 GB instead of 0.64 GB for the student's stem.
#include "common.h"
#include "pw_args.h"

#define SW_TM 64            // output pixels per tile
#define SW_XW 132           // LDS row stride of the input rows (130 used)
#define SW_LDZ 48           // LDS row stride of the dz tile
#define SW_KT 5             // 16-wide column tiles of dW: Kp <= 80
#define SW_BLOCKS_MAX 2048  // workspace slots
static int sw_blocks() { static const int b = getenv("MMD_STEM_WG_BLOCKS") ? atoi(getenv("MMD_STEM_WG_BLOCKS")) : 1024; return b < 1 ? 1 : (b > SW_BLOCKS_MAX ? SW_BLOCKS_MAX : b); }

#ifdef MMD_SWSTAMPS
__device__ unsigned long long g_swst[64];
#define MMD_ST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 64) g_swst[i] = wall_clock64(); } while (0)
extern "C" int mmd_sw_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_swst), sizeof(g_swst)) == hipSuccess ? 0 : -1; }
#else
#define MMD_ST(i)
#endif
// BNP: dz is not a tensor - the stem's BatchNorm(+swish) backward is evaluated from (g = dz argument, bb.z) while the tile is staged (the
// coefficients of a thread's channel quads do not depend on the tile: registers), as in the 1x1 convs' BatchNorm-backward operand launches;
// block 0 adds dgamma / dbeta.  Saves the mmd_bn_bwd_apply launch at the very end of the backward and its [M, Cout] write + read.
template <int COT, bool BNP = false>          // 16-row tiles of dW: Cout <= 16 * COT
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ ws, int B,
                                                        int Cin, int H, int W, int OH, int OW, int pad_t, int pad_l, int Cout, int ntiles,
                                                        BnBwdOp bb = BnBwdOp{}) {
  constexpr int NT = COT * SW_KT;                  // output tiles of the block
  constexpr int TPW = (NT + 3) / 4;                // ... per wave
  constexpr int NXR = 13;                          // x elements per thread and tile: ceil(8 * 3 * 130 / 256)
  constexpr int NZR = (SW_TM * 16 * COT / 4 + 255) / 256;      // dz float4s per thread and tile (rows are Cout <= 16 COT wide)
  __shared__ float sX[8 * 3 * SW_XW];
  __shared__ float sDz[SW_TM * SW_LDZ];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform for the compiler too (tile ownership below)
  const int nrow = Cin * 3, nx = nrow * 130, K9 = Cin * 9;
  const int tpr = OW / SW_TM;                      // tiles per output row (host: OW % 64 == 0)
  const int zq = Cout >> 2;                        // float4s per dz row
  f32x4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the wave's output tiles (every wave walks TPW of them: the ones past NT multiply zeros and are not stored - no branch in the MFMA
  // loop) and the lanes' fixed operand offsets
  int aoff[TPW], boff[TPW];
  bool bok[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int ti = wave + 4 * t;
    const int cot = min(ti / SW_KT, COT - 1), nt = ti % SW_KT;
    const int n = nt * 16 + r;
    aoff[t] = cot * 16 + r;
    bok[t] = ti < NT && n < K9;
    boff[t] = bok[t] ? (n / 3) * SW_XW + (n % 3) : 0;
  }
  // the thread's (input row, column) items of a tile do not depend on the tile: decoded once
  int xi[NXR], xc[NXR], xb[NXR], xs[NXR];
  bool xv[NXR];
#pragma unroll
  for (int k = 0; k < NXR; ++k) {
    const int e = tid + k * 256;
    const int row = e / 130, c = e - row * 130;
    const int ci = row / 3;
    xv[k] = e < nx;
    xi[k] = row - ci * 3; xc[k] = c;
    xb[k] = min(ci, Cin - 1) * H * W;
    xs[k] = row * SW_XW + c;
  }
  int zp[NZR], zc[NZR];
#pragma unroll
  for (int k = 0; k < NZR; ++k) {
    const int e = tid + k * 256;
    zp[k] = e / zq; zc[k] = (e - zp[k] * zq) * 4;
  }
  float xr[NXR];
  float4 zr[NZR], gr[BNP ? NZR : 1];
  BnBwdCoef4 bq[BNP ? NZR : 1];
  if constexpr (BNP) {
#pragma unroll
    for (int k = 0; k < NZR; ++k) bn_bwd_coef4(bb, min(zc[k], Cout - 4), bq[k]);
    if (bb.dgamma && blockIdx.x == 0)
      for (int c = tid; c < Cout; c += 256) { bb.dgamma[c] += (float)bb.sums[Cout + c]; bb.dbeta[c] += (float)bb.sums[c]; }
  }
  // every load is unconditional on a clamped (always valid) address and masked afterwards (a guarded load is a branch + a full drain)
  auto gload = [&](int tile) {
    const int owt = tile % tpr; int q = tile / tpr;
    const int oh = q % OH, b = q / OH;
    const int iw0 = 2 * owt * SW_TM - pad_l, ih0 = 2 * oh - pad_t;
    const float* xb0 = x + (size_t)b * Cin * H * W;
#pragma unroll
    for (int k = 0; k < NXR; ++k) {
      const int ih = ih0 + xi[k], iw = iw0 + xc[k];
      const bool ok = xv[k] && ih >= 0 && ih < H && iw >= 0 && iw < W;
      const float v = xb0[xb[k] + min(max(ih, 0), H - 1) * W + min(max(iw, 0), W - 1)];
      xr[k] = ok ? v : 0.f;
    }
    const float* zb0 = dz + (((size_t)b * OH + oh) * OW + (size_t)owt * SW_TM) * Cout;
#pragma unroll
    for (int k = 0; k < NZR; ++k) {
      const float4 v = mmd_ld4(zb0 + min(zp[k], SW_TM - 1) * Cout + zc[k]);
      zr[k] = (zp[k] < SW_TM) ? v : make_float4(0, 0, 0, 0);
      if constexpr (BNP) gr[k] = mmd_ld4(bb.z + (zb0 - dz) + min(zp[k], SW_TM - 1) * Cout + zc[k]);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int k = 0; k < NXR; ++k)
      if (xv[k]) sX[xs[k]] = xr[k];
#pragma unroll
    for (int k = 0; k < NZR; ++k)
      if (zp[k] < SW_TM) {
        float4 v = zr[k];
        if constexpr (BNP) v = bn_bwd_eval4(v, gr[k], 1.f, bb.act, bq[k]);
        *reinterpret_cast<float4*>(&sDz[zp[k] * SW_LDZ + zc[k]]) = v;
      }
  };
  // dz columns past Cout (Cout < 16 COT) are never staged: zero them once
  for (int i = tid; i < SW_TM * SW_LDZ; i += 256) sDz[i] = 0.f;
  __syncthreads();
  int tile = blockIdx.x;
  MMD_ST(0);
  if (tile < ntiles) gload(tile);
  int it_ = 0;
  for (; tile < ntiles; tile += gridDim.x, ++it_) {
    lstore();
    MMD_ST(1 + 4 * it_);
    __syncthreads();
    MMD_ST(2 + 4 * it_);
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);
    MMD_ST(3 + 4 * it_);
#pragma unroll 8
    for (int s = 0; s < SW_TM / 4; ++s) {
      const int p = 4 * s + g;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const float av = sDz[p * SW_LDZ + aoff[t]];
        const float bx = sX[boff[t] + 2 * p];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bok[t] ? bx : 0.f, acc[t], 0, 0, 0);
      }
    }
    MMD_ST(4 + 4 * it_);
    __syncthreads();
  }
  // partial [16 COT][80] of this block: acc[t][i] = dW[cot*16 + 4g + i][nt*16 + r]
  float* out = ws + (size_t)blockIdx.x * (16 * COT * 16 * SW_KT);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int ti = wave + 4 * t;
    if (ti < NT) {
      const int cot = ti / SW_KT, nt = ti - cot * SW_KT;
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(cot * 16 + 4 * g + i) * (16 * SW_KT) + nt * 16 + r] = acc[t][i];
    }
  }
}

// dW[co][n] += sum over the workspace slots (fixed order): 16 lanes per output element, each a strided sixteenth of the slots with its loads
// unrolled, then a shuffle tree
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ ws, float* dw, int slots, int slot_floats, int Cout, int Kp) {
  const int idx = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
  float s = 0.f;
  if (idx < Cout * Kp) {
    const int co = idx / Kp, n = idx - co * Kp;
    const float* p = ws + (size_t)co * (16 * SW_KT) + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = part;
#pragma unroll 2
    for (; k + 48 < slots; k += 64) {
      s0 += p[(size_t)k * slot_floats]; s1 += p[(size_t)(k + 16) * slot_floats];
      s2 += p[(size_t)(k + 32) * slot_floats]; s3 += p[(size_t)(k + 48) * slot_floats];
    }
    for (; k < slots; k += 16) s0 += p[(size_t)k * slot_floats];
    s = (s0 + s1) + (s2 + s3);
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
  if (part == 0 && idx < Cout * Kp) dw[idx] += s;
}

// 1 when mmd_stem_conv_bwd_weight has a kernel for this geometry (the caller keeps mmd_stem_im2col + the weight-gradient GEMM otherwise)
extern "C" int mmd_stem_conv_bwd_weight_supported(int Cin, int H, int W, int Kp, int Cout) {
  const int OW = (W + 1) / 2;
  return (Cin >= 1 && Cin <= 8 && Kp >= Cin * 9 && Kp <= 16 * SW_KT && (Kp & 3) == 0 && Cout >= 4 && Cout <= 48 && (Cout & 3) == 0 && H >= 2 && W >= 2 &&
          OW % SW_TM == 0) ? 1 : 0;
}
// workspace floats mmd_stem_conv_bwd_weight needs (contents irrelevant on entry)
extern "C" int mmd_stem_wgrad_ws_floats(int Cout) { return SW_BLOCKS_MAX * (16 * ((Cout + 15) / 16)) * (16 * SW_KT); }

// dw [Cout, Kp] (+)= the stem conv's weight gradient from the NCHW image x [B, Cin, H, W] and dz [B*OH*OW, Cout] (OH = ceil(H/2), OW = ceil(W/2)).
static int stem_wgrad_impl(const float* x, const float* dz, float* dw, float* ws, int B, int Cin, int H, int W, int Kp, int Cout, const BnBwdOp* bb,
                           hipStream_t stream) {
  if (!x || !dz || !dw || !ws || B <= 0 || !mmd_stem_conv_bwd_weight_supported(Cin, H, W, Kp, Cout)) return MMD_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  int eh = (OH - 1) * 2 - H + 3, ew = (OW - 1) * 2 - W + 3;
  if (eh < 0) eh = 0; if (ew < 0) ew = 0;
  const int ntiles = B * OH * (OW / SW_TM);
  const int blocks = ntiles < sw_blocks() ? ntiles : sw_blocks();
  const int cot = (Cout + 15) / 16;
  mmd_prof_tag(MMD_FAM_PW_WGRAD, "stemwg B%lld Cin%lld H%lld Co%lld", B, Cin, H, Cout);
  mmd_prof_begin(MMD_FAM_PW_WGRAD, stream);
#define MMD_SW(COT_) do { \
    if (bb) hipLaunchKernelGGL((stem_wgrad_kernel<COT_, true>), dim3(blocks), dim3(256), 0, stream, x, dz, ws, B, Cin, H, W, OH, OW, eh / 2, ew / 2, Cout, ntiles, *bb); \
    else hipLaunchKernelGGL((stem_wgrad_kernel<COT_, false>), dim3(blocks), dim3(256), 0, stream, x, dz, ws, B, Cin, H, W, OH, OW, eh / 2, ew / 2, Cout, ntiles, BnBwdOp{}); } while (0)
  if (cot == 1) MMD_SW(1); else if (cot == 2) MMD_SW(2); else MMD_SW(3);
#undef MMD_SW
  hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3(cdiv(Cout * Kp, 16)), dim3(256), 0, stream, ws, dw, blocks, 16 * cot * 16 * SW_KT, Cout, Kp);
  const double M = (double)B * OH * OW;
  mmd_prof_end(MMD_FAM_PW_WGRAD, stream, 2.0 * M * Cout * Cin * 9, 4.0 * ((double)B * Cin * H * W + M * Cout * (bb ? 2 : 1)));
  return mmd_check_launch();
}
extern "C" int mmd_stem_conv_bwd_weight(const float* x, const float* dz, float* dw, float* ws, int B, int Cin, int H, int W, int Kp, int Cout,
                                        hipStream_t stream) {
  return stem_wgrad_impl(x, dz, dw, ws, B, Cin, H, W, Kp, Cout, nullptr, stream);
}
// The same with the stem's BatchNorm(+act) backward evaluated while the dz tile is staged: dz = BnBwd(g, z) (coefficients as in
// mmd_pwconv_bwd_data_bn: scale, shift, mean, invstd of the BatchNorm, sums = [sum g', sum g' xhat] over `count` rows), dgamma / dbeta (+)= the sums.
extern "C" int mmd_stem_conv_bwd_weight_bn(const float* x, const float* g, const float* z, float* dw, float* ws, int B, int Cin, int H, int W, int Kp,
                                           int Cout, const float* scale, const float* shift, const float* mean, const float* invstd,
                                           const double* sums, long long count, int act, float* dgamma, float* dbeta, hipStream_t stream) {
  if (!g || !z || !scale || !shift || !mean || !invstd || !sums || count <= 0 || (dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  BnBwdOp bb{};
  bb.z = z; bb.scale = scale; bb.shift = shift; bb.mean = mean; bb.invstd = invstd; bb.sums = sums; bb.inv_count = 1.0 / (double)count;
  bb.C = Cout; bb.act = act; bb.mul_b = nullptr; bb.rows_per_image = 1; bb.dz_out = nullptr; bb.dgamma = dgamma; bb.dbeta = dbeta;
  return stem_wgrad_impl(x, g, dw, ws, B, Cin, H, W, Kp, Cout, &bb, stream);
}
