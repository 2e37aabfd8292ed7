// Depthwise k x k convolution (k in {3,5}, stride in {1,2}) on NHWC fp32 with TF-"SAME" zero padding
// folded into index math — CDNA4 / gfx950.
//
// Reference op: Conv2dStaticSamePadding with groups=C (src/YetAnotherEfficientNet.py:27-65, call sites
// :433-435 and src/YetAnotherEfficientDet.py:169-170); the materialised F.pad of the reference is gone.
//
// A block stages an input tile (+halo) for 64 channels in LDS, applying the PRODUCER's
// BatchNorm+swish on the way in (prologue), so every input element is activated once.  Lanes map
// to channels (float4 = 4 channels per lane, 16 lanes = 64 channels = 256 contiguous bytes per
// pixel), 16 pixel-groups per block each own a strip of R outputs along W and reuse the LDS
// reads in registers.  Weights live tap-major [k*k][C] so a lane's 4 channels are one float4.
// Epilogue: raw per-channel sum / sum-of-squares for train-mode BN (double atomics), or eval-mode
// BN+swish and the squeeze-excite average pool.
#include "common.h"
#include <cstdlib>

struct DwArgs {
  const float* x; const float* w; float* y;
  int B, H, W, C, OH, OW;
  int pad_t, pad_l;
  int flip;
  const float* in_scale; const float* in_shift; int in_act; BnLive in_bn;
  const float* out_scale; const float* out_shift; int out_act;
  double* stats; long long* pool; float pool_scale;      // pool: Q36 fixed-point sums (common.h mmd_pool_add)
  double* stats_ws; int ws_slots;            // slotted sums (common.h)
  // input-gradient launch feeding a BatchNorm(+swish) backward: stats become (sum g, sum g*xhat) with g = y * swish'(u),
  // u = bz*bscale + bshift, xhat = (bz - bmean)*binvstd, bz = the BN's forward input at the output position (y itself is stored)
  const float* bz; const float* bscale; const float* bshift; const float* bmean; const float* binvstd;
  int wg_act;        // dw3_rows_kernel<WG>: activation of the weight gradient's x operand (bz = x, bscale / bshift = its producer affine)
  float* dwg;        // input-gradient launch with `bz`: also the depthwise WEIGHT gradient [k*k, C] += sum_q swish(u)[q] * x[q + tap] (x = the launch's input = dY)
  int noswz;
  int tiles_h, tiles_w, cchunks;
  Pyr pyr; long long lev_stride;
  // PRO == 2 (input-gradient launch of an MBConv depthwise conv): the input is not a tensor but the BatchNorm-1(+swish, squeeze-excite)
  // backward evaluated while the tile is staged,
  //   dz1 = scale*( g' - m1 - xhat*m2 ),  g' = (x * q_gate[img,c] + q_add[img,c]) * swish'(q_z*scale + shift),  xhat = (q_z - mean)*invstd,
  // x = g1 (gradient w.r.t. the gated activation), q_z = z1, [m1, m2] = q_sums / count: what mmd_bn_bwd_apply(mul_bc, add_bc) would write
  // to HBM first ([M, C] write + read per block); q_dgamma / q_dbeta (+)= [sum g'*xhat, sum g'] by the first tile's blocks.
  const float* q_z; const float* q_gate; const float* q_add; const float* q_scale; const float* q_shift; const float* q_mean;
  const float* q_invstd; const double* q_sums; double q_inv_count; float* q_dgamma; float* q_dbeta;
  int x16, y16, bz16, qz16;      // bf16 storage (common.h w16) of x, y, bz, q_z - tile kernel only
  int g_images; long long g_w, g_bn;      // grouped frozen nets (common.h MmdGroup): image b reads group b / g_images' taps and folded coefficients
};

// LANES = float4 lanes per pixel (16 -> 64-channel chunks; 8 / 4 -> 32- / 16-channel chunks for the thin early layers,
// where a 64-wide chunk would leave 50-75 % of the threads idle)
template <int K, int S, int LANES = 16> struct DwCfg {
  static constexpr int TH = (S == 1) ? 8 : 4;
  static constexpr int TW = 8;
  static constexpr int CC = 4 * LANES;                      // channels per block
  static constexpr int G = 256 / LANES;                     // pixel groups per block
  static constexpr int R = TH * TW / G;                     // outputs per pixel-group (along W)
  static constexpr int IH = (TH - 1) * S + K;
  static constexpr int IW = (TW - 1) * S + K;
  static constexpr int SEG = (R - 1) * S + K;              // input columns one strip needs
};

// geometry + producer transform a block works with (kernel arguments, or one pyramid level of them) — kept in
// registers: copying the whole argument struct and patching it put it in scratch memory (296 B/lane)
struct DwView {
  const float* x; int H, W, C;
  float4 sc, sh; bool xf; int act; int x16 = 0;      // x16: x is a bf16 array (set by dw_fwd_kernel only; every other user reads fp32)
};

__device__ __forceinline__ void dw_in_coef(const float* in_scale, const float* in_shift, const BnLive& bn, int c, bool cok,
                                           DwView& v) {
  v.sc = make_float4(1, 1, 1, 1); v.sh = make_float4(0, 0, 0, 0);
  v.xf = in_scale || bn.stats;
  if (cok) {
    if (bn.stats) bn_live_coef4(bn, c, v.sc, v.sh);
    else if (in_scale) { v.sc = mmd_ld4(in_scale + c); v.sh = mmd_ld4(in_shift + c); }
  }
}

template <int K, int S, int LANES = 16, bool PRO = true>
__device__ __forceinline__ void dw_stage_input(const DwView& a, float* sIn, int b, int ih0, int iw0, int c0, int tid) {
  using Cf = DwCfg<K, S, LANES>;
  constexpr int NPIX = Cf::IH * Cf::IW;
  constexpr int NIT = (NPIX + Cf::G - 1) / Cf::G;
  const int c4 = (tid & (LANES - 1)) * 4;
  const int c = c0 + c4;
  const bool cok = c < a.C;
  const float4 sc = a.sc, sh = a.sh;
  // all global loads of the tile are issued before the first use (the transform + LDS write), so their latencies overlap
  float4 v[NIT];
  bool ok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = tid / LANES + it * Cf::G;
    const int ih = ih0 + p / Cf::IW, iw = iw0 + p % Cf::IW;
    ok[it] = cok && p < NPIX && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);     // unconditional load, masked below
    v[it] = mmd_ldw4(a.x, (((size_t)b * a.H + ihc) * a.W + iwc) * a.C + (cok ? c : 0), a.x16);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = tid / LANES + it * Cf::G;
    float4 u = ok[it] ? v[it] : make_float4(0, 0, 0, 0);
    if (PRO && ok[it]) {
      if (a.xf) { u.x = u.x * sc.x + sh.x; u.y = u.y * sc.y + sh.y; u.z = u.z * sc.z + sh.z; u.w = u.w * sc.w + sh.w; }
      if (a.act == MMD_ACT_SWISH) { u.x = mmd_swish(u.x); u.y = mmd_swish(u.y); u.z = mmd_swish(u.z); u.w = mmd_swish(u.w); }
    }
    if (p < NPIX) *reinterpret_cast<float4*>(&sIn[p * Cf::CC + c4]) = u;
  }
}

// PRO == 2 staging: see DwArgs::q_*.  Same tile walk as dw_stage_input, two tensors per pixel.
template <int K, int S, int LANES = 16>
__device__ __forceinline__ void dw_stage_input_bnbwd(const DwArgs& a, float* sIn, int b, int ih0, int iw0, int c0, int tid) {
  using Cf = DwCfg<K, S, LANES>;
  constexpr int NPIX = Cf::IH * Cf::IW;
  constexpr int NIT = (NPIX + Cf::G - 1) / Cf::G;
  const int c4 = (tid & (LANES - 1)) * 4;
  const int c = c0 + c4;
  const bool cok = c < a.C;
  const int cc = cok ? c : 0;
  float4 v[NIT], zv[NIT];
  bool ok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = tid / LANES + it * Cf::G;
    const int ih = ih0 + p / Cf::IW, iw = iw0 + p % Cf::IW;
    ok[it] = cok && p < NPIX && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
    const size_t off = (((size_t)b * a.H + ihc) * a.W + iwc) * a.C + cc;
    v[it] = mmd_ldw4(a.x, off, a.x16);
    zv[it] = mmd_ldw4(a.q_z, off, a.qz16);
  }
  const float4 gt = mmd_ld4(a.q_gate + (size_t)b * a.C + cc), ad = mmd_ld4(a.q_add + (size_t)b * a.C + cc);
  const float4 a1 = mmd_ld4(a.q_scale + cc), sh = mmd_ld4(a.q_shift + cc), mu = mmd_ld4(a.q_mean + cc), is = mmd_ld4(a.q_invstd + cc);
  float m1[4], m2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m1[i] = (float)(a.q_sums[cc + i] * a.q_inv_count); m2[i] = (float)(a.q_sums[a.C + cc + i] * a.q_inv_count); }
  const float4 a2 = make_float4(-a1.x * is.x * m2[0], -a1.y * is.y * m2[1], -a1.z * is.z * m2[2], -a1.w * is.w * m2[3]);
  const float4 a3 = make_float4(-a1.x * m1[0], -a1.y * m1[1], -a1.z * m1[2], -a1.w * m1[3]);
  auto ev = [](float g, float z, float gate, float add, float s1, float s2, float s3, float mu_, float sh_) {
    const float gp = (g * gate + add) * mmd_swish_grad(z * s1 + sh_);
    return s1 * gp + s2 * (z - mu_) + s3;
  };
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = tid / LANES + it * Cf::G;
    float4 u = make_float4(0, 0, 0, 0);
    if (ok[it]) {
      u.x = ev(v[it].x, zv[it].x, gt.x, ad.x, a1.x, a2.x, a3.x, mu.x, sh.x); u.y = ev(v[it].y, zv[it].y, gt.y, ad.y, a1.y, a2.y, a3.y, mu.y, sh.y);
      u.z = ev(v[it].z, zv[it].z, gt.z, ad.z, a1.z, a2.z, a3.z, mu.z, sh.z); u.w = ev(v[it].w, zv[it].w, gt.w, ad.w, a1.w, a2.w, a3.w, mu.w, sh.w);
    }
    if (p < NPIX) *reinterpret_cast<float4*>(&sIn[p * Cf::CC + c4]) = u;
  }
}

// PRO / EPI: the launch's prologue and epilogue mode as compile-time parameters (PRO: producer transform; EPI 0 raw, 1 raw + BatchNorm
// sums, 2 raw + `bz` sums, 3 folded BN / activation / pool, 4 any combination at run time) - as for dw3_rows_kernel, the union of all
// modes costs registers and branches on every launch.
// WG (with EPI 2): the weight gradient of the forward conv out of the same launch - the dY tile is in LDS, the forward input a0 = swish(u)
// is recomputed for the BatchNorm sums anyway; k*k float4 sums per thread, one block reduction, k*k*CC atomics per block.  A depthwise
// weight-gradient launch is a leaf whose kernel time the saturated chip pays in full (profiles/r02_notes.md).
// -DMMD_DWSTAMPS (dev build, tools/dev/dw_phases.py): block 0 / thread 0 stamps the 100 MHz wall clock at the phase boundaries
#ifdef MMD_DWSTAMPS
__device__ unsigned long long g_dwst[16];
#define MMD_DT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dwst[i] = wall_clock64(); } while (0)
extern "C" int mmd_dw_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dwst), sizeof(g_dwst)) == hipSuccess ? 0 : -1; }
#else
#define MMD_DT(i)
#endif
template <int K, int S, int LANES = 16, int PRO = 1, int EPI = 4, bool WG = false>
__global__ __launch_bounds__(256) void dw_fwd_kernel(DwArgs a) {
  using Cf = DwCfg<K, S, LANES>;
  constexpr int CC = Cf::CC;
  __shared__ float sIn[Cf::IH * Cf::IW * CC];
  __shared__ float sW[K * K * CC];
  __shared__ float sRed[2 * 4 * CC];
  const int tid = threadIdx.x;
  MMD_DT(0);
  // XCD-aware order: blocks are dealt round-robin to the 8 XCDs; remap so that each XCD's L2 sees a contiguous run of tiles
  // (neighbouring tiles share halo rows / columns)
  int bid = (a.noswz) ? (int)blockIdx.x : mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  // pyramid launch (k=3, s=1): pick this block's level (unrolled selects: no dynamic indexing of the argument arrays)
  int H = a.H, W = a.W, OH = a.OH, OW = a.OW, tiles_h = a.tiles_h, tiles_w = a.tiles_w, lev = 0;
  size_t ro = 0;
  if (a.pyr.n) {
    int blk = 0;
    H = a.pyr.H[0]; W = a.pyr.W[0];
#pragma unroll
    for (int i = 1; i < MMD_MAX_LEV; ++i)
      if (i < a.pyr.n && bid >= a.pyr.blk0[i]) { lev = i; H = a.pyr.H[i]; W = a.pyr.W[i]; ro = (size_t)a.pyr.row0[i]; blk = a.pyr.blk0[i]; }
    bid -= blk; OH = H; OW = W; ro *= a.C;
    tiles_h = (H + Cf::TH - 1) / Cf::TH; tiles_w = (W + Cf::TW - 1) / Cf::TW;
  }
  const int cc = bid % a.cchunks; bid /= a.cchunks;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int b = bid;
  const size_t gw = a.g_images ? (size_t)(b / a.g_images) * a.g_w : 0, gb = a.g_images ? (size_t)(b / a.g_images) * a.g_bn : 0;
  const int c0 = cc * CC, c4 = (tid & (LANES - 1)) * 4, c = c0 + c4;
  const bool cok = c < a.C;
  const int oh0 = th * Cf::TH, ow0 = tw * Cf::TW;
  DwView v;
  v.x = a.x + ro; v.H = H; v.W = W; v.C = a.C; v.act = a.in_act; v.x16 = a.x16;      // (pyramid launches are never w16: ro is in fp32 elements)
  {
    BnLive bn = a.in_bn;
    const long long lo = (long long)lev * a.lev_stride;
    if (bn.stats) { bn.stats += 2 * lo; bn.gamma += lo; bn.beta += lo; if (a.pyr.n) bn.inv_count = 1.0 / ((double)a.B * H * W); }
    dw_in_coef(a.in_scale ? a.in_scale + gb + lo : nullptr, a.in_scale ? a.in_shift + gb + lo : nullptr, bn, c, cok, v);
  }
  float* const yout = a.y + ro;
  constexpr bool E_BZ = EPI == 2 || EPI == 4, E_ST = EPI == 1 || EPI == 4, E_OUT = EPI == 3 || EPI == 4;
  const int p = tid / LANES;
  const int orow = p / (Cf::TW / Cf::R);
  const int ocol0 = (p % (Cf::TW / Cf::R)) * Cf::R;
  // the epilogue's `bz` values: loaded first, unconditionally from clamped positions (behind the staging loads they would be R dependent
  // round trips at the end of the block: 3.4 of a block's 19 us, profiles/r04_dw_bwd_phases.txt)
  constexpr bool Z_PRE = EPI == 2;       // (the run-time-mode body EPI 4 loads them in place: 14 more live registers cost it a block per CU)
  float4 zq[Z_PRE ? Cf::R : 1];
  if constexpr (Z_PRE) {
    if (a.bz) {
      const size_t zrow = ((size_t)b * OH + min(oh0 + orow, OH - 1)) * OW;
#pragma unroll
      for (int o = 0; o < Cf::R; ++o) zq[o] = mmd_ldw4(a.bz, (zrow + min(ow0 + ocol0 + o, OW - 1)) * a.C + (cok ? c : 0), a.bz16);
    }
  }

  for (int i = tid; i < K * K * LANES; i += 256) {
    int tap = i / LANES, q = (i % LANES) * 4;
    int src = a.flip ? (K * K - 1 - tap) : tap;
    float4 wv = (c0 + q < a.C) ? mmd_ld4(a.w + gw + (size_t)src * a.C + c0 + q) : make_float4(0, 0, 0, 0);
    *reinterpret_cast<float4*>(&sW[tap * CC + q]) = wv;
  }
  if constexpr (PRO == 2) {
    dw_stage_input_bnbwd<K, S, LANES>(a, sIn, b, oh0 * S - a.pad_t, ow0 * S - a.pad_l, c0, tid);
    if (a.q_dgamma && b == 0 && th == 0 && tw == 0 && tid < CC && c0 + tid < a.C) {
      a.q_dgamma[c0 + tid] += (float)a.q_sums[a.C + c0 + tid];
      a.q_dbeta[c0 + tid] += (float)a.q_sums[c0 + tid];
    }
  } else {
    dw_stage_input<K, S, LANES, PRO != 0>(v, sIn, b, oh0 * S - a.pad_t, ow0 * S - a.pad_l, c0, tid);
  }
  MMD_DT(1);
  __syncthreads();
  MMD_DT(2);

  float4 acc[Cf::R];
#pragma unroll
  for (int o = 0; o < Cf::R; ++o) acc[o] = make_float4(0, 0, 0, 0);
  // the tap-row loop is NOT unrolled for k=5: a fully unrolled body keeps all 25 weight float4s live
  // (256 VGPRs, one wave per SIMD); rolled it needs ~100 and four waves hide the LDS latency
#pragma unroll(K == 3 ? 3 : 1)
  for (int i = 0; i < K; ++i) {
    float4 in[Cf::SEG];
    const float* prow = &sIn[((orow * S + i) * Cf::IW + ocol0 * S) * CC + c4];
#pragma unroll
    for (int q = 0; q < Cf::SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * CC);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      float4 wv = *reinterpret_cast<const float4*>(&sW[(i * K + j) * CC + c4]);
#pragma unroll
      for (int o = 0; o < Cf::R; ++o) {
        acc[o].x += in[o * S + j].x * wv.x; acc[o].y += in[o * S + j].y * wv.y;
        acc[o].z += in[o * S + j].z * wv.z; acc[o].w += in[o * S + j].w * wv.w;
      }
    }
  }
  MMD_DT(3);
  float4 osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  if (E_OUT && a.out_scale && cok) { osc = mmd_ld4(a.out_scale + gb + c); osh = mmd_ld4(a.out_shift + gb + c); }
  float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0), pl = make_float4(0, 0, 0, 0);
  float4 bsc, bsh, bmu, bis;
  if (E_BZ && a.bz && cok) { bsc = mmd_ld4(a.bscale + c); bsh = mmd_ld4(a.bshift + c); bmu = mmd_ld4(a.bmean + c); bis = mmd_ld4(a.binvstd + c); }
  const int oh = oh0 + orow;
  float4 fq[WG ? Cf::R : 1];
  if (WG) {
#pragma unroll
    for (int o = 0; o < Cf::R; ++o) fq[o] = make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int o = 0; o < Cf::R; ++o) {
    int ow = ow0 + ocol0 + o;
    if (cok && oh < OH && ow < OW) {
      float4 v = acc[o];
      if (E_BZ && a.bz) {
        const float4 zz = Z_PRE ? zq[Z_PRE ? o : 0] : mmd_ldw4(a.bz, (((size_t)b * OH + oh) * OW + ow) * a.C + c, a.bz16);
        if (WG) fq[o] = make_float4(mmd_swish(zz.x * bsc.x + bsh.x), mmd_swish(zz.y * bsc.y + bsh.y), mmd_swish(zz.z * bsc.z + bsh.z),
                                    mmd_swish(zz.w * bsc.w + bsh.w));
        float4 gg;
        gg.x = v.x * mmd_swish_grad(zz.x * bsc.x + bsh.x); gg.y = v.y * mmd_swish_grad(zz.y * bsc.y + bsh.y);
        gg.z = v.z * mmd_swish_grad(zz.z * bsc.z + bsh.z); gg.w = v.w * mmd_swish_grad(zz.w * bsc.w + bsh.w);
        s.x += gg.x; s.y += gg.y; s.z += gg.z; s.w += gg.w;
        ss.x += gg.x * (zz.x - bmu.x) * bis.x; ss.y += gg.y * (zz.y - bmu.y) * bis.y;
        ss.z += gg.z * (zz.z - bmu.z) * bis.z; ss.w += gg.w * (zz.w - bmu.w) * bis.w;
      } else if (E_ST) {
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        ss.x += v.x * v.x; ss.y += v.y * v.y; ss.z += v.z * v.z; ss.w += v.w * v.w;
      }
      float4 t = v;
      if (E_OUT) {
        if (a.out_scale) { t.x = t.x * osc.x + osh.x; t.y = t.y * osc.y + osh.y; t.z = t.z * osc.z + osh.z; t.w = t.w * osc.w + osh.w; }
        if (a.out_act == MMD_ACT_SWISH) { t.x = mmd_swish(t.x); t.y = mmd_swish(t.y); t.z = mmd_swish(t.z); t.w = mmd_swish(t.w); }
        pl.x += t.x; pl.y += t.y; pl.z += t.z; pl.w += t.w;
      }
      mmd_stw4(yout, (((size_t)b * OH + oh) * OW + ow) * a.C + c, t, a.y16);
    }
  }
  MMD_DT(4);
  if (((E_ST || E_BZ) && a.stats) || (E_OUT && a.pool)) {
    // reduce over the 4 pixel-groups of a wave (lanes l, l^16, l^32, l^48 share c4), then over 4 waves in LDS
    auto red4 = [](float4 v) {       // lanes l, l^LANES, l^2LANES, ... of a wave share the channel group
#pragma unroll
      for (int o = LANES; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
      }
      return v;
    };
    const int wave = tid >> 6, lane = tid & 63;
    if ((E_ST || E_BZ) && a.stats) {
      s = red4(s); ss = red4(ss);
      if (lane < LANES) {
        *reinterpret_cast<float4*>(&sRed[wave * CC + c4]) = s;
        *reinterpret_cast<float4*>(&sRed[4 * CC + wave * CC + c4]) = ss;
      }
      __syncthreads();
      if (tid < CC && c0 + tid < a.C) {
        float vs = sRed[tid] + sRed[CC + tid] + sRed[2 * CC + tid] + sRed[3 * CC + tid];
        float vq = sRed[4 * CC + tid] + sRed[5 * CC + tid] + sRed[6 * CC + tid] + sRed[7 * CC + tid];
        double* st = a.stats_ws ? a.stats_ws + (size_t)((blockIdx.x / a.cchunks) % a.ws_slots) * 2 * a.C : a.stats;
        atomicAdd(&st[c0 + tid], (double)vs);
        atomicAdd(&st[a.C + c0 + tid], (double)vq);
      }
      __syncthreads();
    }
    if (E_OUT && a.pool) {
      pl = red4(pl);
      if (lane < LANES) *reinterpret_cast<float4*>(&sRed[wave * CC + c4]) = pl;
      __syncthreads();
      if (tid < CC && c0 + tid < a.C) {
        float v = sRed[tid] + sRed[CC + tid] + sRed[2 * CC + tid] + sRed[3 * CC + tid];
        mmd_pool_add(&a.pool[(size_t)b * a.C + c0 + tid], v, a.pool_scale);
      }
    }
  }
  MMD_DT(5);
  if constexpr (WG) {
    static_assert(S == 1 && Cf::IH * Cf::IW >= 4 * K * K, "weight-gradient reduction aliases the input tile");
    // dw[i][j] += sum_q a0[q] * dY[q - (i - p, j - p)]: the launch's (flipped-tap) window of dY around q, read once more from the LDS tile.
    // One tap row at a time: k float4 sums, butterfly-reduced over the wave's NG = 64 / LANES pixel groups (every lane then holds the
    // wave's sum), and pixel group i % NG keeps row i - ceil(k / NG) x k float4 per lane instead of k*k (100 VGPRs at k = 5, which held
    // the launch at two blocks per CU).
    constexpr int NG = 64 / LANES, NK = (K + NG - 1) / NG;
    const int lane = tid & 63, grp = lane / LANES, wave = tid >> 6;
    float4 keep[NK][K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      float4 in[Cf::SEG];
      const float* prow = &sIn[((orow + i) * Cf::IW + ocol0) * CC + c4];
#pragma unroll
      for (int q = 0; q < Cf::SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * CC);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float4 v = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int o = 0; o < Cf::R; ++o) {
          v.x += fq[o].x * in[o + j].x; v.y += fq[o].y * in[o + j].y; v.z += fq[o].z * in[o + j].z; v.w += fq[o].w * in[o + j].w;
        }
#pragma unroll
        for (int o = LANES; o < 64; o <<= 1) {
          v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
        }
        if (grp == i % NG) keep[i / NG][j] = v;
      }
    }
    MMD_DT(6);
    __syncthreads();                                  // every read of the tile is done: reduce in its place
    float* sRedW = sIn;                               // [4 waves][K*K][CC]
#pragma unroll
    for (int r = 0; r < NK; ++r) {
      const int i = r * NG + grp;
      if (i < K) {
#pragma unroll
        for (int j = 0; j < K; ++j) *reinterpret_cast<float4*>(&sRedW[(wave * K * K + i * K + j) * CC + c4]) = keep[r][j];
      }
    }
    __syncthreads();
    for (int i = tid; i < K * K * CC; i += 256) {
      const int t = i / CC, q = i - t * CC;
      if (c0 + q < a.C)
        atomicAdd(&a.dwg[(size_t)(K * K - 1 - t) * a.C + c0 + q], sRedW[(0 * K * K + t) * CC + q] + sRedW[(1 * K * K + t) * CC + q] +
                                                                    sRedW[(2 * K * K + t) * CC + q] + sRedW[(3 * K * K + t) * CC + q]);
    }
    MMD_DT(7);
  }
}

template <int K, int S, int LANES = 16>
static int dw_fwd_launch(DwArgs& a, hipStream_t st) {
  using Cf = DwCfg<K, S, LANES>;
  a.tiles_h = cdiv(a.OH, Cf::TH); a.tiles_w = cdiv(a.OW, Cf::TW); a.cchunks = cdiv(a.C, Cf::CC);
  long long nb = (long long)a.B * a.tiles_h * a.tiles_w * a.cchunks;
  static const int noswz = getenv("MMD_DW_NOSWZ") ? 1 : 0;
  a.noswz = noswz;
  if (!a.stats || a.ws_slots < 2 || nb / a.cchunks <= MMD_STATS_DEPTH) a.stats_ws = nullptr;
  const bool pro = a.in_scale || a.in_bn.stats || a.in_act != MMD_ACT_NONE;
  const bool out = a.out_scale || a.out_act != MMD_ACT_NONE || a.pool;
  const int epi = (a.stats && out) ? 4 : (a.bz ? 2 : (a.stats ? 1 : (out ? 3 : 0)));
  const dim3 grid((unsigned)nb), blk(256);
#define MMD_DW_TILE(P, E) hipLaunchKernelGGL((dw_fwd_kernel<K, S, LANES, P, E>), grid, blk, 0, st, a)
  if constexpr (S == 1 && LANES >= 8) {
    if (a.q_z) {        // BatchNorm-1 backward evaluated in the prologue (MBConv input gradient; always with the BatchNorm-0 sums + weight gradient)
      if (!(a.dwg && epi == 2 && !pro)) return MMD_EINVAL;
      hipLaunchKernelGGL((dw_fwd_kernel<K, S, LANES, 2, 2, true>), grid, blk, 0, st, a); goto launched;
    }
  }
  if (a.q_z) return MMD_EINVAL;
  if constexpr (S == 1) {
    if (a.dwg && epi == 2 && !pro) { hipLaunchKernelGGL((dw_fwd_kernel<K, S, LANES, false, 2, true>), grid, blk, 0, st, a); goto launched; }
  }
  if (a.dwg) return MMD_EINVAL;
  if (epi == 4) MMD_DW_TILE(true, 4);
  else if (pro) { if (epi == 0) MMD_DW_TILE(true, 0); else if (epi == 1) MMD_DW_TILE(true, 1); else if (epi == 2) MMD_DW_TILE(true, 2); else MMD_DW_TILE(true, 3); }
  else { if (epi == 0) MMD_DW_TILE(false, 0); else if (epi == 1) MMD_DW_TILE(false, 1); else if (epi == 2) MMD_DW_TILE(false, 2); else MMD_DW_TILE(false, 3); }
#undef MMD_DW_TILE
launched:
  if (a.stats_ws) mmd_stats_fold(a.stats, a.stats_ws, a.ws_slots, 2 * a.C, st);
  return mmd_check_launch();
}
// Channel chunk of the tile kernel: 64 channels (16 float4 lanes per pixel).  32-channel chunks where 64-channel ones pad the width by 10 %
// or more beyond what 32-channel ones do - C = 144 (192 against 160 lanes' worth of loads, FMAs and stores per pixel) and C = 288, the stride-1
// blocks of the 128^2 / 64^2 stages - measured no better (round 4: 16.01 - 16.07 against 15.94 - 16.00 ms/step: two outputs per thread reuse
// less of the LDS window than four); MMD_DW_LANES=8 forces them (4 / 8 / 16 for 3x3).
static int dw_lanes_for(int C) {
  static const int force = getenv("MMD_DW_LANES") ? atoi(getenv("MMD_DW_LANES")) : 0;
  (void)C;
  return force == 8 ? 8 : 16;
}
// 3x3 / stride 1: narrow channel chunks when C is small (thin 256x256 layers of the backbone)
static int dw_fwd_launch_31(DwArgs& a, hipStream_t st) {
  static const int force = getenv("MMD_DW_LANES") ? atoi(getenv("MMD_DW_LANES")) : 0;
  if (force == 4) return dw_fwd_launch<3, 1, 4>(a, st);
  if (a.C <= 16 && !force) return dw_fwd_launch<3, 1, 4>(a, st);
  if (a.C <= 32 && !force) return dw_fwd_launch<3, 1, 8>(a, st);
  return dw_lanes_for(a.C) == 8 ? dw_fwd_launch<3, 1, 8>(a, st) : dw_fwd_launch<3, 1, 16>(a, st);
}
static int dw_fwd_launch_51(DwArgs& a, hipStream_t st) {
  return dw_lanes_for(a.C) == 8 ? dw_fwd_launch<5, 1, 8>(a, st) : dw_fwd_launch<5, 1, 16>(a, st);
}

// ---- 3x3 / stride 1 as a row-streaming register window -------------------------------------------------------------------
// The tile kernel above spends ~70 % of a launch in its per-tile skeleton (tile index math, LDS staging writes, two barriers, epilogue
// address math: profiles/r02_notes.md).  Here a thread owns 4 channels x R output columns and walks down `rh` output rows keeping the
// three input rows it needs in registers: per output row it loads the next input row (R + 2 float4, producer transform applied as
// they arrive), does the 9 x R float4 FMAs against register-resident taps and runs the epilogue.  No LDS tile and no barrier in the
// loop; neighbouring threads re-load the two shared columns from L1.  Same prologue / epilogue contract as dw_fwd_kernel (producer
// BN(+swish) incl. live batch statistics, flipped taps for the input gradient, raw BatchNorm sums or folded BN + swish + pool, the
// `bz` sums of a BatchNorm backward).  LW = float4 lanes per pixel (16 / 8 / 4 -> 64- / 32- / 16-channel chunks).
// Stand-alone, frozen-net epilogue, B = 8: 128^2 x 144 47.8 -> 36.7 us, 256^2 x 64 66.1 -> 50.4, 32^2 x 528 13.8 -> 10.9.
struct DwRowsGeom { int colblocks, rowblocks, rh; };

// PRO: producer transform on the input; EPI: 0 raw output, 1 raw + BatchNorm sums, 2 raw + the `bz` sums of a BatchNorm backward,
// 3 folded BN / activation / pool.  Compile-time: the union of all modes needs 244 VGPRs and ran 1.5x slower than the specialised bodies.
// WG (flipped input-gradient launches without a producer transform): the conv's weight gradient out of the same launch,
// dw[8 - tap'] += sum_q f[q] * dY[q + tap' offset] with f = act(x * scale + shift) loaded at the output pixel (a.bz = x) and the dY
// neighbourhood = the register window; one block reduction + 9 x CC atomics per block.
template <int R, int LW, bool PRO, int EPI, bool WG = false>
__global__ __launch_bounds__(256) void dw3_rows_kernel(DwArgs a, DwRowsGeom gm) {
  constexpr int CC = 4 * LW;
  __shared__ float sRed[2 * 4 * CC];
  __shared__ float sRedW[WG ? 4 * 9 * CC : 1];
  const int tid = threadIdx.x, c4 = (tid & (LW - 1)) * 4, strip = tid / LW;
  int bid = (a.noswz) ? (int)blockIdx.x : mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  // pyramid launch: this block's level (unrolled selects, as in dw_fwd_kernel); every level uses the same R and rows per block, a
  // level narrower than a block row simply leaves its right-hand strips idle (the 64^2 level is 75 % of the rows)
  int H = a.H, W = a.W, lev = 0, colblocks = gm.colblocks, rowblocks = gm.rowblocks;
  size_t ro = 0;
  if (a.pyr.n) {
    int blk = 0;
    H = a.pyr.H[0]; W = a.pyr.W[0];
#pragma unroll
    for (int i = 1; i < MMD_MAX_LEV; ++i)
      if (i < a.pyr.n && bid >= a.pyr.blk0[i]) { lev = i; H = a.pyr.H[i]; W = a.pyr.W[i]; ro = (size_t)a.pyr.row0[i]; blk = a.pyr.blk0[i]; }
    bid -= blk; ro *= a.C;
    colblocks = (W + (256 / LW) * R - 1) / ((256 / LW) * R); rowblocks = (H + gm.rh - 1) / gm.rh;
  }
  const int cc = bid % a.cchunks; bid /= a.cchunks;
  const int cb = bid % colblocks; bid /= colblocks;
  const int rb = bid % rowblocks; bid /= rowblocks;
  const int b = bid, c0 = cc * CC, c = c0 + c4;
  const size_t gw = a.g_images ? (size_t)(b / a.g_images) * a.g_w : 0, gb = a.g_images ? (size_t)(b / a.g_images) * a.g_bn : 0;
  const bool cok = c < a.C;
  const int C = a.C;
  const int ow0 = (cb * (256 / LW) + strip) * R, oh0 = rb * gm.rh, oh1 = min(oh0 + gm.rh, H);
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = mmd_ld4(a.w + gw + (size_t)(a.flip ? 8 - t : t) * C + (cok ? c : 0));      // (lanes past C never store)
  DwView v;
  v.act = a.in_act;
  if (PRO) {
    BnLive bn = a.in_bn;
    const long long lo = (long long)lev * a.lev_stride;
    if (bn.stats) { bn.stats += 2 * lo; bn.gamma += lo; bn.beta += lo; if (a.pyr.n) bn.inv_count = 1.0 / ((double)a.B * H * W); }
    dw_in_coef(a.in_scale ? a.in_scale + gb + lo : nullptr, a.in_scale ? a.in_shift + gb + lo : nullptr, bn, c, cok, v);
  }
  float4 osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  if (EPI == 3 && a.out_scale && cok) { osc = mmd_ld4(a.out_scale + gb + c); osh = mmd_ld4(a.out_shift + gb + c); }
  float4 bsc, bsh, bmu, bis;
  if (EPI == 2 && cok) { bsc = mmd_ld4(a.bscale + c); bsh = mmd_ld4(a.bshift + c); bmu = mmd_ld4(a.bmean + c); bis = mmd_ld4(a.binvstd + c); }
  const float* const xb = a.x + ro + (size_t)b * H * W * C + (cok ? c : 0);
  const size_t ob = ro + (size_t)b * H * W * C + c;
  // Row loads are unconditional from clamped coordinates and masked afterwards (a guarded load compiles to a branch with a full
  // wait per load: R + 2 dependent round trips per row), and issued one row ahead of their use.
  unsigned colok = 0;
  int colo[R + 2];
#pragma unroll
  for (int q = 0; q < R + 2; ++q) {
    const int iw = ow0 - 1 + q;
    if (cok && iw >= 0 && iw < W) colok |= 1u << q;
    colo[q] = min(max(iw, 0), W - 1) * C;
  }
  auto issue_row = [&](int ih, float4 (&dst)[R + 2]) {
    const float* p = xb + (long long)min(max(ih, 0), H - 1) * W * C;
#pragma unroll
    for (int q = 0; q < R + 2; ++q) dst[q] = mmd_ld4(p + colo[q]);
  };
  auto finish_row = [&](int ih, float4 (&dst)[R + 2]) {
    const unsigned m = (ih >= 0 && ih < H) ? colok : 0u;
#pragma unroll
    for (int q = 0; q < R + 2; ++q) {
      float4 u = dst[q];
      if (PRO) {
        if (v.xf) { u.x = u.x * v.sc.x + v.sh.x; u.y = u.y * v.sc.y + v.sh.y; u.z = u.z * v.sc.z + v.sh.z; u.w = u.w * v.sc.w + v.sh.w; }
        if (v.act == MMD_ACT_SWISH) { u.x = mmd_swish(u.x); u.y = mmd_swish(u.y); u.z = mmd_swish(u.z); u.w = mmd_swish(u.w); }
      }
      const bool ok = (m >> q) & 1u;
      dst[q] = make_float4(ok ? u.x : 0.f, ok ? u.y : 0.f, ok ? u.z : 0.f, ok ? u.w : 0.f);
    }
  };
  float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0), pl = make_float4(0, 0, 0, 0);
  float4 dwa[WG ? 9 : 1];
  float4 gsc = make_float4(1, 1, 1, 1), gsh = make_float4(0, 0, 0, 0);
  float4 gmu = make_float4(0, 0, 0, 0), gis = make_float4(0, 0, 0, 0);
  if (WG) {
#pragma unroll
    for (int t = 0; t < 9; ++t) dwa[t] = make_float4(0, 0, 0, 0);
    const long long lo = (long long)lev * a.lev_stride;
    if (a.bscale && cok) { gsc = mmd_ld4(a.bscale + lo + c); gsh = mmd_ld4(a.bshift + lo + c); }
    if (a.stats && cok) { gmu = mmd_ld4(a.bmean + lo + c); gis = mmd_ld4(a.binvstd + lo + c); }
  }
  // the `bz` row of an output row (EPI 2 / WG) travels with the input rows: loaded unconditionally one row ahead (issue_z)
  constexpr bool ZROW = EPI == 2 || WG;
  auto issue_z = [&](int oh, float4 (&dst)[ZROW ? R : 1]) {
    if constexpr (ZROW) {
      const float* p = a.bz + ro + (size_t)b * H * W * C + (cok ? c : 0) + (size_t)min(oh, H - 1) * W * C;
#pragma unroll
      for (int o = 0; o < R; ++o) dst[o] = mmd_ld4(p + colo[o + 1]);
    }
  };
  auto out_row = [&](int oh, const float4 (&r0)[R + 2], const float4 (&r1)[R + 2], const float4 (&r2)[R + 2], const float4 (&zr)[ZROW ? R : 1]) {
    if constexpr (WG) {
#pragma unroll
      for (int o = 0; o < R; ++o) {
        const int ow = ow0 + o;
        if (cok && ow < W) {
          float4 f = zr[o];
          if (a.bscale) { f.x = f.x * gsc.x + gsh.x; f.y = f.y * gsc.y + gsh.y; f.z = f.z * gsc.z + gsh.z; f.w = f.w * gsc.w + gsh.w; }
          if (a.wg_act == MMD_ACT_SWISH) { f.x = mmd_swish(f.x); f.y = mmd_swish(f.y); f.z = mmd_swish(f.z); f.w = mmd_swish(f.w); }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            dwa[j].x += f.x * r0[o + j].x; dwa[j].y += f.y * r0[o + j].y; dwa[j].z += f.z * r0[o + j].z; dwa[j].w += f.w * r0[o + j].w;
            dwa[3 + j].x += f.x * r1[o + j].x; dwa[3 + j].y += f.y * r1[o + j].y; dwa[3 + j].z += f.z * r1[o + j].z; dwa[3 + j].w += f.w * r1[o + j].w;
            dwa[6 + j].x += f.x * r2[o + j].x; dwa[6 + j].y += f.y * r2[o + j].y; dwa[6 + j].z += f.z * r2[o + j].z; dwa[6 + j].w += f.w * r2[o + j].w;
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      float4 acc = make_float4(0, 0, 0, 0);
      // explicit FMAs in the tile kernel's tap order (row-major): every instantiation, and dw_fwd_kernel, round alike
      auto tap = [&](const float4& x, const float4& k) {
        acc.x = __fmaf_rn(x.x, k.x, acc.x); acc.y = __fmaf_rn(x.y, k.y, acc.y); acc.z = __fmaf_rn(x.z, k.z, acc.z); acc.w = __fmaf_rn(x.w, k.w, acc.w);
      };
#pragma unroll
      for (int j = 0; j < 3; ++j) tap(r0[o + j], wt[j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) tap(r1[o + j], wt[3 + j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) tap(r2[o + j], wt[6 + j]);
      const int ow = ow0 + o;
      if (cok && ow < W) {
        const size_t off = ob + ((size_t)oh * W + ow) * C;
        if (EPI == 2) {
          const float4 zz = zr[ZROW ? o : 0];
          float4 gg;
          gg.x = acc.x * mmd_swish_grad(zz.x * bsc.x + bsh.x); gg.y = acc.y * mmd_swish_grad(zz.y * bsc.y + bsh.y);
          gg.z = acc.z * mmd_swish_grad(zz.z * bsc.z + bsh.z); gg.w = acc.w * mmd_swish_grad(zz.w * bsc.w + bsh.w);
          s.x += gg.x; s.y += gg.y; s.z += gg.z; s.w += gg.w;
          ss.x += gg.x * (zz.x - bmu.x) * bis.x; ss.y += gg.y * (zz.y - bmu.y) * bis.y;
          ss.z += gg.z * (zz.z - bmu.z) * bis.z; ss.w += gg.w * (zz.w - bmu.w) * bis.w;
        } else if (EPI == 1) {
          s.x += acc.x; s.y += acc.y; s.z += acc.z; s.w += acc.w;
          ss.x += acc.x * acc.x; ss.y += acc.y * acc.y; ss.z += acc.z * acc.z; ss.w += acc.w * acc.w;
        }
        if (WG && a.stats) {      // sums of the BatchNorm(+swish) backward that consumes this launch's output (x's producer BN)
          const float4 zz = zr[ZROW ? o : 0];
          float4 gg;
          gg.x = acc.x * mmd_swish_grad(zz.x * gsc.x + gsh.x); gg.y = acc.y * mmd_swish_grad(zz.y * gsc.y + gsh.y);
          gg.z = acc.z * mmd_swish_grad(zz.z * gsc.z + gsh.z); gg.w = acc.w * mmd_swish_grad(zz.w * gsc.w + gsh.w);
          s.x += gg.x; s.y += gg.y; s.z += gg.z; s.w += gg.w;
          ss.x += gg.x * (zz.x - gmu.x) * gis.x; ss.y += gg.y * (zz.y - gmu.y) * gis.y;
          ss.z += gg.z * (zz.z - gmu.z) * gis.z; ss.w += gg.w * (zz.w - gmu.w) * gis.w;
        }
        float4 t = acc;
        if (EPI == 3) {
          if (a.out_scale) { t.x = t.x * osc.x + osh.x; t.y = t.y * osc.y + osh.y; t.z = t.z * osc.z + osh.z; t.w = t.w * osc.w + osh.w; }
          if (a.out_act == MMD_ACT_SWISH) { t.x = mmd_swish(t.x); t.y = mmd_swish(t.y); t.z = mmd_swish(t.z); t.w = mmd_swish(t.w); }
          pl.x += t.x; pl.y += t.y; pl.z += t.z; pl.w += t.w;
        }
        mmd_st4(a.y + off, t);
      }
    }
  };

  // four row buffers: three hold the window, the fourth receives the next row while the current output row is computed
  float4 w0[R + 2], w1[R + 2], w2[R + 2], w3[R + 2], za[ZROW ? R : 1], zb[ZROW ? R : 1];
  issue_row(oh0 - 1, w0); issue_row(oh0, w1); issue_row(oh0 + 1, w2); issue_z(oh0, za);
  finish_row(oh0 - 1, w0); finish_row(oh0, w1);
  for (int oh = oh0; oh < oh1; oh += 4) {
    issue_row(oh + 2, w3); issue_z(oh + 1, zb); finish_row(oh + 1, w2); out_row(oh, w0, w1, w2, za);
    if (oh + 1 < oh1) { issue_row(oh + 3, w0); issue_z(oh + 2, za); finish_row(oh + 2, w3); out_row(oh + 1, w1, w2, w3, zb); }
    if (oh + 2 < oh1) { issue_row(oh + 4, w1); issue_z(oh + 3, zb); finish_row(oh + 3, w0); out_row(oh + 2, w2, w3, w0, za); }
    if (oh + 3 < oh1) { issue_row(oh + 5, w2); issue_z(oh + 4, za); finish_row(oh + 4, w1); out_row(oh + 3, w3, w0, w1, zb); }
  }
  if ((EPI == 1 || EPI == 2) || (WG && a.stats) || (EPI == 3 && a.pool)) {      // lanes l, l^LW, l^2LW, ... of a wave share the channel quad; then the 4 waves through LDS
    auto red = [](float4 x) {
#pragma unroll
      for (int o = LW; o < 64; o <<= 1) {
        x.x += __shfl_xor(x.x, o, 64); x.y += __shfl_xor(x.y, o, 64); x.z += __shfl_xor(x.z, o, 64); x.w += __shfl_xor(x.w, o, 64);
      }
      return x;
    };
    const int wave = tid >> 6, lane = tid & 63;
    if (EPI == 1 || EPI == 2 || (WG && a.stats)) {
      s = red(s); ss = red(ss);
      if (lane < LW) {
        *reinterpret_cast<float4*>(&sRed[wave * CC + c4]) = s;
        *reinterpret_cast<float4*>(&sRed[4 * CC + wave * CC + c4]) = ss;
      }
      __syncthreads();
      if (tid < CC && c0 + tid < C) {
        const float vs = sRed[tid] + sRed[CC + tid] + sRed[2 * CC + tid] + sRed[3 * CC + tid];
        const float vq = sRed[4 * CC + tid] + sRed[5 * CC + tid] + sRed[6 * CC + tid] + sRed[7 * CC + tid];
        double* st = (a.stats_ws ? a.stats_ws + (size_t)((blockIdx.x / a.cchunks) % a.ws_slots) * 2 * C : a.stats) + 2 * (long long)lev * a.lev_stride;
        atomicAdd(&st[c0 + tid], (double)vs);
        atomicAdd(&st[C + c0 + tid], (double)vq);
      }
      __syncthreads();
    }
    if (EPI == 3 && a.pool) {
      pl = red(pl);
      if (lane < LW) *reinterpret_cast<float4*>(&sRed[wave * CC + c4]) = pl;
      __syncthreads();
      if (tid < CC && c0 + tid < C) {
        const float vv = sRed[tid] + sRed[CC + tid] + sRed[2 * CC + tid] + sRed[3 * CC + tid];
        mmd_pool_add(&a.pool[(size_t)b * C + c0 + tid], vv, a.pool_scale);
      }
    }
  }
  if constexpr (WG) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float4 x = dwa[t];
#pragma unroll
      for (int o = LW; o < 64; o <<= 1) {
        x.x += __shfl_xor(x.x, o, 64); x.y += __shfl_xor(x.y, o, 64); x.z += __shfl_xor(x.z, o, 64); x.w += __shfl_xor(x.w, o, 64);
      }
      if (lane < LW) *reinterpret_cast<float4*>(&sRedW[(wave * 9 + t) * CC + c4]) = x;
    }
    __syncthreads();
    for (int i = tid; i < 9 * CC; i += 256) {
      const int t = i / CC, q = i - t * CC;
      if (c0 + q < C)
        atomicAdd(&a.dwg[(size_t)(8 - t) * C + c0 + q],
                  sRedW[(0 * 9 + t) * CC + q] + sRedW[(1 * 9 + t) * CC + q] + sRedW[(2 * 9 + t) * CC + q] + sRedW[(3 * 9 + t) * CC + q]);
    }
  }
}

template <int R, int LW>
static int dw3_rows_go(DwArgs& a, hipStream_t st) {
  DwRowsGeom gm;
  a.cchunks = cdiv(a.C, 4 * LW);
  gm.colblocks = cdiv(a.W, (256 / LW) * R);
  // rows per block: as many as leave >= ~1024 blocks, at least 4 (every block re-reads two halo rows)
  static const int force_rh = getenv("MMD_DW_ROWS_RH") ? atoi(getenv("MMD_DW_ROWS_RH")) : 0;
  const long long per_row = (long long)a.B * a.cchunks * gm.colblocks;
  int rh = force_rh > 0 ? force_rh : (int)((long long)a.H * per_row / 1024);
  if (rh < 4) rh = 4; if (rh > a.H) rh = a.H;
  gm.rh = rh; gm.rowblocks = cdiv(a.H, rh);
  const long long nb = per_row * gm.rowblocks;
  static const int noswz = getenv("MMD_DW_NOSWZ") ? 1 : 0;
  a.noswz = noswz;
  if (!a.stats || a.ws_slots < 2 || nb / a.cchunks <= MMD_STATS_DEPTH) a.stats_ws = nullptr;
  const bool pro = a.in_scale || a.in_bn.stats || a.in_act != MMD_ACT_NONE;
  const int epi = a.bz ? 2 : (a.stats ? 1 : ((a.out_scale || a.out_act != MMD_ACT_NONE || a.pool) ? 3 : 0));
  const dim3 grid((unsigned)nb), blk(256);
#define MMD_DW3_ROWS(P, E) hipLaunchKernelGGL((dw3_rows_kernel<R, LW, P, E>), grid, blk, 0, st, a, gm)
  if (pro) { if (epi == 0) MMD_DW3_ROWS(true, 0); else if (epi == 1) MMD_DW3_ROWS(true, 1); else if (epi == 2) MMD_DW3_ROWS(true, 2); else MMD_DW3_ROWS(true, 3); }
  else { if (epi == 0) MMD_DW3_ROWS(false, 0); else if (epi == 1) MMD_DW3_ROWS(false, 1); else if (epi == 2) MMD_DW3_ROWS(false, 2); else MMD_DW3_ROWS(false, 3); }
#undef MMD_DW3_ROWS
  if (a.stats_ws) mmd_stats_fold(a.stats, a.stats_ws, a.ws_slots, 2 * a.C, st);
  return mmd_check_launch();
}

// -> 1 when the geometry is left to the tile kernel.  A thread's strip of R columns needs W >= 16 R to fill a block row; with a
// producer transform the R + 2 loaded columns per R outputs make narrow strips (R < 4) more expensive than the tile's 1.56x halo.
static int dw3_rows_launch(DwArgs& a, hipStream_t st) {
  static const int mode = getenv("MMD_DW_ROWS") ? atoi(getenv("MMD_DW_ROWS")) : 1;
  if (!mode || a.x16 || a.y16 || a.bz16 || a.qz16) return 1;      // (bf16 storage: tile kernel)
  if (a.stats && (a.out_scale || a.out_act != MMD_ACT_NONE || a.pool)) return 1;      // sums + folded epilogue together: tile kernel only
  const bool pro = a.in_scale || a.in_bn.stats || a.in_act != MMD_ACT_NONE;
  const int lw = a.C <= 16 ? 4 : (a.C <= 32 ? 8 : 16);
  const int strips = 256 / lw;
  // in the step (us per launch, tile -> rows): 256^2 x 16 37 -> 22, 256^2 x 32 53 -> 38, 32^2 x 528 17.8 -> 15.2, 16^2 x 2112 23.7 -> 16.2,
  // input gradient 128^2 x 144 72 -> 56; the one loser is the student's 128^2 x 144 forward (producer BN + swish, sums: 60 -> 68)
  if (pro && a.C > 64 && (a.C & 63)) return 1;
  if (a.W >= 4 * strips) { if (lw == 16) return dw3_rows_go<4, 16>(a, st); if (lw == 8) return dw3_rows_go<4, 8>(a, st); return dw3_rows_go<4, 4>(a, st); }
  if (pro || lw != 16) return 1;
  if (a.W >= 32) return dw3_rows_go<2, 16>(a, st);
  if (a.W >= 16) return dw3_rows_go<1, 16>(a, st);
  return 1;
}

static int same_pad_lo(int n, int k, int s, int* out) {
  int o = (n + s - 1) / s;
  int extra = (o - 1) * s - n + k;
  if (extra < 0) extra = 0;
  *out = o;
  return extra / 2;
}

// y[B,OH,OW,C] = dwconv_same(pro(x)[B,H,W,C], w[k*k,C]); OH = ceil(H/stride).
extern "C" int mmd_dwconv_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int k, int stride,
                              const float* in_scale, const float* in_shift, int in_act,
                              const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count,
                              const float* out_scale, const float* out_shift, int out_act,
                              double* stats, long long* pool, double* stats_ws, int ws_slots, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr) || (out_scale == nullptr) != (out_shift == nullptr)) return MMD_EINVAL;
  DwArgs a{};
  a.x = x; a.w = w; a.y = y; a.B = B; a.H = H; a.W = W; a.C = C;
  a.pad_t = same_pad_lo(H, k, stride, &a.OH); a.pad_l = same_pad_lo(W, k, stride, &a.OW);
  if (in_stats && (in_scale || !in_gamma || !in_beta || in_count <= 0)) return MMD_EINVAL;
  a.flip = 0; a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.in_bn = mmd_make_bn(in_stats, in_gamma, in_beta, in_count, C);
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act;
  a.stats = stats; a.pool = pool; a.pool_scale = 1.0f / (float)((long long)a.OH * a.OW);
  a.stats_ws = stats_ws; a.ws_slots = ws_slots;
  if (mmd_group_on()) {      // grouped frozen nets (common.h MmdGroup): image b reads group b / images' taps and folded coefficients
    const MmdGroup& gr = mmd_group();
    if (B != gr.n * gr.images || stats || in_stats) return MMD_EINVAL;
    a.g_images = gr.images; a.g_w = gr.w_stride; a.g_bn = gr.bn_stride;
  }
  mmd_prof_tag(MMD_FAM_DW, "dw H%lld C%lld k%lld s%lld", H, C, k, stride);
  mmd_prof_begin(MMD_FAM_DW, stream);
  int rc = 1;
  if (k == 3 && stride == 1) rc = dw3_rows_launch(a, stream);
  if (rc != 1) {}
  else if (k == 3 && stride == 1) rc = dw_fwd_launch_31(a, stream);
  else if (k == 3) rc = dw_fwd_launch<3, 2>(a, stream);
  else if (stride == 1) rc = dw_fwd_launch_51(a, stream);
  else rc = dw_fwd_launch<5, 2>(a, stream);
  mmd_prof_end(MMD_FAM_DW, stream, 2.0 * B * a.OH * a.OW * (double)C * k * k,
               4.0 * ((double)B * H * W * C + (double)B * a.OH * a.OW * C));
  return rc;
}

// same contract with bf16 storage of the wide tensors: w16 bit 0 = x, bit 1 = y are bf16 arrays (common.h)
static int mmd_dwconv_fwd_w16(const float* x, const float* w, float* y, int B, int H, int W, int C, int k, int stride,
                                  const float* in_scale, const float* in_shift, int in_act,
                                  const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count,
                                  const float* out_scale, const float* out_shift, int out_act,
                                  double* stats, long long* pool, double* stats_ws, int ws_slots, int w16, hipStream_t stream) {
  if (w16) return MMD_EINVAL;      // (bf16 storage of the wide tensors was deleted in round 6: the branches are compiled out, common.h)
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr) || (out_scale == nullptr) != (out_shift == nullptr)) return MMD_EINVAL;
  if (in_stats && (in_scale || !in_gamma || !in_beta || in_count <= 0)) return MMD_EINVAL;
  DwArgs a{};
  a.x = x; a.w = w; a.y = y; a.B = B; a.H = H; a.W = W; a.C = C;
  a.pad_t = same_pad_lo(H, k, stride, &a.OH); a.pad_l = same_pad_lo(W, k, stride, &a.OW);
  a.flip = 0; a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.in_bn = mmd_make_bn(in_stats, in_gamma, in_beta, in_count, C);
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act;
  a.stats = stats; a.pool = pool; a.pool_scale = 1.0f / (float)((long long)a.OH * a.OW);
  a.stats_ws = stats_ws; a.ws_slots = ws_slots;
  a.x16 = w16 & 1; a.y16 = (w16 >> 1) & 1;
  mmd_prof_tag(MMD_FAM_DW, "dw16 H%lld C%lld k%lld s%lld", H, C, k, stride);
  mmd_prof_begin(MMD_FAM_DW, stream);
  int rc;
  if (k == 3 && stride == 1) rc = dw_fwd_launch<3, 1, 16>(a, stream);
  else if (k == 3) rc = dw_fwd_launch<3, 2>(a, stream);
  else if (stride == 1) rc = dw_fwd_launch<5, 1>(a, stream);
  else rc = dw_fwd_launch<5, 2>(a, stream);
  mmd_prof_end(MMD_FAM_DW, stream, 2.0 * B * a.OH * a.OW * (double)C * k * k,
               (a.x16 ? 2.0 : 4.0) * (double)B * H * W * C + (a.y16 ? 2.0 : 4.0) * (double)B * a.OH * a.OW * C);
  return rc;
}

extern "C" int mmd_dwconv3_pyr_bwd_weight(const float* x, const float* dy, float* dw, const int* pyr_desc, int C, const float* in_scale,
                                          const float* in_shift, int in_act, long long lev_stride, hipStream_t stream);
extern "C" int mmd_bn_bwd_reduce_pyr(const float* g_in, const float* z, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, int act, const int* pyr_desc, long long lev_stride, float* g_out, double* sums,
                                     int C, hipStream_t stream);      // elt.hip

// Depthwise 3x3/s1 over a whole feature pyramid in ONE launch (shared weights; per-level producer BN via lev_stride).
// flip=1 gives the input gradient.  x, y: pyramid row buffers [row0[n], C].
extern "C" int mmd_dwconv3_pyr(const float* x, const float* w, float* y, const int* pyr_desc, int C, int flip,
                               const float* in_scale, const float* in_shift, int in_act, const double* in_stats,
                               const float* in_gamma, const float* in_beta, long long lev_stride,
                               const float* wg_x, const float* wg_scale, const float* wg_shift, int wg_act, float* dw_grad,
                               const float* wg_mean, const float* wg_invstd, double* bn_sums, hipStream_t stream) {
  if (!x || !w || !y || !pyr_desc || C <= 0 || (C & 3)) return MMD_EINVAL;
  if (dw_grad && (!wg_x || !flip || (wg_scale == nullptr) != (wg_shift == nullptr))) return MMD_EINVAL;
  if (bn_sums && (!dw_grad || !wg_scale || !wg_mean || !wg_invstd || wg_act != MMD_ACT_SWISH)) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MMD_EINVAL;
  if (in_stats && (in_scale || !in_gamma || !in_beta)) return MMD_EINVAL;
  DwArgs a{};
  if (mmd_make_pyr(a.pyr, pyr_desc)) return MMD_EINVAL;
  using Cf = DwCfg<3, 1>;
  a.x = x; a.w = w; a.y = y; a.B = a.pyr.B; a.C = C; a.pad_t = 1; a.pad_l = 1; a.flip = flip;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_bn = mmd_make_bn(in_stats, in_gamma, in_beta, 1, C);
  a.lev_stride = lev_stride; a.cchunks = cdiv(C, 64);
  if (mmd_group_on()) {
    const MmdGroup& gr = mmd_group();
    if (a.B != gr.n * gr.images || flip || dw_grad || in_stats) return MMD_EINVAL;
    a.g_images = gr.images; a.g_w = gr.w_stride; a.g_bn = gr.bn_stride;
  }
  int nb = 0;
  for (int l = 0; l < a.pyr.n; ++l) {
    a.pyr.blk0[l] = nb;
    nb += a.B * cdiv(a.pyr.H[l], Cf::TH) * cdiv(a.pyr.W[l], Cf::TW) * a.cchunks;
  }
  for (int l = a.pyr.n; l <= MMD_MAX_LEV; ++l) a.pyr.blk0[l] = nb;
  mmd_prof_tag(MMD_FAM_DW, "dwpyr n%lld C%lld f%lld b%lld", a.pyr.n, C, flip, nb);
  mmd_prof_begin(MMD_FAM_DW, stream);
  static const int rows_mode = getenv("MMD_DW_ROWS_PYR") ? atoi(getenv("MMD_DW_ROWS_PYR")) : (getenv("MMD_DW_ROWS") ? atoi(getenv("MMD_DW_ROWS")) : 1);
  const bool pro = in_scale || in_stats || in_act != MMD_ACT_NONE;
  if (rows_mode && !pro && C >= 64 && a.pyr.W[0] >= 64) {
    // row-streaming form (dw3_rows_kernel): 4 columns per thread, 4 rows per block on every level.  Only without a producer transform
    // (the heads' input gradients): with 4-row blocks the transform would run on 2.25x the elements, against the tile's 1.56x
    // (in the step: flipped 16.2 -> 14.3 us per launch, forward with BN + swish prologue 17.9 -> 22.7)
    DwRowsGeom gm{1, 1, 4};
    int nr = 0;
    for (int l = 0; l < a.pyr.n; ++l) {
      a.pyr.blk0[l] = nr;
      nr += a.B * a.cchunks * cdiv(a.pyr.W[l], 64) * cdiv(a.pyr.H[l], gm.rh);
    }
    for (int l = a.pyr.n; l <= MMD_MAX_LEV; ++l) a.pyr.blk0[l] = nr;
    if (dw_grad) {
      a.bz = wg_x; a.bscale = wg_scale; a.bshift = wg_shift; a.wg_act = wg_act; a.dwg = dw_grad;
      a.bmean = wg_mean; a.binvstd = wg_invstd; a.stats = bn_sums;
      hipLaunchKernelGGL((dw3_rows_kernel<4, 16, false, 0, true>), dim3(nr), dim3(256), 0, stream, a, gm);
      dw_grad = nullptr; bn_sums = nullptr;               // done
    } else {
      hipLaunchKernelGGL((dw3_rows_kernel<4, 16, false, 0>), dim3(nr), dim3(256), 0, stream, a, gm);
    }
  } else {
    if (pro) hipLaunchKernelGGL((dw_fwd_kernel<3, 1, 16, true, 0>), dim3(nb), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((dw_fwd_kernel<3, 1, 16, false, 0>), dim3(nb), dim3(256), 0, stream, a);
  }
  double rows = a.pyr.row0[a.pyr.n];
  mmd_prof_end(MMD_FAM_DW, stream, 2.0 * rows * C * 9, 8.0 * rows * C);
  int rc = mmd_check_launch();
  if (rc == MMD_OK && dw_grad)      // geometry without the fused form: the weight gradient by its own launch, same stream
    rc = mmd_dwconv3_pyr_bwd_weight(wg_x, x, dw_grad, pyr_desc, C, wg_scale, wg_shift, wg_act, lev_stride, stream);
  if (rc == MMD_OK && bn_sums)      // ... and the BatchNorm backward sums by theirs
    rc = mmd_bn_bwd_reduce_pyr(y, wg_x, wg_scale, wg_shift, wg_mean, wg_invstd, wg_act, pyr_desc, lev_stride, nullptr, bn_sums, C, stream);
  return rc;
}

// ---- input gradient -------------------------------------------------------------------
// stride 1: correlation with the flipped kernel and padding (k-1-pad) -> the forward kernel.
// stride 2: gather form, <= ceil(k/2)^2 taps per input pixel.
template <int K>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int B, int H, int W, int C,
                                                            int OH, int OW, int pad_t, int pad_l, int accumulate) {
  const int c4n = C >> 2;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)B * H * W * c4n;
  if (idx >= total) return;
  int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
  int iw = (int)(pix % W); pix /= W;
  int ih = (int)(pix % H); int b = (int)(pix / H);
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int t = ih + pad_t - i;
    if (t < 0 || (t & 1)) continue;
    int oh = t >> 1;
    if (oh >= OH) continue;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      int u = iw + pad_l - j;
      if (u < 0 || (u & 1)) continue;
      int ow = u >> 1;
      if (ow >= OW) continue;
      float4 g = mmd_ld4(dy + (((size_t)b * OH + oh) * OW + ow) * C + c);
      float4 wv = mmd_ld4(w + (size_t)(i * K + j) * C + c);
      acc.x += g.x * wv.x; acc.y += g.y * wv.y; acc.z += g.z * wv.z; acc.w += g.w * wv.w;
    }
  }
  float* o = dx + (((size_t)b * H + ih) * W + iw) * C + c;
  if (accumulate) { float4 p = mmd_ld4(o); acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
  mmd_st4(o, acc);
}

// The same gather with the BatchNorm(+swish)-backward sums of the tensor the gradient belongs to taken on the way (round 5): dx is the gradient
// w.r.t. a0 = swish(z * scale + shift), so sum g' and sum g' * xhat (g' = dx * swish'(u)) of that BatchNorm ride here, as they do in the stride-1
// launches, instead of a reduce pass that re-reads dx and z (the four stride-2 MBConv blocks: mmd_bn_bwd_reduce launches of 100 / 40 / 15 / 10 us
// on the backward chain).  Sums want few, fat blocks - one block per (image, row block, 64-channel chunk), a thread keeps ONE channel quad and
// walks the block's pixels, two float4 accumulators; the direct form above has one block per 256 (pixel, quad) items, 49 152 blocks on the
// 256^2 map, i.e. as many rounds of same-address f64 atomics.
// WG: the conv's weight gradient rides along too - dW[i][j] = sum dy[oh, ow] * a0[2 oh + i - pad, 2 ow + j - pad] is, seen from the input pixel, the
// product of the gathered dy values with a0 = swish(u) at that pixel, and the sigmoid is at hand from swish'.  A thread walks pixels of ONE
// parity class (pixel lanes 0 .. 3 of a 2 x 2 cell), so its <= ceil(K / 2)^2 reachable taps are the same all along: that many float4 accumulators,
// LDS atomics into a [K * K][64] tile at the end, K * K * 64 global atomics per block (the stride-1 launches do the same; the separate
// dw_wgrad_kernel<K, 2> launches were 4 x ~45 us of chip-filling work on the side stream).
template <int K, bool WG>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_sums_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int B, int H, int W, int C, int OH, int OW,
                                                                 int pad_t, int pad_l, const float* __restrict__ bz,
                                                                 const float* __restrict__ bscale, const float* __restrict__ bshift,
                                                                 const float* __restrict__ bmean, const float* __restrict__ binvstd,
                                                                 double* stats, double* stats_ws, int ws_slots, int rows_per_block,
                                                                 int rowblocks, int cchunks, float* __restrict__ dwg) {
  __shared__ float sRed[2 * 4 * 64];
  __shared__ float sW[K * K * 64];
  __shared__ float sWg[WG ? K * K * 64 : 1];
  const int tid = threadIdx.x, c4 = (tid & 15) * 4, pl_ = tid >> 4;      // 16 pixel lanes x 16 channel quads
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);      // (XCD-aware order: neighbouring tiles, whose halos overlap, behind one L2)
  const int cc = bid % cchunks; bid /= cchunks;
  const int rb = bid % rowblocks; bid /= rowblocks;
  const int b = bid, c = cc * 64 + c4;
  const bool cok = c < C;
  const int cs = cok ? c : 0;
  for (int i = tid; i < K * K * 16; i += 256) {
    const int t = i >> 4, q = (i & 15) * 4;
    *reinterpret_cast<float4*>(&sW[t * 64 + q]) = (cc * 64 + q < C) ? mmd_ld4(w + (size_t)t * C + cc * 64 + q) : make_float4(0, 0, 0, 0);
    if (WG) *reinterpret_cast<float4*>(&sWg[t * 64 + q]) = make_float4(0, 0, 0, 0);
  }
  const float4 sc = mmd_ld4(bscale + cs), sh = mmd_ld4(bshift + cs), mu = mmd_ld4(bmean + cs), is = mmd_ld4(binvstd + cs);
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
  const int ih0 = rb * rows_per_block, ih1 = min(ih0 + rows_per_block, H);      // rows_per_block is even (host)
  __syncthreads();
  constexpr int NT = (K + 1) / 2;      // taps per dimension that can reach an input pixel: those with the parity of (ih + pad_t)
  // pixels by 2 x 2 cells: pixel lane = (cell lane, row parity, column parity); the thread's reachable taps are i0 + 2 ii, j0 + 2 jj
  const int py = (pl_ >> 1) & 1, px = pl_ & 1, cl = pl_ >> 2;
  const int i0 = (ih0 + py + pad_t) & 1, j0 = (px + pad_l) & 1;
  const int cw = (W + 1) >> 1, ncell = ((ih1 - ih0 + 1) >> 1) * cw;
  float4 wacc[WG ? NT * NT : 1];
  if (WG) {
#pragma unroll
    for (int t = 0; t < NT * NT; ++t) wacc[t] = make_float4(0, 0, 0, 0);
  }
  for (int q = cl; q < ncell; q += 4) {
    const int ih = ih0 + 2 * (q / cw) + py, iw = 2 * (q % cw) + px;
    const bool pok = cok && ih < ih1 && iw < W;
    const int ihc = min(ih, H - 1), iwc = min(iw, W - 1);
    const size_t off = (((size_t)b * H + ihc) * W + iwc) * C + cs;
    const float4 zz = mmd_ld4(bz + off);
    // every load of the pixel unconditional from a clamped address, masked afterwards (a guarded load is a dependent round trip)
    float4 gv[NT * NT];
    unsigned vm = 0u;
#pragma unroll
    for (int ii = 0; ii < NT; ++ii) {
      const int i = i0 + 2 * ii, oh = (ih + pad_t - i) >> 1;
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const int j = j0 + 2 * jj, ow = (iw + pad_l - j) >> 1;
        if (pok && i < K && j < K && ih + pad_t - i >= 0 && oh < OH && iw + pad_l - j >= 0 && ow < OW) vm |= 1u << (ii * NT + jj);
        gv[ii * NT + jj] = mmd_ld4(dy + (((size_t)b * OH + min(max(oh, 0), OH - 1)) * OW + min(max(ow, 0), OW - 1)) * C + cs);
      }
    }
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int ii = 0; ii < NT; ++ii)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const int i = min(i0 + 2 * ii, K - 1), j = min(j0 + 2 * jj, K - 1);
        const float4 k_ = *reinterpret_cast<const float4*>(&sW[(i * K + j) * 64 + c4]);
        const float m = ((vm >> (ii * NT + jj)) & 1u) ? 1.f : 0.f;
        float4& g = gv[ii * NT + jj];
        g.x *= m; g.y *= m; g.z *= m; g.w *= m;
        acc.x += g.x * k_.x; acc.y += g.y * k_.y; acc.z += g.z * k_.z; acc.w += g.w * k_.w;
      }
    if (pok) {
      mmd_st4(dx + off, acc);
      const float ux = zz.x * sc.x + sh.x, uy = zz.y * sc.y + sh.y, uz = zz.z * sc.z + sh.z, uw = zz.w * sc.w + sh.w;
      const float sx = mmd_sigmoid(ux), sy = mmd_sigmoid(uy), sz = mmd_sigmoid(uz), sw = mmd_sigmoid(uw);
      const float4 gp = make_float4(acc.x * sx * (1.f + ux * (1.f - sx)), acc.y * sy * (1.f + uy * (1.f - sy)),
                                    acc.z * sz * (1.f + uz * (1.f - sz)), acc.w * sw * (1.f + uw * (1.f - sw)));
      s4.x += gp.x; s4.y += gp.y; s4.z += gp.z; s4.w += gp.w;
      q4.x += gp.x * (zz.x - mu.x) * is.x; q4.y += gp.y * (zz.y - mu.y) * is.y;
      q4.z += gp.z * (zz.z - mu.z) * is.z; q4.w += gp.w * (zz.w - mu.w) * is.w;
      if (WG) {
        const float4 a0 = make_float4(ux * sx, uy * sy, uz * sz, uw * sw);
#pragma unroll
        for (int t = 0; t < NT * NT; ++t) {      // (masked taps hold zeros)
          wacc[t].x += gv[t].x * a0.x; wacc[t].y += gv[t].y * a0.y; wacc[t].z += gv[t].z * a0.z; wacc[t].w += gv[t].w * a0.w;
        }
      }
    }
  }
  // lanes l, l ^ 16, l ^ 32, l ^ 48 of a wave share the channel quad; then the four waves through LDS
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
    q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 16) {
    *reinterpret_cast<float4*>(&sRed[wave * 64 + c4]) = s4;
    *reinterpret_cast<float4*>(&sRed[4 * 64 + wave * 64 + c4]) = q4;
  }
  if (WG) {
#pragma unroll
    for (int ii = 0; ii < NT; ++ii)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const int i = i0 + 2 * ii, j = j0 + 2 * jj;
        if (i < K && j < K) {
          float* d = &sWg[(i * K + j) * 64 + c4];
          const float4 v = wacc[ii * NT + jj];
          atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
        }
      }
  }
  __syncthreads();
  if (tid < 64 && cc * 64 + tid < C) {
    const float vs = sRed[tid] + sRed[64 + tid] + sRed[128 + tid] + sRed[192 + tid];
    const float vq = sRed[256 + tid] + sRed[320 + tid] + sRed[384 + tid] + sRed[448 + tid];
    double* st = stats_ws ? stats_ws + (size_t)((blockIdx.x / cchunks) % ws_slots) * 2 * C : stats;
    atomicAdd(&st[cc * 64 + tid], (double)vs);
    atomicAdd(&st[C + cc * 64 + tid], (double)vq);
  }
  if (WG)
    for (int i = tid; i < K * K * 64; i += 256) {
      const int t = i >> 6, q = i & 63;
      if (cc * 64 + q < C) atomicAdd(&dwg[(size_t)t * C + cc * 64 + q], sWg[i]);
    }
}

// dx[B,H,W,C] (=) dwconv^T(dy[B,OH,OW,C], w)
extern "C" int mmd_dwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int k,
                                   int stride, const float* bn_z, const float* bn_scale, const float* bn_shift,
                                   const float* bn_mean, const float* bn_invstd, double* bn_sums, double* stats_ws,
                                   int ws_slots, float* dw_grad, hipStream_t stream) {
  if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return MMD_EINVAL;
  if (bn_sums && (!bn_z || !bn_scale || !bn_shift || !bn_mean || !bn_invstd)) return MMD_EINVAL;
  if (dw_grad && !bn_sums) return MMD_EINVAL;        // the weight gradient rides on the BatchNorm-sum forms (it needs a0 = swish(u))
  int OH, OW;
  int pt = same_pad_lo(H, k, stride, &OH), pl = same_pad_lo(W, k, stride, &OW);
  mmd_prof_tag(MMD_FAM_DW_BWD, "dwbd H%lld C%lld k%lld s%lld", H, C, k, stride);
  mmd_prof_begin(MMD_FAM_DW_BWD, stream);
  int rc;
  if (stride == 1) {
    DwArgs a{};
    a.x = dy; a.w = w; a.y = dx; a.B = B; a.H = H; a.W = W; a.C = C; a.OH = H; a.OW = W;
    a.pad_t = k - 1 - pt; a.pad_l = k - 1 - pl; a.flip = 1;
    if (bn_sums) {
      a.stats = bn_sums; a.stats_ws = stats_ws; a.ws_slots = ws_slots;
      a.bz = bn_z; a.bscale = bn_scale; a.bshift = bn_shift; a.bmean = bn_mean; a.binvstd = bn_invstd;
    }
    a.dwg = dw_grad;
    rc = (k == 3 && !dw_grad) ? dw3_rows_launch(a, stream) : 1;      // (the row-streaming kernel has no weight-gradient form)
    if (rc == 1) rc = (k == 3) ? dw_fwd_launch_31(a, stream) : dw_fwd_launch_51(a, stream);
  } else if (bn_sums) {
    // ~2048 blocks: rows per block from the map's size; slotted sums when more than MMD_STATS_DEPTH blocks would meet on one address
    const int cch = cdiv(C, 64);
    int rpb = (int)cdiv((long long)B * H * cch, 2048); if (rpb < 2) rpb = 2; rpb += rpb & 1; if (rpb > H) rpb = H + (H & 1);      // even: a thread keeps one row parity
    const int rbl = cdiv(H, rpb);
    const long long per_addr = (long long)B * rbl;
    double* ws = (stats_ws && ws_slots > 1 && per_addr > MMD_STATS_DEPTH) ? stats_ws : nullptr;
    const dim3 grid((unsigned)((long long)B * rbl * cch));
#define MMD_S2_GO(KK, WG_) hipLaunchKernelGGL((dw_bwd_data_s2_sums_kernel<KK, WG_>), grid, dim3(256), 0, stream, dy, w, dx, B, H, W, C, OH, OW, pt, pl, bn_z, bn_scale, \
                                              bn_shift, bn_mean, bn_invstd, bn_sums, ws, ws ? ws_slots : 1, rpb, rbl, cch, dw_grad)
    if (k == 3) { if (dw_grad) MMD_S2_GO(3, true); else MMD_S2_GO(3, false); }
    else { if (dw_grad) MMD_S2_GO(5, true); else MMD_S2_GO(5, false); }
#undef MMD_S2_GO
    if (ws) mmd_stats_fold(bn_sums, ws, ws_slots, 2 * C, stream);
    rc = mmd_check_launch();
  } else {
    size_t total = (size_t)B * H * W * (C >> 2);
    if (k == 3) hipLaunchKernelGGL(dw_bwd_data_s2_kernel<3>, dim3(cdiv(total, 256)), dim3(256), 0, stream, dy, w, dx, B, H, W, C, OH, OW, pt, pl, 0);
    else hipLaunchKernelGGL(dw_bwd_data_s2_kernel<5>, dim3(cdiv(total, 256)), dim3(256), 0, stream, dy, w, dx, B, H, W, C, OH, OW, pt, pl, 0);
    rc = mmd_check_launch();
  }
  mmd_prof_end(MMD_FAM_DW_BWD, stream, 2.0 * B * OH * OW * (double)C * k * k, 4.0 * ((double)B * H * W * C + (double)B * OH * OW * C));
  return rc;
}

static int mmd_dwconv_bwd_data_bn1_w16(const float* g1, const float* z1, const float* w, float* dx, int B, int H, int W, int C, int k,
                                           const float* q_scale, const float* q_shift, const float* q_mean, const float* q_invstd,
                                           const double* q_sums, long long q_count, const float* q_gate, const float* q_add,
                                           float* q_dgamma, float* q_dbeta,
                                           const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                           const float* bn_invstd, double* bn_sums, double* stats_ws, int ws_slots, float* dw_grad,
                                           int w16, hipStream_t stream);
// Input gradient of an MBConv block's stride-1 depthwise conv with the BatchNorm-1 (+swish, squeeze-excite gate / pooled term) backward
// evaluated in the prologue instead of by mmd_bn_bwd_apply: dY = BnBwd1(g1, z1; gate, dpooled, sums1) is never written to HBM.  Always
// together with the BatchNorm-0 sums (bn_*) and the conv's weight gradient, as the engine runs these layers.  C >= 64 channels (64-wide chunks).
extern "C" int mmd_dwconv_bwd_data_bn1(const float* g1, const float* z1, const float* w, float* dx, int B, int H, int W, int C, int k,
                                       const float* q_scale, const float* q_shift, const float* q_mean, const float* q_invstd,
                                       const double* q_sums, long long q_count, const float* q_gate, const float* q_add,
                                       float* q_dgamma, float* q_dbeta,
                                       const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                       const float* bn_invstd, double* bn_sums, double* stats_ws, int ws_slots, float* dw_grad,
                                       hipStream_t stream) {
  return mmd_dwconv_bwd_data_bn1_w16(g1, z1, w, dx, B, H, W, C, k, q_scale, q_shift, q_mean, q_invstd, q_sums, q_count, q_gate, q_add, q_dgamma,
                                     q_dbeta, bn_z, bn_scale, bn_shift, bn_mean, bn_invstd, bn_sums, stats_ws, ws_slots, dw_grad, 0, stream);
}
// bf16 storage: w16 bit 0 = g1, bit 1 = dx, bit 2 = z1, bit 3 = bn_z are bf16 arrays
static int mmd_dwconv_bwd_data_bn1_w16(const float* g1, const float* z1, const float* w, float* dx, int B, int H, int W, int C, int k,
                                           const float* q_scale, const float* q_shift, const float* q_mean, const float* q_invstd,
                                           const double* q_sums, long long q_count, const float* q_gate, const float* q_add,
                                           float* q_dgamma, float* q_dbeta,
                                           const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                           const float* bn_invstd, double* bn_sums, double* stats_ws, int ws_slots, float* dw_grad,
                                           int w16, hipStream_t stream) {
  if (w16) return MMD_EINVAL;      // (bf16 storage of the wide tensors was deleted in round 6: the branches are compiled out, common.h)
  if (!g1 || !z1 || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C < 64 || (C & 3) || (k != 3 && k != 5)) return MMD_EINVAL;
  if (!q_scale || !q_shift || !q_mean || !q_invstd || !q_sums || q_count <= 0 || !q_gate || !q_add) return MMD_EINVAL;
  if ((q_dgamma == nullptr) != (q_dbeta == nullptr)) return MMD_EINVAL;
  if (!bn_z || !bn_scale || !bn_shift || !bn_mean || !bn_invstd || !bn_sums || !dw_grad) return MMD_EINVAL;
  int OH, OW;
  const int pt = same_pad_lo(H, k, 1, &OH), pl = same_pad_lo(W, k, 1, &OW);
  DwArgs a{};
  a.x = g1; a.w = w; a.y = dx; a.B = B; a.H = H; a.W = W; a.C = C; a.OH = H; a.OW = W;
  a.pad_t = k - 1 - pt; a.pad_l = k - 1 - pl; a.flip = 1;
  a.stats = bn_sums; a.stats_ws = stats_ws; a.ws_slots = ws_slots;
  a.bz = bn_z; a.bscale = bn_scale; a.bshift = bn_shift; a.bmean = bn_mean; a.binvstd = bn_invstd;
  a.dwg = dw_grad;
  a.q_z = z1; a.q_gate = q_gate; a.q_add = q_add; a.q_scale = q_scale; a.q_shift = q_shift; a.q_mean = q_mean; a.q_invstd = q_invstd;
  a.q_sums = q_sums; a.q_inv_count = 1.0 / (double)q_count; a.q_dgamma = q_dgamma; a.q_dbeta = q_dbeta;
  a.x16 = w16 & 1; a.y16 = (w16 >> 1) & 1; a.qz16 = (w16 >> 2) & 1; a.bz16 = (w16 >> 3) & 1;
  mmd_prof_tag(MMD_FAM_DW_BWD, "dwbd1 H%lld C%lld k%lld s%lld", H, C, k, 1);
  mmd_prof_begin(MMD_FAM_DW_BWD, stream);
  const int rc = (k == 3) ? (dw_lanes_for(C) == 8 ? dw_fwd_launch<3, 1, 8>(a, stream) : dw_fwd_launch<3, 1, 16>(a, stream))
                          : (dw_lanes_for(C) == 8 ? dw_fwd_launch<5, 1, 8>(a, stream) : dw_fwd_launch<5, 1, 16>(a, stream));
  mmd_prof_end(MMD_FAM_DW_BWD, stream, 2.0 * B * OH * OW * (double)C * k * k, 4.0 * 4 * (double)B * H * W * C);
  return rc;
}

// ---- weight gradient: dw[tap,c] += sum_{b,oh,ow} dy[b,oh,ow,c] * pro(x)[b, oh*s+i-pad, ow*s+j-pad, c]
// Same tiling as the forward; each block walks `tiles_per_block` tiles of one (image, 64-channel chunk)
// keeping k*k float4 partial sums in registers, then one shuffle + LDS reduction and 256-B atomics.
struct DwWgArgs {
  const float* x; const float* dy; float* dw;
  int B, H, W, C, OH, OW, pad_t, pad_l;
  const float* in_scale; const float* in_shift; int in_act;
  int tiles_h, tiles_w, cchunks, nsplit;
  Pyr pyr; long long lev_stride; int nsplit_lev[MMD_MAX_LEV];
};

template <int K, int S, bool PRO>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWgArgs a) {
  using Cf = DwCfg<K, S>;
  static_assert(Cf::IH * Cf::IW >= 4 * K * K, "reduction scratch aliases the input tile");
  __shared__ float sIn[Cf::IH * Cf::IW * 64];
  float* sRed = sIn;                      // reused after the tile loop (behind a barrier)
  const int tid = threadIdx.x;
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);      // (XCD-aware order: neighbouring tiles, whose halos overlap, behind one L2)
  int H = a.H, W = a.W, OH = a.OH, OW = a.OW, tiles_h = a.tiles_h, tiles_w = a.tiles_w, nsplit = a.nsplit, lev = 0;
  size_t ro = 0;
  if (a.pyr.n) {
    int blk = 0;
    H = a.pyr.H[0]; W = a.pyr.W[0]; nsplit = a.nsplit_lev[0];
#pragma unroll
    for (int i = 1; i < MMD_MAX_LEV; ++i)
      if (i < a.pyr.n && bid >= a.pyr.blk0[i]) {
        lev = i; H = a.pyr.H[i]; W = a.pyr.W[i]; ro = (size_t)a.pyr.row0[i]; blk = a.pyr.blk0[i]; nsplit = a.nsplit_lev[i];
      }
    bid -= blk; OH = H; OW = W; ro *= a.C;
    tiles_h = (H + Cf::TH - 1) / Cf::TH; tiles_w = (W + Cf::TW - 1) / Cf::TW;
  }
  const int sp = bid % nsplit; bid /= nsplit;
  const int cc = bid % a.cchunks; bid /= a.cchunks;
  const int b = bid;
  const int c0 = cc * 64, c4 = (tid & 15) * 4, c = c0 + c4;
  const bool cok = c < a.C;
  const int p = tid >> 4;
  const int orow = p / (Cf::TW / Cf::R);
  const int ocol0 = (p % (Cf::TW / Cf::R)) * Cf::R;
  DwView fa;
  fa.x = a.x + ro; fa.H = H; fa.W = W; fa.C = a.C; fa.act = a.in_act;
  {
    BnLive none{};
    const long long lo = (long long)lev * a.lev_stride;
    dw_in_coef(a.in_scale ? a.in_scale + lo : nullptr, a.in_scale ? a.in_shift + lo : nullptr, none, c, cok, fa);
  }
  const float* const dyp = a.dy + ro;

  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = make_float4(0, 0, 0, 0);

  const int ntiles = tiles_h * tiles_w;
  for (int tile = sp; tile < ntiles; tile += nsplit) {
    const int th = tile / tiles_w, tw = tile % tiles_w;
    const int oh0 = th * Cf::TH, ow0 = tw * Cf::TW;
    __syncthreads();
    dw_stage_input<K, S, 16, PRO>(fa, sIn, b, oh0 * S - a.pad_t, ow0 * S - a.pad_l, c0, tid);
    __syncthreads();
    float4 g[Cf::R];
    const int oh = oh0 + orow;
#pragma unroll
    for (int o = 0; o < Cf::R; ++o) {
      int ow = ow0 + ocol0 + o;
      g[o] = (cok && oh < OH && ow < OW) ? mmd_ld4(dyp + (((size_t)b * OH + oh) * OW + ow) * a.C + c)
                                              : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
      float4 in[Cf::SEG];
      const float* prow = &sIn[((orow * S + i) * Cf::IW + ocol0 * S) * 64 + c4];
#pragma unroll
      for (int q = 0; q < Cf::SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * 64);
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int o = 0; o < Cf::R; ++o) {
          acc[i * K + j].x += g[o].x * in[o * S + j].x; acc[i * K + j].y += g[o].y * in[o * S + j].y;
          acc[i * K + j].z += g[o].z * in[o * S + j].z; acc[i * K + j].w += g[o].w * in[o * S + j].w;
        }
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    float4 v = acc[t];
    v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
    v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
    if (lane < 16) *reinterpret_cast<float4*>(&sRed[(wave * K * K + t) * 64 + c4]) = v;
  }
  __syncthreads();
  for (int i = tid; i < K * K * 64; i += 256) {
    int t = i >> 6, q = i & 63;
    if (c0 + q < a.C) {
      float v = sRed[(0 * K * K + t) * 64 + q] + sRed[(1 * K * K + t) * 64 + q] + sRed[(2 * K * K + t) * 64 + q] +
                sRed[(3 * K * K + t) * 64 + q];
      atomicAdd(&a.dw[(size_t)t * a.C + c0 + q], v);
    }
  }
}

// 3x3 / stride 1 weight gradient on the row-streaming window of dw3_rows_kernel: a thread keeps three (transformed) input rows of its
// R + 2 columns and the 9 tap sums of its 4 channels in registers and walks down `rh` rows; per row it loads the next input row and
// the dY row.  One shuffle + LDS reduction and 9 x CC atomics per block at the end, as in the tile kernel.
template <int R, int LW, bool PRO>
__global__ __launch_bounds__(256) void dw3_wgrad_rows_kernel(DwWgArgs a, DwRowsGeom gm) {
  constexpr int CC = 4 * LW;
  __shared__ float sRed[4 * 9 * CC];
  const int tid = threadIdx.x, c4 = (tid & (LW - 1)) * 4, strip = tid / LW;
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);      // (XCD-aware order: neighbouring tiles, whose halos overlap, behind one L2)
  const int cc = bid % a.cchunks; bid /= a.cchunks;
  const int cb = bid % gm.colblocks; bid /= gm.colblocks;
  const int rb = bid % gm.rowblocks; bid /= gm.rowblocks;
  const int b = bid, c0 = cc * CC, c = c0 + c4;
  const bool cok = c < a.C;
  const int H = a.H, W = a.W, C = a.C;
  const int ow0 = (cb * (256 / LW) + strip) * R, oh0 = rb * gm.rh, oh1 = min(oh0 + gm.rh, H);
  DwView v;
  v.act = a.in_act;
  if (PRO) { BnLive none{}; dw_in_coef(a.in_scale, a.in_shift, none, c, cok, v); }
  const float* const xb = a.x + (size_t)b * H * W * C + (cok ? c : 0);
  const float* const gb = a.dy + (size_t)b * H * W * C + (cok ? c : 0);
  // unconditional, clamped, masked row loads issued one row ahead: as in dw3_rows_kernel
  unsigned colok = 0;
  int colo[R + 2];
#pragma unroll
  for (int q = 0; q < R + 2; ++q) {
    const int iw = ow0 - 1 + q;
    if (cok && iw >= 0 && iw < W) colok |= 1u << q;
    colo[q] = min(max(iw, 0), W - 1) * C;
  }
  auto issue_row = [&](int ih, float4 (&dst)[R + 2]) {
    const float* p = xb + (long long)min(max(ih, 0), H - 1) * W * C;
#pragma unroll
    for (int q = 0; q < R + 2; ++q) dst[q] = mmd_ld4(p + colo[q]);
  };
  auto finish_row = [&](int ih, float4 (&dst)[R + 2]) {
    const unsigned m = (ih >= 0 && ih < H) ? colok : 0u;
#pragma unroll
    for (int q = 0; q < R + 2; ++q) {
      float4 u = dst[q];
      if (PRO) {
        if (v.xf) { u.x = u.x * v.sc.x + v.sh.x; u.y = u.y * v.sc.y + v.sh.y; u.z = u.z * v.sc.z + v.sh.z; u.w = u.w * v.sc.w + v.sh.w; }
        if (v.act == MMD_ACT_SWISH) { u.x = mmd_swish(u.x); u.y = mmd_swish(u.y); u.z = mmd_swish(u.z); u.w = mmd_swish(u.w); }
      }
      const bool ok = (m >> q) & 1u;
      dst[q] = make_float4(ok ? u.x : 0.f, ok ? u.y : 0.f, ok ? u.z : 0.f, ok ? u.w : 0.f);
    }
  };
  auto issue_g = [&](int oh, float4 (&dst)[R]) {
    const float* p = gb + (long long)min(oh, H - 1) * W * C;
#pragma unroll
    for (int o = 0; o < R; ++o) dst[o] = mmd_ld4(p + colo[o + 1]);
  };
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0, 0, 0, 0);
  auto out_row = [&](const float4 (&r0)[R + 2], const float4 (&r1)[R + 2], const float4 (&r2)[R + 2], const float4 (&gr)[R]) {
    float4 g[R];
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const bool ok = (colok >> (o + 1)) & 1u;
      g[o] = make_float4(ok ? gr[o].x : 0.f, ok ? gr[o].y : 0.f, ok ? gr[o].z : 0.f, ok ? gr[o].w : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int o = 0; o < R; ++o) {
        acc[j].x += g[o].x * r0[o + j].x; acc[j].y += g[o].y * r0[o + j].y; acc[j].z += g[o].z * r0[o + j].z; acc[j].w += g[o].w * r0[o + j].w;
        acc[3 + j].x += g[o].x * r1[o + j].x; acc[3 + j].y += g[o].y * r1[o + j].y; acc[3 + j].z += g[o].z * r1[o + j].z; acc[3 + j].w += g[o].w * r1[o + j].w;
        acc[6 + j].x += g[o].x * r2[o + j].x; acc[6 + j].y += g[o].y * r2[o + j].y; acc[6 + j].z += g[o].z * r2[o + j].z; acc[6 + j].w += g[o].w * r2[o + j].w;
      }
  };
  float4 w0[R + 2], w1[R + 2], w2[R + 2], w3[R + 2], za[R], zb[R];
  issue_row(oh0 - 1, w0); issue_row(oh0, w1); issue_row(oh0 + 1, w2); issue_g(oh0, za);
  finish_row(oh0 - 1, w0); finish_row(oh0, w1);
  for (int oh = oh0; oh < oh1; oh += 4) {
    issue_row(oh + 2, w3); issue_g(oh + 1, zb); finish_row(oh + 1, w2); out_row(w0, w1, w2, za);
    if (oh + 1 < oh1) { issue_row(oh + 3, w0); issue_g(oh + 2, za); finish_row(oh + 2, w3); out_row(w1, w2, w3, zb); }
    if (oh + 2 < oh1) { issue_row(oh + 4, w1); issue_g(oh + 3, zb); finish_row(oh + 3, w0); out_row(w2, w3, w0, za); }
    if (oh + 3 < oh1) { issue_row(oh + 5, w2); issue_g(oh + 4, za); finish_row(oh + 4, w1); out_row(w3, w0, w1, zb); }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float4 x = acc[t];
#pragma unroll
    for (int o = LW; o < 64; o <<= 1) {
      x.x += __shfl_xor(x.x, o, 64); x.y += __shfl_xor(x.y, o, 64); x.z += __shfl_xor(x.z, o, 64); x.w += __shfl_xor(x.w, o, 64);
    }
    if (lane < LW) *reinterpret_cast<float4*>(&sRed[(wave * 9 + t) * CC + c4]) = x;
  }
  __syncthreads();
  for (int i = tid; i < 9 * CC; i += 256) {
    const int t = i / CC, q = i - t * CC;
    if (c0 + q < C)
      atomicAdd(&a.dw[(size_t)t * C + c0 + q], sRed[(0 * 9 + t) * CC + q] + sRed[(1 * 9 + t) * CC + q] + sRed[(2 * 9 + t) * CC + q] + sRed[(3 * 9 + t) * CC + q]);
  }
}

template <int R, int LW>
static int dw3_wgrad_rows_go(DwWgArgs& a, bool pro, hipStream_t st) {
  DwRowsGeom gm;
  a.cchunks = cdiv(a.C, 4 * LW);
  gm.colblocks = cdiv(a.W, (256 / LW) * R);
  const long long per_row = (long long)a.B * a.cchunks * gm.colblocks;
  int rh = (int)((long long)a.H * per_row / 1024);
  if (rh < 4) rh = 4; if (rh > a.H) rh = a.H;
  gm.rh = rh; gm.rowblocks = cdiv(a.H, rh);
  const dim3 grid((unsigned)(per_row * gm.rowblocks)), blk(256);
  if (pro) hipLaunchKernelGGL((dw3_wgrad_rows_kernel<R, LW, true>), grid, blk, 0, st, a, gm);
  else hipLaunchKernelGGL((dw3_wgrad_rows_kernel<R, LW, false>), grid, blk, 0, st, a, gm);
  return mmd_check_launch();
}

// -> 1 when the geometry is left to the tile kernel.  Stand-alone (B = 8, producer BN + swish on x): 256^2 x 16 67.3 -> 31.9 us, but
// 128^2 x 144 45.2 -> 57.5 and 64^2 x 112 21.0 -> 22.7 (the tile kernel already streams these at 3.3 / 1.4 TB/s; its 64-channel chunk
// is what wastes the thin layer), so only the 16-channel-chunk form is used.
static int dw3_wgrad_rows_launch(DwWgArgs& a, hipStream_t st) {
  static const int mode = getenv("MMD_DW_ROWS") ? atoi(getenv("MMD_DW_ROWS")) : 1;
  // (C in (16, 32] - block 0's depthwise conv on the stem's 256^2 x 32 output, the last leaf of the step's tail - as 32-channel chunks: the
  // tile kernel's 64-channel chunk leaves half of its lanes idle there)
  if (!mode || a.C > 32 || a.W < 256) return 1;
  if (a.C > 16) return dw3_wgrad_rows_go<4, 8>(a, a.in_scale || a.in_act != MMD_ACT_NONE, st);
  return dw3_wgrad_rows_go<4, 4>(a, a.in_scale || a.in_act != MMD_ACT_NONE, st);
}

template <int K, int S>
static int dw_wgrad_launch(DwWgArgs& a, hipStream_t st) {
  using Cf = DwCfg<K, S>;
  a.tiles_h = cdiv(a.OH, Cf::TH); a.tiles_w = cdiv(a.OW, Cf::TW); a.cchunks = cdiv(a.C, 64);
  int ntiles = a.tiles_h * a.tiles_w;
  int base = a.B * a.cchunks;
  int ns = cdiv(2048, base); if (ns > ntiles) ns = ntiles; if (ns < 1) ns = 1;
  a.nsplit = ns;
  if (a.in_scale || a.in_act != MMD_ACT_NONE) hipLaunchKernelGGL((dw_wgrad_kernel<K, S, true>), dim3((unsigned)(base * ns)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((dw_wgrad_kernel<K, S, false>), dim3((unsigned)(base * ns)), dim3(256), 0, st, a);
  return mmd_check_launch();
}

extern "C" int mmd_dwconv_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int k,
                                     int stride, const float* in_scale, const float* in_shift, int in_act,
                                     hipStream_t stream) {
  if (!x || !dy || !dw || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MMD_EINVAL;
  DwWgArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.B = B; a.H = H; a.W = W; a.C = C;
  a.pad_t = same_pad_lo(H, k, stride, &a.OH); a.pad_l = same_pad_lo(W, k, stride, &a.OW);
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  mmd_prof_tag(MMD_FAM_DW_BWD, "dwwg H%lld C%lld k%lld s%lld", H, C, k, stride);
  mmd_prof_begin(MMD_FAM_DW_BWD, stream);
  int rc = 1;
  if (k == 3 && stride == 1) rc = dw3_wgrad_rows_launch(a, stream);
  if (rc != 1) {}
  else if (k == 3 && stride == 1) rc = dw_wgrad_launch<3, 1>(a, stream);
  else if (k == 3) rc = dw_wgrad_launch<3, 2>(a, stream);
  else if (stride == 1) rc = dw_wgrad_launch<5, 1>(a, stream);
  else rc = dw_wgrad_launch<5, 2>(a, stream);
  mmd_prof_end(MMD_FAM_DW_BWD, stream, 2.0 * B * a.OH * a.OW * (double)C * k * k,
               4.0 * ((double)B * H * W * C + (double)B * a.OH * a.OW * C));
  return rc;
}

// Depthwise 3x3/s1 weight gradient over a whole pyramid (shared weights): dw[9,C] += sum over levels.
extern "C" int mmd_dwconv3_pyr_bwd_weight(const float* x, const float* dy, float* dw, const int* pyr_desc, int C,
                                          const float* in_scale, const float* in_shift, int in_act, long long lev_stride,
                                          hipStream_t stream) {
  if (!x || !dy || !dw || !pyr_desc || C <= 0 || (C & 3)) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MMD_EINVAL;
  DwWgArgs a{};
  if (mmd_make_pyr(a.pyr, pyr_desc)) return MMD_EINVAL;
  using Cf = DwCfg<3, 1>;
  a.x = x; a.dy = dy; a.dw = dw; a.B = a.pyr.B; a.C = C; a.pad_t = 1; a.pad_l = 1;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.lev_stride = lev_stride; a.cchunks = cdiv(C, 64);
  int nb = 0;
  for (int l = 0; l < a.pyr.n; ++l) {
    int ntiles = cdiv(a.pyr.H[l], Cf::TH) * cdiv(a.pyr.W[l], Cf::TW);
    int ns = ntiles / 8; if (ns < 1) ns = 1; if (ns > 16) ns = 16;
    a.nsplit_lev[l] = ns;
    a.pyr.blk0[l] = nb;
    nb += a.B * a.cchunks * ns;
  }
  for (int l = a.pyr.n; l <= MMD_MAX_LEV; ++l) a.pyr.blk0[l] = nb;
  mmd_prof_tag(MMD_FAM_DW_BWD, "dwwgpyr C%lld", C, 0, 0, 0);
  mmd_prof_begin(MMD_FAM_DW_BWD, stream);
  if (in_scale || in_act != MMD_ACT_NONE) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1, true>), dim3(nb), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dw_wgrad_kernel<3, 1, false>), dim3(nb), dim3(256), 0, stream, a);
  double rows = a.pyr.row0[a.pyr.n];
  mmd_prof_end(MMD_FAM_DW_BWD, stream, 2.0 * rows * C * 9, 8.0 * rows * C);
  return mmd_check_launch();
}
