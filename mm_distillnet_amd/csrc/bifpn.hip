// BiFPN fast-attention fusion nodes and the TF-SAME 3x3/s2 max-pool (zero padding participates in the
// max) on NHWC fp32 — CDNA4 / gfx950.
// Reference: BiFPN._forward_fast_attention (src/YetAnotherEfficientDet.py:320-392),
// MaxPool2dStaticSamePadding (src/YetAnotherEfficientNet.py:68-104), nn.Upsample(nearest, x2) (:223-226).
//   f = swish( w0*in0 [+ w*in1] [+ w*up2(u)] [+ w*pool(p)] ),  w = relu(theta)/(sum relu(theta) + 1e-4)
// Weights are assigned in the order of the non-null operands (in0, in1, up, pool), which is the
// reference's order for every node (top-down: in,up; bottom-up: in,td,pool; p7: in,pool).
// One thread = one pixel x 4 channels (float4, channel-contiguous, coalesced).
#include "common.h"

#define FUSE_EPS 1e-4f

struct FuseArgs {
  const float* in0; const float* in1; const float* up; const float* pl;
  const float* theta; int ntheta;
  int B, H, W, C, PH, PW, pad_t, pad_l;
  // Round 4, "lazy" operands of the trainable net: operand i (order in0, in1, up, pool) is the RAW output z_i of its producer's 1x1 conv and
  // its train-mode BatchNorm is applied while the operand is loaded, y_i = z_i * scale_i + shift_i - the producer node then needs no
  // mmd_affine_act launch (40 launches on the student's forward chain).  Forward: coefficients derived from the live batch sums (ost / oga /
  // obe / oic, as BnLive); backward: the finalized (osc, osh).  lazy = bit mask of the operands that carry a transform.
  const double* ost[4]; const float* oga[4]; const float* obe[4]; double oic[4];
  const float* osc[4]; const float* osh[4];
  int lazy;
  int g_images; long long g_w, g_bn;      // grouped frozen nets (common.h MmdGroup): whole-node eval kernel and fuse_dw_fwd_kernel
  // node backward, pooled operand (round 5): float offset inside the dynamic LDS of a [17 x 17][64] tile that collects the scattered gradient of
  // the block's fine-map region before it goes to memory (-1: scatter with global atomics, the round-3 form)
  int pl_lds_off;
};
// per-block coefficient table of the lazy operands in LDS: tab[(2 op + {0 scale, 1 shift}) * 64 + channel of the block's 64-channel chunk];
// a thread reads its quad where it needs it (kept out of registers: the node backward kernel runs at 190-240 VGPRs as it is)
struct FuseCoef {
  const float* tab; int c4;
  __device__ __forceinline__ float4 sc(int op) const { return *reinterpret_cast<const float4*>(tab + (2 * op) * 64 + c4); }
  __device__ __forceinline__ float4 sh(int op) const { return *reinterpret_cast<const float4*>(tab + (2 * op + 1) * 64 + c4); }
};
// fills tab [8][64] for the chunk starting at channel c0 (all threads of a 256-thread block call it; followed by a barrier at the caller)
__device__ __forceinline__ void fuse_coef_fill(const FuseArgs& a, int c0, float* tab) {
  for (int i = threadIdx.x; i < 4 * 64; i += 256) {
    const int op = i >> 6, cl = i & 63, c = c0 + cl;
    float sc = 1.f, sh = 0.f;
    if ((a.lazy >> op & 1) && c < a.C) {
      if (a.ost[op]) { BnLive bn; bn.stats = a.ost[op]; bn.gamma = a.oga[op]; bn.beta = a.obe[op]; bn.inv_count = a.oic[op]; bn.C = a.C; bn.eps = 1e-3f;
                       bn_live_coef(bn, c, sc, sh); }
      else { sc = a.osc[op][c]; sh = a.osh[op][c]; }
    }
    tab[(2 * op) * 64 + cl] = sc; tab[(2 * op + 1) * 64 + cl] = sh;
  }
}
__device__ __forceinline__ float4 fuse_aff4(const float4& v, const float4& sc, const float4& sh) {
  return make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
}

__device__ __forceinline__ void fuse_weights(const float* theta, int n, float* w) {
  float r[3] = {0.f, 0.f, 0.f}, s = 0.f;
  for (int i = 0; i < n; ++i) { r[i] = fmaxf(theta[i], 0.f); s += r[i]; }
  for (int i = 0; i < 3; ++i) w[i] = r[i] / (s + FUSE_EPS);
}

// aff: the source holds raw BatchNorm inputs - every in-image element is transformed (x * sc + sh) before the max; the zero padding is
// the padding of the TRANSFORMED map (the reference pads the BatchNorm output) and stays zero
__device__ __forceinline__ float4 pool_window(const float* __restrict__ src, int b, int oh, int ow, int c, int PH, int PW,
                                              int C, int pad_t, int pad_l, bool aff = false, float4 sc = make_float4(1, 1, 1, 1),
                                              float4 sh = make_float4(0, 0, 0, 0)) {
  // every tap is loaded unconditionally from a clamped (always valid) address and masked afterwards: a guarded load is a branch of its
  // own with a full wait behind it - nine dependent round trips per window instead of one (the pooled-operand nodes spent 14 us of a 22 us
  // block there, tools/dev/node_fwd_phases.py)
  float4 v[9];
  unsigned in = 0u;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int y = oh * 2 - pad_t + i;
    const int yc = min(max(y, 0), PH - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = ow * 2 - pad_l + j;
      const int xc = min(max(x, 0), PW - 1);
      if (y >= 0 && y < PH && x >= 0 && x < PW) in |= 1u << (i * 3 + j);
      v[i * 3 + j] = mmd_ld4(src + (((size_t)b * PH + yc) * PW + xc) * C + c);
    }
  }
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float4 u = v[t];
    if (aff) u = fuse_aff4(u, sc, sh);
    if (!((in >> t) & 1u)) u = make_float4(0, 0, 0, 0);   // zero padding takes part in the max
    m.x = fmaxf(m.x, u.x); m.y = fmaxf(m.y, u.y); m.z = fmaxf(m.z, u.z); m.w = fmaxf(m.w, u.w);
  }
  return m;
}

// The same window with the arg-max per channel: arg[q] = tap index (3*i + j) of the FIRST maximum in row-major scan order (torch's max-pool
// backward picks that one), -1 when a zero-padding element wins (it swallows the gradient).
__device__ __forceinline__ void pool_window_arg(const float* __restrict__ src, int b, int oh, int ow, int c, int PH, int PW,
                                                int C, int pad_t, int pad_l, int (&arg)[4], bool aff = false,
                                                float4 sc = make_float4(1, 1, 1, 1), float4 sh = make_float4(0, 0, 0, 0)) {
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int q = 0; q < 4; ++q) arg[q] = -1;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int y = oh * 2 - pad_t + i;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = ow * 2 - pad_l + j;
      const bool in = y >= 0 && y < PH && x >= 0 && x < PW;
      float4 v = make_float4(0, 0, 0, 0);
      if (in) { v = mmd_ld4(src + (((size_t)b * PH + y) * PW + x) * C + c); if (aff) v = fuse_aff4(v, sc, sh); }
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (vv[q] > best[q]) { best[q] = vv[q]; arg[q] = in ? i * 3 + j : -1; }
    }
  }
}

// MODE: -1 = operand set read from the arguments at run time; otherwise bit 0 = in1, bit 1 = up, bit 2 = pool present (compile time: the node
// backward kernel is instantiated per operand set - the run-time form keeps every operand's registers and branches alive)
template <int MODE = -1>
__device__ __forceinline__ float4 fuse_presum(const FuseArgs& a, const float* w, int b, int h, int x, int c, float4* o0,
                                              float4* o1, float4* o2, float4* o3, const FuseCoef* fc = nullptr) {
  const bool has1 = MODE < 0 ? a.in1 != nullptr : (MODE & 1) != 0;
  const bool hasu = MODE < 0 ? a.up != nullptr : (MODE & 2) != 0;
  const bool hasp = MODE < 0 ? a.pl != nullptr : (MODE & 4) != 0;
  size_t off = (((size_t)b * a.H + h) * a.W + x) * a.C + c;
  int wi = 0;
  float4 s = make_float4(0, 0, 0, 0);
  float4 v = mmd_ld4(a.in0 + off);
  if (fc) v = fuse_aff4(v, fc->sc(0), fc->sh(0));
  *o0 = v;
  s.x += w[wi] * v.x; s.y += w[wi] * v.y; s.z += w[wi] * v.z; s.w += w[wi] * v.w; ++wi;
  if (has1) {
    v = mmd_ld4(a.in1 + off);
    if (fc) v = fuse_aff4(v, fc->sc(1), fc->sh(1));
    *o1 = v; s.x += w[wi] * v.x; s.y += w[wi] * v.y; s.z += w[wi] * v.z; s.w += w[wi] * v.w; ++wi;
  }
  if (hasu) {
    v = mmd_ld4(a.up + (((size_t)b * (a.H >> 1) + (h >> 1)) * (a.W >> 1) + (x >> 1)) * a.C + c);
    if (fc) v = fuse_aff4(v, fc->sc(2), fc->sh(2));
    *o2 = v;
    s.x += w[wi] * v.x; s.y += w[wi] * v.y; s.z += w[wi] * v.z; s.w += w[wi] * v.w; ++wi;
  }
  if (hasp) {
    v = fc ? pool_window(a.pl, b, h, x, c, a.PH, a.PW, a.C, a.pad_t, a.pad_l, (a.lazy & 8) != 0, fc->sc(3), fc->sh(3))
           : pool_window(a.pl, b, h, x, c, a.PH, a.PW, a.C, a.pad_t, a.pad_l);
    *o3 = v;
    s.x += w[wi] * v.x; s.y += w[wi] * v.y; s.z += w[wi] * v.z; s.w += w[wi] * v.w; ++wi;
  }
  return s;
}

__global__ __launch_bounds__(256) void fuse_fwd_kernel(FuseArgs a, float* __restrict__ out) {
  float w[3];
  fuse_weights(a.theta, a.ntheta, w);
  const int c4n = a.C >> 2;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)a.B * a.H * a.W * c4n;
  if (idx >= total) return;
  int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
  int x = (int)(pix % a.W); pix /= a.W;
  int h = (int)(pix % a.H); int b = (int)(pix / a.H);
  float4 t0, t1, t2, t3;
  float4 s = fuse_presum(a, w, b, h, x, c, &t0, &t1, &t2, &t3);
  s.x = mmd_swish(s.x); s.y = mmd_swish(s.y); s.z = mmd_swish(s.z); s.w = mmd_swish(s.w);
  mmd_st4(out + (((size_t)b * a.H + h) * a.W + x) * a.C + c, s);
}

// Fusion node + its depthwise 3x3 in ONE kernel: the fused activation f = swish(sum w_i * operand_i) is computed while
// the 10x10x64 input tile (8x8 outputs + halo) is staged in LDS (halo pixels are recomputed by the neighbouring blocks),
// optionally written out for the backward (interior pixels only), and the 3x3 taps run from LDS.
// Replaces fuse_fwd_kernel + dw_fwd_kernel<3,1> (one launch and one read of f less per BiFPN node).
__global__ __launch_bounds__(256) void fuse_dw_fwd_kernel(FuseArgs a, const float* __restrict__ wdw, float* __restrict__ f_out,
                                                         float* __restrict__ zd, int tiles_h, int tiles_w, int cchunks) {
  constexpr int TH = 8, TW = 8, IH = 10, IW = 10, R = 4, SEG = 6;
  __shared__ float sIn[IH * IW * 64];
  __shared__ float sW[9 * 64];
  const int tid = threadIdx.x;
  // (XCD-aware order, round 6: blocks b, b + 8, ... share an XCD and its L2 - a contiguous eighth of the tiles per XCD keeps the neighbours
  // whose halos overlap, and the channel chunks that restage one dz tile, behind one L2)
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int cc = bid % cchunks; bid /= cchunks;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int b = bid;
  if (a.g_images) {      // grouped frozen nets: this image's net (fusion weights and depthwise taps live in the flat parameter buffer)
    const size_t gw = (size_t)(b / a.g_images) * a.g_w;
    a.theta += gw; wdw += gw;
  }
  float w[3];
  fuse_weights(a.theta, a.ntheta, w);
  const int c0 = cc * 64, c4 = (tid & 15) * 4, c = c0 + c4;
  const bool cok = c < a.C;
  const int oh0 = th * TH, ow0 = tw * TW;
  // staging: every operand load of the thread's seven tile pixels is issued before the first use, unconditionally from clamped coordinates
  // (masked afterwards) - as a guarded load -> fuse -> LDS-store loop each pixel was a dependent round trip of its own (two with a pooled
  // operand); the pooled operand's 3x3 windows follow three pixels at a time
  constexpr int NST = (IH * IW + 15) / 16;
  const bool has1 = a.in1 != nullptr, hasu = a.up != nullptr, hasp = a.pl != nullptr;      // block-uniform
  const int cs = cok ? c : 0;
  const int ti = min(tid, 9 * 16 - 1), ttap = ti >> 4, tq = (ti & 15) * 4;
  const bool tok = c0 + tq < a.C;
  const float4 twv = mmd_ld4(wdw + (size_t)ttap * a.C + (tok ? c0 + tq : 0));
  float4 v0[NST], v1[NST];
  unsigned okm = 0u;
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int p = (tid >> 4) + 16 * i;
    const int ih = oh0 - 1 + p / IW, iw = ow0 - 1 + p % IW;
    if (cok && p < IH * IW && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) okm |= 1u << i;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
    const size_t off = (((size_t)b * a.H + ihc) * a.W + iwc) * a.C + cs;
    v0[i] = mmd_ld4(a.in0 + off);
    v1[i] = make_float4(0, 0, 0, 0);
    if (has1) v1[i] = mmd_ld4(a.in1 + off);
    else if (hasu) v1[i] = mmd_ld4(a.up + (((size_t)b * (a.H >> 1) + (ihc >> 1)) * (a.W >> 1) + (iwc >> 1)) * a.C + cs);
  }
  if (tid < 9 * 16) *reinterpret_cast<float4*>(&sW[ttap * 64 + tq]) = tok ? twv : make_float4(0, 0, 0, 0);
  const float w1 = (has1 || hasu) ? w[1] : 0.f, wp = hasp ? w[(has1 || hasu) ? 2 : 1] : 0.f;
#pragma unroll
  for (int i0 = 0; i0 < NST; i0 += 3) {
    float4 m[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = i0 + u;
      m[u] = make_float4(0, 0, 0, 0);
      if (i < NST && hasp) {
        const int p = (tid >> 4) + 16 * i;
        const int ihc = min(max(oh0 - 1 + p / IW, 0), a.H - 1), iwc = min(max(ow0 - 1 + p % IW, 0), a.W - 1);
        m[u] = pool_window(a.pl, b, ihc, iwc, cs, a.PH, a.PW, a.C, a.pad_t, a.pad_l);
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = i0 + u;
      if (i >= NST) break;
      const int p = (tid >> 4) + 16 * i;
      if (p >= IH * IW) break;
      // (the summation order and contraction of fuse_presum: the two-launch path's f is bit-identical)
      auto mix = [&](float x0, float x1, float xm) {
        float t = w[0] * x0;
        if (has1 || hasu) t = __fmaf_rn(w1, x1, t);
        if (hasp) t = __fmaf_rn(wp, xm, t);
        return t;
      };
      float4 v;
      v.x = mix(v0[i].x, v1[i].x, m[u].x); v.y = mix(v0[i].y, v1[i].y, m[u].y);
      v.z = mix(v0[i].z, v1[i].z, m[u].z); v.w = mix(v0[i].w, v1[i].w, m[u].w);
      const bool ok = (okm >> i) & 1u;
      v.x = ok ? mmd_swish(v.x) : 0.f; v.y = ok ? mmd_swish(v.y) : 0.f; v.z = ok ? mmd_swish(v.z) : 0.f; v.w = ok ? mmd_swish(v.w) : 0.f;
      const int py = p / IW, px = p % IW;
      if (ok && f_out && py >= 1 && py <= TH && px >= 1 && px <= TW)
        mmd_st4(f_out + (((size_t)b * a.H + oh0 - 1 + py) * a.W + ow0 - 1 + px) * a.C + c, v);
      *reinterpret_cast<float4*>(&sIn[p * 64 + c4]) = v;
    }
  }
  __syncthreads();
  const int p = tid >> 4;
  const int orow = p / (TW / R);
  const int ocol0 = (p % (TW / R)) * R;
  float4 acc[R];
#pragma unroll
  for (int o = 0; o < R; ++o) acc[o] = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float4 in[SEG];
    const float* prow = &sIn[((orow + i) * IW + ocol0) * 64 + c4];
#pragma unroll
    for (int q = 0; q < SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * 64);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4 wv = *reinterpret_cast<const float4*>(&sW[(i * 3 + j) * 64 + c4]);
#pragma unroll
      for (int o = 0; o < R; ++o) {
        acc[o].x += in[o + j].x * wv.x; acc[o].y += in[o + j].y * wv.y;
        acc[o].z += in[o + j].z * wv.z; acc[o].w += in[o + j].w * wv.w;
      }
    }
  }
  const int oh = oh0 + orow;
#pragma unroll
  for (int o = 0; o < R; ++o) {
    int ow = ow0 + ocol0 + o;
    if (cok && oh < a.H && ow < a.W) mmd_st4(zd + (((size_t)b * a.H + oh) * a.W + ow) * a.C + c, acc[o]);
  }
}

static int fuse_fill(FuseArgs& a, const float* in0, const float* in1, const float* up, const float* pl,
                     const float* theta, int B, int H, int W, int C) {
  if (!in0 || !theta || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  int n = 1 + (in1 != nullptr) + (up != nullptr) + (pl != nullptr);
  if (n < 2 || n > 3) return MMD_EINVAL;
  if (up && ((H & 1) || (W & 1))) return MMD_EINVAL;
  a.in0 = in0; a.in1 = in1; a.up = up; a.pl = pl; a.theta = theta; a.ntheta = n;
  a.B = B; a.H = H; a.W = W; a.C = C; a.PH = 2 * H; a.PW = 2 * W;
  // SAME pool of an even-sized map: extra = 1 -> pad_lo = 0
  a.pad_t = 0; a.pad_l = 0;
  a.pl_lds_off = -1;
  return MMD_OK;
}

extern "C" int mmd_bifpn_fuse_fwd(const float* in0, const float* in1, const float* up, const float* pool,
                                  const float* theta, float* out, int B, int H, int W, int C, hipStream_t stream) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (rc || !out) return MMD_EINVAL;
  size_t total = (size_t)B * H * W * (C >> 2);
  hipLaunchKernelGGL(fuse_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a, out);
  return mmd_check_launch();
}

extern "C" int mmd_bifpn_node_dw_fwd(const float* in0, const float* in1, const float* up, const float* pool,
                                     const float* theta, const float* w_dw, float* f_out, float* zd, int B, int H, int W,
                                     int C, hipStream_t stream) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (rc || !w_dw || !zd) return MMD_EINVAL;
  int th = cdiv(H, 8), tw = cdiv(W, 8), cc = cdiv(C, 64);
  if (mmd_group_on()) {
    const MmdGroup& gr = mmd_group();
    if (B != gr.n * gr.images) return MMD_EINVAL;
    a.g_images = gr.images; a.g_w = gr.w_stride;
  }
  hipLaunchKernelGGL(fuse_dw_fwd_kernel, dim3((unsigned)(B * th * tw * cc)), dim3(256), 0, stream, a, w_dw, f_out, zd, th, tw, cc);
  return mmd_check_launch();
}

// -DMMD_NODE_TIMING (dev build, tools/dev/node_phases.py): block 0 / thread 0 stamps the 100 MHz wall clock at the phase boundaries
#ifdef MMD_NODE_TIMING
__device__ unsigned long long g_node_t[16];
#define NODE_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); g_node_t[i] = wall_clock64(); } } while (0)
extern "C" int mmd_node_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_node_t), sizeof(g_node_t)) == hipSuccess ? 0 : -1;
}
#else
#define NODE_T(i)
#endif
// ---- whole BiFPN node of a FROZEN net in one kernel ---------------------------------------------------------------------------
//   y = BN_folded( pw( dw3x3( swish( sum_i w_i * operand_i ) ) ) + bias )      SeparableConvBlock(norm=True, activation=False) after the
// fast-attention fusion (src/YetAnotherEfficientDet.py:150-185, 338-390), eval mode.  The two-kernel path writes the depthwise output,
// reads it back in a 1x1-conv GEMM launch and pays two launches on the net's serial chain per node (40 nodes per net).  Here a block of
// 8 waves owns an 8x8 pixel tile and all C = 112 channels:
//   phase 0  fused + activated input tile with halo (10x10 px) -> LDS            (VALU; 28 channel quads x 100 px over 512 threads)
//   phase 1  depthwise 3x3 from LDS -> zd tile [64 px][C] in LDS                  (VALU)
//   phase 2  zd · Wpwᵀ on v_mfma_f32_16x16x4_f32: A = zd rows (16 px), B = Wpw rows (16 output channels), k split so that lane group g owns
//            the contiguous run [28 g, 28 g + 28) of both; Wpw is fetched into registers before phase 0 and parked in the LDS region the
//            input tile no longer needs (B fragments straight from L2 instead - 78 KB, two blocks per CU - measured slower: 16.5 -> 18.2 us
//            per launch); 28 (row, column) tiles over 8 waves
//   epilogue (acc + bias) * scale + shift straight from the accumulators (16 lanes = 64 contiguous bytes of one pixel).
// 86 KB of LDS (one block per CU).
// Round 4: the width is a template parameter.  FC in {64, 112, 160, 224} (EfficientDet-D0 / D2 / D3 / D4; a multiple of 16).  Up to 160 the 1x1
// weights are parked in LDS as described; at FC = 224 (BASELINE configs[4]: 157 KB for the two tiles alone) they do not fit, and phase 2
// takes its B fragments straight from L2 (the [C, C] matrix is 200 KB: L2-resident after the first tile of a launch).
constexpr int FN_NT = 512;
template <int FC> struct FnCfg {
  // row stride of the two MFMA operand tiles: FC + 8.  A ds_read_b128 is banked per 16-lane group {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
  // (MI355X_MICROARCH.md, LDS table), i.e. rows r in {0-3, 12-15} of one k group with rows {4-11} of the next: with the fragments interleaved
  // (lane (r, g) reads k = 16 j + 4 g .. + 3) a stride = 8 (mod 16) floats puts the 16 lanes on 16 different 16-B slots for every width; FC + 4
  // (chosen for contiguous 16-lane groups) is 2-way on every fragment read
  static constexpr int Q = FC / 4, FS = FC, ZS = FC + 8, WS = FC + 8, KR = FC / 4, CT = FC / 16;
  static constexpr bool PARK = FC <= 160;
  static constexpr int U = (PARK && FC * WS > 100 * FS) ? FC * WS : 100 * FS;      // input tile, later (PARK) the 1x1 weights
  static constexpr bool LAZY_OK = FC <= 160;                                       // room for the lazy operands' coefficient table (train form)
  static constexpr size_t lds(bool train) { return (size_t)(U + 64 * ZS + 9 * FC + (train ? 2 * FC + (LAZY_OK ? 8 * FC : 0) : 0)) * sizeof(float); }
};

// TRAIN (the student's nodes): y = the RAW 1x1-conv output z (+ bias), its per-channel sums (sum z, sum z^2: the node's train-mode BatchNorm
// statistics) go to `stats`, and the depthwise output tile is also stored (zd_out: the 1x1 conv's weight gradient reads it in the backward).
// PK = false: the 1x1 weights are NOT parked in LDS even where they would fit (B fragments from L2 as at FC = 224): 78 KB instead of 86 KB, two
// blocks per CU - for the chip-filling launches (the 64^2 level of the teacher pack: 1536 blocks = six rounds of one block per CU)
template <int MODE, bool TRAIN = false, int FC = 112, bool PK = true>      // operand set: bit 0 = in1, bit 1 = up, bit 2 = pool (as fuse_presum)
__global__ __launch_bounds__(FN_NT) void bifpn_node_fused_kernel(FuseArgs a, const float* __restrict__ wdw, const float* __restrict__ wpw,
                                                                const float* __restrict__ bias, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ y,
                                                                int tiles_h, int tiles_w, float* __restrict__ zd_out, double* stats) {
  using Cf = FnCfg<FC>;
  constexpr bool PARK = Cf::PARK && PK;
  constexpr int FN_C = FC, FN_Q = Cf::Q, FN_FS = Cf::FS, FN_ZS = Cf::ZS, FN_WS = Cf::WS, FN_U = PARK ? Cf::U : 100 * Cf::FS, KR = Cf::KR, CT = Cf::CT;
  extern __shared__ float smem[];
  float* const sU = smem;                       // [100][FS] fused input tile | [C][WS] pointwise weights
  float* const sZ = smem + FN_U;                // [64][ZS] depthwise output tile
  float* const sWd = sZ + 64 * FN_ZS;           // [9][C] depthwise taps
  float* const sSt = sWd + 9 * FN_C;            // TRAIN: [2][C] block sums
  float* const sAf = sSt + 2 * FN_C;            // TRAIN, lazy operands: [4 operands][scale | shift][C]
  constexpr bool LZ = TRAIN && Cf::LAZY_OK;
  NODE_T(0);
  float w[3];
  fuse_weights(a.theta, a.ntheta, w);
  const int tid = threadIdx.x;
  // (XCD-aware order, round 6: blocks b, b + 8, ... share an XCD and its L2 - a contiguous eighth of the tiles per XCD keeps the neighbours
  // whose halos overlap, and the channel chunks that restage one dz tile, behind one L2)
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int b = bid, oh0 = th * 8, ow0 = tw * 8;
  if (!TRAIN && a.g_images) {      // grouped frozen nets: this image's net
    const size_t gw = (size_t)(b / a.g_images) * a.g_w, gb = (size_t)(b / a.g_images) * a.g_bn;
    wdw += gw; wpw += gw; if (bias) bias += gw; scale += gb; shift += gb;
    fuse_weights(a.theta + gw, a.ntheta, w);
  }
  // the 1x1 weights of the whole node, in flight while phases 0 and 1 run
  constexpr int NW4 = FN_C * FN_Q, WPT = PARK ? (NW4 + FN_NT - 1) / FN_NT : 1;
  float4 wreg[WPT];
  if constexpr (PARK) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * FN_NT;
      wreg[i] = idx < NW4 ? mmd_ld4(wpw + (size_t)idx * 4) : make_float4(0, 0, 0, 0);
    }
  }
  // depthwise taps and the epilogue coefficients of the wave's column tiles: loaded here, used (LDS store / epilogue) only after the
  // operand loads of phase 0 are in flight - a store or a guarded `bias ? bias[n] : 0` right here is a full wait in front of them
  static_assert(9 * FN_Q <= 2 * FN_NT, "two tap quads per thread");
  const float4 wd0 = mmd_ld4(wdw + (size_t)min(tid, 9 * FN_Q - 1) * 4), wd1 = mmd_ld4(wdw + (size_t)min(tid + FN_NT, 9 * FN_Q - 1) * 4);
  constexpr int NJ = (CT + 1) / 2;               // column tiles per wave
  float ebi[NJ], esc[NJ], esh[NJ];               // epilogue coefficients of the wave's column tiles (ct = (wave >> 2) + 2 j)
  {
    const int wv = tid >> 6, r_ = tid & 15;
    const float* const bsrc = bias ? bias : wdw;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = min(((wv >> 2) + 2 * j) * 16 + r_, FN_C - 1);
      const float t = bsrc[n];
      ebi[j] = bias ? t : 0.f;
      if (!TRAIN) { esc[j] = scale[n]; esh[j] = shift[n]; } else { esc[j] = 1.f; esh[j] = 0.f; }
    }
  }
  if (TRAIN) for (int i = tid; i < 2 * FN_C; i += FN_NT) sSt[i] = 0.f;
  // ---- phase 0: every global load of the thread's (pixel, quad) items is issued before the first use (one block per CU: nothing else
  // would hide six dependent load latencies); the pooled operand's 3x3 window is gathered in the second pass
  {
    constexpr int NI = (100 * FN_Q + FN_NT - 1) / FN_NT;
    float4 v0[NI], v1[NI];
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = tid + i * FN_NT;
      const int p = it / FN_Q, q = it - p * FN_Q;
      const int ih = oh0 - 1 + p / 10, iw = ow0 - 1 + p % 10;
      ok[i] = it < 100 * FN_Q && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      v0[i] = make_float4(0, 0, 0, 0); v1[i] = make_float4(0, 0, 0, 0);
      if (ok[i]) {
        const size_t off = (((size_t)b * a.H + ih) * a.W + iw) * a.C + q * 4;
        v0[i] = mmd_ld4(a.in0 + off);
        if (MODE & 1) v1[i] = mmd_ld4(a.in1 + off);
        else if (MODE & 2) v1[i] = mmd_ld4(a.up + (((size_t)b * (a.H >> 1) + (ih >> 1)) * (a.W >> 1) + (iw >> 1)) * a.C + q * 4);
      }
    }
    constexpr int n2 = (MODE & 3) ? 1 : 0;             // operand order (in0, in1, up, pool): in1 and up never occur together
    const float wp_ = (MODE & 4) ? w[1 + n2] : 0.f;
    const bool lz = LZ && a.lazy != 0;                 // block-uniform
    if (tid < 9 * FN_Q) *reinterpret_cast<float4*>(&sWd[tid * 4]) = wd0;
    if (tid + FN_NT < 9 * FN_Q) *reinterpret_cast<float4*>(&sWd[(tid + FN_NT) * 4]) = wd1;
    if constexpr (LZ) {
      if (lz) {
        // the lazy operands' coefficient table - filled HERE, behind the operand loads issued above, so that its own dependent loads
        // (batch sums -> coefficients) overlap them instead of preceding them on the block's critical path
        for (int i = tid; i < 4 * FN_C; i += FN_NT) {
          const int op = i / FN_C, c = i - op * FN_C;
          float sc = 1.f, sh = 0.f;
          if (a.lazy >> op & 1) {
            if (a.ost[op]) { BnLive bn; bn.stats = a.ost[op]; bn.gamma = a.oga[op]; bn.beta = a.obe[op]; bn.inv_count = a.oic[op]; bn.C = FN_C; bn.eps = 1e-3f;
                             bn_live_coef(bn, c, sc, sh); }
            else { sc = a.osc[op][c]; sh = a.osh[op][c]; }
          }
          sAf[(op * 2) * FN_C + c] = sc; sAf[(op * 2 + 1) * FN_C + c] = sh;
        }
        __syncthreads();
      }
    }
    constexpr int OP1 = (MODE & 1) ? 1 : 2;            // which operand v1 holds
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += 3) {               // the pooled operand's 3x3 windows of three items are gathered together (27 loads in flight)
      float4 m[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = i0 + u;
        m[u] = make_float4(0, 0, 0, 0);
        if (i < NI && (MODE & 4) && ok[i]) {
          const int it = tid + i * FN_NT;
          const int p = it / FN_Q, q = it - p * FN_Q;
          if (lz && (a.lazy & 8))
            m[u] = pool_window(a.pl, b, oh0 - 1 + p / 10, ow0 - 1 + p % 10, q * 4, a.PH, a.PW, a.C, a.pad_t, a.pad_l, true,
                               *reinterpret_cast<const float4*>(&sAf[6 * FN_C + q * 4]), *reinterpret_cast<const float4*>(&sAf[7 * FN_C + q * 4]));
          else
            m[u] = pool_window(a.pl, b, oh0 - 1 + p / 10, ow0 - 1 + p % 10, q * 4, a.PH, a.PW, a.C, a.pad_t, a.pad_l);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = i0 + u;
        if (i >= NI) break;
        const int it = tid + i * FN_NT;
        if (it >= 100 * FN_Q) break;
        const int p = it / FN_Q, q = it - p * FN_Q;
        float4 v = make_float4(0, 0, 0, 0);
        if (ok[i]) {
          if (lz) {
            v0[i] = fuse_aff4(v0[i], *reinterpret_cast<const float4*>(&sAf[q * 4]), *reinterpret_cast<const float4*>(&sAf[FN_C + q * 4]));
            if (n2) v1[i] = fuse_aff4(v1[i], *reinterpret_cast<const float4*>(&sAf[(2 * OP1) * FN_C + q * 4]),
                                      *reinterpret_cast<const float4*>(&sAf[(2 * OP1 + 1) * FN_C + q * 4]));
          }
          v.x = w[0] * v0[i].x; v.y = w[0] * v0[i].y; v.z = w[0] * v0[i].z; v.w = w[0] * v0[i].w;
          if (n2) { v.x += w[1] * v1[i].x; v.y += w[1] * v1[i].y; v.z += w[1] * v1[i].z; v.w += w[1] * v1[i].w; }
          if (MODE & 4) { v.x += wp_ * m[u].x; v.y += wp_ * m[u].y; v.z += wp_ * m[u].z; v.w += wp_ * m[u].w; }
          v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w);
        }
        *reinterpret_cast<float4*>(&sU[p * FN_FS + q * 4]) = v;
      }
    }
  }
  NODE_T(1);
  __syncthreads();
  NODE_T(2);
  // ---- phase 1
  // Thread -> (pixel, quad): with FN_Q = 28 quads per pixel a 16-lane LDS pass that starts at quad 16 / 20 / 24 runs on into the next
  // pixel's quads 0 .. 3 / 7 / 11, and those lanes read the SAME tap row of sWd at addresses that share banks with quads 16+ (4 q mod 64
  // banks wrap at q = 16): two-way conflicts on 3 of 7 passes of every tap read - the 31 - 44 % SQ_LDS_BANK_CONFLICT of every variant of
  // this kernel (profiles/r04_sq_counters_by_kernel.txt).  No tap-row layout avoids it (a 16-window-injective bank assignment over 28
  // cyclic quads would need period gcd(16, 28) = 4), so the lanes are dealt 32 per pixel instead (quads 28 .. 31 idle): every pass
  // stays inside one pixel.  Same trip count (64 x 32 / 512 = 4 = ceil(64 x 28 / 512)); widths where padding would add a trip keep the
  // dense mapping.
  constexpr int FN_QP = (FN_Q + 15) / 16 * 16;
  constexpr bool QPAD = (FN_QP != FN_Q) && ((64 * FN_QP + FN_NT - 1) / FN_NT == (64 * FN_Q + FN_NT - 1) / FN_NT);
  constexpr int FN_QM = QPAD ? FN_QP : FN_Q;
  for (int it = tid; it < 64 * FN_QM; it += FN_NT) {
    const int p = it / FN_QM, q = it - p * FN_QM;
    if (QPAD && q >= FN_Q) continue;
    const int orow = p >> 3, ocol = p & 7;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 x = mmd_lds_ld4(&sU[((orow + i) * 10 + ocol + j) * FN_FS + q * 4]);      // (one ds_read_b128 each: common.h)
        const float4 k = mmd_lds_ld4(&sWd[(i * 3 + j) * FN_C + q * 4]);
        acc.x += x.x * k.x; acc.y += x.y * k.y; acc.z += x.z * k.z; acc.w += x.w * k.w;
      }
    *reinterpret_cast<float4*>(&sZ[p * FN_ZS + q * 4]) = acc;
    if (TRAIN) {
      const int oh = oh0 + orow, ow = ow0 + ocol;
      if (oh < a.H && ow < a.W) mmd_st4(zd_out + (((size_t)b * a.H + oh) * a.W + ow) * FN_C + q * 4, acc);
    }
  }
  NODE_T(3);
  __syncthreads();                                   // every read of the input tile is done: park the 1x1 weights in its place
  NODE_T(4);
  if constexpr (PARK) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * FN_NT;
      if (idx < NW4) { const int n = idx / FN_Q, k4 = idx - n * FN_Q; *reinterpret_cast<float4*>(&sU[n * FN_WS + k4 * 4]) = wreg[i]; }
    }
    __syncthreads();
  }
  NODE_T(5);
  // ---- phase 2
  const int lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave & 3;
  float af[KR];
  {
    const float* ap = &sZ[(rt * 16 + r) * FN_ZS + g * 4];      // k = 16 j + 4 g + i: the same assignment on both operands (any is valid)
#pragma unroll
    for (int j = 0; j < KR / 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(ap + 16 * j);
      af[4 * j] = v.x; af[4 * j + 1] = v.y; af[4 * j + 2] = v.z; af[4 * j + 3] = v.w;
    }
  }
#pragma unroll
  for (int j4 = 0; j4 < NJ; ++j4) {
    const int ct = (wave >> 2) + 2 * j4;
    if (ct >= CT) break;                                 // wave-uniform
    float bf[KR];
    const float* bp = PARK ? &sU[(ct * 16 + r) * FN_WS + g * 4] : wpw + (size_t)(ct * 16 + r) * FN_C + g * 4;      // LDS-parked / straight from L2
#pragma unroll
    for (int j = 0; j < KR / 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(bp + 16 * j);
      bf[4 * j] = v.x; bf[4 * j + 1] = v.y; bf[4 * j + 2] = v.z; bf[4 * j + 3] = v.w;
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};      // two chains: a dependent MFMA waits ~8 passes for its accumulator
#pragma unroll
    for (int k = 0; k < KR; k += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bf[k], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k + 1], bf[k + 1], acc1, 0, 0, 0);
    }
    acc += acc1;
    const int n = ct * 16 + r;
    const float bi = ebi[j4], sc = esc[j4], sh = esh[j4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = rt * 16 + 4 * g + i;
      const int oh = oh0 + (p >> 3), ow = ow0 + (p & 7);
      if (oh < a.H && ow < a.W) {
        const float v = TRAIN ? acc[i] + bi : (acc[i] + bi) * sc + sh;
        y[(((size_t)b * a.H + oh) * a.W + ow) * FN_C + n] = v;
        if (TRAIN) { s1 += v; s2 += v * v; }
      }
    }
    if (TRAIN) {      // lanes r, r+16, r+32, r+48 hold the same channel (pixel groups g): fold them, then the four row-tile waves through LDS
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) { atomicAdd(&sSt[n], s1); atomicAdd(&sSt[FN_C + n], s2); }
    }
  }
  NODE_T(6);
  if (TRAIN) {
    __syncthreads();
    for (int i = tid; i < 2 * FN_C; i += FN_NT) atomicAdd(&stats[i], (double)sSt[i]);
  }
  NODE_T(7);
}

// 1 when mmd_bifpn_node_fwd_fused has a kernel for this width (the caller keeps mmd_bifpn_node_dw_fwd + mmd_pwconv_fwd otherwise)
extern "C" int mmd_bifpn_node_fused_supported(int C) { return (C == 64 || C == 112 || C == 160 || C == 224) ? 1 : 0; }

// Whole frozen-net BiFPN node: y[B*H*W, C] = ((dw3x3(swish(fuse(operands))) · w_pw[C,C]ᵀ) + bias) * scale + shift.
extern "C" int mmd_bifpn_node_fwd_fused(const float* in0, const float* in1, const float* up, const float* pool, const float* theta,
                                        const float* w_dw, const float* w_pw, const float* bias, const float* scale, const float* shift,
                                        float* y, int B, int H, int W, int C, hipStream_t stream) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (rc || !w_dw || !w_pw || !scale || !shift || !y || !mmd_bifpn_node_fused_supported(C) || (in1 && up)) return MMD_EINVAL;      // (no BiFPN node fuses in1 AND up)
  const int th = cdiv(H, 8), tw = cdiv(W, 8);
  if (mmd_group_on()) {
    const MmdGroup& gr = mmd_group();
    if (B != gr.n * gr.images) return MMD_EINVAL;
    a.g_images = gr.images; a.g_w = gr.w_stride; a.g_bn = gr.bn_stride;
  }
  mmd_prof_tag(MMD_FAM_MBX, "node H%lld C%lld ops%lld", H, C, a.ntheta, 0);
  mmd_prof_begin(MMD_FAM_MBX, stream);
  const int mode = (in1 ? 1 : 0) | (up ? 2 : 0) | (pool ? 4 : 0);
  const dim3 grid((unsigned)(B * th * tw)), blk(FN_NT);
  rc = MMD_EINVAL;
#define MMD_NODE_FWD(M, FC_) do { static bool attr = false; \
    if (!attr) { hipFuncSetAttribute((const void*)bifpn_node_fused_kernel<M, false, FC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((bifpn_node_fused_kernel<M, false, FC_>), grid, blk, FnCfg<FC_>::lds(false), stream, a, w_dw, w_pw, bias, scale, shift, y, th, tw, nullptr, nullptr); \
    rc = MMD_OK; } while (0)
#define MMD_NODE_FWD_NP(M, FC_) do { static bool attr = false; \
    if (!attr) { hipFuncSetAttribute((const void*)bifpn_node_fused_kernel<M, false, FC_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((bifpn_node_fused_kernel<M, false, FC_, false>), grid, blk, (size_t)(100 * FnCfg<FC_>::FS + 64 * FnCfg<FC_>::ZS + 9 * FC_) * sizeof(float), stream, a, w_dw, w_pw, bias, scale, shift, y, th, tw, nullptr, nullptr); \
    rc = MMD_OK; } while (0)
  static const int nopark_min = getenv("MMD_NODE_NOPARK_MIN") ? atoi(getenv("MMD_NODE_NOPARK_MIN")) : 1024;
#define MMD_NODE_FWD_C(FC_) do { \
    if (mode == 2 && FC_ == 112 && (int)grid.x >= nopark_min) MMD_NODE_FWD_NP(2, 112); \
    else if (mode == 2) MMD_NODE_FWD(2, FC_);            /* (in, up): top-down nodes */ \
    else if (mode == 5) MMD_NODE_FWD(5, FC_);       /* (in, td, pool): bottom-up nodes */ \
    else if (mode == 4) MMD_NODE_FWD(4, FC_);       /* (in, pool): p7_out */ \
    else if (mode == 1) MMD_NODE_FWD(1, FC_); } while (0)
  if (C == 112) MMD_NODE_FWD_C(112);
  else if (C == 224) MMD_NODE_FWD_C(224);
  else if (C == 160) MMD_NODE_FWD_C(160);
  else MMD_NODE_FWD_C(64);
#undef MMD_NODE_FWD_C
#undef MMD_NODE_FWD_NP
#undef MMD_NODE_FWD
  if (rc) return rc;
  const double rows = (double)B * H * W;
  mmd_prof_end(MMD_FAM_MBX, stream, rows * C * (2.0 * 9 + 2.0 * C), 4.0 * rows * C * (a.ntheta + 1 + (pool ? 3 : 0)));
  return mmd_check_launch();
}

// Whole TRAINABLE-net BiFPN node forward (train mode): z[B*H*W, C] = dw3x3(swish(fuse(operands))) · w_pw[C,C]ᵀ + bias (raw, pre-BatchNorm),
// stats[2C] (+)= [sum z, sum z^2] (the BatchNorm's batch statistics), zd[B*H*W, C] = the depthwise output (kept for the backward).  One
// launch instead of mmd_bifpn_node_dw_fwd + mmd_pwconv_fwd(stats) on the student's forward chain.
// host arrays [4] in operand order (in0, in1, up, pool); a null entry = that operand is a plain tensor
static int fuse_fill_lazy_fwd(FuseArgs& a, const void* op_stats4, const void* op_gamma4, const void* op_beta4, const void* op_count4) {
  if (!op_stats4) return MMD_OK;
  if (!op_gamma4 || !op_beta4 || !op_count4) return MMD_EINVAL;
  const double* const* st = (const double* const*)op_stats4;
  const float* const* ga = (const float* const*)op_gamma4;
  const float* const* be = (const float* const*)op_beta4;
  const long long* cn = (const long long*)op_count4;
  const float* ops[4] = {a.in0, a.in1, a.up, a.pl};
  for (int i = 0; i < 4; ++i) {
    if (!st[i]) continue;
    if (!ops[i] || !ga[i] || !be[i] || cn[i] <= 0) return MMD_EINVAL;
    a.ost[i] = st[i]; a.oga[i] = ga[i]; a.obe[i] = be[i]; a.oic[i] = 1.0 / (double)cn[i]; a.lazy |= 1 << i;
  }
  return MMD_OK;
}
static int fuse_fill_lazy_bwd(FuseArgs& a, const void* op_scale4, const void* op_shift4) {
  if (!op_scale4) return MMD_OK;
  if (!op_shift4) return MMD_EINVAL;
  const float* const* sc = (const float* const*)op_scale4;
  const float* const* sh = (const float* const*)op_shift4;
  const float* ops[4] = {a.in0, a.in1, a.up, a.pl};
  for (int i = 0; i < 4; ++i) {
    if (!sc[i]) continue;
    if (!ops[i] || !sh[i]) return MMD_EINVAL;
    a.osc[i] = sc[i]; a.osh[i] = sh[i]; a.lazy |= 1 << i;
  }
  return MMD_OK;
}
static int node_fwd_fused_train_impl(const float* in0, const float* in1, const float* up, const float* pool, const float* theta,
                                     const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats,
                                     int B, int H, int W, int C, const void* op_stats4, const void* op_gamma4, const void* op_beta4,
                                     const void* op_count4, hipStream_t stream);
extern "C" int mmd_bifpn_node_fwd_fused_train(const float* in0, const float* in1, const float* up, const float* pool, const float* theta,
                                              const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats,
                                              int B, int H, int W, int C, hipStream_t stream) {
  return node_fwd_fused_train_impl(in0, in1, up, pool, theta, w_dw, w_pw, bias, z, zd, stats, B, H, W, C, nullptr, nullptr, nullptr, nullptr, stream);
}
// Round 4, "lazy" operands: operand i is the RAW 1x1-conv output of its producer and its train-mode BatchNorm (no activation) is applied
// while it is loaded, from the live batch sums op_stats4[i] (+ op_gamma4[i], op_beta4[i], op_count4[i] rows) - host arrays of 4 entries in
// operand order (in0, in1, up, pool), null entry = plain tensor.  The producer node then needs no mmd_affine_act launch.  C <= 160.
extern "C" int mmd_bifpn_node_fwd_fused_train_lz(const float* in0, const float* in1, const float* up, const float* pool, const float* theta,
                                                 const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats,
                                                 int B, int H, int W, int C, const void* op_stats4, const void* op_gamma4,
                                                 const void* op_beta4, const void* op_count4, hipStream_t stream) {
  return node_fwd_fused_train_impl(in0, in1, up, pool, theta, w_dw, w_pw, bias, z, zd, stats, B, H, W, C, op_stats4, op_gamma4, op_beta4, op_count4, stream);
}
static int node_fwd_fused_train_impl(const float* in0, const float* in1, const float* up, const float* pool, const float* theta,
                                     const float* w_dw, const float* w_pw, const float* bias, float* z, float* zd, double* stats,
                                     int B, int H, int W, int C, const void* op_stats4, const void* op_gamma4, const void* op_beta4,
                                     const void* op_count4, hipStream_t stream) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (!rc) rc = fuse_fill_lazy_fwd(a, op_stats4, op_gamma4, op_beta4, op_count4);
  if (a.lazy && C > 160) return MMD_EINVAL;
  if (rc || !w_dw || !w_pw || !z || !zd || !stats || !mmd_bifpn_node_fused_supported(C) || (in1 && up)) return MMD_EINVAL;
  const int th = cdiv(H, 8), tw = cdiv(W, 8);
  const int mode = (in1 ? 1 : 0) | (up ? 2 : 0) | (pool ? 4 : 0);
  const dim3 grid((unsigned)(B * th * tw)), blk(FN_NT);
  rc = MMD_EINVAL;
#define MMD_NODE_FWD_T(M, FC_) do { static bool attr = false; \
    if (!attr) { hipFuncSetAttribute((const void*)bifpn_node_fused_kernel<M, true, FC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((bifpn_node_fused_kernel<M, true, FC_>), grid, blk, FnCfg<FC_>::lds(true), stream, a, w_dw, w_pw, bias, nullptr, nullptr, z, th, tw, zd, stats); \
    rc = MMD_OK; } while (0)
#define MMD_NODE_FWD_TC(FC_) do { \
    if (mode == 2) MMD_NODE_FWD_T(2, FC_); else if (mode == 5) MMD_NODE_FWD_T(5, FC_); else if (mode == 4) MMD_NODE_FWD_T(4, FC_); \
    else if (mode == 1) MMD_NODE_FWD_T(1, FC_); } while (0)
  if (C == 112) MMD_NODE_FWD_TC(112);
  else if (C == 224) MMD_NODE_FWD_TC(224);
  else if (C == 160) MMD_NODE_FWD_TC(160);
  else MMD_NODE_FWD_TC(64);
#undef MMD_NODE_FWD_TC
#undef MMD_NODE_FWD_T
  if (rc) return rc;
  return mmd_check_launch();
}

// backward part 1: dx = df * swish'(x) (x recomputed from the operands); wdot[i] += <dx, operand_i>
// The gradients of the same-resolution operands (in0, in1) are w_i * dx: written / accumulated here (d0, d1 nullable) instead
// of by a scale_acc launch each; dx itself is stored only when an upsampled or pooled operand still needs it.
__global__ __launch_bounds__(256) void fuse_bwd_kernel(FuseArgs a, const float* __restrict__ df, float* __restrict__ dx,
                                                       float* wdot, float* __restrict__ d0, int acc0,
                                                       float* __restrict__ d1, int acc1) {
  __shared__ float sred[4 * 3];
  float w[3];
  fuse_weights(a.theta, a.ntheta, w);
  const int c4n = a.C >> 2;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)a.B * a.H * a.W * c4n;
  float d[3] = {0.f, 0.f, 0.f};
  if (idx < total) {
    int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
    int x = (int)(pix % a.W); pix /= a.W;
    int h = (int)(pix % a.H); int b = (int)(pix / a.H);
    float4 t[4];
    float4 s = fuse_presum(a, w, b, h, x, c, &t[0], &t[1], &t[2], &t[3]);
    size_t off = (((size_t)b * a.H + h) * a.W + x) * a.C + c;
    float4 g = mmd_ld4(df + off);
    g.x *= mmd_swish_grad(s.x); g.y *= mmd_swish_grad(s.y); g.z *= mmd_swish_grad(s.z); g.w *= mmd_swish_grad(s.w);
    if (dx) mmd_st4(dx + off, g);
    if (d0) {
      float4 v = make_float4(g.x * w[0], g.y * w[0], g.z * w[0], g.w * w[0]);
      if (acc0) { float4 p = mmd_ld4(d0 + off); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
      mmd_st4(d0 + off, v);
    }
    if (d1) {      // in1 present: its weight is w[1]
      float4 v = make_float4(g.x * w[1], g.y * w[1], g.z * w[1], g.w * w[1]);
      if (acc1) { float4 p = mmd_ld4(d1 + off); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
      mmd_st4(d1 + off, v);
    }
    int wi = 0;
    d[wi++] = g.x * t[0].x + g.y * t[0].y + g.z * t[0].z + g.w * t[0].w;
    if (a.in1) d[wi++] = g.x * t[1].x + g.y * t[1].y + g.z * t[1].z + g.w * t[1].w;
    if (a.up) d[wi++] = g.x * t[2].x + g.y * t[2].y + g.z * t[2].z + g.w * t[2].w;
    if (a.pl) d[wi++] = g.x * t[3].x + g.y * t[3].y + g.z * t[3].z + g.w * t[3].w;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 3; ++i) { float v = wave_sum(d[i]); if (lane == 0) sred[wave * 3 + i] = v; }
  __syncthreads();
  if (threadIdx.x < a.ntheta)
    atomicAdd(&wdot[threadIdx.x], sred[threadIdx.x] + sred[3 + threadIdx.x] + sred[6 + threadIdx.x] + sred[9 + threadIdx.x]);
}
extern "C" int mmd_bifpn_fuse_bwd(const float* in0, const float* in1, const float* up, const float* pool,
                                  const float* theta, const float* df, float* dx, float* wdot, int B, int H, int W,
                                  int C, float* d0, int acc0, float* d1, int acc1, hipStream_t stream) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (rc || !df || !wdot || (!dx && !d0) || (d1 && !in1)) return MMD_EINVAL;
  size_t total = (size_t)B * H * W * (C >> 2);
  hipLaunchKernelGGL(fuse_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a, df, dx, wdot, d0, acc0, d1, acc1);
  return mmd_check_launch();
}

// Fused node backward: the depthwise 3x3 input gradient  df = dwconv^T(dzd, w)  is computed from an LDS tile of dzd (8x8
// pixels + halo, 64-channel chunk) and fed straight into the fusion backward above - one launch instead of two and no df
// round trip through HBM (mirror of fuse_dw_fwd_kernel; 40 launches on the backward's serial chain).
// GEMM = true (round 4, "whole-node backward"): the launch also runs the node's 1x1 conv's input gradient, which used to be a launch of its
// own in front of this one (mmd_pwconv_bwd_data_bn on the skinny GEMM kernel: 40 launches of 8-25 us on the backward's serial chain).  The
// block evaluates the node's BatchNorm backward dz = a1 g + a2 (z - mu) + a3 on its 10x10-pixel halo tile for ALL channels into LDS
// ([112][C + 8], dynamic), multiplies it by W[:, chunk] on v_mfma_f32_16x16x4_f32 (wave w owns 16 channels of the block's 64-channel chunk,
// all 7 pixel tiles; its 28 B values per lane come straight from L2 - the transposed weight, 7 float4 loads - and stay in registers) and writes the dzd tile into sIn - where the
// plain form stages it from HBM.  The halo is recomputed by the neighbouring blocks (1.56x of a 2.5 MFLOP product); dzd never exists in
// HBM.  The chunk-0 blocks store dz (interior pixels) for the conv's weight-gradient GEMM; block 0 adds dgamma / dbeta.
struct NodeGemm {
  const float* g; const float* z; const float* scale; const float* mean; const float* invstd; const double* sums; double inv_count;
  const float* w;          // the 1x1 conv's weight TRANSPOSED, w_t[C in][C out] (what mmd_pwconv_bwd_data takes): dzd[p, c] = sum_n dz[p, n] w_t[c, n]
  float* dz_out; float* dgamma; float* dbeta;
};
// NKK (GEMM form): the width in 16-channel groups as a compile-time constant (7 = D2's 112, 14 = D4's 224; 0 = read from the arguments) - the MFMA
// loop is then straight-line code: with the run-time bound every k group sat behind its own branch, and the LDS reads of the next group
// could not be hoisted over it
// CW (round 5): channels per block.  64 = the form described above.  16 (GEMM form only) = the SMALL-MAP form: on the 4^2 .. 16^2 levels the
// 64-channel blocks are 16 .. 64 per launch, a quarter of the chip at best, and a launch lasts as long as one block's dependent chain
// (19 - 31 us, tools/dev/node_phases.py).  With 16-channel blocks there are 3.5x as many; a block stages the same dz tile, but its four
// waves share the ONE 16-channel column group of the 1x1 product (wave w takes pixel tiles w and w + 4: 56 MFMAs instead of 196 on the
// block's critical path) and a thread owns one pixel x one channel quad of the operand pass instead of four (one round of loads, a quarter
// of the scattered atomics of a pooled operand).
template <int MODE, bool GEMM = false, int NKK = 0, int CW = 64>
__global__ __launch_bounds__(256) void fuse_dw_bwd_kernel(FuseArgs a, const float* __restrict__ wdw,
                                                         const float* __restrict__ dzd, float* __restrict__ dx, float* wdot,
                                                         float* __restrict__ d0, int acc0, float* __restrict__ d1, int acc1,
                                                         float* __restrict__ dup, int acc_up, float* __restrict__ dwg,
                                                         int tiles_h, int tiles_w, int cchunks, BnSumDst x0, BnSumDst x1, BnSumDst xu,
                                                         float* dpl, BnSumDst xp, int own, NodeGemm ng) {
  // dpl (MODE & 4): the POOLED operand's gradient [B, 2H, 2W, C], scattered from here - every output pixel adds w_pool * g to the arg-max
  // element of its 3x3 window with fp32 atomics (dpl holds zeros or the earlier contributions) - instead of materialising dx for a gather
  // launch over the 4x larger source map (mmd_maxpool_same_bwd_acc: 22 launches of ~22 us per D2 step).  A scattered gradient has no last
  // writer, so the BatchNorm-backward sums of such a tensor are kept LINEARLY: every contribution adds the sums of its own share -
  // own bit 0 / 1 / 2: the sums of d0 / d1 / dup are taken over this launch's share, not over the accumulated total; xp: this scatter's share.
  static_assert(CW == 64 || (CW == 16 && GEMM), "channel chunk");
  constexpr int TH = 8, TW = 8, IH = 10, IW = 10, R = CW == 64 ? 4 : 1, SEG = R + 2;
  constexpr int QL = CW / 4;                        // lanes (channel quads) per pixel
  constexpr int PR = R >= 2 ? 2 : 1;                // pixels of a thread whose loads are in flight together
  __shared__ float sIn[IH * IW * CW];
  __shared__ float sW[9 * CW];
  __shared__ float sred[4 * 3];
  // BatchNorm-backward sums of the operand gradients this launch completes: [operand][s|q][wave][channel].  GEMM form: they live in the dz
  // tile's place (dead behind the MFMA phase) - with the 100-row tile that brings the block to 78 KB, two blocks per CU on the large maps
  extern __shared__ float sDyn[];
  __shared__ float sBnS[GEMM ? 4 : 4 * 2 * 4 * 64];
  float* const sBn = GEMM ? sDyn : sBnS;
  __shared__ float sCo[8 * 64];                    // lazy operands' (scale, shift) of this block's channel chunk
  static_assert(IH * IW >= 4 * 9, "the weight-gradient reduction aliases the dzd tile");
  NODE_T(0);
  const int tid = threadIdx.x;
  // (XCD-aware order, round 6: blocks b, b + 8, ... share an XCD and its L2 - a contiguous eighth of the tiles per XCD keeps the neighbours
  // whose halos overlap, and the channel chunks that restage one dz tile, behind one L2)
  int bid = mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int cc = bid % cchunks; bid /= cchunks;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int b = bid;
  const int c0 = cc * CW, c4 = (tid & (QL - 1)) * 4, c = c0 + c4;
  const bool cok = c < a.C;
  const int oh0 = th * TH, ow0 = tw * TW;
  // ---- head: everything the block needs before its first barrier (lazy operands' coefficients, flipped taps, the BatchNorm backward's
  // coefficient sources) is LOADED here and stored to LDS behind the loads of the 1x1 weights and of the dz tile's first trip: as four
  // load -> LDS-store groups in a row they were four dependent round trips, 4.5 of a block's 19 us (profiles/r04_node_bwd_phases_after.txt)
  const int hop = __builtin_amdgcn_readfirstlane(tid >> 6), hcl = tid & 63;      // lazy table: this thread's (operand, channel)
  const bool hlz = (a.lazy >> hop) & 1, hlive = hlz && a.ost[hop];      // (live batch statistics: the forward form; the backward passes finalized coefficients)
  const int hlc = min(c0 + hcl, a.C - 1);
  // unconditional loads (stand-in address when the operand carries no finalized coefficients): a value loaded inside a branch is waited for at the join
  float hsc = ((hlz && !hlive) ? a.osc[hop] : wdw)[(hlz && !hlive) ? hlc : 0], hsh = ((hlz && !hlive) ? a.osh[hop] : wdw)[(hlz && !hlive) ? hlc : 0];
  const int ti = min(tid, 9 * QL - 1), ttap = ti / QL, tq = (ti % QL) * 4;      // flipped taps: the transpose of a stride-1 SAME correlation
  const bool tok = c0 + tq < a.C;
  const float4 twv = mmd_ld4(wdw + (size_t)(8 - ttap) * a.C + (tok ? c0 + tq : 0));
  // Pooled operand's gradient (MODE & 4), round 5.  Every output pixel adds w_pool * g to the arg-max element of its 3 x 3 window: as
  // 16 scattered 4-byte GLOBAL atomics per thread and pixel pair that was 33 of a block's 63 us (tools/dev/node_phases.py at H 48 / C 224:
  // "consume" 18.7 + 14.1 us, against 2 us for a node without a pooled operand).  The block's 8 x 8 output tile touches a 17 x 17 region of
  // the fine map (windows 2 oh .. 2 oh + 2): the contributions are collected in an LDS tile [17 x 17][64] with LDS atomics and written out
  // once - interior elements (rows / columns 1 .. 15: no other block's windows reach them) by plain coalesced read-add-write, the shared rim
  // (rows / columns 0 and 16) with global atomics.  74 KB more LDS: these variants hold 256 + 34 .. 76 registers, one block per CU anyway.
  // The tile's INTERIOR starts from dpl's current values, fetched here by LDS-DMA (global_load_lds_dwordx4: no registers, in flight behind
  // everything up to the operand pass), so that the write-out is a plain store - as a read-add-write it was three dependent round trips at
  // the end of the block (6.8 of the 22.7 us operand pass at H 32 / C 112); the rim and out-of-range slots start from zero.
  // (GEMM form only: in the two-launch form - config 5's large levels - the DMA loop's index arithmetic pushed the pooled variant, 256 + 70
  // registers, into 17 spills and cost what it saved: that form zero-fills the tile and writes out by read-add-write, FINE_DMA = false)
  constexpr int FT = 17;
  constexpr bool FINE_DMA = GEMM;
  float* const sFine = ((MODE & 4) && CW == 64 && a.pl_lds_off >= 0) ? sDyn + a.pl_lds_off : nullptr;      // (64-channel form only)
  if ((MODE & 4) && sFine && !FINE_DMA)
    for (int i = tid; i < FT * FT * 16; i += 256) *reinterpret_cast<float4*>(&sFine[i * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((MODE & 4) && sFine && FINE_DMA) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int wv0 = __builtin_amdgcn_readfirstlane(tid >> 6), ln0 = tid & 63;
    for (int it = wv0; it * 64 < FT * FT * 16; it += 4) {      // one wave instruction = 64 consecutive quads of the tile image
      const int i = it * 64 + ln0;
      const int px = i >> 4, qq = i & 15;
      const int fy = px / FT, fx = px - fy * FT;
      const int gy = 2 * oh0 + fy, gx = 2 * ow0 + fx, cq = c0 + qq * 4;
      const bool inr = i < FT * FT * 16 && gy < a.PH && gx < a.PW && cq < a.C;
      const bool inner = fy >= 1 && fy <= FT - 2 && fx >= 1 && fx <= FT - 2;
      if (inr && inner)
        __builtin_amdgcn_global_load_lds((glb_vp)(uintptr_t)(dpl + (((size_t)b * a.PH + gy) * a.PW + gx) * a.C + cq),
                                         (lds_vp)(uintptr_t)(sFine + it * 256), 16, 0, 0);
      else if (i < FT * FT * 16)
        *reinterpret_cast<float4*>(&sFine[i * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  double hs0 = 0, hs1 = 0;
  float hscl = 0.f, hinv = 0.f, hmu = 0.f;
  if constexpr (GEMM) {
    const int n = min(tid, a.C - 1);
    hs0 = ng.sums[n]; hs1 = ng.sums[a.C + n]; hscl = ng.scale[n]; hinv = ng.invstd[n]; hmu = ng.mean[n];
  }
  auto head_store = [&]() {
    if (tid < 9 * QL) *reinterpret_cast<float4*>(&sW[ttap * CW + tq]) = tok ? twv : make_float4(0, 0, 0, 0);
    if (a.lazy) {
      if (hlive) {
        BnLive bn; bn.stats = a.ost[hop]; bn.gamma = a.oga[hop]; bn.beta = a.obe[hop]; bn.inv_count = a.oic[hop]; bn.C = a.C; bn.eps = 1e-3f;
        bn_live_coef(bn, hlc, hsc, hsh);
      }
      const bool in = hlz && c0 + hcl < a.C;
      sCo[(2 * hop) * 64 + hcl] = in ? hsc : 1.f; sCo[(2 * hop + 1) * 64 + hcl] = in ? hsh : 0.f;
    }
  };
  if constexpr (GEMM) {
    const int C = a.C, LDZ = C + 8, NQ = C >> 2;      // (row stride: as FnCfg::ZS above - conflict-free ds_read_b128 fragments)
    float* const sDz = sDyn;                        // [100 pixel rows][LDZ] (the seventh 16-row MFMA tile re-reads row 99 for its rows 100 .. 111)
    float* const sCf = sDyn + IH * IW * LDZ;        // [4][C]: a1, a2, a3, mu of the BatchNorm backward
    // the wave's B operand: w[n = 16 kk + 4 g + j][c = c0 + 16 wave + r], kk = 0 .. C / 16 - 1 (at most 14 k groups: C <= 224)
    const int lane = tid & 63, r = lane & 15, gq = lane >> 4, wv_ = tid >> 6;
    const int cb = c0 + (CW == 64 ? wv_ * 16 : 0) + r;
    const bool cvalid = cb < C;
    constexpr int KKN = NKK ? NKK : 14;
    // (round 5: from the TRANSPOSED weight w_t[c][n], as the input-gradient GEMMs take it - one 16-byte load per k group and lane instead of
    // four strided dwords: 7 instead of 28 load instructions per lane in front of the block's first barrier)
    float bw[KKN][4];
#pragma unroll
    for (int kk = 0; kk < KKN; ++kk) {
      const bool on = kk * 16 < C && cvalid;
      const float4 t = mmd_ld4(ng.w + (size_t)(cvalid ? cb : 0) * C + (kk * 16 < C ? kk * 16 : 0) + 4 * gq);
      bw[kk][0] = on ? t.x : 0.f; bw[kk][1] = on ? t.y : 0.f; bw[kk][2] = on ? t.z : 0.f; bw[kk][3] = on ? t.w : 0.f;
    }
    // dz tile: U items (pixel, channel quad) per trip, all of a trip's loads in flight together; the FIRST trip is issued before the
    // coefficient barrier, so its round trip overlaps the one of the batch sums / weights above (C = 112: two trips in all, one exposed)
    constexpr int U = 6;
    const int TOT = IH * IW * NQ;
    float4 gv[U], zv[U];
    size_t off[U];
    int ppv[U], qv[U];
    bool ok[U];
    auto issue = [&](int it0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + u * 256;
        const int pp = it / NQ, q = it - pp * NQ;
        const int ih = oh0 - 1 + pp / IW, iw = ow0 - 1 + pp % IW;
        ppv[u] = pp; qv[u] = q;
        ok[u] = it < TOT && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        off[u] = (((size_t)b * a.H + min(max(ih, 0), a.H - 1)) * a.W + min(max(iw, 0), a.W - 1)) * C + min(q, NQ - 1) * 4;
        gv[u] = mmd_ld4(ng.g + off[u]); zv[u] = mmd_ld4(ng.z + off[u]);      // unconditional (clamped address), masked by ok[] below
      }
    };
    auto finish = [&](int it0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (it0 + u * 256 < TOT) {
          const int pp = ppv[u], q = qv[u];
          float4 d = make_float4(0, 0, 0, 0);
          if (ok[u]) {
            const float4 a1 = *reinterpret_cast<const float4*>(&sCf[q * 4]), a2 = *reinterpret_cast<const float4*>(&sCf[C + q * 4]);
            const float4 a3 = *reinterpret_cast<const float4*>(&sCf[2 * C + q * 4]), mu = *reinterpret_cast<const float4*>(&sCf[3 * C + q * 4]);
            d.x = a1.x * gv[u].x + a2.x * (zv[u].x - mu.x) + a3.x; d.y = a1.y * gv[u].y + a2.y * (zv[u].y - mu.y) + a3.y;
            d.z = a1.z * gv[u].z + a2.z * (zv[u].z - mu.z) + a3.z; d.w = a1.w * gv[u].w + a2.w * (zv[u].w - mu.w) + a3.w;
            const int py = pp / IW, px = pp % IW;      // interior pixels, once per tile: the stored dz the weight-gradient GEMM reads
            if (cc == 0 && ng.dz_out && py >= 1 && py <= TH && px >= 1 && px <= TW) mmd_st4(ng.dz_out + off[u], d);
          }
          *reinterpret_cast<float4*>(&sDz[pp * LDZ + q * 4]) = d;      // pixels outside the image: dz = 0 -> dzd = 0 (what the plain form stages)
        }
      }
    };
    issue(tid);
    head_store();
    if (tid < C) {
      const float m1 = (float)(hs0 * ng.inv_count), m2 = (float)(hs1 * ng.inv_count);
      sCf[tid] = hscl; sCf[C + tid] = -hscl * hinv * m2; sCf[2 * C + tid] = -hscl * m1; sCf[3 * C + tid] = hmu;
    }
    for (int n = tid + 256; n < C; n += 256) {      // (C > 256 only)
      const float m1 = (float)(ng.sums[n] * ng.inv_count), m2 = (float)(ng.sums[C + n] * ng.inv_count);
      const float sc = ng.scale[n];
      sCf[n] = sc; sCf[C + n] = -sc * ng.invstd[n] * m2; sCf[2 * C + n] = -sc * m1; sCf[3 * C + n] = ng.mean[n];
    }
    __syncthreads();                                // coefficients visible
    NODE_T(1);
    if (ng.dgamma && blockIdx.x == 0)
      for (int n = tid; n < C; n += 256) { ng.dgamma[n] += (float)ng.sums[C + n]; ng.dbeta[n] += (float)ng.sums[n]; }
    finish(tid);
    for (int it0 = tid + 256 * U; it0 < TOT; it0 += 256 * U) { issue(it0); finish(it0); }
    __syncthreads();
    NODE_T(2);
    // dzd tile: 7 pixel tiles x this wave's 16 channels (CW = 16: the block's 16 channels, pixel tiles wave and wave + 4)
    constexpr int NMT = CW == 64 ? 7 : 2;
    f32x4 acc7[NMT];
#pragma unroll
    for (int mi = 0; mi < NMT; ++mi) acc7[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkk = C >> 4;
#pragma unroll
    for (int kk = 0; kk < KKN; ++kk) {
      if (NKK || kk < nkk) {
#pragma unroll
        for (int mi = 0; mi < NMT; ++mi) {
          const int mt = CW == 64 ? mi : wv_ + 4 * mi;      // (tile 7 of the last wave: computed on clamped rows, never stored)
          const float4 av = *reinterpret_cast<const float4*>(&sDz[min(mt * 16 + r, IH * IW - 1) * LDZ + kk * 16 + 4 * gq]);
          acc7[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bw[kk][0], acc7[mi], 0, 0, 0);
          acc7[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bw[kk][1], acc7[mi], 0, 0, 0);
          acc7[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bw[kk][2], acc7[mi], 0, 0, 0);
          acc7[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bw[kk][3], acc7[mi], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < NMT; ++mi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mt = CW == 64 ? mi : wv_ + 4 * mi;
        const int pp = mt * 16 + 4 * gq + i;
        if (pp < IH * IW) sIn[pp * CW + (CW == 64 ? wv_ * 16 : 0) + r] = cvalid ? acc7[mi][i] : 0.f;
      }
  } else {
  // the dzd tile: all seven loads of a thread in flight together (unconditional, clamped addresses, masked) - guarded they were seven
  // load -> LDS-store round trips in a row
  constexpr int NST = (IH * IW + 15) / 16;
  float4 sv[NST];
  unsigned sok = 0u;
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int p = (tid >> 4) + 16 * i;
    const int ih = oh0 - 1 + p / IW, iw = ow0 - 1 + p % IW;
    if (cok && p < IH * IW && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) sok |= 1u << i;
    sv[i] = mmd_ld4(dzd + (((size_t)b * a.H + min(max(ih, 0), a.H - 1)) * a.W + min(max(iw, 0), a.W - 1)) * a.C + (cok ? c : 0));
  }
  head_store();
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int p = (tid >> 4) + 16 * i;
    const bool ok = (sok >> i) & 1u;
    if (p < IH * IW) *reinterpret_cast<float4*>(&sIn[p * 64 + c4]) = make_float4(ok ? sv[i].x : 0.f, ok ? sv[i].y : 0.f, ok ? sv[i].z : 0.f, ok ? sv[i].w : 0.f);
  }
  }
  if ((MODE & 4) && sFine && FINE_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the fine tile's LDS-DMA pieces have landed (they count on vmcnt) ...
  __syncthreads();                                                                  // ... in every wave, before the first LDS add below
  NODE_T(3);
  float w[3];
  fuse_weights(a.theta, a.ntheta, w);
  const int p = tid / QL;
  const int orow = p / (TW / R);
  const int ocol0 = (p % (TW / R)) * R;
  float4 acc[R];
#pragma unroll
  for (int o = 0; o < R; ++o) acc[o] = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float4 in[SEG];
    const float* prow = &sIn[((orow + i) * IW + ocol0) * CW + c4];
#pragma unroll
    for (int q = 0; q < SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * CW);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4 wv = *reinterpret_cast<const float4*>(&sW[(i * 3 + j) * CW + c4]);
#pragma unroll
      for (int o = 0; o < R; ++o) {
        acc[o].x += in[o + j].x * wv.x; acc[o].y += in[o + j].y * wv.y;
        acc[o].z += in[o + j].z * wv.z; acc[o].w += in[o + j].w * wv.w;
      }
    }
  }
  NODE_T(4);
  float d[3] = {0.f, 0.f, 0.f};
  const int oh = oh0 + orow;
  const FuseCoef fco{sCo, c4};                     // lazy operands: the finalized BatchNorm coefficients of the block's channel chunk (LDS)
  float4 gq[R], fq[R];
  // BatchNorm-backward sums of the totals written below (x0 / x1 / xu.z != nullptr: this launch is the last contribution to that operand's
  // gradient, the operand is a BatchNorm output; replaces the mmd_bn_bwd_reduce launch of the operand's node)
  const float4 z4 = make_float4(0, 0, 0, 0);
  float4 bs0 = z4, bq0 = z4, bs1 = z4, bq1 = z4, bsu = z4, bqu = z4, mu0 = z4, is0 = z4, mu1 = z4, is1 = z4, muu = z4, isu = z4;
  float4 bsp = z4, bqp = z4, mup = z4, isp = z4;
  if (cok) {
    if ((MODE & 4) && xp.z) { mup = mmd_ld4(xp.mean + c); isp = mmd_ld4(xp.invstd + c); }
    if (x0.z) { mu0 = mmd_ld4(x0.mean + c); is0 = mmd_ld4(x0.invstd + c); }
    if ((MODE & 1) && x1.z) { mu1 = mmd_ld4(x1.mean + c); is1 = mmd_ld4(x1.invstd + c); }
    if ((MODE & 2) && xu.z) { muu = mmd_ld4(xu.mean + c); isu = mmd_ld4(xu.invstd + c); }
  }
  // The thread's four pixels in two pairs: every global load of a pair - operands, the pooled operand's 3x3 windows (loaded ONCE: value,
  // arg-max and the raw element at the arg-max come from the same registers), the running gradients, the BatchNorm inputs of the sums -
  // is issued before anything of the pair is consumed: one round trip per pair where the pixel-by-pixel form took three per pixel
  // (18 of the 35 us of a pooled-operand node on the small maps, tools/dev/node_phases.py)
  constexpr int NWIN = (MODE & 4) ? 9 : 1;
#pragma unroll
  for (int h2 = 0; h2 < R; h2 += PR) {
    float4 r0[PR], r1[PR], ru[PR], qd0[PR], qd1[PR], zb0[PR], zb1[PR], win[PR][NWIN];
    unsigned inb[PR];
    bool okp[PR];
    size_t offp[PR];
#pragma unroll
    for (int oo = 0; oo < PR; ++oo) {
      inb[oo] = 0u;
      const int ow = ow0 + ocol0 + h2 + oo;
      okp[oo] = cok && oh < a.H && ow < a.W;
      offp[oo] = (((size_t)b * a.H + oh) * a.W + ow) * a.C + c;
      r0[oo] = r1[oo] = ru[oo] = qd0[oo] = qd1[oo] = zb0[oo] = zb1[oo] = z4;
      if (okp[oo]) {
        r0[oo] = mmd_ld4(a.in0 + offp[oo]);
        if (MODE & 1) r1[oo] = mmd_ld4(a.in1 + offp[oo]);
        if (MODE & 2) ru[oo] = mmd_ld4(a.up + (((size_t)b * (a.H >> 1) + (oh >> 1)) * (a.W >> 1) + (ow >> 1)) * a.C + c);
        if (MODE & 4) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int y = oh * 2 - a.pad_t + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int x = ow * 2 - a.pad_l + j;
              // (unconditional load from a clamped address, masked afterwards: a guarded load is a branch with a full wait behind it)
              const bool inw = y >= 0 && y < a.PH && x >= 0 && x < a.PW;
              const float4 wv = mmd_ld4(a.pl + (((size_t)b * a.PH + min(max(y, 0), a.PH - 1)) * a.PW + min(max(x, 0), a.PW - 1)) * a.C + c);
              win[oo][i * 3 + j] = inw ? wv : z4;
              if (inw) inb[oo] |= 1u << (i * 3 + j);
            }
          }
        }
        if (d0 && acc0) qd0[oo] = mmd_ld4(d0 + offp[oo]);
        if (d1 && acc1) qd1[oo] = mmd_ld4(d1 + offp[oo]);
        if (d0 && x0.z) zb0[oo] = (x0.z == a.in0) ? r0[oo] : mmd_ld4(x0.z + offp[oo]);      // (a lazy operand IS its BatchNorm's input)
        if ((MODE & 1) && d1 && x1.z) zb1[oo] = (x1.z == a.in1) ? r1[oo] : mmd_ld4(x1.z + offp[oo]);
      }
    }
    NODE_T(9 + h2);
#pragma unroll
    for (int oo = 0; oo < PR; ++oo) {
      const int o = h2 + oo;
      const int ow = ow0 + ocol0 + o;
      gq[o] = z4; fq[o] = z4;
      if (okp[oo]) {
        const size_t off = offp[oo];
        float4 t[4];
        float4 sv = z4;
        int wi = 0;
        {
          float4 v = r0[oo];
          if (a.lazy) v = fuse_aff4(v, fco.sc(0), fco.sh(0));
          t[0] = v; sv.x += w[wi] * v.x; sv.y += w[wi] * v.y; sv.z += w[wi] * v.z; sv.w += w[wi] * v.w; ++wi;
        }
        if (MODE & 1) {
          float4 v = r1[oo];
          if (a.lazy) v = fuse_aff4(v, fco.sc(1), fco.sh(1));
          t[1] = v; sv.x += w[wi] * v.x; sv.y += w[wi] * v.y; sv.z += w[wi] * v.z; sv.w += w[wi] * v.w; ++wi;
        }
        if (MODE & 2) {
          float4 v = ru[oo];
          if (a.lazy) v = fuse_aff4(v, fco.sc(2), fco.sh(2));
          t[2] = v; sv.x += w[wi] * v.x; sv.y += w[wi] * v.y; sv.z += w[wi] * v.z; sv.w += w[wi] * v.w; ++wi;
        }
        int arg[4] = {-1, -1, -1, -1};
        float praw[4] = {0.f, 0.f, 0.f, 0.f}, zarg[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE & 4) {
          // value = max over the window with the zero padding taking part; arg = tap of the FIRST maximum in row-major order, -1 when a
          // padding element wins (pool_window / pool_window_arg above, from one set of loads)
          const bool paff = (a.lazy & 8) != 0;
          const float4 psc = a.lazy ? fco.sc(3) : make_float4(1, 1, 1, 1), psh = a.lazy ? fco.sh(3) : z4;
          float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const bool in = (inb[oo] >> tap) & 1u;
            const float4 raw = win[oo][tap];
            float4 v = raw;
            if (in && paff) v = fuse_aff4(v, psc, psh);
            const float vv[4] = {v.x, v.y, v.z, v.w}, rv[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (vv[q] > best[q]) { best[q] = vv[q]; arg[q] = in ? tap : -1; praw[q] = rv[q]; }
          }
          const float4 v = make_float4(best[0], best[1], best[2], best[3]);
          t[3] = v; sv.x += w[wi] * v.x; sv.y += w[wi] * v.y; sv.z += w[wi] * v.z; sv.w += w[wi] * v.w; ++wi;
          // the BatchNorm input at the arg-max element (the sums of this scatter's share), when it is not the pooled tensor itself (a
          // materialised - non-lazy - operand: D4): the four gathers are issued HERE, together and unconditionally (clamped tap), and used at
          // the end of the pixel - inside the scatter loop below each was a dependent round trip of its own (16 per pixel pair: the 6 x 6
          // node of config 5 took 65 us, the 4 x 4 node of D2 - lazy operands - 33)
          if (dpl && xp.z && xp.z != a.pl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int tp = max(arg[q], 0);
              const int yy = oh * 2 - a.pad_t + tp / 3, xx = ow * 2 - a.pad_l + tp % 3;
              zarg[q] = xp.z[(((size_t)b * a.PH + min(max(yy, 0), a.PH - 1)) * a.PW + min(max(xx, 0), a.PW - 1)) * a.C + c + q];
            }
          }
        }
        if (dwg) fq[o] = make_float4(mmd_swish(sv.x), mmd_swish(sv.y), mmd_swish(sv.z), mmd_swish(sv.w));      // the node's fused activation
        float4 g = acc[o];
        g.x *= mmd_swish_grad(sv.x); g.y *= mmd_swish_grad(sv.y); g.z *= mmd_swish_grad(sv.z); g.w *= mmd_swish_grad(sv.w);
        gq[o] = g;
        if (dx) mmd_st4(dx + off, g);
        if (d0) {
          float4 v = make_float4(g.x * w[0], g.y * w[0], g.z * w[0], g.w * w[0]);
          const float4 mine = v;
          if (acc0) { const float4 q = qd0[oo]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          mmd_st4(d0 + off, v);
          if (x0.z) {
            const float4 gs = (own & 1) ? mine : v, zz = zb0[oo];
            bs0.x += gs.x; bs0.y += gs.y; bs0.z += gs.z; bs0.w += gs.w;
            bq0.x += gs.x * (zz.x - mu0.x) * is0.x; bq0.y += gs.y * (zz.y - mu0.y) * is0.y;
            bq0.z += gs.z * (zz.z - mu0.z) * is0.z; bq0.w += gs.w * (zz.w - mu0.w) * is0.w;
          }
        }
        if (d1) {
          float4 v = make_float4(g.x * w[1], g.y * w[1], g.z * w[1], g.w * w[1]);
          const float4 mine = v;
          if (acc1) { const float4 q = qd1[oo]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          mmd_st4(d1 + off, v);
          if ((MODE & 1) && x1.z) {
            const float4 gs = (own & 2) ? mine : v, zz = zb1[oo];
            bs1.x += gs.x; bs1.y += gs.y; bs1.z += gs.z; bs1.w += gs.w;
            bq1.x += gs.x * (zz.x - mu1.x) * is1.x; bq1.y += gs.y * (zz.y - mu1.y) * is1.y;
            bq1.z += gs.z * (zz.z - mu1.z) * is1.z; bq1.w += gs.w * (zz.w - mu1.w) * is1.w;
          }
        }
        if ((MODE & 4) && dpl) {
          const float wp = w[1 + ((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0)];
          const float gv[4] = {g.x * wp, g.y * wp, g.z * wp, g.w * wp};
          const float muv[4] = {mup.x, mup.y, mup.z, mup.w}, isv[4] = {isp.x, isp.y, isp.z, isp.w};
          const bool zsame = xp.z == a.pl;          // a lazy pooled operand is its BatchNorm's input: the arg-max element is in registers
          float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (arg[q] >= 0) {
              const int yy = oh * 2 - a.pad_t + arg[q] / 3, xx = ow * 2 - a.pad_l + arg[q] % 3;
              const size_t so = (((size_t)b * a.PH + yy) * a.PW + xx) * a.C + c + q;
              if (sFine) atomicAdd(&sFine[((yy - 2 * oh0) * FT + (xx - 2 * ow0)) * 64 + c4 + q], gv[q]);      // (pad 0: host-checked)
              else atomicAdd(&dpl[so], gv[q]);
              if (xp.z) { s4[q] = gv[q]; q4[q] = gv[q] * ((zsame ? praw[q] : zarg[q]) - muv[q]) * isv[q]; }
            }
          }
          bsp.x += s4[0]; bsp.y += s4[1]; bsp.z += s4[2]; bsp.w += s4[3];
          bqp.x += q4[0]; bqp.y += q4[1]; bqp.z += q4[2]; bqp.w += q4[3];
        }
        int wj = 0;
        d[wj++] += g.x * t[0].x + g.y * t[0].y + g.z * t[0].z + g.w * t[0].w;
        if (MODE & 1) d[wj++] += g.x * t[1].x + g.y * t[1].y + g.z * t[1].z + g.w * t[1].w;
        if (MODE & 2) d[wj++] += g.x * t[2].x + g.y * t[2].y + g.z * t[2].z + g.w * t[2].w;
        if (MODE & 4) d[wj++] += g.x * t[3].x + g.y * t[3].y + g.z * t[3].z + g.w * t[3].w;
      }
    }
    NODE_T(10 + h2);
  }
  if ((MODE & 4) && sFine) {
    // the collected fine-map tile -> dpl.  Interior (rows / columns 1 .. 15): this block is the only writer of the launch and the tile
    // started from dpl's values - a plain coalesced store.  Rim: shared with the neighbouring blocks' windows - atomics.
    __syncthreads();
    constexpr int NIN = 15 * 15 * 16;
    if constexpr (FINE_DMA) {
      for (int i = tid; i < NIN; i += 256) {
        const int px = i >> 4, qq = i & 15;
        const int fy = 1 + px / 15, fx = 1 + px % 15;
        const int gy = 2 * oh0 + fy, gx = 2 * ow0 + fx, cq = c0 + qq * 4;
        if (gy < a.PH && gx < a.PW && cq < a.C)
          mmd_st4(dpl + (((size_t)b * a.PH + gy) * a.PW + gx) * a.C + cq, *reinterpret_cast<const float4*>(&sFine[(fy * FT + fx) * 64 + qq * 4]));
      }
    } else {
      // (zero-filled tile: read-add-write, five quads per thread in flight per round - unconditional loads from clamped addresses; only
      // non-zero quads are stored: each output pixel lands on ONE element of its window, about a quarter of the tile)
      for (int i0 = tid; i0 < NIN; i0 += 256 * 5) {
        float4 t[5], v[5];
        float* dst[5];
        bool okq[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = i0 + u * 256, ii = min(i, NIN - 1);
          const int px = ii >> 4, qq = ii & 15;
          const int fy = 1 + px / 15, fx = 1 + px % 15;
          const int gy = 2 * oh0 + fy, gx = 2 * ow0 + fx, cq = c0 + qq * 4;
          okq[u] = i < NIN && gy < a.PH && gx < a.PW && cq < a.C;
          v[u] = *reinterpret_cast<const float4*>(&sFine[(fy * FT + fx) * 64 + qq * 4]);
          dst[u] = dpl + (((size_t)b * a.PH + min(gy, a.PH - 1)) * a.PW + min(gx, a.PW - 1)) * a.C + min(cq, a.C - 4);
          t[u] = mmd_ld4(dst[u]);
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
          if (okq[u] && (v[u].x != 0.f || v[u].y != 0.f || v[u].z != 0.f || v[u].w != 0.f))
            mmd_st4(dst[u], make_float4(t[u].x + v[u].x, t[u].y + v[u].y, t[u].z + v[u].z, t[u].w + v[u].w));
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // rim: 64 pixels x 16 quads
      const int j = tid + k * 256, bp = j >> 4, qq = j & 15;
      int fy, fx;
      if (bp < FT) { fy = 0; fx = bp; }
      else if (bp < 2 * FT) { fy = FT - 1; fx = bp - FT; }
      else { const int rr = bp - 2 * FT; fy = 1 + (rr >> 1); fx = (rr & 1) ? FT - 1 : 0; }
      const int gy = 2 * oh0 + fy, gx = 2 * ow0 + fx, cq = c0 + qq * 4;
      if (gy < a.PH && gx < a.PW && cq < a.C) {
        const float4 vv = *reinterpret_cast<const float4*>(&sFine[(fy * FT + fx) * 64 + qq * 4]);
        float* dp = dpl + (((size_t)b * a.PH + gy) * a.PW + gx) * a.C + cq;
        if (vv.x != 0.f) atomicAdd(dp, vv.x);
        if (vv.y != 0.f) atomicAdd(dp + 1, vv.y);
        if (vv.z != 0.f) atomicAdd(dp + 2, vv.z);
        if (vv.w != 0.f) atomicAdd(dp + 3, vv.w);
      }
    }
  }
  NODE_T(5);
  const int wave = tid >> 6, lane = tid & 63;
  if ((MODE & 2) && dup) {
    // gradient of the nearest-upsampled operand: w_up * (sum over the 2x2 block).  A thread holds 4 pixels of one row (two
    // horizontal pairs); the row below belongs to lane ^ 32 of the same wave (pixel groups 4w..4w+3 = rows 2w, 2w, 2w+1, 2w+1),
    // tiles start on even rows / columns, so every 2x2 block is complete inside one wave: no atomics
    const float wu = w[1 + ((MODE & 1) ? 1 : 0)];
    // (CW = 16: one pixel per thread, pixel p = tid / 4 - the pixel to the right is lane + 4, the one below lane + 32; the even-row /
    // even-column lane of each 2x2 block writes)
#pragma unroll
    for (int q = 0; q < (R == 4 ? 2 : 1); ++q) {
      float4 sv;
      if constexpr (R == 4) {
        sv = make_float4(gq[2 * q].x + gq[2 * q + 1].x, gq[2 * q].y + gq[2 * q + 1].y, gq[2 * q].z + gq[2 * q + 1].z, gq[2 * q].w + gq[2 * q + 1].w);
      } else {
        sv = gq[0];
        sv.x += __shfl_xor(sv.x, 4, 64); sv.y += __shfl_xor(sv.y, 4, 64); sv.z += __shfl_xor(sv.z, 4, 64); sv.w += __shfl_xor(sv.w, 4, 64);
      }
      sv.x += __shfl_xor(sv.x, 32, 64); sv.y += __shfl_xor(sv.y, 32, 64); sv.z += __shfl_xor(sv.z, 32, 64); sv.w += __shfl_xor(sv.w, 32, 64);
      const int uy = oh >> 1, ux = ((ow0 + ocol0) >> 1) + q;
      if (lane < 32 && (R == 4 || (lane & 4) == 0) && cok && uy < (a.H >> 1) && ux < (a.W >> 1)) {
        float* o = dup + (((size_t)b * (a.H >> 1) + uy) * (a.W >> 1) + ux) * a.C + c;
        sv.x *= wu; sv.y *= wu; sv.z *= wu; sv.w *= wu;
        const float4 mine = sv;
        if (acc_up) { float4 pv = mmd_ld4(o); sv.x += pv.x; sv.y += pv.y; sv.z += pv.z; sv.w += pv.w; }
        mmd_st4(o, sv);
        if (xu.z) bnsum_acc4(xu, (size_t)(o - dup), (own & 4) ? mine : sv, muu, isu, bsu, bqu);
      }
    }
  }
  {
    // per-channel block sums: lanes l, l^16, l^32, l^48 of a wave share the channel quad, then the 4 waves through LDS
    auto put = [&](float4 v, int slot) {
#pragma unroll
      for (int o = QL; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
      }
      if (lane < QL) *reinterpret_cast<float4*>(&sBn[(slot * 4 + wave) * 64 + c4]) = v;
    };
    if (x0.z) { put(bs0, 0); put(bq0, 1); }
    if ((MODE & 1) && x1.z) { put(bs1, 2); put(bq1, 3); }
    if ((MODE & 2) && xu.z) { put(bsu, 4); put(bqu, 5); }
    if ((MODE & 4) && xp.z) { put(bsp, 6); put(bqp, 7); }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { float v = wave_sum(d[i]); if (lane == 0) sred[wave * 3 + i] = v; }
  // depthwise weight gradient of the node, from what this launch already holds: dw[i][j] += sum_q f[q] * dzd[q - (i-1, j-1)], f recomputed
  // above and the dzd neighbourhood still in the LDS tile (it is the flipped-tap window of the input gradient: tap' = 8 - tap).
  // Replaces a mmd_dwconv_bwd_weight launch per node and the materialised f it read.
  float4 dwa[9];
  if (dwg) {
#pragma unroll
    for (int t = 0; t < 9; ++t) dwa[t] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float4 in[SEG];
      const float* prow = &sIn[((orow + i) * IW + ocol0) * CW + c4];
#pragma unroll
      for (int q = 0; q < SEG; ++q) in[q] = *reinterpret_cast<const float4*>(prow + q * CW);
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int o = 0; o < R; ++o) {
          dwa[i * 3 + j].x += fq[o].x * in[o + j].x; dwa[i * 3 + j].y += fq[o].y * in[o + j].y;
          dwa[i * 3 + j].z += fq[o].z * in[o + j].z; dwa[i * 3 + j].w += fq[o].w * in[o + j].w;
        }
    }
  }
  NODE_T(6);
  __syncthreads();                                  // sred / sBn complete; every read of the dzd tile done
  NODE_T(7);
  if (tid < a.ntheta) atomicAdd(&wdot[tid], sred[tid] + sred[3 + tid] + sred[6 + tid] + sred[9 + tid]);
  if (tid < 128 && (tid & 63) < CW && c0 + (tid & 63) < a.C) {         // threads 0..63: sum g, 64..127: sum g*xhat
    const int q = tid & 63, hq = tid >> 6;
    auto flush = [&](const BnSumDst& x, int op) {
      const float* r = &sBn[((op * 2 + hq) * 4) * 64 + q];
      atomicAdd(&x.sums[hq * a.C + c0 + q], (double)(r[0] + r[64] + r[128] + r[192]));
    };
    if (x0.z) flush(x0, 0);
    if ((MODE & 1) && x1.z) flush(x1, 1);
    if ((MODE & 2) && xu.z) flush(xu, 2);
    if ((MODE & 4) && xp.z) flush(xp, 3);
  }
  if (dwg) {
    float* sRedW = sIn;                             // [4 waves][9][CW]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float4 v = dwa[t];
#pragma unroll
      for (int o = QL; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
      }
      if (lane < QL) *reinterpret_cast<float4*>(&sRedW[(wave * 9 + t) * CW + c4]) = v;
    }
    __syncthreads();
    for (int i = tid; i < 9 * CW; i += 256) {
      const int t = i / CW, q = i - t * CW;
      if (c0 + q < a.C)
        atomicAdd(&dwg[(size_t)(8 - t) * a.C + c0 + q], sRedW[(0 * 9 + t) * CW + q] + sRedW[(1 * 9 + t) * CW + q] + sRedW[(2 * 9 + t) * CW + q] + sRedW[(3 * 9 + t) * CW + q]);
    }
  }
  NODE_T(8);
}
// launches of the whole-node backward with fewer than this many 64-channel blocks take the 16-channel (small-map) form; 0 = never.
// A per-call argument (NodeForm) since round 6: the library keeps no process-global state besides the communicator.
struct NodeForm { int small_below; int pool_lds; };      // < 0: the library's default (128 blocks; LDS-tile scatter on launches of >= 128 blocks)
static int node_dw_bwd_impl(const float* in0, const float* in1, const float* up, const float* pool,
                            const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B,
                            int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                            float* dw_grad, BnSumDst x0, BnSumDst x1, BnSumDst xu, hipStream_t stream,
                            float* dpl = nullptr, BnSumDst xp = BnSumDst{}, int own = 0, const void* op_scale4 = nullptr,
                            const void* op_shift4 = nullptr, const NodeGemm* ng = nullptr, NodeForm form = NodeForm{-1, -1}) {
  FuseArgs a{};
  int rc = fuse_fill(a, in0, in1, up, pool, theta, B, H, W, C);
  if (!rc) rc = fuse_fill_lazy_bwd(a, op_scale4, op_shift4);
  if (rc || !w_dw || (!dzd && !ng) || !wdot || (!dx && !d0) || (d1 && !in1) || (dup && !up)) return MMD_EINVAL;
  if (ng && (!ng->g || !ng->z || !ng->scale || !ng->mean || !ng->invstd || !ng->sums || !ng->w || (C & 15) || C > 224 ||
             (ng->dgamma == nullptr) != (ng->dbeta == nullptr)))
    return MMD_EINVAL;
  if ((dpl && !pool) || (xp.z && (!dpl || !xp.mean || !xp.invstd || !xp.sums))) return MMD_EINVAL;
  if ((x0.z && (!d0 || !x0.mean || !x0.invstd || !x0.sums)) || (x1.z && (!d1 || !x1.mean || !x1.invstd || !x1.sums)) ||
      (xu.z && (!dup || !xu.mean || !xu.invstd || !xu.sums)))
    return MMD_EINVAL;
  int th = cdiv(H, 8), tw = cdiv(W, 8), cc = cdiv(C, 64);
  const int mode = (in1 ? 1 : 0) | (up ? 2 : 0) | (pool ? 4 : 0);
  // small maps (whole-node form): 16-channel blocks, 3.5x as many, each with a shorter dependent chain (kernel comment, CW)
  static const int cw16_env = getenv("MMD_NODE_CW16_BELOW") ? atoi(getenv("MMD_NODE_CW16_BELOW")) : 128;
  const int node_cw16_below = form.small_below >= 0 ? form.small_below : cw16_env;
  const bool cw16 = ng && (C == 112 || C == 224) && (mode == 2 || mode == 5 || mode == 4) && B * th * tw * cc < node_cw16_below;
  if (cw16) cc = cdiv(C, 16);
  const dim3 grid((unsigned)(B * th * tw * cc)), blk(256);
  // algorithmic bytes: the operand maps read (the up / pooled operand at a quarter / four times the node's size), the upstream gradient
  // (two tensors g, z when the 1x1 input gradient rides in this launch) and one gradient map written per operand
  const double rows = (double)B * H * W;
  // Round 6: the bytes of the launch's CONTRACT, as for the GEMM family (every tensor it must move once by definition): the operand maps
  // read; per operand gradient one write, + one read where it accumulates into a running gradient, + the z of the BatchNorm whose backward
  // sums ride along; the pooled operand's scattered gradient read + written over the fine map; upstream (g, z) read + dz written (whole-node
  // form) or dzd read.  The earlier formula (2 x maps + upstream) left out the accumulate reads, the sums' z tensors and the stored dz:
  // it priced the 64^2 (in, up) launch at 66 MB where the contract is 118 MB and the counters read 146 MB (profiles/r06_notes.md).
  const double s0 = 1.0, s1 = in1 ? 1.0 : 0.0, su = up ? 0.25 : 0.0, sp = pool ? 4.0 : 0.0;
  double maps = s0 + s1 + su + sp;                                       // operand reads
  maps += (d0 ? s0 * (1.0 + (acc0 ? 1.0 : 0.0)) : 0.0) + (d1 ? s1 * (1.0 + (acc1 ? 1.0 : 0.0)) : 0.0) + (dup ? su * (1.0 + (acc_up ? 1.0 : 0.0)) : 0.0);
  maps += (dpl ? 2.0 * sp : 0.0) + (dx ? 1.0 : 0.0);
  maps += (x0.z ? s0 : 0.0) + (x1.z ? s1 : 0.0) + (xu.z ? su : 0.0) + (xp.z ? sp : 0.0);
  maps += ng ? (2.0 + (ng->dz_out ? 1.0 : 0.0)) : 1.0;
  const double nbytes = 4.0 * rows * C * maps;
  const double nflops = rows * C * (2.0 * 9 * 2 + (ng ? 2.0 * C : 0.0));
  mmd_prof_tag(MMD_FAM_NODE_BWD, "nodebwd H%lld C%lld mode%lld full%lld", H, C, mode, ng ? 1 : 0);
  mmd_prof_begin(MMD_FAM_NODE_BWD, stream);
  // pooled operand's gradient through an LDS tile of the fine map (kernel comment): needs pad 0 (even map sizes: every BiFPN level of the
  // 512^2 / 768^2 inputs), 17 x 17 x 64 floats of dynamic LDS behind the GEMM form's dz tile, and the block within 160 KB
  static const int pool_lds_env = getenv("MMD_NO_POOL_LDS") ? 0 : 1;
  const int pool_lds_on = form.pool_lds >= 0 ? form.pool_lds : pool_lds_env;
  const size_t fine_bytes = (size_t)17 * 17 * 64 * sizeof(float);
  const size_t gemm_floats = (size_t)100 * (C + 8) + 4 * C;
  // (only where the launch fills the chip: on the small maps - 16 / 64 blocks, one round of single blocks - the tile's set-up and
  // write-out are on the block's critical path and cost 3 - 5 us more than the scattered atomics they replace; 256 blocks: 60 -> 44 us)
  static const int pool_lds_min = getenv("MMD_POOL_LDS_MIN") ? atoi(getenv("MMD_POOL_LDS_MIN")) : 128;
  const bool pool_lds = pool_lds_on && !cw16 && dpl && a.pad_t == 0 && a.pad_l == 0 && ((int)grid.x >= pool_lds_min || form.pool_lds == 1) &&
                        (ng ? gemm_floats * sizeof(float) + fine_bytes + 30 * 1024 : fine_bytes + 40 * 1024) <= 160 * 1024;
  if (pool_lds) a.pl_lds_off = ng ? (int)gemm_floats : 0;
  if (ng) {       // whole-node backward: the 1x1 conv's input gradient inside this launch
    const size_t lds = gemm_floats * sizeof(float) + (pool_lds ? fine_bytes : 0);
#define MMD_NODE_BWD_GK(M, NK) do { static bool attr = false; \
      if (!attr) { hipFuncSetAttribute((const void*)fuse_dw_bwd_kernel<M, true, NK>, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024); attr = true; } \
      hipLaunchKernelGGL((fuse_dw_bwd_kernel<M, true, NK>), grid, blk, lds, stream, a, w_dw, dzd, dx, wdot, d0, acc0, d1, acc1, dup, acc_up, dw_grad, th, tw, cc, x0, x1, xu, dpl, xp, own, *ng); } while (0)
#define MMD_NODE_BWD_GS(M, NK) do { static bool attr = false; \
      if (!attr) { hipFuncSetAttribute((const void*)fuse_dw_bwd_kernel<M, true, NK, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024); attr = true; } \
      hipLaunchKernelGGL((fuse_dw_bwd_kernel<M, true, NK, 16>), grid, blk, lds, stream, a, w_dw, dzd, dx, wdot, d0, acc0, d1, acc1, dup, acc_up, dw_grad, th, tw, cc, x0, x1, xu, dpl, xp, own, *ng); } while (0)
#define MMD_NODE_BWD_G(M) do { if (cw16) { if (C == 112) MMD_NODE_BWD_GS(M, 7); else MMD_NODE_BWD_GS(M, 14); } \
                               else if (C == 112) MMD_NODE_BWD_GK(M, 7); else if (C == 224) MMD_NODE_BWD_GK(M, 14); else MMD_NODE_BWD_GK(M, 0); } while (0)
    if (mode == 2) MMD_NODE_BWD_G(2); else if (mode == 5) MMD_NODE_BWD_G(5); else if (mode == 4) MMD_NODE_BWD_G(4); else return MMD_EINVAL;
#undef MMD_NODE_BWD_G
#undef MMD_NODE_BWD_GS
#undef MMD_NODE_BWD_GK
    mmd_prof_end(MMD_FAM_NODE_BWD, stream, nflops, nbytes);
    return mmd_check_launch();
  }
  const NodeGemm ng0{};
#define MMD_NODE_BWD(M) do { static bool attr = false; \
    if (!attr) { hipFuncSetAttribute((const void*)fuse_dw_bwd_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
    hipLaunchKernelGGL(fuse_dw_bwd_kernel<M>, grid, blk, pool_lds ? fine_bytes : 0, stream, a, w_dw, dzd, dx, wdot, d0, acc0, d1, acc1, dup, acc_up, dw_grad, th, tw, cc, x0, x1, xu, dpl, xp, own, ng0); } while (0)
  switch (mode) {          // the operand sets of BiFPN._forward_fast_attention: (in, up), (in, td, pool), (in, pool); others through the generic forms
    case 2: MMD_NODE_BWD(2); break;
    case 5: MMD_NODE_BWD(5); break;
    case 4: MMD_NODE_BWD(4); break;
    case 1: MMD_NODE_BWD(1); break;
    case 3: MMD_NODE_BWD(3); break;
    case 6: MMD_NODE_BWD(6); break;
    default: MMD_NODE_BWD(7); break;
  }
#undef MMD_NODE_BWD
  mmd_prof_end(MMD_FAM_NODE_BWD, stream, nflops, nbytes);
  return mmd_check_launch();
}
extern "C" int mmd_bifpn_node_dw_bwd(const float* in0, const float* in1, const float* up, const float* pool,
                                     const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B,
                                     int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                     float* dw_grad, hipStream_t stream) {
  return node_dw_bwd_impl(in0, in1, up, pool, theta, w_dw, dzd, dx, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad,
                          BnSumDst{}, BnSumDst{}, BnSumDst{}, stream);
}
// Same launch; for each operand gradient it writes (d0, d1, dup) an optional (z, mean, invstd, sums): when this launch is the LAST
// contribution to that gradient and the operand is the output of a BatchNorm, sums [2C] (+)= [sum g, sum g*xhat] of the completed
// gradient - the reduce pass of the operand node's BatchNorm backward without a launch of its own (common.h BnSumDst).
extern "C" int mmd_bifpn_node_dw_bwd2(const float* in0, const float* in1, const float* up, const float* pool,
                                      const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B,
                                      int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                      float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                      const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                      const float* zu, const float* meanu, const float* invstdu, double* sumsu, hipStream_t stream) {
  return node_dw_bwd_impl(in0, in1, up, pool, theta, w_dw, dzd, dx, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad,
                          BnSumDst{z0, mean0, invstd0, sums0}, BnSumDst{z1, mean1, invstd1, sums1}, BnSumDst{zu, meanu, invstdu, sumsu}, stream);
}

// mmd_bifpn_node_dw_bwd2 + the POOLED operand's gradient out of the same launch: dpool [B, 2H, 2W, C] (holding zeros or the earlier
// contributions) += w_pool * g at the arg-max element of every output pixel's 3x3 / stride-2 window (fp32 atomics; replaces dx +
// mmd_maxpool_same_bwd_acc).  A scattered gradient has no last writer: the BatchNorm-backward sums of a pooled tensor are therefore kept
// linearly - (zp, meanp, invstdp, sumsp) receive the sums of THIS launch's share, and `own` (bit 0 / 1 / 2 for d0 / d1 / dup) makes the
// other operands' sums cover this launch's share as well, instead of the accumulated total (for operands that are pooled elsewhere).
extern "C" int mmd_bifpn_node_dw_bwd3(const float* in0, const float* in1, const float* up, const float* pool,
                                      const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B,
                                      int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                      float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                      const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                      const float* zu, const float* meanu, const float* invstdu, double* sumsu,
                                      float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own,
                                      hipStream_t stream) {
  return node_dw_bwd_impl(in0, in1, up, pool, theta, w_dw, dzd, dx, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad,
                          BnSumDst{z0, mean0, invstd0, sums0}, BnSumDst{z1, mean1, invstd1, sums1}, BnSumDst{zu, meanu, invstdu, sumsu}, stream,
                          dpool, BnSumDst{zp, meanp, invstdp, sumsp}, own);
}

// mmd_bifpn_node_dw_bwd3 with "lazy" operands (round 4): operand i holds the RAW BatchNorm input and is read as z_i * op_scale4[i] +
// op_shift4[i] (the finalized coefficients; host arrays of 4 entries in operand order, null entry = plain tensor) wherever the launch
// needs the operand's VALUE - the fused activation, the fusion-weight dot products, the pool window's arg-max.  Gradients and BatchNorm
// sums are unchanged (they are w.r.t. the BatchNorm OUTPUT; z0 / z1 / zu / zp are the raw tensors themselves).
extern "C" int mmd_bifpn_node_dw_bwd3_lz(const float* in0, const float* in1, const float* up, const float* pool,
                                         const float* theta, const float* w_dw, const float* dzd, float* dx, float* wdot, int B,
                                         int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                         float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                         const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                         const float* zu, const float* meanu, const float* invstdu, double* sumsu,
                                         float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own,
                                         const void* op_scale4, const void* op_shift4, hipStream_t stream) {
  return node_dw_bwd_impl(in0, in1, up, pool, theta, w_dw, dzd, dx, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad,
                          BnSumDst{z0, mean0, invstd0, sums0}, BnSumDst{z1, mean1, invstd1, sums1}, BnSumDst{zu, meanu, invstdu, sumsu}, stream,
                          dpool, BnSumDst{zp, meanp, invstdp, sumsp}, own, op_scale4, op_shift4);
}

// Whole-node backward (round 4): mmd_bifpn_node_dw_bwd3_lz with the node's 1x1 conv's input gradient inside the launch - dzd is not an
// argument but computed per tile as BnBwd(g, z) . w_pw, with g the gradient w.r.t. the node's BatchNorm output, z its raw 1x1 output,
// (scale, mean, invstd) of that BatchNorm, sums = [sum g, sum g xhat] over `count` rows, w_pw_t [C in, C out] the conv's weight transposed.  dz_out [B*H*W, C]
// receives the evaluated BatchNorm backward (the conv's weight-gradient GEMM reads it), dgamma / dbeta (+)= the sums.  Replaces
// mmd_pwconv_bwd_data_bn + mmd_bifpn_node_dw_bwd3(_lz): one launch per node on the backward's chain instead of two.  C % 16 == 0, C <= 224.
static int node_bwd_full_impl(const float* in0, const float* in1, const float* up, const float* pool,
                                       const float* theta, const float* w_dw, float* wdot, int B,
                                       int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                       float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                       const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                       const float* zu, const float* meanu, const float* invstdu, double* sumsu,
                                       float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own,
                                       const void* op_scale4, const void* op_shift4,
                                       const float* g, const float* z, const float* bn_scale, const float* bn_mean, const float* bn_invstd,
                                       const double* bn_sums, long long count, const float* w_pw, float* dz_out, float* dgamma, float* dbeta,
                                       NodeForm form, hipStream_t stream) {
  if (count <= 0) return MMD_EINVAL;
  NodeGemm ng{g, z, bn_scale, bn_mean, bn_invstd, bn_sums, 1.0 / (double)count, w_pw, dz_out, dgamma, dbeta};
  return node_dw_bwd_impl(in0, in1, up, pool, theta, w_dw, nullptr, nullptr, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad,
                          BnSumDst{z0, mean0, invstd0, sums0}, BnSumDst{z1, mean1, invstd1, sums1}, BnSumDst{zu, meanu, invstdu, sumsu}, stream,
                          dpool, BnSumDst{zp, meanp, invstdp, sumsp}, own, op_scale4, op_shift4, &ng, form);
}
extern "C" int mmd_bifpn_node_bwd_full(const float* in0, const float* in1, const float* up, const float* pool,
                                       const float* theta, const float* w_dw, float* wdot, int B,
                                       int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                       float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                       const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                       const float* zu, const float* meanu, const float* invstdu, double* sumsu,
                                       float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own,
                                       const void* op_scale4, const void* op_shift4,
                                       const float* g, const float* z, const float* bn_scale, const float* bn_mean, const float* bn_invstd,
                                       const double* bn_sums, long long count, const float* w_pw, float* dz_out, float* dgamma, float* dbeta,
                                       hipStream_t stream) {
  return node_bwd_full_impl(in0, in1, up, pool, theta, w_dw, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad, z0, mean0, invstd0, sums0,
                            z1, mean1, invstd1, sums1, zu, meanu, invstdu, sumsu, dpool, zp, meanp, invstdp, sumsp, own, op_scale4, op_shift4,
                            g, z, bn_scale, bn_mean, bn_invstd, bn_sums, count, w_pw, dz_out, dgamma, dbeta, NodeForm{-1, -1}, stream);
}
// the same launch with its block shape chosen by the caller instead of by the launch size (tests, timing): small_below = launches with fewer
// 64-channel blocks than this take the 16-channel form (0: never, 1 << 30: always, < 0: default); pool_lds = the pooled operand's gradient
// through the LDS tile of the fine map (1) or by global atomics (0), < 0: default
extern "C" int mmd_bifpn_node_bwd_full_form(const float* in0, const float* in1, const float* up, const float* pool,
                                       const float* theta, const float* w_dw, float* wdot, int B,
                                       int H, int W, int C, float* d0, int acc0, float* d1, int acc1, float* dup, int acc_up,
                                       float* dw_grad, const float* z0, const float* mean0, const float* invstd0, double* sums0,
                                       const float* z1, const float* mean1, const float* invstd1, double* sums1,
                                       const float* zu, const float* meanu, const float* invstdu, double* sumsu,
                                       float* dpool, const float* zp, const float* meanp, const float* invstdp, double* sumsp, int own,
                                       const void* op_scale4, const void* op_shift4,
                                       const float* g, const float* z, const float* bn_scale, const float* bn_mean, const float* bn_invstd,
                                       const double* bn_sums, long long count, const float* w_pw, float* dz_out, float* dgamma, float* dbeta,
                                       int small_below, int pool_lds, hipStream_t stream) {
  return node_bwd_full_impl(in0, in1, up, pool, theta, w_dw, wdot, B, H, W, C, d0, acc0, d1, acc1, dup, acc_up, dw_grad, z0, mean0, invstd0, sums0,
                            z1, mean1, invstd1, sums1, zu, meanu, invstdu, sumsu, dpool, zp, meanp, invstdp, sumsp, own, op_scale4, op_shift4,
                            g, z, bn_scale, bn_mean, bn_invstd, bn_sums, count, w_pw, dz_out, dgamma, dbeta, NodeForm{small_below, pool_lds}, stream);
}

// d theta_k += [theta_k > 0] * sum_i wdot_i * (delta_ik * S - r_i) / S^2,  S = sum r + eps   
__global__ void fuse_theta_bwd_kernel(const float* theta, const float* wdot, float* dtheta, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float r[3], S = FUSE_EPS;
  for (int i = 0; i < n; ++i) { r[i] = fmaxf(theta[i], 0.f); S += r[i]; }
  for (int k = 0; k < n; ++k) {
    if (!(theta[k] > 0.f)) continue;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += wdot[i] * ((i == k ? S : 0.f) - r[i]) / (S * S);
    dtheta[k] += acc;
  }
}
extern "C" int mmd_bifpn_theta_bwd(const float* theta, const float* wdot, float* dtheta, int n, hipStream_t stream) {
  if (!theta || !wdot || !dtheta || n < 2 || n > 3) return MMD_EINVAL;
  hipLaunchKernelGGL(fuse_theta_bwd_kernel, dim3(1), dim3(64), 0, stream, theta, wdot, dtheta, n);
  return mmd_check_launch();
}

__device__ __forceinline__ float theta_weight(const float* theta, int n, int idx) {
  if (!theta) return 1.f;
  float w[3];
  fuse_weights(theta, n, w);
  return w[idx];
}

// dst (+)= w * src  (same resolution operand of a fusion node)
__global__ void scale_acc_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* theta, int n, int widx,
                                 int accumulate, size_t n4) {
  float w = theta_weight(theta, n, widx);
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = mmd_ld4(src + i * 4);
  v.x *= w; v.y *= w; v.z *= w; v.w *= w;
  if (accumulate) { float4 p = mmd_ld4(dst + i * 4); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
  mmd_st4(dst + i * 4, v);
}
extern "C" int mmd_scale_acc(const float* src, float* dst, const float* theta, int ntheta, int widx, int accumulate,
                             long long numel, hipStream_t stream) {
  if (!src || !dst || numel <= 0 || (numel & 3) || (theta && (widx < 0 || widx >= ntheta || ntheta > 3))) return MMD_EINVAL;
  size_t n4 = (size_t)numel / 4;
  hipLaunchKernelGGL(scale_acc_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, stream, src, dst, theta, ntheta, widx, accumulate, n4);
  return mmd_check_launch();
}

// nearest x2 upsample backward: dst[b,uh,uw,c] (+)= w * sum of the 2x2 children of dx[B,H,W,C]
__global__ void up_bwd_acc_kernel(const float* __restrict__ dx, float* __restrict__ dst, const float* theta, int n, int widx,
                                  int accumulate, int B, int H, int W, int C) {
  float w = theta_weight(theta, n, widx);
  const int c4n = C >> 2, UH = H >> 1, UW = W >> 1;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)B * UH * UW * c4n;
  if (idx >= total) return;
  int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
  int ux = (int)(pix % UW); pix /= UW;
  int uy = (int)(pix % UH); int b = (int)(pix / UH);
  float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = mmd_ld4(dx + (((size_t)b * H + uy * 2 + i) * W + ux * 2 + j) * C + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  s.x *= w; s.y *= w; s.z *= w; s.w *= w;
  float* o = dst + (((size_t)b * UH + uy) * UW + ux) * C + c;
  if (accumulate) { float4 p = mmd_ld4(o); s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
  mmd_st4(o, s);
}
extern "C" int mmd_upsample2_bwd_acc(const float* dx, float* dst, const float* theta, int ntheta, int widx,
                                     int accumulate, int B, int H, int W, int C, hipStream_t stream) {
  if (!dx || !dst || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return MMD_EINVAL;
  if (theta && (widx < 0 || widx >= ntheta || ntheta > 3)) return MMD_EINVAL;
  size_t total = (size_t)B * (H / 2) * (W / 2) * (C >> 2);
  hipLaunchKernelGGL(up_bwd_acc_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, dx, dst, theta, ntheta, widx, accumulate, B, H, W, C);
  return mmd_check_launch();
}

// standalone SAME max-pool 3x3/s2 forward: out[B,OH,OW,C], OH = ceil(PH/2)
__global__ void maxpool_fwd_kernel(const float* __restrict__ src, float* __restrict__ out, int B, int PH, int PW, int C, int OH,
                                   int OW, int pad_t, int pad_l) {
  const int c4n = C >> 2;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)B * OH * OW * c4n;
  if (idx >= total) return;
  int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
  int ow = (int)(pix % OW); pix /= OW;
  int oh = (int)(pix % OH); int b = (int)(pix / OH);
  mmd_st4(out + (((size_t)b * OH + oh) * OW + ow) * C + c, pool_window(src, b, oh, ow, c, PH, PW, C, pad_t, pad_l));
}
static void pool_geom(int n, int* o, int* lo) {
  *o = (n + 1) / 2;
  int extra = (*o - 1) * 2 - n + 3;
  if (extra < 0) extra = 0;
  *lo = extra / 2;
}
extern "C" int mmd_maxpool_same_fwd(const float* src, float* out, int B, int PH, int PW, int C, hipStream_t stream) {
  if (!src || !out || B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  int OH, OW, pt, pl;
  pool_geom(PH, &OH, &pt); pool_geom(PW, &OW, &pl);
  size_t total = (size_t)B * OH * OW * (C >> 2);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, src, out, B, PH, PW, C, OH, OW, pt, pl);
  return mmd_check_launch();
}

// max-pool backward in gather form: every source pixel looks at the (<=4) windows that contain it and takes the
// window's gradient iff it is the FIRST maximum of that window in row-major scan order (torch semantics; a
// zero-padding element that wins swallows the gradient).
// xs.z != nullptr: this launch completes the gradient of `src` = BN(xs.z); it then also accumulates that BatchNorm's backward sums
// (LDS float atomics per block, one double atomic per channel and block; `iters` items per thread keep the block count - the depth of
// the same-address atomics - in the hundreds).  C <= MP_MAXC in that mode.
#define MP_MAXC 512
__global__ __launch_bounds__(256) void maxpool_bwd_acc_kernel(const float* __restrict__ src, const float* __restrict__ dout, float* __restrict__ dst,
                                       const float* theta, int n, int widx, int accumulate, int B, int PH, int PW, int C,
                                       int OH, int OW, int pad_t, int pad_l, BnSumDst xs, int iters) {
  __shared__ float sS[2 * MP_MAXC];
  if (xs.z) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) sS[i] = 0.f;
    __syncthreads();
  }
  float w = theta_weight(theta, n, widx);
  const int c4n = C >> 2;
  const size_t total = (size_t)B * PH * PW * c4n;
  for (int it = 0; it < iters; ++it) {
  size_t idx = ((size_t)blockIdx.x * iters + it) * 256 + threadIdx.x;
  if (idx >= total) break;
  int c = (int)(idx % c4n) * 4; size_t pix = idx / c4n;
  int x = (int)(pix % PW); pix /= PW;
  int y = (int)(pix % PH); int b = (int)(pix / PH);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // windows o with 2o - pad <= y <= 2o - pad + 2
  int oy_lo = (y + pad_t - 2 + 1) >> 1; if (oy_lo < 0) oy_lo = 0;
  int oy_hi = (y + pad_t) >> 1; if (oy_hi > OH - 1) oy_hi = OH - 1;
  int ox_lo = (x + pad_l - 2 + 1) >> 1; if (ox_lo < 0) ox_lo = 0;
  int ox_hi = (x + pad_l) >> 1; if (ox_hi > OW - 1) ox_hi = OW - 1;
  for (int oy = oy_lo; oy <= oy_hi; ++oy)
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int by[4] = {-9, -9, -9, -9}, bx[4] = {-9, -9, -9, -9};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int yy = oy * 2 - pad_t + i;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          int xx = ox * 2 - pad_l + j;
          float4 v = make_float4(0, 0, 0, 0);
          if (yy >= 0 && yy < PH && xx >= 0 && xx < PW) v = mmd_ld4(src + (((size_t)b * PH + yy) * PW + xx) * C + c);
          float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (vv[q] > best[q]) { best[q] = vv[q]; by[q] = yy; bx[q] = xx; }
        }
      }
      float4 g = mmd_ld4(dout + (((size_t)b * OH + oy) * OW + ox) * C + c);
      float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (by[q] == y && bx[q] == x) acc[q] += gg[q];
    }
  float4 o = make_float4(acc[0] * w, acc[1] * w, acc[2] * w, acc[3] * w);
  const size_t doff = (((size_t)b * PH + y) * PW + x) * C + c;
  float* d = dst + doff;
  if (accumulate) { float4 p = mmd_ld4(d); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
  mmd_st4(d, o);
  if (xs.z) {
    const float4 zz = mmd_ld4(xs.z + doff), mu = mmd_ld4(xs.mean + c), is = mmd_ld4(xs.invstd + c);
    atomicAdd(&sS[c], o.x); atomicAdd(&sS[c + 1], o.y); atomicAdd(&sS[c + 2], o.z); atomicAdd(&sS[c + 3], o.w);
    atomicAdd(&sS[C + c], o.x * (zz.x - mu.x) * is.x); atomicAdd(&sS[C + c + 1], o.y * (zz.y - mu.y) * is.y);
    atomicAdd(&sS[C + c + 2], o.z * (zz.z - mu.z) * is.z); atomicAdd(&sS[C + c + 3], o.w * (zz.w - mu.w) * is.w);
  }
  }
  if (xs.z) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&xs.sums[i], (double)sS[i]);
  }
}
static int maxpool_bwd_impl(const float* src, const float* dout, float* dst, const float* theta, int ntheta,
                            int widx, int accumulate, int B, int PH, int PW, int C, BnSumDst xs, hipStream_t stream) {
  if (!src || !dout || !dst || B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  if (theta && (widx < 0 || widx >= ntheta || ntheta > 3)) return MMD_EINVAL;
  if (xs.z && (!xs.mean || !xs.invstd || !xs.sums || C > MP_MAXC)) return MMD_EINVAL;
  int OH, OW, pt, pl;
  pool_geom(PH, &OH, &pt); pool_geom(PW, &OW, &pl);
  size_t total = (size_t)B * PH * PW * (C >> 2);
  // one item per thread: serial iterations cost more latency than they save in same-address atomics (measured: 15 -> 26 us average with up
  // to 16 iterations); a launch with sums therefore wants <= ~1024 blocks (mmd_maxpool_bwd_sums_ok) - the 64 x 64 source level keeps its
  // own reduce pass
  const int iters = 1;
  hipLaunchKernelGGL(maxpool_bwd_acc_kernel, dim3(cdiv(total, 256 * (size_t)iters)), dim3(256), 0, stream, src, dout, dst, theta, ntheta, widx,
                     accumulate, B, PH, PW, C, OH, OW, pt, pl, xs, iters);
  return mmd_check_launch();
}
// 1 when mmd_maxpool_same_bwd_acc2's sums are worth taking for a [B, PH, PW, C] source (block count = depth of the same-address f64 atomics)
extern "C" int mmd_maxpool_bwd_sums_ok(int B, int PH, int PW, int C) {
  return (C <= MP_MAXC && (long long)B * PH * PW * (C >> 2) <= 1024ll * 256) ? 1 : 0;
}
extern "C" int mmd_maxpool_same_bwd_acc(const float* src, const float* dout, float* dst, const float* theta, int ntheta,
                                        int widx, int accumulate, int B, int PH, int PW, int C, hipStream_t stream) {
  return maxpool_bwd_impl(src, dout, dst, theta, ntheta, widx, accumulate, B, PH, PW, C, BnSumDst{}, stream);
}
// + the BatchNorm-backward sums of the completed gradient when this launch is its last contribution and src = BN(z) (common.h BnSumDst)
extern "C" int mmd_maxpool_same_bwd_acc2(const float* src, const float* dout, float* dst, const float* theta, int ntheta,
                                         int widx, int accumulate, int B, int PH, int PW, int C, const float* z, const float* mean,
                                         const float* invstd, double* sums, hipStream_t stream) {
  return maxpool_bwd_impl(src, dout, dst, theta, ntheta, widx, accumulate, B, PH, PW, C, BnSumDst{z, mean, invstd, sums}, stream);
}

// d theta of EVERY fusion node of a net in one launch (the per-node launch above is a leaf that forks off the backward's main chain 40
// times per step): node i has its fusion weights at theta_base + off[i] (n[i] of them), their gradient at dtheta_base + off[i] and its
// dot products at wdot_all + 4*i.
__global__ void fuse_theta_bwd_batched_kernel(const float* theta_base, float* dtheta_base, const float* wdot_all, const long long* desc, int nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nodes) return;
  const long long off = desc[2 * i]; const int n = (int)desc[2 * i + 1];
  const float* theta = theta_base + off; float* dtheta = dtheta_base + off; const float* wdot = wdot_all + 4 * i;
  float r[3], S = FUSE_EPS;
  for (int k = 0; k < n; ++k) { r[k] = fmaxf(theta[k], 0.f); S += r[k]; }
  for (int k = 0; k < n; ++k) {
    if (!(theta[k] > 0.f)) continue;
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc += wdot[j] * ((j == k ? S : 0.f) - r[j]) / (S * S);
    dtheta[k] += acc;
  }
}
extern "C" int mmd_bifpn_theta_bwd_batched(const float* theta_base, float* dtheta_base, const float* wdot_all, const long long* desc,
                                           int nodes, hipStream_t stream) {
  if (!theta_base || !dtheta_base || !wdot_all || !desc || nodes <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(fuse_theta_bwd_batched_kernel, dim3(cdiv(nodes, 64)), dim3(64), 0, stream, theta_base, dtheta_base, wdot_all, desc, nodes);
  return mmd_check_launch();
}
