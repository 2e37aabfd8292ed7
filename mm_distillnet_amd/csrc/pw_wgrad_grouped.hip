// Every 1x1-conv weight gradient of a backward segment in ONE persistent launch + one deterministic fold — CDNA4 / gfx950.
//
//   dW_l[N,K] = sum_m dY_l[m,n] * pro_l(X_l)[m,k]        for all layers l of the segment
//
// Round 1 issued one pw_wgrad_kernel per layer on a side stream (~120 launches per step).  Each launch paid ~8 us of skeleton
// (launch, first load, staging, barriers), ended in N*K fp32 atomics per block (~40 % of a launch, non-deterministic sums) and, being
// GPU-filling, cost the step its kernel time almost 1:1 although it ran "beside" the main chain (profiles/r01_notes.md).  The weight
// gradients are leaves - only the optimizer reads them - and every operand (dY, X: arena tensors) stays alive until the step ends, so
// they can all wait for the end of their backward segment and run as one grid:
//   * work item = (layer, 64x64 output tile, M split): the item table is static (same arena addresses every step) and lives in
//     device memory; persistent blocks walk it with a fixed stride (item i -> block i mod grid): no per-layer launch, small and large
//     layers interleave, the tail of one layer overlaps the head of the next;
//   * an item writes its partial tile with plain stores into a workspace slot of its own; `wgrad_fold_kernel` then adds the splits of
//     every output tile in split order and WRITES dW (no atomics anywhere: gradients are bit-reproducible run to run; dW needs no
//     zeroing).
// Inner loop = pw_wgrad_kernel's (pw_gemm.hip): 32-row steps, dY and pro(X) tiles staged through LDS, v_mfma_f32_32x32x2_f32, each
// of the 4 waves owns a 32x32 sub-tile.  Reference: autograd of nn.Conv2d(k=1) weight (src/YetAnotherEfficientNet.py:427,446;
// src/YetAnotherEfficientDet.py:171,238-265).
#include "common.h"
#include <cstdlib>
#include <cstring>

// Public ABI (include/mmdistill.h repeats this struct).
struct MmdWgradLayer {
  const float* dy; const float* x; float* dw;
  const float* in_scale; const float* in_shift; const float* gate;
  int M, K, N; int in_act; int rows_per_image;
  int mchunk;      // rows per split (multiple of 32)
  int nsplit;      // cdiv(M, mchunk)
  int ntn, ntk;    // 64-wide tiles over N and K
  int item0;       // first work item: item = item0 + (split * ntn + tn) * ntk + tk
  int tile0;       // first output tile (fold): tile = tile0 + tn * ntk + tk
  int pad_;
  long long ws_off;   // floats: partial of item i at ws + ws_off + (i - item0) * 4096
};

#define GW_LD 68
#define GW_BR 32
#define GW_MAXL 512

__global__ __launch_bounds__(256) void wgrad_grouped_kernel(const MmdWgradLayer* __restrict__ L, int nl, int nitems, float* __restrict__ ws) {
  __shared__ float sD[GW_BR * GW_LD];
  __shared__ float sX[GW_BR * GW_LD];
  __shared__ int sItem0[GW_MAXL + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < nl; i += 256) sItem0[i] = L[i].item0;
  if (tid == 0) sItem0[nl] = nitems;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const int c4 = (tid & 15) * 4;         // column offset inside the 64-wide tile
  const int lrow = tid >> 4;             // 0..15
  int li = 0;
  // (an XCD-aware walk - consecutive items, i.e. the tiles that re-read one M slab, on one XCD - measured SLOWER: family 1.33-1.45 ->
  // 1.72-1.79 ms per step; the plain stride spreads every layer over all XCDs and balances better than the L2 reuse is worth)
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    while (sItem0[li + 1] <= item) ++li;              // items are visited in increasing order: amortised O(1)
    li = __builtin_amdgcn_readfirstlane(li);          // block-uniform: the layer record comes in through scalar loads
    const MmdWgradLayer a = L[li];
    int b = item - a.item0;
    const int tk = b % a.ntk; b /= a.ntk;
    const int tn = b % a.ntn; b /= a.ntn;
    const int mbeg = b * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int n0 = tn * 64, k0 = tk * 64;
    const bool nok = (n0 + c4) < a.N, kok = (k0 + c4) < a.K;
    float4 xsc = make_float4(1, 1, 1, 1), xsh = make_float4(0, 0, 0, 0);
    if (a.in_scale && kok) { xsc = mmd_ld4(a.in_scale + k0 + c4); xsh = mmd_ld4(a.in_shift + k0 + c4); }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    constexpr int NL = GW_BR / 16;
    float4 rd[NL], rx[NL], rg[NL]; bool rok[NL];
    auto gload = [&](int mb) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int row = mb + lrow + i * 16;
        rok[i] = row < mend;
        const int rc = rok[i] ? row : mbeg;                     // clamped: loads are unconditional, masked in lstore
        rd[i] = mmd_ld4(a.dy + (size_t)rc * a.N + (nok ? n0 + c4 : 0));
        rx[i] = mmd_ld4(a.x + (size_t)rc * a.K + (kok ? k0 + c4 : 0));
        if (a.gate) rg[i] = mmd_ld4(a.gate + (size_t)(rc / a.rows_per_image) * a.K + (kok ? k0 + c4 : 0));
      }
    };
    auto lstore = [&]() {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        float4 v = rx[i];
        if (a.in_scale) { v.x = v.x * xsc.x + xsh.x; v.y = v.y * xsc.y + xsh.y; v.z = v.z * xsc.z + xsh.z; v.w = v.w * xsc.w + xsh.w; }
        if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
        if (a.gate) { v.x *= rg[i].x; v.y *= rg[i].y; v.z *= rg[i].z; v.w *= rg[i].w; }
        if (!(rok[i] && kok)) v = make_float4(0, 0, 0, 0);
        *reinterpret_cast<float4*>(&sX[(lrow + i * 16) * GW_LD + c4]) = v;
        *reinterpret_cast<float4*>(&sD[(lrow + i * 16) * GW_LD + c4]) = (rok[i] && nok) ? rd[i] : make_float4(0, 0, 0, 0);
      }
    };
    const bool idle = n0 + wn * 32 >= a.N || k0 + wk * 32 >= a.K;      // this wave's 32x32 sub-tile is all N / K padding
    gload(mbeg);
    for (int mb = mbeg; mb < mend; mb += GW_BR) {
      lstore();
      __syncthreads();
      if (mb + GW_BR < mend) gload(mb + GW_BR);
      if (!idle) {
        const float* pd = &sD[h * GW_LD + wn * 32 + r];
        const float* px = &sX[h * GW_LD + wk * 32 + r];
#pragma unroll
        for (int tt = 0; tt < GW_BR / 2; ++tt)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[tt * 2 * GW_LD], px[tt * 2 * GW_LD], acc, 0, 0, 0);
      }
      __syncthreads();
    }
    // partial tile [64 n][64 k] -> this item's workspace slot (128-B row segments per wave instruction)
    float* out = ws + a.ws_off + (size_t)(item - a.item0) * 4096;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = wn * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      out[n * 64 + wk * 32 + r] = acc[q];
    }
  }
}

// one block per output tile: dW[n0.., k0..] = sum over the splits (in split order) of the partial tiles
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const MmdWgradLayer* __restrict__ L, int nl, int ntiles, const float* __restrict__ ws) {
  __shared__ int s_li;
  const int tile = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int lo = 0, hi = nl - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L[mid].tile0 <= tile) lo = mid; else hi = mid - 1; }
    s_li = lo;
  }
  __syncthreads();
  const MmdWgradLayer a = L[s_li];
  const int t = tile - a.tile0;
  const int tn = t / a.ntk, tk = t - tn * a.ntk;
  const int tiles = a.ntn * a.ntk;
  const float* p = ws + a.ws_off + (size_t)t * 4096 + tid * 4;
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < a.nsplit; ++sp) {
    const float* q = p + (size_t)sp * tiles * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = mmd_ld4(q + u * 1024);
      s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = u * 1024 + tid * 4, n = tn * 64 + (e >> 6), k = tk * 64 + (e & 63);
    if (n < a.N && k < a.K) mmd_st4(a.dw + (size_t)n * a.K + k, s[u]);       // K % 4 == 0: a float4 is all-valid or all-out
  }
}

// Host planner: fills mchunk / nsplit / ntn / ntk / item0 / tile0 / ws_off of `layers` (host memory) for splits of about
// `rows_per_item` rows.  -> total items, output tiles and workspace floats.
extern "C" int mmd_wgrad_plan(MmdWgradLayer* layers, int n, int rows_per_item, int* n_items, int* n_tiles, long long* ws_floats) {
  if (!layers || n <= 0 || n > GW_MAXL || !n_items || !n_tiles || !ws_floats) return MMD_EINVAL;
  if (rows_per_item < GW_BR) rows_per_item = GW_BR;
  int items = 0, tiles = 0; long long ws = 0;
  for (int i = 0; i < n; ++i) {
    MmdWgradLayer& a = layers[i];
    if (a.M <= 0 || a.K <= 0 || a.N <= 0 || (a.K & 3) || (a.N & 3) || !a.dy || !a.x || !a.dw) return MMD_EINVAL;
    if (a.gate && a.rows_per_image <= 0) return MMD_EINVAL;
    if ((a.in_scale == nullptr) != (a.in_shift == nullptr)) return MMD_EINVAL;
    if (a.rows_per_image <= 0) a.rows_per_image = 1;
    a.ntn = cdiv(a.N, 64); a.ntk = cdiv(a.K, 64);
    int splits = cdiv(a.M, rows_per_item);
    a.mchunk = cdiv(cdiv(a.M, splits), GW_BR) * GW_BR;
    a.nsplit = cdiv(a.M, a.mchunk);
    a.item0 = items; a.tile0 = tiles; a.ws_off = ws; a.pad_ = 0;
    items += a.nsplit * a.ntn * a.ntk;
    tiles += a.ntn * a.ntk;
    ws += (long long)a.nsplit * a.ntn * a.ntk * 4096;
  }
  *n_items = items; *n_tiles = tiles; *ws_floats = ws;
  return MMD_OK;
}

// layers_dev: the planned table in device memory; ws: workspace of ws_floats floats (needs no initialisation).
extern "C" int mmd_wgrad_grouped(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks,
                                 double flops, double bytes, hipStream_t stream) {
  if (!layers_dev || n_layers <= 0 || n_layers > GW_MAXL || n_items <= 0 || n_tiles <= 0 || !ws) return MMD_EINVAL;
  if (blocks <= 0) blocks = 1024;
  if (blocks > n_items) blocks = n_items;
  mmd_prof_tag(MMD_FAM_PW_WGRAD, "wgrouped L%lld items%lld tiles%lld b%lld", n_layers, n_items, n_tiles, blocks);
  mmd_prof_begin(MMD_FAM_PW_WGRAD, stream);
  hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3(n_tiles), dim3(256), 0, stream, layers_dev, n_layers, n_tiles, ws);
  mmd_prof_end(MMD_FAM_PW_WGRAD, stream, flops, bytes);
  return mmd_check_launch();
}
