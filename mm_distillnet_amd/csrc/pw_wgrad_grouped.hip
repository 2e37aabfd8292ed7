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
#include <type_traits>

// Public ABI (include/mmdistill.h repeats this struct).
struct MmdWgradLayer {
  const float* dy; const float* x; float* dw;
  const float* in_scale; const float* in_shift; const float* gate;
  int M, K, N; int in_act; int rows_per_image;
  int mchunk;      // rows per split (multiple of 32)
  int nsplit;      // cdiv(M, mchunk)
  int ntn, ntk;    // T-wide tiles over N and K
  int item0;       // first work item: item = item0 + (split * ntn + tn) * ntk + tk
  int tile0;       // first output tile (fold): tile = tile0 + tn * ntk + tk
  int pad_;
  long long ws_off;   // floats: partial of item i at ws + ws_off + (i - item0) * T * T  (T = pad_: the planner's tile edge, 64 or 128)
};

#define GW_BR 32
#define GW_MAXL 512

// Output tile edge: 64 (a wave owns one 32x32 sub-tile; default) or 128 (a wave owns 2x2 of them; MMD_WG_TILE=128).  An item reads
// rows x (T + T) floats for a T x T output: with 64-wide tiles the step's weight gradients move 5.4 GB for 2.5 GB of operands (every dY
// slab is re-read per K tile, every X slab per N tile; 6.3 GB measured), 128-wide tiles would move 3.4 GB and make the 112-wide BiFPN /
// head layers single-tile - and measured SLOWER all the same: step 18.5 -> 20.2-20.7 ms at the same MFMA time per item, 19.2 with 4x the
// rows per item (232 VGPRs, two waves per SIMD, a quarter of the items to balance over the chip).  The launch is not HBM-bound.
// MMD_WG_TILE: 64 / 128 = the square forms (one tile edge for every layer); unset = the rectangular form (round 5, further down): 128-wide
// N tiles for the layers with N > 64, 64 x 64 for the thin ones, in one launch.
static int wg_tile() {
  static const int t = getenv("MMD_WG_TILE") ? atoi(getenv("MMD_WG_TILE")) : 0;
  return t == 128 ? 128 : t == 64 ? 64 : 0;
}

// BF (precision "bf16": BASELINE configs[4]): both operands are rounded to bf16 (RNE) at the MFMA input, fp32 accumulate - the arithmetic of
// pw_wgrad_kernel<BF = true> (pw_gemm.hip) and of the oracle's BF16_PW rule; tensors and LDS tiles stay fp32.  Round 4: until then the
// bf16 mode ran one launch per layer (D4 / 768^2: 130 launches, 5.9 ms of kernel time per step).
typedef __bf16 gw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gw_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned gw_pk(float a, float b) {
  gw_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gw_bf16x2));
}
// lane (r, h) of the 32x32x16 bf16 MFMA supplies k = 8h .. 8h + 7 of its row / column r: here 8 consecutive ROWS of the staged slab
__device__ __forceinline__ gw_bf16x8 gw_gather8(const float* q, int ld) {
  gw_u32x4 u = {gw_pk(q[0], q[ld]), gw_pk(q[2 * ld], q[3 * ld]), gw_pk(q[4 * ld], q[5 * ld]), gw_pk(q[6 * ld], q[7 * ld])};
  return __builtin_bit_cast(gw_bf16x8, u);
}

template <int T, bool BF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void wgrad_grouped_kernel(const MmdWgradLayer* __restrict__ L, int nl, int nitems, float* __restrict__ ws) {
  constexpr int LD = T + 4, NTH = T / 4, RPP = 256 / NTH, NL = GW_BR / RPP, S = T / 64;
  __shared__ float sD[GW_BR * LD];
  __shared__ float sX[GW_BR * LD];
  __shared__ int sItem0[GW_MAXL + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < nl; i += 256) sItem0[i] = L[i].item0;
  if (tid == 0) sItem0[nl] = nitems;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const int c4 = (tid & (NTH - 1)) * 4;  // column offset inside the T-wide tile
  const int lrow = tid / NTH;            // 0 .. RPP-1
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
    const int n0 = tn * T, k0 = tk * T;
    const bool nok = (n0 + c4) < a.N, kok = (k0 + c4) < a.K;
    float4 xsc = make_float4(1, 1, 1, 1), xsh = make_float4(0, 0, 0, 0);
    if (a.in_scale && kok) { xsc = mmd_ldg4(a.in_scale + k0 + c4); xsh = mmd_ldg4(a.in_shift + k0 + c4); }
    f32x16 acc[S][S];
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
      for (int v = 0; v < S; ++v)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][v][q] = 0.f;
    float4 rd[NL], rx[NL], rg[NL]; bool rok[NL];
    auto gload = [&](int mb) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int row = mb + lrow + i * RPP;
        rok[i] = row < mend;
        const int rc = rok[i] ? row : mbeg;                     // clamped: loads are unconditional, masked in lstore
        rd[i] = mmd_ldg4(a.dy + (size_t)rc * a.N + (nok ? n0 + c4 : 0));
        rx[i] = mmd_ldg4(a.x + (size_t)rc * a.K + (kok ? k0 + c4 : 0));
        if (a.gate) rg[i] = mmd_ldg4(a.gate + (size_t)(rc / a.rows_per_image) * a.K + (kok ? k0 + c4 : 0));
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
        *reinterpret_cast<float4*>(&sX[(lrow + i * RPP) * LD + c4]) = v;
        *reinterpret_cast<float4*>(&sD[(lrow + i * RPP) * LD + c4]) = (rok[i] && nok) ? rd[i] : make_float4(0, 0, 0, 0);
      }
    };
    // 32x32 sub-tiles of this wave that are all N / K padding are skipped: the live ones are a prefix in each dimension (padding sits at
    // the end), so the MFMA loop is instantiated per (live n sub-tiles, live k sub-tiles) - no per-MFMA branch
    int nu = 0, nv = 0;
#pragma unroll
    for (int u = 0; u < S; ++u) { nu += (n0 + (wn * S + u) * 32 < a.N) ? 1 : 0; nv += (k0 + (wk * S + u) * 32 < a.K) ? 1 : 0; }
    nu = __builtin_amdgcn_readfirstlane(nu); nv = __builtin_amdgcn_readfirstlane(nv);
    const float* const pd = &sD[h * LD + wn * S * 32 + r];
    const float* const px = &sX[h * LD + wk * S * 32 + r];
    auto mma = [&](auto nu_c, auto nv_c) {
      constexpr int NU = decltype(nu_c)::value, NV = decltype(nv_c)::value;
      if constexpr (BF) {
#pragma unroll
        for (int gq = 0; gq < GW_BR / 16; ++gq) {
          gw_bf16x8 dv[NU], xv[NV];
#pragma unroll
          for (int u = 0; u < NU; ++u) dv[u] = gw_gather8(&sD[(gq * 16 + h * 8) * LD + (wn * S + u) * 32 + r], LD);
#pragma unroll
          for (int v = 0; v < NV; ++v) xv[v] = gw_gather8(&sX[(gq * 16 + h * 8) * LD + (wk * S + v) * 32 + r], LD);
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv[u], xv[v], acc[u][v], 0, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int tt = 0; tt < GW_BR / 2; ++tt) {
        float dv[NU], xv[NV];
#pragma unroll
        for (int u = 0; u < NU; ++u) dv[u] = pd[tt * 2 * LD + u * 32];
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[v] = px[tt * 2 * LD + v * 32];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[u], xv[v], acc[u][v], 0, 0, 0);
      }
    };
    gload(mbeg);
    for (int mb = mbeg; mb < mend; mb += GW_BR) {
      lstore();
      __syncthreads();
      if (mb + GW_BR < mend) gload(mb + GW_BR);
      if (nu && nv) {
        if constexpr (S == 1) mma(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        else {
          if (nu == 2 && nv == 2) mma(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
          else if (nu == 2) mma(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
          else if (nv == 2) mma(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
          else mma(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
      }
      __syncthreads();
    }
    // partial tile [T n][T k] -> this item's workspace slot (128-B row segments per wave instruction)
    float* out = ws + a.ws_off + (size_t)(item - a.item0) * (T * T);
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
      for (int v = 0; v < S; ++v)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = (wn * S + u) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          out[n * T + (wk * S + v) * 32 + r] = acc[u][v][q];
        }
  }
}

// ---- round 5: rectangular output tiles, 128 (N) x 64 (K) for the layers with N > 64 ------------------------------------------------
// A 64 x 64 item reads rows x (64 + 64) floats for 4096 outputs: every dY slab is re-read once per K tile, every X slab once per N tile -
// 5.1 GB of HBM traffic per step for 2.5 GB of operands at ~4.1 TB/s (profiles/r04_pmc_hbm_traffic.csv): the launch runs at what the
// memory system delivers.  Here a layer with N > 64 takes 128-wide N tiles (per-layer tile width `pad_`, 64 or 128): an item reads rows x
// (128 + 64) floats for 8192 outputs (21 flop per operand byte instead of 16, the X slabs re-read half as often), a wave owns two
// 32 x 32 sub-tiles along N (32 accumulator registers; the 128 x 128 form needed 232 VGPRs and measured slower), 32 MFMAs per wave
// between the two barriers of a 32-row step instead of 16.  Thin layers (N <= 64: the 256^2 / 128^2 project convs) keep 64 x 64 items in
// the same launch - all four waves stay busy there.  Same table, planner, workspace scheme and split-ordered fold: bit-reproducible.
#ifndef WG_RECT_WAVES
#define WG_RECT_WAVES 4      /* waves per SIMD the register allocation aims at: 4 = at most 128 VGPRs (137 without the bound: three waves) */
#endif
// R32: every item's row range is whole 32-row steps (every layer's M a multiple of 32 - the caller's promise, mmd_wgrad_rows32): the row
// clamps / masks and the per-load address multiplies are compiled out.  (Both forms in ONE kernel, chosen per item, spilled 51 registers.)
template <bool BF, bool R32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WG_RECT_WAVES, 8))) void wgrad_grouped_rect_kernel(const MmdWgradLayer* __restrict__ L, int nl, int nitems, float* __restrict__ ws) {
  constexpr int LDD = 128 + 4, LDX = 64 + 4;
  __shared__ float sD[GW_BR * LDD];
  __shared__ float sX[GW_BR * LDX];
  __shared__ int sItem0[GW_MAXL + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < nl; i += 256) sItem0[i] = L[i].item0;
  if (tid == 0) sItem0[nl] = nitems;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const int c4x = (tid & 15) * 4, lrowx = tid >> 4;      // X tile: 16 float4 per row, 16 rows per pass, 2 passes
  int li = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    while (sItem0[li + 1] <= item) ++li;
    li = __builtin_amdgcn_readfirstlane(li);
    const MmdWgradLayer a = L[li];
    const int TN = a.pad_;                                // 64 or 128 (block-uniform)
    const bool wide = TN == 128;
    int b = item - a.item0;
    const int tk = b % a.ntk; b /= a.ntk;
    const int tn = b % a.ntn; b /= a.ntn;
    const int mbeg = b * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int n0 = tn * TN, k0 = tk * 64;
    // dY tile: TN / 4 float4 per row -> 8 (wide) or 16 rows per pass, 4 or 2 passes
    const int c4d = wide ? (tid & 31) * 4 : (tid & 15) * 4, lrowd = wide ? tid >> 5 : tid >> 4, rppd = wide ? 8 : 16;
    const bool nok = (n0 + c4d) < a.N, kok = (k0 + c4x) < a.K;
    float4 xsc = make_float4(1, 1, 1, 1), xsh = make_float4(0, 0, 0, 0);
    if (a.in_scale && kok) { xsc = mmd_ldg4(a.in_scale + k0 + c4x); xsh = mmd_ldg4(a.in_shift + k0 + c4x); }
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;
    float4 rd[4], rx[2], rg[2]; bool rokd[4], rokx[2];
    int gimg[2] = {0, 0}, grem[2] = {0, 0}, gimg0 = 0;          // gate rows: image index / row inside the image of the X tile's two row slots
    if (a.gate) {
      gimg0 = mbeg / a.rows_per_image;
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int r0_ = mbeg + lrowx + i * 16; gimg[i] = r0_ / a.rows_per_image; grem[i] = r0_ - gimg[i] * a.rows_per_image; }
    }
    // Round 6: running operand pointers (a row step is one 64-bit add per load instead of a 64-bit multiply-add: 16 v_mad_i64_i32 + 27
    // v_lshl_add_u64 per 32-row step in the ISA) and - FULL: the item's row range is whole 32-row steps, i.e. every layer of a B >= 2 net -
    // no row clamps and no row masks at all (57 v_cndmask per step).  On this kernel VALU instructions are MFMA time.
    // (ONE running pointer per operand - the thread's first row; its other rows are wave-uniform offsets from it: pointer arrays cost the
    // 128-register bound 18 - 25 spilled registers)
    const float* pdy0 = nullptr; const float* pxx0 = nullptr;
    if constexpr (R32) {
      pdy0 = a.dy + (size_t)(mbeg + lrowd) * a.N + (nok ? n0 + c4d : 0);
      pxx0 = a.x + (size_t)(mbeg + lrowx) * a.K + (kok ? k0 + c4x : 0);
    }
    const int dro = rppd * a.N, xro = 16 * a.K;          // element offsets between a thread's rows (wave-uniform)
    const int dstep = GW_BR * a.N, xstep = GW_BR * a.K;
    auto gload = [&](int mb, auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < 2 || wide) {
          if constexpr (FULL) {
            rd[i] = mmd_ldg4(pdy0 + i * dro);
          } else {
            const int row = mb + lrowd + i * rppd;
            rokd[i] = row < mend;
            rd[i] = mmd_ldg4(a.dy + (size_t)(rokd[i] ? row : mbeg) * a.N + (nok ? n0 + c4d : 0));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = mb + lrowx + i * 16;
        if constexpr (FULL) {
          rx[i] = mmd_ldg4(pxx0 + i * xro);
        } else {
          rokx[i] = row < mend;
          const int rc = rokx[i] ? row : mbeg;
          rx[i] = mmd_ldg4(a.x + (size_t)rc * a.K + (kok ? k0 + c4x : 0));
        }
        if (a.gate) {
          // image of the row, carried from step to step (rows advance by GW_BR: an integer division per load was ~20 VALU instructions)
          while (grem[i] >= a.rows_per_image) { grem[i] -= a.rows_per_image; ++gimg[i]; }
          rg[i] = mmd_ldg4(a.gate + (size_t)((FULL || rokx[i]) ? gimg[i] : gimg0) * a.K + (kok ? k0 + c4x : 0));
          grem[i] += GW_BR;
        }
      }
      if constexpr (FULL) { pdy0 += dstep; pxx0 += xstep; }
    };
    // Only ROWS past the item's end are zeroed.  Columns past N / K need no mask: their loads come from clamped (valid, finite) addresses
    // and only feed output elements n >= N / k >= K of the partial tile, which the fold never stores - 24 selects per step instead of 48
    // on a kernel whose VALU instructions are MFMA time.
    auto lstore = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < 2 || wide)
          *reinterpret_cast<float4*>(&sD[(lrowd + i * rppd) * LDD + c4d]) = (FULL || rokd[i]) ? rd[i] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 v = rx[i];
        if (a.in_scale) { v.x = v.x * xsc.x + xsh.x; v.y = v.y * xsc.y + xsh.y; v.z = v.z * xsc.z + xsh.z; v.w = v.w * xsc.w + xsh.w; }
        if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
        if (a.gate) { v.x *= rg[i].x; v.y *= rg[i].y; v.z *= rg[i].z; v.w *= rg[i].w; }
        if constexpr (!FULL) { if (!rokx[i]) v = make_float4(0, 0, 0, 0); }
        *reinterpret_cast<float4*>(&sX[(lrowx + i * 16) * LDX + c4x]) = v;
      }
    };
    // this wave's n sub-tiles: 64-wide tile -> one (wn * 32), 128-wide -> two (wn * 64, wn * 64 + 32); the live ones are a prefix
    const int nb = wide ? wn * 64 : wn * 32;
    int nu = (n0 + nb < a.N ? 1 : 0) + ((wide && n0 + nb + 32 < a.N) ? 1 : 0);
    int nv = (k0 + wk * 32 < a.K) ? 1 : 0;
    nu = __builtin_amdgcn_readfirstlane(nu); nv = __builtin_amdgcn_readfirstlane(nv);
    const float* const pd = &sD[h * LDD + nb + r];
    const float* const px = &sX[h * LDX + wk * 32 + r];
    auto mma = [&](auto nu_c) {
      constexpr int NU = decltype(nu_c)::value;
      if constexpr (BF) {
#pragma unroll
        for (int gq = 0; gq < GW_BR / 16; ++gq) {
          gw_bf16x8 dv[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) dv[u] = gw_gather8(&sD[(gq * 16 + h * 8) * LDD + nb + u * 32 + r], LDD);
          const gw_bf16x8 xv = gw_gather8(&sX[(gq * 16 + h * 8) * LDX + wk * 32 + r], LDX);
#pragma unroll
          for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv[u], xv, acc[u], 0, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int tt = 0; tt < GW_BR / 2; ++tt) {
        float dv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) dv[u] = pd[tt * 2 * LDD + u * 32];
        const float xv = px[tt * 2 * LDX];
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[u], xv, acc[u], 0, 0, 0);
      }
    };
    // The row loop is instantiated per number of live sub-tiles (item-uniform), not branched inside: with `if (nu == 2) ... else ...` in
    // the loop body hipcc kept the accumulators of the two variants in different registers and copied all 32 of them at every merge -
    // 40 v_mov per 32-row step beside 32 MFMAs (ISA of the first version), on a kernel whose VALU and fp32-MFMA work share one pipe.
    auto rows = [&](auto nu_c, auto full_c) {
      constexpr int NU = decltype(nu_c)::value;
      gload(mbeg, full_c);
      for (int mb = mbeg; mb < mend; mb += GW_BR) {
        lstore(full_c);
        __syncthreads();
        if (mb + GW_BR < mend) gload(mb + GW_BR, full_c);
        if constexpr (NU > 0) mma(nu_c);
        __syncthreads();
      }
    };
    // BARRIER CONTRACT: the four waves of a block may run DIFFERENT instantiations of `rows` (nu / nv are wave-uniform, not
    // block-uniform; `full` is item-uniform), so they meet at __syncthreads() calls at different program counters.  That is sound on gfx950
    // because s_barrier counts arrivals per workgroup, whatever the PC - and ONLY because every instantiation executes exactly the same number
    // of barriers: two per 32-row step (one behind lstore, one behind the MFMAs), (mend - mbeg) / GW_BR steps, with item-uniform mbeg / mend.
    // Anything added to one instantiation's loop that contains a barrier must be added to all of them.
    const int nuw = (!nv || nu == 0) ? 0 : nu;      // (a wave without live sub-tiles still stages its share of the slabs)
    if (nuw == 0) rows(std::integral_constant<int, 0>{}, std::integral_constant<bool, R32>{});
    else if (nuw == 2) rows(std::integral_constant<int, 2>{}, std::integral_constant<bool, R32>{});
    else rows(std::integral_constant<int, 1>{}, std::integral_constant<bool, R32>{});
    // partial tile [TN n][64 k] -> this item's workspace slot
    float* out = ws + a.ws_off + (size_t)(item - a.item0) * (TN * 64);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (u == 0 || wide)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = nb + u * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          out[n * 64 + wk * 32 + r] = acc[u][q];
        }
  }
}

// ---- round 6: the rectangular form with fp32 products on the bf16 matrix pipe (split form, common.h) --------------------------------------
// The fp32 kernel above is MFMA-bound (54 % of its SIMD cycles in v_mfma_f32_32x32x2_f32: 32 x 64 cycles per wave and 32-row step).  Here a
// step's slabs are split into three bf16 planes while they are staged and stored TRANSPOSED, [column][k], so that a lane's eight k of a
// 16-deep MFMA group are one ds_read_b128 per plane (the bf16 form above gathers them with eight ds_read_b32):
//   * the reduction index of a weight gradient is the ROW, and the order of k inside an MFMA is free as long as both operands agree:
//     k slot w (0..15) of a 32-row step holds rows (w, w + 16) as one bf16 pair - exactly the two rows a staging thread of the X tile (and
//     of a 64-wide dY tile) holds, so one 32-bit store per column and plane; a thread of a 128-wide dY tile holds rows 2l, 2l + 1, 2l + 16,
//     2l + 17 -> slots 2l, 2l + 1: one 64-bit store;
//   * sT[column][3 planes x 16 words + 4 pad] with the 4-word groups of a plane XOR-ed by (column >> 5) & 3: fragment reads conflict-free,
//     the staging stores 2-way (unswizzled: 16-way; tools/dev/lds_bank_model.py);
//   * 24 bf16 MFMAs x 32 cycles per wave and step instead of 32 x 64.
// Whole 32-row steps only (the caller's rows32 promise: no row masks); same table, workspace and fold - bit-reproducible.
#define GW_LD3 52
#define GW_MAXL3 255      // layers per launch on this kernel: its table copy must leave four 40 KB blocks per CU
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void wgrad_grouped_split_kernel(const MmdWgradLayer* __restrict__ L, int nl, int nitems, float* __restrict__ ws, int xcd8) {
  __shared__ unsigned sD[128 * GW_LD3];
  __shared__ unsigned sX[64 * GW_LD3];
  __shared__ int sItem0[GW_MAXL3 + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < nl; i += 256) sItem0[i] = L[i].item0;
  if (tid == 0) sItem0[nl] = nitems;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const int c4x = (tid & 15) * 4, lx = tid >> 4;      // X tile: rows lx, lx + 16 of a step -> k slot lx
  int li = 0;
  // XCD-aware item order (round 6): blocks b, b + 8, ... share an XCD and its L2.  Consecutive items are the K tiles / N tiles of ONE row range -
  // they re-read the same dY / X slabs - so each XCD takes runs of 2^xcd8 = 8 consecutive items inside every group of 64 blocks: the launch's
  // HBM traffic 3.96 -> 2.67 GB per step (rocprofv3 FETCH_SIZE / WRITE_SIZE; 1.97x -> 1.33x its algorithmic bytes), step -0.05 .. -0.08 ms over
  // ten alternating pairs.  (Round 4's XCD-aware walk gave every XCD one contiguous EIGHTH of the table and lost to the imbalance between the
  // layers; runs of 8 keep every layer spread over all XCDs.  Runs of 4 / 16 / 32 measured the same within noise.  MMD_WG_XCD8=0: plain order.)
  const int b0 = blockIdx.x;
  const int rl = xcd8, gm = (8 << rl) - 1;      // run length 2^rl, group of 8 runs
  const int vb = (rl && b0 < (int)(gridDim.x & ~(unsigned)gm)) ? ((b0 & ~gm) | ((b0 & 7) << rl) | ((b0 >> 3) & ((1 << rl) - 1))) : b0;      // (a trailing partial group keeps its order)
  for (int item = vb; item < nitems; item += gridDim.x) {
    while (sItem0[li + 1] <= item) ++li;
    li = __builtin_amdgcn_readfirstlane(li);
    const MmdWgradLayer a = L[li];
    const int TN = a.pad_;
    const bool wide = TN == 128;
    int b = item - a.item0;
    const int tk = b % a.ntk; b /= a.ntk;
    const int tn = b % a.ntn; b /= a.ntn;
    const int mbeg = b * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int n0 = tn * TN, k0 = tk * 64;
    // dY tile: 128-wide -> thread (column quad tid & 31, l = tid >> 5) holds rows 2l, 2l + 1, 2l + 16, 2l + 17; 64-wide -> (tid & 15, l = tid >> 4), rows l, l + 16
    const int c4d = wide ? (tid & 31) * 4 : (tid & 15) * 4, ld_ = wide ? tid >> 5 : tid >> 4;
    const bool nok = (n0 + c4d) < a.N, kok = (k0 + c4x) < a.K;
    float4 xsc = make_float4(1, 1, 1, 1), xsh = make_float4(0, 0, 0, 0);
    if (a.in_scale && kok) { xsc = mmd_ldg4(a.in_scale + k0 + c4x); xsh = mmd_ldg4(a.in_shift + k0 + c4x); }
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;
    float4 rd[4], rx[2], rg[2];
    int gimg[2] = {0, 0}, grem[2] = {0, 0};
    if (a.gate) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int r0_ = mbeg + lx + i * 16; gimg[i] = r0_ / a.rows_per_image; grem[i] = r0_ - gimg[i] * a.rows_per_image; }
    }
    // columns past N / K: loads from clamped (valid, finite) addresses; they only feed output elements the fold never stores
    const float* pdy0 = a.dy + (size_t)(mbeg + (wide ? 2 * ld_ : ld_)) * a.N + (nok ? n0 + c4d : 0);
    const float* pxx0 = a.x + (size_t)(mbeg + lx) * a.K + (kok ? k0 + c4x : 0);
    const int d1 = wide ? a.N : 16 * a.N, d2 = 16 * a.N, d3 = 17 * a.N, xro = 16 * a.K;      // element offsets of a thread's other rows (wave-uniform)
    const int dstep = GW_BR * a.N, xstep = GW_BR * a.K;
    auto gload = [&]() {
      rd[0] = mmd_ldg4(pdy0); rd[1] = mmd_ldg4(pdy0 + d1);
      if (wide) { rd[2] = mmd_ldg4(pdy0 + d2); rd[3] = mmd_ldg4(pdy0 + d3); }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rx[i] = mmd_ldg4(pxx0 + i * xro);
        if (a.gate) {
          while (grem[i] >= a.rows_per_image) { grem[i] -= a.rows_per_image; ++gimg[i]; }
          rg[i] = mmd_ldg4(a.gate + (size_t)gimg[i] * a.K + (kok ? k0 + c4x : 0));
          grem[i] += GW_BR;
        }
      }
      pdy0 += dstep; pxx0 += xstep;
    };
    // word address of k slot w of column c inside a tile: plane p adds 16 p
    auto slot = [](int c, int w) { return c * GW_LD3 + ((((w >> 2) ^ (c >> 5)) & 3) << 2) + (w & 3); };
    auto lstore = [&]() {
      if (wide) {
        const float e0[4] = {rd[0].x, rd[0].y, rd[0].z, rd[0].w}, e1[4] = {rd[1].x, rd[1].y, rd[1].z, rd[1].w};
        const float e2[4] = {rd[2].x, rd[2].y, rd[2].z, rd[2].w}, e3[4] = {rd[3].x, rd[3].y, rd[3].z, rd[3].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned h0, m0, l0, h1, m1, l1;
          mmd_split3_pk(e0[j], e2[j], h0, m0, l0);      // slot 2l:     rows 2l,     2l + 16
          mmd_split3_pk(e1[j], e3[j], h1, m1, l1);      // slot 2l + 1: rows 2l + 1, 2l + 17
          unsigned* q = &sD[slot(c4d + j, 2 * ld_)];
          *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(q + 16) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(q + 32) = make_uint2(l0, l1);
        }
      } else {
        const float e0[4] = {rd[0].x, rd[0].y, rd[0].z, rd[0].w}, e1[4] = {rd[1].x, rd[1].y, rd[1].z, rd[1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned h0, m0, l0;
          mmd_split3_pk(e0[j], e1[j], h0, m0, l0);      // slot l: rows l, l + 16
          unsigned* q = &sD[slot(c4d + j, ld_)];
          q[0] = h0; q[16] = m0; q[32] = l0;
        }
      }
      float4 v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[i] = rx[i];
        if (a.in_scale) { v[i].x = v[i].x * xsc.x + xsh.x; v[i].y = v[i].y * xsc.y + xsh.y; v[i].z = v[i].z * xsc.z + xsh.z; v[i].w = v[i].w * xsc.w + xsh.w; }
        if (a.in_act == MMD_ACT_SWISH) { v[i].x = mmd_swish(v[i].x); v[i].y = mmd_swish(v[i].y); v[i].z = mmd_swish(v[i].z); v[i].w = mmd_swish(v[i].w); }
        if (a.gate) { v[i].x *= rg[i].x; v[i].y *= rg[i].y; v[i].z *= rg[i].z; v[i].w *= rg[i].w; }
      }
      const float x0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, x1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h0, m0, l0;
        mmd_split3_pk(x0[j], x1[j], h0, m0, l0);
        unsigned* q = &sX[slot(c4x + j, lx)];
        q[0] = h0; q[16] = m0; q[32] = l0;
      }
    };
    const int nb = wide ? wn * 64 : wn * 32;
    int nu = (n0 + nb < a.N ? 1 : 0) + ((wide && n0 + nb + 32 < a.N) ? 1 : 0);
    int nv = (k0 + wk * 32 < a.K) ? 1 : 0;
    nu = __builtin_amdgcn_readfirstlane(nu); nv = __builtin_amdgcn_readfirstlane(nv);
    auto ldf = [](const unsigned* p) { return __builtin_bit_cast(gw_bf16x8, *reinterpret_cast<const gw_u32x4*>(p)); };
    auto mma = [&](auto nu_c) {
      constexpr int NU = decltype(nu_c)::value;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int w0 = (g * 2 + h) * 4;      // this lane's four k slots of the group
        const unsigned* qx = &sX[slot(wk * 32 + r, w0)];
        const gw_bf16x8 xh = ldf(qx), xm = ldf(qx + 16), xl = ldf(qx + 32);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const unsigned* qd = &sD[slot(nb + u * 32 + r, w0)];
          const gw_bf16x8 dh = ldf(qd), dm = ldf(qd + 16), dl = ldf(qd + 32);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl, xh, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, xl, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dm, xm, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dm, xh, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, xm, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, xh, acc[u], 0, 0, 0);
        }
      }
    };
    // (instantiated per number of live sub-tiles, two barriers per step in every instantiation: the BARRIER CONTRACT of the kernel above)
    auto rows = [&](auto nu_c) {
      constexpr int NU = decltype(nu_c)::value;
      gload();
      for (int mb = mbeg; mb < mend; mb += GW_BR) {
        lstore();
        __syncthreads();
        if (mb + GW_BR < mend) gload();
        if constexpr (NU > 0) mma(nu_c);
        __syncthreads();
      }
    };
    const int nuw = (!nv || nu == 0) ? 0 : nu;
    if (nuw == 0) rows(std::integral_constant<int, 0>{});
    else if (nuw == 2) rows(std::integral_constant<int, 2>{});
    else rows(std::integral_constant<int, 1>{});
    float* out = ws + a.ws_off + (size_t)(item - a.item0) * (TN * 64);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (u == 0 || wide)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = nb + u * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          out[n * 64 + wk * 32 + r] = acc[u][q];
        }
  }
}

// fold of the rectangular form: one block per output tile [TN][64] (TN = the layer's pad_), splits added in split order
__global__ __launch_bounds__(256) void wgrad_fold_rect_kernel(const MmdWgradLayer* __restrict__ L, int nl, int ntiles, const float* __restrict__ ws) {
  __shared__ int s_li;
  const int tile = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int lo = 0, hi = nl - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L[mid].tile0 <= tile) lo = mid; else hi = mid - 1; }
    s_li = lo;
  }
  __syncthreads();
  const MmdWgradLayer a = L[s_li];
  const int TN = a.pad_, TT = TN * 64;
  const int t = tile - a.tile0;
  const int tn = t / a.ntk, tk = t - tn * a.ntk;
  const int tiles = a.ntn * a.ntk;
  const size_t sstride = (size_t)tiles * TT;
  const float* p = ws + a.ws_off + (size_t)t * TT + tid * 4;
  // two passes of 4096 floats for a 128-wide tile, one for a 64-wide one; eight splits' loads in flight per round, added in split order
  for (int half = 0; half * 4096 < TT; ++half) {
    float4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* ph = p + half * 4096;
    constexpr int FU = 8;
    int sp = 0;
    for (; sp + FU <= a.nsplit; sp += FU) {
      float4 v[FU][4];
#pragma unroll
      for (int f = 0; f < FU; ++f)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[f][u] = mmd_ld4(ph + (size_t)(sp + f) * sstride + u * 1024);
#pragma unroll
      for (int f = 0; f < FU; ++f)
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[u].x += v[f][u].x; s[u].y += v[f][u].y; s[u].z += v[f][u].z; s[u].w += v[f][u].w; }
    }
    for (; sp < a.nsplit; ++sp) {
      const float* q = ph + (size_t)sp * sstride;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = mmd_ld4(q + u * 1024);
        s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = half * 4096 + u * 1024 + tid * 4, n = tn * TN + e / 64, k = tk * 64 + e % 64;
      if (n < a.N && k < a.K) mmd_stg4(a.dw + (size_t)n * a.K + k, s[u]);
    }
  }
}

// one block per output tile: dW[n0.., k0..] = sum over the splits (in split order) of the partial tiles
template <int T>
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const MmdWgradLayer* __restrict__ L, int nl, int ntiles, const float* __restrict__ ws) {
  constexpr int TT = T * T, NU = TT / 1024;
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
  const float* p = ws + a.ws_off + (size_t)t * TT + tid * 4;
  float4 s[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  // The thin high-resolution layers have up to 128 splits per tile (M = 524288 at 4096 rows per item) and one load round per split made
  // their blocks a chain of 128 dependent round trips - the whole fold launch (at the exposed end of the step) waited for them: 63 -> 45 us
  // with FU splits per round (FU * NU loads in flight per thread), added in split order as before (same sums, bit for bit).  Measured no
  // better: four blocks per tile with 16 splits per round (45 us), the tile -> layer search on an LDS copy of the tile0 column (68 us).
  constexpr int FU = NU <= 4 ? 8 : 2;
  const size_t sstride = (size_t)tiles * TT;
  int sp = 0;
  for (; sp + FU <= a.nsplit; sp += FU) {
    float4 v[FU][NU];
#pragma unroll
    for (int f = 0; f < FU; ++f)
#pragma unroll
      for (int u = 0; u < NU; ++u) v[f][u] = mmd_ld4(p + (size_t)(sp + f) * sstride + u * 1024);
#pragma unroll
    for (int f = 0; f < FU; ++f)
#pragma unroll
      for (int u = 0; u < NU; ++u) { s[u].x += v[f][u].x; s[u].y += v[f][u].y; s[u].z += v[f][u].z; s[u].w += v[f][u].w; }
  }
  for (; sp < a.nsplit; ++sp) {
    const float* q = p + (size_t)sp * sstride;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float4 v = mmd_ld4(q + u * 1024);
      s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e = u * 1024 + tid * 4, n = tn * T + e / T, k = tk * T + e % T;
    if (n < a.N && k < a.K) mmd_stg4(a.dw + (size_t)n * a.K + k, s[u]);       // K % 4 == 0: a float4 is all-valid or all-out
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
    const int T = wg_tile();
    if (T == 0) {      // rectangular form: TN x 64 tiles, TN = 128 for N > 64; half the rows per item there (same MFMA time per item)
      static const int rect_min = getenv("MMD_WG_RECT_MIN") ? atoi(getenv("MMD_WG_RECT_MIN")) : 65;
      static const int rect_rows = getenv("MMD_WG_RECT_ROWS") ? atoi(getenv("MMD_WG_RECT_ROWS")) : 0;
      const int TN = a.N >= rect_min ? 128 : 64;
      a.ntn = cdiv(a.N, TN); a.ntk = cdiv(a.K, 64);
      const int rpi = TN == 128 ? max(rect_rows ? rect_rows : rows_per_item / 2, GW_BR) : rows_per_item;
      const int splits = cdiv(a.M, rpi);
      a.mchunk = cdiv(cdiv(a.M, splits), GW_BR) * GW_BR;
      a.nsplit = cdiv(a.M, a.mchunk);
      a.item0 = items; a.tile0 = tiles; a.ws_off = ws; a.pad_ = TN;
      items += a.nsplit * a.ntn * a.ntk;
      tiles += a.ntn * a.ntk;
      ws += (long long)a.nsplit * a.ntn * a.ntk * TN * 64;
      continue;
    }
    a.ntn = cdiv(a.N, T); a.ntk = cdiv(a.K, T);
    int splits = cdiv(a.M, T == 128 ? max(rows_per_item / 4, GW_BR) : rows_per_item);      // same MFMA time per item for either tile
    a.mchunk = cdiv(cdiv(a.M, splits), GW_BR) * GW_BR;
    a.nsplit = cdiv(a.M, a.mchunk);
    a.item0 = items; a.tile0 = tiles; a.ws_off = ws; a.pad_ = T;
    items += a.nsplit * a.ntn * a.ntk;
    tiles += a.ntn * a.ntk;
    ws += (long long)a.nsplit * a.ntn * a.ntk * T * T;
  }
  *n_items = items; *n_tiles = tiles; *ws_floats = ws;
  return MMD_OK;
}

// layers_dev: the planned table in device memory; ws: workspace of ws_floats floats (needs no initialisation).
static int wgrad_grouped_impl(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks,
                              double flops, double bytes, int bf16, hipStream_t stream, int rows32 = 0) {
  if (!layers_dev || n_layers <= 0 || n_layers > GW_MAXL || n_items <= 0 || n_tiles <= 0 || !ws) return MMD_EINVAL;
  if (blocks <= 0) blocks = 1024;
  if (blocks > n_items) blocks = n_items;
  mmd_prof_tag(MMD_FAM_PW_WGRAD, "wgrouped L%lld items%lld tiles%lld b%lld", n_layers, n_items, n_tiles, blocks);
  mmd_prof_begin(MMD_FAM_PW_WGRAD, stream);
  if (wg_tile() == 0) {
    if (bf16 == 0 && rows32 && mmd_split_default() && n_layers <= GW_MAXL3) {
      static const int w3 = getenv("MMD_WG_SPLIT_W3") ? 1 : 0;
      static const int xcd8_env = getenv("MMD_WG_XCD8") ? atoi(getenv("MMD_WG_XCD8")) : 3;      // log2 of the run length (3 = runs of 8, 0 = plain order)
      const int xcd8 = xcd8_env;
      if (w3) hipLaunchKernelGGL(wgrad_grouped_split_kernel<3>, dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws, xcd8);
      else hipLaunchKernelGGL(wgrad_grouped_split_kernel<4>, dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws, xcd8);
    } else if (bf16 == 1) {
      if (rows32) hipLaunchKernelGGL((wgrad_grouped_rect_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
      else hipLaunchKernelGGL((wgrad_grouped_rect_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    } else {
      if (rows32) hipLaunchKernelGGL((wgrad_grouped_rect_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
      else hipLaunchKernelGGL((wgrad_grouped_rect_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    }
    hipLaunchKernelGGL(wgrad_fold_rect_kernel, dim3(n_tiles), dim3(256), 0, stream, layers_dev, n_layers, n_tiles, ws);
  } else if (wg_tile() == 128) {
    if (bf16 == 1) hipLaunchKernelGGL((wgrad_grouped_kernel<128, true>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    else hipLaunchKernelGGL((wgrad_grouped_kernel<128, false>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    hipLaunchKernelGGL(wgrad_fold_kernel<128>, dim3(n_tiles), dim3(256), 0, stream, layers_dev, n_layers, n_tiles, ws);
  } else {
    if (bf16 == 1) hipLaunchKernelGGL((wgrad_grouped_kernel<64, true>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    else hipLaunchKernelGGL((wgrad_grouped_kernel<64, false>), dim3(blocks), dim3(256), 0, stream, layers_dev, n_layers, n_items, ws);
    hipLaunchKernelGGL(wgrad_fold_kernel<64>, dim3(n_tiles), dim3(256), 0, stream, layers_dev, n_layers, n_tiles, ws);
  }
  mmd_prof_end(MMD_FAM_PW_WGRAD, stream, flops, bytes);
  return mmd_check_launch();
}
extern "C" int mmd_wgrad_grouped(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks,
                                 double flops, double bytes, hipStream_t stream) {
  return wgrad_grouped_impl(layers_dev, n_layers, n_items, n_tiles, ws, blocks, flops, bytes, 0, stream);
}
// the same launch with two choices made by the caller.  rows32 = 1: the caller promises that every layer's M is a multiple of 32 - the kernel
// then runs without row clamps / masks and with running operand pointers (round 6).  bf16: 0 = fp32 products (with rows32: the split form -
// six bf16 MFMAs on a three-way exact split of both operands, common.h - unless MMD_MFMA_F32=1), 1 = operands rounded to bf16 (as
// mmd_wgrad_grouped_bf16), 2 = fp32 products on v_mfma_f32_32x32x2_f32 (the per-call form of MMD_MFMA_F32=1)
extern "C" int mmd_wgrad_grouped_form(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks,
                                      double flops, double bytes, int bf16, int rows32, hipStream_t stream) {
  if (bf16 < 0 || bf16 > 2) return MMD_EINVAL;
  return wgrad_grouped_impl(layers_dev, n_layers, n_items, n_tiles, ws, blocks, flops, bytes, bf16, stream, rows32 ? 1 : 0);
}
// precision "bf16": operands rounded to bf16 at the MFMA input (mmd_pwconv_bwd_weight_bf16's arithmetic), same table / workspace / fold
extern "C" int mmd_wgrad_grouped_bf16(const MmdWgradLayer* layers_dev, int n_layers, int n_items, int n_tiles, float* ws, int blocks,
                                      double flops, double bytes, hipStream_t stream) {
  return wgrad_grouped_impl(layers_dev, n_layers, n_items, n_tiles, ws, blocks, flops, bytes, 1, stream);
}
