// Backward of an MBConv EXPAND conv (1x1, thin input) with its train-mode BatchNorm-0 + swish, in ONE kernel — CDNA4 / gfx950 (round 4).
//
// Reference ops: autograd of `_expand_conv` -> `_bn0` -> swish in MBConvBlock.forward, src/YetAnotherEfficientNet.py:456-460, for the
// high-resolution blocks of the trainable net (D2: 16 -> 96 at 256^2, 24 -> 144 at 128^2; D4: 24 -> 144, 32 -> 192).
//
// The GEMM-shaped path runs two launches over the 6x expanded gradient: the input-gradient GEMM evaluates dz0 = BnBwd(g0, z0) in its
// operand prologue AND stores it (one [M, C] write), so that the weight-gradient GEMM - grouped with the other layers' into a launch that
// runs after the backward's last kernel - can read it back (one [M, C] read, plus x again).  For these layers the wide tensor is the whole
// cost (16 -> 96 at 256^2, B = 8: 201 MB per pass) and the products are tiny (dW is 96 x 16).  Here one pass over (g0, z0):
//   dz0 = a1 g0 swish'(z0 a1 + sh) + a2 (z0 - mu) + a3          BatchNorm-0 + swish backward, evaluated once per element -> LDS tile [64][C]
//   dx  = dz0 . W          (+ residual; may be in place)        v_mfma_f32_16x16x4_f32, A = dz0 rows from the tile, B = W[C, Cin] resident in LDS
//   dW += dz0^T . x                                             same tile read with rows as the reduction; accumulators live in registers
//                                                               across the block's row tiles (persistent grid), C x Cin atomics per block at the end
//   xs  = [sum dx' , sum dx' xhat']                             optional: dx completes the gradient of the PREVIOUS block's BatchNorm-2 output
//                                                               (BnSumOp of pw_args.h: "the last writer takes the sums")
// dz0 never reaches HBM: 2 wide passes instead of 4 (read g0, z0 | write dz0 | read dz0), and the layer leaves the grouped weight-gradient
// launch at the exposed end of the backward.
#include "common.h"
#include <cstdlib>

struct MbwArgs {
  const float* g0; const float* z0; const float* x; const float* w;
  float* dx; const float* residual; float* dw;
  const float* scale; const float* shift; const float* mean; const float* invstd; const double* sums; double inv_count;
  float* dgamma; float* dbeta;
  const float* xs_z; const float* xs_mean; const float* xs_invstd; const float* xs_mul_b; int xs_rpi; double* xs_sums;
  int M, ntiles;
};

// TM = rows per tile: 64, or 32 where the larger tile leaves only two blocks per CU (LDS, prefetch registers)
template <int C, int CIN, int MBW_TM>
struct MbwCfg {
  // LDS row strides, from the banking of the instruction that reads each tile (MI355X_MICROARCH.md, LDS table; round 5 - the earlier strides
  // assumed contiguous 16-lane groups and 64 banks for every read: 17 - 34 % SQ_LDS_BANK_CONFLICT):
  //   ds_read_b128 (GEMM 1's A): 64 banks, 16-lane groups {0-3, 12-15, 20-27}, ... = rows {0-3, 12-15} of k group g with rows {4-11} of g + 1:
  //     stride = 8 (mod 16) floats puts them on 16 different 16-B slots;
  //   ds_read_b32 (GEMM 1's B, GEMM 2's A and B): 32 banks, 32-lane halves = two k groups x 16 consecutive floats: the two groups' rows must
  //     be 16 (mod 32) floats apart - W rows 4 apart: stride = 4 (mod 8); dz0 / x rows: the half-wave's groups take rows 2 apart (GEMM 2's
  //     k assignment below), and 2 x stride = 16 (mod 32) is the same stride = 8 (mod 16).
  static constexpr int LD = C + 8;
  static constexpr int LDW = CIN + 4;
  static constexpr int LDX = (CIN % 16 == 8) ? CIN : CIN + 8;
  static constexpr int LDO = CIN + 4;                                       // dx staging
  static constexpr int NT = (CIN + 15) / 16;                                // 16-column tiles of dx / dW
  static constexpr int MT = C / 16;                                         // 16-row tiles of dW
  static constexpr int TPW = (MT + 3) / 4;                                  // dW row tiles per wave
  static constexpr int NQ = C / 4, RG = 256 / NQ;                           // staging: channel quads x row groups
  static constexpr int NQX = CIN / 4, RGX = 256 / NQX;                      // epilogue: dx quads x row groups
  static constexpr size_t lds = (size_t)(MBW_TM * LD + MBW_TM * LDX + C * LDW + MBW_TM * LDO + 2 * CIN) * sizeof(float);
};

template <int C, int CIN, int MBW_TM>
__global__ __launch_bounds__(256, (C >= 288 ? 1 : 2)) void mbconv_expand_bwd_kernel(MbwArgs a) {      // (C = 288: 112 KB of LDS, one block per CU - no register cap)
  using Cf = MbwCfg<C, CIN, MBW_TM>;
  constexpr int LD = Cf::LD, LDW = Cf::LDW, LDX = Cf::LDX, LDO = Cf::LDO, NT = Cf::NT, MT = Cf::MT, TPW = Cf::TPW;
  constexpr int NQ = Cf::NQ, RG = Cf::RG, NQX = Cf::NQX, RGX = Cf::RGX;
  static_assert(C % 16 == 0 && CIN % 8 == 0 && CIN >= 16 && CIN <= 48, "geometry");
  extern __shared__ float smem[];
  float* const sDz = smem;                          // [64][LD]
  float* const sX = sDz + MBW_TM * LD;              // [64][LDX]
  float* const sW = sX + MBW_TM * LDX;              // [C][LDW]
  float* const sO = sW + C * LDW;                   // [64][LDO]
  float* const sXs = sO + MBW_TM * LDO;             // [2][CIN] block sums of the upstream BatchNorm backward
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- once per block: W -> LDS, BatchNorm-backward coefficients of this thread's channel quad -> registers
  for (int i = tid; i < C * (CIN / 4); i += 256) {
    const int c = i / (CIN / 4), k4 = i - c * (CIN / 4);
    *reinterpret_cast<float4*>(&sW[c * LDW + k4 * 4]) = mmd_ld4(a.w + (size_t)c * CIN + k4 * 4);
  }
  if (tid < 2 * CIN) sXs[tid] = 0.f;
  const bool stager = tid < NQ * RG;
  const int sq = tid % NQ, srg = tid / NQ;          // staging role: channel quad, first row
  float4 a1, a2, a3, mu, sh;
  {
    const int c = sq * 4;
    float v[5][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float m1 = (float)(a.sums[c + j] * a.inv_count), m2 = (float)(a.sums[C + c + j] * a.inv_count);
      const float is = a.invstd[c + j], sc = a.scale[c + j];
      v[0][j] = sc; v[1][j] = -sc * is * m2; v[2][j] = -sc * m1; v[3][j] = a.mean[c + j]; v[4][j] = a.shift[c + j];
    }
    a1 = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]); a2 = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
    a3 = make_float4(v[2][0], v[2][1], v[2][2], v[2][3]); mu = make_float4(v[3][0], v[3][1], v[3][2], v[3][3]);
    sh = make_float4(v[4][0], v[4][1], v[4][2], v[4][3]);
  }
  if (a.dgamma && blockIdx.x == 0)                  // dgamma / dbeta of BatchNorm-0 from the reduce pass' sums
    for (int c = tid; c < C; c += 256) { a.dgamma[c] += (float)a.sums[C + c]; a.dbeta[c] += (float)a.sums[c]; }
  const bool epi = tid < NQX * RGX;
  const int eq = tid % NQX, erg = tid / NQX;        // epilogue role: dx column quad, first row
  float4 xmu = make_float4(0, 0, 0, 0), xis = xmu, xs4 = xmu, xq4 = xmu;
  if (a.xs_z && epi) { xmu = mmd_ld4(a.xs_mean + eq * 4); xis = mmd_ld4(a.xs_invstd + eq * 4); }
  f32x4 accW[TPW][NT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) accW[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  // software pipeline: the NEXT tile's (g0, z0, x) loads are issued before the current tile's MFMA phases and consumed after them - with two
  // blocks per CU (LDS) nothing else hides a tile's first-load latency (first version, loads at the top of the tile: 2.0 TB/s)
  constexpr int NR = (MBW_TM + RG - 1) / RG;            // rows per staging thread
  constexpr int NXI = (MBW_TM * NQX + 255) / 256;       // x quads per thread
  float4 pg[NR], pz[NR], px[NXI];
  auto prefetch = [&](int tile) {
    const int m0 = tile * MBW_TM;
    if (stager) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int rl = srg + i * RG, row = m0 + rl;
        pg[i] = make_float4(0, 0, 0, 0); pz[i] = make_float4(0, 0, 0, 0);
        if (rl < MBW_TM && row < a.M) {
          const size_t off = (size_t)row * C + sq * 4;
          pg[i] = mmd_ld4(a.g0 + off); pz[i] = mmd_ld4(a.z0 + off);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int it = tid + i * 256, rl = it / NQX, q = it - rl * NQX, row = m0 + rl;
      px[i] = (it < MBW_TM * NQX && row < a.M) ? mmd_ld4(a.x + (size_t)row * CIN + q * 4) : make_float4(0, 0, 0, 0);
    }
  };
  if ((int)blockIdx.x < a.ntiles) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * MBW_TM;
    __syncthreads();                                // the previous tile's readers are done (and W / sXs are staged on the first pass)
    // ---- stage: dz0 tile (evaluated from the prefetched registers), x tile
    if (stager) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int rl = srg + i * RG, row = m0 + rl;
        if (rl < MBW_TM) {
          const float4 gq = pg[i], zq = pz[i];
          float4 d = make_float4(0, 0, 0, 0);
          if (row < a.M) {
            d.x = a1.x * (gq.x * mmd_swish_grad(zq.x * a1.x + sh.x)) + a2.x * (zq.x - mu.x) + a3.x;
            d.y = a1.y * (gq.y * mmd_swish_grad(zq.y * a1.y + sh.y)) + a2.y * (zq.y - mu.y) + a3.y;
            d.z = a1.z * (gq.z * mmd_swish_grad(zq.z * a1.z + sh.z)) + a2.z * (zq.z - mu.z) + a3.z;
            d.w = a1.w * (gq.w * mmd_swish_grad(zq.w * a1.w + sh.w)) + a2.w * (zq.w - mu.w) + a3.w;
          }
          *reinterpret_cast<float4*>(&sDz[rl * LD + sq * 4]) = d;      // rows past M: zeros (they multiply into dW)
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int it = tid + i * 256, rl = it / NQX, q = it - rl * NQX;
      if (it < MBW_TM * NQX) *reinterpret_cast<float4*>(&sX[rl * LDX + q * 4]) = px[i];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) prefetch(tile + gridDim.x);      // in flight during this tile's MFMA phases and epilogue
    // the epilogue's own operands (residual, the upstream BatchNorm's z) are fetched now as well: their latency hides behind the MFMA phases
    constexpr int NRX = (MBW_TM + RGX - 1) / RGX;
    float4 er[NRX], ez[NRX];
    if (epi) {
#pragma unroll
      for (int i = 0; i < NRX; ++i) {
        const int rl = erg + i * RGX, row = m0 + rl;
        er[i] = make_float4(0, 0, 0, 0); ez[i] = make_float4(0, 0, 0, 0);
        if (rl < MBW_TM && row < a.M) {
          const size_t off = (size_t)row * CIN + eq * 4;
          if (a.residual) er[i] = mmd_ld4(a.residual + off);
          if (a.xs_z) ez[i] = mmd_ld4(a.xs_z + off);
        }
      }
    }
    // ---- GEMM 1: dx[TM][CIN] = dz0 . W as (TM / 16) x NT output tiles dealt over the waves   (k = channels: lane group g owns channels
    // 16 kk + 4 g + j of both operands)
    for (int job = wave; job < (MBW_TM / 16) * NT; job += 4) {
      const int rt = job % (MBW_TM / 16), u = job / (MBW_TM / 16);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* ap = &sDz[(rt * 16 + r) * LD + 4 * g];
      const float* bp = &sW[(4 * g) * LDW + u * 16 + r];
      const bool cok = CIN % 16 == 0 || u * 16 + r < CIN;
#pragma unroll 3
      for (int kk = 0; kk < C / 16; ++kk) {
        const float4 av = *reinterpret_cast<const float4*>(ap + kk * 16);
        const float b0 = cok ? bp[(kk * 16 + 0) * LDW] : 0.f, b1 = cok ? bp[(kk * 16 + 1) * LDW] : 0.f;
        const float b2 = cok ? bp[(kk * 16 + 2) * LDW] : 0.f, b3 = cok ? bp[(kk * 16 + 3) * LDW] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc1, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc1, 0, 0, 0);
      }
      acc += acc1;
      if (cok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sO[(rt * 16 + 4 * g + i) * LDO + u * 16 + r] = acc[i];
      }
    }
    // ---- GEMM 2: dW[16 t + ..][CIN] += dz0^T . x   (k = the tile's 64 rows: lane group g owns row 4 kk + pg of both operands, pg = 0, 2, 1, 3:
    // the two groups of a half-wave read rows two apart - MbwCfg)
    {
      const int pg = ((g & 1) << 1) | (g >> 1);
#pragma unroll 4
      for (int kk = 0; kk < MBW_TM / 4; ++kk) {
        const float* arow = &sDz[(4 * kk + pg) * LD + r];
        const float* brow = &sX[(4 * kk + pg) * LDX + r];
        float b[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) b[u] = (CIN % 16 == 0 || u * 16 + r < CIN) ? brow[u * 16] : 0.f;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int mt = wave + 4 * t;
          if (mt < MT) {                            // wave-uniform
            const float av = arow[mt * 16];
#pragma unroll
            for (int u = 0; u < NT; ++u) accW[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u], accW[t][u], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    // ---- dx epilogue, row-major: residual, store, upstream BatchNorm-backward sums
    if (epi) {
#pragma unroll
      for (int i = 0; i < NRX; ++i) {
        const int rl = erg + i * RGX, row = m0 + rl;
        if (rl < MBW_TM && row < a.M) {
          float4 v = *reinterpret_cast<const float4*>(&sO[rl * LDO + eq * 4]);
          const size_t off = (size_t)row * CIN + eq * 4;
          if (a.residual) { v.x += er[i].x; v.y += er[i].y; v.z += er[i].z; v.w += er[i].w; }
          mmd_st4(a.dx + off, v);
          if (a.xs_z) {
            const float4 zz = ez[i];
            if (a.xs_mul_b) { const float rs = a.xs_mul_b[row / a.xs_rpi]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
            xs4.x += v.x; xs4.y += v.y; xs4.z += v.z; xs4.w += v.w;
            xq4.x += v.x * (zz.x - xmu.x) * xis.x; xq4.y += v.y * (zz.y - xmu.y) * xis.y;
            xq4.z += v.z * (zz.z - xmu.z) * xis.z; xq4.w += v.w * (zz.w - xmu.w) * xis.w;
          }
        }
      }
    }
  }
  // ---- once per block: dW partials and the upstream sums
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int mt = wave + 4 * t;
    if (mt < MT) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = mt * 16 + 4 * g + i, k = u * 16 + r;
          if (CIN % 16 == 0 || k < CIN) atomicAdd(&a.dw[(size_t)c * CIN + k], accW[t][u][i]);
        }
    }
  }
  if (a.xs_z) {
    if (epi) {
      float* p = &sXs[eq * 4];
      atomicAdd(p, xs4.x); atomicAdd(p + 1, xs4.y); atomicAdd(p + 2, xs4.z); atomicAdd(p + 3, xs4.w);
      p += CIN;
      atomicAdd(p, xq4.x); atomicAdd(p + 1, xq4.y); atomicAdd(p + 2, xq4.z); atomicAdd(p + 3, xq4.w);
    }
    __syncthreads();
    if (tid < 2 * CIN) atomicAdd(&a.xs_sums[tid], (double)sXs[tid]);
  }
}

// (Cin, Cmid) pairs with a kernel instantiation
static bool mbw_has_kernel(int Cin, int Cmid) {
  return (Cin == 16 && Cmid == 96) || (Cin == 24 && Cmid == 144) || (Cin == 32 && Cmid == 192) || (Cin == 48 && Cmid == 288);
}
// 1 when mmd_mbconv_expand_bwd_fused has a kernel for (Cin, Cmid) AND the single pass measured faster than the two GEMM launches it
// replaces.  (48, 288) - the 64^2 blocks of D2, blocks 6 - 8, M = 32768 at B = 8 - has a kernel (round 6, VERDICT r5 item 3: 32-row tiles,
// W [288][52] alone is 60 KB of LDS, one block per CU) but does NOT pay: 88 us per launch against 40 us for the input-gradient GEMM + ~15 us
// of the grouped weight-gradient launch, step 14.17 vs 13.98 ms in alternating runs (profiles/r06_notes.md); MMD_MBW48=1 switches it on.
extern "C" int mmd_mbconv_expand_bwd_supported(int Cin, int Cmid) {
  static const int on48 = getenv("MMD_MBW48") ? 1 : 0;
  if (Cin == 48 && Cmid == 288) return on48;
  return mbw_has_kernel(Cin, Cmid) ? 1 : 0;
}

template <int C, int CIN, int TM>
static int mbw_launch(MbwArgs& a, hipStream_t st, int gdef) {
  using Cf = MbwCfg<C, CIN, TM>;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)mbconv_expand_bwd_kernel<C, CIN, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  static const int gmax = getenv("MMD_MBW_GRID") ? atoi(getenv("MMD_MBW_GRID")) : 0;      // persistent over the row tiles
  a.ntiles = cdiv(a.M, TM);
  const int cap = gmax > 0 ? gmax : gdef;
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  hipLaunchKernelGGL((mbconv_expand_bwd_kernel<C, CIN, TM>), dim3(grid), dim3(256), Cf::lds, st, a);
  return mmd_check_launch();
}

// dx[M, Cin] = BnBwd0(g0, z0)[M, Cmid] . w[Cmid, Cin] (+ residual, which may be dx itself);  dw[Cmid, Cin] += BnBwd0(g0, z0)^T . x[M, Cin];
// dgamma / dbeta (+)= the BatchNorm-0 sums;  optional xs_*: dx is the complete gradient w.r.t. BN'(xs_z) * xs_mul_b[image] (+ skip) and
// xs_sums[2 Cin] (+)= [sum g', sum g' xhat'] as in mmd_pwconv_bwd_data_bn2.  BnBwd0: train-mode BatchNorm + swish backward with
// (scale, shift, mean, invstd) of the forward and sums = [sum g', sum g' xhat] of the reduce pass over `count` rows.
extern "C" int mmd_mbconv_expand_bwd_fused(const float* g0, const float* z0, const float* x, const float* w, float* dx, const float* residual,
                                           float* dw, int M, int Cin, int Cmid, const float* scale, const float* shift, const float* mean,
                                           const float* invstd, const double* sums, long long count, float* dgamma, float* dbeta,
                                           const float* xs_z, const float* xs_mean, const float* xs_invstd, const float* xs_mul_b,
                                           int xs_rows_per_image, double* xs_sums, hipStream_t stream) {
  if (!g0 || !z0 || !x || !w || !dx || !dw || M <= 0 || !scale || !shift || !mean || !invstd || !sums || count <= 0) return MMD_EINVAL;
  if (!mbw_has_kernel(Cin, Cmid) || (dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  if (xs_z && (!xs_mean || !xs_invstd || !xs_sums || (xs_mul_b && xs_rows_per_image <= 0))) return MMD_EINVAL;
  MbwArgs a{};
  a.g0 = g0; a.z0 = z0; a.x = x; a.w = w; a.dx = dx; a.residual = residual; a.dw = dw;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.sums = sums; a.inv_count = 1.0 / (double)count;
  a.dgamma = dgamma; a.dbeta = dbeta;
  a.xs_z = xs_z; a.xs_mean = xs_mean; a.xs_invstd = xs_invstd; a.xs_mul_b = xs_mul_b; a.xs_rpi = xs_rows_per_image > 0 ? xs_rows_per_image : 1;
  a.xs_sums = xs_sums;
  a.M = M;
  mmd_prof_tag(MMD_FAM_PW, "mbw M%lld K%lld N%lld f%lld", M, Cmid, Cin, (residual ? 8 : 0) | (xs_z ? 4 : 0));
  mmd_prof_begin(MMD_FAM_PW, stream);
  int rc;
  // 64-row tiles (two blocks per CU at C = 144) measured faster than 32-row ones (three blocks): 88 vs 99.9 us per launch in the graph
  static const int tm32 = getenv("MMD_MBW_TM32") ? 1 : 0;
  if (Cin == 16) rc = mbw_launch<96, 16, 64>(a, stream, 768);                    // 43 KB of LDS: three blocks per CU
  else if (Cin == 24) rc = tm32 ? mbw_launch<144, 24, 32>(a, stream, 768) : mbw_launch<144, 24, 64>(a, stream, 512);
  else if (Cin == 32) rc = tm32 ? mbw_launch<192, 32, 32>(a, stream, 512) : mbw_launch<192, 32, 64>(a, stream, 256);
  else rc = mbw_launch<288, 48, 32>(a, stream, 256);                                  // 112 KB of LDS: one block per CU
  // two GEMMs' worth of products; bytes: g0, z0 read, x read, dx written (+ residual, + the sums' z)
  mmd_prof_end(MMD_FAM_PW, stream, 4.0 * M * (double)Cmid * Cin, 4.0 * M * (2.0 * Cmid + Cin * (2.0 + (residual ? 1 : 0) + (xs_z ? 1 : 0))));
  return rc;
}
