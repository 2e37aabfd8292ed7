// All-N, K-sliced "slab" GEMM for the 1x1 convs whose A operand carries an ARITHMETIC prologue and whose M is small (round 6).
//
//   Y[M,N] = pro(X)[M,K] * W[N,K]^T   (+ epilogue),   pro = BatchNorm backward (+ swish') of (g, z)            (PRO = 1: expand-conv input gradients)
//                                                     or  swish(x * scale + shift) * gate[image]              (PRO = 0: the student's project convs, live BN)
//
// Why: on the LDS-tiled skinny kernel (pw_gemm.hip, 32 x 64 tiles) these launches re-evaluate their prologue once per 64-wide column tile -
// 2 to 6 times per element - and fp32 MFMA and VALU instructions share a SIMD's issue (profiles/r04_notes.md section 17), so the redundant
// transcendentals are MFMA time: `pw M2048 K1248 N208 f12` ran at 24 TFLOP/s with 12.4 VALU instructions per MFMA ("What's
// missing" 2).  Here a block owns a 32-row slab x ALL N columns (NT32 32-wide MFMA tiles per wave, the four waves split the k of every
// granule) x a SLICE of K, so every (row, k) element is transformed exactly once chip-wide; launches with few row slabs (M = 2048: 64) are
// cut along K, never along N, to fill the chip (64 slabs x 4 slices = 256 blocks), and the slices' partial slabs [slice][M][N] (L2-resident,
// a few MB) are added in slice order by a second, tiny launch that owns the epilogue (a kernel boundary costs ~1.5 us, the in-launch
// last-arriver seam 5 - 13 us at these slab sizes: MI355X_MICROARCH.md, price list rows boundary / splitk-seam) - deterministic, no atomics.
// N > 256 (one launch per step: 2112 -> 352) splits into column chunks of <= 8 tiles; only those re-evaluate the prologue (2x instead of 6x).
//
// LDS: A tile [32][BK + 4] and B tile [NT32 * 32][BK + 4] (k contiguous, 4-float row pad: conflict-free ds_read_b128, as pw_gemm.hip), and a
// per-block table of the slice's per-channel coefficients (BatchNorm-backward a1, a2, a3, mean, shift / live-BN scale, shift), filled once.
// k order inside a step: granule gi (32 k) -> wave w takes k = gi*32 + w*8 + h*4 + {0..3} (h = lane >> 5), so a slice whose length is not a
// multiple of BK skips whole granules in EVERY wave (no idle waves in the tail step).
// Reference op: autograd of the expand / project convs behind train-mode BatchNorm + swish, src/YetAnotherEfficientNet.py:427-447.
#include "common.h"
#include "pw_args.h"
#include <cstdlib>

#define SL_BM 32

// -DMMD_SLSTAMPS (dev build, tools/dev/slab_phases.py): block 0 / thread 0 stamps the 100 MHz wall clock along the kernel
#ifdef MMD_SLSTAMPS
__device__ unsigned long long g_slst[128];
#define SL_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 128) g_slst[i] = wall_clock64(); } while (0)
#define SL_TW(i) do { __builtin_amdgcn_s_waitcnt(0); SL_T(i); } while (0)
extern "C" int mmd_slab_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slst), sizeof(g_slst)) == hipSuccess ? 0 : -1; }
#else
#define SL_T(i)
#define SL_TW(i)
#endif

struct SlabArgs {
  PwArgs p;
  int nchunk;        // column chunks (blocks along N; 1 unless N > 256)
  int nslice;        // K slices (blocks along K)
  int gran;          // 32-wide k granules per slice
  int ktab;          // table row length (floats): the slice's channel count rounded up to BK
  float* part;       // nslice > 1: partial slabs [nslice][nslab * 32][nchunk * NT32 * 32]
  int has_aff;       // PRO 0: the operand has a scale / shift (live BatchNorm or given coefficients)
};

__device__ __forceinline__ float4 sl_mask(bool ok, const float4& v) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// Epilogue of one 32-row x ncols block: bias, BatchNorm sums (of the output, or the BnSumOp form), folded BN / activation, residual, store.
// `ld(rl, c4)` returns the finished dot products of row rl, columns 4 c4 .. 4 c4 + 3 of the block.  NC4 = float4 columns of the block,
// lg = log2 of NC4 rounded up to a power of two (threads are dealt c4 = tid & (2^lg - 1), row group tid >> lg); sRed: (2 * 256 / 2^lg) * NC4 * 4 floats.
template <class LD4>
__device__ __forceinline__ void slab_epilogue(const PwArgs& a, int m0, int n0, int NC4, int lg, LD4 ld, float* sRed) {
  const int tid = threadIdx.x;
  const int c4 = tid & ((1 << lg) - 1), rg = tid >> lg, RG = 256 >> lg;
  const int col = n0 + c4 * 4;
  const bool cok = c4 < NC4 && col < a.N;
  float4 b4 = make_float4(0, 0, 0, 0), osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  float4 xmu = make_float4(0, 0, 0, 0), xis = make_float4(0, 0, 0, 0);
  if (cok) {
    if (a.bias) b4 = mmd_ld4(a.bias + col);
    if (a.out_scale) { osc = mmd_ld4(a.out_scale + col); osh = mmd_ld4(a.out_shift + col); }
    if (a.xs.z) { xmu = mmd_ld4(a.xs.mean + col); xis = mmd_ld4(a.xs.invstd + col); }
  }
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
  if (cok) {
    for (int rl = rg; rl < SL_BM; rl += RG) {
      const int row = m0 + rl;
      if (row >= a.M) break;
      float4 v = ld(rl, c4);
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (!a.xs.z) {
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        q4.x += v.x * v.x; q4.y += v.y * v.y; q4.z += v.z * v.z; q4.w += v.w * v.w;
      }
      if (a.out_scale) { v.x = v.x * osc.x + osh.x; v.y = v.y * osc.y + osh.y; v.z = v.z * osc.z + osh.z; v.w = v.w * osc.w + osh.w; }
      if (a.out_act) { v.x = mmd_act(v.x, a.out_act); v.y = mmd_act(v.y, a.out_act); v.z = mmd_act(v.z, a.out_act); v.w = mmd_act(v.w, a.out_act); }
      const size_t off = (size_t)row * a.N + col;
      if (a.residual) { const float4 rr = mmd_ld4(a.residual + off); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
      mmd_st4(a.y + off, v);
      if (a.xs.z) pw_xs_acc(a.xs, v, off, row, xmu, xis, s4, q4);
    }
  }
  if (a.stats) {
    // fixed-order block sums: row group rg's partial -> sRed[rg][c], then one thread per column adds the RG partials and issues the two
    // f64 atomics (one pair per column and block, as the skinny kernel)
    const int NW = NC4 * 4;
    if (c4 < NC4) {
      *reinterpret_cast<float4*>(&sRed[(size_t)rg * NW + c4 * 4]) = s4;
      *reinterpret_cast<float4*>(&sRed[(size_t)(RG + rg) * NW + c4 * 4]) = q4;
    }
    __syncthreads();
    for (int c = tid; c < NW; c += 256) {
      if (n0 + c < a.N) {
        float s = 0.f, q = 0.f;
        for (int g = 0; g < RG; ++g) { s += sRed[(size_t)g * NW + c]; q += sRed[(size_t)(RG + g) * NW + c]; }
        atomicAdd(&a.stats[n0 + c], (double)s);
        atomicAdd(&a.stats[a.N + n0 + c], (double)q);
      }
    }
  }
}

template <int NT32, int BK, int PRO>
__global__ __launch_bounds__(256) void pw_slab_kernel(SlabArgs sa) {
  constexpr int LD = BK + 4;
  constexpr int F4R = BK / 4;             // float4 per tile row
  constexpr int RSTEP = 256 / F4R;        // rows covered by one pass of the 256 threads: 8 (BK 128) / 16 (BK 64)
  constexpr int NA = SL_BM / RSTEP;       // A rows per thread: 4 / 2
  constexpr int NB = NT32 * 32 / RSTEP;   // B rows per thread
  constexpr int NW = NT32 * 32;           // columns of the block
  constexpr int NG = BK / 32;             // k granules per step
  constexpr int NCO = (PRO == 1) ? 5 : 2; // table rows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                       // [32][LD]
  float* sB = smem + SL_BM * LD;          // [NW][LD]; after the K loop: reduction scratch, finished tile, epilogue sums
  float* sTab = sB + NW * LD;             // [NCO][ktab]
  const PwArgs& a = sa.p;
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  const int slice = t % sa.nslice, rest = t / sa.nslice, chunk = rest % sa.nchunk, slab = rest / sa.nchunk;
  const int m0 = slab * SL_BM, n0 = chunk * NW;
  const int kbeg = slice * sa.gran * 32;
  const int kend = min(a.K, kbeg + sa.gran * 32);
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int kq = (tid % F4R) * 4, lrow = tid / F4R;

  // ---- per-channel coefficients of the slice, once per block
  if constexpr (PRO == 1) {
    if (a.bb.dgamma && t == 0)
      for (int c = tid; c < a.K; c += 256) { a.bb.dgamma[c] += (float)a.bb.sums[a.K + c]; a.bb.dbeta[c] += (float)a.bb.sums[c]; }
    for (int c = kbeg + tid; c < kend; c += 256) {
      float a1, a2, a3, mu, sh;
      bn_bwd_coef(a.bb, c, a1, a2, a3, mu, sh);
      const int j = c - kbeg;
      sTab[j] = a1; sTab[sa.ktab + j] = a2; sTab[2 * sa.ktab + j] = a3; sTab[3 * sa.ktab + j] = mu; sTab[4 * sa.ktab + j] = sh;
    }
  } else {
    if (sa.has_aff)
      for (int c = kbeg + tid; c < kend; c += 256) {
        float sc, sh;
        if (a.in_bn.stats) bn_live_coef(a.in_bn, c, sc, sh);
        else { sc = a.in_scale[c]; sh = a.in_shift[c]; }
        sTab[c - kbeg] = sc; sTab[sa.ktab + c - kbeg] = sh;
      }
  }

  // ---- operand rows of this thread (clamped: rows / columns past the end are computed and never stored)
  const float* xrow[NA]; const float* grow[NA]; float rowsc[NA]; bool rok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = m0 + lrow + i * RSTEP;
    rok[i] = row < a.M;
    const int rr = rok[i] ? row : a.M - 1;
    xrow[i] = a.x + (size_t)rr * a.K;
    if constexpr (PRO == 1) {
      grow[i] = a.bb.z + (size_t)rr * a.K;
      rowsc[i] = a.bb.mul_b ? a.bb.mul_b[rr / a.bb.rows_per_image] : 1.f;
    } else {
      grow[i] = a.gate ? a.gate + (size_t)(rr / a.rows_per_image) * a.K : nullptr;
      rowsc[i] = 1.f;
    }
  }
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int col = n0 + lrow + i * RSTEP;
    wrow[i] = a.w + (size_t)(col < a.N ? col : a.N - 1) * a.K;
  }
  f32x16 acc[NT32];
#pragma unroll
  for (int j = 0; j < NT32; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
  const bool dz_here = PRO == 1 && a.bb.dz_out != nullptr && chunk == 0;
  const bool has_gate = PRO == 0 && a.gate != nullptr;
  const bool swish_in = PRO == 0 && a.in_act == MMD_ACT_SWISH;
  const bool swish_bb = PRO == 1 && a.bb.act == MMD_ACT_SWISH;

  float4 ra[NA], rg4[NA], rb[NB];
  auto gload = [&](int k0) {
    // unconditional loads from clamped (always valid) addresses, masked at the LDS store of a tail step only
    const int k = k0 + kq;
    const int kc = k < kend ? k : kend - 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      ra[i] = mmd_ld4(xrow[i] + kc);
      if constexpr (PRO == 1) rg4[i] = mmd_ld4(grow[i] + kc);
      else if (has_gate) rg4[i] = mmd_ld4(grow[i] + kc);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = mmd_ld4(wrow[i] + kc);
  };
  auto lstore = [&](int k0) {
    const int k = k0 + kq;
    const bool kok = k < kend;
    const int j = (kok ? k : kend - 4) - kbeg;
    if constexpr (PRO == 1) {
      BnBwdCoef4 bq;
      bn_bwd_tab4(sTab, sa.ktab, j, bq);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float4 v = swish_bb ? bn_bwd_eval4(ra[i], rg4[i], rowsc[i], MMD_ACT_SWISH, bq) : bn_bwd_eval4(ra[i], rg4[i], rowsc[i], MMD_ACT_NONE, bq);
        if (dz_here && kok && rok[i]) mmd_st4(a.bb.dz_out + (size_t)(m0 + lrow + i * RSTEP) * a.K + k, v);
        *reinterpret_cast<float4*>(&sA[(lrow + i * RSTEP) * LD + kq]) = sl_mask(kok, v);
      }
    } else {
      float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
      if (sa.has_aff) { sc = *reinterpret_cast<const float4*>(sTab + j); sh = *reinterpret_cast<const float4*>(sTab + sa.ktab + j); }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float4 v = ra[i];
        if (sa.has_aff) { v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w; }
        if (swish_in) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
        if (has_gate) { v.x *= rg4[i].x; v.y *= rg4[i].y; v.z *= rg4[i].z; v.w *= rg4[i].w; }
        *reinterpret_cast<float4*>(&sA[(lrow + i * RSTEP) * LD + kq]) = sl_mask(kok, v);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&sB[(lrow + i * RSTEP) * LD + kq]) = rb[i];
  };
  auto mma = [&](int ng) {      // ng = populated granules of the step (block-uniform)
    const float* pa = &sA[r * LD + wave * 8 + h * 4];
    const float* pb = &sB[r * LD + wave * 8 + h * 4];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi < ng) {
        const float4 av = *reinterpret_cast<const float4*>(pa + gi * 32);
#pragma unroll
        for (int j = 0; j < NT32; ++j) {
          const float4 bv = *reinterpret_cast<const float4*>(pb + j * 32 * LD + gi * 32);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[j], 0, 0, 0);
        }
      }
    }
  };

  SL_T(0);
  gload(kbeg);
  __syncthreads();                                   // the coefficient table is complete
  SL_T(1);
  int sti = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    SL_TW(2 + 5 * sti);                              // (dev build only: the step's loads have landed)
    lstore(k0);
    SL_T(3 + 5 * sti);
    __syncthreads();
    SL_T(4 + 5 * sti);
    if (k0 + BK < kend) gload(k0 + BK);
    mma(min(NG, (kend - k0 + 31) >> 5));
    SL_T(5 + 5 * sti);
    __syncthreads();
    SL_T(6 + 5 * sti);
    ++sti;
  }
  SL_T(60);

  // ---- cross-wave K reduction in the accumulator layout (two rounds, 2 x 32 x NW floats of scratch), then the finished tile row-major
  float* scr = sB;                                    // [2][NT32][16][64]: lane-contiguous, conflict-free
  auto put = [&](int slot) {
#pragma unroll
    for (int j = 0; j < NT32; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) scr[((slot * NT32 + j) * 16 + q) * 64 + lane] = acc[j][q];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int j = 0; j < NT32; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] += scr[((slot * NT32 + j) * 16 + q) * 64 + lane];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  float* tile = sB;                                   // [32][NW]
  if (wave == 0) add(0);
  __syncthreads();                                    // (every wave is past its scratch reads before wave 0 overwrites the scratch)
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < NT32; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        tile[row * NW + j * 32 + r] = acc[j][q];
      }
  }
  __syncthreads();
  SL_T(61);
  constexpr int NC4 = NW / 4;
  constexpr int LG = NC4 <= 16 ? 4 : NC4 <= 32 ? 5 : NC4 <= 64 ? 6 : 7;
  if (sa.nslice > 1) {
    // partial slab of this K slice -> workspace (plain coalesced stores; the combine launch behind the kernel boundary reads them)
    const int NWT = sa.nchunk * NW;
    float* dst = sa.part + ((size_t)slice * (gridDim.x / (sa.nslice * sa.nchunk)) * SL_BM + m0) * NWT + n0;
    for (int i = tid; i < SL_BM * NC4; i += 256) {
      const int rl = i / NC4, c4 = i - rl * NC4;
      mmd_st4(dst + (size_t)rl * NWT + c4 * 4, *reinterpret_cast<const float4*>(&tile[rl * NW + c4 * 4]));
    }
    SL_TW(62);
    return;
  }
  slab_epilogue(a, m0, n0, NC4, LG, [&](int rl, int c4) { return *reinterpret_cast<const float4*>(&tile[rl * NW + c4 * 4]); }, tile + SL_BM * NW);
  SL_TW(62);
}

// Combine launch of a K-sliced slab GEMM: adds the slices' partial slabs in slice order and runs the epilogue.  One block per 32 rows x
// column chunk (the same decomposition, so the BatchNorm-sum atomics per address are what the GEMM launch itself would issue).
__global__ __launch_bounds__(256) void pw_slab_combine_kernel(SlabArgs sa, int nw, int lg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PwArgs& a = sa.p;
  const int chunk = blockIdx.x % sa.nchunk, slab = blockIdx.x / sa.nchunk;
  const int m0 = slab * SL_BM, n0 = chunk * nw;
  const int NWT = sa.nchunk * nw;
  const size_t sstride = (size_t)(gridDim.x / sa.nchunk) * SL_BM * NWT;
  const float* src = sa.part + (size_t)m0 * NWT + n0;
  const int ns = sa.nslice;
  SL_T(64);
  slab_epilogue(a, m0, n0, nw / 4, lg, [&](int rl, int c4) {
    const float* p = src + (size_t)rl * NWT + c4 * 4;
    float4 v = mmd_ld4(p);
    for (int s = 1; s < ns; ++s) { const float4 u = mmd_ld4(p + s * sstride); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    return v;
  }, smem);
  SL_TW(65);
}

typedef void (*SlabKern)(SlabArgs);
template <int BK, int PRO>
static SlabKern slab_pick_nt(int nt32) {
  switch (nt32) {
    case 2: return pw_slab_kernel<2, BK, PRO>;
    case 3: return pw_slab_kernel<3, BK, PRO>;
    case 4: return pw_slab_kernel<4, BK, PRO>;
    case 5: return pw_slab_kernel<5, BK, PRO>;
    case 6: return pw_slab_kernel<6, BK, PRO>;
    case 7: return pw_slab_kernel<7, BK, PRO>;
    case 8: return pw_slab_kernel<8, BK, PRO>;
    default: return nullptr;
  }
}

// Geometry of a launch: -> false when the slab kernel does not cover it
struct SlabPlan { int nt32, nchunk, nslab, nslice, gran, bk, ktab; size_t lds; long long part_floats; };
static bool slab_plan(int M, int K, int N, int pro, SlabPlan& p) {
  if (M <= 0 || (K & 3) || (N & 3) || K < 128 || N < 36) return false;
  const int tiles = cdiv(N, 32);
  p.nchunk = cdiv(tiles, 8);
  p.nt32 = cdiv(tiles, p.nchunk);
  if (p.nt32 < 2) return false;
  p.nslab = cdiv(M, SL_BM);
  const int blocks = p.nslab * p.nchunk;
  const int gtot = cdiv(K, 32);
  static const int target = getenv("MMD_SLAB_BLOCKS") ? atoi(getenv("MMD_SLAB_BLOCKS")) : 256;
  // K slices: fill the chip (about one block per CU), at least 4 granules (128 k) per slice
  int ns = blocks >= target ? 1 : target / blocks;
  ns = max(1, min(ns, gtot / 4));
  p.gran = cdiv(gtot, ns);
  p.nslice = cdiv(gtot, p.gran);
  const int nco = pro == 1 ? 5 : 2;
  // BK = 128 where the tiles + the coefficient table fit (one block per CU either way), BK = 64 otherwise
  static const int bk_env = getenv("MMD_SLAB_BK") ? atoi(getenv("MMD_SLAB_BK")) : 0;
  for (int bk = (bk_env == 64 ? 64 : 128); bk >= 64; bk -= 64) {
    p.bk = bk;
    p.ktab = cdiv(p.gran * 32, bk) * bk;
    p.lds = ((size_t)(SL_BM + p.nt32 * 32) * (bk + 4) + (size_t)nco * p.ktab) * sizeof(float);
    // after the K loop the B tile's space holds: reduction scratch 2 x 32 x NW floats, then the finished tile 32 x NW + the epilogue's
    // sums (2 x RG x NW <= 16 x NW with RG <= 8): 48 x NW floats - both within NW x (BK + 4)
    if (p.lds <= 158 * 1024) break;
    if (bk == 64) return false;
  }
  p.part_floats = p.nslice > 1 ? (long long)p.nslice * p.nslab * SL_BM * p.nchunk * p.nt32 * 32 : 0;
  return true;
}

// workspace floats a launch of this shape needs for its K slices (0: it runs unsliced, or the slab kernel does not take the shape)
extern "C" int mmd_pwconv_slab_ws_floats(int M, int K, int N, int bn_operand) {
  SlabPlan p;
  if (!slab_plan(M, K, N, bn_operand ? 1 : 0, p)) return 0;
  return p.part_floats > 0x7fffffffLL ? 0 : (int)p.part_floats;      // (a sliced launch has < 256 row slabs x <= 8 slices: a few M floats)
}

// -> 1 when the launch was taken, 0 when the shape / operand mode is not covered (the caller then uses the LDS-tiled kernels), < 0 on error.
// ws / ws_floats: the caller's workspace for the K slices' partial slabs (a.form == MMD_PW_FORM_SLAB forces every supported launch here;
// auto mode: the launches that would run the skinny kernel with an arithmetic prologue)
int pw_slab_try(PwArgs& a, float* ws, long long ws_floats, bool auto_ok, hipStream_t stream) {
  static const int off = getenv("MMD_NO_SLAB") ? 1 : 0;
  if (off && a.form != MMD_PW_FORM_SLAB) return 0;
  if (a.form != MMD_PW_FORM_AUTO && a.form != MMD_PW_FORM_SLAB) return 0;
  if (a.form == MMD_PW_FORM_AUTO && !auto_ok) return 0;
  if (a.bf16 || a.st.Cin || a.pyr.n || a.y_batch_stride || a.p5.z || a.stats_ws || a.g_images) return 0;
  if (a.x16 || a.y16 || a.z16 || a.dz16 || a.p5z16) return 0;
  if (a.in_act != MMD_ACT_NONE && a.in_act != MMD_ACT_SWISH) return 0;
  if (a.bb.z && a.bb.act != MMD_ACT_NONE && a.bb.act != MMD_ACT_SWISH) return 0;
  const int pro = a.bb.z ? 1 : 0;
  if (!pro && a.gate && a.rows_per_image <= 0) return 0;
  SlabPlan p;
  if (!slab_plan(a.M, a.K, a.N, pro, p)) return 0;
  if (p.nslice > 1 && (!ws || ws_floats < p.part_floats)) {
    if (a.form == MMD_PW_FORM_SLAB && !ws) return MMD_EINVAL;       // forced without a workspace: an error, not a silent other kernel
    return 0;
  }
  SlabArgs sa{};
  sa.p = a; sa.nchunk = p.nchunk; sa.nslice = p.nslice; sa.gran = p.gran; sa.ktab = p.ktab; sa.part = ws;
  sa.has_aff = (!pro && (a.in_scale || a.in_bn.stats)) ? 1 : 0;
  SlabKern kern = p.bk == 128 ? (pro ? slab_pick_nt<128, 1>(p.nt32) : slab_pick_nt<128, 0>(p.nt32))
                              : (pro ? slab_pick_nt<64, 1>(p.nt32) : slab_pick_nt<64, 0>(p.nt32));
  if (!kern) return 0;
  {       // raise the kernel's dynamic-LDS limit once per instantiation (at most 28 of them)
    static SlabKern done[32]; static int ndone = 0;
    bool seen = false;
    for (int i = 0; i < ndone; ++i) seen |= done[i] == kern;
    if (!seen) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MMD_ELAUNCH;
      if (ndone < 32) done[ndone++] = kern;
    }
  }
  const int nblk = p.nslab * p.nchunk * p.nslice;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), p.lds, stream, sa);
  if (p.nslice > 1) {
    const int nw = p.nt32 * 32, nc4 = nw / 4;
    const int lg = nc4 <= 16 ? 4 : nc4 <= 32 ? 5 : nc4 <= 64 ? 6 : 7;
    const size_t lds = (size_t)2 * (256 >> lg) * nw * sizeof(float);
    hipLaunchKernelGGL(pw_slab_combine_kernel, dim3(p.nslab * p.nchunk), dim3(256), lds, stream, sa, nw, lg);
  }
  return 1;
}
