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
#include <type_traits>

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

// Epilogue of one BR-row x ncols block: bias, BatchNorm sums (of the output, or the BnSumOp form), folded BN / activation, residual, store.
// `ld(rl, c4)` returns the finished dot products of row rl, columns 4 c4 .. 4 c4 + 3 of the block.  NC4 = float4 columns of the block,
// lg = log2 of NC4 rounded up to a power of two (threads are dealt c4 = tid & (2^lg - 1), row group tid >> lg; R = BR / (256 >> lg) rows per
// thread); sRed: (2 * 256 / 2^lg) * NC4 * 4 floats.  Every global load of the thread's rows (the operand through `ld`, residual, the z of the
// BnSumOp sums) is issued before the first use: the first version walked its rows one dependent round trip at a time (5.8 us per block).
template <int BR, int R, class LD4>
__device__ __forceinline__ void slab_epilogue(const PwArgs& a, int m0, int n0, int NC4, int lg, LD4 ld, float* sRed) {
  const int tid = threadIdx.x;
  const int c4 = tid & ((1 << lg) - 1), rg = tid >> lg, RG = 256 >> lg;
  const int col = n0 + c4 * 4;
  const bool cok = c4 < NC4 && col < a.N;
  float4 b4 = make_float4(0, 0, 0, 0), osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  float4 xmu = make_float4(0, 0, 0, 0), xis = make_float4(0, 0, 0, 0);
  const int cc = cok ? col : 0;
  if (a.bias) b4 = mmd_ld4(a.bias + cc);
  if (a.out_scale) { osc = mmd_ld4(a.out_scale + cc); osh = mmd_ld4(a.out_shift + cc); }
  if (a.xs.z) { xmu = mmd_ld4(a.xs.mean + cc); xis = mmd_ld4(a.xs.invstd + cc); }
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
  float4 v[R], rr[R], zz[R];
  float rs[R];
  // clamped (always valid) addresses, masked afterwards: guarded loads compile to one dependent round trip each
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int rl = rg + i * RG, row = min(m0 + rl, a.M - 1);
    const size_t off = (size_t)row * a.N + cc;
    v[i] = ld(rl, cok ? c4 : 0);
    rr[i] = make_float4(0, 0, 0, 0); zz[i] = rr[i]; rs[i] = 1.f;
    if (a.residual) rr[i] = mmd_ld4(a.residual + off);
    if (a.xs.z) { zz[i] = mmd_ld4(a.xs.z + off); if (a.xs.mul_b) rs[i] = a.xs.mul_b[row / a.xs.rows_per_image]; }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int rl = rg + i * RG, row = m0 + rl;
    if (cok && row < a.M) {
      float4 u = v[i];
      u.x += b4.x; u.y += b4.y; u.z += b4.z; u.w += b4.w;
      if (!a.xs.z) {
        s4.x += u.x; s4.y += u.y; s4.z += u.z; s4.w += u.w;
        q4.x += u.x * u.x; q4.y += u.y * u.y; q4.z += u.z * u.z; q4.w += u.w * u.w;
      }
      if (a.out_scale) { u.x = u.x * osc.x + osh.x; u.y = u.y * osc.y + osh.y; u.z = u.z * osc.z + osh.z; u.w = u.w * osc.w + osh.w; }
      if (a.out_act) mmd_act4(u, a.out_act);
      if (a.residual) { u.x += rr[i].x; u.y += rr[i].y; u.z += rr[i].z; u.w += rr[i].w; }
      mmd_st4(a.y + (size_t)row * a.N + col, u);
      if (a.xs.z) {      // (pw_xs_acc's arithmetic on the prefetched z)
        const float4 g = make_float4(u.x * rs[i], u.y * rs[i], u.z * rs[i], u.w * rs[i]);
        s4.x += g.x; s4.y += g.y; s4.z += g.z; s4.w += g.w;
        q4.x += g.x * (zz[i].x - xmu.x) * xis.x; q4.y += g.y * (zz[i].y - xmu.y) * xis.y;
        q4.z += g.z * (zz[i].z - xmu.z) * xis.z; q4.w += g.w * (zz[i].w - xmu.w) * xis.w;
      }
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

// DZK: the BatchNorm-backward launches' stored dz - 0 = none, 1 = every row slab whole (M % 32 == 0: stored without a per-lane guard), 2 = guarded
template <int NT32, int PRO, int DZK>
__global__ __launch_bounds__(256) void pw_slab_kernel(SlabArgs sa) {
  constexpr int NW = NT32 * 32;           // columns of the block
  constexpr int NCO = (PRO == 1) ? 5 : 2; // table rows
  constexpr int GLD = 36;                 // floats per staged row: a granule's 32 k + 4 pad (conflict-free ds_read_b128 of the MFMA fragments)
  constexpr int WREG = (32 + NW) * GLD;   // floats of one wave's private staging region: A [32][36] | B [NW][36]
  constexpr int NBI = NW / 8;             // B load instructions per granule (8 rows x 128 bytes each)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PwArgs& a = sa.p;
  float* sTab = smem;                     // [NCO][ktab]
  float* stage = smem + NCO * sa.ktab;    // [4 waves][WREG]; after the K loop: reduction scratch | finished tile | epilogue sums
  float* scr = stage;                     // [NT32][3][4][64] float4: the non-owners' accumulators of every tile
  float* tile = scr + NT32 * 3 * 1024;    // [32][NW] finished dot products, row-major
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  const int slice = t % sa.nslice, rest = t / sa.nslice, chunk = rest % sa.nchunk, slab = rest / sa.nchunk;
  const int m0 = slab * SL_BM, n0 = chunk * NW;
  const int kbeg = slice * sa.gran * 32;
  const int kend = min(a.K, kbeg + sa.gran * 32);
  const int G = (kend - kbeg + 31) >> 5;  // 32-wide k granules of the slice; wave w takes granules w, w + 4, ...
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const wA = stage + (size_t)wave * WREG;      // this wave's staged A granule [32][GLD]
  float* const wB = wA + 32 * GLD;                    // ... and B granule [NW][GLD]
  SL_T(0);

  // ---- A wave owns whole granules (32 k = one 128-byte line per row) of the slice and stages them through its OWN LDS region: loads are
  // row-contiguous (lane = 16-byte piece p of row q + 8 it: eight lanes cover a row's line - fully coalesced; loading straight in the MFMA
  // operand layout, one row per lane, ran at one lane per cycle in the texture addresser and bound the loop), the prologue runs on the
  // coalesced registers (every (row, k) element of the launch is evaluated by exactly one lane), the MFMA fragments are read back from
  // LDS.  Nothing is shared between waves, so the K loop has no barrier; rows / columns past the end are clamped: computed, never stored.
  const int p = lane & 7, q = lane >> 3;
  const float* xrow[4]; const float* grow[4]; float rowsc[4]; bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + q + 8 * i;
    rok[i] = row < a.M;
    const int rr = rok[i] ? row : a.M - 1;
    xrow[i] = a.x + (size_t)rr * a.K;
    grow[i] = nullptr; rowsc[i] = 1.f;
    if constexpr (PRO == 1) {
      grow[i] = a.bb.z + (size_t)rr * a.K;
      rowsc[i] = a.bb.mul_b ? a.bb.mul_b[rr / a.bb.rows_per_image] : 1.f;
    } else {
      if (a.gate) grow[i] = a.gate + (size_t)(rr / a.rows_per_image) * a.K;
    }
  }
  int wofs[NBI];                                     // B row q + 8 it (clamped to the last column), as a 32-bit element offset
#pragma unroll
  for (int it = 0; it < NBI; ++it) wofs[it] = min(n0 + q + 8 * it, a.N - 1) * a.K;
  const bool dz_here = PRO == 1 && a.bb.dz_out != nullptr && chunk == 0;
  const bool has_gate = PRO == 0 && a.gate != nullptr;
  const bool swish_in = PRO == 0 && a.in_act == MMD_ACT_SWISH;
  const bool swish_bb = PRO == 1 && a.bb.act == MMD_ACT_SWISH;

  // this lane's k (from the row start) in granule g, clamped inside the slice (a tail granule's surplus elements are zeroed on the A side)
  auto koff = [&](int g) { return min(kbeg + g * 32 + 4 * p, kend - 4); };
  float4 ra[4], rz[4], rb[NBI];
  auto load_a = [&](int g) {
    const int k = koff(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = mmd_ld4(xrow[i] + k);
      if constexpr (PRO == 1) rz[i] = mmd_ld4(grow[i] + k);
      else if (has_gate) rz[i] = mmd_ld4(grow[i] + k);
    }
  };
  auto load_b = [&](int g) {
    const int k = koff(g);
#pragma unroll
    for (int it = 0; it < NBI; ++it) rb[it] = mmd_ld4(a.w + (wofs[it] + k));
  };
  // first granule's loads in flight while the coefficient table is filled
  const int g0 = wave < G ? wave : 0;
  load_a(g0);
  load_b(g0);

  // ---- per-channel coefficients of the slice, once per block.  A thread owns channels kbeg + tid + 256 u; the loads of all its channels
  // are issued before the first use (clamped addresses): one round trip for the table instead of one per 256 channels.
  constexpr int TU = 3;                              // channels per thread (slices of up to 768 channels; longer ones loop)
  for (int cb = kbeg; cb < kend; cb += 256 * TU) {
    if constexpr (PRO == 1) {
      // dgamma / dbeta (+)= the reduce pass' sums: channel group `it` (256 channels) of a slice by that slice's block of row slab `it` - one
      // round per block, riding on the sums the table fill loads anyway (block 0 doing all K channels was the launch's slowest block by 5+ us)
      const bool dg = a.bb.dgamma != nullptr && chunk == 0;
      double s0[TU], s1[TU]; float is[TU], scl[TU], mn[TU], shf[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int c = min(cb + tid + 256 * u, kend - 1);
        s0[u] = a.bb.sums[c]; s1[u] = a.bb.sums[a.bb.C + c]; is[u] = a.bb.invstd[c]; scl[u] = a.bb.scale[c]; mn[u] = a.bb.mean[c]; shf[u] = a.bb.shift[c];
      }
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int c = cb + tid + 256 * u;
        if (c < kend) {
          const float m1 = (float)(s0[u] * a.bb.inv_count), m2 = (float)(s1[u] * a.bb.inv_count);      // (bn_bwd_coef's arithmetic)
          const int j = c - kbeg;
          sTab[j] = scl[u]; sTab[sa.ktab + j] = -scl[u] * is[u] * m2; sTab[2 * sa.ktab + j] = -scl[u] * m1;
          sTab[3 * sa.ktab + j] = mn[u]; sTab[4 * sa.ktab + j] = shf[u];
          if (dg && (c - kbeg) / 256 == slab) { a.bb.dgamma[c] += (float)s1[u]; a.bb.dbeta[c] += (float)s0[u]; }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int c = cb + tid + 256 * u;
        if (c < kend) {
          float sc = 1.f, sh = 0.f;
          if (a.in_bn.stats) bn_live_coef(a.in_bn, c, sc, sh);
          else if (a.in_scale) { sc = a.in_scale[c]; sh = a.in_shift[c]; }
          sTab[c - kbeg] = sc; sTab[sa.ktab + c - kbeg] = sh;
        }
      }
    }
  }
  f32x16 acc[NT32];
#pragma unroll
  for (int j = 0; j < NT32; ++j)
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) acc[j][qq] = 0.f;
  __syncthreads();                                   // the coefficient table is complete
  SL_T(1);

  // One granule: prologue on the coalesced registers -> this wave's LDS region (+ the stored dz), B rows -> LDS, then (NEXT) the loads of the
  // wave's next granule into the registers just stored - they have the granule's NT32 x 16 MFMAs (x 64 cycles) to land - and the MFMAs on
  // fragments read back from LDS.  DS operations of one wave execute in issue order, so the write -> read hand-off inside the wave needs
  // no barrier (the fence keeps the compiler from reordering them).
  // DZ: 0 = no dz side output, 1 = stored unconditionally (whole 32-row slab, whole granule: the common case), 2 = per-lane guard.  A guarded
  // store is a branch, and behind a branch hipcc's wait counting starts over: every later use of a prefetched register waited vmcnt(0),
  // i.e. for the stores just issued - so the guard is compiled in only where a tail needs it.
  auto granule = [&](int g, int gn, auto next, auto dzc) __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(next)::value;
    constexpr int DZ = decltype(dzc)::value;
    const int kk = kbeg + g * 32 + 4 * p;                      // this lane's k (may lie past the slice's end in a tail granule)
    const bool kok = kk < kend;
    const int jt = (kok ? kk : kend - 4) - kbeg;
    if constexpr (PRO == 1) {
      BnBwdCoef4 bq;
      bn_bwd_tab4(sTab, sa.ktab, jt, bq);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 vs = bn_bwd_eval4(ra[i], rz[i], rowsc[i], MMD_ACT_SWISH, bq), vn = bn_bwd_eval4(ra[i], rz[i], rowsc[i], MMD_ACT_NONE, bq);
        const float4 v = swish_bb ? vs : vn;
        // (DZ 1 stores from every column chunk's block: the chunks evaluate identical values - N > 224 only)
        if constexpr (DZ == 1) mmd_st4(a.bb.dz_out + (size_t)(m0 + q + 8 * i) * a.K + kk, v);
        else if constexpr (DZ == 2) { if (dz_here && kok && rok[i]) mmd_st4(a.bb.dz_out + (size_t)(m0 + q + 8 * i) * a.K + kk, v); }
        *reinterpret_cast<float4*>(&wA[(q + 8 * i) * GLD + 4 * p]) = sl_mask(kok, v);
      }
    } else {
      const float4 sc = *reinterpret_cast<const float4*>(sTab + jt), sh = *reinterpret_cast<const float4*>(sTab + sa.ktab + jt);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = make_float4(ra[i].x * sc.x + sh.x, ra[i].y * sc.y + sh.y, ra[i].z * sc.z + sh.z, ra[i].w * sc.w + sh.w);
        if (swish_in) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
        if (has_gate) { v.x *= rz[i].x; v.y *= rz[i].y; v.z *= rz[i].z; v.w *= rz[i].w; }
        *reinterpret_cast<float4*>(&wA[(q + 8 * i) * GLD + 4 * p]) = sl_mask(kok, v);
      }
    }
#pragma unroll
    for (int it = 0; it < NBI; ++it) *reinterpret_cast<float4*>(&wB[(q + 8 * it) * GLD + 4 * p]) = rb[it];
    if constexpr (NEXT) { load_a(gn); load_b(gn); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fragments: lane (r, h) supplies row r, k = 16 h + 4 i + {0..3}
    const float* pa = wA + r * GLD + h * 16;
    const float* pb = wB + r * GLD + h * 16;
    // (the fragments of sub-block i + 1 - one A and NT32 B float4 - are read before the 4 NT32 MFMAs of sub-block i are issued, pinned by
    // sched_barrier: written the plain way hipcc puts every read right in front of its four MFMAs with an lgkmcnt(0) - ~100 cycles of LDS
    // latency exposed per 256 cycles of MFMA issue)
    float4 av[2], bv[2][NT32];
    auto frag = [&](int i, int sl) {
      av[sl] = *reinterpret_cast<const float4*>(pa + 4 * i);
#pragma unroll
      for (int j = 0; j < NT32; ++j) bv[sl][j] = *reinterpret_cast<const float4*>(pb + j * 32 * GLD + 4 * i);
    };
    frag(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sl = i & 1;
      if (i + 1 < 4) frag(i + 1, sl ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT32; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].x, bv[sl][j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].y, bv[sl][j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].z, bv[sl][j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].w, bv[sl][j].w, acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the next granule's LDS stores stay behind these reads)
    __builtin_amdgcn_wave_barrier();
  };
  if (wave < G) {
    int g = wave;
    int sti = 0;
    for (; g + 4 < G; g += 4) { granule(g, g + 4, std::true_type{}, std::integral_constant<int, DZK>{}); SL_T(2 + sti); ++sti; }
    // (only the slice's very last granule can be cut by kend - always some wave's LAST one: it runs the guarded form)
    if (DZK == 1 && kbeg + g * 32 + 32 > kend) granule(g, g, std::false_type{}, std::integral_constant<int, 2>{});
    else granule(g, g, std::false_type{}, std::integral_constant<int, DZK>{});
  }
  SL_T(60);
  __syncthreads();                                   // every wave is done with its staging region: the space becomes the reduction scratch

  // ---- cross-wave K reduction as a reduce-scatter: tile j is finished by wave j % 4; every other wave hands its accumulators of that tile
  // over in LDS (lane-contiguous float4: conflict-free ds_write_b128 / ds_read_b128), one barrier, then the owner adds the three and
  // writes the tile row-major for the vectorised epilogue
#pragma unroll
  for (int j = 0; j < NT32; ++j) {
    const int owner = j & 3;
    if (wave != owner) {
      const int idx = (wave - owner - 1) & 3;       // 0..2
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(&scr[(((j * 3 + idx) * 4 + qq) * 64 + lane) * 4]) =
            make_float4(acc[j][4 * qq], acc[j][4 * qq + 1], acc[j][4 * qq + 2], acc[j][4 * qq + 3]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NT32; ++j) {
    if (wave == (j & 3)) {
#pragma unroll
      for (int idx = 0; idx < 3; ++idx)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4 u = *reinterpret_cast<const float4*>(&scr[(((j * 3 + idx) * 4 + qq) * 64 + lane) * 4]);
          acc[j][4 * qq] += u.x; acc[j][4 * qq + 1] += u.y; acc[j][4 * qq + 2] += u.z; acc[j][4 * qq + 3] += u.w;
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) tile[((q & 3) + 8 * (q >> 2) + 4 * h) * NW + j * 32 + r] = acc[j][q];
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
  slab_epilogue<SL_BM, (SL_BM >> (8 - LG))>(a, m0, n0, NC4, LG, [&](int rl, int c4) { return *reinterpret_cast<const float4*>(&tile[rl * NW + c4 * 4]); },
                                            tile + SL_BM * NW);
  SL_TW(62);
}

// Combine launch of a K-sliced slab GEMM: adds the slices' partial slabs in slice order and runs the epilogue.  One block per 16 rows x
// column chunk (twice the GEMM launch's row slabs: 128 blocks at M = 2048, every thread with all its loads - NS partials, residual, the
// BatchNorm sums' z for each of its rows - in flight at once; the first version, 64 blocks walking 8 rows one dependent round trip at a
// time, took 16 us).
template <int NS, int LG>
__global__ __launch_bounds__(256) void pw_slab_combine_kernel(SlabArgs sa, int nw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BR = 16, R = (BR >> (8 - LG)) > 0 ? (BR >> (8 - LG)) : 1;
  const PwArgs& a = sa.p;
  const int chunk = blockIdx.x % sa.nchunk, sub = blockIdx.x / sa.nchunk;
  const int m0 = sub * BR, n0 = chunk * nw;
  const int NWT = sa.nchunk * nw;
  const size_t sstride = (size_t)((gridDim.x / sa.nchunk + 1) / 2) * SL_BM * NWT;      // rows of a slice's slab: whole 32-row slabs
  const float* src = sa.part + (size_t)m0 * NWT + n0;
  SL_T(64);
  slab_epilogue<BR, R>(a, m0, n0, nw / 4, LG, [&](int rl, int c4) {
    const float* p = src + (size_t)rl * NWT + c4 * 4;
    float4 u[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) u[s] = mmd_ld4(p + s * sstride);
    float4 v = u[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) { v.x += u[s].x; v.y += u[s].y; v.z += u[s].z; v.w += u[s].w; }
    return v;
  }, smem);
  SL_TW(65);
}

typedef void (*SlabKern)(SlabArgs);
typedef void (*SlabComb)(SlabArgs, int);
template <int PRO, int DZK>
static SlabKern slab_pick_nt(int nt32) {
  switch (nt32) {
    case 2: return pw_slab_kernel<2, PRO, DZK>;
    case 3: return pw_slab_kernel<3, PRO, DZK>;
    case 4: return pw_slab_kernel<4, PRO, DZK>;
    case 5: return pw_slab_kernel<5, PRO, DZK>;
    case 6: return pw_slab_kernel<6, PRO, DZK>;
    case 7: return pw_slab_kernel<7, PRO, DZK>;
    default: return nullptr;
  }
}
template <int NS>
static SlabComb slab_pick_comb_lg(int lg) {
  return lg == 4 ? pw_slab_combine_kernel<NS, 4> : lg == 5 ? pw_slab_combine_kernel<NS, 5> : lg == 6 ? pw_slab_combine_kernel<NS, 6> : nullptr;
}
static SlabComb slab_pick_comb(int ns, int lg) {
  switch (ns) {
    case 2: return slab_pick_comb_lg<2>(lg);
    case 3: return slab_pick_comb_lg<3>(lg);
    case 4: return slab_pick_comb_lg<4>(lg);
    case 5: return slab_pick_comb_lg<5>(lg);
    case 6: return slab_pick_comb_lg<6>(lg);
    case 7: return slab_pick_comb_lg<7>(lg);
    case 8: return slab_pick_comb_lg<8>(lg);
    default: return nullptr;
  }
}

// Geometry of a launch: -> false when the slab kernel does not cover it
struct SlabPlan { int nt32, nchunk, nslab, nslice, gran, ktab, lg; size_t lds; long long part_floats; };
static bool slab_plan(int M, int K, int N, int pro, SlabPlan& p) {
  if (M <= 0 || (K & 3) || (N & 3) || K < 128 || N < 36) return false;
  const int tiles = cdiv(N, 32);
  p.nchunk = cdiv(tiles, 7);      // at most 7 tiles per block: 4 waves x (32 + 224) staged rows x 144 bytes = 147 KB of LDS
  p.nt32 = cdiv(tiles, p.nchunk);
  if (p.nt32 < 2) return false;
  p.nslab = cdiv(M, SL_BM);
  const int blocks = p.nslab * p.nchunk;
  const int gtot = cdiv(K, 32);
  static const int target = getenv("MMD_SLAB_BLOCKS") ? atoi(getenv("MMD_SLAB_BLOCKS")) : 256;
  // K slices: fill the chip (about one block per CU); at least 8 granules per slice (two per wave), at most 8 slices
  int ns = blocks >= target ? 1 : target / blocks;
  ns = max(1, min(min(ns, 8), gtot / 8));
  p.gran = cdiv(gtot, ns);
  p.nslice = cdiv(gtot, p.gran);
  const int nco = pro == 1 ? 5 : 2;
  p.ktab = p.gran * 32;
  const int nw = p.nt32 * 32, nc4 = nw / 4;
  p.lg = nc4 <= 16 ? 4 : nc4 <= 32 ? 5 : 6;
  // table | max(the four waves' staging regions [32 + NW][36], the non-owners' accumulators [NT32][3][4 KB] + finished tile [32][NW] +
  // epilogue sums [2 RG][NW])
  const size_t stage_f = (size_t)4 * (32 + nw) * 36, red_f = (size_t)p.nt32 * 3 * 1024 + (size_t)SL_BM * nw + (size_t)2 * (256 >> p.lg) * nw;
  p.lds = ((size_t)nco * p.ktab + (stage_f > red_f ? stage_f : red_f)) * sizeof(float);
  if (p.lds > 158 * 1024) return false;
  if (pro == 1 && cdiv(p.gran * 32, 256) > p.nslab) return false;      // (dgamma / dbeta: one 256-channel group per row slab's block)
  p.part_floats = p.nslice > 1 ? (long long)p.nslice * p.nslab * SL_BM * p.nchunk * nw : 0;
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
  SlabKern kern = !pro ? slab_pick_nt<0, 0>(p.nt32) : (a.bb.dz_out && a.M % SL_BM == 0) ? slab_pick_nt<1, 1>(p.nt32) : slab_pick_nt<1, 2>(p.nt32);
  SlabComb comb = p.nslice > 1 ? slab_pick_comb(p.nslice, p.lg) : nullptr;
  if (!kern || (p.nslice > 1 && !comb)) return 0;
  {       // raise the kernel's dynamic-LDS limit once per instantiation (18 of them)
    static SlabKern done[24]; static int ndone = 0;
    bool seen = false;
    for (int i = 0; i < ndone; ++i) seen |= done[i] == kern;
    if (!seen) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MMD_ELAUNCH;
      if (ndone < 24) done[ndone++] = kern;
    }
  }
  const int nblk = p.nslab * p.nchunk * p.nslice;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), p.lds, stream, sa);
  if (p.nslice > 1) {
    const int nw = p.nt32 * 32;
    const size_t lds = (size_t)2 * (256 >> p.lg) * nw * sizeof(float);
    hipLaunchKernelGGL(comb, dim3(2 * p.nslab * p.nchunk), dim3(256), lds, stream, sa, nw);
  }
  return 1;
}
