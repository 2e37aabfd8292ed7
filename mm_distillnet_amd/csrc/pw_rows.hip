// Thin-K pointwise (1x1) convolution: Y[M,N] = pro(X)[M,K] * W[N,K]^T (+ epilogue) for K <= 128 — CDNA4 / gfx950.
//
// Most 1x1 convs of EfficientDet have a SHORT reduction (the 112-channel BiFPN / head layers: 230 launches per step; the MBConv
// expand convs 16->96 ... 120->720) and a tall M (B*H*W rows).  An LDS-tiled K loop has 1-4 iterations there, so its blocks run
// load -> LDS -> barrier -> MFMA -> barrier -> store in lock step and the MFMA pipe idles ~60 % of the time (profiles/r01_notes.md).
// This kernel removes the K loop and every barrier from the steady state:
//   * the weight panel W[pcols, K] is staged ONCE per block into LDS (row stride K+4 floats: a 32-lane group's ds_read_b64
//     fragments cover the 64 banks exactly once - as long as the reads stay single ds_read_b64, see the fragment loop) and stays resident;
//   * each WAVE owns 16-row slabs of X: the whole-K A fragment of a slab lives in registers (K/4 VGPRs), loaded straight
//     from global memory in the MFMA operand layout (lane (r, g) -> row r, k = 8kk + 2g + {0,1}: 32 B per row per
//     instruction, all bytes of the 16 rows over the K/8 instructions issued back to back), the NEXT slab's fragment is
//     in flight while the current one is multiplied;
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue): 16-row / 16-column granularity, so N = 112 / 144 / 528 / 720 pay
//     no column padding and the tail of M is quantised in 16-row units across 2048 wave slots;
//   * waves never synchronise after the panel is staged: two blocks per CU (2 waves per SIMD) interleave one wave's epilogue
//     with the other's MFMAs.
// The producer's BatchNorm + swish + squeeze-excite gate (pro) are applied to the A fragment in registers, the epilogue
// (bias / BN statistics / folded BN / activation / residual / strided head output / feature-pyramid levels) matches
// pw_gemm.hip's.  Reference op: nn.Conv2d(k=1) inside Conv2dStaticSamePadding (src/YetAnotherEfficientNet.py:27-65; call
// sites :427,446, src/YetAnotherEfficientDet.py:171,238-265).
#include "common.h"
#include "pw_args.h"
#include <cstdlib>

#define RW_AFF 1
#define RW_SWISH 2
#define RW_GATE 4
// dev ablations (MMD_ROWS_ABL bit mask, timing experiments only - results are wrong): 256 no output stores, 512 no A fetch after the
// first slab, 1024 a single k group of MFMAs
#define RW_ABL_NOSTORE 256
#define RW_ABL_NOFETCH 512
#define RW_ABL_NOMFMA 1024

struct RowsArgs {
  int LDB;          // LDS row stride of the weight panel (K + 4 floats)
  int nslabs;       // 16-row slabs
  int spw;          // consecutive slabs per wave
  int cpp;          // column chunks (C*16 columns each) per panel
  int pcols;        // cpp * C * 16
  int npanels, bpp; // column panels x blocks per panel
  int mode;         // RW_* prologue bits
  int nlev;         // statistic sets kept in LDS (pyramid levels, else 1); 0 = no statistics
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

template <int NKK, int C>
__global__ __launch_bounds__(512, 2) void pw_rows_kernel(PwArgs a, RowsArgs ra) {
  extern __shared__ float smem[];
  const int K = NKK * 8, N = a.N, LDB = ra.LDB, pcols = ra.pcols;
  float* const sB = smem;                          // [pcols][LDB]
  float* const sCoef = sB + pcols * LDB;           // [2][K]   producer scale / shift
  float* const sEp = sCoef + 2 * K;                // [3][pcols] bias, out_scale, out_shift
  float* const sStat = sEp + 3 * pcols;            // [nlev][2][pcols]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform for the compiler too: slab indices, level lookups and
                                                                  // loop branches then live in SGPRs
  const int panel = blockIdx.x / ra.bpp, bi = blockIdx.x - panel * ra.bpp;
  const int n0p = panel * pcols;

  // 8 waves per block = 2 per SIMD (waves w and w + 4 share a SIMD): the slabs are dealt evenly over the panel's SIMD slots
  // (consecutive slabs per slot), and a slot's two waves split its run, so while one wave is in its epilogue / fragment transform the
  // other keeps the MFMA pipe busy
  const int slot = bi * 4 + (wave & 3), nslot = ra.bpp * 4;
  const int q_beg = (int)((long long)slot * ra.nslabs / nslot), q_end = (int)((long long)(slot + 1) * ra.nslabs / nslot);
  const int q_mid = q_beg + (q_end - q_beg + 1) / 2;
  const int s_beg = (wave < 4) ? q_beg : q_mid, s_end = (wave < 4) ? q_mid : q_end;
  f32x2 an[NKK], gn[NKK];
  int lev_n = 0, Mv_n = a.M, srow0_n = 0, rpi_n = a.rows_per_image;
    auto fetch = [&](int s) {       // raw A fragment (+ gate fragment) of slab s
      const int row0 = s * 16;
      lev_n = 0; Mv_n = a.M; srow0_n = 0; rpi_n = a.rows_per_image;
      if (a.pyr.n) {
        lev_n = __builtin_amdgcn_readfirstlane(pyr_level_of_row(a.pyr, row0));
        srow0_n = a.pyr.row0[lev_n]; rpi_n = a.pyr.H[lev_n] * a.pyr.W[lev_n]; Mv_n = srow0_n + a.pyr.B * rpi_n;
      }
      const int row = min(row0 + r, Mv_n - 1);          // clamped: rows past the end are computed but never stored
      const float* xp = a.x + (size_t)row * K + 2 * g;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) an[kk] = *reinterpret_cast<const f32x2*>(xp + kk * 8);
      if (ra.mode & RW_GATE) {
        const float* gp = a.gate + (size_t)(row / rpi_n) * K + 2 * g;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) gn[kk] = *reinterpret_cast<const f32x2*>(gp + kk * 8);
      }
    };
  if (s_beg < s_end) fetch(s_beg);      // the first slab's fragment is in flight while the panel is staged
  // ---- once per block: weight panel, producer coefficients, epilogue vectors
  {
    // LDS-DMA (global_load_lds_dwordx4): no VGPR staging, every piece of the panel in flight at once, ONE wait.  A wave instruction
    // fills 64 consecutive 16-byte LDS slots (wave-uniform base + lane * 16); slot q of the panel image is row q / (K/4 + 1),
    // 16-byte column q % (K/4 + 1), and the last column of every row is the 4-float pad: its lane is masked off, which leaves the
    // hole the padded row stride needs.  Rows past N are clamped copies of row N-1: they only feed output columns that are never
    // stored or summed.  (A register-staged loop - load, wait, ds_write - cost ~7 us per block: 7 dependent L2 round trips.)
    const int f4 = K >> 2, spr = f4 + 1, total = pcols * spr;
    const int nI = (total + 63) >> 6;
    for (int ii = wave; ii < nI; ii += 8) {
      const int slot = ii * 64 + lane;
      const int n = slot / spr, k4 = slot - n * spr;
      const float* src = a.w + (size_t)min(n0p + n, N - 1) * K + min(k4, f4 - 1) * 4;
      if (slot < total && k4 < f4)
        __builtin_amdgcn_global_load_lds((glb_vptr)(uintptr_t)src, (lds_vptr)(uintptr_t)(sB + ii * 256), 16, 0, 0);
    }
    if (ra.mode & RW_AFF)
      for (int k = tid; k < K; k += 512) {
        float sc, sh;
        if (a.in_bn.stats) bn_live_coef(a.in_bn, k, sc, sh);
        else { sc = a.in_scale[k]; sh = a.in_shift[k]; }
        sCoef[k] = sc; sCoef[K + k] = sh;
      }
    for (int c = tid; c < pcols; c += 512) {
      const int col = n0p + c;
      const bool ok = col < N;
      sEp[c] = (a.bias && ok) ? a.bias[col] : 0.f;
      sEp[pcols + c] = (a.out_scale && ok) ? a.out_scale[col] : 1.f;
      sEp[2 * pcols + c] = (a.out_scale && ok) ? a.out_shift[col] : 0.f;
    }
    for (int i = tid; i < ra.nlev * 2 * pcols; i += 512) sStat[i] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the panel pieces (LDS-DMA counts on vmcnt) have landed ...
  __syncthreads();                                       // ... in every wave

  if (s_beg < s_end) {
    for (int s = s_beg; s < s_end; ++s) {
      const int row0 = s * 16;
      const int lev = lev_n, Mv = Mv_n, srow0 = srow0_n, rpi = rpi_n;
      // ---- producer transform on the fragment: act(x*scale + shift) * gate
      f32x2 ac[NKK];
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) ac[kk] = an[kk];
      if (ra.mode & RW_AFF) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          const f32x2 sc = *reinterpret_cast<const f32x2*>(&sCoef[kk * 8 + 2 * g]);
          const f32x2 sh = *reinterpret_cast<const f32x2*>(&sCoef[K + kk * 8 + 2 * g]);
          ac[kk].x = ac[kk].x * sc.x + sh.x; ac[kk].y = ac[kk].y * sc.y + sh.y;
        }
      }
      if (ra.mode & RW_SWISH) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) { ac[kk].x = mmd_swish(ac[kk].x); ac[kk].y = mmd_swish(ac[kk].y); }
      }
      if (ra.mode & RW_GATE) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) { ac[kk].x *= gn[kk].x; ac[kk].y *= gn[kk].y; }
      }
      if (s + 1 < s_end && !(ra.mode & RW_ABL_NOFETCH)) fetch(s + 1);                  // in flight during this slab's MFMAs
      // ---- output row bookkeeping: lane (r, g) owns rows row0 + 4g + i, i = 0..3, of column (tile*16 + r).
      // Everything below is phrased as whole-chunk passes behind wave-uniform branches: a wave64 VALU instruction costs 4 cycles, so a
      // per-element "if (valid) ... if (stats) ... if (act)" epilogue (~20 instructions x 112 elements) took longer than the MFMAs.
      const bool rows_full = row0 + 16 <= Mv;            // wave-uniform
      size_t off[4]; bool rok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * g + i;
        rok[i] = row < Mv;
        const int rr = rok[i] ? row : row0;
        if (a.y_batch_stride) {
          const int img = (rr - srow0) / rpi;
          const long long yoff = a.pyr.n ? a.yoff_lev[lev] : a.y_offset;
          off[i] = (size_t)img * a.y_batch_stride + yoff + (size_t)(rr - srow0 - img * rpi) * N;
        } else {
          off[i] = (size_t)rr * N;
        }
      }
      for (int ch = 0; ch < ra.cpp; ++ch) {
        const int cbase = ch * C * 16;
        if (n0p + cbase >= N) break;                    // wave-uniform
        const bool full = rows_full && (n0p + cbase + C * 16 <= N);       // no row / column of this 16 x 16C block is padding
        const int c0 = n0p + cbase + r;                  // this lane's column in tile 0 (tile j: + 16 j)
        float res[C][4];
        if (a.residual) {
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) res[j][i] = (c0 + j * 16 < N && rok[i]) ? a.residual[off[i] + c0 + j * 16] : 0.f;
        }
        f32x4 acc[C];
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // B fragments as single ds_read_b64 (32-lane groups, bank = dword address mod 64: rows r * (K+4) + 2g + {0,1} cover the 64 banks
        // exactly once per group).  Left to itself the compiler pairs the reads of k groups kk and kk+1 (32 B apart) into ds_read2_b64,
        // whose 16-lane groups bank modulo 32 - rows r and r+8 then collide (SQ_LDS_BANK_CONFLICT 139 % of the LDS cycles in
        // profiles/r02_sq_counters_by_kernel.txt); an opaque LDS byte address per k group keeps them apart.
        typedef const __attribute__((address_space(3))) f32x2* lds_f2p;
        const unsigned bp0 = (unsigned)(uintptr_t)(lds_vptr)(uintptr_t)(sB + (cbase + r) * LDB + 2 * g);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          if ((ra.mode & RW_ABL_NOMFMA) && kk > 0) break;
          f32x2 b[C];
          unsigned bpk = bp0 + kk * 32;
          asm volatile("" : "+v"(bpk));
#pragma unroll
          for (int j = 0; j < C; ++j) b[j] = *(lds_f2p)(uintptr_t)(bpk + j * 16 * LDB * 4);
#pragma unroll
          for (int j = 0; j < C; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[kk].x, b[j].x, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < C; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[kk].y, b[j].y, acc[j], 0, 0, 0);
        }
        // ---- epilogue of the chunk, straight from the accumulators (64-B row segments per 16 lanes)
        const float* ep = sEp + cbase + r;
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < C; ++j) { const float bias = ep[j * 16]; acc[j] += bias; }
        }
        if (ra.nlev) {            // BatchNorm statistics of the raw (pre-affine) output
#pragma unroll
          for (int j = 0; j < C; ++j) {
            float ssum = 0.f, ssq = 0.f;
            const bool cok = c0 + j * 16 < N;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = (full || (cok && rok[i])) ? acc[j][i] : 0.f;
              ssum += v; ssq += v * v;
            }
            ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
            ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
            if (g == 0 && cok) {
              atomicAdd(&sStat[(lev * 2) * pcols + cbase + j * 16 + r], ssum);
              atomicAdd(&sStat[(lev * 2 + 1) * pcols + cbase + j * 16 + r], ssq);
            }
          }
        }
        if (a.out_scale) {
#pragma unroll
          for (int j = 0; j < C; ++j) { const float osc = ep[pcols + j * 16], osh = ep[2 * pcols + j * 16]; acc[j] = acc[j] * osc + osh; }
        }
        if (a.out_act == MMD_ACT_SWISH) {
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mmd_swish(acc[j][i]);
        } else if (a.out_act == MMD_ACT_SIGMOID) {
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mmd_sigmoid(acc[j][i]);
        }
        if (a.residual) {
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] += res[j][i];
        }
        if (!(ra.mode & RW_ABL_NOSTORE)) {
          float* yp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) yp[i] = a.y + off[i] + c0;
          if (full) {
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
              for (int i = 0; i < 4; ++i) yp[i][j * 16] = acc[j][i];
          } else {
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c0 + j * 16 < N && rok[i]) yp[i][j * 16] = acc[j][i];
          }
        }
      }
    }
  }
  if (ra.nlev) {
    __syncthreads();
    double* base = a.stats_ws ? a.stats_ws + (size_t)(blockIdx.x % a.ws_slots) * 2 * N : a.stats;
    for (int i = tid; i < ra.nlev * 2 * pcols; i += 512) {
      const int l = i / (2 * pcols), rem = i - l * 2 * pcols, which = rem / pcols, c = rem - which * pcols;
      const int col = n0p + c;
      const float v = sStat[i];
      if (col < N && v != 0.f) atomicAdd(&base[(a.pyr.n ? 2 * (size_t)l * a.lev_stride : 0) + (size_t)which * N + col], (double)v);
    }
  }
}

typedef void (*RowsKern)(PwArgs, RowsArgs);
template <int C>
static RowsKern rows_pick_k(int nkk) {
  switch (nkk) {
    case 2: return pw_rows_kernel<2, C>;
    case 3: return pw_rows_kernel<3, C>;
    case 4: return pw_rows_kernel<4, C>;
    case 6: return pw_rows_kernel<6, C>;
    case 11: return pw_rows_kernel<11, C>;
    case 12: return pw_rows_kernel<12, C>;
    case 14: return pw_rows_kernel<14, C>;
    case 15: return pw_rows_kernel<15, C>;
    case 16: return pw_rows_kernel<16, C>;
    default: return nullptr;
  }
}
static RowsKern rows_pick(int nkk, int c) {
  return c == 7 ? rows_pick_k<7>(nkk) : c == 4 ? rows_pick_k<4>(nkk) : c == 3 ? rows_pick_k<3>(nkk) : c == 1 ? rows_pick_k<1>(nkk) : nullptr;
}

// a.form (pw_args.h, per call): MMD_PW_FORM_AUTO = the measured shape filter decides; _ROWS = every supported launch takes the row-slab
// kernel; _TILED = none does (tests, A/B timing)

int pw_rows_try(PwArgs& a, hipStream_t stream) {
  static const int off = getenv("MMD_NO_ROWS") ? 1 : 0;
  static const int lds_blk = getenv("MMD_ROWS_LDS") ? atoi(getenv("MMD_ROWS_LDS")) : 120 * 1024;     // panel budget (one 8-wave block per CU)
  if (off || a.bf16 || a.bb.z || a.xs.z || a.st.Cin) return 0;
  const int M = a.M, K = a.K, N = a.N;
  if (K < 16 || K > 128 || (K & 7)) return 0;
  // Measured against the LDS-tiled kernels per shape of the step (tools/dev/gemm_bench.py, profiles/r02_notes.md): the row-slab kernel
  // wins by 5-8 % on tall launches without BatchNorm statistics and N >= 48 (M43648 K112 N112 21.2 -> 19.8 us, M131072 K24 N144
  // 23.6 -> 22.0, M32768 K48 N288 19.5 -> 18.1, M524288 K16 N96 42.2 -> 40.5) and loses on the short ones (M <= 8192: the panel
  // staging + one slab per wave do not amortise).  MMD_ROWS_ALL=1 lifts the filter (tests run both ways).
  static const int all = getenv("MMD_ROWS_ALL") ? 1 : 0;
  if (a.form == MMD_PW_FORM_TILED || a.form == MMD_PW_FORM_LONGK || a.form == MMD_PW_FORM_SLAB) return 0;
  if (!all && a.form != MMD_PW_FORM_ROWS && (M < 32768 || a.stats || N < 48)) return 0;
  if (a.in_act != MMD_ACT_NONE && a.in_act != MMD_ACT_SWISH) return 0;
  if (a.gate && (a.pyr.n || (a.rows_per_image & 15))) return 0;        // a 16-row slab must lie inside one image
  const int nkk = K / 8;
  const int tiles = cdiv(N, 16);
  const int nslabs = cdiv(M, 16);
  // columns per chunk (C 16-wide tiles share one A fragment pass): least padding, wider chunks on ties
  int C = 1; double best = -1.0;
  const int cands[4] = {7, 4, 3, 1};
  for (int ci = 0; ci < 4; ++ci) {
    const int c = cands[ci];
    const double eff = (double)tiles / (cdiv(tiles, c) * c) * (c == 1 ? 0.85 : 1.0);
    if (eff > best + 1e-9) { best = eff; C = c; }
  }
  static const int small_slabs = getenv("MMD_ROWS_SMALL") ? atoi(getenv("MMD_ROWS_SMALL")) : 512;
  if (nslabs * cdiv(tiles, C) < small_slabs && nslabs < small_slabs) C = 1;   // tiny M: spread the columns over more waves instead
  RowsKern kern = rows_pick(nkk, C);
  if (!kern) return 0;
  RowsArgs ra{};
  ra.LDB = K + 4;
  ra.nslabs = nslabs;
  const int nchunks = cdiv(tiles, C);
  const int chunk_bytes = C * 16 * ra.LDB * 4;
  int cpp = lds_blk / chunk_bytes; if (cpp < 1) cpp = 1; if (cpp > nchunks) cpp = nchunks;
  while (cpp > 1 && (long long)nslabs * cdiv(nchunks, cpp) < 1024) --cpp;      // small M: more panels = more waves
  ra.cpp = cpp; ra.pcols = cpp * C * 16; ra.npanels = cdiv(nchunks, cpp);
  ra.mode = ((a.in_scale || a.in_bn.stats) ? RW_AFF : 0) | (a.in_act == MMD_ACT_SWISH ? RW_SWISH : 0) | (a.gate ? RW_GATE : 0);
  static const int abl = (getenv("MMD_DEV") && getenv("MMD_ROWS_ABL")) ? atoi(getenv("MMD_ROWS_ABL")) : 0;      // honoured under MMD_DEV=1 only (_lib.py WORK_SKIPPING)
  ra.mode |= abl;
  ra.nlev = a.stats ? (a.pyr.n ? a.pyr.n : 1) : 0;
  const size_t lds = ((size_t)ra.pcols * ra.LDB + 2 * K + 3 * ra.pcols + (size_t)ra.nlev * 2 * ra.pcols) * sizeof(float);
  if (lds > 150 * 1024) return 0;
  // one block per CU (one wave per SIMD) unless told otherwise: the slabs are dealt evenly, so every SIMD carries
  // ceil(nslabs / 1024) or one less
  static const int blk_target = getenv("MMD_ROWS_BLK") ? atoi(getenv("MMD_ROWS_BLK")) : 256;
  int maxblk = blk_target;
  if (a.stats && !a.stats_ws && maxblk > 128 * ra.npanels) maxblk = 128 * ra.npanels;   // every block ends in same-address f64 atomics (~17 ns each)
  int bpp = maxblk / ra.npanels; if (bpp < 1) bpp = 1;
  if (bpp > cdiv(nslabs, 8)) bpp = cdiv(nslabs, 8);            // at least two slabs per SIMD slot: one for each of its waves
  ra.spw = cdiv(nslabs, bpp * 4);
  ra.bpp = bpp;
  static bool attr_done[17][8];
  if (!attr_done[nkk][C]) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[nkk][C] = true;
  }
  hipLaunchKernelGGL(kern, dim3(ra.npanels * bpp), dim3(512), lds, stream, a, ra);
  return 1;
}
