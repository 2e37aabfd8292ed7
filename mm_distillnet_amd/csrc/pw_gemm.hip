// Pointwise (1x1) convolution as an fp32 MFMA GEMM on NHWC rows — CDNA4 / gfx950.
//
//   forward / input-gradient:  Y[M,N] = pro(X)[M,K] * W[N,K]^T   (+ epilogue)        -> pw_gemm_kernel
//   weight-gradient:           dW[N,K] += dY[M,N]^T * pro(X)[M,K]                     -> pw_wgrad_kernel
//
// pro(X) = act(X*in_scale[k] + in_shift[k]) * gate[image(row), k]  is applied while the A tile is staged
// through registers into LDS, so a producer's BatchNorm+swish (and the MBConv squeeze-excite gate)
// never round-trip through HBM as a separate tensor.
// Reference op: nn.Conv2d(k=1) inside Conv2dStaticSamePadding (src/YetAnotherEfficientNet.py:27-65,
// call sites :427,446 and src/YetAnotherEfficientDet.py:171,238-265); BN/swish fused here replace
// src/YetAnotherEfficientNet.py:428,447,126-143.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A and B tiles sit in LDS as
// [row][k] with k contiguous and a 4-float row pad (stride 36 floats -> conflict-free ds_read_b128);
// each lane reads 4 consecutive k for its row, lanes 32..63 take the next 4 k, which feeds 4 MFMAs.
#include "common.h"
#include "pw_args.h"
#include <cstdlib>

// ---- bf16 mixed precision (BASELINE config 5): the SAME kernels with the inner product on v_mfma_f32_32x32x16_bf16.
// Activations and weights stay fp32 in HBM and LDS; each lane converts its 8 consecutive k (RNE, v_cvt_pk_bf16_f32) right
// before the MFMA, accumulation and every prologue / epilogue stay fp32.  16x the fp32 MFMA rate: the layers become
// load/store-bound.  A lane (r = lane & 31, h = lane >> 5) supplies row r, k = 8h .. 8h+7 of a 16-wide k group, for A and B.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// ---- split mode (BF == 2): fp32 products as six bf16 MFMAs on a three-way exact split of both operands (common.h)
// one quad of a tile row -> its words in the three planes of the row ([h: KW words][m: KW words][l: KW words], KW = tile k / 2)
template <int KW>
__device__ __forceinline__ void split3_store(unsigned* p, const float4& v) {
  unsigned h0, m0, l0, h1, m1, l1;
  mmd_split3_pk(v.x, v.y, h0, m0, l0); mmd_split3_pk(v.z, v.w, h1, m1, l1);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + KW) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2 * KW) = make_uint2(l0, l1);
}
__device__ __forceinline__ bf16x8 pack_bf16x8(float4 a, float4 b) {
  u32x4_t u = {pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w)};
  return __builtin_bit_cast(bf16x8, u);
}

// (pw_xs_acc, the epilogue helper of the BnSumOp mode: pw_args.h)

// Pool5Op epilogue: one output quad v (= g1) and the z1 quad at the same position -> the five running sums (chan_pool_bwd_kernel's arithmetic)
struct P5Coef { float4 sc, sh, mu, is; };
__device__ __forceinline__ void pw_p5_acc_pre(const P5Coef& q, const float4& v, const float4& zz, float4 (&acc)[5]);
__device__ __forceinline__ void pw_p5_acc(const Pool5Op& p, const P5Coef& q, const float4& v, size_t off, float4 (&acc)[5], int z16) {
  pw_p5_acc_pre(q, v, mmd_ldw4(p.z, off, z16), acc);
}
__device__ __forceinline__ void pw_p5_acc_pre(const P5Coef& q, const float4& v, const float4& zz, float4 (&acc)[5]) {
  const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gv[4] = {v.x, v.y, v.z, v.w};
  const float scv[4] = {q.sc.x, q.sc.y, q.sc.z, q.sc.w}, shv[4] = {q.sh.x, q.sh.y, q.sh.z, q.sh.w};
  const float muv[4] = {q.mu.x, q.mu.y, q.mu.z, q.mu.w}, isv[4] = {q.is.x, q.is.y, q.is.z, q.is.w};
  float r[5][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u = zv[i] * scv[i] + shv[i];
    const float sg = mmd_sigmoid(u);
    const float sp = sg * (1.0f + u * (1.0f - sg));
    const float xh = (zv[i] - muv[i]) * isv[i];
    r[0][i] = gv[i] * (u * sg); r[1][i] = gv[i] * sp; r[2][i] = gv[i] * sp * xh; r[3][i] = sp; r[4][i] = sp * xh;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) { acc[k].x += r[k][0]; acc[k].y += r[k][1]; acc[k].z += r[k][2]; acc[k].w += r[k][3]; }
}

#define PW_BM 128
#define PW_BK 32
#define PW_LD 36
#define PW_LDH 20      // bf16 mode: words per tile row (PW_BK / 2 data + 4 pad)
// split mode: a tile row = three planes of PW_BK / 2 = 16 words, NO pad; the four 4-word groups of a plane are XOR-ed with (row >> 2) & 3.
// Staging stores (8 lanes per row, one 64-bit store per plane): four consecutive rows cover banks 0 / 48 / 32 / 16 + 16 - all 64, no
// conflict (the padded stride 52 overlapped neighbouring rows by 4 banks: 21 - 24 % of the LDS cycles were conflicts); fragment reads
// (ds_read_b128, lane = row): rows with equal r & 3 share a 16-bank window and take different groups of it (tools/dev/lds_bank_model.py).
// (Measured flat on the step: the LDS pipe is not what these kernels wait for.  Kept for the smaller tiles.)
#define PW_LD3 48

// NKL = 8-wide k groups of the LAST K tile that hold data (1..4): fp32 MFMA runs at the vector rate (64 cycles per
// 32x32x2), so multiplying the zero padding of K = 112 / 48 / 24 ... is real time.  Compile-time so the hot loop keeps its schedule
// (a run-time trip count cost more than the padding).
// BM_T = 128: the 4 waves stack along M (32 rows each, all BN_T columns); BM_T = 64: 2 x 2 waves (32 rows x BN_T/2 columns
// each) - twice the blocks for the small-M layers (16x16 / 32x32 stages), whose 128-row tiling leaves most SIMDs with
// one wave or none.
// PRO = 1: the A operand is BnBwdOp(a.x = g, a.bb.z = z) evaluated while staging (input-gradient GEMM behind a BatchNorm).
// PRO = 3: "plain" A operand (no producer transform, no gate); PRO = 4: squeeze-excite gate only (the frozen nets' project convs).
// The unused gate / coefficient registers and branches are compiled out: 62-80 VGPRs instead of 90-128, i.e. five to six waves
// per SIMD instead of four (pw_gemm_kernel_lean) - the K loops of these layers are 1-7 tiles long, so what hides the load ->
// LDS -> MFMA chain of one block is the other blocks on the CU (weighted over a step's shapes: 8.78 -> 8.29 ms).
template <int BM_T, int BN_T, int NKL, int BF, int PRO>
__device__ __forceinline__ void pw_gemm_body(const PwArgs& a) {
  constexpr int WM = BM_T / 32;          // waves along M
  constexpr int WN = 4 / WM;             // waves along N
  constexpr int NS = BN_T / (32 * WN);   // 32-col slabs per wave
  constexpr int NA = BM_T / 32;          // A float4 loads per thread (BM_T*8/256)
  constexpr int NB = BN_T / 32;          // B float4 loads per thread (BN_T*8/256)
  constexpr int LDC = BN_T + 4;            // C staging row stride (floats)
  constexpr int LDT = BF == 2 ? PW_LD3 : PW_LD;      // words per tile row in the K loop
  constexpr int SM = (BM_T * LDC > (BM_T + BN_T) * LDT) ? BM_T * LDC : (BM_T + BN_T) * LDT;
  __shared__ float smem[SM];              // A|B tiles in the K loop, then the C tile for the vectorised epilogue
  // split mode: the K loop's tiles are larger than the C tile - the reduction scratch lives behind the C tile instead of beside the tiles
  // (39.9 KB per 128 x 64 block: four blocks per CU, which 42 KB would not allow)
  constexpr bool RED_IN = BF == 2 && BM_T * LDC + 8 * BN_T + ((PRO == 1) ? 20 * BN_T : 0) <= SM;
  __shared__ float sRedS[RED_IN ? 1 : 2 * 4 * BN_T];
  __shared__ float sRed5S[(PRO == 1 && !RED_IN) ? 5 * 4 * BN_T : 1];      // Pool5Op sums (BatchNorm-backward operand launches only)
  float* const sRed = RED_IN ? smem + BM_T * LDC : sRedS;
  float* const sRed5 = RED_IN ? smem + BM_T * LDC + 8 * BN_T : sRed5S;
  float* const sA = smem;
  float* const sB = smem + BM_T * PW_LD;
  unsigned* const sAu = reinterpret_cast<unsigned*>(smem);                   // BF: bf16 tiles, two elements per word, [row][PW_LDH]
  unsigned* const sBu = reinterpret_cast<unsigned*>(smem) + BM_T * (BF == 2 ? PW_LD3 : PW_LDH);

  const int tid = threadIdx.x;
  const int t = mmd_xcd_swizzle(blockIdx.x, a.nblk);
  const int tn = t % a.ntn, tm = t / a.ntn;
  const int m0 = tm * BM_T, n0 = tn * BN_T;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // pyramid launch: this tile lies inside one level; rows beyond the level's valid count are padding
  int Mv = a.M, srow0 = 0, rpi = a.rows_per_image; long long yoff = a.y_offset; double* stats = a.stats;
  if (a.pyr.n) {
    const int lev = pyr_level_of_row(a.pyr, m0);
    srow0 = a.pyr.row0[lev]; rpi = a.pyr.H[lev] * a.pyr.W[lev]; Mv = srow0 + a.pyr.B * rpi; yoff = a.yoff_lev[lev];
    if (stats) stats += 2 * lev * a.lev_stride;
  }
  if (a.stats_ws) stats = a.stats_ws + (size_t)(tm % a.ws_slots) * 2 * a.N;
  const int kq = (tid & 7) * 4;          // this thread's k offset inside a K tile
  const int lrow = tid >> 3;             // 0..31

  // per-thread row bookkeeping for the NA A loads
  const float* xrow[NA]; const float* grow[NA]; bool rok[NA]; float rowsc[NA];
  int sih[NA], siw[NA];                  // PRO == 2: top-left input coordinate of the row's 3x3 window
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int row = m0 + lrow + i * 32;
    rok[i] = row < Mv;
    int rr = rok[i] ? row : 0;
    xrow[i] = mmd_roww(a.x, rr, a.K, a.x16);
    if constexpr (PRO == 2) {
      const int ow = rr % a.st.OW, t1 = rr / a.st.OW, oh = t1 % a.st.OH, b = t1 / a.st.OH;
      sih[i] = oh * 2 - a.st.pad_t; siw[i] = ow * 2 - a.st.pad_l;
      xrow[i] = a.x + (size_t)b * a.st.Cin * a.st.H * a.st.W;
      grow[i] = nullptr;
    } else if constexpr (PRO == 1) {
      grow[i] = mmd_roww(a.bb.z, rr, a.K, a.z16);          // the second A tensor rides in the gate's registers
      rowsc[i] = a.bb.mul_b ? a.bb.mul_b[rr / a.bb.rows_per_image] : 1.f;
    } else if constexpr (PRO == 3) {
      grow[i] = nullptr;
    } else if constexpr (PRO == 4) {
      grow[i] = a.gate + (size_t)(rr / a.rows_per_image) * a.K;
    } else {
      grow[i] = a.gate ? a.gate + (size_t)(rr / a.rows_per_image) * a.K : nullptr;
    }
  }
  // grouped frozen nets: this row tile's group (a tile never straddles two groups: host-checked) selects the parameter set
  size_t gw = 0, gb = 0;
  if (a.g_images) { const int gi = ((m0 - srow0) / rpi) / a.g_images; gw = (size_t)gi * a.g_w; gb = (size_t)gi * a.g_bn; }
  const float* wrow[NB]; bool wok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int col = n0 + lrow + i * 32;
    wok[i] = col < a.N;
    wrow[i] = a.w + gw + (size_t)(wok[i] ? col : 0) * a.K;
  }

  f32x16 acc[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  float4 ra[NA], rg[NA], rb[NB], rsc, rsh;
  BnBwdCoef4 bq;
  bool kok;
  unsigned smask = 0;                     // PRO == 2: validity bit per (row slot, element), applied in lstore
  int kcur = 0;                           // PRO == 1: k of the staged group (for the dz side output)
  extern __shared__ float sBqTab[];       // PRO == 1 with a.bq_lds: [5][K] BatchNorm-backward coefficients (pw_args.h)
  if constexpr (PRO == 1) {
    if (a.bb.dgamma && t == 0)
      for (int c = tid; c < a.K; c += 256) { a.bb.dgamma[c] += (float)a.bb.sums[a.K + c]; a.bb.dbeta[c] += (float)a.bb.sums[c]; }
    if (a.bq_lds) { bn_bwd_tab_fill(a.bb, a.K, sBqTab, tid, 256); __syncthreads(); }
  }
  auto gload = [&](int k0) {
    // every load is unconditional on a clamped (always valid) address and masked afterwards: guarded loads compile to a
    // branch per load and a full vmcnt(0) drain, which serialises the prefetch
    int k = k0 + kq;
    kok = k < a.K;
    const int kc = kok ? k : 0;
    if constexpr (PRO == 2) {
      smask = 0;
      const int HW = a.st.H * a.st.W;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ke = kc + e, ci = ke / 9, t9 = ke - ci * 9, di = t9 / 3, dj = t9 - di * 3;
        const bool kv = ci < a.st.Cin;                      // k beyond Cin*9 is the weight matrix' zero padding
        const int cic = kv ? ci : 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int ih = sih[i] + di, iw = siw[i] + dj;
          const bool ok = kv && ih >= 0 && ih < a.st.H && iw >= 0 && iw < a.st.W;
          const int ihc = min(max(ih, 0), a.st.H - 1), iwc = min(max(iw, 0), a.st.W - 1);
          const float v = xrow[i][(size_t)cic * HW + (size_t)ihc * a.st.W + iwc];
          if (e == 0) ra[i].x = v; else if (e == 1) ra[i].y = v; else if (e == 2) ra[i].z = v; else ra[i].w = v;
          smask |= (ok ? 1u : 0u) << (i * 4 + e);
        }
      }
    } else if constexpr (PRO == 1) {
      kcur = kc;
      if (!a.bq_lds) bn_bwd_coef4(a.bb, kc, bq);
#pragma unroll
      for (int i = 0; i < NA; ++i) { ra[i] = mmd_ldw4(xrow[i], kc, a.x16); rg[i] = mmd_ldw4(grow[i], kc, a.z16); }
    } else if constexpr (PRO == 3) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = mmd_ldw4(xrow[i], kc, a.x16);
    } else if constexpr (PRO == 4) {
#pragma unroll
      for (int i = 0; i < NA; ++i) { ra[i] = mmd_ldw4(xrow[i], kc, a.x16); rg[i] = mmd_ld4(grow[i] + kc); }
    } else {
      if (a.in_bn.stats) bn_live_coef4(a.in_bn, kc, rsc, rsh);
      else if (a.in_scale) { rsc = mmd_ld4(a.in_scale + kc); rsh = mmd_ld4(a.in_shift + kc); }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ra[i] = mmd_ldw4(xrow[i], kc, a.x16);
        if (a.gate) rg[i] = mmd_ld4(grow[i] + kc);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = mmd_ld4(wrow[i] + kc);
  };
  // split mode: this thread's word offset inside a plane of its rows (rows lrow + 32 i share (row >> 2) & 3)
  const int wofs3 = (((((tid & 7) >> 1) ^ (lrow >> 2)) & 3) << 2) + ((tid & 1) << 1);
  auto lstore = [&]() {
    if constexpr (PRO == 1) { if (a.bq_lds) bn_bwd_tab4(sBqTab, a.K, kcur, bq); }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = ra[i];
      if constexpr (PRO == 2) {
        const unsigned mk = smask >> (i * 4);
        v.x = (mk & 1) ? v.x : 0.f; v.y = (mk & 2) ? v.y : 0.f; v.z = (mk & 4) ? v.z : 0.f; v.w = (mk & 8) ? v.w : 0.f;
      } else if constexpr (PRO == 1) {
        v = bn_bwd_eval4(v, rg[i], rowsc[i], a.bb.act, bq);
        if (a.bb.dz_out && tn == 0 && kok && rok[i]) mmd_stw4(a.bb.dz_out, (size_t)(m0 + lrow + i * 32) * a.K + kcur, v, a.dz16);
      } else if constexpr (PRO == 3) {
      } else if constexpr (PRO == 4) {
        v.x *= rg[i].x; v.y *= rg[i].y; v.z *= rg[i].z; v.w *= rg[i].w;
      } else {
      if (a.in_scale || a.in_bn.stats) {
        v.x = v.x * rsc.x + rsh.x; v.y = v.y * rsc.y + rsh.y; v.z = v.z * rsc.z + rsh.z; v.w = v.w * rsc.w + rsh.w;
      }
      if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
      if (a.gate) { v.x *= rg[i].x; v.y *= rg[i].y; v.z *= rg[i].z; v.w *= rg[i].w; }
      }
      // (round 6) No mask on the A tile.  k >= K: the weight tile is zero there and the A value comes from a clamped - valid, finite -
      // address, finite x 0 = 0; rows >= M: their outputs are neither stored nor summed.  16 v_cndmask per thread and K tile less.
      // (The stem gather, PRO == 2, masks per element above: its out-of-image taps are real zeros of in-range rows.)
      if constexpr (BF == 2) {
        split3_store<PW_BK / 2>(&sAu[(lrow + i * 32) * PW_LD3 + wofs3], v);
      } else if constexpr (BF) {
        // bf16 mode (round 6): the tile is stored as bf16 - rounded ONCE here (RNE, v_cvt_pk_bf16_f32) instead of by every lane in front of
        // every MFMA (12 converts + 6 ds_read_b128 per two MFMAs: the 16x faster pipe bought 24 %) - rows of PW_BK bf16 + 8 pad = 20 words
        *reinterpret_cast<uint2*>(&sAu[(lrow + i * 32) * PW_LDH + (kq >> 1)]) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
      } else {
        *reinterpret_cast<float4*>(&sA[(lrow + i * 32) * PW_LD + kq]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float4 w4 = (kok && wok[i]) ? rb[i] : make_float4(0, 0, 0, 0);
      if constexpr (BF == 2) split3_store<PW_BK / 2>(&sBu[(lrow + i * 32) * PW_LD3 + wofs3], w4);
      else if constexpr (BF) *reinterpret_cast<uint2*>(&sBu[(lrow + i * 32) * PW_LDH + (kq >> 1)]) = make_uint2(pk_bf16(w4.x, w4.y), pk_bf16(w4.z, w4.w));
      else *reinterpret_cast<float4*>(&sB[(lrow + i * 32) * PW_LD + kq]) = w4;
    }
  };

  const int nk = (a.K + PW_BK - 1) / PW_BK;
  gload(0);
  const float* const pa = &sA[(wm * 32 + r) * PW_LD + h * 4];
  const float* const pb = &sB[(wn * NS * 32 + r) * PW_LD + h * 4];
  auto mma = [&](int kk) {
    float4 av = *reinterpret_cast<const float4*>(pa + kk * 8);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      float4 bv = *reinterpret_cast<const float4*>(pb + j * 32 * PW_LD + kk * 8);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[j], 0, 0, 0);
    }
  };
  // lane (r, h) supplies row r, k = 16 g + 8 h .. + 7: eight bf16 = ONE 16-byte LDS read per operand (row stride 20 words: conflict-free
  // for the hardware's ds_read_b128 lane groups, tools/dev/lds_bank_model.py)
  const unsigned* const pa16 = &sAu[(wm * 32 + r) * PW_LDH + h * 4];
  const unsigned* const pb16 = &sBu[(wn * NS * 32 + r) * PW_LDH + h * 4];
  auto mma16 = [&](int g) {      // one 16-wide k group on the bf16 MFMA
    const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(pa16 + g * 8));
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const bf16x8 bv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(pb16 + j * 32 * PW_LDH + g * 8));
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
    }
  };
  // split mode: the three planes of both operands, six MFMAs per column slab - smallest partial products first
  const unsigned* const pa3 = &sAu[(wm * 32 + r) * PW_LD3];
  const unsigned* const pb3 = &sBu[(wn * NS * 32 + r) * PW_LD3];
  const int rofs3[2] = {((h ^ (r >> 2)) & 3) << 2, (((2 + h) ^ (r >> 2)) & 3) << 2};      // 16-deep group g, lane half h -> 4-word group 2 g + h, swizzled
  auto ldf = [](const unsigned* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(p)); };
  auto mma3 = [&](int g) {
    const bf16x8 ah = ldf(pa3 + rofs3[g]), am = ldf(pa3 + PW_BK / 2 + rofs3[g]), al = ldf(pa3 + PW_BK + rofs3[g]);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const unsigned* q = pb3 + j * 32 * PW_LD3 + rofs3[g];
      const bf16x8 bh = ldf(q), bm = ldf(q + PW_BK / 2), bl = ldf(q + PW_BK);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
    }
  };
  for (int kt = 0; kt < nk - 1; ++kt) {
    lstore();
    __syncthreads();
    gload((kt + 1) * PW_BK);
#ifdef MMD_PIN_PREFETCH      // (measured round 5: 14.74 vs 14.67 ms/step without - the other blocks on the CU hide the latency better than the pinned loads do: left off)
    // (dev: keep the next K tile's loads IN FRONT of this tile's MFMAs - left alone, hipcc sinks them behind the MFMAs in the register-lean
    // variants, where the staging registers double as fragment registers, so a block's K step pays the full load latency)
    if constexpr ((PRO == 3 || PRO == 4) && !(PRO == 4 && BM_T == 128 && BN_T == 64)) __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (BF == 2) {
#pragma unroll
      for (int g = 0; g < PW_BK / 16; ++g) mma3(g);
    } else if constexpr (BF) {
#pragma unroll
      for (int g = 0; g < PW_BK / 16; ++g) mma16(g);
    } else {
#pragma unroll
      for (int kk = 0; kk < PW_BK / 8; ++kk) mma(kk);
    }
    __syncthreads();
  }
  lstore();                      // last K tile: only its populated 8-wide groups (the tile is zero-filled beyond K)
  __syncthreads();
  if constexpr (BF == 2) {
#pragma unroll
    for (int g = 0; g < (NKL + 1) / 2; ++g) mma3(g);
  } else if constexpr (BF) {
#pragma unroll
    for (int g = 0; g < (NKL + 1) / 2; ++g) mma16(g);
  } else {
#pragma unroll
    for (int kk = 0; kk < NKL; ++kk) mma(kk);
  }
  __syncthreads();

  // ---- epilogue.  The accumulator (col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)) is staged through LDS so every
  // thread then owns 4 consecutive columns: dwordx4 stores / residual loads (4x fewer store instructions than
  // per-lane dword stores, which are issue-bound), per-row index math amortised over 4 values.
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      smem[(wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * LDC + (wn * NS + j) * 32 + r] = acc[j][q];
  __syncthreads();
  constexpr int CGN = BN_T / 4;            // column groups of 4
  constexpr int RSTEP = 256 / CGN;         // row groups
  const int cg = tid % CGN, rgrp = tid / CGN;
  const int col = n0 + cg * 4;
  const bool cok = col < a.N;              // N % 4 == 0: a column group is all-valid or all-out
  float4 b4 = make_float4(0, 0, 0, 0), osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  if (cok) {
    if (a.bias) b4 = mmd_ld4(a.bias + gw + col);
    if (a.out_scale) { osc = mmd_ld4(a.out_scale + gb + col); osh = mmd_ld4(a.out_shift + gb + col); }
  }
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
  float4 xmu = make_float4(0, 0, 0, 0), xis = make_float4(0, 0, 0, 0);
  if (a.xs.z && cok) { xmu = mmd_ld4(a.xs.mean + col); xis = mmd_ld4(a.xs.invstd + col); }
  float4 p5a[(PRO == 1) ? 5 : 1];
  P5Coef p5q;
  if constexpr (PRO == 1) {
#pragma unroll
    for (int k = 0; k < 5; ++k) p5a[k] = make_float4(0, 0, 0, 0);
    if (a.p5.z && cok) { p5q.sc = mmd_ld4(a.p5.scale + col); p5q.sh = mmd_ld4(a.p5.shift + col); p5q.mu = mmd_ld4(a.p5.mean + col); p5q.is = mmd_ld4(a.p5.invstd + col); }
  }
  // Round 6: every global load of the epilogue (residual, the z of the BnSumOp sums, the z1 of the pooled pass) is issued for ALL of the
  // thread's rows before the first use, from clamped (always valid) addresses.  Inside the per-row guard each of them was a dependent round
  // trip - `if (ok) v = ld` compiles to a branch with a full s_waitcnt vmcnt(0) (profiles/r04_notes.md sections 23 - 29) - i.e. 4 - 8 round
  // trips in sequence per block, longer than the K loop of the K = 88 .. 144 layers.
  // (in chunks of CH rows: all NI = 4 .. 8 rows at once cost the register-lean variants their occupancy - 16 - 24 spilled registers at the
  // 80-register bound)
  constexpr int NI = BM_T / RSTEP;
  constexpr int CH = 2;
  const int colc = cok ? col : 0;
#pragma unroll
  for (int c0 = 0; c0 < NI; c0 += CH) {
    size_t offs[CH];
    float4 rr[CH], xz[(PRO == 1) ? CH : 1], pz[(PRO == 1) ? CH : 1];
    float xrs[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int rowc = min(m0 + rgrp + RSTEP * (c0 + u), Mv - 1);
      if (a.y_batch_stride) {
        const int img = (rowc - srow0) / rpi;
        offs[u] = (size_t)img * a.y_batch_stride + yoff + (size_t)(rowc - srow0 - img * rpi) * a.N + colc;
      } else {
        offs[u] = (size_t)rowc * a.N + colc;
      }
      xrs[u] = 1.f;
      if (a.residual) rr[u] = mmd_ld4(a.residual + offs[u]);
      if constexpr (PRO == 1) {      // (the BnSumOp sums and the pooled pass only ride on the BatchNorm-backward operand launches)
        if (a.xs.z) { xz[u] = mmd_ld4(a.xs.z + offs[u]); if (a.xs.mul_b) xrs[u] = a.xs.mul_b[rowc / a.xs.rows_per_image]; }
        if (a.p5.z) pz[u] = mmd_ldw4(a.p5.z, offs[u], a.p5z16);
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int rl = rgrp + RSTEP * (c0 + u), row = m0 + rl;
      if (cok && row < Mv) {
        float4 v = *reinterpret_cast<const float4*>(&smem[rl * LDC + cg * 4]);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        if (a.stats && !a.xs.z) {      // (frozen nets take no statistics: 8 VALU instructions per quad less)
          s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
          q4.x += v.x * v.x; q4.y += v.y * v.y; q4.z += v.z * v.z; q4.w += v.w * v.w;
        }
        if (a.out_scale) { v.x = v.x * osc.x + osh.x; v.y = v.y * osc.y + osh.y; v.z = v.z * osc.z + osh.z; v.w = v.w * osc.w + osh.w; }
        if (a.out_act) mmd_act4(v, a.out_act);
        if (a.residual) { v.x += rr[u].x; v.y += rr[u].y; v.z += rr[u].z; v.w += rr[u].w; }
        mmd_stw4(a.y, offs[u], v, a.y16);
        if constexpr (PRO == 1) {
          if (a.xs.z) pw_xs_acc_pre(v, xz[u], xrs[u], xmu, xis, s4, q4);
          if (a.p5.z) pw_p5_acc_pre(p5q, v, pz[u], p5a);
        } else {
          if (a.xs.z) pw_xs_acc(a.xs, v, offs[u], row, xmu, xis, s4, q4);
        }
      }
    }
  }
  if constexpr (PRO == 1) {
    if (a.p5.z) {        // block-uniform; the tile's rows lie inside one image (host-checked)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float4 t = p5a[k];
#pragma unroll
        for (int o = CGN; o < 64; o <<= 1) {
          t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
        }
        if (lane < CGN) *reinterpret_cast<float4*>(&sRed5[(k * 4 + wave) * BN_T + lane * 4]) = t;
      }
      __syncthreads();
      const int img = m0 / a.p5.rows_per_image;
      for (int i = tid; i < 5 * BN_T; i += 256) {
        const int k = i / BN_T, cl = i - k * BN_T;
        if (n0 + cl < a.N) {
          const float* r5 = &sRed5[k * 4 * BN_T + cl];
          atomicAdd(&a.p5.out[((size_t)k * a.p5.B + img) * a.N + n0 + cl], r5[0] + r5[BN_T] + r5[2 * BN_T] + r5[3 * BN_T]);
        }
      }
    }
  }
  if (a.stats) {
    // lanes with equal (lane % CGN) share a column group: fold the row groups of a wave, then the 4 waves
#pragma unroll
    for (int o = CGN; o < 64; o <<= 1) {
      s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
      q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
    }
    if (lane < CGN) {
      *reinterpret_cast<float4*>(&sRed[wave * BN_T + lane * 4]) = s4;
      *reinterpret_cast<float4*>(&sRed[4 * BN_T + wave * BN_T + lane * 4]) = q4;
    }
    __syncthreads();
    if (tid < BN_T) {
      int c = n0 + tid;
      if (c < a.N) {
        float s = sRed[tid] + sRed[BN_T + tid] + sRed[2 * BN_T + tid] + sRed[3 * BN_T + tid];
        float ss = sRed[4 * BN_T + tid] + sRed[5 * BN_T + tid] + sRed[6 * BN_T + tid] + sRed[7 * BN_T + tid];
        atomicAdd(&stats[c], (double)s);
        atomicAdd(&stats[a.N + c], (double)ss);
      }
    }
  }
}

template <int BM_T, int BN_T, int NKL, int BF, int PRO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void pw_gemm_kernel(PwArgs a) {
  pw_gemm_body<BM_T, BN_T, NKL, BF, PRO>(a);
}
// register-lean operand modes (PRO 3 / 4) at a higher occupancy target (WAVES per SIMD)
template <int BM_T, int BN_T, int NKL, int BF, int PRO, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void pw_gemm_kernel_lean(PwArgs a) {
  pw_gemm_body<BM_T, BN_T, NKL, BF, PRO>(a);
}
template <int BM_T, int BN_T, int PRO, int WAVES>
static void (*pw_pick_lean(int nkl, int bf))(PwArgs) {
  // (split mode: 33 - 40 KB of tiles per 128-row block - four blocks per CU - so the register budget of four waves per SIMD there; the
  // 64 x 64 blocks' 26.6 KB leave six)
  constexpr int WS = BM_T == 64 ? WAVES : 4;
  if (bf == 2) return nkl == 1 ? pw_gemm_kernel_lean<BM_T, BN_T, 1, 2, PRO, WS> : nkl == 2 ? pw_gemm_kernel_lean<BM_T, BN_T, 2, 2, PRO, WS>
                    : nkl == 3 ? pw_gemm_kernel_lean<BM_T, BN_T, 3, 2, PRO, WS> : pw_gemm_kernel_lean<BM_T, BN_T, 4, 2, PRO, WS>;
  if (bf) return nkl == 1 ? pw_gemm_kernel_lean<BM_T, BN_T, 1, true, PRO, WAVES> : nkl == 2 ? pw_gemm_kernel_lean<BM_T, BN_T, 2, true, PRO, WAVES>
               : nkl == 3 ? pw_gemm_kernel_lean<BM_T, BN_T, 3, true, PRO, WAVES> : pw_gemm_kernel_lean<BM_T, BN_T, 4, true, PRO, WAVES>;
  return nkl == 1 ? pw_gemm_kernel_lean<BM_T, BN_T, 1, false, PRO, WAVES> : nkl == 2 ? pw_gemm_kernel_lean<BM_T, BN_T, 2, false, PRO, WAVES>
       : nkl == 3 ? pw_gemm_kernel_lean<BM_T, BN_T, 3, false, PRO, WAVES> : pw_gemm_kernel_lean<BM_T, BN_T, 4, false, PRO, WAVES>;
}

// BatchNorm-backward operand launches: dynamic LDS bytes of the per-block coefficient table (sets a.bq_lds), 0 when the table is off (not such
// a launch, MMD_NO_BQ_LDS, or the table would not fit beside `static_bytes` of the kernel's own tiles).
#include <unordered_set>
static size_t pw_bq_lds(PwArgs& a, const void* kern, size_t static_bytes) {
  static const int off = getenv("MMD_NO_BQ_LDS") ? 1 : 0;
  a.bq_lds = 0;
  if (!a.bb.z || off || (a.K & 3)) return 0;
  const size_t bytes = (size_t)5 * a.K * sizeof(float);
  // (the tiled kernels hide their latencies behind 3 - 4 co-resident blocks per CU: the table may not push a block beyond 48 KB)
  static const size_t cap = getenv("MMD_BQ_LDS_CAP_KB") ? (size_t)atoi(getenv("MMD_BQ_LDS_CAP_KB")) * 1024 : 48 * 1024;
  if (static_bytes + bytes > cap) return 0;
  if (static_bytes + bytes > 64 * 1024) {         // beyond the default per-block limit: raise it once per kernel
    static std::unordered_set<const void*> raised;
    if (!raised.count(kern)) { hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); raised.insert(kern); }
  }
  a.bq_lds = 1;
  return bytes;
}

// ------------------------------------------------------------------------------------------
// "Skinny" variant for layers whose 128x64 tiling gives too few blocks (deep 16x16 / 32x32 stages, BiFPN and
// head layers): block tile 32(M) x 64(N); the four waves split K (wave w owns k in [32w,32w+32) of every
// 128-wide K step), partial accumulators are summed through LDS and the epilogue is done row-major by all
// 256 threads (coalesced 256-B row segments).  4x more blocks and a 4x shorter serial MFMA chain per block.
#define SK_BM 32
#define SK_BN 64
#define SK_BK 128
#define SK_LD 132
#define SK_LDH 68      // bf16 mode: words per tile row (SK_BK / 2 data + 4 pad)

// -DMMD_KSTAMPS (dev build, tools/dev/skinny_phases.py): block 0 / thread 0 stamps the 100 MHz wall clock along the K loop
#ifdef MMD_KSTAMPS
__device__ unsigned long long g_kst[128];
#define MMD_KT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 128) g_kst[i] = wall_clock64(); } while (0)
extern "C" int mmd_k_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kst), sizeof(g_kst)) == hipSuccess ? 0 : -1; }
#else
#define MMD_KT(i)
#endif
// PF = register prefetch depth.  The skinny launches have at most one or two blocks per CU, so nothing but the block itself hides
// its load latency: with PF = 2 the loads of K steps t+1 and t+2 are in flight while step t is multiplied (two register stages,
// ~250 VGPRs - occupancy is irrelevant at these grid sizes).
// (split form - common.h - tried here in round 6 and not kept: a block re-splits its 64 x 128 weight tile at every K step, 2/3 of the staging
// VALU work, and the step measured 13.44 vs 13.44 ms with it.  With the weight tile copied from PRE-SPLIT bf16 planes instead - a per-store
// mmd_split3_planes pass, three 8-byte loads per quad - this kernel and the LDS-tiled ones measured SLOWER, step 13.56 vs 13.23 ms, layer
// by layer up to +16 %: three load instructions and 1.5x the bytes through the address path cost more than the 22 VALU instructions they
// save.  profiles/r06_notes.md section 10.)
template <bool BF, int PRO, int PF>
__global__ __launch_bounds__(256) void pw_gemm_skinny_kernel(PwArgs a) {
  __shared__ float sA[SK_BM * SK_LD];          // 16.5 KB
  __shared__ float sB[SK_BN * SK_LD];          // 33 KB ; reused as the 4 x [32][64] partial-sum buffer (32 KB)
  __shared__ float sRed[2 * 4 * SK_BN];
  __shared__ float sRed5[(PRO == 1) ? 5 * 4 * SK_BN : 1];
  unsigned* const sAu = reinterpret_cast<unsigned*>(sA);      // BF: bf16 tiles [row][SK_LDH words]
  unsigned* const sBu = reinterpret_cast<unsigned*>(sB);
  const int tid = threadIdx.x;
  const int t = mmd_xcd_swizzle(blockIdx.x, a.nblk);
  const int tn = t % a.ntn, tm = t / a.ntn;
  const int m0 = tm * SK_BM, n0 = tn * SK_BN;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  int Mv = a.M, srow0 = 0, rpi = a.rows_per_image; long long yoff = a.y_offset; double* stats = a.stats;
  if (a.pyr.n) {
    const int lev = pyr_level_of_row(a.pyr, m0);
    srow0 = a.pyr.row0[lev]; rpi = a.pyr.H[lev] * a.pyr.W[lev]; Mv = srow0 + a.pyr.B * rpi; yoff = a.yoff_lev[lev];
    if (stats) stats += 2 * lev * a.lev_stride;
  }
  const int kq = (tid & 31) * 4;          // k offset inside the 128-wide step
  const int lrow = tid >> 5;              // 0..7

  const float* xrow[4]; const float* grow[4]; bool rok[4]; float rowsc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = m0 + lrow + i * 8;
    rok[i] = row < Mv;
    int rr = rok[i] ? row : 0;
    xrow[i] = mmd_roww(a.x, rr, a.K, a.x16);
    if constexpr (PRO == 1) {
      grow[i] = mmd_roww(a.bb.z, rr, a.K, a.z16);
      rowsc[i] = a.bb.mul_b ? a.bb.mul_b[rr / a.bb.rows_per_image] : 1.f;
    } else if constexpr (PRO == 3) {
      grow[i] = nullptr;
    } else if constexpr (PRO == 4) {
      grow[i] = a.gate + (size_t)(rr / a.rows_per_image) * a.K;
    } else {
      grow[i] = a.gate ? a.gate + (size_t)(rr / a.rows_per_image) * a.K : nullptr;
    }
  }
  size_t gw = 0, gb = 0;      // grouped frozen nets: the tile's parameter set
  if (a.g_images) { const int gi = ((m0 - srow0) / rpi) / a.g_images; gw = (size_t)gi * a.g_w; gb = (size_t)gi * a.g_bn; }
  const float* wrow[8]; bool wok[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int col = n0 + lrow + i * 8;
    wok[i] = col < a.N;
    wrow[i] = a.w + gw + (size_t)(wok[i] ? col : 0) * a.K;
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
  extern __shared__ float sBqTabS[];      // PRO == 1 with a.bq_lds: [5][K] BatchNorm-backward coefficients (pw_args.h)
  if constexpr (PRO == 1) {
    if (a.bb.dgamma && t == 0)
      for (int c = tid; c < a.K; c += 256) { a.bb.dgamma[c] += (float)a.bb.sums[a.K + c]; a.bb.dbeta[c] += (float)a.bb.sums[c]; }
    if (a.bq_lds) { bn_bwd_tab_fill(a.bb, a.K, sBqTabS, tid, 256); __syncthreads(); }
  }

  struct Stage { float4 ra[4], rg[4], rb[8], rsc, rsh; BnBwdCoef4 bq; bool kok; int kc; };
  Stage st[PF];
  auto gload = [&](int k0, Stage& s) {
    // every load is unconditional on a clamped (always valid) address and masked afterwards: guarded loads compile to a
    // branch per load and a full vmcnt(0) drain, which serialises the prefetch
    int k = k0 + kq;
    s.kok = k < a.K;
    const int kc = s.kok ? k : 0;
    s.kc = kc;
    if constexpr (PRO == 1) {
      if (!a.bq_lds) bn_bwd_coef4(a.bb, kc, s.bq);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s.ra[i] = mmd_ldw4(xrow[i], kc, a.x16); s.rg[i] = mmd_ldw4(grow[i], kc, a.z16); }
    } else if constexpr (PRO == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s.ra[i] = mmd_ldw4(xrow[i], kc, a.x16);
    } else if constexpr (PRO == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { s.ra[i] = mmd_ldw4(xrow[i], kc, a.x16); s.rg[i] = mmd_ld4(grow[i] + kc); }
    } else {
      if (a.in_bn.stats) bn_live_coef4(a.in_bn, kc, s.rsc, s.rsh);
      else if (a.in_scale) { s.rsc = mmd_ld4(a.in_scale + kc); s.rsh = mmd_ld4(a.in_shift + kc); }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s.ra[i] = mmd_ldw4(xrow[i], kc, a.x16);
        if (a.gate) s.rg[i] = mmd_ld4(grow[i] + kc);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s.rb[i] = mmd_ld4(wrow[i] + kc);
  };
  auto lstore = [&](const Stage& s) {
    BnBwdCoef4 bqs;
    if constexpr (PRO == 1) { if (a.bq_lds) bn_bwd_tab4(sBqTabS, a.K, s.kc, bqs); else bqs = s.bq; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = s.ra[i];
      if constexpr (PRO == 1) {
        v = bn_bwd_eval4(v, s.rg[i], rowsc[i], a.bb.act, bqs);
        if (a.bb.dz_out && tn == 0 && s.kok && rok[i]) mmd_stw4(a.bb.dz_out, (size_t)(m0 + lrow + i * 8) * a.K + s.kc, v, a.dz16);
      } else if constexpr (PRO == 3) {
      } else if constexpr (PRO == 4) {
        v.x *= s.rg[i].x; v.y *= s.rg[i].y; v.z *= s.rg[i].z; v.w *= s.rg[i].w;
      } else {
      if (a.in_scale || a.in_bn.stats) {
        v.x = v.x * s.rsc.x + s.rsh.x; v.y = v.y * s.rsc.y + s.rsh.y; v.z = v.z * s.rsc.z + s.rsh.z; v.w = v.w * s.rsc.w + s.rsh.w;
      }
      if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
      if (a.gate) { v.x *= s.rg[i].x; v.y *= s.rg[i].y; v.z *= s.rg[i].z; v.w *= s.rg[i].w; }
      }
      // (no mask on the A tile: k >= K meets the zeroed weight tile, rows >= M are never stored - see pw_gemm_body)
      if constexpr (BF) *reinterpret_cast<uint2*>(&sAu[(lrow + i * 8) * SK_LDH + (kq >> 1)]) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
      else *reinterpret_cast<float4*>(&sA[(lrow + i * 8) * SK_LD + kq]) = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 w4 = (s.kok && wok[i]) ? s.rb[i] : make_float4(0, 0, 0, 0);
      if constexpr (BF) *reinterpret_cast<uint2*>(&sBu[(lrow + i * 8) * SK_LDH + (kq >> 1)]) = make_uint2(pk_bf16(w4.x, w4.y), pk_bf16(w4.z, w4.w));
      else *reinterpret_cast<float4*>(&sB[(lrow + i * 8) * SK_LD + kq]) = w4;
    }
  };
  auto mma = [&]() {
    if constexpr (BF) {
      // bf16 tiles (rounded once at the LDS store): one 16-byte read per operand and MFMA, no per-lane conversion
      const unsigned* pa = &sAu[r * SK_LDH + wave * 16 + h * 4];
      const unsigned* pb = &sBu[r * SK_LDH + wave * 16 + h * 4];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(pa + g * 8));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(pb + j * 32 * SK_LDH + g * 8));
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
        }
      }
    } else {
      const float* pa = &sA[r * SK_LD + wave * 32 + h * 4];
      const float* pb = &sB[r * SK_LD + wave * 32 + h * 4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float4 av = *reinterpret_cast<const float4*>(pa + kk * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 bv = *reinterpret_cast<const float4*>(pb + j * 32 * SK_LD + kk * 8);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[j], 0, 0, 0);
        }
      }
    }
  };

  const int nk = (a.K + SK_BK - 1) / SK_BK;
  MMD_KT(0);
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    if (s < nk) gload(s * SK_BK, st[s]);
    if constexpr (PF > 1) __builtin_amdgcn_sched_barrier(0);     // keep the stages' loads in issue order (the counted waits rely on it)
  }
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int kt = kt0 + s;
      if (kt < nk) {                               // block-uniform
#ifdef MMD_KSTAMPS
        __builtin_amdgcn_s_waitcnt(0); MMD_KT(64 + kt);      // (dev build only: the step's loads have landed)
#endif
        lstore(st[s]);
        MMD_KT(1 + 4 * kt);
        __syncthreads();
        MMD_KT(2 + 4 * kt);
        if (kt + PF < nk) gload((kt + PF) * SK_BK, st[s]);
        if constexpr (PF > 1) __builtin_amdgcn_sched_barrier(0);
        mma();
        MMD_KT(3 + 4 * kt);
        __syncthreads();
        MMD_KT(4 + 4 * kt);
      }
    }
  }
  // ---- cross-wave K reduction through LDS: part[wave][row][col], row-major 32 x 64
  MMD_KT(120);
  float* part = sB;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int row = (q & 3) + 8 * (q >> 2) + 4 * h;
      part[(wave * SK_BM + row) * SK_BN + j * 32 + r] = acc[j][q];
    }
  __syncthreads();
  // ---- epilogue: thread -> 4 consecutive columns (cg = tid & 15), rows (tid >> 4) + 16 i: dwordx4 stores
  const int cg = tid & 15, rgrp = tid >> 4, col = n0 + cg * 4;
  const bool cok = col < a.N;
  float4 b4 = make_float4(0, 0, 0, 0), osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  if (cok) {
    if (a.bias) b4 = mmd_ld4(a.bias + gw + col);
    if (a.out_scale) { osc = mmd_ld4(a.out_scale + gb + col); osh = mmd_ld4(a.out_shift + gb + col); }
  }
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
  float4 xmu = make_float4(0, 0, 0, 0), xis = make_float4(0, 0, 0, 0);
  if (a.xs.z && cok) { xmu = mmd_ld4(a.xs.mean + col); xis = mmd_ld4(a.xs.invstd + col); }
  float4 p5a[(PRO == 1) ? 5 : 1];
  P5Coef p5q;
  if constexpr (PRO == 1) {
#pragma unroll
    for (int k = 0; k < 5; ++k) p5a[k] = make_float4(0, 0, 0, 0);
    if (a.p5.z && cok) { p5q.sc = mmd_ld4(a.p5.scale + col); p5q.sh = mmd_ld4(a.p5.shift + col); p5q.mu = mmd_ld4(a.p5.mean + col); p5q.is = mmd_ld4(a.p5.invstd + col); }
  }
  // (round 6: the epilogue's global loads - residual, the sums' z, the pooled pass' z1 - for both of the thread's rows up front, from
  // clamped addresses: inside the per-row guard each was a dependent round trip)
  size_t offs[2];
  float4 rr[2], xz[2], pz[(PRO == 1) ? 2 : 1];
  float xrs[2];
  const int colc = cok ? col : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rowc = min(m0 + rgrp + 16 * i, Mv - 1);
    if (a.y_batch_stride) {
      const int img = (rowc - srow0) / rpi;
      offs[i] = (size_t)img * a.y_batch_stride + yoff + (size_t)(rowc - srow0 - img * rpi) * a.N + colc;
    } else {
      offs[i] = (size_t)rowc * a.N + colc;
    }
    xrs[i] = 1.f;
    if (a.residual) rr[i] = mmd_ld4(a.residual + offs[i]);
    if (a.xs.z) { xz[i] = mmd_ld4(a.xs.z + offs[i]); if (a.xs.mul_b) xrs[i] = a.xs.mul_b[rowc / a.xs.rows_per_image]; }
    if constexpr (PRO == 1) { if (a.p5.z) pz[i] = mmd_ldw4(a.p5.z, offs[i], a.p5z16); }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = rgrp + 16 * i, row = m0 + rl;
    float4 v = *reinterpret_cast<const float4*>(&part[rl * SK_BN + cg * 4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      float4 u = *reinterpret_cast<const float4*>(&part[(w * SK_BM + rl) * SK_BN + cg * 4]);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (cok && row < Mv) {
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (a.stats && !a.xs.z) {
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        q4.x += v.x * v.x; q4.y += v.y * v.y; q4.z += v.z * v.z; q4.w += v.w * v.w;
      }
      if (a.out_scale) { v.x = v.x * osc.x + osh.x; v.y = v.y * osc.y + osh.y; v.z = v.z * osc.z + osh.z; v.w = v.w * osc.w + osh.w; }
      if (a.out_act) mmd_act4(v, a.out_act);
      if (a.residual) { v.x += rr[i].x; v.y += rr[i].y; v.z += rr[i].z; v.w += rr[i].w; }
      mmd_stw4(a.y, offs[i], v, a.y16);
      if (a.xs.z) pw_xs_acc_pre(v, xz[i], xrs[i], xmu, xis, s4, q4);
      if constexpr (PRO == 1) { if (a.p5.z) pw_p5_acc_pre(p5q, v, pz[i], p5a); }
    }
  }
  if constexpr (PRO == 1) {
    if (a.p5.z) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float4 tq = p5a[k];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          tq.x += __shfl_xor(tq.x, o, 64); tq.y += __shfl_xor(tq.y, o, 64); tq.z += __shfl_xor(tq.z, o, 64); tq.w += __shfl_xor(tq.w, o, 64);
        }
        if (lane < 16) *reinterpret_cast<float4*>(&sRed5[(k * 4 + wave) * SK_BN + lane * 4]) = tq;
      }
      __syncthreads();
      const int img = m0 / a.p5.rows_per_image;
      for (int i = tid; i < 5 * SK_BN; i += 256) {
        const int k = i / SK_BN, cl = i - k * SK_BN;
        if (n0 + cl < a.N) {
          const float* r5 = &sRed5[k * 4 * SK_BN + cl];
          atomicAdd(&a.p5.out[((size_t)k * a.p5.B + img) * a.N + n0 + cl], r5[0] + r5[SK_BN] + r5[2 * SK_BN] + r5[3 * SK_BN]);
        }
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
      q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
    }
    if (lane < 16) {
      *reinterpret_cast<float4*>(&sRed[wave * SK_BN + lane * 4]) = s4;
      *reinterpret_cast<float4*>(&sRed[4 * SK_BN + wave * SK_BN + lane * 4]) = q4;
    }
    __syncthreads();
    if (tid < SK_BN && n0 + tid < a.N) {
      float s2 = sRed[tid] + sRed[SK_BN + tid] + sRed[2 * SK_BN + tid] + sRed[3 * SK_BN + tid];
      float q2 = sRed[4 * SK_BN + tid] + sRed[5 * SK_BN + tid] + sRed[6 * SK_BN + tid] + sRed[7 * SK_BN + tid];
      atomicAdd(&stats[n0 + tid], (double)s2);
      atomicAdd(&stats[a.N + n0 + tid], (double)q2);
    }
  }
  MMD_KT(121);
}

// ------------------------------------------------------------------------------------------
// Streaming variant for the HBM-bound layers (K <= 128, large M): a persistent block keeps its weight panel
// B[BN, K] resident in LDS for its whole life and walks M tiles with the WHOLE K extent of the A tile double-buffered
// (global loads of tile t+1 are in flight during the MFMAs and epilogue of tile t; one barrier per tile).  The A
// stream is read once per 128-wide column panel; BN statistics are accumulated in registers across tiles and
// leave as one atomic round per block.
//   WM x WN waves: (2 x 2): BM = 64, wave tile 32 x (BN/2), BN in {64, 128};  (4 x 1): BM = 128, BN = 32.
struct StArgs { PwArgs p; int KP; int LD; int mtiles; int gm; };   // KP = K rounded up to 8, LD = KP + 4, gm = blocks per column panel

template <int WM, int NSW>
__global__ __launch_bounds__(256) void pw_stream_kernel(StArgs sa) {
  constexpr int WN = 4 / WM;
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * NSW * WN;
  constexpr int NLD = 8;                         // max float4 A loads per thread: BM*KP/4/256 <= 128*128/4/256 = 16 for WM=4
  constexpr int NLA = (WM == 4) ? 16 : 8;
  extern __shared__ float smem[];
  const PwArgs& a = sa.p;
  const int KP = sa.KP, LD = sa.LD, f4row = KP >> 2;
  float* sB = smem;
  float* sA0 = sB + BN * LD;
  float* sA1 = sA0 + BM * LD;
  float* sSc = sA1 + BM * LD;                    // [KP] scale, [KP] shift
  float* sRed = sSc + 2 * KP;                    // [2][WM][BN]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const int tn = blockIdx.x % a.ntn, bm = blockIdx.x / a.ntn;
  const int n0 = tn * BN;
  const bool xf = a.in_scale || a.in_bn.stats;

  // ---- one-time: weight panel (zero-padded to [BN][KP]) and the prologue coefficients
  for (int i = tid; i < BN * f4row; i += 256) {
    int row = i / f4row, k = (i - row * f4row) * 4;
    float4 v = make_float4(0, 0, 0, 0);
    if (n0 + row < a.N && k < a.K) v = mmd_ld4(a.w + (size_t)(n0 + row) * a.K + k);
    *reinterpret_cast<float4*>(&sB[row * LD + k]) = v;
  }
  if (xf)
    for (int k = tid; k < KP; k += 256) {
      float sc = 0.f, sh = 0.f;
      if (k < a.K) {
        if (a.in_bn.stats) bn_live_coef(a.in_bn, k, sc, sh);
        else { sc = a.in_scale[k]; sh = a.in_shift[k]; }
      }
      sSc[k] = sc; sSc[KP + k] = sh;
    }
  // per-thread load slots (same pattern every tile)
  int lrow[NLA], lk[NLA];
  const int nld = (BM * f4row + 255) / 256;
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    int idx = tid + i * 256;
    lrow[i] = idx / f4row; lk[i] = (idx - lrow[i] * f4row) * 4;
    if (i >= nld || lrow[i] >= BM) lrow[i] = -1;
  }
  float4 ra[NLA], rg[NLA];
  auto gload = [&](int mt) {
    const int m0 = mt * BM;
    int Mv = a.M;
    if (a.pyr.n) { const int lev = pyr_level_of_row(a.pyr, m0); Mv = a.pyr.row0[lev] + a.pyr.B * a.pyr.H[lev] * a.pyr.W[lev]; }
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      ra[i] = make_float4(0, 0, 0, 0);
      if (lrow[i] >= 0) {
        int row = m0 + lrow[i];
        bool ok = row < Mv && lk[i] < a.K;
        if (ok) ra[i] = mmd_ld4(a.x + (size_t)row * a.K + lk[i]);
        if (a.gate) rg[i] = ok ? mmd_ld4(a.gate + (size_t)(row / a.rows_per_image) * a.K + lk[i]) : make_float4(0, 0, 0, 0);
        if (!ok) lrow[i] = -2 - lrow[i];          // mark invalid for this tile (restored in lstore)
      }
    }
  };
  auto lstore = [&](float* sA) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int lr = lrow[i];
      if (lr == -1) continue;
      bool ok = lr >= 0;
      if (!ok) { lr = -2 - lr; lrow[i] = lr; }
      float4 v = ra[i];
      if (ok) {
        if (xf) {
          float4 sc = *reinterpret_cast<const float4*>(&sSc[lk[i]]), sh = *reinterpret_cast<const float4*>(&sSc[KP + lk[i]]);
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
        if (a.gate) { v.x *= rg[i].x; v.y *= rg[i].y; v.z *= rg[i].z; v.w *= rg[i].w; }
      } else v = make_float4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(&sA[lr * LD + lk[i]]) = v;
    }
  };

  float cs[NSW], css[NSW];
#pragma unroll
  for (int j = 0; j < NSW; ++j) { cs[j] = 0.f; css[j] = 0.f; }
  double* stats0 = a.stats;          // level 0 base; per-level offset applied at flush time
  int cur_lev = -1;

  auto flush_stats = [&](int lev) {   // block-level reduce of the per-lane column sums and one atomic round
    if (!a.stats) return;
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
      float s = cs[j] + __shfl_xor(cs[j], 32, 64), q = css[j] + __shfl_xor(css[j], 32, 64);
      if (h == 0) { sRed[wm * BN + (wn * NSW + j) * 32 + r] = s; sRed[WM * BN + wm * BN + (wn * NSW + j) * 32 + r] = q; }
      cs[j] = 0.f; css[j] = 0.f;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { s += sRed[w * BN + tid]; q += sRed[WM * BN + w * BN + tid]; }
      double* st = stats0 + (a.pyr.n ? 2 * (long long)lev * a.lev_stride : 0);
      atomicAdd(&st[n0 + tid], (double)s);
      atomicAdd(&st[a.N + n0 + tid], (double)q);
    }
    __syncthreads();
  };

  int mt = bm;
  if (mt < sa.mtiles) gload(mt);
  __syncthreads();                       // sB / sSc visible
  if (mt < sa.mtiles) lstore(sA0);
  __syncthreads();
  int cur = 0;
  for (; mt < sa.mtiles; mt += sa.gm) {
    const int m0 = mt * BM;
    const int nxt = mt + sa.gm;
    if (nxt < sa.mtiles) gload(nxt);
    int Mv = a.M, srow0 = 0, rpi = a.rows_per_image, lev = 0; long long yoff = a.y_offset;
    if (a.pyr.n) {
      lev = pyr_level_of_row(a.pyr, m0);
      srow0 = a.pyr.row0[lev]; rpi = a.pyr.H[lev] * a.pyr.W[lev]; Mv = srow0 + a.pyr.B * rpi; yoff = a.yoff_lev[lev];
      if (a.stats && cur_lev >= 0 && lev != cur_lev) flush_stats(cur_lev);      // wave-uniform, all threads take it together
      cur_lev = lev;
    }
    const float* sA = cur ? sA1 : sA0;
    f32x16 acc[NSW];
#pragma unroll
    for (int j = 0; j < NSW; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    const float* pa = &sA[(wm * 32 + r) * LD + h * 4];
    const float* pb = &sB[(wn * NSW * 32 + r) * LD + h * 4];
    for (int kk = 0; kk < KP; kk += 8) {
      float4 av = *reinterpret_cast<const float4*>(pa + kk);
#pragma unroll
      for (int j = 0; j < NSW; ++j) {
        float4 bv = *reinterpret_cast<const float4*>(pb + j * 32 * LD + kk);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
      const int col = n0 + (wn * NSW + j) * 32 + r;
      const bool cok = col < a.N;
      const float bias = (a.bias && cok) ? a.bias[col] : 0.f;
      const float osc = (a.out_scale && cok) ? a.out_scale[col] : 1.f;
      const float osh = (a.out_scale && cok) ? a.out_shift[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (cok && row < Mv) {
          float v = acc[j][q] + bias;
          cs[j] += v; css[j] += v * v;
          if (a.out_scale) v = v * osc + osh;
          v = mmd_act(v, a.out_act);
          size_t off;
          if (a.y_batch_stride) {
            int img = (row - srow0) / rpi;
            off = (size_t)img * a.y_batch_stride + yoff + (size_t)(row - srow0 - img * rpi) * a.N + col;
          } else {
            off = (size_t)row * a.N + col;
          }
          if (a.residual) v += a.residual[off];
          a.y[off] = v;
        }
      }
    }
    if (nxt < sa.mtiles) lstore(cur ? sA0 : sA1);
    __syncthreads();
    cur ^= 1;
  }
  if (a.stats && (cur_lev >= 0 || !a.pyr.n) && bm < sa.mtiles) flush_stats(cur_lev < 0 ? 0 : cur_lev);
}

template <int WM, int NSW>
static int pw_stream_launch(PwArgs& a, hipStream_t stream) {
  constexpr int WN = 4 / WM, BM = 32 * WM, BN = 32 * NSW * WN;
  StArgs sa; sa.p = a;
  sa.KP = (a.K + 7) / 8 * 8; sa.LD = sa.KP + 4;
  sa.mtiles = cdiv(a.M, BM);
  sa.p.ntn = cdiv(a.N, BN);
  size_t lds = ((size_t)(BN + 2 * BM) * sa.LD + 2 * sa.KP + 2 * WM * BN) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)pw_stream_kernel<WM, NSW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  int per_cu = (int)((160 * 1024) / lds); if (per_cu > 4) per_cu = 4; if (per_cu < 1) per_cu = 1;
  int gm = (256 * per_cu) / sa.p.ntn; if (gm < 1) gm = 1; if (gm > sa.mtiles) gm = sa.mtiles;
  sa.gm = gm;
  hipLaunchKernelGGL((pw_stream_kernel<WM, NSW>), dim3(gm * sa.p.ntn), dim3(256), lds, stream, sa);
  return 0;
}

static int pw_dispatch(PwArgs& a, hipStream_t stream);

// (internal, called by mmd_stem_conv_fwd in elt.hip)  y[B*OH*OW, Cout] = im2col(x) * w[Cout, Kp]^T with the 1x1 kernel's epilogue
int mmd_pw_stem_gemm(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int OH, int OW, int pad_t, int pad_l,
                     int Kp, int Cout, const float* out_scale, const float* out_shift, int out_act, double* stats,
                     double* stats_ws, int ws_slots, hipStream_t stream);

template <int BM_T, int BN_T, int PRO>
static void (*pw_pick(int nkl, int bf))(PwArgs) {
  if (bf == 2) return nkl == 1 ? pw_gemm_kernel<BM_T, BN_T, 1, 2, PRO> : nkl == 2 ? pw_gemm_kernel<BM_T, BN_T, 2, 2, PRO>
                    : nkl == 3 ? pw_gemm_kernel<BM_T, BN_T, 3, 2, PRO> : pw_gemm_kernel<BM_T, BN_T, 4, 2, PRO>;
  if (bf) return nkl == 1 ? pw_gemm_kernel<BM_T, BN_T, 1, true, PRO> : nkl == 2 ? pw_gemm_kernel<BM_T, BN_T, 2, true, PRO>
               : nkl == 3 ? pw_gemm_kernel<BM_T, BN_T, 3, true, PRO> : pw_gemm_kernel<BM_T, BN_T, 4, true, PRO>;
  return nkl == 1 ? pw_gemm_kernel<BM_T, BN_T, 1, false, PRO> : nkl == 2 ? pw_gemm_kernel<BM_T, BN_T, 2, false, PRO>
       : nkl == 3 ? pw_gemm_kernel<BM_T, BN_T, 3, false, PRO> : pw_gemm_kernel<BM_T, BN_T, 4, false, PRO>;
}

static int pw_fwd_impl(const float* x, const float* w, float* y, int M, int K, int N,
                              const float* in_scale, const float* in_shift, int in_act,
                              const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count,
                              const float* gate, int rows_per_image,
                              const float* bias, const float* out_scale, const float* out_shift, int out_act,
                              const float* residual, double* stats,
                              long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots,
                              hipStream_t stream, int bf16, int w16 = 0, int form = MMD_PW_FORM_AUTO, float* ws = nullptr,
                              long long ws_floats = 0) {
  if (M <= 0 || K <= 0 || N <= 0 || (K & 3) || (N & 3) || !x || !w || !y) return MMD_EINVAL;
  if (form < 0 || (form & ~MMD_PW_FORM_NATIVE) > MMD_PW_FORM_SLAB) return MMD_EINVAL;
  if ((w16 & 2) && (residual || y_batch_stride)) return MMD_EINVAL;      // a bf16 output has no residual / strided form
  if ((gate || y_batch_stride) && rows_per_image <= 0) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MMD_EINVAL;
  if ((out_scale == nullptr) != (out_shift == nullptr)) return MMD_EINVAL;
  if (in_stats && (in_scale || !in_gamma || !in_beta || in_count <= 0)) return MMD_EINVAL;
  PwArgs a{x, w, y, M, K, N, in_scale, in_shift, in_act, mmd_make_bn(in_stats, in_gamma, in_beta, in_count, K), gate,
           rows_per_image > 0 ? rows_per_image : 1,
           bias, out_scale, out_shift, out_act, residual, stats, nullptr, 0, y_batch_stride, y_offset, 0, 0, Pyr{},
           {0, 0, 0, 0, 0}, 0, bf16};
  if (stats && stats_ws && ws_slots > 1 && cdiv(M, PW_BM) > MMD_STATS_DEPTH) { a.stats_ws = stats_ws; a.ws_slots = ws_slots; }
  a.x16 = w16 & 1; a.y16 = (w16 >> 1) & 1;
  a.form = form & ~MMD_PW_FORM_NATIVE; a.native = (form & MMD_PW_FORM_NATIVE) ? 1 : 0; a.slab_ws = ws; a.slab_ws_floats = ws ? ws_floats : 0;
  return pw_dispatch(a, stream);
}

#define PW_FWD_PARAMS const float* x, const float* w, float* y, int M, int K, int N, \
                      const float* in_scale, const float* in_shift, int in_act, \
                      const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count, \
                      const float* gate, int rows_per_image, \
                      const float* bias, const float* out_scale, const float* out_shift, int out_act, \
                      const float* residual, double* stats, \
                      long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots, hipStream_t stream
#define PW_FWD_ARGS x, w, y, M, K, N, in_scale, in_shift, in_act, in_stats, in_gamma, in_beta, in_count, gate, rows_per_image, \
                    bias, out_scale, out_shift, out_act, residual, stats, y_batch_stride, y_offset, stats_ws, ws_slots, stream
extern "C" int mmd_pwconv_fwd(PW_FWD_PARAMS) { return pw_fwd_impl(PW_FWD_ARGS, 0); }
// mmd_pwconv_fwd with the kernel family chosen by the caller (pw_args.h MMD_PW_FORM_*: 0 auto, 1 thin-K row-slab kernel, 2 LDS-tiled kernels only,
// 3 long-K LDS-DMA kernel, 4 all-N K-sliced slab kernel); a family that does not support the launch falls through to the LDS-tiled kernels.
// ws / ws_floats (nullable): workspace for the slab kernel's K slices (mmd_pwconv_slab_ws_floats); without one a launch that would need
// slices keeps the LDS-tiled kernels (form 0) or is refused (form 4)
extern "C" int mmd_pwconv_fwd_form(const float* x, const float* w, float* y, int M, int K, int N,
                                   const float* in_scale, const float* in_shift, int in_act,
                                   const double* in_stats, const float* in_gamma, const float* in_beta, long long in_count,
                                   const float* gate, int rows_per_image,
                                   const float* bias, const float* out_scale, const float* out_shift, int out_act,
                                   const float* residual, double* stats,
                                   long long y_batch_stride, long long y_offset, double* stats_ws, int ws_slots,
                                   float* ws, long long ws_floats, int form, hipStream_t stream) {
  return pw_fwd_impl(PW_FWD_ARGS, 0, 0, form, ws, ws_floats);
}
// same contract, operands rounded to bf16 at the MFMA input (fp32 accumulate, fp32 in/out tensors)
extern "C" int mmd_pwconv_fwd_bf16(PW_FWD_PARAMS) { return pw_fwd_impl(PW_FWD_ARGS, 1); }

int mmd_pw_stem_gemm(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int OH, int OW, int pad_t, int pad_l,
                     int Kp, int Cout, const float* out_scale, const float* out_shift, int out_act, double* stats,
                     double* stats_ws, int ws_slots, hipStream_t stream) {
  PwArgs a{};
  a.x = x; a.w = w; a.y = y; a.M = B * OH * OW; a.K = Kp; a.N = Cout; a.rows_per_image = 1;
  a.in_bn = mmd_make_bn(nullptr, nullptr, nullptr, 0, Kp);
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act; a.stats = stats;
  a.st = StemOp{Cin, H, W, OH, OW, pad_t, pad_l};
  const int ntm = cdiv(a.M, PW_BM);
  if (stats && stats_ws && ws_slots > 1 && ntm > MMD_STATS_DEPTH) { a.stats_ws = stats_ws; a.ws_slots = ws_slots; }
  a.ntn = cdiv(Cout, 32); a.nblk = ntm * a.ntn;
  const int nkl = ((Kp - 1) % PW_BK) / 8 + 1;
  hipLaunchKernelGGL((pw_pick<128, 32, 2>(nkl, 0)), dim3(a.nblk), dim3(256), 0, stream, a);
  if (a.stats_ws) mmd_stats_fold(a.stats, a.stats_ws, a.ws_slots, 2 * Cout, stream);
  return mmd_check_launch();
}

static int pw_dispatch(PwArgs& a, hipStream_t stream) {
  const int M = a.M, K = a.K, N = a.N;
  int ntm = cdiv(M, PW_BM);
  mmd_prof_tag(MMD_FAM_PW, "pw M%lld K%lld N%lld f%lld", M, K, N, (a.in_act ? 1 : 0) | (a.gate ? 2 : 0) | (a.stats ? 4 : 0) | (a.residual ? 8 : 0) | (a.out_scale ? 16 : 0) | (a.pyr.n ? 32 : 0));
  mmd_prof_begin(MMD_FAM_PW, stream);
  const long long big_tiles = (long long)ntm * cdiv(N, 64);
  static const int k_small = getenv("MMD_SKINNY_K") ? atoi(getenv("MMD_SKINNY_K")) : 0;
  // measured r01: the streaming variant is ~2x SLOWER than the tiled kernel on every shape (1 block/CU, one wave per
  // SIMD, nothing hides the LDS/epilogue latency) -> off unless MMD_STREAM=1; kept as the starting point for a
  // software-pipelined version (profiles/r01_notes.md)
  static const int use_stream = getenv("MMD_STREAM") ? 1 : 0;
  static const int skinny_tiles = getenv("MMD_SKINNY_TILES") ? atoi(getenv("MMD_SKINNY_TILES")) : 160;
  // 64x64 tiles (2x2 waves) for layers with sq_min <= big_tiles < sq_tiles
  static const int sq_tiles = getenv("MMD_SQ_TILES") ? atoi(getenv("MMD_SQ_TILES")) : 800;
  static const int sq_min = getenv("MMD_SQ_MIN") ? atoi(getenv("MMD_SQ_MIN")) : 160;
  const bool w16 = a.x16 || a.y16 || a.z16 || a.dz16 || a.p5z16;      // bf16 storage: the LDS-tiled kernels only
  // grouped frozen nets (common.h MmdGroup): plain forward launches only, on the LDS-tiled kernels; every group's rows must be whole tiles
  const MmdGroup& gr = mmd_group();
  const bool grouped = gr.n > 1;
  // which LDS-tiled kernel this launch takes (decided here: the grouped mode needs its row-tile height): 0 = skinny 32 x 64 (K split over
  // the waves), 1 = 128 x 32, 2 = 64 x 64, 3 = 128 x 64
  const bool take_skinny = (big_tiles < skinny_tiles || K <= k_small) && N > 16 && !(sq_tiles > 0 && big_tiles >= sq_min && big_tiles < sq_tiles && N > 32);
  static const int bn32_gain_env = getenv("MMD_BN32_GAIN") ? atoi(getenv("MMD_BN32_GAIN")) : 0;
  const int bn32_gain = bn32_gain_env ? bn32_gain_env : (a.bf16 ? 20 : 10);
  const int pad64 = cdiv(N, 64) * 64, pad32 = cdiv(N, 32) * 32;
  const int variant = take_skinny ? 0
                    : ((!a.bb.z && (N <= 32 || ((pad64 - pad32) * 100 > bn32_gain * N && !(sq_tiles > 0 && big_tiles < sq_tiles)))) || (a.bb.z && N <= 32)) ? 1
                    : ((sq_tiles > 0 && big_tiles < sq_tiles) || a.bb.z) ? 2 : 3;
  if (grouped) {
    if (w16 || a.bb.z || a.st.Cin || a.stats || a.in_bn.stats || a.in_scale) return MMD_EINVAL;
    const int imgs = gr.n * gr.images;
    // a row tile must not straddle two groups: every group's rows are whole tiles of the kernel taken (round 5: of THAT kernel - 32 rows on
    // the skinny one, 64 on the 64 x 64 one - so that the 6 x 6 level of D4 / 768^2, 288 rows per group at B = 8, packs too)
    const int bm = variant == 0 ? SK_BM : variant == 2 ? 64 : PW_BM;
    if (a.pyr.n) {
      if (a.pyr.B != imgs) return MMD_EINVAL;
      for (int l = 0; l < a.pyr.n; ++l) if (((long long)gr.images * a.pyr.H[l] * a.pyr.W[l]) % bm) return MMD_EINVAL;
    } else {
      if (a.rows_per_image <= 0 || (long long)imgs * a.rows_per_image != M || ((long long)gr.images * a.rows_per_image) % bm) return MMD_EINVAL;
    }
    a.g_images = gr.images; a.g_w = gr.w_stride; a.g_bn = gr.bn_stride;
  }
  int slab_rc = 0;
  if (!w16 && !grouped) {
    // all-N K-sliced slab kernel (pw_slab.hip, round 6): the launches the skinny kernel would run with an arithmetic prologue (BatchNorm
    // backward operand, or affine / swish / gate) and more than one column tile - there the prologue is re-evaluated per 64-wide tile
    // (measured per shape, profiles/r06_notes.md: the BatchNorm-backward operand launches gain 10 - 24 %; the forward form - affine / swish /
    // gate operand - only where N > 224 splits into column chunks, 58 vs 70 us, and is within +-5 % of the skinny kernel elsewhere)
    const bool slab_auto = take_skinny && !a.p5.z && N > 64 && K >= 256 &&
                           (a.bb.z || ((a.in_scale || a.in_bn.stats || a.in_act != MMD_ACT_NONE) && N > 224));
    slab_rc = pw_slab_try(a, a.slab_ws, a.slab_ws_floats, slab_auto, stream);
    if (slab_rc < 0) return slab_rc;
  }
  if (slab_rc == 1) {
    // taken
  } else if (!w16 && !grouped && pw_rows_try(a, stream) == 1) {
    // thin-K row-slab kernel (pw_rows.hip) took the launch
  } else if (!w16 && !grouped && pw_longk_try(a, stream) == 1) {
    // long-K small-M kernel with the LDS-DMA pipelined K loop (pw_longk.hip)
  } else if (use_stream && !w16 && !grouped && K <= 128 && big_tiles >= 160) {
    if (N <= 32) pw_stream_launch<4, 1>(a, stream);
    else if (N <= 64) pw_stream_launch<2, 1>(a, stream);
    else pw_stream_launch<2, 2>(a, stream);
  } else if (variant == 0) {
    a.ntn = cdiv(N, SK_BN); a.nblk = cdiv(M, SK_BM) * a.ntn;
    static const int sk_lean = getenv("MMD_NO_LEAN") ? 0 : 1;
    const bool noxf = sk_lean && !a.bb.z && !a.in_scale && !a.in_bn.stats && a.in_act == MMD_ACT_NONE;
    const int pro = a.bb.z ? 1 : noxf ? (a.gate ? 4 : 3) : 0;
    void (*sk)(PwArgs);
#define SK_PICK(P, F) (a.bf16 ? pw_gemm_skinny_kernel<true, P, F> : pw_gemm_skinny_kernel<false, P, F>)
    // (PF = 2 instantiates and runs, but hipcc drains both register stages before each LDS store - its waitcnt pass counts down to
    // vmcnt(0) across the per-lane masking branches - so it measured 1 % slower than PF = 1; see profiles/r01_notes.md)
    static const int sk_pf2 = getenv("MMD_SK_PF2") ? atoi(getenv("MMD_SK_PF2")) : 0;      // (dev: bit 0 = BatchNorm-backward operand launches, bit 1 = the others, with two register stages)
    sk = pro == 1 ? ((sk_pf2 & 1) ? SK_PICK(1, 2) : SK_PICK(1, 1)) : pro == 3 ? SK_PICK(3, 1) : pro == 4 ? SK_PICK(4, 1)
       : ((sk_pf2 & 2) ? SK_PICK(0, 2) : SK_PICK(0, 1));
#undef SK_PICK
    // (the per-block coefficient table of the BatchNorm-backward operand launches pays on the tiled kernels below - 25 launches, -8 % - and
    //  not here: 19 launches 641 -> 659 us per step; MMD_SK_BQ_LDS=1 switches it on)
    static const int sk_tab = getenv("MMD_SK_BQ_LDS") ? 1 : 0;
    const size_t bql = sk_tab ? pw_bq_lds(a, (const void*)sk, 58 * 1024) : 0;      // (sets a.bq_lds: before `a` is copied into the launch)
    hipLaunchKernelGGL(sk, dim3(a.nblk), dim3(256), bql, stream, a);
  } else {
    const int nkl = ((K - 1) % PW_BK) / 8 + 1;      // populated 8-wide groups of the last K tile
    const bool split_on = mmd_split_default() && !a.native;
    const int bfm = a.bf16 ? 1 : (split_on && !w16 && !a.st.Cin && K >= 64 && N > 48) ? 2 : 0;      // MFMA form of the tiles: 0 fp32, 1 bf16 operands, 2 fp32 by three-way bf16 split
    // plain A operand (no producer transform, no gate) or gate only: register-lean variants at a higher occupancy
    static const int lean_on = getenv("MMD_NO_LEAN") ? 0 : 1;
    const bool noxf = lean_on && !a.bb.z && !a.in_scale && !a.in_bn.stats && a.in_act == MMD_ACT_NONE;
    const bool plain = noxf && !a.gate, gated = noxf && a.gate;
    // 32-wide column tiles when they waste clearly fewer padded columns than 64-wide ones (N = 88, 144, 208, ...)
    // (bf16 operands: the A tile's conversion and L2 re-reads per column tile outweigh the padded MFMA work sooner - D4's N = 224 nodes on four
    // 64-wide tiles instead of seven 32-wide ones: config 5 49.9 -> 49.5 ms/step)
    void (*kern)(PwArgs);
    if (variant == 1) {
      // (BatchNorm-backward operand launches take the 128x32 variant only for the thin layers, N <= 32, where 64-wide tiles
      // would multiply 2-4x padding; it holds two VGPRs in scratch there)
      a.ntn = cdiv(N, 32);
      kern = a.bb.z ? pw_pick<128, 32, 1>(nkl, bfm) : plain ? pw_pick_lean<128, 32, 3, 6>(nkl, bfm)
           : gated ? pw_pick_lean<128, 32, 4, 4>(nkl, bfm) : pw_pick<128, 32, 0>(nkl, bfm);
    } else if (variant == 2) {      // 64x64 tiles: small-M layers (and every BatchNorm-
      // backward operand launch: its two-tensor prologue does not fit the 128-row variants' 128-VGPR budget)
      a.ntn = cdiv(N, 64); ntm = cdiv(M, 64);
      if (plain) kern = pw_pick_lean<64, 64, 3, 6>(nkl, bfm);
      else if (gated) kern = pw_pick_lean<64, 64, 4, 6>(nkl, bfm);
      else
      kern = a.bb.z ? pw_pick<64, 64, 1>(nkl, bfm) : pw_pick<64, 64, 0>(nkl, bfm);
    } else {
      a.ntn = cdiv(N, 64);
      kern = plain ? pw_pick_lean<128, 64, 3, 5>(nkl, bfm) : gated ? pw_pick_lean<128, 64, 4, 5>(nkl, bfm) : pw_pick<128, 64, 0>(nkl, bfm);
    }
    a.nblk = ntm * a.ntn;
    const size_t bql = pw_bq_lds(a, (const void*)kern, (bfm == 2 ? 37 : 28) * 1024);     // (sets a.bq_lds: before `a` is copied into the launch)
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(256), bql, stream, a);
  }
  if (a.stats_ws) mmd_stats_fold(a.stats, a.stats_ws, a.ws_slots, 2 * N, stream);
  // algorithmic bytes of the launch's contract: A, B, Y once each, plus what the fused prologue / epilogue jobs move by definition - the
  // second A tensor (z) and the stored dz of a BatchNorm-backward operand, a residual, the z of the output's BatchNorm sums, the z1 of
  // the pooled squeeze-excite pass
  const double a_elems = (double)M * K * (a.bb.z ? (a.bb.dz_out ? 3.0 : 2.0) : 1.0);
  const double y_elems = (double)M * N * (1.0 + (a.residual ? 1.0 : 0.0) + (a.xs.z ? 1.0 : 0.0) + (a.p5.z ? 1.0 : 0.0));
  mmd_prof_end(MMD_FAM_PW, stream, 2.0 * M * (double)K * N, 4.0 * (a_elems + y_elems + (double)N * K));
  return mmd_check_launch();
}

// Shared-weight head layer over a whole feature pyramid in one launch (see Pyr in common.h).  x, y: pyramid row
// buffers [row0[n], K] / [row0[n], N] (or, with y_batch_stride != 0, the concatenated [B, A_total, c] head output with
// per-level offsets y_off_lev).  stats (nullable): level l accumulates into stats + 2*l*lev_stride.
static int pw_fwd_pyr_impl(const float* x, const float* w, float* y, const int* pyr_desc, int K, int N,
                                  const float* bias, int out_act, double* stats, long long lev_stride,
                                  long long y_batch_stride, const long long* y_off_lev, hipStream_t stream, int bf16) {
  if (!x || !w || !y || !pyr_desc || K <= 0 || N <= 0 || (K & 3) || (N & 3)) return MMD_EINVAL;
  if (y_batch_stride && !y_off_lev) return MMD_EINVAL;
  PwArgs a{};
  if (mmd_make_pyr(a.pyr, pyr_desc)) return MMD_EINVAL;
  a.x = x; a.w = w; a.y = y; a.M = a.pyr.row0[a.pyr.n]; a.K = K; a.N = N; a.rows_per_image = 1;
  a.bias = bias; a.out_act = out_act; a.stats = stats; a.lev_stride = lev_stride; a.y_batch_stride = y_batch_stride;
  a.in_bn = mmd_make_bn(nullptr, nullptr, nullptr, 0, K);
  for (int l = 0; l < a.pyr.n; ++l) a.yoff_lev[l] = y_off_lev ? y_off_lev[l] : 0;
  a.bf16 = bf16;
  return pw_dispatch(a, stream);
}
extern "C" int mmd_pwconv_fwd_pyr(const float* x, const float* w, float* y, const int* pyr_desc, int K, int N,
                                  const float* bias, int out_act, double* stats, long long lev_stride,
                                  long long y_batch_stride, const long long* y_off_lev, hipStream_t stream) {
  return pw_fwd_pyr_impl(x, w, y, pyr_desc, K, N, bias, out_act, stats, lev_stride, y_batch_stride, y_off_lev, stream, 0);
}
extern "C" int mmd_pwconv_fwd_pyr_bf16(const float* x, const float* w, float* y, const int* pyr_desc, int K, int N,
                                       const float* bias, int out_act, double* stats, long long lev_stride,
                                       long long y_batch_stride, const long long* y_off_lev, hipStream_t stream) {
  return pw_fwd_pyr_impl(x, w, y, pyr_desc, K, N, bias, out_act, stats, lev_stride, y_batch_stride, y_off_lev, stream, 1);
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[N,K] += sum_m dY[m,n] * pro(X)[m,k]; output tile 64(n) x 64(k) per block,
// each of the 4 waves owns a 32x32 sub-tile; the reduction over M is split across blocks
// (mchunk rows each) and combined with fp32 atomics (256-B contiguous per wave-instruction).
struct WgArgs {
  const float* dy; const float* x; float* dw;
  int M, K, N;
  const float* in_scale; const float* in_shift; int in_act;
  const float* gate; int rows_per_image;
  int mchunk; int ntn; int ntk; int nblk;
  BnBwdOp bb; float* dgamma; float* dbeta;      // BNP: dy is BnBwd(dy = g, bb.z) evaluated on the fly; dgamma/dbeta (+)= from bb.sums
  int dy16, x16;                                // bf16 storage of dy / x (common.h w16)
};
#define WG_LD 68
#ifndef WG_BR
#define WG_BR 32          // rows of the M reduction per step (64 measured no faster)
#endif

template <bool BF, bool BNP, int PF>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgArgs a) {
  __shared__ float sD[WG_BR * WG_LD];
  __shared__ float sX[WG_BR * WG_LD];
  const int tid = threadIdx.x;
  // XCD-aware order: the ntn*ntk tiles that re-read one M slab sit on ONE XCD (one L2), slabs are dealt to the XCDs in
  // contiguous runs; the grid is padded to a multiple of 8 (the surplus blocks leave)
  int b = mmd_xcd_swizzle(blockIdx.x, gridDim.x);
  if (b >= a.nblk) return;
  const int tk = b % a.ntk; b /= a.ntk;
  const int tn = b % a.ntn; b /= a.ntn;
  const int mbeg = b * a.mchunk;
  const int mend = min(a.M, mbeg + a.mchunk);
  const int n0 = tn * 64, k0 = tk * 64;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const int c4 = (tid & 15) * 4;         // column offset inside the 64-wide tile
  const int lrow = tid >> 4;             // 0..15
  const bool nok = (n0 + c4) < a.N, kok = (k0 + c4) < a.K;
  float4 xsc = make_float4(1, 1, 1, 1), xsh = make_float4(0, 0, 0, 0);
  if (a.in_scale && kok) { xsc = mmd_ld4(a.in_scale + k0 + c4); xsh = mmd_ld4(a.in_shift + k0 + c4); }
  BnBwdCoef4 bq;
  if constexpr (BNP) {
    bn_bwd_coef4(a.bb, nok ? n0 + c4 : 0, bq);       // this thread's 4 dY columns are fixed for the whole M loop
    if (tk == 0 && b == 0 && lrow == 0 && nok && a.dgamma) {      // BatchNorm affine gradients straight from the reduce pass' sums
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a.dgamma[n0 + c4 + i] += (float)a.bb.sums[a.bb.C + n0 + c4 + i];
        a.dbeta[n0 + c4 + i] += (float)a.bb.sums[n0 + c4 + i];
      }
    }
  }

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;

  constexpr int NL = WG_BR / 16;      // float4 loads per thread and operand
  // PF register stages: the loads of the next PF row steps are in flight while one step is multiplied.  A block needs
  // PF * (load latency / MFMA time per step) >= 1 to stay MFMA-bound on its own, and few blocks mean few atomics at the
  // end (N*K fp32 atomics per block at ~1.3 TB/s chip-wide are what bounded the many-small-blocks form of this kernel).
  struct Stage { float4 rd[NL], rx[NL], rg[NL], rz[NL]; float rrs[NL]; bool rok[NL]; };
  Stage st[PF];
  auto gload = [&](int mb, Stage& s) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int row = mb + lrow + i * 16;
      s.rok[i] = row < mend;
      const int rc = s.rok[i] ? row : mbeg;                     // clamped: loads are unconditional, masked in lstore
      s.rd[i] = mmd_ldw4(a.dy, (size_t)rc * a.N + (nok ? n0 + c4 : 0), a.dy16);
      if constexpr (BNP) {
        s.rz[i] = mmd_ld4(a.bb.z + (size_t)rc * a.N + (nok ? n0 + c4 : 0));
        s.rrs[i] = a.bb.mul_b ? a.bb.mul_b[rc / a.bb.rows_per_image] : 1.f;
      }
      s.rx[i] = mmd_ldw4(a.x, (size_t)rc * a.K + (kok ? k0 + c4 : 0), a.x16);
      if (a.gate) s.rg[i] = mmd_ld4(a.gate + (size_t)(rc / a.rows_per_image) * a.K + (kok ? k0 + c4 : 0));
    }
  };
  auto lstore = [&](const Stage& s) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float4 v = s.rx[i];
      if (a.in_scale) {
        v.x = v.x * xsc.x + xsh.x; v.y = v.y * xsc.y + xsh.y; v.z = v.z * xsc.z + xsh.z; v.w = v.w * xsc.w + xsh.w;
      }
      if (a.in_act == MMD_ACT_SWISH) { v.x = mmd_swish(v.x); v.y = mmd_swish(v.y); v.z = mmd_swish(v.z); v.w = mmd_swish(v.w); }
      if (a.gate) { v.x *= s.rg[i].x; v.y *= s.rg[i].y; v.z *= s.rg[i].z; v.w *= s.rg[i].w; }
      if (!(s.rok[i] && kok)) v = make_float4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(&sX[(lrow + i * 16) * WG_LD + c4]) = v;
      float4 d = s.rd[i];
      if constexpr (BNP) d = bn_bwd_eval4(d, s.rz[i], s.rrs[i], a.bb.act, bq);
      *reinterpret_cast<float4*>(&sD[(lrow + i * 16) * WG_LD + c4]) = (s.rok[i] && nok) ? d : make_float4(0, 0, 0, 0);
    }
  };
  const bool idle = n0 + wn * 32 >= a.N || k0 + wk * 32 >= a.K;      // a wave whose 32x32 sub-tile is all N / K padding (thin layers)
  auto mma = [&]() {
    const float* pd = &sD[h * WG_LD + wn * 32 + r];
    const float* px = &sX[h * WG_LD + wk * 32 + r];
    if (idle) {
    } else if constexpr (BF) {
      // the reduction runs over rows: lane (r, h) gathers rows 16g + 8h .. + 7 of its column (8 ds_read_b32 per operand)
#pragma unroll
      for (int g = 0; g < WG_BR / 16; ++g) {
        const float* qd = &sD[(g * 16 + h * 8) * WG_LD + wn * 32 + r];
        const float* qx = &sX[(g * 16 + h * 8) * WG_LD + wk * 32 + r];
        const bf16x8 dv = pack_bf16x8(make_float4(qd[0], qd[WG_LD], qd[2 * WG_LD], qd[3 * WG_LD]),
                                      make_float4(qd[4 * WG_LD], qd[5 * WG_LD], qd[6 * WG_LD], qd[7 * WG_LD]));
        const bf16x8 xv = pack_bf16x8(make_float4(qx[0], qx[WG_LD], qx[2 * WG_LD], qx[3 * WG_LD]),
                                      make_float4(qx[4 * WG_LD], qx[5 * WG_LD], qx[6 * WG_LD], qx[7 * WG_LD]));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv, xv, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < WG_BR / 2; ++tt)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[tt * 2 * WG_LD], px[tt * 2 * WG_LD], acc, 0, 0, 0);
    }
  };

#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (mbeg + s * WG_BR < mend) gload(mbeg + s * WG_BR, st[s]);
  for (int mb = mbeg; mb < mend; mb += PF * WG_BR) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int m = mb + s * WG_BR;
      if (m < mend) {                              // block-uniform
        lstore(st[s]);
        __syncthreads();
        if (m + PF * WG_BR < mend) gload(m + PF * WG_BR, st[s]);
        mma();
        __syncthreads();
      }
    }
  }
  const int kcol = k0 + wk * 32 + r;
  if (kcol < a.K) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int n = n0 + wn * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (n < a.N) atomicAdd(&a.dw[(size_t)n * a.K + kcol], acc[q]);
    }
  }
}

static int pw_wgrad_impl(const float* dy, const float* x, float* dw, int M, int K, int N,
                                     const float* in_scale, const float* in_shift, int in_act,
                                     const float* gate, int rows_per_image, hipStream_t stream, int bf16,
                                     const BnBwdOp* bb = nullptr, float* dgamma = nullptr, float* dbeta = nullptr, int w16 = 0) {
  if (M <= 0 || K <= 0 || N <= 0 || (K & 3) || (N & 3) || !dy || !x || !dw) return MMD_EINVAL;
  if (gate && rows_per_image <= 0) return MMD_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MMD_EINVAL;
  WgArgs a{dy, x, dw, M, K, N, in_scale, in_shift, in_act, gate, rows_per_image > 0 ? rows_per_image : 1, 0, 0, 0, 0, BnBwdOp{}, nullptr, nullptr};
  if (bb) { a.bb = *bb; a.dgamma = dgamma; a.dbeta = dbeta; }
  a.dy16 = w16 & 1; a.x16 = (w16 >> 1) & 1;
  a.ntn = cdiv(N, 64); a.ntk = cdiv(K, 64);
  int tiles = a.ntn * a.ntk;
  // enough blocks to fill 256 CUs a few times over, but every split ends in N*K fp32 atomics (1.3 TB/s chip-wide):
  // keep at least 256 rows per split so the atomic traffic stays well below the streamed bytes
  static const int wg_blocks = getenv("MMD_WG_BLOCKS") ? atoi(getenv("MMD_WG_BLOCKS")) : 1024;
  int splits = wg_blocks / tiles; if (splits < 1) splits = 1;
  // small M (the 16x16-and-below BiFPN levels): 64-row splits - a handful of blocks walking 256 rows each is a chain of exposed
  // load latencies, and N*K atomics per split are cheap next to that
  int maxs = cdiv(M, M <= 8192 ? 64 : 256); if (splits > maxs) splits = maxs;
  a.mchunk = cdiv(cdiv(M, splits), WG_BR) * WG_BR;
  splits = cdiv(M, a.mchunk);
  mmd_prof_tag(MMD_FAM_PW_WGRAD, "wg M%lld K%lld N%lld s%lld", M, K, N, splits);
  mmd_prof_begin(MMD_FAM_PW_WGRAD, stream);
  // PF = 1: deeper register prefetch (2, 4 stages) with 256-512 blocks measured no faster (1.89 -> 1.99-2.27 ms weighted): the
  // launch is base (~8 us: launch, first load, staging + barriers) + MFMA (~12-14 us) + atomics, added up rather than overlapped
  void (*wk)(WgArgs) = bb ? (bf16 ? pw_wgrad_kernel<true, true, 1> : pw_wgrad_kernel<false, true, 1>)
                          : (bf16 ? pw_wgrad_kernel<true, false, 1> : pw_wgrad_kernel<false, false, 1>);
  a.nblk = tiles * splits;
  hipLaunchKernelGGL(wk, dim3((a.nblk + 7) / 8 * 8), dim3(256), 0, stream, a);
  mmd_prof_end(MMD_FAM_PW_WGRAD, stream, 2.0 * M * (double)K * N, 4.0 * ((double)M * K + (double)M * N + (double)N * K));
  return mmd_check_launch();
}
extern "C" int mmd_pwconv_bwd_weight(const float* dy, const float* x, float* dw, int M, int K, int N,
                                     const float* in_scale, const float* in_shift, int in_act,
                                     const float* gate, int rows_per_image, hipStream_t stream) {
  return pw_wgrad_impl(dy, x, dw, M, K, N, in_scale, in_shift, in_act, gate, rows_per_image, stream, 0);
}
// dW[N,K] += BnBwd(g, z)^T * pro(X): weight gradient of a 1x1 conv in front of a BatchNorm, with the BatchNorm backward evaluated
// on the dY operand while staging; dgamma / dbeta (nullable) (+)= [sum g'*xhat, sum g'] taken from `sums`.
static int pw_wgrad_bn_impl(const float* g, const float* z, const float* x, float* dw, int M, int K, int N,
                            const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image,
                            const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums,
                            long long count, int act, const float* mul_b, int bn_rows_per_image, float* dgamma, float* dbeta,
                            hipStream_t stream, int bf16) {
  if (!z || !scale || !shift || !mean || !invstd || !sums || count <= 0) return MMD_EINVAL;
  if (mul_b && bn_rows_per_image <= 0) return MMD_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  BnBwdOp bb{z, scale, shift, mean, invstd, sums, 1.0 / (double)count, N, act, mul_b, bn_rows_per_image > 0 ? bn_rows_per_image : 1,
             nullptr, nullptr, nullptr};
  return pw_wgrad_impl(g, x, dw, M, K, N, in_scale, in_shift, in_act, gate, rows_per_image, stream, bf16, &bb, dgamma, dbeta);
}
#define PW_WGBN_PARAMS const float* g, const float* z, const float* x, float* dw, int M, int K, int N, \
                       const float* in_scale, const float* in_shift, int in_act, const float* gate, int rows_per_image, \
                       const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums, \
                       long long count, int act, const float* mul_b, int bn_rows_per_image, float* dgamma, float* dbeta, hipStream_t stream
#define PW_WGBN_ARGS g, z, x, dw, M, K, N, in_scale, in_shift, in_act, gate, rows_per_image, scale, shift, mean, invstd, sums, count, act, \
                     mul_b, bn_rows_per_image, dgamma, dbeta, stream
extern "C" int mmd_pwconv_bwd_weight_bn(PW_WGBN_PARAMS) { return pw_wgrad_bn_impl(PW_WGBN_ARGS, 0); }
extern "C" int mmd_pwconv_bwd_weight_bn_bf16(PW_WGBN_PARAMS) { return pw_wgrad_bn_impl(PW_WGBN_ARGS, 1); }
extern "C" int mmd_pwconv_bwd_weight_bf16(const float* dy, const float* x, float* dw, int M, int K, int N,
                                          const float* in_scale, const float* in_shift, int in_act,
                                          const float* gate, int rows_per_image, hipStream_t stream) {
  return pw_wgrad_impl(dy, x, dw, M, K, N, in_scale, in_shift, in_act, gate, rows_per_image, stream, 1);
}

// dX[M,K] = dY[M,N] * W[N,K]: same MFMA kernel with the transposed weight copy Wt[K,N]
// (kept per layer by the engine and refreshed after each optimizer step, see mmd_transpose2d).
extern "C" int mmd_pwconv_bwd_data(const float* dy, const float* wt, float* dx, int M, int K, int N,
                                   int accumulate, hipStream_t stream) {
  // Y'=dx [M,K], X'=dy [M,N], W'=wt [K,N] -> reduction dim is N
  return mmd_pwconv_fwd(dy, wt, dx, M, /*K=*/N, /*N=*/K, nullptr, nullptr, MMD_ACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0,
                        nullptr, nullptr, nullptr, MMD_ACT_NONE, accumulate ? dx : nullptr, nullptr, 0, 0, nullptr, 0, stream);
}
// dX[M,K] = BnBwd(g, z)[M,N] * W[N,K]: input gradient of a 1x1 conv whose output goes through BatchNorm (+swish, +per-image
// drop-connect scale), with the BatchNorm backward evaluated in the operand prologue (see BnBwdOp).  sums = [sum g', sum g'*xhat].
static int pw_bwd_data_bn_impl(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N,
                               const float* scale, const float* shift, const float* mean, const float* invstd,
                               const double* sums, long long count, int act, const float* mul_b, int rows_per_image,
                               float* dz_out, float* dgamma, float* dbeta, hipStream_t stream, int bf16) {
  if (M <= 0 || K <= 0 || N <= 0 || (K & 3) || (N & 3) || !g || !z || !wt || !dx) return MMD_EINVAL;
  if (!scale || !shift || !mean || !invstd || !sums || count <= 0) return MMD_EINVAL;
  if (mul_b && rows_per_image <= 0) return MMD_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  PwArgs a{};
  a.x = g; a.w = wt; a.y = dx; a.M = M; a.K = N; a.N = K; a.rows_per_image = 1;
  a.in_bn = mmd_make_bn(nullptr, nullptr, nullptr, 0, N);
  a.bf16 = bf16;
  a.bb = BnBwdOp{z, scale, shift, mean, invstd, sums, 1.0 / (double)count, N, act, mul_b, rows_per_image > 0 ? rows_per_image : 1,
                 dz_out, dgamma, dbeta};
  return pw_dispatch(a, stream);
}
// Same launch with two more jobs for its epilogue (both optional):
//   residual: dx = BnBwd(g, z) * W + residual   (residual may be dx itself: in-place accumulation into a gradient that already holds
//             the skip branch's / other consumers' contributions);
//   xs_*:     dx is then the COMPLETE gradient w.r.t. a tensor BN'(xs_z) * xs_mul_b[image] (+ skip), and xs_sums [2K] (+)=
//             [sum g', sum g' * xhat'], g' = dx * xs_mul_b[image], xhat' = (xs_z - xs_mean) * xs_invstd: the reduce pass of that upstream
//             BatchNorm's backward (mmd_bn_bwd_reduce with act = NONE) without a launch of its own.  stats_ws / ws_slots as in mmd_pwconv_fwd.
//   p5_*:     (MBConv project conv; not together with residual / xs) dx = g1 is the gradient w.r.t. the squeeze-excite-gated activation
//             a1 * gate: p5_out [5][p5_B][K] (+)= mmd_chan_pool_bwd(p5_z = z1, ..., g1) - the pooled pass of the squeeze-excite /
//             BatchNorm-1 backward - from the output tiles (by its own launch when an image's rows are not a multiple of 128).
extern "C" int mmd_chan_pool_bwd(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd,
                                 const float* g1, float* out5, int B, int rows_per_image, int C, hipStream_t stream);      // elt.hip
static int pw_bwd_data_bn2_impl(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                const double* sums, long long count, int act, const float* mul_b, int rows_per_image,
                                float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z,
                                const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image,
                                double* xs_sums, double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale,
                                const float* p5_shift, const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B,
                                hipStream_t stream, int bf16, int w16 = 0, int form = MMD_PW_FORM_AUTO, float* ws = nullptr,
                                long long ws_floats = 0) {
  if (M <= 0 || K <= 0 || N <= 0 || (K & 3) || (N & 3) || !g || !z || !wt || !dx) return MMD_EINVAL;
  if (!scale || !shift || !mean || !invstd || !sums || count <= 0) return MMD_EINVAL;
  if (mul_b && rows_per_image <= 0) return MMD_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MMD_EINVAL;
  if (xs_z && (!xs_mean || !xs_invstd || !xs_sums || (xs_mul_b && xs_rows_per_image <= 0))) return MMD_EINVAL;
  PwArgs a{};
  a.x = g; a.w = wt; a.y = dx; a.M = M; a.K = N; a.N = K; a.rows_per_image = 1;
  a.in_bn = mmd_make_bn(nullptr, nullptr, nullptr, 0, N);
  a.bf16 = bf16;
  a.residual = residual;
  a.bb = BnBwdOp{z, scale, shift, mean, invstd, sums, 1.0 / (double)count, N, act, mul_b, rows_per_image > 0 ? rows_per_image : 1,
                 dz_out, dgamma, dbeta};
  a.x16 = w16 & 1; a.y16 = (w16 >> 1) & 1; a.z16 = (w16 >> 2) & 1; a.dz16 = (w16 >> 3) & 1; a.p5z16 = (w16 >> 4) & 1;
  if (form < 0 || (form & ~MMD_PW_FORM_NATIVE) > MMD_PW_FORM_SLAB) return MMD_EINVAL;
  a.form = form & ~MMD_PW_FORM_NATIVE; a.native = (form & MMD_PW_FORM_NATIVE) ? 1 : 0; a.slab_ws = ws; a.slab_ws_floats = ws ? ws_floats : 0;
  if (a.y16 && (residual || xs_z)) return MMD_EINVAL;
  if (xs_z) {
    a.xs = BnSumOp{xs_z, xs_mean, xs_invstd, xs_mul_b, xs_rows_per_image > 0 ? xs_rows_per_image : 1};
    a.stats = xs_sums;
    // (the BatchNorm-backward operand launches run 64-row tiles, or 128-row ones for N <= 32: count the blocks per address with 64)
    if (stats_ws && ws_slots > 1 && cdiv(M, 64) > MMD_STATS_DEPTH) { a.stats_ws = stats_ws; a.ws_slots = ws_slots; }
  }
  if (p5_z) {
    if (!p5_scale || !p5_shift || !p5_mean || !p5_invstd || !p5_out || p5_B <= 0 || M % p5_B || residual || xs_z) return MMD_EINVAL;
    const int rpi = M / p5_B;
    if (rpi % PW_BM == 0) {       // every row tile (32 / 64 / 128 rows) inside one image: the sums ride in the epilogue
      a.p5 = Pool5Op{p5_z, p5_scale, p5_shift, p5_mean, p5_invstd, p5_out, p5_B, rpi};
      return pw_dispatch(a, stream);
    }
    if (a.y16 || a.p5z16) return MMD_EINVAL;      // (the stand-alone pooled pass reads fp32 tensors)
    const int rc = pw_dispatch(a, stream);      // ragged image size: the pooled pass as its own launch
    return rc ? rc : mmd_chan_pool_bwd(p5_z, p5_scale, p5_shift, p5_mean, p5_invstd, dx, p5_out, p5_B, rpi, K, stream);
  }
  return pw_dispatch(a, stream);
}
#define PW_BD2_PARAMS const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift, \
                      const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b, \
                      int rows_per_image, float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z, \
                      const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums, \
                      double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale, const float* p5_shift, \
                      const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B, hipStream_t stream
#define PW_BD2_ARGS g, z, wt, dx, M, K, N, scale, shift, mean, invstd, sums, count, act, mul_b, rows_per_image, dz_out, dgamma, dbeta, residual, \
                    xs_z, xs_mean, xs_invstd, xs_mul_b, xs_rows_per_image, xs_sums, stats_ws, ws_slots, p5_z, p5_scale, p5_shift, p5_mean, \
                    p5_invstd, p5_out, p5_B, stream
extern "C" int mmd_pwconv_bwd_data_bn2(PW_BD2_PARAMS) { return pw_bwd_data_bn2_impl(PW_BD2_ARGS, 0); }
extern "C" int mmd_pwconv_bwd_data_bn2_bf16(PW_BD2_PARAMS) { return pw_bwd_data_bn2_impl(PW_BD2_ARGS, 1); }
// mmd_pwconv_bwd_data_bn2 with a workspace for the slab kernel's K slices (ws / ws_floats, nullable: mmd_pwconv_slab_ws_floats) and the
// kernel family chosen per call (form as mmd_pwconv_fwd_form; 0 = the shape filters decide)
extern "C" int mmd_pwconv_bwd_data_bn2_form(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N, const float* scale, const float* shift,
                      const float* mean, const float* invstd, const double* sums, long long count, int act, const float* mul_b,
                      int rows_per_image, float* dz_out, float* dgamma, float* dbeta, const float* residual, const float* xs_z,
                      const float* xs_mean, const float* xs_invstd, const float* xs_mul_b, int xs_rows_per_image, double* xs_sums,
                      double* stats_ws, int ws_slots, const float* p5_z, const float* p5_scale, const float* p5_shift,
                      const float* p5_mean, const float* p5_invstd, float* p5_out, int p5_B, float* ws, long long ws_floats, int form,
                      hipStream_t stream) {
  return pw_bwd_data_bn2_impl(PW_BD2_ARGS, 0, 0, form, ws, ws_floats);
}

extern "C" int mmd_pwconv_bwd_data_bn(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N,
                                      const float* scale, const float* shift, const float* mean, const float* invstd,
                                      const double* sums, long long count, int act, const float* mul_b, int rows_per_image,
                                      float* dz_out, float* dgamma, float* dbeta, hipStream_t stream) {
  return pw_bwd_data_bn_impl(g, z, wt, dx, M, K, N, scale, shift, mean, invstd, sums, count, act, mul_b, rows_per_image, dz_out, dgamma,
                             dbeta, stream, 0);
}
extern "C" int mmd_pwconv_bwd_data_bn_bf16(const float* g, const float* z, const float* wt, float* dx, int M, int K, int N,
                                           const float* scale, const float* shift, const float* mean, const float* invstd,
                                           const double* sums, long long count, int act, const float* mul_b, int rows_per_image,
                                           float* dz_out, float* dgamma, float* dbeta, hipStream_t stream) {
  return pw_bwd_data_bn_impl(g, z, wt, dx, M, K, N, scale, shift, mean, invstd, sums, count, act, mul_b, rows_per_image, dz_out, dgamma,
                             dbeta, stream, 1);
}
extern "C" int mmd_pwconv_bwd_data_bf16(const float* dy, const float* wt, float* dx, int M, int K, int N,
                                        int accumulate, hipStream_t stream) {
  return mmd_pwconv_fwd_bf16(dy, wt, dx, M, /*K=*/N, /*N=*/K, nullptr, nullptr, MMD_ACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0,
                             nullptr, nullptr, nullptr, MMD_ACT_NONE, accumulate ? dx : nullptr, nullptr, 0, 0, nullptr, 0, stream);
}

__global__ void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8)
    if (r0 + i < R && c < C) tile[i][threadIdx.x] = src[(size_t)(r0 + i) * C + c];
  __syncthreads();
  int rr = r0 + threadIdx.x, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8)
    if (c0 + i < C && rr < R) dst[(size_t)(c0 + i) * R + rr] = tile[threadIdx.x][i];
}

// dst[C,R] = src[R,C]^T
extern "C" int mmd_transpose2d(const float* src, float* dst, int R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0 || !src || !dst) return MMD_EINVAL;
  hipLaunchKernelGGL(transpose2d_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream, src, dst, R, C);
  return mmd_check_launch();
}

// All student 1x1 weights transposed in ONE launch: desc[l] = {src_off, dst_off, R, C, first_tile} (floats / 32x32 tiles)
__global__ void transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst, const long long* __restrict__ desc,
                                         int n) {
  __shared__ float tile[32][33];
  const int tb = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (desc[mid * 5 + 4] <= tb) lo = mid; else hi = mid - 1; }
  const long long* d = desc + lo * 5;
  const int R = (int)d[2], C = (int)d[3];
  const int tl = tb - (int)d[4], tcx = (C + 31) / 32;
  const int bx = tl % tcx, by = tl / tcx;
  const float* s = src + d[0];
  float* o = dst + d[1];
  int c = bx * 32 + threadIdx.x, r0 = by * 32;
  for (int i = threadIdx.y; i < 32; i += 8)
    if (r0 + i < R && c < C) tile[i][threadIdx.x] = s[(size_t)(r0 + i) * C + c];
  __syncthreads();
  int rr = r0 + threadIdx.x, c0 = bx * 32;
  for (int i = threadIdx.y; i < 32; i += 8)
    if (c0 + i < C && rr < R) o[(size_t)(c0 + i) * R + rr] = tile[threadIdx.x][i];
}
extern "C" int mmd_transpose_batched(const float* src_base, float* dst_base, const long long* desc, int n, int total_tiles,
                                     hipStream_t stream) {
  if (!src_base || !dst_base || !desc || n <= 0 || total_tiles <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(32, 8), 0, stream, src_base, dst_base, desc, n);
  return mmd_check_launch();
}
