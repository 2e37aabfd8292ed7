// Device-side pseudo-label generation (no per-image host sync) — CDNA4 / gfx950.
//   decode + clip + threshold + class filter ... src/YetAnotherEfficientDet.py:574-602, src/utils/utils.py:123-231
//   class-aware NMS per teacher, int() truncation . src/utils/utils.py:205-231,285-323 (torchvision batched_nms semantics)
//   cross-teacher merge + class-agnostic NMS ..... src/optimization/train_methods.py:361-411
// Integer / index results are meant to be bit-identical to the CPU path, so the box arithmetic uses
// explicitly rounded fp32 ops (no FMA contraction) in the reference's operation order, candidates are
// compacted in ANCHOR ORDER (block prefix sums, not atomics) and the NMS sort is (score desc, index asc).
// Reference quirk kept: the score emitted for a kept box is the score of the idx-th OVER-THRESHOLD
// candidate, idx being the box's index in the class-filtered list (src/utils/utils.py:193-213).
#include "common.h"

#define PP_CAP 1024          // candidates per image the single-pass (all-in-LDS) NMS handles; longer lists take the chunked path
// Capacities are run-time arguments (`cap` = rows per image of the candidate / label arrays).  The reference has no cap at all
// (src/utils/utils.py:179-205 runs torchvision's NMS over every over-threshold anchor), so the host sizes the arrays for
// the worst case (cap = A anchors) and nothing is ever truncated; lists of up to PP_CAP candidates - every realistic
// teacher - stay on the one-pass kernel path.

// ---- pass 1: per anchor best class/score and flags
// A block's 256 anchors x NC class scores are one contiguous run: staged through LDS with coalesced float4 loads (row stride NC + 1:
// conflict-free per-anchor reads).  The per-thread form read 80-byte-strided scalars: 80 us for the pack's 24 x 49104 x 20 scores (1.2 TB/s)
// on the frozen teachers' chain, which is the forward phase's critical path (profiles/r04_notes.md section 22).
#define PPS_MAXNC 64
__global__ __launch_bounds__(256) void pp_score_kernel(const float* __restrict__ cls, int A, int NC, float thr,
                                                       unsigned long long valid_mask, float* __restrict__ score,
                                                       unsigned char* __restrict__ clsid, unsigned char* __restrict__ flags) {
  extern __shared__ float spp[];                  // [256][NC + 1]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int a0 = blockIdx.x * 256;
  const int na = min(256, A - a0);
  const size_t base = ((size_t)b * A + a0) * NC;  // (16-byte aligned when 256 NC % 4 == 0: NC % 4 == 0 or handled by the scalar tail below)
  const int total = na * NC;
  const int ld = NC + 1;
  if (((base | (size_t)NC) & 3) == 0) {
    for (int e = tid * 4; e < total; e += 1024) {
      const float4 v = mmd_ld4(cls + base + e);
      const int r = e / NC, c = e - r * NC;         // NC % 4 == 0: the four elements lie in one row
      float* d = &spp[r * ld + c];
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int e = tid; e < total; e += 256) { const int r = e / NC; spp[r * ld + (e - r * NC)] = cls[base + e]; }
  }
  __syncthreads();
  const int a = a0 + tid;
  if (a >= A) return;
  const float* p = &spp[tid * ld];
  float best = p[0]; int bi = 0;
  for (int c = 1; c < NC; ++c) { float v = p[c]; if (v > best) { best = v; bi = c; } }
  size_t o = (size_t)b * A + a;
  score[o] = best; clsid[o] = (unsigned char)bi;
  unsigned char f = 0;
  if (best > thr) { f = 1; if ((valid_mask >> bi) & 1ull) f |= 2; }
  flags[o] = f;
}

// ---- pass 2: ordered compaction (one 1024-thread block per image)
// over_scores[b, i]      : score of the i-th over-threshold anchor (anchor order)
// cand[b, i, 0..5]       : (x1,y1,x2,y2,score,class) of the i-th over-threshold AND valid-class anchor
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {      // exclusive prefix sum over the 64 lanes
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  return x - v;
}
// One block per (image, chunk of 1024 anchors), 256 threads x 4 consecutive anchors (round 4; until then ONE 1024-thread block per image
// walked the anchors in 12 barrier rounds: 49 us for 24 blocks on the teachers' chain).  The chunk's base positions in the image's two
// ordered lists are the counts of the flags in front of it, which the block takes from the flag bytes themselves (<= 48 KB: popcounts of
// 16-byte loads) - no workspace, no cross-block hand-off; the rows land exactly where the one-block walk put them.
#define PP_PER 4
#define PP_CHUNK (256 * PP_PER)
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void pp_count_word(unsigned w, int& co, int& ck) { co += __popc(w & 0x01010101u); ck += __popc(w & 0x02020202u); }
__global__ __launch_bounds__(256) void pp_compact_kernel(const float* __restrict__ reg, const float* __restrict__ anchors,
                                                         const float* __restrict__ score, const unsigned char* __restrict__ clsid,
                                                         const unsigned char* __restrict__ flags, int A, float image_size,
                                                         float* __restrict__ over_scores, float* __restrict__ cand,
                                                         int* __restrict__ n_over, int* __restrict__ n_keep, int* overflow,
                                                         int cap) {
  __shared__ int s_o[4], s_k[4], s_bo[4], s_bk[4];
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int a0 = blockIdx.x * PP_CHUNK;
  const unsigned char* fb = flags + (size_t)b * A;
  // ---- flags in front of the chunk
  int co0 = 0, ck0 = 0;
  if ((((size_t)b * A) & 15) == 0) {
    const uint4* f16 = reinterpret_cast<const uint4*>(fb);
    for (int i = tid; i < a0 / 16; i += 256) {          // a0 is a multiple of 1024
      const uint4 v = f16[i];
      pp_count_word(v.x, co0, ck0); pp_count_word(v.y, co0, ck0); pp_count_word(v.z, co0, ck0); pp_count_word(v.w, co0, ck0);
    }
  } else {
    for (int i = tid; i < a0; i += 256) { const unsigned char f = fb[i]; co0 += f & 1; ck0 += (f >> 1) & 1; }
  }
  // ---- the chunk's own flags
  const int a = a0 + tid * PP_PER;
  unsigned char f[PP_PER];
  int co = 0, ck = 0;
#pragma unroll
  for (int i = 0; i < PP_PER; ++i) {
    f[i] = (a + i < A) ? fb[a + i] : 0;
    co += f[i] & 1; ck += (f[i] >> 1) & 1;
  }
  const int po = wave_excl_scan(co, lane), pk = wave_excl_scan(ck, lane);
  const int wb_o = wave_sum_i(co0), wb_k = wave_sum_i(ck0);
  if (lane == 63) { s_o[wave] = po + co; s_k[wave] = pk + ck; }
  if (lane == 0) { s_bo[wave] = wb_o; s_bk[wave] = wb_k; }
  __syncthreads();
  int wo = 0, wk = 0, to = 0, tk = 0, bo = 0, bk = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { if (i < wave) { wo += s_o[i]; wk += s_k[i]; } to += s_o[i]; tk += s_k[i]; bo += s_bo[i]; bk += s_bk[i]; }
  int pos = bo + wo + po, kp = bk + wk + pk;
#pragma unroll
  for (int i = 0; i < PP_PER; ++i) {
    if (!(f[i] & 1)) continue;
    const int ai = a + i;
    float sc = score[(size_t)b * A + ai];
    // only the first `cap` entries can ever be indexed (index < n_keep <= cap), so a longer list is not an error
    if (pos < cap) over_scores[(size_t)b * cap + pos] = sc;
    ++pos;
    if (f[i] & 2) {
      if (kp < cap) {
        float4 an = mmd_ld4(anchors + (size_t)ai * 4);                 // y1,x1,y2,x2
        float4 r = mmd_ld4(reg + ((size_t)b * A + ai) * 4);            // dy,dx,dh,dw
        float yca = __fdiv_rn(__fadd_rn(an.x, an.z), 2.f), xca = __fdiv_rn(__fadd_rn(an.y, an.w), 2.f);
        float ha = __fsub_rn(an.z, an.x), wa = __fsub_rn(an.w, an.y);
        float w = __fmul_rn(expf(r.w), wa), h = __fmul_rn(expf(r.z), ha);
        float yc = __fadd_rn(__fmul_rn(r.x, ha), yca), xc = __fadd_rn(__fmul_rn(r.y, wa), xca);
        float x1 = __fsub_rn(xc, __fdiv_rn(w, 2.f)), y1 = __fsub_rn(yc, __fdiv_rn(h, 2.f));
        float x2 = __fadd_rn(xc, __fdiv_rn(w, 2.f)), y2 = __fadd_rn(yc, __fdiv_rn(h, 2.f));
        x1 = fmaxf(x1, 0.f); y1 = fmaxf(y1, 0.f); x2 = fminf(x2, image_size); y2 = fminf(y2, image_size);
        float* o = cand + ((size_t)b * cap + kp) * 6;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = sc; o[5] = (float)clsid[(size_t)b * A + ai];
      } else *overflow = 1;
      ++kp;
    }
  }
  if (tid == 0 && a0 + PP_CHUNK >= A) { n_over[b] = min(bo + to, cap); n_keep[b] = min(bk + tk, cap); }      // the image's last chunk
}

extern "C" int mmd_decode_filter(const float* cls, const float* reg, const float* anchors, int B, int A, int NC,
                                 float conf_threshold, unsigned long long valid_class_mask, float image_size,
                                 float* score_ws, unsigned char* clsid_ws, unsigned char* flags_ws,
                                 float* over_scores, float* cand, int* n_over, int* n_keep, int* overflow, int cap,
                                 hipStream_t stream) {
  if (!cls || !reg || !anchors || !score_ws || !clsid_ws || !flags_ws || !over_scores || !cand || !n_over || !n_keep || !overflow)
    return MMD_EINVAL;
  if (B <= 0 || A <= 0 || NC <= 0 || NC > 64 || cap <= 0) return MMD_EINVAL;
  const size_t score_lds = (size_t)256 * (NC + 1) * sizeof(float);
  if (score_lds > 64 * 1024) {      // NC = 64: 66 560 B, just above the default dynamic-LDS limit (the CU has 160 KB)
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)pp_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 65 * (int)sizeof(float)) != hipSuccess) return MMD_ELAUNCH;
      attr = true;
    }
  }
  hipLaunchKernelGGL(pp_score_kernel, dim3(cdiv(A, 256), B), dim3(256), score_lds, stream, cls, A, NC, conf_threshold,
                     valid_class_mask, score_ws, clsid_ws, flags_ws);
  hipLaunchKernelGGL(pp_compact_kernel, dim3(cdiv(A, PP_CHUNK), B), dim3(256), 0, stream, reg, anchors, score_ws, clsid_ws, flags_ws, A,
                     image_size, over_scores, cand, n_over, n_keep, overflow, cap);
  return mmd_check_launch();
}

// ---- greedy NMS over rows [x1,y1,x2,y2,score,label] (one 1024-thread block per image)
// mode 0 (per-teacher): class-aware (boxes offset by label*(max_coord+1) like torchvision.batched_nms);
//         out row = (int(x1),int(y1),int(min(x2,S)),int(min(y2,S)), over_scores[orig idx], label_map[label])
// mode 1 (cross-teacher merge): rows gathered from up to 3 sources in order, class-agnostic;
//         out row = (x1,y1,x2,y2,label)  [5 columns]
// n <= PP_CAP rows: everything lives in LDS (sort, boxes, suppression bitmask).  n > PP_CAP (an untrained student at evaluation time, a badly
// calibrated teacher): the same greedy algorithm in 1024-row chunks of the sorted list - a chunk is first tested against every box
// kept so far, then resolved internally with the bitmask - with the sorted order, the kept list and the chunk bitmask in a global
// workspace.  Greedy NMS only ever compares a candidate with EARLIER KEPT boxes, so the chunked result is the sequential one, bit for bit.
struct NmsArgs {
  const float* src[4]; const int* cnt[4]; int nsrc;
  int mode; float thr; int inclusive;
  const float* over_scores; const int* label_map; float image_size;
  float* out; int* out_cnt; int out_cols; int out_cap;
  unsigned long long* mask_ws; int* overflow;
  int merge01;        // mode 1: image 1 takes image 0's rows in front of its own when both have rows
                      // (ModelWithNMSLossAugmented.forward, augment=True: src/optimization/train_methods.py:379-387)
  int in_cap;         // rows per image of every source array (and of over_scores)
  float* big_ws; long long big_stride; int P;      // chunked path: per image key[P] | idx[P] | kbox[4*nmax] | karea[nmax] | kidx[nmax]
  int nmax;           // upper bound of n: nsrc * in_cap (* 2 with merge01)
};

__device__ __forceinline__ bool nms_hit(float ix1, float iy1, float ix2, float iy2, float ia, float jx1, float jy1, float jx2, float jy2,
                                        float ja, float thr, int inclusive) {
  float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
  float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
  float ww = fmaxf(0.f, __fsub_rn(xx2, xx1)), hh = fmaxf(0.f, __fsub_rn(yy2, yy1));
  float inter = __fmul_rn(ww, hh);
  float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ia, ja), inter));
  return inclusive ? (ovr >= thr) : (ovr > thr);
}

__global__ __launch_bounds__(1024) void pp_nms_kernel(NmsArgs a) {
  __shared__ float skey[PP_CAP];
  __shared__ int sidx[PP_CAP];
  __shared__ float sbox[PP_CAP * 4];
  __shared__ float sarea[PP_CAP];
  __shared__ unsigned long long smask[3584];      // 28 KB: used when n * words <= 3584
  __shared__ int skeep[PP_CAP];
  __shared__ float sred[16];
  __shared__ unsigned long long sdead[16];
  __shared__ int s_n, s_nk;
  const int b = blockIdx.x, tid = threadIdx.x;
  // gather source rows (concatenation order = source order)
  int cnts[4] = {0, 0, 0, 0}, cnts0[4] = {0, 0, 0, 0}, n = 0, n0 = 0;
  for (int s = 0; s < a.nsrc; ++s) { cnts[s] = min(a.cnt[s][b], a.in_cap); n += cnts[s]; }
  if (a.merge01 && b == 1 && n > 0) {
    for (int s = 0; s < a.nsrc; ++s) { cnts0[s] = min(a.cnt[s][0], a.in_cap); n0 += cnts0[s]; }
    n += n0;
  }
  if (n > PP_CAP && (!a.big_ws || n > a.nmax)) { if (tid == 0) *a.overflow = 1; n = PP_CAP; }      // no workspace: the old hard limit
  auto row_ptr = [&](int i) -> const float* {
    int s = 0, img = b;
    if (i < n0) {           // image 0's rows first, teacher order preserved
      while (s < a.nsrc - 1 && i >= cnts0[s]) { i -= cnts0[s]; ++s; }
      img = 0;
    } else {
      i -= n0;
      while (s < a.nsrc - 1 && i >= cnts[s]) { i -= cnts[s]; ++s; }
    }
    return a.src[s] + ((size_t)img * a.in_cap + i) * 6;
  };
  if (n == 0) { if (tid == 0) a.out_cnt[b] = 0; return; }
  // max coordinate for the class offset (mode 0), over ALL n rows
  float maxc = -INFINITY;
  if (a.mode == 0) {
    for (int i = tid; i < n; i += 1024) { const float* r = row_ptr(i); maxc = fmaxf(maxc, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]))); }
    maxc = wave_max(maxc);
    if ((tid & 63) == 0) sred[tid >> 6] = maxc;
    __syncthreads();
    maxc = sred[0];
    for (int i = 1; i < 16; ++i) maxc = fmaxf(maxc, sred[i]);
  }
  auto emit = [&](int k, int oi) {          // k-th kept row <- source row oi
    const float* r = row_ptr(oi);
    float* o = a.out + ((size_t)b * a.out_cap + k) * a.out_cols;
    if (a.mode == 0) {
      o[0] = (float)(int)fmaxf(r[0], 0.f); o[1] = (float)(int)fmaxf(r[1], 0.f);
      o[2] = (float)(int)fminf(r[2], a.image_size); o[3] = (float)(int)fminf(r[3], a.image_size);
      o[4] = a.over_scores[(size_t)b * a.in_cap + oi];            // reference quirk, see header
      o[5] = (float)a.label_map[(int)r[5]];
    } else {
      o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[5];
    }
  };
  auto stage_box = [&](int t, int oi) {     // LDS slot t <- (class-offset) box of source row oi
    const float* r = row_ptr(oi);
    float off = 0.f;
    if (a.mode == 0) off = __fmul_rn(r[5], __fadd_rn(maxc, 1.f));
    float x1 = __fadd_rn(r[0], off), y1 = __fadd_rn(r[1], off), x2 = __fadd_rn(r[2], off), y2 = __fadd_rn(r[3], off);
    sbox[t * 4] = x1; sbox[t * 4 + 1] = y1; sbox[t * 4 + 2] = x2; sbox[t * 4 + 3] = y2;
    sarea[t] = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
  };

  if (n <= PP_CAP) {
    // ---------------- one pass, all in LDS
    float myscore = -INFINITY;
    if (tid < n) myscore = row_ptr(tid)[4];
    skey[tid] = myscore; sidx[tid] = tid;
    __syncthreads();
    // bitonic sort, descending score, ascending index on ties, over the smallest power of two >= n (>= 64) slots: the slots past n hold
    // -inf keys and sort behind every row; a teacher's ~100 candidates need 28 barrier stages instead of the 55 of a 1024-wide sort
    int PS = 64;
    while (PS < n) PS <<= 1;
    for (int k = 2; k <= PS; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        int ixj = tid ^ j;
        if (ixj > tid && ixj < PS) {
          float k0 = skey[tid], k1 = skey[ixj]; int i0 = sidx[tid], i1 = sidx[ixj];
          bool first_before = (k0 > k1) || (k0 == k1 && i0 < i1);     // "tid element sorts before ixj element"
          bool up = ((tid & k) == 0);
          if (up ? !first_before : first_before) { skey[tid] = k1; skey[ixj] = k0; sidx[tid] = i1; sidx[ixj] = i0; }
        }
        __syncthreads();
      }
    if (tid < n) stage_box(tid, sidx[tid]);
    __syncthreads();
    const int words = (n + 63) >> 6;
    const bool in_lds = n * words <= 3584;
    unsigned long long* mask = in_lds ? smask : a.mask_ws + (size_t)b * PP_CAP * (PP_CAP / 64);
    for (int e = tid; e < n * words; e += 1024) {
      int i = e / words, w = e % words;
      unsigned long long bits = 0ull;
      float ix1 = sbox[i * 4], iy1 = sbox[i * 4 + 1], ix2 = sbox[i * 4 + 2], iy2 = sbox[i * 4 + 3], ia = sarea[i];
      int j0 = w * 64;
      for (int q = 0; q < 64; ++q) {
        int j = j0 + q;
        if (j <= i || j >= n) continue;
        if (nms_hit(ix1, iy1, ix2, iy2, ia, sbox[j * 4], sbox[j * 4 + 1], sbox[j * 4 + 2], sbox[j * 4 + 3], sarea[j], a.thr, a.inclusive))
          bits |= 1ull << q;
      }
      mask[(size_t)i * words + w] = bits;
    }
    __syncthreads();
    if (!in_lds) __threadfence();
    __syncthreads();
    if (tid < 64) {
      unsigned long long removed = 0ull;     // lane w owns word w (w < words <= 16)
      int nk = 0;
      for (int i = 0; i < n; ++i) {
        unsigned long long rw = __shfl(removed, i >> 6, 64);
        if (!((rw >> (i & 63)) & 1ull)) {
          if (tid == 0) skeep[nk] = i;
          ++nk;
          if (tid < words) removed |= mask[(size_t)i * words + tid];
        }
      }
      if (tid == 0) { s_nk = nk; s_n = n; }
    }
    __syncthreads();
    int nk = s_nk;
    if (nk > a.out_cap) { if (tid == 0) *a.overflow = 1; nk = a.out_cap; }
    if (tid == 0) a.out_cnt[b] = nk;
    for (int k = tid; k < nk; k += 1024) emit(k, sidx[skeep[k]]);
    return;
  }

  // ---------------- chunked path (n > PP_CAP)
  float* ws = a.big_ws + (size_t)b * a.big_stride;
  float* gkey = ws;
  int* gidx = reinterpret_cast<int*>(ws + a.P);
  float* kbox = ws + 2 * (size_t)a.P;
  float* karea = kbox + 4 * (size_t)a.nmax;
  int* kidx = reinterpret_cast<int*>(karea + a.nmax);
  int P = 2048; while (P < n) P <<= 1;          // sort width for this image (<= a.P)
  for (int e = tid; e < P; e += 1024) { gkey[e] = e < n ? row_ptr(e)[4] : -INFINITY; gidx[e] = e; }
  __syncthreads();
  // bitonic sort in global memory (one workgroup = one CU = one vector L1: workgroup-scope visibility through the barrier)
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = tid; e < P; e += 1024) {
        int ixj = e ^ j;
        if (ixj > e) {
          float k0 = gkey[e], k1 = gkey[ixj]; int i0 = gidx[e], i1 = gidx[ixj];
          bool first_before = (k0 > k1) || (k0 == k1 && i0 < i1);
          bool up = ((e & k) == 0);
          if (up ? !first_before : first_before) { gkey[e] = k1; gkey[ixj] = k0; gidx[e] = i1; gidx[ixj] = i0; }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
  unsigned long long* mask = a.mask_ws + (size_t)b * PP_CAP * (PP_CAP / 64);
  int nkept = 0;
  for (int c0 = 0; c0 < n; c0 += PP_CAP) {
    const int m = min(PP_CAP, n - c0);
    if (tid < m) { const int oi = gidx[c0 + tid]; sidx[tid] = oi; stage_box(tid, oi); }
    __syncthreads();
    // against every box kept by the earlier chunks
    bool alive = tid < m;
    if (alive) {
      const float ix1 = sbox[tid * 4], iy1 = sbox[tid * 4 + 1], ix2 = sbox[tid * 4 + 2], iy2 = sbox[tid * 4 + 3], ia = sarea[tid];
      for (int kk = 0; kk < nkept; ++kk) {
        const float4 kb = *reinterpret_cast<const float4*>(kbox + 4 * (size_t)kk);
        // operand order as in the sequential algorithm: the kept (earlier, higher-score) box is "i", the candidate is "j"
        if (nms_hit(kb.x, kb.y, kb.z, kb.w, karea[kk], ix1, iy1, ix2, iy2, ia, a.thr, a.inclusive)) { alive = false; break; }
      }
    }
    const unsigned long long live = __ballot(alive);
    if ((tid & 63) == 0) sdead[tid >> 6] = ~live;
    // inside the chunk: suppression bitmask of the pairs (i < j)
    const int words = (m + 63) >> 6;
    for (int e = tid; e < m * words; e += 1024) {
      int i = e / words, w = e % words;
      unsigned long long bits = 0ull;
      float ix1 = sbox[i * 4], iy1 = sbox[i * 4 + 1], ix2 = sbox[i * 4 + 2], iy2 = sbox[i * 4 + 3], ia = sarea[i];
      int j0 = w * 64;
      for (int q = 0; q < 64; ++q) {
        int j = j0 + q;
        if (j <= i || j >= m) continue;
        if (nms_hit(ix1, iy1, ix2, iy2, ia, sbox[j * 4], sbox[j * 4 + 1], sbox[j * 4 + 2], sbox[j * 4 + 3], sarea[j], a.thr, a.inclusive))
          bits |= 1ull << q;
      }
      mask[(size_t)i * words + w] = bits;
    }
    __threadfence_block();
    __syncthreads();
    if (tid < 64) {
      unsigned long long removed = tid < words ? sdead[tid] : 0ull;
      int nk = 0;
      for (int i = 0; i < m; ++i) {
        unsigned long long rw = __shfl(removed, i >> 6, 64);
        if (!((rw >> (i & 63)) & 1ull)) {
          if (tid == 0) skeep[nk] = i;
          ++nk;
          if (tid < words) removed |= mask[(size_t)i * words + tid];
        }
      }
      if (tid == 0) s_nk = nk;
    }
    __syncthreads();
    const int nkc = s_nk;
    for (int k = tid; k < nkc; k += 1024) {
      const int i = skeep[k];
      *reinterpret_cast<float4*>(kbox + 4 * (size_t)(nkept + k)) = make_float4(sbox[i * 4], sbox[i * 4 + 1], sbox[i * 4 + 2], sbox[i * 4 + 3]);
      karea[nkept + k] = sarea[i];
      kidx[nkept + k] = sidx[i];
    }
    nkept += nkc;
    __threadfence_block();
    __syncthreads();
  }
  int nk = nkept;
  if (nk > a.out_cap) { if (tid == 0) *a.overflow = 1; nk = a.out_cap; }
  if (tid == 0) a.out_cnt[b] = nk;
  for (int k = tid; k < nk; k += 1024) emit(k, kidx[k]);
}

static int pp_pow2(int n) { int p = 2048; while (p < n) p <<= 1; return p; }
// floats of chunked-path workspace per image for lists of up to nmax rows
// (rounded up to a multiple of 4 floats: kbox is read and written as float4, so every image's slice must start 16-byte aligned -
// with an odd nmax, e.g. cap = 9 * 341 anchors at image_size 128, 6 * nmax alone leaves odd images 8-byte aligned)
extern "C" int mmd_nms_ws_floats(int nmax) { return nmax <= PP_CAP ? 0 : (int)((2ll * pp_pow2(nmax) + 6ll * nmax + 3) / 4 * 4); }

// per-teacher NMS: cand [B,cap,6] + n_keep -> out [B,cap,6] (truncated coords, quirk score, mapped label), out_cnt [B].
// mask_ws: B * 1024 * 16 words.  big_ws (nullable when cap <= 1024): B * mmd_nms_ws_floats(cap) floats.
extern "C" int mmd_nms_teacher(const float* cand, const int* n_keep, const float* over_scores, const int* label_map,
                               float nms_threshold, int inclusive, float image_size, int B, float* out, int* out_cnt,
                               unsigned long long* mask_ws, int* overflow, int cap, float* big_ws, hipStream_t stream) {
  if (!cand || !n_keep || !over_scores || !label_map || !out || !out_cnt || !mask_ws || !overflow || B <= 0 || cap <= 0) return MMD_EINVAL;
  NmsArgs a{};
  a.src[0] = cand; a.cnt[0] = n_keep; a.nsrc = 1; a.mode = 0; a.thr = nms_threshold; a.inclusive = inclusive;
  a.over_scores = over_scores; a.label_map = label_map; a.image_size = image_size;
  a.out = out; a.out_cnt = out_cnt; a.out_cols = 6; a.out_cap = cap; a.mask_ws = mask_ws; a.overflow = overflow;
  a.in_cap = cap; a.nmax = cap; a.big_ws = big_ws; a.P = pp_pow2(a.nmax); a.big_stride = mmd_nms_ws_floats(a.nmax);
  hipLaunchKernelGGL(pp_nms_kernel, dim3(B), dim3(1024), 0, stream, a);
  return mmd_check_launch();
}

// cross-teacher merge: up to 3 per-teacher outputs ([B,cap,6] + counts, in teacher order) ->
// boxes [B,maxg,5] (x1,y1,x2,y2,label) in NMS keep order, nbox [B].  big_ws (nullable when the concatenation cannot exceed 1024
// rows): B * mmd_nms_ws_floats(nteachers * cap * (merge01 ? 2 : 1)) floats.
extern "C" int mmd_nms_merge(const float* t0, const int* c0, const float* t1, const int* c1, const float* t2,
                             const int* c2, int nteachers, float iou_threshold, int inclusive, int B, float* boxes,
                             int* nbox, int maxg, unsigned long long* mask_ws, int* overflow, int merge01, int cap, float* big_ws,
                             hipStream_t stream) {
  if (!t0 || !c0 || !boxes || !nbox || !mask_ws || !overflow || B <= 0 || nteachers < 1 || nteachers > 3 || maxg <= 0 || cap <= 0) return MMD_EINVAL;
  if ((nteachers > 1 && (!t1 || !c1)) || (nteachers > 2 && (!t2 || !c2))) return MMD_EINVAL;
  NmsArgs a{};
  a.src[0] = t0; a.cnt[0] = c0; a.src[1] = t1; a.cnt[1] = c1; a.src[2] = t2; a.cnt[2] = c2; a.nsrc = nteachers;
  a.mode = 1; a.thr = iou_threshold; a.inclusive = inclusive; a.merge01 = (merge01 && B >= 2) ? 1 : 0;
  a.out = boxes; a.out_cnt = nbox; a.out_cols = 5; a.out_cap = maxg; a.mask_ws = mask_ws; a.overflow = overflow;
  a.in_cap = cap; a.nmax = nteachers * cap * (a.merge01 ? 2 : 1); a.big_ws = big_ws; a.P = pp_pow2(a.nmax);
  a.big_stride = mmd_nms_ws_floats(a.nmax);
  hipLaunchKernelGGL(pp_nms_kernel, dim3(B), dim3(1024), 0, stream, a);
  return mmd_check_launch();
}

// Same merge over up to 4 sources given as host arrays of device pointers (the 4th: the "augmentation" pass of
// ModelWithNMSKDListLossAugmented, src/optimization/train_methods.py:73-110, whose labels are concatenated after the teachers').
extern "C" int mmd_nms_merge_n(const float* const* srcs, const int* const* cnts, int nsrc, float iou_threshold, int inclusive, int B,
                               float* boxes, int* nbox, int maxg, unsigned long long* mask_ws, int* overflow, int merge01, int cap,
                               float* big_ws, hipStream_t stream) {
  if (!srcs || !cnts || !boxes || !nbox || !mask_ws || !overflow || B <= 0 || nsrc < 1 || nsrc > 4 || maxg <= 0 || cap <= 0) return MMD_EINVAL;
  NmsArgs a{};
  for (int i = 0; i < nsrc; ++i) { if (!srcs[i] || !cnts[i]) return MMD_EINVAL; a.src[i] = srcs[i]; a.cnt[i] = cnts[i]; }
  a.nsrc = nsrc; a.mode = 1; a.thr = iou_threshold; a.inclusive = inclusive; a.merge01 = (merge01 && B >= 2) ? 1 : 0;
  a.out = boxes; a.out_cnt = nbox; a.out_cols = 5; a.out_cap = maxg; a.mask_ws = mask_ws; a.overflow = overflow;
  a.in_cap = cap; a.nmax = nsrc * cap * (a.merge01 ? 2 : 1); a.big_ws = big_ws; a.P = pp_pow2(a.nmax);
  a.big_stride = mmd_nms_ws_floats(a.nmax);
  hipLaunchKernelGGL(pp_nms_kernel, dim3(B), dim3(1024), 0, stream, a);
  return mmd_check_launch();
}

extern "C" int mmd_pp_cap(void) { return PP_CAP; }
