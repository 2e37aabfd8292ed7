// Distillation and detection losses, forward + gradient in the same pass — CDNA4 / gfx950.
//   MTA attention transfer ...... src/loss/MTALoss.py:15-77   (quirk kept: kl_div is fed probabilities)
//   focal + smooth-L1 ........... src/loss/YetAnotherFocalLoss.py:6-190
// Both are HBM-streaming reductions: the MTA channel reduction uses one wave per pixel row with a
// shuffle tree; the per-image softmax/KL rows and the anchor assignment use block reductions.
#include "common.h"

// ------------------------------------------------------------------ MTA: a[b,j] = mean_c f[b,j,c]^p
__global__ __launch_bounds__(256) void mta_attention_kernel(const float* __restrict__ f, float* __restrict__ a, int rows, int C,
                                                            float p) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows) return;
  const float* fr = f + (size_t)wave * C;
  float acc = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float4 v = mmd_ld4(fr + c);
    if (p == 2.f) acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    else acc += powf(v.x, p) + powf(v.y, p) + powf(v.z, p) + powf(v.w, p);
  }
  acc = wave_sum(acc);
  if (lane == 0) a[wave] = acc / (float)C;
}
extern "C" int mmd_mta_attention(const float* f, float* a, int rows, int C, float p, hipStream_t stream) {
  if (!f || !a || rows <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  hipLaunchKernelGGL(mta_attention_kernel, dim3(cdiv((long long)rows * 64, 256)), dim3(256), 0, stream, f, a, rows, C, p);
  return mmd_check_launch();
}

__device__ __forceinline__ float block_sum(float v, float* sm, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sm, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  __syncthreads();
  float r = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sm[i]);
  return r;
}

// One block per image. a_s [B,HW]; a_t0..2 [B,HW] raw attention means of nt teachers (nt==1: pairwise;
// nt>1: list mode = L1-normalised product of the L2-normalised maps).  loss[0] += sum_j v*(log v - u) / B.
// If da_s != null: da_s[b,j] (+)= dL/d a_s[b,j] * gscale   (student side only; teachers are constants).
// accumulate: 0 = store, 1 = read-modify-write (sequential launches), 2 = atomic (concurrent blocks write the same da_s)
__device__ __forceinline__ void mta_kl_body(const float* __restrict__ a_s, const float* __restrict__ t0,
                                            const float* __restrict__ t1, const float* __restrict__ t2, const float* __restrict__ t3, int nt,
                                            int HW, int B, float T, float* loss, float* da_s, float gscale,
                                            int accumulate, int b, float* sm) {
  const int tid = threadIdx.x;
  const float* as = a_s + (size_t)b * HW;
  const float* tp[4] = {t0 + (size_t)b * HW, t1 ? t1 + (size_t)b * HW : nullptr, t2 ? t2 + (size_t)b * HW : nullptr,
                        t3 ? t3 + (size_t)b * HW : nullptr};
  // L2 norms
  float acc = 0.f;
  for (int j = tid; j < HW; j += 256) acc += as[j] * as[j];
  const float ns = fmaxf(sqrtf(block_sum(acc, sm, tid)), 1e-12f);
  float nk[4] = {1.f, 1.f, 1.f, 1.f};
  for (int k = 0; k < nt; ++k) {
    acc = 0.f;
    for (int j = tid; j < HW; j += 256) acc += tp[k][j] * tp[k][j];
    nk[k] = fmaxf(sqrtf(block_sum(acc, sm, tid)), 1e-12f);
  }
  float l1 = 1.f;
  if (nt > 1) {
    acc = 0.f;
    for (int j = tid; j < HW; j += 256) {
      float q = 1.f;
      for (int k = 0; k < nt; ++k) q *= tp[k][j] / nk[k];
      acc += fabsf(q);
    }
    l1 = fmaxf(block_sum(acc, sm, tid), 1e-12f);
  }
  auto that = [&](int j) {
    float q = 1.f;
    for (int k = 0; k < nt; ++k) q *= tp[k][j] / nk[k];
    return nt > 1 ? q / l1 : q;
  };
  // softmax(x/T) over j for student and teacher
  float ms = -INFINITY, mt = -INFINITY;
  for (int j = tid; j < HW; j += 256) { ms = fmaxf(ms, as[j] / ns / T); mt = fmaxf(mt, that(j) / T); }
  ms = block_max(ms, sm, tid); mt = block_max(mt, sm, tid);
  float es = 0.f, et = 0.f;
  for (int j = tid; j < HW; j += 256) { es += __expf(as[j] / ns / T - ms); et += __expf(that(j) / T - mt); }
  es = block_sum(es, sm, tid); et = block_sum(et, sm, tid);
  // KL (as the reference calls it) and the two dot products the backward needs
  float kl = 0.f, dot_gu = 0.f;
  for (int j = tid; j < HW; j += 256) {
    float u = __expf(as[j] / ns / T - ms) / es;
    float v = __expf(that(j) / T - mt) / et;
    kl += (v > 0.f ? v * __logf(v) : 0.f) - v * u;
    dot_gu += (-v) * u;            // g_j = -v_j (scaled later)
  }
  kl = block_sum(kl, sm, tid); dot_gu = block_sum(dot_gu, sm, tid);
  if (tid == 0) atomicAdd(loss, kl / (float)B);
  if (!da_s) return;
  // d ahat_j = u_j (g_j - <g,u>) / T ;  da_j = (d ahat_j - ahat_j <ahat, d ahat>) / ns
  float dot_ad = 0.f;
  for (int j = tid; j < HW; j += 256) {
    float ah = as[j] / ns;
    float u = __expf(ah / T - ms) / es;
    float v = __expf(that(j) / T - mt) / et;
    float dah = u * (-v - dot_gu) / T;
    dot_ad += ah * dah;
  }
  dot_ad = block_sum(dot_ad, sm, tid);
  const float sc = gscale / (float)B;
  for (int j = tid; j < HW; j += 256) {
    float ah = as[j] / ns;
    float u = __expf(ah / T - ms) / es;
    float v = __expf(that(j) / T - mt) / et;
    float dah = u * (-v - dot_gu) / T;
    float d = (dah - ah * dot_ad) / ns * sc;
    size_t o = (size_t)b * HW + j;
    if (accumulate == 2) atomicAdd(&da_s[o], d);
    else da_s[o] = accumulate ? da_s[o] + d : d;
  }
}
// Pairwise form (one teacher map) with both maps held in REGISTERS (HW <= 256 * MTA_RC): the generic body above re-reads them from global
// memory in each of its ten reduction passes (45 us for the 64 x 64 level's blocks, on the critical path between forward and backward).
// Same per-thread element order (j = tid + 256 i) and the same block reductions: the same numbers.
#define MTA_RC 16
__device__ __forceinline__ void mta_kl_body_cached(const float* __restrict__ a_s, const float* __restrict__ t0, int HW, int B, float T,
                                                   float* loss, float* da_s, float gscale, int accumulate, int b, float* sm) {
  const int tid = threadIdx.x;
  const float* as = a_s + (size_t)b * HW;
  const float* tp = t0 + (size_t)b * HW;
  float ra[MTA_RC], rt[MTA_RC];
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) { const int j = tid + 256 * i; ra[i] = j < HW ? as[j] : 0.f; rt[i] = j < HW ? tp[j] : 0.f; }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) acc += ra[i] * ra[i];
  const float ns = fmaxf(sqrtf(block_sum(acc, sm, tid)), 1e-12f);
  acc = 0.f;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) acc += rt[i] * rt[i];
  const float nk = fmaxf(sqrtf(block_sum(acc, sm, tid)), 1e-12f);
  float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) { ms = fmaxf(ms, ra[i] / ns / T); mt = fmaxf(mt, (1.f * rt[i] / nk) / T); }
  ms = block_max(ms, sm, tid); mt = block_max(mt, sm, tid);
  float es = 0.f, et = 0.f;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) { es += __expf(ra[i] / ns / T - ms); et += __expf((1.f * rt[i] / nk) / T - mt); }
  es = block_sum(es, sm, tid); et = block_sum(et, sm, tid);
  float kl = 0.f, dot_gu = 0.f;
  float ru[MTA_RC], rv[MTA_RC];
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) {
    ru[i] = 0.f; rv[i] = 0.f;
    if (tid + 256 * i < HW) {
      const float u = __expf(ra[i] / ns / T - ms) / es;
      const float v = __expf((1.f * rt[i] / nk) / T - mt) / et;
      ru[i] = u; rv[i] = v;
      kl += (v > 0.f ? v * __logf(v) : 0.f) - v * u;
      dot_gu += (-v) * u;
    }
  }
  kl = block_sum(kl, sm, tid); dot_gu = block_sum(dot_gu, sm, tid);
  if (tid == 0) atomicAdd(loss, kl / (float)B);
  if (!da_s) return;
  float dot_ad = 0.f;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) {
    const float ah = ra[i] / ns;
    const float dah = ru[i] * (-rv[i] - dot_gu) / T;
    dot_ad += ah * dah;
  }
  dot_ad = block_sum(dot_ad, sm, tid);
  const float sc = gscale / (float)B;
#pragma unroll
  for (int i = 0; i < MTA_RC; ++i) if (tid + 256 * i < HW) {
    const float ah = ra[i] / ns;
    const float dah = ru[i] * (-rv[i] - dot_gu) / T;
    const float d = (dah - ah * dot_ad) / ns * sc;
    const size_t o = (size_t)b * HW + tid + 256 * i;
    if (accumulate == 2) atomicAdd(&da_s[o], d);
    else da_s[o] = accumulate ? da_s[o] + d : d;
  }
}
__global__ __launch_bounds__(256) void mta_kl_kernel(const float* __restrict__ a_s, const float* __restrict__ t0,
                                                     const float* __restrict__ t1, const float* __restrict__ t2, int nt,
                                                     int HW, int B, float T, float* loss, float* da_s, float gscale,
                                                     int accumulate) {
  __shared__ float sm[8];
  mta_kl_body(a_s, t0, t1, t2, nullptr, nt, HW, B, T, loss, da_s, gscale, accumulate, blockIdx.x, sm);
}

// Every (level, teacher) pair of a step in ONE launch: grid (B, levels, pairs).  The per-pair launches are 8-block kernels
// of ~10 dependent block reductions each (5-19 us apiece, 15 of them back to back on the step's critical path).
//   pairwise mode (ModelWithNMSLoss): pairs = teachers, loss[t*nlev + l]; da_s[l] accumulates over teachers with atomics (zero on entry)
//   list mode (ModelWithNMSKDListLoss): pairs = 1, each block multiplies the nt teacher maps, loss[l]
#define MTA_MAX_LEV 5
#define MTA_MAX_T 4      // 3 teachers + the "augmentation" pass of ModelWithNMSKDListLossAugmented (src/optimization/train_methods.py:73-110)
struct MtaMulti {
  const float* a_s[MTA_MAX_LEV]; const float* a_t[MTA_MAX_T][MTA_MAX_LEV]; float* da[MTA_MAX_LEV]; int HW[MTA_MAX_LEV];
  int nt, list_mode, B; float T, gscale; float* loss;
};
__global__ __launch_bounds__(256) void mta_kl_multi_kernel(MtaMulti m) {
  __shared__ float sm[8];
  const int l = blockIdx.y, t = blockIdx.z, nlev = gridDim.y;
  if (m.list_mode)
    mta_kl_body(m.a_s[l], m.a_t[0][l], m.nt > 1 ? m.a_t[1][l] : nullptr, m.nt > 2 ? m.a_t[2][l] : nullptr,
                m.nt > 3 ? m.a_t[3][l] : nullptr, m.nt, m.HW[l], m.B, m.T, m.loss + l, m.da[l], m.gscale, 0, blockIdx.x, sm);
  else if (m.HW[l] <= 256 * MTA_RC)
    mta_kl_body_cached(m.a_s[l], m.a_t[t][l], m.HW[l], m.B, m.T, m.loss + t * nlev + l, m.da[l], m.gscale, m.nt > 1 ? 2 : 0, blockIdx.x, sm);
  else
    mta_kl_body(m.a_s[l], m.a_t[t][l], nullptr, nullptr, nullptr, 1, m.HW[l], m.B, m.T, m.loss + t * nlev + l, m.da[l], m.gscale,
                m.nt > 1 ? 2 : 0, blockIdx.x, sm);
}
extern "C" int mmd_mta_kl_multi(const float* const* a_s, const float* const* a_t, float* const* da_s, const int* HW, int nlev,
                                int nteachers, int list_mode, int B, float T, float* loss, float gscale, hipStream_t stream) {
  if (!a_s || !a_t || !HW || !loss || nlev < 1 || nlev > MTA_MAX_LEV || nteachers < 1 || nteachers > MTA_MAX_T || B <= 0 || !(T > 0.f))
    return MMD_EINVAL;
  MtaMulti m{};
  for (int l = 0; l < nlev; ++l) {
    if (!a_s[l] || HW[l] <= 0) return MMD_EINVAL;
    m.a_s[l] = a_s[l]; m.HW[l] = HW[l]; m.da[l] = da_s ? da_s[l] : nullptr;
    for (int t = 0; t < nteachers; ++t) {
      if (!a_t[t * nlev + l]) return MMD_EINVAL;
      m.a_t[t][l] = a_t[t * nlev + l];
    }
  }
  m.nt = nteachers; m.list_mode = list_mode; m.B = B; m.T = T; m.gscale = gscale; m.loss = loss;
  hipLaunchKernelGGL(mta_kl_multi_kernel, dim3(B, nlev, list_mode ? 1 : nteachers), dim3(256), 0, stream, m);
  return mmd_check_launch();
}
extern "C" int mmd_mta_kl(const float* a_s, const float* a_t0, const float* a_t1, const float* a_t2, int nteachers,
                          int B, int HW, float T, float* loss, float* da_s, float gscale, int accumulate,
                          hipStream_t stream) {
  if (!a_s || !a_t0 || !loss || B <= 0 || HW <= 0 || nteachers < 1 || nteachers > 3 || !(T > 0.f)) return MMD_EINVAL;
  if ((nteachers > 1 && !a_t1) || (nteachers > 2 && !a_t2)) return MMD_EINVAL;
  hipLaunchKernelGGL(mta_kl_kernel, dim3(B), dim3(256), 0, stream, a_s, a_t0, a_t1, a_t2, nteachers, HW, B, T, loss, da_s,
                     gscale, accumulate);
  return mmd_check_launch();
}

// df[b,j,c] (+)= da[b,j] * p * f^(p-1) / C
__global__ __launch_bounds__(256) void mta_attention_bwd_kernel(const float* __restrict__ f, const float* __restrict__ da,
                                                                float* __restrict__ df, int rows, int C, float p, int accumulate) {
  const int c4n = C >> 2;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)rows * c4n) return;
  int row = (int)(idx / c4n), c = (int)(idx % c4n) * 4;
  float k = da[row] * p / (float)C;
  size_t off = (size_t)row * C + c;
  float4 v = mmd_ld4(f + off), o;
  if (p == 2.f) { o.x = k * v.x; o.y = k * v.y; o.z = k * v.z; o.w = k * v.w; }
  else { o.x = k * powf(v.x, p - 1.f); o.y = k * powf(v.y, p - 1.f); o.z = k * powf(v.z, p - 1.f); o.w = k * powf(v.w, p - 1.f); }
  if (accumulate) { float4 q = mmd_ld4(df + off); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
  mmd_st4(df + off, o);
}
extern "C" int mmd_mta_attention_bwd(const float* f, const float* da, float* df, int rows, int C, float p, int accumulate,
                                     hipStream_t stream) {
  if (!f || !da || !df || rows <= 0 || C <= 0 || (C & 3)) return MMD_EINVAL;
  hipLaunchKernelGGL(mta_attention_bwd_kernel, dim3(cdiv((long long)rows * (C >> 2), 256)), dim3(256), 0, stream, f, da, df,
                     rows, C, p, accumulate);
  return mmd_check_launch();
}

// ------------------------------------------------------------------ focal loss
// boxes [B, maxg, 5] (x1,y1,x2,y2,label), nbox [B].  assign[b,a]: >=0 positive (box index), -1 ignore, -2 negative.
#define FA_LDS_BOXES 1024
__global__ __launch_bounds__(256) void focal_assign_kernel(const float* __restrict__ anchors, const float* __restrict__ boxes,
                                                           const int* __restrict__ nbox, int maxg, int A, int* __restrict__ assign,
                                                           int* npos) {
  // the image's boxes (+ areas) staged in LDS once per block: the per-anchor loop read them as scalars from global memory, five dependent
  // L1 round trips per box and anchor (48 us for ~70 boxes per image, on the critical path between forward and backward)
  __shared__ float4 sbx[FA_LDS_BOXES];
  __shared__ float sar[FA_LDS_BOXES];
  const int b = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int G = min(nbox[b], maxg);
  const float* bx = boxes + (size_t)b * maxg * 5;
  const int GL = min(G, FA_LDS_BOXES);
  for (int g = threadIdx.x; g < GL; g += 256) {
    const float x1 = bx[g * 5], y1 = bx[g * 5 + 1], x2 = bx[g * 5 + 2], y2 = bx[g * 5 + 3];
    sbx[g] = make_float4(x1, y1, x2, y2);
    sar[g] = (x2 - x1) * (y2 - y1);
  }
  __syncthreads();
  int cnt = 0;
  if (a < A && G > 0) {
    float4 an = mmd_ld4(anchors + (size_t)a * 4);     // y1,x1,y2,x2
    float aarea = (an.z - an.x) * (an.w - an.y);
    float best = -INFINITY; int bi = 0;
    for (int g = 0; g < G; ++g) {
      float x1, y1, x2, y2, area;
      if (g < GL) { const float4 q = sbx[g]; x1 = q.x; y1 = q.y; x2 = q.z; y2 = q.w; area = sar[g]; }
      else { x1 = bx[g * 5]; y1 = bx[g * 5 + 1]; x2 = bx[g * 5 + 2]; y2 = bx[g * 5 + 3]; area = (x2 - x1) * (y2 - y1); }
      float iw = fmaxf(fminf(an.w, x2) - fmaxf(an.y, x1), 0.f);
      float ih = fmaxf(fminf(an.z, y2) - fmaxf(an.x, y1), 0.f);
      // a disjoint pair has IoU exactly 0, which replaces `best` only while that is still -inf (the first box): skip its division
      if (g > 0 && !(iw * ih > 0.f)) continue;
      float ua = fmaxf(aarea + area - iw * ih, 1e-8f);
      float iou = iw * ih / ua;
      if (iou > best) { best = iou; bi = g; }      // first maximum wins, like torch.max on CPU
    }
    int v = best >= 0.5f ? bi : (best < 0.4f ? -2 : -1);
    assign[(size_t)b * A + a] = v;
    cnt = v >= 0;
  }
  unsigned long long m = __ballot(cnt);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&npos[b], __popcll(m));
}

// loss sums: acc[b*2+0] += cls-sum, acc[b*2+1] += reg-sum (double).  Gradients written for every element.
// Two passes per block of 256 anchors: (1) thread = anchor: assignment / label into LDS, smooth-L1 + its gradient; (2) every thread walks
// float4s of the block's contiguous run of 256 x NC class probabilities (coalesced loads and stores; the per-anchor form touched 64
// cache lines per wave instruction: 59 us for 2 x 31 MB on the critical path between forward and backward).  A thread's focal terms are
// summed in fp32, the reduction across threads runs in fp64.
__global__ __launch_bounds__(256) void focal_loss_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                                         const float* __restrict__ anchors, const float* __restrict__ boxes,
                                                         const int* __restrict__ nbox, const int* __restrict__ assign,
                                                         const int* __restrict__ npos, int maxg, int A, int NC, int B,
                                                         double* acc, float* __restrict__ dcls, float* __restrict__ dreg,
                                                         float gscale, int to_logit) {
  __shared__ double sd[8];
  __shared__ short s_as[256], s_lab[256];       // per anchor: -2 negative / -1 ignore / 0 positive; label of a positive
  const int b = blockIdx.y, tid = threadIdx.x;
  const int a0 = blockIdx.x * 256;
  const int a = a0 + tid;
  int total = 0;
  for (int i = 0; i < B; ++i) total += min(nbox[i], maxg);
  const int G = min(nbox[b], maxg);
  const float alpha = 0.25f;
  const int np = G > 0 ? npos[b] : 0;
  const float norm = G > 0 ? fmaxf((float)np, 1.f) : 1.f;     // image w/o boxes: un-normalised (reference quirk)
  const float gs = gscale / (float)B / norm;
  double lc = 0.0, lr = 0.0;
  // ---- pass 1
  {
    int as = -2, lab = -1;
    if (a < A) {
      float* dr = dreg ? dreg + ((size_t)b * A + a) * 4 : nullptr;
      float g4[4] = {0.f, 0.f, 0.f, 0.f};
      if (total != 0) {
        as = G > 0 ? assign[(size_t)b * A + a] : -2;
        if (as >= 0) {
          const float* bx = boxes + ((size_t)b * maxg + as) * 5;
          lab = (int)bx[4];
          float4 an = mmd_ld4(anchors + (size_t)a * 4);
          float aw = an.w - an.y, ah = an.z - an.x;
          float acx = an.y + 0.5f * aw, acy = an.x + 0.5f * ah;
          float gw = bx[2] - bx[0], gh = bx[3] - bx[1];
          float gcx = bx[0] + 0.5f * gw, gcy = bx[1] + 0.5f * gh;
          gw = fmaxf(gw, 1.f); gh = fmaxf(gh, 1.f);
          float t[4] = {(gcy - acy) / ah, (gcx - acx) / aw, __logf(gh / ah), __logf(gw / aw)};
          const float* r = reg + ((size_t)b * A + a) * 4;
          const float gr = gscale / (float)B / (4.f * (float)np);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float diff = r[k] - t[k];
            float ad = fabsf(diff);
            float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            if (ad <= 1.f / 9.f) { lr += 4.5 * (double)ad * ad; g4[k] = 9.f * ad * sgn * gr; }
            else { lr += (double)ad - 0.5 / 9.0; g4[k] = sgn * gr; }
          }
          lr /= (4.0 * (double)np);
        }
      }
      if (dr) { dr[0] = g4[0]; dr[1] = g4[1]; dr[2] = g4[2]; dr[3] = g4[3]; }
    }
    s_as[tid] = (short)(as >= 0 ? 0 : as); s_lab[tid] = (short)lab;
  }
  __syncthreads();
  // ---- pass 2
  auto term = [&](float raw, int as, bool is_lab, float& gv) -> float {      // focal term of one (anchor, class) and its gradient
    float q = fminf(fmaxf(raw, 1e-4f), 1.f - 1e-4f);
    bool inside = raw >= 1e-4f && raw <= 1.f - 1e-4f;
    float l = 0.f, d = 0.f;
    if (as != -1) {
      if (is_lab) {
        float om = 1.f - q, lg = __logf(q);
        l = -alpha * om * om * lg;
        d = alpha * (2.f * om * lg - om * om / q);
      } else {
        float om = 1.f - q, lg = __logf(om);
        l = -(1.f - alpha) * q * q * lg;
        d = (1.f - alpha) * (-2.f * q * lg + q * q / om);
      }
    }
    gv = inside ? d * gs : 0.f;
    if (to_logit) gv *= raw * (1.f - raw);
    return l;
  };
  const int na = min(256, A - a0);
  const size_t base = ((size_t)b * A + a0) * NC;
  const int run = na * NC;
  float ls = 0.f;
  if (total == 0) {                         // whole batch without boxes: zero loss, no gradient
    if (dcls) for (int e = tid; e < run; e += 256) dcls[base + e] = 0.f;
  } else if ((NC & 3) == 0) {
    for (int e = tid * 4; e < run; e += 1024) {
      const int r = e / NC, c = e - r * NC;                 // NC % 4 == 0: the four elements belong to one anchor
      const int as = s_as[r], lab = s_lab[r];
      const float4 r4 = mmd_ld4(cls + base + e);
      float4 g4v;
      ls += term(r4.x, as, c == lab, g4v.x) + term(r4.y, as, c + 1 == lab, g4v.y) + term(r4.z, as, c + 2 == lab, g4v.z) +
            term(r4.w, as, c + 3 == lab, g4v.w);
      if (dcls) mmd_st4(dcls + base + e, g4v);
    }
  } else {
    for (int e = tid; e < run; e += 256) {
      const int r = e / NC, c = e - r * NC;
      float gv;
      ls += term(cls[base + e], s_as[r], c == s_lab[r], gv);
      if (dcls) dcls[base + e] = gv;
    }
  }
  lc = (double)ls / (double)norm;
  lc = wave_sum_d(lc); lr = wave_sum_d(lr);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sd[wave * 2] = lc; sd[wave * 2 + 1] = lr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc[b * 2], sd[0] + sd[2] + sd[4] + sd[6]);
    atomicAdd(&acc[b * 2 + 1], sd[1] + sd[3] + sd[5] + sd[7]);
  }
}

// out[0] = regression loss, out[1] = classification loss (means over images; zeros if the batch has no box)
__global__ void focal_finalize_kernel(const double* acc, const int* nbox, int maxg, int B, float* out, int* any_boxes) {
  if (threadIdx.x != 0) return;
  int total = 0;
  for (int i = 0; i < B; ++i) total += min(nbox[i], maxg);
  double c = 0, r = 0;
  for (int i = 0; i < B; ++i) { c += acc[i * 2]; r += acc[i * 2 + 1]; }
  out[0] = total ? (float)(r / B) : 0.f;
  out[1] = total ? (float)(c / B) : 0.f;
  if (any_boxes && total) *any_boxes = 1;     // sticky: head parameters have received a gradient
}

__global__ void focal_zero_kernel(int* npos, double* acc, int B) {
  for (int i = threadIdx.x; i < B; i += blockDim.x) npos[i] = 0;
  for (int i = threadIdx.x; i < 2 * B; i += blockDim.x) acc[i] = 0.0;
}
// cls [B,A,NC] post-sigmoid probabilities, reg [B,A,4], anchors [A,4] (y1,x1,y2,x2).
// workspace: assign int[B*A], npos int[B] (zeroed here), acc double[2B] (zeroed here).
extern "C" int mmd_focal_loss(const float* cls, const float* reg, const float* anchors, const float* boxes,
                              const int* nbox, int maxg, int B, int A, int NC, int* assign_ws, int* npos_ws,
                              double* acc_ws, float* loss_out, float* dcls, float* dreg, float grad_scale,
                              int to_logit, int* any_boxes, hipStream_t stream) {
  if (!cls || !reg || !anchors || !boxes || !nbox || !assign_ws || !npos_ws || !acc_ws || !loss_out) return MMD_EINVAL;
  if (B <= 0 || A <= 0 || NC <= 0 || maxg <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(focal_zero_kernel, dim3(1), dim3(256), 0, stream, npos_ws, acc_ws, B);      // (one launch: both sit on the critical path)
  dim3 grid(cdiv(A, 256), B);
  hipLaunchKernelGGL(focal_assign_kernel, grid, dim3(256), 0, stream, anchors, boxes, nbox, maxg, A, assign_ws, npos_ws);
  hipLaunchKernelGGL(focal_loss_kernel, grid, dim3(256), 0, stream, cls, reg, anchors, boxes, nbox, assign_ws, npos_ws, maxg,
                     A, NC, B, acc_ws, dcls, dreg, grad_scale, to_logit);
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, stream, acc_ws, nbox, maxg, B, loss_out, any_boxes);
  return mmd_check_launch();
}
