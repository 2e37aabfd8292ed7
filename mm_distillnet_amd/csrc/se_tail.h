// Squeeze-excite FC pair WITHOUT launches of its own (round 4) — CDNA4 / gfx950.
//
// Reference: MBConvBlock.forward, src/YetAnotherEfficientNet.py:469-474 (adaptive_avg_pool2d -> _se_reduce -> swish -> _se_expand ->
// sigmoid), and autograd's backward of the same four ops.
//
// The two FC layers of a squeeze-excite block are ~10^4..10^5 MACs per image behind a GLOBAL dependency: every workgroup of the
// launch that pools an image (depthwise epilogue / fused expand+depthwise kernel / chan_pool in the forward, the project conv's
// input-gradient GEMM epilogue in the backward) must have added its partial sums first.  As launches of their own (se_hidden + se_gate,
// se_bwd_a + se_bwd_b: 230 launches per step) they sit on all four nets' serial chains at ~5 us apiece for ~0 work.  Here the pooling
// launch keeps an arrival counter per image; the workgroup whose agent-scope add comes LAST runs the image's FC pair once - nothing is
// recomputed per block - and writes the gate (forward) / dpooled + the BatchNorm-1 sums (backward) the next launch reads.
//
// Hand-off form (cdna_hip_programming.md §6 Guideline 16, MI355X_MICROARCH.md § visibility, table row 1): the payload (partial sums) is
// written with agent-scope atomic adds, every adding wave drains (s_waitcnt vmcnt(0)), the workgroup barrier orders its waves, ONE lane
// adds to the image's counter; the workgroup told "last" by the value its add returned loads the sums with sc1 (agent-scope) loads
// only - they bypass this CU's L1, and no plain load of those bytes exists in the tail.  Counters are zeroed before every launch
// (they live in the per-step accumulator arena that one memset clears).
#pragma once
#include "common.h"

struct SeTail {
  unsigned* cnt;            // [B] arrival counters, zero before the launch; nullptr = no tail (the FCs run as launches)
  int nblk;                 // workgroups that add to ONE image's pooled sums in this launch
  const float* pooled;      // [B, C] the completed sums (forward: the launch's own `pool` output)
  const float* wr; const float* br;      // reduce FC  [S, C], [S]
  const float* wet; const float* be;     // expand FC, transposed [S, C], [C]
  float* hpre; float* gate;              // [B, S] pre-activation hidden vector (kept for the backward), [B, C] sigmoid gate
  int C, S;
};
#define MMD_SE_MAXC 3072                  // widest expanded tensor (EfficientNet-b6: 2064; b4 2688)
#define MMD_SE_MAXS 256

__device__ __forceinline__ float mmd_ld_agent(const float* p) {       // sc1 load: served by L2 / memory, never by this CU's L1
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// true in every thread of the workgroup whose arrival completed image b's sums.  Called by ALL threads, after the block's last atomic add.
__device__ __forceinline__ bool mmd_last_arriver(unsigned* cnt, int nblk) {
  __shared__ int sLast;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's adds have been performed at the coherence point
  __syncthreads();                                        // ... and every other wave's of this workgroup
  if (threadIdx.x == 0)
    sLast = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblk - 1);
  __syncthreads();
  return sLast != 0;
}

// Forward FC pair of image b.  smem: >= C + S floats of LDS the caller no longer needs.  Arithmetic and summation order are those of
// se_hidden_kernel / se_gate_kernel (elt.hip): bit-identical gate for identical pooled sums.
template <int NT>
__device__ __forceinline__ void mmd_se_tail_fwd(const SeTail& t, int b, float* smem) {
  if (!mmd_last_arriver(t.cnt + b, t.nblk)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = t.C, S = t.S;
  float* const sp = smem;          // pooled [C]
  float* const shd = smem + C;     // swish(hidden) [S]
  for (int c = tid; c < C; c += NT) sp[c] = mmd_ld_agent(t.pooled + (size_t)b * C + c);
  __syncthreads();
  for (int j = wave; j < S; j += NT / 64) {
    const float* w = t.wr + (size_t)j * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 q = mmd_ld4(w + c), a = *reinterpret_cast<const float4*>(sp + c);
      acc += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float h = acc + t.br[j];
      t.hpre[(size_t)b * S + j] = h;
      shd[j] = mmd_swish(h);
    }
  }
  __syncthreads();
  for (int c = tid * 4; c < C; c += NT * 4) {
    float4 acc = mmd_ld4(t.be + c);
    const float* w = t.wet + c;
#pragma unroll 4
    for (int j = 0; j < S; ++j) {
      const float4 q = mmd_ld4(w + (size_t)j * C);
      const float h = shd[j];
      acc.x += q.x * h; acc.y += q.y * h; acc.z += q.z * h; acc.w += q.w * h;
    }
    *reinterpret_cast<float4*>(t.gate + (size_t)b * C + c) =
        make_float4(mmd_sigmoid(acc.x), mmd_sigmoid(acc.y), mmd_sigmoid(acc.z), mmd_sigmoid(acc.w));
  }
}

// Backward of the FC pair by the last-arriving workgroup of the project conv's input-gradient GEMM, whose epilogue pools
// pool5 [5][B][C] (Pool5Op, pw_args.h).  Arithmetic of se_bwd_a_kernel / se_bwd_b_kernel (elt.hip):
//   dpe[c] = dgate[c] gate[c] (1 - gate[c]),  dgate = pool5[0]          dh[j] = sum_c wet[j,c] dpe[c]
//   dpr[j] = dh[j] swish'(hpre[j])                                       dpooled[c] = dpool_scale sum_j wr[j,c] dpr[j]
//   bn_sums[c] += gate pool5[1] + dpooled pool5[3],  bn_sums[C + c] += gate pool5[2] + dpooled pool5[4]      (BatchNorm-1 backward sums)
struct SeTailBwd {
  unsigned* cnt; int nblk;
  const float* pool5;       // [5][B][C]
  const float* gate; const float* hpre; const float* wr; const float* wet;
  float* dpe; float* dpr; float* dpooled; double* bn_sums;
  float dpool_scale; int B, C, S;
};
template <int NT>
__device__ __forceinline__ void mmd_se_tail_bwd(const SeTailBwd& t, int b, float* smem) {
  if (!mmd_last_arriver(t.cnt + b, t.nblk)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = t.C, S = t.S;
  float* const sd = smem;          // dpe [C]
  float* const sr = smem + C;      // dpr [S]
  const size_t n = (size_t)t.B * C, bc = (size_t)b * C;
  for (int c = tid; c < C; c += NT) {
    const float g = t.gate[bc + c];
    const float d = mmd_ld_agent(t.pool5 + bc + c) * g * (1.f - g);
    sd[c] = d;
    t.dpe[bc + c] = d;
  }
  __syncthreads();
  for (int j = wave; j < S; j += NT / 64) {
    const float* w = t.wet + (size_t)j * C;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w[c] * sd[c];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float d = acc * mmd_swish_grad(t.hpre[(size_t)b * S + j]);
      sr[j] = d;
      t.dpr[(size_t)b * S + j] = d;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    float acc = 0.f;
#pragma unroll 4
    for (int j = 0; j < S; ++j) acc += t.wr[(size_t)j * C + c] * sr[j];
    const float dp = acc * t.dpool_scale;
    t.dpooled[bc + c] = dp;
    const float gt = t.gate[bc + c];
    atomicAdd(&t.bn_sums[c], (double)(gt * mmd_ld_agent(t.pool5 + n + bc + c) + dp * mmd_ld_agent(t.pool5 + 3 * n + bc + c)));
    atomicAdd(&t.bn_sums[C + c], (double)(gt * mmd_ld_agent(t.pool5 + 2 * n + bc + c) + dp * mmd_ld_agent(t.pool5 + 4 * n + bc + c)));
  }
}
