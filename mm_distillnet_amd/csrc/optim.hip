// Flat-buffer optimizer and utilities — CDNA4 / gfx950.
// All student parameters, gradients and Adam moments live in single flat fp32 buffers (one 32 MB
// stream each), so the optimizer is ONE coalesced float4 pass and the gradient all-reduce works on
// contiguous buckets.  Reference: torch.optim.Adam as configured in
// src/optimization/train_methods.py:825-833 (no weight decay passed, no amsgrad), stepped at
// src/optimization/traditional.py:190.
#include "common.h"

// state[0] = step (as float, exact up to 2^24), state[1] = lr/bc1, state[2] = 1/sqrt(bc2)
// hyper: device floats [lr, beta1, beta2, eps] so a captured graph sees scheduler updates.
__global__ void adam_prep_kernel(float* state, const float* hyper, const int* active) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (active && *active == 0) return;
  float step = state[0] + 1.f;
  state[0] = step;
  double b1 = hyper[1], b2 = hyper[2];
  double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  state[1] = (float)((double)hyper[0] / bc1);
  state[2] = (float)(1.0 / sqrt(bc2));
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* state, const float* hyper,
                                                   const int* active, float gscale, size_t n4) {
  if (active && *active == 0) return;
  const float step_size = state[1], inv_sqrt_bc2 = state[2];
  const float b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = mmd_ld4(p + i * 4), gg = mmd_ld4(g + i * 4), mm = mmd_ld4(m + i * 4), vv = mmd_ld4(v + i * 4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = G[k] * gscale;
      M[k] = b1 * M[k] + (1.f - b1) * gk;
      V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
      float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
      P[k] -= step_size * (M[k] / denom);
    }
    mmd_st4(p + i * 4, pp); mmd_st4(m + i * 4, mm); mmd_st4(v + i * 4, vv);
  }
}
// One Adam step over a contiguous segment of the flat buffers (n % 4 == 0, 16-byte aligned).
// `active` (nullable): device int; the segment is skipped while it is 0 (parameters that have never
// received a gradient are skipped by torch.optim.Adam as well: p.grad is None).
extern "C" int mmd_adam_step(float* p, const float* g, float* m, float* v, float* state, const float* hyper,
                             const int* active, float grad_scale, long long n, hipStream_t stream) {
  if (!p || !g || !m || !v || !state || !hyper || n <= 0 || (n & 3)) return MMD_EINVAL;
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, stream, state, hyper, active);
  size_t n4 = (size_t)n / 4;
  int blocks = cdiv(n4, 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, state, hyper, active, grad_scale, n4);
  return mmd_check_launch();
}

// Zero fill as a plain kernel: hipMemsetAsync nodes recorded by stream capture did not reliably clear the large
// accumulator arenas on replay (sums kept growing across replays), a kernel node has no such ambiguity.
__global__ __launch_bounds__(256) void zero_fill_kernel(float4* __restrict__ p, size_t n16, unsigned char* tail, int ntail) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
int mmd_zero_bytes(void* p, size_t bytes, hipStream_t stream) {
  uintptr_t a = (uintptr_t)p;
  if (a & 15) {            // unaligned start: byte loop for the head
    return hipMemsetAsync(p, 0, bytes, stream) == hipSuccess ? MMD_OK : MMD_ELAUNCH;
  }
  size_t n16 = bytes / 16;
  int ntail = (int)(bytes - n16 * 16);
  int blocks = cdiv(n16 ? n16 : 1, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, stream, (float4*)p, n16, (unsigned char*)p + n16 * 16, ntail);
  return mmd_check_launch();
}
extern "C" int mmd_memset_async(void* p, int value, long long bytes, hipStream_t stream) {
  if (!p || bytes <= 0) return MMD_EINVAL;
  if (value == 0) return mmd_zero_bytes(p, (size_t)bytes, stream);
  return hipMemsetAsync(p, value, (size_t)bytes, stream) == hipSuccess ? MMD_OK : MMD_ELAUNCH;
}

// sum of squares of a flat buffer into out[0] (double), for clip_grad_norm_ (traditional.py:184-188)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, double* out) {
  __shared__ double sd[4];
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += (double)x[i] * x[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sd[0] + sd[1] + sd[2] + sd[3]);
}
__global__ void clip_scale_kernel(float* __restrict__ x, size_t n, const double* sumsq, float max_norm) {
  float norm = (float)sqrt(*sumsq);
  float coef = max_norm / (norm + 1e-6f);
  if (coef >= 1.f) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= coef;
}
extern "C" int mmd_clip_grad_norm(float* g, long long n, float max_norm, double* sumsq_ws, hipStream_t stream) {
  if (!g || n <= 0 || !sumsq_ws || !(max_norm > 0.f)) return MMD_EINVAL;
  mmd_zero_bytes(sumsq_ws, sizeof(double), stream);
  int blocks = cdiv(n, 1024); if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, g, (size_t)n, sumsq_ws);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, stream, g, (size_t)n, sumsq_ws, max_norm);
  return mmd_check_launch();
}

// Whole-buffer Adam with the "head" parameter ranges gated by a device flag.  torch.optim.Adam skips
// parameters whose .grad is None and keeps a per-parameter step count; in the reference the
// regressor/classifier parameters have no gradient until the first batch with pseudo-labels
// (src/loss/YetAnotherFocalLoss.py:179-187 returns constants), after which zero_grad() keeps zero
// tensors around.  ranges: 3 x [begin,end) in floats (conv weights, BN gammas, BN betas of the heads).
struct AdamRanges { long long b[3]; long long e[3]; };
__global__ void adam_prep2_kernel(float* st_main, float* st_head, const float* hyper, const int* head_active) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int s = 0; s < 2; ++s) {
    float* state = s ? st_head : st_main;
    if (s && *head_active == 0) continue;
    float step = state[0] + 1.f;
    state[0] = step;
    double b1 = hyper[1], b2 = hyper[2];
    double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    state[1] = (float)((double)hyper[0] / bc1);
    state[2] = (float)(1.0 / sqrt(bc2));
  }
}
__global__ __launch_bounds__(256) void adam2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const float* st_main, const float* st_head,
                                                    const float* hyper, const int* head_active, AdamRanges r, float gscale,
                                                    size_t n4) {
  const int hact = *head_active;
  const float b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    long long e0 = (long long)i * 4;
    bool head = (e0 >= r.b[0] && e0 < r.e[0]) || (e0 >= r.b[1] && e0 < r.e[1]) || (e0 >= r.b[2] && e0 < r.e[2]);
    if (head && !hact) continue;
    const float* st = head ? st_head : st_main;
    const float step_size = st[1], inv_sqrt_bc2 = st[2];
    float4 pp = mmd_ld4(p + i * 4), gg = mmd_ld4(g + i * 4), mm = mmd_ld4(m + i * 4), vv = mmd_ld4(v + i * 4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = G[k] * gscale;
      M[k] = b1 * M[k] + (1.f - b1) * gk;
      V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
      float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
      P[k] -= step_size * (M[k] / denom);
    }
    mmd_st4(p + i * 4, pp); mmd_st4(m + i * 4, mm); mmd_st4(v + i * 4, vv);
  }
}
extern "C" int mmd_adam_step_gated(float* p, const float* g, float* m, float* v, float* state_main, float* state_head,
                                   const float* hyper, const int* head_active, long long b0, long long e0, long long b1,
                                   long long e1, long long b2, long long e2, float grad_scale, long long n,
                                   hipStream_t stream) {
  if (!p || !g || !m || !v || !state_main || !state_head || !hyper || !head_active || n <= 0 || (n & 3)) return MMD_EINVAL;
  if ((b0 | e0 | b1 | e1 | b2 | e2) & 3) return MMD_EINVAL;
  AdamRanges r{{b0, b1, b2}, {e0, e1, e2}};
  hipLaunchKernelGGL(adam_prep2_kernel, dim3(1), dim3(64), 0, stream, state_main, state_head, hyper, head_active);
  size_t n4 = (size_t)n / 4;
  int blocks = cdiv(n4, 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam2_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, state_main, state_head, hyper,
                     head_active, r, grad_scale, n4);
  return mmd_check_launch();
}


// The reference's other optimizers behind the same flat, head-gated pass (src/optimization/train_methods.py:808-836: cfg `optimizer`
// = SGD(lr, momentum, weight_decay) | Adam(lr, betas) | AdamW(lr, betas; torch's default weight_decay 1e-2)).
//   mode 0 Adam, 1 AdamW (decoupled decay: p *= 1 - lr*wd before the Adam update), 2 SGD (d = g + wd*p; buf = d on a parameter's
//   first step, else momentum*buf + d; p -= lr*buf - torch.optim.SGD with dampening 0, no nesterov; `m` is the momentum buffer, `v` unused).
// hyper: device floats [lr, beta1, beta2, eps, weight_decay, momentum].
template <int MODE>
__global__ __launch_bounds__(256) void opt2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* st_main, const float* st_head,
                                                   const float* hyper, const int* head_active, AdamRanges r, float gscale,
                                                   size_t n4) {
  const int hact = *head_active;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], mom = hyper[5];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    long long e0 = (long long)i * 4;
    bool head = (e0 >= r.b[0] && e0 < r.e[0]) || (e0 >= r.b[1] && e0 < r.e[1]) || (e0 >= r.b[2] && e0 < r.e[2]);
    if (head && !hact) continue;
    const float* st = head ? st_head : st_main;
    const float step_size = st[1], inv_sqrt_bc2 = st[2];
    const bool first = st[0] == 1.f;
    float4 pp = mmd_ld4(p + i * 4), gg = mmd_ld4(g + i * 4), mm = mmd_ld4(m + i * 4), vv = make_float4(0, 0, 0, 0);
    if (MODE != 2) vv = mmd_ld4(v + i * 4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = G[k] * gscale;
      if (MODE == 2) {
        float d = gk + wd * P[k];
        if (mom != 0.f) { M[k] = first ? d : mom * M[k] + d; d = M[k]; }
        P[k] -= lr * d;
      } else {
        if (MODE == 1) P[k] *= 1.f - lr * wd;
        M[k] = b1 * M[k] + (1.f - b1) * gk;
        V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
        float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
        P[k] -= step_size * (M[k] / denom);
      }
    }
    mmd_st4(p + i * 4, pp); mmd_st4(m + i * 4, mm);
    if (MODE != 2) mmd_st4(v + i * 4, vv);
  }
}
extern "C" int mmd_opt_step_gated(int mode, float* p, const float* g, float* m, float* v, float* state_main, float* state_head,
                                  const float* hyper6, const int* head_active, long long b0, long long e0, long long b1,
                                  long long e1, long long b2, long long e2, float grad_scale, long long n, hipStream_t stream) {
  if (!p || !g || !m || !v || !state_main || !state_head || !hyper6 || !head_active || n <= 0 || (n & 3)) return MMD_EINVAL;
  if ((b0 | e0 | b1 | e1 | b2 | e2) & 3) return MMD_EINVAL;
  if (mode < 0 || mode > 2) return MMD_EINVAL;
  AdamRanges r{{b0, b1, b2}, {e0, e1, e2}};
  hipLaunchKernelGGL(adam_prep2_kernel, dim3(1), dim3(64), 0, stream, state_main, state_head, hyper6, head_active);
  size_t n4 = (size_t)n / 4;
  int blocks = cdiv(n4, 256); if (blocks > 2048) blocks = 2048;
  auto k = mode == 0 ? opt2_kernel<0> : mode == 1 ? opt2_kernel<1> : opt2_kernel<2>;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, stream, p, g, m, v, state_main, state_head, hyper6, head_active, r, grad_scale, n4);
  return mmd_check_launch();
}

// ---------------------------------------------------------------- drop-connect masks on the device
// reference: drop_connect (src/YetAnotherEfficientNet.py:173-182): per sample and skip block, mask = floor(keep + U[0, 1)), scale = mask / keep.
// One launch INSIDE the captured step draws all of them, so the replay loop holds no ATen RNG / floor / div / copy launches:
// Philox4x32-10 (Salmon et al., SC'11) keyed by `seed`, counter = (draws so far, element index); `state` [2] on the device:
//   state[0] = number of draws so far (advanced by this launch), state[1] != 0 = "injected": the caller wrote out[] itself (tests, a resumed
//   run that replays recorded masks) and this launch leaves it alone.  One block: the counter update is ordered behind every read of it.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__global__ __launch_bounds__(256) void drop_scale_kernel(float* __restrict__ out, const float* __restrict__ keep, int n_skip, int B,
                                                         unsigned long long seed, unsigned long long* state) {
  const unsigned long long draw = state[0];
  const bool injected = state[1] != 0ull;
  if (!injected) {
    const int n = n_skip * B;
    for (int q = threadIdx.x; q * 4 < n; q += 256) {      // one Philox block = four uniforms
      unsigned c[4] = {(unsigned)draw, (unsigned)(draw >> 32), (unsigned)q, 0u};
      philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = q * 4 + e;
        if (i < n) {
          const float kp = keep[i / B];
          const float u = (float)(c[e] >> 8) * (1.0f / 16777216.0f);      // 24 random bits: [0, 1)
          out[i] = floorf(kp + u) / kp;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && !injected) state[0] = draw + 1ull;
}
extern "C" int mmd_drop_scale(float* out, const float* keep, int n_skip, int batch, unsigned long long seed, unsigned long long* state,
                              hipStream_t stream) {
  if (!out || !keep || !state || n_skip <= 0 || batch <= 0) return MMD_EINVAL;
  hipLaunchKernelGGL(drop_scale_kernel, dim3(1), dim3(256), 0, stream, out, keep, n_skip, batch, seed, state);
  return mmd_check_launch();
}
