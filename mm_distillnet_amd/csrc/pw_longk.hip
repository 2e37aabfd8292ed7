// Long-K, small-M pointwise (1x1) convolution with a software-pipelined K loop — CDNA4 / gfx950.
//
//   Y[M,N] = (X * gate)[M,K] * W[N,K]^T (+ epilogue),   K >= 384, few row tiles (the 16x16 / 32x32 stages: M = 2048 / 8192)
//
// These launches have about one block per CU, so nothing but the block itself can hide its loads, and pw_gemm_skinny_kernel's one
// register stage leaves ~1 us of load latency exposed in every 128-wide K step (M2048 K1248 N208: 24 us for 8 us of MFMA work,
// profiles/r02_notes.md).  Register double-buffering is not expressible through hipcc (its waitcnt pass drains both stages; loads issued
// from inline asm get their in-flight destination registers copied by the register allocator before the hand-placed wait).  So the
// in-flight data never touches a VGPR: the A and B tiles of a K step travel global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a
// THREE-slot ring, two steps ahead of the MFMAs, and the loop runs on counted waits:
//     wait  vmcnt(12 | 0)     this wave's 12 pieces of step t have landed (the 12 of step t+1 may still fly)
//     s_barrier               ... in every wave; and every wave is done reading the slot step t+2 will overwrite
//     issue step t+2          12 LDS-DMA pieces per wave (48 KB per block and step)
//     MFMAs of step t         ds_read_b128 fragments out of slot t % 3
// One barrier per step, never a vmcnt(0) inside the loop.  The LDS image of a tile is lane-linear (1 KB = two 512-B rows per piece), so
// the bank-conflict-free layout comes from the SOURCE side: the 16-byte chunk j of row r is fetched into slot j ^ (r & 31) and read
// back from there (the same involution on both sides).  Tile 32(M) x 64(N), the four waves split each step's K range and their partial
// accumulators are summed through LDS like in the skinny kernel; the squeeze-excite gate of the block's image sits in LDS and multiplies
// the A fragment.  Reference op: nn.Conv2d(k=1) of MBConv's project conv (src/YetAnotherEfficientNet.py:446, gate :469-474).
#include "common.h"
#include "pw_args.h"
#include <cstdlib>

#define LK_BM 32
#define LK_BN 64
#define LK_BK 128
#define LK_STAGE (LK_BM * LK_BK + LK_BN * LK_BK)      // floats per ring slot: 12288 = 48 KB

typedef __attribute__((address_space(3))) void* lk_lds_vptr;
typedef const __attribute__((address_space(1))) void* lk_glb_vptr;

template <int PRO>      // 3: plain A operand, 4: squeeze-excite gate on A
__global__ __launch_bounds__(256) void pw_longk_kernel(PwArgs a) {
  extern __shared__ float smem[];                  // ONE array: [3][LK_STAGE] ring | gate[K]
  float* const sGate = smem + 3 * LK_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = mmd_xcd_swizzle(blockIdx.x, a.nblk);
  const int tn = t % a.ntn, tm = t / a.ntn;
  const int m0 = tm * LK_BM, n0 = tn * LK_BN;
  const int K = a.K, N = a.N, Mv = a.M;
  const int nk = (K + LK_BK - 1) / LK_BK;

  if constexpr (PRO == 4) {      // the block's 32 rows lie inside one image (rows_per_image % 32 == 0, checked on the host)
    const float* gp = a.gate + (size_t)(m0 / a.rows_per_image) * K;
    for (int k = tid * 4; k < nk * LK_BK; k += 1024)       // zero-padded to whole K steps: the tail fragments read inside the array
      *reinterpret_cast<float4*>(&sGate[k]) = k < K ? mmd_ld4(gp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();             // ordinary loads are finished before the first LDS-DMA is issued (hipcc would otherwise drain to vmcnt(0))
  }

  // ---- LDS-DMA piece bookkeeping.  A ring slot is 48 pieces of 1 KB: pieces 0..15 = A rows (2 per piece), 16..47 = B rows.  Wave w
  // issues pieces w, w+4, ... (12 per step).  Lane l of a piece writes LDS row (2 * pair + (l >> 5)), 16-byte slot (l & 31), and fetches
  // the chunk (slot ^ (row & 31)) of that row from global memory.
  const float* src[12];
  int kofs;                       // this lane's k offset (floats) inside a K step, before the tail clamp
  {
    const int sub = lane >> 5, slot = lane & 31;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = wave + 4 * i;
      if (piece < 16) {
        const int row = 2 * piece + sub;
        src[i] = a.x + (size_t)min(m0 + row, Mv - 1) * K;
      } else {
        const int row = 2 * (piece - 16) + sub;
        src[i] = a.w + (size_t)min(n0 + row, N - 1) * K;
      }
    }
    // (row & 31) of the lane's row: A piece p rows 2p + sub (p < 16 -> row < 32); B piece rows 2(p-16) + sub (0..63 -> & 31).
    // For both, row & 31 == (2 * (piece & 15) + sub) & 31 and piece & 15 == (wave + 4 i) & 15: computed per piece below.
    kofs = slot;
  }
  auto issue = [&](int step, int ring) {
    const int k0 = step * LK_BK;
    float* base = smem + ring * LK_STAGE;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = wave + 4 * i;
      const int rsw = (2 * (piece & 15) + (lane >> 5)) & 31;
      int k = k0 + ((kofs ^ rsw) << 2);
      k = min(k, K - 4);                                   // tail step: stay inside the row (the surplus products are zeroed on the A side)
      __builtin_amdgcn_global_load_lds((lk_glb_vptr)(uintptr_t)(src[i] + k), (lk_lds_vptr)(uintptr_t)(base + piece * 256), 16, 0, 0);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int ring = 0;
  for (int step = 0; step < nk; ++step) {
    if (step + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of step - 1 have returned
    __builtin_amdgcn_s_barrier();
    if (step + 2 < nk) issue(step + 2, ring == 0 ? 2 : ring - 1);      // (ring + 2) % 3: the slot read at step - 1
    // Fragment reads through inline asm: hipcc knows that an LDS-DMA writes LDS and, unable to tell the ring slots apart, puts a
    // vmcnt(0) in front of every ordinary ds_read while pieces are in flight - which would drain the prefetch each step.  The 12 (+4
    // gate) reads of the step and their lgkmcnt(0) sit in ONE statement, so the outputs are valid when it ends.
    // chunk(kk) = (8 wave + 2 kk + h) ^ r = ((8 wave + h) ^ r) ^ 2 kk (disjoint bits): one address register per kk, the B slabs and the
    // ring slot are immediate / scalar offsets.
    const unsigned abase = (unsigned)(uintptr_t)smem + (unsigned)ring * (LK_STAGE * 4) + (unsigned)r * (LK_BK * 4);
    const unsigned c0 = (unsigned)((wave * 8 + h) ^ r);
    const unsigned ad0 = abase + ((c0 ^ 0u) << 4), ad1 = abase + ((c0 ^ 2u) << 4), ad2 = abase + ((c0 ^ 4u) << 4), ad3 = abase + ((c0 ^ 6u) << 4);
    const int kw = step * LK_BK + wave * 32 + h * 4;        // this lane's k of kk = 0 (kk adds 8)
    f32x4 av[4], b0[4], b1[4], gv[4];
    if constexpr (PRO == 4) {
      const unsigned gad = (unsigned)(uintptr_t)sGate + (unsigned)kw * 4;
      asm volatile(
          "ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:16384\n\tds_read_b128 %2, %16 offset:32768\n\t"
          "ds_read_b128 %3, %17\n\tds_read_b128 %4, %17 offset:16384\n\tds_read_b128 %5, %17 offset:32768\n\t"
          "ds_read_b128 %6, %18\n\tds_read_b128 %7, %18 offset:16384\n\tds_read_b128 %8, %18 offset:32768\n\t"
          "ds_read_b128 %9, %19\n\tds_read_b128 %10, %19 offset:16384\n\tds_read_b128 %11, %19 offset:32768\n\t"
          "ds_read_b128 %12, %20\n\tds_read_b128 %13, %20 offset:32\n\tds_read_b128 %14, %20 offset:64\n\tds_read_b128 %15, %20 offset:96\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(av[0]), "=&v"(b0[0]), "=&v"(b1[0]), "=&v"(av[1]), "=&v"(b0[1]), "=&v"(b1[1]), "=&v"(av[2]), "=&v"(b0[2]), "=&v"(b1[2]),
            "=&v"(av[3]), "=&v"(b0[3]), "=&v"(b1[3]), "=&v"(gv[0]), "=&v"(gv[1]), "=&v"(gv[2]), "=&v"(gv[3])
          : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3), "v"(gad) : "memory");
    } else {
      asm volatile(
          "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:16384\n\tds_read_b128 %2, %12 offset:32768\n\t"
          "ds_read_b128 %3, %13\n\tds_read_b128 %4, %13 offset:16384\n\tds_read_b128 %5, %13 offset:32768\n\t"
          "ds_read_b128 %6, %14\n\tds_read_b128 %7, %14 offset:16384\n\tds_read_b128 %8, %14 offset:32768\n\t"
          "ds_read_b128 %9, %15\n\tds_read_b128 %10, %15 offset:16384\n\tds_read_b128 %11, %15 offset:32768\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(av[0]), "=&v"(b0[0]), "=&v"(b1[0]), "=&v"(av[1]), "=&v"(b0[1]), "=&v"(b1[1]), "=&v"(av[2]), "=&v"(b0[2]), "=&v"(b1[2]),
            "=&v"(av[3]), "=&v"(b0[3]), "=&v"(b1[3])
          : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);          // nothing that uses the fragments may be scheduled above the statement
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a4 = av[kk];
      if constexpr (PRO == 4) {
        a4 = a4 * gv[kk];
      }
      if (kw + kk * 8 >= K) a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b0[kk][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b1[kk][e], acc[1], 0, 0, 0);
      }
    }
    ring = ring == 2 ? 0 : ring + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // every wave is done with the ring: reuse it for the reduction

  // ---- cross-wave K reduction through LDS: part[wave][row][col], row-major 32 x 64
  float* part = smem;
  float* sRed = smem + LK_STAGE;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
      part[(wave * LK_BM + row) * LK_BN + j * 32 + r] = acc[j][q];
    }
  __syncthreads();
  // ---- epilogue: thread -> 4 consecutive columns (cg = tid & 15), rows (tid >> 4) + 16 i: dwordx4 stores
  const int cg = tid & 15, rgrp = tid >> 4, col = n0 + cg * 4;
  const bool cok = col < N;
  float4 b4 = make_float4(0, 0, 0, 0), osc = make_float4(1, 1, 1, 1), osh = make_float4(0, 0, 0, 0);
  if (cok) {
    if (a.bias) b4 = mmd_ld4(a.bias + col);
    if (a.out_scale) { osc = mmd_ld4(a.out_scale + col); osh = mmd_ld4(a.out_shift + col); }
  }
  float4 s4 = make_float4(0, 0, 0, 0), q4 = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = rgrp + 16 * i, row = m0 + rl;
    float4 v = *reinterpret_cast<const float4*>(&part[rl * LK_BN + cg * 4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(&part[(w * LK_BM + rl) * LK_BN + cg * 4]);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (cok && row < Mv) {
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
      q4.x += v.x * v.x; q4.y += v.y * v.y; q4.z += v.z * v.z; q4.w += v.w * v.w;
      if (a.out_scale) { v.x = v.x * osc.x + osh.x; v.y = v.y * osc.y + osh.y; v.z = v.z * osc.z + osh.z; v.w = v.w * osc.w + osh.w; }
      if (a.out_act) mmd_act4(v, a.out_act);
      const size_t off = (size_t)row * N + col;
      if (a.residual) { const float4 rr = mmd_ld4(a.residual + off); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
      mmd_st4(a.y + off, v);
    }
  }
  if (a.stats) {
    const int wv = tid >> 6;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
      q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
    }
    if (lane < 16) {
      *reinterpret_cast<float4*>(&sRed[wv * LK_BN + lane * 4]) = s4;
      *reinterpret_cast<float4*>(&sRed[4 * LK_BN + wv * LK_BN + lane * 4]) = q4;
    }
    __syncthreads();
    if (tid < LK_BN && n0 + tid < N) {
      const float s2 = sRed[tid] + sRed[LK_BN + tid] + sRed[2 * LK_BN + tid] + sRed[3 * LK_BN + tid];
      const float q2 = sRed[4 * LK_BN + tid] + sRed[5 * LK_BN + tid] + sRed[6 * LK_BN + tid] + sRed[7 * LK_BN + tid];
      atomicAdd(&a.stats[n0 + tid], (double)s2);
      atomicAdd(&a.stats[N + n0 + tid], (double)q2);
    }
  }
}


// -> 1 when the launch was taken.  Supported: fp32, plain or gate-only A operand, no pyramid / strided output, K >= 256, K % 4 == 0.
int pw_longk_try(PwArgs& a, hipStream_t stream) {
  static const int off = getenv("MMD_NO_LONGK") ? 1 : 0;
  if (off || a.form == MMD_PW_FORM_TILED || a.form == MMD_PW_FORM_ROWS || a.form == MMD_PW_FORM_SLAB || a.bf16 || a.bb.z || a.xs.z || a.st.Cin || a.pyr.n || a.y_batch_stride) return 0;
  if (a.in_scale || a.in_bn.stats || a.in_act != MMD_ACT_NONE) return 0;
  if (a.stats_ws) return 0;
  const int M = a.M, K = a.K, N = a.N;
  if (K < 256 || (K & 3) || K > 3072 || N < 4) return 0;
  if (a.gate && (a.rows_per_image % LK_BM)) return 0;
  const long long blocks = (long long)cdiv(M, LK_BM) * cdiv(N, LK_BN);
  // the regime this kernel is for: at most one block per CU - the 144 KB ring allows one resident block, so a 257th block waits for a
  // whole block time.  Measured (tools/dev/one_rows.py, profiles/r02_notes.md): M2048 K1248 N208 24.5 -> 20.5 us, M2048 K720 N208 16.4 ->
  // 13.9, but M8192 K528 N88 (512 blocks) 20.8 -> 22.4 and M2048 K2112 N352 (384 blocks) 57.2 -> 58.9.
  if (a.form != MMD_PW_FORM_LONGK && (K < 512 || blocks > 256)) return 0;
  a.ntn = cdiv(N, LK_BN); a.nblk = (int)blocks;
  const size_t lds = ((size_t)3 * LK_STAGE + (a.gate ? (size_t)cdiv(K, LK_BK) * LK_BK : 0)) * sizeof(float);
  if (lds > 160 * 1024) return 0;
  auto kern = a.gate ? pw_longk_kernel<4> : pw_longk_kernel<3>;
  static bool attr_done[2];
  if (!attr_done[a.gate ? 1 : 0]) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[a.gate ? 1 : 0] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(256), lds, stream, a);
  return 1;
}
