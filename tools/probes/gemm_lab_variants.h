// gemm_lab_variants.h — candidate prefill GEMM kernels that were measured in tools/probes/gemm_lab.hip and did NOT enter the library (round 5).
// (gemm_dma8i_kernel, the full-line form, moved to kernels/gemm_dma.h.)
// Included after kernels/gemm_dma.h (uses its helpers).  Numbers: profiles/r05_prefill.txt.
#pragma once
#include "archive/gemm_dma_r05_lab.h"      // round 5's header with its lab-only template switches (DIS / PP / TEPI / WJ); the product header dropped them in round 6

namespace tgx {

template <int DT, int EPI, int DIS = 0, int PP = 0>
__global__ __launch_bounds__(512) void gemm_dma8ip_kernel(const GemmArgs a) {
  constexpr int TMN = 256, NSLOT = 5, UNIT = TMN * 64;     // 16-bit elements per unit: 256 rows x 128 bytes
  constexpr int WI = 2, WJ = 4;
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  int tile_m, tile_n;
  xcd_tile(a, tile_m, tile_n);
  const int m0 = tile_m * TMN, n0 = tile_n * TMN;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;
  const bool inter = EPI == GEMM_SILU;

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // DMA map: a unit = 32 pieces of 1 KiB = 8 rows x 128 bytes each; wave w takes pieces w, w + 8, w + 16, w + 24; chunk c of row r sits in slot c ^ ((r >> 1) & 7)
  const int prow = lane >> 3, pslot = lane & 7;
  const bf16_t *srcA[4], *srcB[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int row = (wv + 8 * p) * 8 + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    srcA[p] = a.A_hi + (size_t)min(m0 + row, a.M - 1) * (2 * (size_t)a.K) + chunk * 8;        // interleaved rows are 2 K elements long
    const int nb = min(n0 + row, a.N - 1);
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    srcB[p] = a.B + brow * a.K + chunk * 8;
  }
  const int nk = a.K / 32, nblk = a.K / 64;
  // unit u = 3 b + j: j = 0 the A lines of step 2b, 1 the B lines of block b, 2 the A lines of step 2b + 1 (units past the end reload the last block)
  auto issue = [&](int b, int j, int p) {
    if ((DIS & 2) && b > 0) return;
    const int bb = min(b, nblk - 1);
    const unsigned dst = lds_base + (unsigned)(((3 * b + j) % NSLOT) * UNIT * 2) + (unsigned)((wv + 8 * p) * 1024);
    if (j == 1) dma_1k(srcB[p] + (size_t)bb * 64, dst);
    else dma_1k(srcA[p] + (size_t)(2 * bb + (j >> 1)) * 64, dst);
  };
  const int swz = ((lane & 31) >> 1) & 7;
  const int arow = (wm * 64 + (lane & 31)) * 64, brow_l = (wn * 128 + (lane & 31)) * 64;
  // fragments of k16 step kk of k32 step s: A chunks (term * 4 + kk * 2 + half), B chunks ((s & 1) * 4 + kk * 2 + half) of the block's lines
  auto read_frags = [&](int s, int kk, bf16x8* fa, bf16x8* fb) {
    if (DIS & 4) return;
    const int b = s >> 1;
    const bf16_t* ua = dma_lds + (size_t)((3 * b + ((s & 1) << 1)) % NSLOT) * UNIT;
    const bf16_t* ub = dma_lds + (size_t)((3 * b + 1) % NSLOT) * UNIT;
    const int ca = kk * 2 + (lane >> 5), cb = (s & 1) * 4 + ca;
#pragma unroll
    for (int j = 0; j < WJ; j++) fb[j] = *reinterpret_cast<const bf16x8*>(ub + brow_l + j * 32 * 64 + ((cb ^ swz) << 3));
#pragma unroll
    for (int i = 0; i < WI; i++) {
      fa[2 * i] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + ((ca ^ swz) << 3));
      fa[2 * i + 1] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + (((4 + ca) ^ swz) << 3));
    }
  };
  // two fragments of k16 step kk of k32 step s: part 0, 1 = B blocks (2 part, 2 part + 1); part 2, 3 = both terms of A block part - 2
  auto read_pair = [&](int s, int kk, int part, bf16x8* fa, bf16x8* fb) __attribute__((always_inline)) {
    if (DIS & 4) return;
    const int b = s >> 1;
    const bf16_t* ua = dma_lds + (size_t)((3 * b + ((s & 1) << 1)) % NSLOT) * UNIT;
    const bf16_t* ub = dma_lds + (size_t)((3 * b + 1) % NSLOT) * UNIT;
    const int ca = kk * 2 + (lane >> 5), cb = (s & 1) * 4 + ca;
    if (part < 2) {
#pragma unroll
      for (int j = 2 * part; j < 2 * part + 2; j++) fb[j] = *reinterpret_cast<const bf16x8*>(ub + brow_l + j * 32 * 64 + ((cb ^ swz) << 3));
    } else {
      const int i = part - 2;
      fa[2 * i] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + ((ca ^ swz) << 3));
      fa[2 * i + 1] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + (((4 + ca) ^ swz) << 3));
    }
  };
  // ping-pong: waves w and w + 4 (one SIMD) half a k32 step apart; a phase ends with one workgroup barrier; group 1 enters one phase late.  Step s is read in
  // phases 2s (group 0) and 2s + 1 (group 1).  Per k64 block b a wave issues, in its memory phase of step 2b, its pieces of units 3b+3, 3b+4 (A of step 2b+2,
  // B of block b+1: freed by phase 4b-1) and in that of step 2b+1 unit 3b+5 (A of step 2b+3, freed by phase 4b+1); each is read from three phases later.
  const int grp = wv >> 2;
#ifdef LAB_TIMING
  long long tm[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_readcyclecounter();
#define LAB_TICK(i) do { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define LAB_TICK(i) do {} while (0)
#endif
#pragma unroll
  for (int u = 0; u < 3; u++)
#pragma unroll
    for (int p = 0; p < 4; p++) issue(0, u, p);
  bf16x8 fa[2][2 * WI], fb[2][WJ];
  if (DIS & 4) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int i = 0; i < 2 * WI; i++) fa[h][i] = bf16x8{};
#pragma unroll
      for (int j = 0; j < WJ; j++) fb[h][j] = bf16x8{};
    }
  }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // units 0, 1 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  if (PP & 4) { if (grp) __builtin_amdgcn_s_setprio(1); }
  if (grp) __builtin_amdgcn_s_barrier();
  auto matrix_phase = [&]() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(3);
    if (PP & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int i = 0; i < WI; i++)
#pragma unroll
        for (int j = 0; j < WJ; j++) {
          if (!(DIS & 1)) {
            acc[i][j] = mfma16<DT>(fa[h][2 * i + 1], fb[h][j], acc[i][j]);   // small term first
            acc[i][j] = mfma16<DT>(fa[h][2 * i], fb[h][j], acc[i][j]);
          }
        }
    if (PP & 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(4);
    __builtin_amdgcn_s_barrier();
    LAB_TICK(5);
  };
#ifdef LAB_TIMING
  tprev = __builtin_readcyclecounter();
#endif
  for (int b = 0; b < nblk; b++) {
    // ---- step 2b: memory phase
    if constexpr (PP & 16) {      // one DMA piece behind every two fragment reads: the wave reaches its next piece when the CU's memory pipeline has taken the others'
#pragma unroll
      for (int q = 0; q < 8; q++) {
        read_pair(2 * b, q >> 2, q & 3, fa[q >> 2], fb[q >> 2]);
        __builtin_amdgcn_sched_barrier(0);
        issue(b + 1, q >> 2, q & 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      LAB_TICK(0);
    } else {
    read_frags(2 * b, 0, fa[0], fb[0]);
    read_frags(2 * b, 1, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(0);
#pragma unroll
    for (int p = 0; p < 4; p++) issue(b + 1, 0, p);
#pragma unroll
    for (int p = 0; p < 4; p++) issue(b + 1, 1, p);
    __builtin_amdgcn_sched_barrier(0);
    }
    LAB_TICK(1);
    if (DIS & 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // unit 3b+2 landed
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(2);
    matrix_phase();
    // ---- step 2b+1
    if constexpr (PP & 16) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        read_pair(2 * b + 1, q >> 2, q & 3, fa[q >> 2], fb[q >> 2]);
        __builtin_amdgcn_sched_barrier(0);
        if (q & 1) { issue(b + 1, 2, q >> 1); __builtin_amdgcn_sched_barrier(0); }
      }
      LAB_TICK(0);
    } else {
    read_frags(2 * b + 1, 0, fa[0], fb[0]);
    read_frags(2 * b + 1, 1, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(0);
#pragma unroll
    for (int p = 0; p < 4; p++) issue(b + 1, 2, p);
    __builtin_amdgcn_sched_barrier(0);
    }
    LAB_TICK(1);
    if (DIS & 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // units 3b+3, 3b+4 landed
    __builtin_amdgcn_sched_barrier(0);
    LAB_TICK(2);
    matrix_phase();
  }
  if (!grp) __builtin_amdgcn_s_barrier();
#ifdef LAB_TIMING
  if (a.ssq_out && blockIdx.x == 3 && blockIdx.y == 2 && lane == 0) {
    long long* o = reinterpret_cast<long long*>(a.ssq_out) + wv * 8;
#pragma unroll
    for (int i = 0; i < 6; i++) o[i] = tm[i];
    o[6] = nblk;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (DIS & 8) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < WI; i++)
#pragma unroll
      for (int j = 0; j < WJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) t += acc[i][j][r];
    if (t == 12345.678f) a.C[0] = t;
    return;
  }
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++) {
      const int col = n0 + wn * 128 + j * 32 + (lane & 31);
      if (EPI == GEMM_SILU) {
        silu_block_store<DT>(acc[i][j], lane, col, m0 + wm * 64 + i * 32, a);
        continue;
      }
      if (col >= a.N) continue;
      const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const float v = acc[i][j][r] + bv;
        if (EPI == GEMM_GELU) {
          const size_t o = (size_t)row * a.N + col;
          split16<DT>(gelu_new_fast(v), a.out_hi[o], a.out_lo[o]);
          continue;
        }
        float* dst = a.C + (size_t)row * a.ldc + col;
        *dst = (EPI == GEMM_RESIDUAL) ? (*dst + v) : v;
      }
    }
}


}  // namespace tgx
