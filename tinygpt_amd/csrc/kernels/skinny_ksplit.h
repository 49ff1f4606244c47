// skinny_ksplit.h — Y[M][N] = X[M][K] . W[N][K]^T for M <= 16 activation rows and a WIDE product (N >= 64 x CUs: gate_up, lm_head) without
// a barrier in the K loop.
//
// Why a second skinny kernel: skinny.h stages every 256-k activation panel in LDS for the four waves of a workgroup (two barriers, RMSNorm and
// 16-bit split per panel, redone by each of the N / 64 workgroups) and is bound by that serial chain, not by the weight stream
// (tools/probes/skinny_probe.hip: 15.2 of 18.5 µs remain with the weight refills compiled out).  Here the four waves split K instead:
//   * every wave owns all 64 weight rows of its workgroup for a quarter of K; weight tiles (64 rows x 64 k: one 128-byte line per row) go
//     global -> registers (three slots in flight per wave, non-temporal) -> a wave-private LDS tile (transposition into MFMA fragments only);
//   * the activation fragments come straight from memory as 16-bit terms prepared ONCE per layer by rmsnorm_split_kernel (lane (m, g) reads
//     16 bytes of row m), prefetched as deep as the weight slots — loads retire in order, so a fragment fetched later than a weight tile
//     would drain that tile with it;
//   * the four partial accumulators meet once, through LDS, in wave order (deterministic); wave w finishes weight-row block w.
// Prototype numbers (tools/probes/skinny3_probe.hip, gate_up of Llama-3.2-1B, 8 rows): 12.8-13.5 µs = 5.0-5.2 TB/s against 16.6-18.8 µs.
// Every load of the K loop is unconditional from a clamped address and every register slot a compile-time constant (the loop body covers the
// three slots; a trailing partial trip multiplies by zeroed activation fragments), so the compiler's vmcnt waits stay counted.
//
// Needs K % 256 == 0 (a whole number of 64-k tiles per wave) and M <= 32; the launcher falls back to skinny.h otherwise.
#pragma once
#include "skinny.h"

namespace tgx {

// KFIX: K known at compile time (2048 / 3072 / 4096: the hidden sizes of the benchmark configs) — the K loop is then fully unrolled, no refill
// is issued past the range and no trailing dead tile exists (prototype rate: 12.8 µs); 0 = run-time K (14.9 µs on the same product).
// MB = 16-row blocks of activation rows (1: M <= 16 with three weight slots in flight per wave; 2: M <= 32 with two — a third slot next to
// the second block's fragments exceeds 256 VGPRs, and the refills would then bounce through AGPRs with a full drain each).
template <int DT, int EPI, int KFIX, int MB>
__global__ __launch_bounds__(256) void skinny_ksplit_kernel(const GemmArgs a) {
  constexpr int KT = 64, LDW = KT + 8, SLOTS = MB == 1 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) bf16_t lds[4 * 64 * LDW];          // wave-private weight tiles; reused for the final reduction
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  bf16_t* sW = lds + wv * 64 * LDW;
  const int n0 = blockIdx.x * 64;
  const int kw = a.K >> 2, kw0 = wv * kw, tiles = kw / KT;
  const bool inter = EPI == GEMM_SILU;

  // weight tile loads: instruction i covers tile rows 8 i .. 8 i + 7 (8 lanes x 16 B = one line per row)
  const int lrow = lane >> 3, chunk = lane & 7;
  const bf16_t* wrow[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int nb = min(n0 + 8 * i + lrow, a.N - 1);           // clamped: rows past N reload the last one, masked at the store
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    wrow[i] = a.B + brow * a.K + kw0 + 8 * chunk;
  }
  // refills past the K range stay unconditional (the waits count them) but all lanes then read ONE 16-byte word: a single cache line, not a tile
  auto load_w = [&](int t, u32x4* r) {
    const bool real = t < tiles;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = load_nt(reinterpret_cast<const u32x4*>(real ? wrow[i] + t * KT : a.B));
  };
  // activation fragments of one tile: 2 k-steps x 2 terms (row m = lane & 15 clamped to M - 1, masked at the store; k = ... + 8 (lane >> 4))
  const int am = lane & 15, ag = lane >> 4;
  size_t aoff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++) aoff[mb] = (size_t)min(16 * mb + am, a.M - 1) * a.K + kw0 + 8 * ag;
  auto load_a = [&](int t, u32x4 (*dst)[MB][2]) {
    const bool real = t < tiles;
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int mb = 0; mb < MB; mb++) {
        dst[ks][mb][0] = *reinterpret_cast<const u32x4*>(real ? a.A_hi + aoff[mb] + t * KT + ks * 32 : a.A_hi);
        dst[ks][mb][1] = *reinterpret_cast<const u32x4*>(real ? a.A_lo + aoff[mb] + t * KT + ks * 32 : a.A_lo);
      }
  };
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < 4; nb++) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 w[SLOTS][8], fa[SLOTS][2][MB][2];
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) { load_a(sl, fa[sl]); load_w(sl, w[sl]); }
  __builtin_amdgcn_sched_barrier(0);

  auto tile = [&](int t, u32x4* ws, u32x4 (*fs)[MB][2], bool refill) {
    const bool live = t < tiles;                                // a trailing partial trip: the tile is a reload, its activations count as zero
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4*>(&sW[(8 * i + lrow) * LDW + chunk * 8]) = ws[i];
    u32x4 fc[2][MB][2];                                          // this tile's fragments leave their slot before it is refilled
    const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int mb = 0; mb < MB; mb++) { fc[ks][mb][0] = live ? fs[ks][mb][0] : zero; fc[ks][mb][1] = live ? fs[ks][mb][1] : zero; }
    if (refill) { load_a(t + SLOTS, fs); load_w(t + SLOTS, ws); }
    __builtin_amdgcn_sched_barrier(0);                           // the refill is issued HERE: the scheduler would sink it towards its use
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 fb[4];
#pragma unroll
      for (int nb = 0; nb < 4; nb++) fb[nb] = *reinterpret_cast<const bf16x8*>(&sW[(16 * nb + am) * LDW + ks * 32 + 8 * ag]);
#pragma unroll
      for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int nb = 0; nb < 4; nb++) {
          acc[mb][nb] = mfma16x16<DT>(__builtin_bit_cast(bf16x8, fc[ks][mb][1]), fb[nb], acc[mb][nb]);      // small term first
          acc[mb][nb] = mfma16x16<DT>(__builtin_bit_cast(bf16x8, fc[ks][mb][0]), fb[nb], acc[mb][nb]);
        }
    }
  };
  if constexpr (KFIX > 0) {
    constexpr int T = KFIX / 4 / KT;
#pragma unroll
    for (int t = 0; t < T; t++) tile(t, w[t % SLOTS], fa[t % SLOTS], t + SLOTS < T);      // every slot and every refill decision is a constant
  } else {
    for (int t0 = 0; t0 < tiles; t0 += SLOTS) {
      tile(t0, w[0], fa[0], true);
      tile(t0 + 1, w[1], fa[1], true);
      if constexpr (SLOTS == 3) tile(t0 + 2, w[2], fa[2], true);
    }
  }

  // the four k-quarters meet in LDS; wave w finishes weight-row block w (sum in wave order: deterministic)
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);                    // [wave][mb][nb][r][lane]
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < 4; nb++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[(((wv * MB + mb) * 4 + nb) * 4 + r) * 64 + lane] = acc[mb][nb][r];
  __syncthreads();
  const int col = n0 + 16 * wv + am;
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; w2++) v += red[(((w2 * MB + mb) * 4 + wv) * 4 + r) * 64 + lane];
      const int row = 16 * mb + 4 * ag + r;
      if (EPI == GEMM_SILU) {        // even lanes hold gate_i, odd lanes up_i (i = col / 2): the pair meets over the DPP crossbar
        const float other = dpp_mov<0xB1, 0xf>(v);
        if ((lane & 1) || col >= a.N || row >= a.M) continue;
        const size_t o = (size_t)row * a.inter + (size_t)(col >> 1);
        split16<DT>((v / (1.0f + expf(-v))) * other, a.out_hi[o], a.out_lo[o]);
        continue;
      }
      if (col >= a.N || row >= a.M) continue;
      a.C[(size_t)row * a.ldc + col] = v + (a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f);
    }
}

}  // namespace tgx
