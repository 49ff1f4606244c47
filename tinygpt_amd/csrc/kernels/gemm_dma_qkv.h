// gemm_dma_qkv.h — the QKV product of a 16-bit prompt on the LDS-DMA rings of gemm_dma.h: Q columns as two activation terms, K | V columns as three (their
// results are rounded into the cache), as ONE balanced launch — four waves with separate Q and K | V workgroups (gemm_dma_qkv_kernel), or eight waves with the Q
// tile and the K | V tile of a row block on one staging of the activation lines (gemm_dma_qkv8_kernel; template ROPE: + RoPE, the cache append — paged or not —
// and the q split in its epilogue).  Split from gemm_dma.h in round 6 (no code change); == the q_proj | k_proj | v_proj slices of MergedLinear (Linear.h:64-79)
// followed, with ROPE, by Attention.h:96-106.
#pragma once
#include "gemm_dma.h"

namespace tgx {

// The QKV product of a bf16 prompt as ONE balanced launch (round 3).  Its Q columns take two split terms, its K / V columns three (their results are
// rounded into the cache): as one grid of 128 x 128 tiles that is 256 two-term + 128 three-term workgroups on 256 CUs — 1.5 per CU, and a CU that
// draws two three-term tiles carries 6 units of work against an average of 3.5.  Here the first `nq` workgroups take the Q columns in 128-row tiles
// (2 units each), the rest take the K / V columns in 64-ROW tiles with three terms (1.5 units each): 256 + 256 workgroups on Llama-3.2-1B at 2048
// tokens, one of each kind co-resident per CU (49 + 41 KB of LDS) — 3.5 units everywhere.
template <int DT, int DBK, int NS>
__global__ __launch_bounds__(256) void gemm_dma_qkv_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int qcols = a.three_from / GBN, nq = qcols * ((a.M + 127) / 128);
  const int b = (int)blockIdx.x;
  if (b < nq) {
    const int ry = b / qcols, cx = b - ry * qcols;
    gemm_dma_tile<DT, GEMM_STORE, 2, DBK, NS>(a, ry * 128, cx * GBN, false, dma_lds, 0, a.K, 0);
  } else {
    const int kcols = (a.N - a.three_from + GBN - 1) / GBN, k = b - nq;
    const int ry = k / kcols, cx = k - ry * kcols;
    gemm_dma_tile<DT, GEMM_STORE, 1, DBK, NS>(a, ry * 64, a.three_from + cx * GBN, true, dma_lds, 0, a.K, 0);
  }
}

// The QKV product of a bf16 prompt as ONE launch of eight-wave workgroups that SHARE the activation lines (round 5).  gemm_dma_qkv_kernel above is bound by operand
// delivery — its two co-resident workgroups ask 44 KB of half lines per k32 step of a CU for 56 MFMAs per SIMD-quad (MfmaUtil 30 %) — and each of them fetches its own
// rows of the activation terms.  Here workgroup (row block rb, column group g) owns 128 rows and BOTH the g-th 128-column tile of Q (waves 0-3, two terms, 64 x 64 per
// wave) and the g-th 64-column tile of K | V (waves 4-7, three terms, 32 x 64 per wave): the three 16-KB term tiles of a k64 stage are staged once for both, next to the
// two weight tiles (16 + 8 KB), 72 KB of whole 128-byte lines per stage, two stages.  Needs as many 128-column Q tiles as 64-column K | V tiles (q_dim = 4 kv_dim:
// Llama-3.2-1B, Mistral-7B); one workgroup per CU at S = 2048, 2 + 1.5 units of work each.  Same MFMAs per accumulator in the same order as gemm_dma_qkv_kernel.
// ROPE (head_dim 64: a Q wave's 64 columns are one query head, a K | V wave's 64 columns one kv head, and the rotation partners (d, d + 32) are the two column blocks
// of ONE lane): the epilogue adds the bias, rotates q and k at the row's position, splits q into its two 16-bit terms, rounds k and v into the cache — the
// rope_kv_split launch and the fp32 QKV matrix disappear.  Values pass through the idle ring ([rows][72] 16-bit: 144-byte rows) and leave as whole 128-byte rows.
template <int DT, bool ROPE = false>
__global__ __launch_bounds__(512) void gemm_dma_qkv8_kernel(const GemmArgs a) {
  constexpr int DBK = 64, NS = 2;
  constexpr int T_A = 128 * DBK, T_BQ = 128 * DBK, T_BK = 64 * DBK;       // 16-bit elements: a term tile, the Q weight tile, the K | V weight tile
  constexpr int STAGE = 3 * T_A + T_BQ + T_BK;                            // 72 KB
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)dma_lds;
  const int ncg = a.three_from / GBN;                                     // column groups
  // workgroup id -> (row block, column group): XCD (id % 8) takes whole row blocks when they divide (its L2 then holds a row block's activation lines for all of
  // the block's column groups, and only the weights are fetched once per XCD); performance only
  int rb = (int)blockIdx.x / ncg, cg = (int)blockIdx.x - rb * ncg;
  {
    const int nrb = (int)gridDim.x / ncg;
    if ((nrb & 7) == 0) {
      const int xcd = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
      rb = xcd * (nrb >> 3) + l / ncg; cg = l % ncg;
    }
  }
  const int m0 = rb * 128, nq0 = cg * 128, nk0 = a.three_from + cg * 64;

  // DMA map: a piece = 8 rows of 128 bytes; lane l -> row (l >> 3) of the piece, LDS slot l & 7 <- global chunk slot ^ ((row >> 1) & 7).  Wave w moves pieces w, w + 8 of
  // each term tile and of the Q weight tile, and piece w of the K | V weight tile: nine per stage
  const int prow = lane >> 3, pslot = lane & 7;
  const bf16_t* gsrc[9];
  unsigned ldst[9];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int piece = wv + 8 * p, row = piece * 8 + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    const size_t ga = (size_t)min(m0 + row, a.M - 1) * a.K + chunk * 8;
    gsrc[4 * p] = a.A_hi + ga; gsrc[4 * p + 1] = a.A_lo + ga; gsrc[4 * p + 2] = a.A_lo2 + ga;
    gsrc[4 * p + 3] = a.B + (size_t)min(nq0 + row, a.N - 1) * a.K + chunk * 8;
    ldst[4 * p] = (unsigned)(piece * 1024); ldst[4 * p + 1] = ldst[4 * p] + (unsigned)(T_A * 2); ldst[4 * p + 2] = ldst[4 * p] + (unsigned)(2 * T_A * 2);
    ldst[4 * p + 3] = ldst[4 * p] + (unsigned)(3 * T_A * 2);
  }
  {
    const int row = wv * 8 + prow, chunk = pslot ^ ((row >> 1) & 7);
    gsrc[8] = a.B + (size_t)min(nk0 + row, a.N - 1) * a.K + chunk * 8;
    ldst[8] = (unsigned)((3 * T_A + T_BQ) * 2 + wv * 1024);
  }
  auto issue_stage = [&](int k0, int stage) {
    const unsigned sb = lds_base + (unsigned)(stage * STAGE * 2);
#pragma unroll
    for (int q = 0; q < 9; q++) dma_1k(gsrc[q] + k0, sb + ldst[q]);
  };
  auto frag = [&](const bf16_t* tile, int row, int kchunk) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(tile + row * DBK + ((kchunk ^ ((row >> 1) & 7)) << 3));
  };

  const bool qwave = wv < 4;                                             // wave-uniform
  const int w4 = wv & 3, wm = w4 >> 1, wn = w4 & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nk = a.K / DBK;
  issue_stage(0, 0);
  for (int k = 0; k < nk; k++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // stage k landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                      // ... every wave's; and every wave is done reading stage k - 1
    const bool more = k + 1 < nk;
    const int nk0 = (k + 1) * DBK;
    const unsigned nsb = lds_base + (unsigned)(((k + 1) % NS) * STAGE * 2);
    const bf16_t* st = dma_lds + (size_t)(k % NS) * STAGE;
    const bf16_t *tAh = st, *tAl = st + T_A, *tAl2 = st + 2 * T_A, *tBq = st + 3 * T_A, *tBk = tBq + T_BQ;
    // the nine pieces of the next stage leave three at a time behind the first three k16 steps (their issue hides under the matrix pipe)
    if (qwave) {
#pragma unroll
      for (int kk = 0; kk < DBK / 16; kk++) {
        if (more && kk < 3) {
#pragma unroll
          for (int q = 3 * kk; q < 3 * kk + 3; q++) dma_1k(gsrc[q] + nk0, nsb + ldst[q]);
        }
        const int kchunk = kk * 2 + (lane >> 5);
        bf16x8 fah[2], fal[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const int row = wm * 64 + i * 32 + (lane & 31);
          fah[i] = frag(tAh, row, kchunk);
          fal[i] = frag(tAl, row, kchunk);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) fb[j] = frag(tBq, wn * 64 + j * 32 + (lane & 31), kchunk);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[i][j] = mfma16<DT>(fal[i], fb[j], acc[i][j]);   // small term first
            acc[i][j] = mfma16<DT>(fah[i], fb[j], acc[i][j]);
          }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < DBK / 16; kk++) {
        if (more && kk < 3) {
#pragma unroll
          for (int q = 3 * kk; q < 3 * kk + 3; q++) dma_1k(gsrc[q] + nk0, nsb + ldst[q]);
        }
        const int kchunk = kk * 2 + (lane >> 5);
        const int row = w4 * 32 + (lane & 31);
        const bf16x8 fah = frag(tAh, row, kchunk), fal = frag(tAl, row, kchunk), fal2 = frag(tAl2, row, kchunk);
        bf16x8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; j++) fb[j] = frag(tBk, j * 32 + (lane & 31), kchunk);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[0][j] = mfma16<DT>(fal2, fb[j], acc[0][j]);       // smallest term first
          acc[0][j] = mfma16<DT>(fal, fb[j], acc[0][j]);
          acc[0][j] = mfma16<DT>(fah, fb[j], acc[0][j]);
        }
      }
    }
  }

  if constexpr (ROPE) {
    constexpr int RS = 72;                                    // 16-bit elements per staged row
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave is done with the ring
    bf16_t* const sh = dma_lds + (size_t)wv * (2 * 64 * RS);  // this wave's slice: hi (or the cache image) rows, then lo rows
    bf16_t* const sl = sh + 64 * RS;
    const int p = lane & 31, hh = lane >> 5;
    if (qwave) {
      const int head = (nq0 + wn * 64) >> 6;
      const float b0 = a.bias ? elem_to_f32<DT>(a.bias[head * 64 + p]) : 0.f, b1 = a.bias ? elem_to_f32<DT>(a.bias[head * 64 + 32 + p]) : 0.f;
      float cs[2][16], sn[2][16];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = min(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, a.M - 1);
          cs[i][r] = a.rope_cos[(size_t)(a.rope_past + row) * 32 + p]; sn[i][r] = a.rope_sin[(size_t)(a.rope_past + row) * 32 + p];
        }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float x0 = acc[i][0][r] + b0, x1 = acc[i][1][r] + b1;
          rope_rotate_pair(x0, x1, cs[i][r], sn[i][r]);
          bf16_t h0, l0, h1, l1;
          split16<DT>(x0, h0, l0); split16<DT>(x1, h1, l1);
          const int o = (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * RS + p;
          sh[o] = h0; sh[o + 32] = h1; sl[o] = l0; sl[o + 32] = l1;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the wave reads only what it wrote itself
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int rl = q * 8 + (lane >> 3), row = m0 + wm * 64 + rl;
        const u32x4 vh = *reinterpret_cast<const u32x4*>(sh + rl * RS + (lane & 7) * 8);
        const u32x4 vl = *reinterpret_cast<const u32x4*>(sl + rl * RS + (lane & 7) * 8);
        if (row < a.M) {
          const size_t o = (size_t)row * a.three_from + head * 64 + (lane & 7) * 8;
          *reinterpret_cast<u32x4*>(a.rope_q_hi + o) = vh;
          *reinterpret_cast<u32x4*>(a.rope_q_lo + o) = vl;
        }
      }
    } else {
      const bool isk = cg < a.rope_kv_heads;                  // column group cg: key head cg, or value head cg - kv_heads
      const int kvh = isk ? cg : cg - a.rope_kv_heads;
      const float b0 = a.bias ? elem_to_f32<DT>(a.bias[nk0 + p]) : 0.f, b1 = a.bias ? elem_to_f32<DT>(a.bias[nk0 + 32 + p]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh, row = min(m0 + w4 * 32 + rl, a.M - 1);
        float x0 = acc[0][0][r] + b0, x1 = acc[0][1][r] + b1;
        if (isk) {
          const float c = a.rope_cos[(size_t)(a.rope_past + row) * 32 + p], s = a.rope_sin[(size_t)(a.rope_past + row) * 32 + p];
          rope_rotate_pair(x0, x1, c, s);
        }
        sh[rl * RS + p] = f32_to_elem<DT>(x0); sh[rl * RS + 32 + p] = f32_to_elem<DT>(x1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // paged KV (rope_tbl set): rope_k / rope_v are the layer's pools and a row's page comes from the sequence's block table (common.h kv_paged_off)
      bf16_t* const cache = (isk ? a.rope_k : a.rope_v) + (a.rope_tbl ? (size_t)0 : (size_t)kvh * a.rope_max_ctx * 64);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int rl = q * 8 + (lane >> 3), row = m0 + w4 * 32 + rl, pos = a.rope_past + min(row, a.M - 1);
        const u32x4 v = *reinterpret_cast<const u32x4*>(sh + rl * RS + (lane & 7) * 8);
        const int page = a.rope_tbl ? a.rope_tbl[pos >> KV_BLOCK_SHIFT] : 0;
        const size_t o = a.rope_tbl ? (((size_t)page * a.rope_kv_heads + kvh) * KV_BLOCK + (pos & (KV_BLOCK - 1))) * 64 : (size_t)pos * 64;
        if (row < a.M) *reinterpret_cast<u32x4*>(cache + o + (lane & 7) * 8) = v;
      }
    }
    return;
  }
  // epilogue: fp32 rows of the QKV matrix (+ bias); C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  auto store_block = [&](const f32x16& v, int row0, int col) __attribute__((always_inline)) {
    if (col >= a.N) return;
    const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < a.M) a.C[(size_t)row * a.ldc + col] = v[r] + bv;
    }
  };
  if (qwave) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) store_block(acc[i][j], m0 + wm * 64 + i * 32, nq0 + wn * 64 + j * 32 + (lane & 31));
  } else {
#pragma unroll
    for (int j = 0; j < 2; j++) store_block(acc[0][j], m0 + w4 * 32, nk0 + j * 32 + (lane & 31));
  }
}

}  // namespace tgx
