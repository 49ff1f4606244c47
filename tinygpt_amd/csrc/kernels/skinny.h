// skinny.h — Y[M][N] = X[M][K] · W[N][K]ᵀ for a handful of activation rows (M <= 32) on the matrix cores: the nn::Linear of a [B,1]
// decode batch (GPTEngine.cpp:154-168 pushes any batch through each Linear) and of prompts of a few tokens.  The batched GEMV covers
// B <= 4 rows per pass over the weights (its R activation vectors live in registers); beyond that this kernel streams every weight
// byte ONCE for up to 32 rows.
//
// Roofline: HBM — 2*N*K bytes of weights per launch; the activations (M*K*4 bytes) are re-read per row group from L2.  MFMA work is
// 2-3 x 2*32*N*K flop at most (M padded to 16 or 32): far below the matrix peak at HBM rate.
//
// Structure (gfx950): one workgroup = 4 waves = 64 or 128 weight rows (tile columns) x a K range; grid = (N / rows, K splits).
//   weights  wave w owns NBW blocks of 16 rows.  A W tile (8 KB: 32 rows x 128 k or 16 rows x 256 k) is fetched with 8 non-temporal 16-byte
//            loads per lane (>= 256 contiguous bytes per row and wave-load) two tiles ahead of use (16 KB in flight per wave), then passes
//            through a wave-private LDS tile only to be transposed into MFMA fragments (no workgroup barrier on the weight path).
//   x        panels of 256 k of all M rows are staged in LDS by the whole workgroup as hi / lo / optional third 16-bit term (prefill.h's
//            exact split of fp32 activations), register-prefetched one panel ahead; two barriers per panel.  The source is either the
//            terms in memory (ASRC 0), fp32 rows split on the way (ASRC 1), or fp32 rows with RMSNorm applied on the way (ASRC 2) —
//            the reference's RMSNorm -> Linear pair (DecoderLayer.h:40-41) in one launch.
//   math     v_mfma_f32_16x16x32 (bf16 or f16): A = x rows (16 per block, MB blocks), B = 16 weight rows; fp32 accumulators
//            D[m][n]: col n = lane&15, row m = 4*(lane>>4)+reg — so a store instruction writes 16 consecutive n for 4 rows.
//   epilogue as the prefill GEMM: store(+bias), residual add, siluMul on gate/up-interleaved columns, or a split-K partial slab
//            finished by reduce_rows_kernel / rope_kv_rows_kernel below (z-ordered sums: deterministic).
#pragma once
#include "prefill.h"

namespace tgx {

template <int DT>
__device__ __forceinline__ f32x4 mfma16x16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (DT == DT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int SK_NCB = 8;             // column blocks of the row-finishing kernels = partial sums of squares per row

// Tile geometries (template CFG): weight rows per wave x k per weight tile, register slots in flight per wave, k per activation panel.
//   0  16 x 256 (8 KB tiles), 2 slots, panel 256: 64-row workgroups, 51 KB of LDS at 16 activation rows
//   1  16 x 128 (4 KB tiles), 4 slots, panel 128: 64-row workgroups, 26 KB of LDS — more workgroups per CU, for grids of many small ones
//   2  32 x 128 (8 KB tiles), 2 slots, panel 256: 128-row workgroups — half the activation re-reads, for very tall products (lm_head)
template <int CFG> struct SkinnyCfg;
template <> struct SkinnyCfg<0> { static constexpr int NBW = 1, KT = 256, SLOTS = 2, KP = 256; };
template <> struct SkinnyCfg<1> { static constexpr int NBW = 1, KT = 128, SLOTS = 4, KP = 128; };
template <> struct SkinnyCfg<2> { static constexpr int NBW = 2, KT = 128, SLOTS = 2, KP = 256; };
__host__ __device__ constexpr int skinny_nbw(int cfg) { return cfg == 2 ? 2 : 1; }
__host__ __device__ constexpr int skinny_kt(int cfg) { return cfg == 0 ? 256 : 128; }
__host__ __device__ constexpr int skinny_kp(int cfg) { return cfg == 1 ? 128 : 256; }
__host__ __device__ constexpr int skinny_rows(int cfg) { return 64 * skinny_nbw(cfg); }            // weight rows per workgroup
__host__ __device__ constexpr size_t skinny_lds_bytes(int mb, int nt, int cfg) {
  return (size_t)(4 * 16 * skinny_nbw(cfg) * (skinny_kt(cfg) + 8) + nt * mb * 16 * (skinny_kp(cfg) + 8)) * 2;
}

// MB = 16-row blocks of activation rows (1: M <= 16, 2: M <= 32); NT = 16-bit terms per activation (2, or 3 for a product whose result is
// rounded to 16 bits again: the K/V rows of the QKV product); NBW = 16-row weight blocks per wave; ASRC: see above.
// Uses GemmArgs (prefill.h).
#ifndef TGX_SKINNY_DIS
#define TGX_SKINNY_DIS 0    // experiments only (tools/probes/skinny_probe.hip): 1 activation panel staged once, 2 no fragment reads / MFMAs, 8 no W refills, 16 no panel barriers
#endif
template <int DT, int EPI, int MB, int NT, int CFG, int ASRC>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const GemmArgs a) {
  constexpr int DIS = TGX_SKINNY_DIS;
  constexpr int NBW = SkinnyCfg<CFG>::NBW, KT = SkinnyCfg<CFG>::KT, SLOTS = SkinnyCfg<CFG>::SLOTS, SK_KP = SkinnyCfg<CFG>::KP;
  constexpr int SK_LDX = SK_KP + 8;                    // 16-bit elements per LDS row of an activation panel
  constexpr int LPT = 16 * NBW * KT * 2 / 1024;        // 16-byte loads per lane and weight tile (4 or 8)
  constexpr int LDW = KT + 8;
  constexpr int WR = 16 * NBW;                         // weight rows per wave
  constexpr int TPP = SK_KP / KT;                      // weight tiles per activation panel
  constexpr int XI = MB * 16 * (SK_KP / 8) / 256;      // 8-element activation chunks per thread and panel
  constexpr int LPR = KT / 8;                          // lanes per row of a wave-load (16 or 32)
  constexpr int RPL = 64 / LPR;                        // rows per wave-load (4 or 2)
  extern __shared__ __attribute__((aligned(16))) bf16_t sk_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  bf16_t* sW = sk_lds + wv * WR * LDW;
  bf16_t* sX = sk_lds + 4 * WR * LDW;                  // term t at sX + t * MB*16*SK_LDX
  const int n0 = blockIdx.x * (4 * WR) + wv * WR;      // first tile column of this wave
  const int k_begin = EPI == GEMM_PARTIAL ? (int)blockIdx.y * a.k_per : 0;
  const int k_end = EPI == GEMM_PARTIAL ? min(a.K, k_begin + a.k_per) : a.K;
  const bool inter = EPI == GEMM_SILU || (EPI == GEMM_PARTIAL && a.interleave);

  // weight tile loads: instruction i covers tile rows RPL*i .. RPL*i + RPL-1 (LPR lanes x 16 B contiguous per row)
  const int chunk = lane % LPR, lrow = lane / LPR;
  const bf16_t* wrow[LPT];
#pragma unroll
  for (int i = 0; i < LPT; i++) {
    const int nb = min(n0 + RPL * i + lrow, a.N - 1);   // clamped: rows past N reload the last one, masked at the store
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    wrow[i] = a.B + brow * a.K;
  }
  // Every load of the main loop is issued unconditionally from a clamped (always legal) address: a load under a branch makes the
  // compiler's vmcnt bookkeeping conservative, and the loop then drains ALL outstanding loads at every panel (first version: 3 TB/s).
  // A refill past the K range re-reads the range's last tile (cache-resident, never used); activations past it are staged as zeros.
  const int kt_last = k_begin + max(0, (k_end - k_begin - 1) / KT) * KT;
  auto load_w = [&](int kt, u32x4* r) {
    const int k = min(min(kt, kt_last) + 8 * chunk, a.K - 8);
#pragma unroll
    for (int i = 0; i < LPT; i++) r[i] = load_nt(reinterpret_cast<const u32x4*>(wrow[i] + k));
  };

  // activation panel: chunk c = tid + 256 i -> row c / 32, 8-element column c % 32 (the rows of a thread are the same in every panel)
  float inv_row[XI];
  if constexpr (ASRC == 2) {
#pragma unroll
    for (int i = 0; i < XI; i++) {
      const int row = min((tid + 256 * i) / (SK_KP / 8), a.M - 1);
      float ss = 0.f;
#pragma unroll
      for (int cb = 0; cb < SK_NCB; cb++) ss += a.ssq_part[(size_t)row * SK_NCB + cb];     // fixed order; ssq_ncb == SK_NCB
      inv_row[i] = 1.0f / sqrtf(ss / (float)a.K + a.eps);                                      // HF RMSNorm: w * (x * rsqrt(mean(x^2) + eps))
    }
  }
  u32x4 xr[ASRC == 0 ? NT : 2][XI];                    // ASRC 0: the stored terms; else the raw fp32 chunk (2 x 16 bytes)
  u32x4 xw[ASRC == 2 ? XI : 1];                        // norm weights of the chunk (16-bit)
  int x_kp = k_begin;                                  // panel the registers hold (store_x masks against it)
  auto load_x = [&](int kp) {
    x_kp = kp;
#pragma unroll
    for (int i = 0; i < XI; i++) {
      const int c = tid + 256 * i, row = min(c / (SK_KP / 8), a.M - 1), kc = c % (SK_KP / 8);
      const int k = min(kp + 8 * kc, a.K - 8);          // clamped; rows >= M and k >= k_end become zeros in store_x
      if constexpr (ASRC == 0) {
        const size_t off = (size_t)row * a.K + k;
        xr[0][i] = *reinterpret_cast<const u32x4*>(a.A_hi + off);
        xr[1][i] = *reinterpret_cast<const u32x4*>(a.A_lo + off);
        if (NT == 3) xr[NT - 1][i] = *reinterpret_cast<const u32x4*>(a.A_lo2 + off);
      } else {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.A_f32 + (size_t)row * a.lda + k);
        xr[0][i] = src[0];
        xr[1][i] = src[1];
        if constexpr (ASRC == 2) xw[i] = *reinterpret_cast<const u32x4*>(a.norm_w + k);
      }
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < XI; i++) {
      const int c = tid + 256 * i, row = c / (SK_KP / 8), kc = c - row * (SK_KP / 8);
      const bool ok = row < a.M && x_kp + 8 * kc < k_end;
      const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
      if constexpr (ASRC == 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) *reinterpret_cast<u32x4*>(&sX[(t * MB * 16 + row) * SK_LDX + kc * 8]) = ok ? xr[t][i] : zero;
      } else {
        u32x4 o[NT];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float y = __uint_as_float(e < 4 ? xr[0][i][e] : xr[1][i][e - 4]);
          if constexpr (ASRC == 2) {
            const unsigned int wp = xw[i][e >> 1];
            y = ((e & 1) ? pair_hi<DT>(wp) : pair_lo<DT>(wp)) * (y * inv_row[i]);
          }
          const bf16_t h = f32_to_elem<DT>(y);
          const float r1 = y - elem_to_f32<DT>(h);
          const bf16_t l = f32_to_elem<DT>(r1);
          bf16_t terms[3] = {h, l, 0};
          if (NT == 3) terms[2] = f32_to_elem<DT>(r1 - elem_to_f32<DT>(l));
#pragma unroll
          for (int t = 0; t < NT; t++) {
            if (e & 1) o[t][e >> 1] |= (unsigned int)terms[t] << 16; else o[t][e >> 1] = terms[t];
          }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) *reinterpret_cast<u32x4*>(&sX[(t * MB * 16 + row) * SK_LDX + kc * 8]) = ok ? o[t] : zero;
      }
    }
  };

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // SLOTS register slots of one weight tile each: 16 KB of weights in flight per wave
  u32x4 w[SLOTS][LPT];
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) load_w(k_begin + sl * KT, w[sl]);
  load_x(k_begin);

  // one weight tile: registers -> the wave's LDS tile, its register slot refilled SLOTS tiles ahead, then the tile's MFMAs;
  // xoff = offset of the tile's first k inside the staged activation panel
  auto tile = [&](u32x4* ws, int kt, int xoff) {
#pragma unroll
    for (int i = 0; i < LPT; i++) *reinterpret_cast<u32x4*>(&sW[(RPL * i + lrow) * LDW + chunk * 8]) = ws[i];
    if (!(DIS & 8)) load_w(kt + SLOTS * KT, ws);
#pragma unroll
    for (int ks = 0; ks < ((DIS & 2) ? 0 : KT / 32); ks++) {
      const int kcol = ks * 32 + 8 * (lane >> 4);
      bf16x8 fb[NBW], fa[MB][NT];
#pragma unroll
      for (int nb = 0; nb < NBW; nb++) fb[nb] = *reinterpret_cast<const bf16x8*>(&sW[(16 * nb + (lane & 15)) * LDW + kcol]);
#pragma unroll
      for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int t = 0; t < NT; t++)
          fa[mb][t] = *reinterpret_cast<const bf16x8*>(&sX[(t * MB * 16 + 16 * mb + (lane & 15)) * SK_LDX + xoff + kcol]);
#pragma unroll
      for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) {
#pragma unroll
          for (int t = NT - 1; t >= 0; t--) acc[mb][nb] = mfma16x16<DT>(fa[mb][t], fb[nb], acc[mb][nb]);     // small terms first
        }
    }
  };

  // The register slot of every tile is a compile-time constant: the loop body covers SLOTS / TPP panels and leaves through `break`
  // (with a run-time slot index the compiler's vmcnt bookkeeping merges "slot 0 is the youngest load" with "slot 1 is", and every tile
  // then waits for ALL outstanding loads — the prefetch depth collapses to one tile: measured 3.0 vs 5 TB/s).
  auto panel_head = [&](int kp) {
    if ((DIS & 1) && kp != k_begin) return;
    if (!(DIS & 16)) __syncthreads();                 // every wave is done with the previous panel
    store_x();
    if (!(DIS & 16)) __syncthreads();
    load_x(kp + SK_KP);
  };
  if (k_begin < k_end) {
    for (int kp = k_begin;;) {
      if constexpr (TPP == 2) {        // SLOTS == 2: the panel's two tiles are the two slots
        panel_head(kp);
        tile(w[0], kp, 0);
        if (kp + KT < k_end) tile(w[1], kp + KT, KT);
        kp += SK_KP; if (kp >= k_end) break;
      } else {
        panel_head(kp); tile(w[0], kp, 0); kp += SK_KP; if (kp >= k_end) break;
        panel_head(kp); tile(w[1], kp, 0); kp += SK_KP; if (kp >= k_end) break;
        if constexpr (SLOTS == 4) {
          panel_head(kp); tile(w[2], kp, 0); kp += SK_KP; if (kp >= k_end) break;
          panel_head(kp); tile(w[3], kp, 0); kp += SK_KP; if (kp >= k_end) break;
        }
      }
    }
  }

  // D: col n = lane&15, row m = 4*(lane>>4) + r
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
      const int col = n0 + 16 * nb + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * mb + 4 * (lane >> 4) + r;
        const float v = acc[mb][nb][r];
        if (EPI == GEMM_SILU) {      // even lanes hold gate_i, odd lanes up_i (i = col / 2): the pair meets over the DPP crossbar
          const float other = dpp_mov<0xB1, 0xf>(v);
          if ((lane & 1) || col >= a.N || row >= a.M) continue;
          const size_t o = (size_t)row * a.inter + (size_t)(col >> 1);
          split16<DT>((v / (1.0f + expf(-v))) * other, a.out_hi[o], a.out_lo[o]);
          continue;
        }
        if (col >= a.N || row >= a.M) continue;
        if (EPI == GEMM_PARTIAL) { a.part[((size_t)blockIdx.y * a.M + row) * a.N + col] = v; continue; }
        float* dst = a.C + (size_t)row * a.ldc + col;
        const float o = v + (a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f);
        *dst = (EPI == GEMM_RESIDUAL) ? (*dst + o) : o;
      }
    }
}

// ---- finishing a split-K product row-wise: C[m][:] (+)= sum_z part[z][m][:] (+ bias), plus the row's partial sums of squares for the
// RMSNorm the next product applies while staging (ASRC 2).  grid = (M, SK_NCB): workgroup (m, cb) owns a contiguous quarter of the columns.
template <int DT, int EPI>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const GemmArgs a) {
  __shared__ float sc[4];
  const int m = blockIdx.x, per = ((a.N + (int)gridDim.y * 4 - 1) / ((int)gridDim.y * 4)) * 4;
  const int c0 = blockIdx.y * per, c1 = min(a.N, c0 + per);
  const size_t slab = (size_t)a.M * a.N;
  float ss = 0.f;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    const float* p = a.part + (size_t)m * a.N + c;
    float t[16];
#pragma unroll
    for (int z = 0; z < 16; z++) t[z] = z < a.nsplit ? p[z * slab] : 0.f;      // all slabs in flight together; summed in z order
    float v = t[0];
#pragma unroll
    for (int z = 1; z < 16; z++) v += t[z];
    for (int z = 16; z < a.nsplit; z++) v += p[z * slab];
    if (a.bias) v += elem_to_f32<DT>(a.bias[c]);
    float* dst = a.C + (size_t)m * a.ldc + c;
    if (EPI == GEMM_RESIDUAL) v += *dst;
    *dst = v;
    ss = fmaf(v, v, ss);
  }
  ss = block_sum_256(ss, sc);
  if (threadIdx.x == 0 && a.ssq_out) a.ssq_out[(size_t)m * gridDim.y + blockIdx.y] = ss;
}

// partial sums of squares of fp32 rows (the embedding rows a decode step starts from): same (M, SK_NCB) layout
static __global__ __launch_bounds__(256) void row_ssq_kernel(const float* X, long long ldx, int n, float* ssq_out) {
  __shared__ float sc[4];
  const int m = blockIdx.x, per = ((n + (int)gridDim.y * 4 - 1) / ((int)gridDim.y * 4)) * 4;
  const int c0 = blockIdx.y * per, c1 = min(n, c0 + per);
  float ss = 0.f;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) { const float v = X[(size_t)m * ldx + c]; ss = fmaf(v, v, ss); }
  ss = block_sum_256(ss, sc);
  if (threadIdx.x == 0) ssq_out[(size_t)m * gridDim.y + blockIdx.y] = ss;
}

// ---- row-wise glue of a batched decode step (rows = independent sequences, each at its own position with its own cache) -------------
// [sum of the QKV product's split-K slabs + bias] -> split -> [Qwen3 q/k RMSNorm] -> RoPE(q), RoPE(k) at pos[row] ->
// KVCacheManager::append into THAT row's cache; q stays fp32 for the decode attention kernel (Attention.h:94-106, AttentionWithQKNorm :156-163)
struct RopeRowsArgs {
  const float* QKV;        // [rows][qd + 2*kvd] fp32 (bias already added), or nullptr: take the sums of `part`
  const float* part;       // [nsplit][rows][qd + 2*kvd] split-K slabs of the QKV product
  int nsplit, rows;
  const void* bias;        // [qd + 2*kvd] storage dtype or nullptr (only with part)
  float* q_out;            // [rows][q_stride]
  void *k_cache, *v_cache; // this layer, row 0: [kv_heads][max_ctx][hd]; rows kv_stride elements apart
  const float *rope_cos, *rope_sin;
  const int* pos;          // [rows]
  int heads, kv_heads, hd, max_ctx;
  long long q_stride, kv_stride;
  const void *q_norm_w, *k_norm_w;
  float eps;
  const int* blk_tbl;      // paged KV (common.h kv_paged_off): [rows][tbl_stride] block tables, k_cache / v_cache = the layer's pools (kv_stride 0) — or nullptr
  long long tbl_stride;
};
template <int DT>
__global__ __launch_bounds__(64) void rope_kv_rows_kernel(const RopeRowsArgs a) {      // grid (rows, heads + 2*kv_heads): one wave per head vector
  typedef elem_t<DT> E;
  const int r = blockIdx.x, hh = blockIdx.y, pos = a.pos[r], half = a.hd >> 1;
  const int qd = a.heads * a.hd, kvd = a.kv_heads * a.hd, N = qd + 2 * kvd;
  const size_t slab = (size_t)a.rows * N;
  const int p = threadIdx.x;
  const bool act = p < half;
  const int pc = act ? p : 0;
  auto value = [&](int idx) -> float {
    if (a.QKV) return a.QKV[(size_t)r * N + idx];
    const float* src = a.part + (size_t)r * N + idx;
    float t[16];
#pragma unroll
    for (int z = 0; z < 16; z++) t[z] = z < a.nsplit ? src[z * slab] : 0.f;    // all slabs in flight together; summed in z order
    float v = t[0];
#pragma unroll
    for (int z = 1; z < 16; z++) v += t[z];
    for (int z = 16; z < a.nsplit; z++) v += src[z * slab];
    if (a.bias) v += elem_to_f32<DT>(static_cast<const E*>(a.bias)[idx]);
    return v;
  };
  float x0 = value(hh * a.hd + pc), x1 = value(hh * a.hd + pc + half);
  if (!act) { x0 = 0.f; x1 = 0.f; }
  if (a.q_norm_w != nullptr && hh < a.heads + a.kv_heads) {     // per-head RMSNorm over head_dim (lanes beyond hd/2 hold zeros)
    const float ss = wave_sum(head_sq_pair(x0, x1));
    const float inv = head_rms_inv(ss, a.hd, a.eps);
    const E* w = static_cast<const E*>(hh < a.heads ? a.q_norm_w : a.k_norm_w);
    x0 = elem_to_f32<DT>(w[pc]) * (x0 * inv);
    x1 = elem_to_f32<DT>(w[pc + half]) * (x1 * inv);
  }
  if (!act) return;
  if (hh < a.heads + a.kv_heads) {
    const float cs = a.rope_cos[(size_t)pos * half + p], sn = a.rope_sin[(size_t)pos * half + p];
    rope_rotate_pair(x0, x1, cs, sn);
  }
  if (hh < a.heads) {
    float* q = a.q_out + (size_t)r * a.q_stride + hh * a.hd;
    q[p] = x0; q[p + half] = x1;
  } else {
    E* base = static_cast<E*>(hh < a.heads + a.kv_heads ? a.k_cache : a.v_cache) + (size_t)r * a.kv_stride;
    const int kh = hh < a.heads + a.kv_heads ? hh - a.heads : hh - a.heads - a.kv_heads;
    E* dst = base + (a.blk_tbl ? kv_paged_off(a.blk_tbl + (size_t)r * a.tbl_stride, a.kv_heads, kh, pos, a.hd) : ((size_t)kh * a.max_ctx + pos) * a.hd);
    dst[p] = f32_to_elem<DT>(x0);
    dst[p + half] = f32_to_elem<DT>(x1);
  }
}

// per-workgroup argmax partials of every row's logits (what the GEMV lm_head's epilogue leaves for the greedy finalize / the sampler).  The row's `n_part`
// slots (the GEMV lm_head's grid: ~2000) are filled by gridDim.x workgroups — as many as have a whole 16-byte load per thread to do (~125 at V = 128k; one
// workgroup per slot was 64k nearly idle workgroups at 32 rows: 26 us) — slots from gridDim.x on are set to the reduction's identity.
static __global__ __launch_bounds__(256) void argmax_partials_rows_kernel(const float* logits, long long logits_stride, int V, float* part_val, int* part_idx, long long part_stride, int n_part) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* lg = logits + (size_t)blockIdx.y * logits_stride;
  float bv = -INFINITY; int bi = 0x7fffffff;
  if ((logits_stride & 3) == 0 && (V & 3) == 0) {       // 16-byte loads (64 rows x 128k logits: 52 -> 12 us)
    const f32x4* lg4 = reinterpret_cast<const f32x4*>(lg);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (V >> 2); i += gridDim.x * 256) {
      const f32x4 v4 = lg4[i];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float v = v4[j];
        if (v > bv || (v == bv && 4 * i + j < bi)) { bv = v; bi = 4 * i + j; }
      }
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
      const float v = lg[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s]; const int ix = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = ix; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_val[(size_t)blockIdx.y * part_stride + blockIdx.x] = sv[0]; part_idx[(size_t)blockIdx.y * part_stride + blockIdx.x] = si[0]; }
  for (int p = (int)gridDim.x + (int)blockIdx.x * 256 + (int)threadIdx.x; p < n_part; p += (int)gridDim.x * 256) {
    part_val[(size_t)blockIdx.y * part_stride + p] = -INFINITY; part_idx[(size_t)blockIdx.y * part_stride + p] = 0x7fffffff;
  }
}

}  // namespace tgx
