// gemm_f32.h — the batched prefill for fp32 storage (--dtype fp32: the reference's own runnable configuration is GPT-2 fp32,
// examples/inference/main.cpp:82-88, BASELINE.json configs[0]) and the fp32 row-wise glue around it.
//
//   C[M][N] (+)= A[M][K] · B[N][K]ᵀ with fp32 A, B, C on v_mfma_f32_32x32x2_f32: exact fp32 products with fp32 accumulation —
//   the arithmetic of the decode path's fp32 FMA chains in another summation order (no operand splitting: the weights ARE fp32).
//
// Roofline: MFMA at the f32-input rate (157 TFLOP/s = the fp32 vector peak, 1/16 of bf16): flops per launch = 2*M*N*K.
// Tile: 64*MI x 128 x 32 per workgroup (4 waves as 2 x 2, each MI x 2 MFMA tiles of 32 x 32), operands staged global -> registers
// -> LDS k-major ([k][row], row stride TM+1 / 129 floats: the A / B fragment of one MFMA — lane = row, k = lane >> 5 — is 32
// consecutive floats per half-wave, conflict-free), the next K step's global loads in flight during the MFMAs.
#pragma once
#include "prefill.h"

namespace tgx {

enum { F32_STORE = 0, F32_RESIDUAL = 1, F32_SILU = 2, F32_GELU = 3 };
struct GemmF32Args {
  const float* A;          // [M][K]
  const float* B;          // [N][K] (torch Linear weight, fp32 storage)
  const float* bias;       // [N] or nullptr
  float* C;                // [M][ldc]   (F32_SILU: [M][inter] — tile columns alternate gate row i / up row i; F32_GELU: gelu_new(acc + bias))
  int M, N, K, ldc;
  int inter;
  // split-K (a prompt that gives the product fewer tiles than two per CU): blockIdx.z covers k_per of K and stores its raw fp32 tile to
  // part[z][M][N]; gemm_f32_reduce_kernel sums the slabs in z order and applies the epilogue.  nsplit <= 1: the epilogue runs here.
  float* part;
  int k_per, nsplit;
};

constexpr int FBN = 128, FBK = 32;

__device__ __forceinline__ float gelu_new_f32(float x) {   // HF "gelu_new" (GPT-2): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

template <int EPI, int MI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args a) {
  constexpr int TM = 64 * MI;
  constexpr int LDA = TM + 1, LDB = FBN + 1;
  constexpr int AI = TM * (FBK / 4) / 256;          // 16-byte A chunks per thread and K step (2 or 4)
  constexpr int BI = FBN * (FBK / 4) / 256;         // 4
  __shared__ float sA[FBK * LDA];
  __shared__ float sB[FBK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * FBN;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // staging map: chunk c = tid + 256 i -> tile row c / 8, 4-float column c % 8
  f32x4 ra[AI], rb[BI];
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      const int m = min(m0 + row, a.M - 1), k = min(k0 + 4 * kc, a.K - 4);      // clamped (always legal); masked at the LDS store
      ra[i] = *reinterpret_cast<const f32x4*>(a.A + (size_t)m * a.K + k);
    }
#pragma unroll
    for (int i = 0; i < BI; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      const int nb = min(n0 + row, a.N - 1), k = min(k0 + 4 * kc, a.K - 4);
      const size_t brow = EPI == F32_SILU ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
      rb[i] = *reinterpret_cast<const f32x4*>(a.B + brow * a.K + k);
    }
  };
  auto store_tiles = [&](int k0, int k_end) {
#pragma unroll
    for (int i = 0; i < AI; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      const f32x4 v = (m0 + row < a.M && k0 + 4 * kc < k_end) ? ra[i] : zero;
#pragma unroll
      for (int t = 0; t < 4; t++) sA[(4 * kc + t) * LDA + row] = v[t];
    }
#pragma unroll
    for (int i = 0; i < BI; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      const f32x4 v = (n0 + row < a.N && k0 + 4 * kc < k_end) ? rb[i] : zero;
#pragma unroll
      for (int t = 0; t < 4; t++) sB[(4 * kc + t) * LDB + row] = v[t];
    }
  };

  const int k_begin = a.nsplit > 1 ? (int)blockIdx.z * a.k_per : 0;
  const int k_end = a.nsplit > 1 ? min(a.K, k_begin + a.k_per) : a.K;
  load_tiles(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += FBK) {
    __syncthreads();                 // everyone is done reading the previous tile
    store_tiles(k0, k_end);
    __syncthreads();
    load_tiles(k0 + FBK);            // next K step's global loads fly under the MFMAs (clamped past the end)
#pragma unroll
    for (int kk = 0; kk < FBK / 2; kk++) {
      const int krow = 2 * kk + (lane >> 5);
      float fa[MI], fb[2];
#pragma unroll
      for (int i = 0; i < MI; i++) fa[i] = sA[krow * LDA + wm * (32 * MI) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 2; j++) fb[j] = sB[krow * LDB + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[i][j][r];
        if (a.nsplit > 1) {          // raw slab (tile-column order; the reducer pairs gate / up columns itself)
          if (col < a.N && row < a.M) a.part[((size_t)blockIdx.z * a.M + row) * a.N + col] = v;
          continue;
        }
        if (EPI == F32_SILU) {       // even lanes hold gate_i, odd lanes up_i (i = col / 2): the pair meets over the DPP crossbar
          const float other = dpp_mov<0xB1, 0xf>(v);
          if ((lane & 1) || col >= a.N || row >= a.M) continue;
          a.C[(size_t)row * a.inter + (size_t)(col >> 1)] = (v / (1.0f + expf(-v))) * other;
          continue;
        }
        if (col >= a.N || row >= a.M) continue;
        float* dst = a.C + (size_t)row * a.ldc + col;
        const float o = v + (a.bias ? a.bias[col] : 0.f);
        if (EPI == F32_GELU) *dst = gelu_new_f32(o);
        else *dst = (EPI == F32_RESIDUAL) ? (*dst + o) : o;
      }
    }
}

// Sums the split-K slabs in z order and applies the epilogue the unsplit kernel would have applied.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const GemmF32Args a) {
  const int ncol = EPI == F32_SILU ? a.N / 2 : a.N;               // SILU: one thread per (gate, up) pair
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.M * ncol) return;
  const int row = (int)(idx / ncol), c = (int)(idx - (size_t)row * ncol);
  const size_t slab = (size_t)a.M * a.N;
  if (EPI == F32_SILU) {
    const float* p = a.part + (size_t)row * a.N + 2 * c;
    float g = 0.f, u = 0.f;
    for (int z = 0; z < a.nsplit; z++) { g += p[z * slab]; u += p[z * slab + 1]; }
    a.C[(size_t)row * a.inter + c] = (g / (1.0f + expf(-g))) * u;
    return;
  }
  const float* p = a.part + (size_t)row * a.N + c;
  float v = 0.f;
  for (int z = 0; z < a.nsplit; z++) v += p[z * slab];
  if (a.bias) v += a.bias[c];
  float* dst = a.C + (size_t)row * a.ldc + c;
  if (EPI == F32_GELU) *dst = gelu_new_f32(v);
  else *dst = (EPI == F32_RESIDUAL) ? (*dst + v) : v;
}

// ---- causal GQA flash attention in fp32 over the fp32 cache, queries past .. past+S-1 (the fp32 counterpart of attn_prefill_kernel,
// prefill.h: same decomposition, same register-resident online softmax).  One workgroup = 128 queries of one head (4 waves x 32);
// K / V tiles of 64 keys go global -> registers -> LDS and are shared by the four waves.  S^T = K . Q^T and O^T += V^T . P^T run on
// v_mfma_f32_32x32x2_f32: every operand element is ONE float per lane, so K is read [key][d] (row stride HD + 1: the 32 keys of a
// fragment fall on 32 banks), V straight from its [key][d] tile (lane = d), Q lives in registers (lane = query, d parity = lane >> 5),
// and the probabilities are already the B operand of the next MFMA: register r of a score sub-tile holds key (r&3) + 8 (r>>2) for
// lanes 0-31 and that key + 4 for lanes 32-63 — exactly the two k slots of one 32x32x2 step.
struct AttnPrefillF32Args {
  const float* q;                // [S][heads*hd] rotated queries
  const float *k_cache, *v_cache;   // [kv_heads][max_ctx][hd]
  float* out;                    // [S][heads*hd]
  int S, heads, kv_heads, max_ctx, past;
  float scale;
  int qblk_mirror;
};
template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_f32_kernel(const AttnPrefillF32Args a) {
  constexpr int LK = HD + 1;                  // K tile row stride (floats)
  constexpr int LV = HD;                      // V tile row stride
  constexpr int NB = HD / 32;                 // 32-row output-dim blocks
  constexpr int CH = HD / 4;                  // 16-byte chunks per head row
  constexpr int KT = HD == 64 ? 64 : 32;       // keys per tile (LDS: 2 x KT x HD floats = 33 KB)
  constexpr int NSUB = KT / 32;
  constexpr int NCH = KT * CH / 256;          // chunks per thread and tile
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) float smem[KT * LV + KT * LK];     // V tile | K tile; the output rows pass through it at the end
  float* const sV = smem;
  float* const sK = smem + KT * LV;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, G = a.heads / a.kv_heads, kvh = h / G;
  const int qd = a.heads * HD;
  const int qblk = (a.qblk_mirror && h >= a.heads / 2) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int q0 = qblk * 128 + wv * 32;
  const int qi = q0 + ql;
  const bool qvalid = qi < a.S;
  const int qpos = a.past + qi;

  float qreg[HD / 2];                         // q[query][2 i + hh]
#pragma unroll
  for (int i = 0; i < HD / 2; i++) qreg[i] = qvalid ? a.q[(size_t)qi * qd + (size_t)h * HD + 2 * i + hh] : 0.f;
  f32x16 oacc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[b][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = a.k_cache + (size_t)kvh * a.max_ctx * HD;
  const float* vbase = a.v_cache + (size_t)kvh * a.max_ctx * HD;
  const int wg_last_pos = a.past + min(qblk * 128 + 127, a.S - 1);
  const int n_kt = wg_last_pos / KT + 1;
  const int wave_last_pos = a.past + min(q0 + 31, a.S - 1);
  const bool wave_live = q0 < a.S;
  const float qs = a.scale * LOG2E;

  f32x4 kvr[NCH], vvr[NCH];
  auto fetch_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = tid + 256 * i, row = c / CH, kc = c - row * CH;
      const int key = min(kt * KT + row, wg_last_pos);          // clamped; keys past the workgroup's last position are masked below
      kvr[i] = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * HD + kc * 4);
      vvr[i] = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * HD + kc * 4);
    }
  };
  fetch_tile(0);
  for (int kt = 0; kt < n_kt; kt++) {
    const int key0 = kt * KT;
    __syncthreads();                           // the previous tile is consumed by every wave
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = tid + 256 * i, row = c / CH, kc = c - row * CH;
#pragma unroll
      for (int t = 0; t < 4; t++) sK[row * LK + kc * 4 + t] = kvr[i][t];
      *reinterpret_cast<f32x4*>(&sV[row * LV + kc * 4]) = vvr[i];
    }
    __syncthreads();
    if (kt + 1 < n_kt) fetch_tile(kt + 1);     // in flight during this tile's MFMAs
    if (!wave_live || key0 > wave_last_pos) continue;    // wave-uniform: nothing of this tile is visible to the wave's queries

    f32x16 sacc[NSUB];
    float mx = -INFINITY;
    const bool diag = key0 + KT - 1 > a.past + q0;
#pragma unroll
    for (int sub = 0; sub < NSUB; sub++) {
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[sub][r] = 0.f;
      const int kb = key0 + 32 * sub;
      if (kb <= wave_last_pos) {
#pragma unroll
        for (int i = 0; i < HD / 2; i++)
          sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(32 * sub + ql) * LK + 2 * i + hh], qreg[i], sacc[sub], 0, 0, 0);
      }
      if (diag || !qvalid || kb > wave_last_pos) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (!(qvalid && key <= qpos)) sacc[sub][r] = -INFINITY;     // isCausal (Attention.h:108) with the cache offset
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[sub][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
    const float m_new = fmaxf(m_run, mx);
    const bool dead = m_new == -INFINITY;
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
    const float neg_m = dead ? 0.f : -m_new;
    float sum = 0.f;
#pragma unroll
    for (int sub = 0; sub < NSUB; sub++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[sub][r], qs, neg_m));
        sacc[sub][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[b][r] *= alpha;

    // O^T += V^T . P^T: step r of a sub-tile contracts keys (r&3) + 8 (r>>2) [lanes 0-31] and + 4 [lanes 32-63]
#pragma unroll
    for (int sub = 0; sub < NSUB; sub++) {
      if (key0 + 32 * sub > wave_last_pos) continue;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kloc = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
        for (int b = 0; b < NB; b++)
          oacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[kloc * LV + 32 * b + ql], sacc[sub][r], oacc[b], 0, 0, 0);
      }
    }
  }

  // A lane holds one query's output dims: direct stores would be 4-byte writes a whole row apart (64 cache lines per instruction).  Each wave
  // passes its 32 x 32 blocks through LDS (row stride 33 floats: conflict-free column writes) and stores 128-byte row segments.
  __syncthreads();                              // every wave is done with the last K / V tile
  static_assert(4 * 32 * 33 <= KT * LV + KT * LK, "output staging exceeds the K / V tiles");
  float* const wblk = smem + wv * 32 * 33;
  const float inv_l = qvalid ? 1.0f / l_run : 0.f;
#pragma unroll
  for (int b = 0; b < NB; b++) {
#pragma unroll
    for (int r = 0; r < 16; r++) wblk[ql * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = oacc[b][r] * inv_l;
#pragma unroll
    for (int i = 0; i < 16; i++) {               // 32 rows x 32 floats over 64 lanes: lane -> (row = 2 i + hh, column ql)
      const int row = 2 * i + hh;
      if (q0 + row < a.S) a.out[(size_t)(q0 + row) * qd + (size_t)h * HD + 32 * b + ql] = wblk[row * 33 + ql];
    }
  }
}

// ---- row-wise glue ----------------------------------------------------------------------------------------------------------------
// X[s][:] = embed[ids[s]] (+ wpe[past + s] for GPT-2, ModelGPT2.h:165-169) as fp32, any storage dtype
template <int DT>
__global__ __launch_bounds__(256) void embed_rows_any_kernel(const long long* ids, const void* embed, const void* wpe, float* X, int H, int S, long long ids_stride, int past) {
  const int b = blockIdx.x / S, s = blockIdx.x % S;
  const long long t = ids[(size_t)b * ids_stride + s];
  const elem_t<DT>* row = static_cast<const elem_t<DT>*>(embed) + (size_t)t * H;
  const elem_t<DT>* prow = static_cast<const elem_t<DT>*>(wpe) + (size_t)(past + s) * H;
  f32x4* dst = reinterpret_cast<f32x4*>(X + (size_t)blockIdx.x * H);
  for (int c = threadIdx.x; c < (H >> 3); c += 256) {
    float f[8];
    slice_unpack<DT>(load_slice<DT>(row, c), f);
    if (wpe) {
      float g[8];
      slice_unpack<DT>(load_slice<DT>(prow, c), g);
#pragma unroll
      for (int k = 0; k < 8; k++) f[k] += g[k];
    }
    dst[2 * c] = f32x4{f[0], f[1], f[2], f[3]};
    dst[2 * c + 1] = f32x4{f[4], f[5], f[6], f[7]};
  }
}

// Row-wise RMSNorm (LN = 0; HF order w * (x * rsqrt(mean(x^2) + eps))) or torch LayerNorm (LN = 1: biased variance, two passes,
// ((x - mean) * rsqrt(var + eps)) * w + b — the formulas of the decode kernels' prologues, gemv.h).  Output: fp32 rows (OUT16 = 0, the
// fp32 GEMM's A operand) or the hi / lo (/ third) 16-bit terms of the bf16 / fp16 GEMM (OUT16 = 1).
template <int DT, int LN, int OUT16>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* X, const void* w_, const void* b_, float eps, int H, float* out, bf16_t* hi, bf16_t* lo, bf16_t* lo2) {
  __shared__ float sc[4];
  const elem_t<DT>* w = static_cast<const elem_t<DT>*>(w_);
  const elem_t<DT>* bias = static_cast<const elem_t<DT>*>(b_);
  const float* x = X + (size_t)blockIdx.x * H;
  float mean = 0.f, inv;
  if (LN) {
    float sm = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) sm += x[i];
    mean = block_sum_256(sm, sc) / (float)H;
    float sq = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) { const float dl = x[i] - mean; sq = fmaf(dl, dl, sq); }
    inv = 1.0f / sqrtf(block_sum_256(sq, sc) / (float)H + eps);
  } else {
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) ss = fmaf(x[i], x[i], ss);
    inv = 1.0f / sqrtf(block_sum_256(ss, sc) / (float)H + eps);
  }
  for (int i = threadIdx.x; i < H; i += 256) {
    float y;
    if (LN) y = __fadd_rn(__fmul_rn(__fmul_rn(x[i] - mean, inv), elem_to_f32<DT>(w[i])), elem_to_f32<DT>(bias[i]));
    else y = elem_to_f32<DT>(w[i]) * (x[i] * inv);
    const size_t o = (size_t)blockIdx.x * H + i;
    if (OUT16) {
      if constexpr (DT != DT_F32) {
        split16<DT>(y, hi[o], lo[o]);
        if (lo2) lo2[o] = f32_to_elem<DT>(y - elem_to_f32<DT>(hi[o]) - elem_to_f32<DT>(lo[o]));
      }
    } else out[o] = y;
  }
}

static __global__ void iota_pos_kernel(int* pos, int first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] = first + i;
}

}  // namespace tgx
