// prefill.h — batched prefill (seq > 1): the S×H·Wᵀ products on the matrix cores, causal GQA flash attention,
// and the row-wise glue between them.  Same math as S single-position passes, different schedule:
//
//   reference op (per layer, [B,S,*] tensors)                 here
//   nn::Embedding                     GPTModel.h:52            embed_rows_kernel
//   RMSNorm                           DecoderLayer.h:40-41     rmsnorm_split_kernel      (fp32 -> bf16 hi/lo pair)
//   MergedLinear qkv / o_proj /       Attention.h:95,90        gemm_bf16x2_kernel        (MFMA 32x32x16 bf16)
//     gate_up / down_proj             GatedMLP.h:38,40
//   split + RoPE(q,k) + cache append  Attention.h:96-106       rope_kv_split_kernel
//   flashAttention(causal)            Attention.h:108-109      attn_prefill_kernel       (MFMA QKᵀ and PV, online softmax)
//   siluMul                           Activation.h:16          fused into the gate_up product's epilogue (GEMM_SILU)
//
// Precision: activations are fp32 between ops, the matrix cores take bf16.  Every fp32 MFMA operand x is split as
// x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) and multiplied in two MFMAs against the exact bf16 weights / K / V,
// so products carry ~16 mantissa bits and accumulate in fp32: the batched path agrees with the step path to ~1e-5.
//
// Roofline: MFMA (bf16 dense peak ~2.5 PFLOP/s); flops per GEMM launch = 2 (hi,lo) * 2*M*N*K.
#pragma once
#include "common.h"

namespace tgx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// one 32x32x16 matrix-core product on 16-bit operands of the storage dtype (same register image for both dtypes)
template <int DT>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (DT == DT_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// x = hi + lo with hi = round16(x), lo = round16(x - hi) in the 16-bit storage dtype DT (bf16: 16 significant bits in two
// terms; fp16: 22, with an absolute floor of 2^-25 where lo goes subnormal)
template <int DT>
__device__ __forceinline__ void split16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_elem<DT>(x);
  lo = f32_to_elem<DT>(x - elem_to_f32<DT>(hi));
}

// silu(g) * u and gelu_new(v) = v * sigmoid(2 y) with the hardware exp2 / rcp (1 ulp each, ~2e-7 relative).  The MFMA prefill epilogues run them
// for every output element — with expf + an IEEE division (~35 instructions) the epilogue of a 256 x 256 tile issued as many instructions as
// its whole K loop; the value is rounded to two 16-bit terms (2^-17) right after.
__device__ __forceinline__ float silu_mul_fast(float g, float u) {
  return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g)) * u;
}
__device__ __forceinline__ float gelu_new_fast(float v) {
  const float y = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * y));     // 0.5 v (1 + tanh y) = v / (1 + e^(-2y))
}

// ---- X[s][:] = embed[ids[s]] (fp32) ------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void embed_rows_kernel(const long long* ids, const bf16_t* embed, float* X, int H, int S, long long ids_stride) {
  // stacked prompts: workspace row blockIdx.x = (batch row, position); the ids of batch row b start at ids + b * ids_stride
  const long long t = ids[(size_t)(blockIdx.x / S) * ids_stride + blockIdx.x % S];
  const u32x4* src = reinterpret_cast<const u32x4*>(embed + (size_t)t * H);
  f32x4* dst = reinterpret_cast<f32x4*>(X + (size_t)blockIdx.x * H);
  for (int c = threadIdx.x; c < (H >> 3); c += 256) {
    const u32x4 v = src[c];
    dst[2 * c] = f32x4{pair_lo<DT>(v[0]), pair_hi<DT>(v[0]), pair_lo<DT>(v[1]), pair_hi<DT>(v[1])};
    dst[2 * c + 1] = f32x4{pair_lo<DT>(v[2]), pair_hi<DT>(v[2]), pair_lo<DT>(v[3]), pair_hi<DT>(v[3])};
  }
}

// ---- row-wise RMSNorm, output as bf16 hi/lo ---------------------------------------------------------------------
// `lo2` (optional) receives a third term: x = hi + lo + lo2 carries ~24 mantissa bits.  The QKV projection uses it,
// because its output is the only one that is rounded to bf16 again (the KV cache): with two terms (2^-18) a few
// per cent of the cache entries round differently from the single-position path, with three the schedules agree.
// `part` (optional): the row first takes a pending split-K residual — x += sum_z part[z][row][:] (+ bias), slabs `slab` floats apart, summed in z
// order exactly as gemm_splitk_reduce_kernel<GEMM_RESIDUAL> does — and is written back: the reduce launch and its pass over x disappear.
template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_split_kernel(float* X, const bf16_t* w, float eps, int H, bf16_t* hi, bf16_t* lo, bf16_t* lo2,
                                                            const float* part = nullptr, int nsplit = 0, long long slab = 0, const bf16_t* bias = nullptr, int inter = 0) {
  // inter = 1 (round 5): the two terms leave INTERLEAVED per k32 block — hi[m][H / 32][hi 32 | lo 32], one 128-byte line per row and k32 step — for the consumer
  // that stages full lines (kernels/gemm_dma.h gemm_dma8i_kernel); `lo` is not written
  // one workgroup per row; a thread owns 8-element slices (two 16-byte loads, one 16-byte store per term), kept in registers between the
  // sum of squares and the scaling for rows up to 8192 elements (H % 8 == 0: checked by the caller)
  __shared__ float sc[4];
  constexpr int NV = 4;
  float* x = X + (size_t)blockIdx.x * H;
  const int nch = H >> 3;
  auto fetch = [&](int c, f32x4& v0, f32x4& v1) {
    v0 = *reinterpret_cast<const f32x4*>(x + 8 * c); v1 = *reinterpret_cast<const f32x4*>(x + 8 * c + 4);
    if (part) {
      const float* p = part + (size_t)blockIdx.x * H + 8 * c;
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
      int z = 0;
      for (; z + 8 <= nsplit; z += 8) {          // eight slabs in flight together (a batched step has one workgroup per row: the loop was a chain of L2 round trips); summed in z order
        f32x4 a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a0[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(z + u) * slab); a1[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(z + u) * slab + 4); }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
          for (int t = 0; t < 4; t++) { s0[t] += a0[u][t]; s1[t] += a1[u][t]; }
      }
      for (; z < nsplit; z++) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + (size_t)z * slab), a1 = *reinterpret_cast<const f32x4*>(p + (size_t)z * slab + 4);
#pragma unroll
        for (int t = 0; t < 4; t++) { s0[t] += a0[t]; s1[t] += a1[t]; }
      }
      if (bias) {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + 8 * c);
#pragma unroll
        for (int t = 0; t < 2; t++) { s0[2 * t] += pair_lo<DT>(bv[t]); s0[2 * t + 1] += pair_hi<DT>(bv[t]); s1[2 * t] += pair_lo<DT>(bv[2 + t]); s1[2 * t + 1] += pair_hi<DT>(bv[2 + t]); }
      }
#pragma unroll
      for (int t = 0; t < 4; t++) { v0[t] += s0[t]; v1[t] += s1[t]; }
      *reinterpret_cast<f32x4*>(x + 8 * c) = v0; *reinterpret_cast<f32x4*>(x + 8 * c + 4) = v1;
    }
  };
  f32x4 xr[NV][2];
  u32x4 wr[NV];                       // the norm weights of the thread's chunks: loaded with the row, not after the workgroup's sum (one L2 round trip off the chain)
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = threadIdx.x + 256 * j;
    xr[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; xr[j][1] = xr[j][0];
    wr[j] = u32x4{0u, 0u, 0u, 0u};
    if (c < nch) wr[j] = *reinterpret_cast<const u32x4*>(w + 8 * c);
    if (c < nch) fetch(c, xr[j][0], xr[j][1]);
#pragma unroll
    for (int t = 0; t < 4; t++) { ss = fmaf(xr[j][0][t], xr[j][0][t], ss); ss = fmaf(xr[j][1][t], xr[j][1][t], ss); }
  }
  for (int c = threadIdx.x + 256 * NV; c < nch; c += 256) {
    f32x4 v0, v1;
    fetch(c, v0, v1);                 // (rows wider than 8192: the residual lands in memory here and is re-read below)
#pragma unroll
    for (int t = 0; t < 4; t++) { ss = fmaf(v0[t], v0[t], ss); ss = fmaf(v1[t], v1[t], ss); }
  }
  ss = block_sum_256(ss, sc);
  const float inv = 1.0f / sqrtf(ss / (float)H + eps);
  auto emit = [&](int c, const f32x4& v0, const f32x4& v1, const u32x4 wv) {
    unsigned int h[4], l[4], l2[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float xa = t < 2 ? v0[2 * t] : v1[2 * t - 4], xb = t < 2 ? v0[2 * t + 1] : v1[2 * t - 3];
      const float ya = pair_lo<DT>(wv[t]) * (xa * inv), yb = pair_hi<DT>(wv[t]) * (xb * inv);
      bf16_t ha, la, hb, lb;
      split16<DT>(ya, ha, la);
      split16<DT>(yb, hb, lb);
      h[t] = (unsigned int)ha | ((unsigned int)hb << 16);
      l[t] = (unsigned int)la | ((unsigned int)lb << 16);
      if (lo2) {
        const bf16_t ta = f32_to_elem<DT>(ya - elem_to_f32<DT>(ha) - elem_to_f32<DT>(la)), tb = f32_to_elem<DT>(yb - elem_to_f32<DT>(hb) - elem_to_f32<DT>(lb));
        l2[t] = (unsigned int)ta | ((unsigned int)tb << 16);
      }
    }
    const size_t o = (size_t)blockIdx.x * H + 8 * c;
    if (inter) {
      const size_t oi = ((size_t)blockIdx.x * (H >> 5) + (c >> 2)) * 64 + (c & 3) * 8;
      *reinterpret_cast<u32x4*>(hi + oi) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4*>(hi + oi + 32) = u32x4{l[0], l[1], l[2], l[3]};
      return;
    }
    *reinterpret_cast<u32x4*>(hi + o) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(lo + o) = u32x4{l[0], l[1], l[2], l[3]};
    if (lo2) *reinterpret_cast<u32x4*>(lo2 + o) = u32x4{l2[0], l2[1], l2[2], l2[3]};
  };
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = threadIdx.x + 256 * j;
    if (c < nch) emit(c, xr[j][0], xr[j][1], wr[j]);
  }
  for (int c = threadIdx.x + 256 * NV; c < nch; c += 256)
    emit(c, *reinterpret_cast<const f32x4*>(x + 8 * c), *reinterpret_cast<const f32x4*>(x + 8 * c + 4), *reinterpret_cast<const u32x4*>(w + 8 * c));
}

// ---- split -> RoPE(q), RoPE(k) at pastLength+s -> cache append; q as bf16 hi/lo ------------------------------------
struct RopeKvArgs {
  const float* QKV;        // [S][qd + 2*kvd] fp32 (bias already added)
  bf16_t *q_hi, *q_lo;     // [S][qd]
  bf16_t *k_cache, *v_cache;   // this layer/row: [kv_heads][max_ctx][hd]
  const float *rope_cos, *rope_sin;
  int heads, kv_heads, hd, max_ctx, past;
  const bf16_t *q_norm_w, *k_norm_w;   // Qwen3 per-head RMSNorm weights [hd] (nullptr: no QK-norm)
  float eps;
  // the QKV product left as split-K slabs (QKV == nullptr): row s = sum_z part[z][s][:] + bias, z order as gemm_splitk_reduce_kernel<GEMM_STORE>
  const float* part; int nsplit; long long slab; const bf16_t* bias;
  const int* blk_tbl;      // paged KV (common.h kv_paged_off): this sequence's block table, k_cache / v_cache = the layer's pools — or nullptr
};
template <int DT>
__global__ __launch_bounds__(256) void rope_kv_split_kernel(const RopeKvArgs a) {
  // one workgroup per position; a thread owns FOUR adjacent RoPE pairs (p..p+3, p+half..p+half+3) of one head: 16-byte loads, 8-byte stores
  const int s = blockIdx.x, pos = a.past + s, half = a.hd >> 1, q4 = half >> 2;
  const int qd = a.heads * a.hd, kvd = a.kv_heads * a.hd;
  const float* row = (a.QKV ? a.QKV : a.part) + (size_t)s * (qd + 2 * kvd);
  const int units = (a.heads + 2 * a.kv_heads) * q4;
  auto fetch4 = [&](int col) {
    if (a.QKV) return *reinterpret_cast<const f32x4*>(row + col);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= a.nsplit; z += 8) {        // eight slabs in flight together, summed in z order (as rmsnorm_split_kernel)
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = *reinterpret_cast<const f32x4*>(row + (size_t)(z + u) * a.slab + col);
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += t[u][e];
    }
    if (z + 4 <= a.nsplit) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) t[u] = *reinterpret_cast<const f32x4*>(row + (size_t)(z + u) * a.slab + col);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += t[u][e];
      z += 4;
    }
    for (; z < a.nsplit; z++) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(row + (size_t)z * a.slab + col);
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] += t[e];
    }
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] += elem_to_f32<DT>(a.bias[col + e]);
    }
    return v;
  };
  auto pack4 = [](const bf16_t* e) { return u32x2{(unsigned int)e[0] | ((unsigned int)e[1] << 16), (unsigned int)e[2] | ((unsigned int)e[3] << 16)}; };
  for (int u = threadIdx.x; u < units; u += 256) {
    const int hh = u / q4, p = 4 * (u - hh * q4);
    f32x4 x0 = fetch4(hh * a.hd + p), x1 = fetch4(hh * a.hd + p + half);
    if (a.q_norm_w != nullptr && hh < a.heads + a.kv_heads) {
      // AttentionWithQKNorm (Attention.h:156-163): RMSNorm over head_dim; the hd/8 lanes of one head are adjacent and
      // aligned (8 or 16), so the mean of squares is a sub-wave butterfly
      float ss = 0.f;
#pragma unroll
      for (int t = 0; t < 4; t++) ss += x0[t] * x0[t] + x1[t] * x1[t];
      for (int o = q4 >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float inv = 1.0f / sqrtf(ss / (float)a.hd + a.eps);
      const bf16_t* w = hh < a.heads ? a.q_norm_w : a.k_norm_w;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        x0[t] = elem_to_f32<DT>(w[p + t]) * (x0[t] * inv);
        x1[t] = elem_to_f32<DT>(w[p + t + half]) * (x1[t] * inv);
      }
    }
    if (hh < a.heads + a.kv_heads) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos * half + p), sn = *reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos * half + p);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        float r0 = x0[t], r1 = x1[t];
        rope_rotate_pair(r0, r1, cs[t], sn[t]);         // one spelling of the rotation in every RoPE + append site (common.h): the paths append identical rows
        x0[t] = r0; x1[t] = r1;
      }
    }
    if (hh < a.heads) {
      bf16_t h0[4], l0[4], h1[4], l1[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { split16<DT>(x0[t], h0[t], l0[t]); split16<DT>(x1[t], h1[t], l1[t]); }
      const size_t o = (size_t)s * qd + hh * a.hd + p;
      *reinterpret_cast<u32x2*>(a.q_hi + o) = pack4(h0); *reinterpret_cast<u32x2*>(a.q_lo + o) = pack4(l0);
      *reinterpret_cast<u32x2*>(a.q_hi + o + half) = pack4(h1); *reinterpret_cast<u32x2*>(a.q_lo + o + half) = pack4(l1);
    } else {
      const bool is_k = hh < a.heads + a.kv_heads;
      const int kh = is_k ? hh - a.heads : hh - a.heads - a.kv_heads;
      bf16_t* dst = (is_k ? a.k_cache : a.v_cache) + (a.blk_tbl ? kv_paged_off(a.blk_tbl, a.kv_heads, kh, pos, a.hd) : ((size_t)kh * a.max_ctx + pos) * a.hd);
      bf16_t e0[4], e1[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { e0[t] = f32_to_elem<DT>(x0[t]); e1[t] = f32_to_elem<DT>(x1[t]); }
      *reinterpret_cast<u32x2*>(dst + p) = pack4(e0);
      *reinterpret_cast<u32x2*>(dst + p + half) = pack4(e1);
    }
  }
}

// ---- C[M][N] (+)= (Ahi + Alo)[M][K] · B[N][K]ᵀ on v_mfma_f32_32x32x16_bf16 ------------------------------------------
// 128x128 workgroup tile, BK = 64, 4 waves as 2x2 (each 64x64 = 2x2 MFMA tiles, 64 accumulator VGPRs).  Tiles are
// staged global -> registers -> LDS with the next K-step's global loads in flight during the MFMAs; LDS rows are
// padded to 144 B so the 16-byte fragment reads of a 16-lane group fall on 16 distinct bank quads.
enum { GEMM_STORE = 0, GEMM_RESIDUAL = 1, GEMM_SILU = 2, GEMM_PARTIAL = 3, GEMM_GELU = 4 };   // PARTIAL: split-K slab, finished by gemm_splitk_reduce_kernel
// GEMM_GELU (GPT-2's c_fc, ModelGPT2.h:96-107): out_hi / out_lo [M][N] = the 16-bit terms of gelu_new(acc + bias) — the c_proj product's A operand
struct GemmArgs {
  const bf16_t *A_hi, *A_lo;   // [M][K]
  const bf16_t* A_lo2;         // optional third term (nullptr: two-term product)
  const bf16_t* B;             // [N][K] (torch Linear weight)
  const bf16_t* bias;          // [N] or nullptr
  float* C;                    // [M][ldc]
  int M, N, K, ldc;
  // GEMM_SILU (the gate_up product): tile columns alternate gate row i / up row i (B row of column n = (n odd ? inter : 0) + n/2),
  // so a lane pair holds (gate_i, up_i) and the epilogue emits siluMul directly as the hi / lo operand of the down product:
  // the [M][2*inter] fp32 intermediate and the separate siluMul pass disappear (GatedMLP.h:37-39, Activation.h:16)
  int inter;
  bf16_t *out_hi, *out_lo;     // [M][inter]
  // split-K (few row tiles: a short prompt): blockIdx.z covers k_per elements of K and stores its fp32 partial tile to part[z][M][N];
  // gemm_splitk_reduce_kernel sums the slabs in z order and applies the epilogue.  interleave = 1: gate/up column order as for GEMM_SILU
  float* part;
  int k_per, nsplit, interleave;
  // skinny_gemm_kernel (skinny.h) may take its activations as fp32 rows and split them into 16-bit terms while staging them in LDS
  // (ASRC 1), optionally applying RMSNorm on the way (ASRC 2: y = norm_w[k] * (x * rsqrt(sum_cb ssq_part[m][cb] / K + eps)), the sums of
  // squares left by the kernel that produced x): no separate norm / split launch, no 16-bit copy of the activations in memory
  const float* A_f32;          // [M][lda]
  int lda;
  const bf16_t* norm_w;        // [K]
  const float* ssq_part;       // [M][ssq_ncb]
  int ssq_ncb;
  float eps;
  float* ssq_out;              // reduce_rows_kernel: [M][gridDim.y] partial sums of squares of the rows it writes
  // gemm_x2_kernel with A_lo2: tile columns below three_from take two terms only.  The third term exists for results that are rounded
  // to 16 bits again — the K / V columns of the QKV product; its Q columns (the first heads*head_dim) stay fp32.
  int three_from;
  // gemm_dma_qkv8_kernel<DT, ROPE = true> (head_dim 64, one sequence): split + RoPE(q), RoPE(k) + cache append + q as two 16-bit terms run in the product's epilogue
  // (== rope_kv_split_kernel on the finished rows, same arithmetic: Attention.h:96-106); row m sits at position rope_past + m
  bf16_t *rope_q_hi, *rope_q_lo;     // [M][three_from]
  bf16_t *rope_k, *rope_v;           // this layer: [kv_heads][rope_max_ctx][64]
  const float *rope_cos, *rope_sin;  // [max_ctx][32]
  int rope_past, rope_max_ctx, rope_kv_heads;
  const int* rope_tbl;           // paged KV: the sequence's block table (rope_k / rope_v = the layer's pools); nullptr = one slab per row
};

// GEMM_SILU epilogue of one 32 x 32 accumulator block (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)): even lanes hold
// gate_i, odd lanes up_i (column parity).  The lane pair shares the block's 16 registers: the even lane finishes r = 0, 2, ..., the odd lane
// r + 1 (one DPP exchange per register pair), so every lane computes and stores a result.  row_base = the block's first row.
template <int DT>
__device__ __forceinline__ void silu_block_store(const f32x16& acc, int lane, int col, int row_base, const GemmArgs& a) {
  const bool odd = lane & 1;
  const bool col_ok = col < a.N;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const float a0 = acc[r], a1 = acc[r + 1];
    const float recv = dpp_mov<0xB1, 0xf>(odd ? a0 : a1);        // quad_perm [1,0,3,2]: what the neighbour lane needs from this one
    const float g = odd ? recv : a0, u = odd ? a1 : recv;
    const int row = row_base + (r & 3) + (odd ? 1 : 0) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (col_ok && row < a.M) {
      const size_t o = (size_t)row * a.inter + (size_t)(col >> 1);
      split16<DT>(silu_mul_fast(g, u), a.out_hi[o], a.out_lo[o]);
    }
  }
}

constexpr int GBM = 128, GBN = 128, GBK = 64, GLD = GBK + 8;

// MI = 32-row blocks per wave along M: 2 -> the 128x128 tile; 1 -> a 64x128 tile for the products with few column tiles
// (N = hidden: o_proj, down_proj), which would otherwise put one workgroup on each CU and leave the matrix cores waiting
// on every barrier.
template <int DT, int EPI, int MI>
__global__ __launch_bounds__(256) void gemm_x2_kernel(const GemmArgs a) {
  constexpr int TM = 64 * MI;                 // tile rows
  constexpr int AI = TM * 8 / 256;            // 16-byte A chunks per thread and K-step
  __shared__ __attribute__((aligned(16))) bf16_t sAh[TM * GLD];
  __shared__ __attribute__((aligned(16))) bf16_t sAl[TM * GLD];
  __shared__ __attribute__((aligned(16))) bf16_t sB[GBN * GLD];
  extern __shared__ __attribute__((aligned(16))) bf16_t sAl2[];   // [TM*GLD] only when the launch asks for it
  const bool three = a.A_lo2 != nullptr && (int)(blockIdx.x + 1) * GBN > a.three_from;      // workgroup-uniform
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * GBN;
  const int kch = a.K >> 3;                                    // 16-byte chunks per row

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // staging map: chunk c = tid + 256*i -> tile row c/8, 16-byte column c%8
  u32x4 rah[AI], ral[AI], ral2[AI], rb[4];
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      const size_t koff = (size_t)(k0 >> 3) + kc;
      if (i < AI) {
        const bool am = m0 + row < a.M;
        rah[i < AI ? i : 0] = am ? reinterpret_cast<const u32x4*>(a.A_hi)[(size_t)(m0 + row) * kch + koff] : zero;
        ral[i < AI ? i : 0] = am ? reinterpret_cast<const u32x4*>(a.A_lo)[(size_t)(m0 + row) * kch + koff] : zero;
        ral2[i < AI ? i : 0] = (three && am) ? reinterpret_cast<const u32x4*>(a.A_lo2)[(size_t)(m0 + row) * kch + koff] : zero;
      }
      const bool bn = n0 + row < a.N;
      const int nb = n0 + row;
      const size_t brow = (EPI == GEMM_SILU || (EPI == GEMM_PARTIAL && a.interleave)) ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
      rb[i] = bn ? reinterpret_cast<const u32x4*>(a.B)[brow * kch + koff] : zero;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
      if (i < AI) {
        *reinterpret_cast<u32x4*>(&sAh[row * GLD + kc * 8]) = rah[i < AI ? i : 0];
        *reinterpret_cast<u32x4*>(&sAl[row * GLD + kc * 8]) = ral[i < AI ? i : 0];
        if (three) *reinterpret_cast<u32x4*>(&sAl2[row * GLD + kc * 8]) = ral2[i < AI ? i : 0];
      }
      *reinterpret_cast<u32x4*>(&sB[row * GLD + kc * 8]) = rb[i];
    }
  };

  const int k_begin = EPI == GEMM_PARTIAL ? (int)blockIdx.z * a.k_per : 0;
  const int k_end = EPI == GEMM_PARTIAL ? min(a.K, k_begin + a.k_per) : a.K;
  if (k_begin < k_end) load_tiles(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
    __syncthreads();                 // everyone is done reading the previous tile
    store_tiles();
    __syncthreads();
    if (k0 + GBK < k_end) load_tiles(k0 + GBK);   // next K-step's global loads fly under the MFMAs
#pragma unroll
    for (int kk = 0; kk < GBK / 16; kk++) {
      const int kcol = kk * 16 + 8 * (lane >> 5);
      bf16x8 fah[MI], fal[MI], fal2[MI], fb[2];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int row = wm * (32 * MI) + i * 32 + (lane & 31);
        fah[i] = *reinterpret_cast<const bf16x8*>(&sAh[row * GLD + kcol]);
        fal[i] = *reinterpret_cast<const bf16x8*>(&sAl[row * GLD + kcol]);
        if (three) fal2[i] = *reinterpret_cast<const bf16x8*>(&sAl2[row * GLD + kcol]);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int row = wn * 64 + j * 32 + (lane & 31);
        fb[j] = *reinterpret_cast<const bf16x8*>(&sB[row * GLD + kcol]);
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (three) acc[i][j] = mfma16<DT>(fal2[i], fb[j], acc[i][j]);
          acc[i][j] = mfma16<DT>(fal[i], fb[j], acc[i][j]);   // small terms first
          acc[i][j] = mfma16<DT>(fah[i], fb[j], acc[i][j]);
        }
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (EPI == GEMM_SILU) {
        silu_block_store<DT>(acc[i][j], lane, col, m0 + wm * (32 * MI) + i * 32, a);
        continue;
      }
      if (col >= a.N) continue;
      if (EPI == GEMM_PARTIAL) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < a.M) a.part[((size_t)blockIdx.z * a.M + row) * a.N + col] = acc[i][j][r];
        }
        continue;
      }
      const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

// Sums the split-K slabs in z order and applies the epilogue the unsplit kernel would have applied (EPI as above).
template <int DT, int EPI>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmArgs a) {
  const int ncol = EPI == GEMM_SILU ? a.N / 2 : a.N;               // SILU: one thread per (gate, up) pair
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.M * ncol) return;
  const int row = (int)(idx / ncol), c = (int)(idx - (size_t)row * ncol);
  const size_t slab = (size_t)a.M * a.N;
  if (EPI == GEMM_SILU) {
    const float* p = a.part + (size_t)row * a.N + 2 * c;
    float g = 0.f, u = 0.f;
    for (int z = 0; z < a.nsplit; z++) { g += p[z * slab]; u += p[z * slab + 1]; }
    const size_t o = (size_t)row * a.inter + c;
    split16<DT>(silu_mul_fast(g, u), a.out_hi[o], a.out_lo[o]);      // the same formulation as the unsplit epilogues: a prompt's activations do not depend on whether split-K was chosen
    return;
  }
  const float* p = a.part + (size_t)row * a.N + c;
  float v = 0.f;
  for (int z = 0; z < a.nsplit; z++) v += p[z * slab];
  if (a.bias) v += elem_to_f32<DT>(a.bias[c]);
  if (EPI == GEMM_GELU) {
    const size_t o = (size_t)row * a.N + c;
    split16<DT>(gelu_new_fast(v), a.out_hi[o], a.out_lo[o]);
    return;
  }
  float* dst = a.C + (size_t)row * a.ldc + c;
  *dst = (EPI == GEMM_RESIDUAL) ? (*dst + v) : v;
}

// ---- causal GQA flash attention over the cache, queries past..past+S-1 -------------------------------------------------
struct AttnPrefillArgs {
  const bf16_t *q_hi, *q_lo;     // [S][heads*hd]
  const bf16_t *k_cache, *v_cache;   // [kv_heads][max_ctx][hd]
  bf16_t *o_hi, *o_lo;           // [S][heads*hd]
  int S, heads, kv_heads, max_ctx, past;
  float scale;                   // hd^-1/2
  int qblk_mirror;               // 1: see the block order in the kernel
  int heavy_first;               // 1 (the key-split form): grid = (heads, query blocks), blockIdx.y = rank of the block from the heaviest down — see the kernel
  const int* blk_tbl;            // paged KV (kernel template PAGED): this sequence's block table, k_cache / v_cache = the layer's pools
};

// Scores and probabilities never leave the registers (the first version of this round routed them through LDS with four
// barriers per tile: 262 us per layer at S = 2048 against 165 us for this one).
// One workgroup = 128 queries of one query head (4 waves x 32 queries); K / V tiles of 64 keys go global -> registers -> LDS
// as they are ([key][d], padded rows), shared by the four waves.  Each wave computes S^T = K.Q^T — the MFMA's output layout then gives
// every lane ONE query column (16 keys per 32-key sub-tile; the other half-wave holds the other 16), so the online softmax is lane-local
// plus one exchange with lane^32, the running maximum / sum / rescale are per-lane scalars, and the probabilities are already
// in the B-operand order of the next MFMA: O^T += V^T.P^T, where the V^T fragment comes out of the [key][d] tile through the transposing
// LDS read (ds_read_b64_tr_b16) in the SAME key permutation (keys 4hh..4hh+3 and 8+4hh..8+4hh+3 of every 16: two 8-byte reads).
// Q (hi, lo) lives in registers for the whole kernel; the output rows leave through an LDS transpose as whole rows.
#ifndef TGX_ATTN_DIS
#define TGX_ATTN_DIS 0      // experiments only (tools/probes/attn_prefill_probe.hip): 1 no LDS staging, 2 no softmax arithmetic, 4 no PV, 8 no QK^T, 16 no tile fetch
#endif
// LA = K / V tiles of look-ahead.  2 (two register sets) is the faster form while a CU holds two workgroups (S <= ~2k: 61 vs 65 µs per layer at 2048);
// long prompts give every CU three or more, and the leaner LA = 1 form (162 VGPRs) then runs three waves per SIMD: 251 -> 203 µs per layer at S = 4096, 820 -> 720 at 8192.
// KP = key split inside the workgroup (round 5).  A wave walks its tiles one after the other (QK^T chain -> softmax -> PV chain: ~2-3 thousand cycles per tile
// whatever else the CU does), so a workgroup's time is tiles x that latency, and at S = 2048 the grid is ONE round of 512 workgroups whose last query block walks
// 32 tiles: the launch lasted as long as its heaviest workgroup (56-64 us) although the CUs' average load is half of that.  KP = 2: eight waves — waves 4-7 repeat
// the query sub-blocks of waves 0-3 on the ODD tiles, 0-3 take the even ones; two tiles are staged per barrier pair; the two online-softmax streams of a
// query meet once, through LDS, after the loop ((m, l, O) merged as flash-decoding does).  One workgroup per CU instead of two, the same matrix work per CU and
// cycle, half the chain; blocks are dispatched heaviest first (AttnPrefillArgs.heavy_first) so that the second round fills the CUs as they free up.
// PAGED (round 6): the workgroup copies the part of the sequence's block table it needs (<= 1024 entries) into LDS first; a K / V row's address then takes one LDS
// read (it does not touch the counted vmcnt waits of the tile pipeline) — a 64-key tile never straddles a 128-token page.
template <int DT, int HD, int LA = 2, int KP = 1, bool PAGED = false>
__global__ __launch_bounds__(256 * KP, KP == 2 ? 2 : (LA == 1 ? (HD == 64 ? 3 : 2) : 1)) void attn_prefill_kernel(const AttnPrefillArgs a) {      // head_dim 128: LA = 1 fits two waves per SIMD (LA = 2 needs 292 registers: one)
  constexpr int DIS = TGX_ATTN_DIS;
  constexpr int LQ = HD + 8;                  // 16-bit row stride of the K tile (144 / 272 B: conflict-free 16-byte fragment reads)
  constexpr int LV = HD + 32;                 // 16-bit row stride of the V tile ([key][d], 64 B more than a row: the four key rows of a transposing read fall on four bank quarters)
  constexpr int KS = HD / 16;                 // MFMA k-steps over the head dimension
  constexpr int NB = HD / 32;                 // 32-row output-dim blocks
  constexpr int CH = HD / 8;                  // 16-byte chunks per head row
  constexpr int NCH = 64 * CH / 256;          // chunks per thread and tile
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr int REG = 64 * LQ + 64 * LV;      // one K tile | V tile pair, 16-bit elements
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];     // KP x (K tile | V tile); the output rows (and the KP = 2 merge) pass through it at the end
  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, kp = tid >> 8, hh = lane >> 5, ql = lane & 31;      // kp: this wave's tile parity (0 when KP == 1)
  bf16_t* const sK = smem + kp * REG;          // the tile this wave multiplies — and the one its thread group stages
  bf16_t* const sV = sK + 64 * LQ;

  const int h = a.heavy_first ? blockIdx.x : blockIdx.y, G = a.heads / a.kv_heads, kvh = h / G;
  const int qd = a.heads * HD;
  // causal work grows with the query block: the upper half of the heads walks the blocks in reverse, so that workgroups b and b + half
  // the grid (which tend to share a CU) carry complementary amounts.  heavy_first: every head's last block first (linear workgroup order = dispatch order)
  const int qblk = a.heavy_first ? (int)gridDim.y - 1 - (int)blockIdx.y
                                 : ((a.qblk_mirror && h >= a.heads / 2) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
  const int q0 = qblk * 128 + wv * 32;                 // first query of this wave
  const int qi = q0 + ql;                              // this lane's query
  const bool qvalid = qi < a.S;
  const int qpos = a.past + qi;

  bf16x8 qh[KS], qlo[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    u32x4 vh = u32x4{0u, 0u, 0u, 0u}, vl = vh;
    if (qvalid) {
      const size_t o = (size_t)qi * qd + (size_t)h * HD + kk * 16 + 8 * hh;
      vh = *reinterpret_cast<const u32x4*>(a.q_hi + o);
      vl = *reinterpret_cast<const u32x4*>(a.q_lo + o);
    }
    qh[kk] = __builtin_bit_cast(bf16x8, vh);
    qlo[kk] = __builtin_bit_cast(bf16x8, vl);
  }
  f32x16 oacc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[b][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* kbase = a.k_cache + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD);
  const bf16_t* vbase = a.v_cache + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD);
  const int wg_last_pos = a.past + min(qblk * 128 + 127, a.S - 1);   // keys beyond it are never attended by this workgroup
  __shared__ int stbl[PAGED ? 1024 : 1];
  if constexpr (PAGED) {
    for (int i = tid; i <= (wg_last_pos >> KV_BLOCK_SHIFT); i += 256 * KP) stbl[i] = a.blk_tbl[i];
    __syncthreads();
  }
  auto key_off = [&](int key) -> size_t {      // element offset of a key's row from kbase / vbase
    if constexpr (PAGED) return (((size_t)stbl[key >> KV_BLOCK_SHIFT] * a.kv_heads + kvh) * KV_BLOCK + (key & (KV_BLOCK - 1))) * (size_t)HD;
    else return (size_t)key * HD;
  };
  const int n_kt = (wg_last_pos / 64 + 1 + KP - 1) / KP;      // steps of KP tiles
  const int wave_last_pos = a.past + min(q0 + 31, a.S - 1);
  const bool wave_live = q0 < a.S;
  const float qs = a.scale * LOG2E;

  // K / V tiles travel global -> registers -> LDS, TWO tiles ahead of the one being multiplied (two register sets, static roles): with one
  // tile of look-ahead a workgroup's tile time was the load latency itself (~2.4 µs per tile, MfmaUtil 16 %) whatever the arithmetic cost.
  // Keys past the workgroup's range are clamped to its last key (a written cache row: finite values, masked by index below).
  u32x4 kvA[NCH], vvA[NCH], kvB[NCH], vvB[NCH];
  // (KP = 2: step kt = tiles 2 kt and 2 kt + 1; thread group kp fetches, stages and multiplies tile 2 kt + kp)
  auto fetch_tile = [&](int kt, u32x4* kr, u32x4* vr) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = (tid & 255) + 256 * i, row = c / CH, kc = c - row * CH;
      const int key = min((KP * kt + kp) * 64 + row, wg_last_pos);
      const size_t ko = key_off(key) + kc * 8;
      kr[i] = *reinterpret_cast<const u32x4*>(kbase + ko);
      __builtin_amdgcn_sched_barrier(0);       // the same issue order at every call site: the counted vmcnt waits rely on it
      vr[i] = *reinterpret_cast<const u32x4*>(vbase + ko);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage_tile = [&](const u32x4* kr, const u32x4* vr) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = (tid & 255) + 256 * i, row = c / CH, kc = c - row * CH;
      const u32x4 kv = kr[i], vv = vr[i];
      *reinterpret_cast<u32x4*>(&sK[row * LQ + kc * 8]) = kv;
      *reinterpret_cast<u32x4*>(&sV[row * LV + kc * 8]) = vv;       // V stays [key][d]: the PV step reads it with the transposing LDS read
    }
  };
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; r++) zero16[r] = 0.f;

  auto compute_tile = [&](int kt) {
    const int key0 = (KP * kt + kp) * 64;
    if (!wave_live || key0 > wave_last_pos) return;      // wave-uniform: nothing of this tile is visible to the wave's queries

    // S^T sub-tiles: sacc[sub][r] = raw score (q.k) of key key0 + 32 sub + (r&3) + 8 (r>>2) + 4 hh for this lane's query
    f32x16 sacc[2];
    float mx = -INFINITY;
    const bool diag = key0 + 63 > a.past + q0;           // wave-uniform: some (key, query) pair of this tile needs the causal mask
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const int kb = key0 + 32 * sub;
      // both sub-tiles always (keys past the workgroup's range are zero rows in LDS and masked below): a skip for the half-masked diagonal
      // tile costs register copies at the join on every tile
      if (DIS & 8) sacc[sub] = zero16;
#pragma unroll
      for (int kk = 0; kk < ((DIS & 8) ? 0 : KS); kk++) {
        const bf16x8 fk = *reinterpret_cast<const bf16x8*>(&sK[(32 * sub + ql) * LQ + kk * 16 + 8 * hh]);
        sacc[sub] = mfma16<DT>(fk, qlo[kk], kk == 0 ? zero16 : sacc[sub]);      // the chain starts from the constant 0: no register clearing
        sacc[sub] = mfma16<DT>(fk, qh[kk], sacc[sub]);
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
    if (DIS & 2) {                                        // experiment: scores go to PV as they are
      l_run += 1.f;
    } else {
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;          // qs > 0: scaling commutes with the maximum
    const float m_new = fmaxf(m_run, mx);
    const bool dead = m_new == -INFINITY;               // padded query: nothing attended yet
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
    const float neg_m = dead ? 0.f : -m_new;
    float sum = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[sub][r], qs, neg_m));   // exp2(-inf) = 0 for masked entries
        sacc[sub][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (__any(alpha != 1.f)) {                   // the running maximum settles after the first tiles: most steps leave O as it is
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[b][r] *= alpha;
    }

    }
    // O^T += V^T . P^T over the four 16-key steps of the tile; B-operand element j of step s is register 8 s' + j of its sub-tile
#pragma unroll
    for (int sub = 0; sub < ((DIS & 4) ? 0 : 2); sub++) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        unsigned int wh[4], wl[4];
        if constexpr (DT == DT_BF16) {          // pairs: one packed convert per term, the residual from the packed hi word itself
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float p0 = sacc[sub][8 * s2 + 2 * j], p1 = sacc[sub][8 * s2 + 2 * j + 1];
            wh[j] = pack_bf16(p0, p1);
            wl[j] = pack_bf16(p0 - bf16_lo(wh[j]), p1 - bf16_hi(wh[j]));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            bf16_t ph, pl;
            split16<DT>(sacc[sub][8 * s2 + j], ph, pl);
            if (j & 1) { wh[j >> 1] |= (unsigned int)ph << 16; wl[j >> 1] |= (unsigned int)pl << 16; }
            else { wh[j >> 1] = ph; wl[j >> 1] = pl; }
          }
        }
        const bf16x8 fph = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
        const bf16x8 fpl = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
        const int kloc = 32 * sub + 16 * s2 + 4 * hh;     // tile-local key of element 0; elements 4..7 are 8 keys further
#pragma unroll
        for (int b = 0; b < NB; b++) {
          // ds_read_b64_tr_b16: the 16 lanes of a group point at the [4 keys][16 dims] block (lane i: key i >> 2, dims 4 (i & 3)..+3) and
          // lane i receives column i — the four consecutive keys of ITS dim (tools/probes/tr_read_probe.hip)
          const bf16_t* vblk = &sV[(kloc + ((lane & 15) >> 2)) * LV + 32 * b + (lane & 16) + 4 * (lane & 3)];
          const u32x2 v0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk)));
          const u32x2 v1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 8 * LV)));
          const bf16x8 fv = __builtin_bit_cast(bf16x8, u32x4{v0[0], v0[1], v1[0], v1[1]});
          oacc[b] = mfma16<DT>(fv, fpl, oacc[b]);
          oacc[b] = mfma16<DT>(fv, fph, oacc[b]);
        }
      }
    }
  };

  if constexpr (LA == 1) {
    fetch_tile(0, kvA, vvA);
    for (int kt = 0; kt < n_kt; kt++) {
      __syncthreads();
      if (!(DIS & 1)) stage_tile(kvA, vvA);
      __syncthreads();
      if (!(DIS & 16)) fetch_tile(kt + 1, kvA, vvA);            // (past the range: a clamped reload)
      compute_tile(kt);
    }
  } else {
  __builtin_amdgcn_sched_barrier(0);           // the loop's counted waits assume the issue order: set A, then set B (older loads retire first)
  fetch_tile(0, kvA, vvA);
  __builtin_amdgcn_sched_barrier(0);
  fetch_tile(1, kvB, vvB);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < n_kt; kt += 2) {
    __syncthreads();                           // the previous tile is consumed by every wave
    if (!(DIS & 1)) stage_tile(kvA, vvA);
    __syncthreads();
    if (!(DIS & 16)) fetch_tile(kt + 2, kvA, vvA);              // in flight during two tiles of MFMAs
    compute_tile(kt);
    // no early exit for an odd tile count: a break here gives the loop a second back edge on which set A is the youngest, and every wait
    // above becomes a drain; the extra tile is a clamped reload that compute_tile skips (key0 > wave_last_pos)
    __syncthreads();
    if (!(DIS & 1)) stage_tile(kvB, vvB);
    __syncthreads();
    if (!(DIS & 16)) fetch_tile(kt + 3, kvB, vvB);
    compute_tile(kt + 1);
  }
  }

  // normalise and emit as hi / lo 16-bit pairs (the o_proj GEMM's A operand).  A lane holds ONE query's output dims, so direct stores would be
  // 2-byte writes 4 KB apart (64 cache lines per instruction: ~20 µs of the kernel at S = 2048); instead each wave transposes its 32 x HD block
  // through LDS (rows of 2 HD + 8 bytes: conflict-free 4-byte column writes) and stores whole rows, 16 bytes per lane.
  __syncthreads();                              // every wave is done with the last K / V tile
  if constexpr (KP == 2) {
    // the odd-tile stream of every query joins the even-tile one: lane-to-lane (the two waves hold the same queries in the same layout)
    constexpr int NV = NB * 16 + 2;
    static_assert(4 * NV * 64 * 4 <= KP * REG * 2, "the merge buffer exceeds the K / V tiles");
    float* const mo = reinterpret_cast<float*>(smem) + (size_t)wv * NV * 64 + lane;
    if (kp == 1) {
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) mo[(b * 16 + r) * 64] = oacc[b][r];
      mo[(NB * 16) * 64] = m_run; mo[(NB * 16 + 1) * 64] = l_run;
    }
    __syncthreads();
    if (kp == 0) {
      const float m_b = mo[(NB * 16) * 64], l_b = mo[(NB * 16 + 1) * 64];
      const float m_new = fmaxf(m_run, m_b);
      const bool dead = m_new == -INFINITY;
      const float fa = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new), fb = dead ? 1.f : __builtin_amdgcn_exp2f(m_b - m_new);
      l_run = l_run * fa + l_b * fb;
      m_run = m_new;
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[b][r] = oacc[b][r] * fa + mo[(b * 16 + r) * 64] * fb;
    }
    __syncthreads();                            // the merge buffer is read: the staging rows below reuse it
    if (kp == 1) return;
  }
  constexpr int RS = HD + 4;                    // staged row stride in 16-bit elements
  static_assert(4 * 32 * RS <= 64 * LQ + 64 * LV, "output staging exceeds the K / V tiles");
  bf16_t* const wrow = smem + wv * 32 * RS;
  const float inv_l = qvalid ? 1.0f / l_run : 0.f;
  unsigned int whi[NB * 8], wlo[NB * 8];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float v0 = oacc[b][r] * inv_l, v1 = oacc[b][r + 1] * inv_l;
      bf16_t h0, l0, h1, l1;
      split16<DT>(v0, h0, l0);
      split16<DT>(v1, h1, l1);
      whi[b * 8 + (r >> 1)] = (unsigned int)h0 | ((unsigned int)h1 << 16);
      wlo[b * 8 + (r >> 1)] = (unsigned int)l0 | ((unsigned int)l1 << 16);
    }
#pragma unroll
  for (int term = 0; term < 2; term++) {
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int d = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hh;                 // dims d, d + 1
        *reinterpret_cast<unsigned int*>(wrow + ql * RS + d) = term ? wlo[b * 8 + (r >> 1)] : whi[b * 8 + (r >> 1)];
      }
    bf16_t* const dst = term ? a.o_lo : a.o_hi;
#pragma unroll
    for (int i = 0; i < HD / 16; i++) {          // 32 rows x HD / 8 chunks of 16 bytes over 64 lanes
      const int c = lane + 64 * i, row = c / CH, cc = c - row * CH;
      const u32x2 p0 = *reinterpret_cast<const u32x2*>(wrow + row * RS + cc * 8);
      const u32x2 p1 = *reinterpret_cast<const u32x2*>(wrow + row * RS + cc * 8 + 4);
      if (q0 + row < a.S)
        *reinterpret_cast<u32x4*>(dst + (size_t)(q0 + row) * qd + (size_t)h * HD + cc * 8) = u32x4{p0[0], p0[1], p1[0], p1[1]};
    }
  }
}

}  // namespace tgx
