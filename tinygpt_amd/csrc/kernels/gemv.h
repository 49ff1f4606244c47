// gemv.h — the batch-1 weight-streaming kernel of the decode path: y = W[N,K] · x (bf16 weights, fp32
// accumulate), with the ops the reference issues around each nn::Linear fused in as prologue/epilogue.
//
//   reference op sequence (per decode step)                         fused form here
//   RMSNorm -> MergedLinear qkv -> split -> RoPE(q),RoPE(k)          PRO_RMSNORM + EPI_QKV_ROPE
//     -> KVCacheManager::append (Attention.h:94-106)                   (K/V land directly in the cache slot)
//   Linear o_proj -> x + .  (Attention.h:90, DecoderLayer.h:40)      PRO_PLAIN   + EPI_RESIDUAL
//   RMSNorm -> MergedLinear gate_up -> siluMul (GatedMLP.h:37-39)    PRO_RMSNORM + EPI_SILU_MUL
//   Linear down_proj -> x + .  (GatedMLP.h:40, DecoderLayer.h:41)    PRO_PLAIN   + EPI_RESIDUAL
//   RMSNorm -> Linear lm_head -> argmax (GPTModel.h:56-57,            PRO_RMSNORM + EPI_LOGITS
//     Sampler.cpp:28)
//
// Roofline: HBM.  Algorithmic bytes per launch = 2*N*K (each weight byte is read exactly once, with the
// non-temporal policy); x (<= 28 KB) is staged once per workgroup in LDS as fp32.
//
// Work decomposition: a *unit* is the pair of rows whose results one epilogue needs together
// (RoPE partners i / i+hd/2; gate row i / up row i; two adjacent rows otherwise).  One wave owns a unit:
// every lane streams 16-byte slices of both rows (coalesced 1 KiB per wave-load), accumulates in fp32,
// and a 64-lane butterfly finishes the two dot products.  Units are dealt round-robin to the
// gridDim.x*4 waves of the launch.
#pragma once
#include "common.h"

namespace tgx {

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1 };
enum { EPI_QKV_ROPE = 0, EPI_RESIDUAL = 1, EPI_SILU_MUL = 2, EPI_LOGITS = 3 };

struct GemvArgs {
  const bf16_t* W;        // [N][K] row-major (torch Linear layout)
  const bf16_t* bias;     // [N] or nullptr
  const float* x;         // [K] input activations (fp32 between ops)
  const bf16_t* norm_w;   // [K] RMSNorm weight (PRO_RMSNORM)
  float eps;
  int N, K;
  int units;              // number of row pairs
  // EPI_QKV_ROPE
  float* q_out;           // [heads*hd] fp32
  bf16_t* k_cache;        // this layer, this row: [kv_heads][max_ctx][hd]
  bf16_t* v_cache;
  const float* rope_cos;  // [max_ctx][hd/2] fp32
  const float* rope_sin;
  const int* pos;         // device-resident pastLength of this row
  int heads, kv_heads, hd, max_ctx;
  // EPI_RESIDUAL: out[n] += acc;  EPI_SILU_MUL: out[i] = silu(g) * u      (fp32)
  float* out;
  // EPI_LOGITS
  float* logits;          // [N] fp32 accumulators
  float* part_val;        // [gridDim.x] best logit of this workgroup
  int* part_idx;
};

template <int PRO>
__device__ __forceinline__ void stage_x(const GemvArgs& a, float* xs, float* scratch4) {
  const int nchunk = a.K >> 3;
  const f32x4* xg = reinterpret_cast<const f32x4*>(a.x);
  float inv = 1.f;
  if (PRO == PRO_RMSNORM) {
    float ss = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += blockDim.x) {
      const f32x4 v0 = xg[2 * c], v1 = xg[2 * c + 1];
#pragma unroll
      for (int j = 0; j < 4; j++) { ss = fmaf(v0[j], v0[j], ss); }
#pragma unroll
      for (int j = 0; j < 4; j++) { ss = fmaf(v1[j], v1[j], ss); }
    }
    ss = block_sum_256(ss, scratch4);
    inv = 1.0f / sqrtf(ss / (float)a.K + a.eps);
  }
  const u32x4* wg = reinterpret_cast<const u32x4*>(a.norm_w);
  for (int c = threadIdx.x; c < nchunk; c += blockDim.x) {
    f32x4 v0 = xg[2 * c], v1 = xg[2 * c + 1];
    if (PRO == PRO_RMSNORM) {
      const u32x4 w = wg[c];
      // HF LlamaRMSNorm order: weight * (x * rsqrt(var+eps)), all fp32
      v0[0] = bf16_lo(w[0]) * (v0[0] * inv); v0[1] = bf16_hi(w[0]) * (v0[1] * inv);
      v0[2] = bf16_lo(w[1]) * (v0[2] * inv); v0[3] = bf16_hi(w[1]) * (v0[3] * inv);
      v1[0] = bf16_lo(w[2]) * (v1[0] * inv); v1[1] = bf16_hi(w[2]) * (v1[1] * inv);
      v1[2] = bf16_lo(w[3]) * (v1[2] * inv); v1[3] = bf16_hi(w[3]) * (v1[3] * inv);
    }
    f32x4* dst = reinterpret_cast<f32x4*>(xs + (c << 3));
    dst[0] = v0;
    dst[1] = v1;
  }
  __syncthreads();
}

template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);                     // [K] fp32
  float* scratch = reinterpret_cast<float*>(smem + (size_t)a.K * 4);  // 16 floats

  stage_x<PRO>(a, xs, scratch);

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nchunk = a.K >> 3;
  const int total_waves = gridDim.x * 4;
  const u32x4* W4 = reinterpret_cast<const u32x4*>(a.W);
  const int half = a.hd >> 1;

  float best_val = -INFINITY;
  int best_idx = 0x7fffffff;

  for (int u = blockIdx.x * 4 + wv; u < a.units; u += total_waves) {
    int ra, rb;
    bool rb_valid = true;
    if (EPI == EPI_QKV_ROPE) {
      const int hh = u / half, p = u - hh * half;
      ra = hh * a.hd + p;
      rb = ra + half;
    } else if (EPI == EPI_SILU_MUL) {
      ra = u;
      rb = u + (a.N >> 1);
    } else {
      ra = 2 * u;
      rb = ra + 1;
      if (rb >= a.N) { rb = ra; rb_valid = false; }
    }
    const u32x4* wa = W4 + (size_t)ra * nchunk;
    const u32x4* wb = W4 + (size_t)rb * nchunk;
    float acc_a0 = 0.f, acc_b0 = 0.f, acc_a1 = 0.f, acc_b1 = 0.f;
    int base = 0;
    // main loop (wave-uniform trip count): 2 rows x 4 slices = 8 independent 16-byte loads in flight per lane
    for (; base + 256 <= nchunk; base += 256) {
      const int c = base + lane;
      u32x4 va0 = load_nt(wa + c), vb0 = load_nt(wb + c);
      u32x4 va1 = load_nt(wa + c + 64), vb1 = load_nt(wb + c + 64);
      u32x4 va2 = load_nt(wa + c + 128), vb2 = load_nt(wb + c + 128);
      u32x4 va3 = load_nt(wa + c + 192), vb3 = load_nt(wb + c + 192);
      const f32x4* x0 = reinterpret_cast<const f32x4*>(xs + ((c) << 3));
      const f32x4* x1 = reinterpret_cast<const f32x4*>(xs + ((c + 64) << 3));
      const f32x4* x2 = reinterpret_cast<const f32x4*>(xs + ((c + 128) << 3));
      const f32x4* x3 = reinterpret_cast<const f32x4*>(xs + ((c + 192) << 3));
      f32x4 p0 = x0[0], q0 = x0[1], p1 = x1[0], q1 = x1[1], p2 = x2[0], q2 = x2[1], p3 = x3[0], q3 = x3[1];
      acc_a0 = dot8(acc_a0, va0, p0, q0); acc_b0 = dot8(acc_b0, vb0, p0, q0);
      acc_a1 = dot8(acc_a1, va1, p1, q1); acc_b1 = dot8(acc_b1, vb1, p1, q1);
      acc_a0 = dot8(acc_a0, va2, p2, q2); acc_b0 = dot8(acc_b0, vb2, p2, q2);
      acc_a1 = dot8(acc_a1, va3, p3, q3); acc_b1 = dot8(acc_b1, vb3, p3, q3);
    }
    // tail: up to 4 more (possibly partial) slices; addresses clamped, x masked to zero beyond K
    for (; base < nchunk; base += 64) {
      const int c = base + lane;
      const bool ok = c < nchunk;
      const int cc = ok ? c : nchunk - 1;
      u32x4 va = load_nt(wa + cc), vb = load_nt(wb + cc);
      const f32x4* xp = reinterpret_cast<const f32x4*>(xs + (cc << 3));
      f32x4 p = xp[0], q = xp[1];
      if (!ok) { p = f32x4{0.f, 0.f, 0.f, 0.f}; q = p; }
      acc_a0 = dot8(acc_a0, va, p, q);
      acc_b0 = dot8(acc_b0, vb, p, q);
    }
    float sa = group_sum<64>(acc_a0 + acc_a1);
    float sb = group_sum<64>(acc_b0 + acc_b1);

    if (EPI == EPI_LOGITS) {
      if (lane == 0) {
        a.logits[ra] = sa;
        if (sa > best_val) { best_val = sa; best_idx = ra; }     // rows ascend within a wave: '>' keeps the first
        if (rb_valid) {
          a.logits[rb] = sb;
          if (sb > best_val) { best_val = sb; best_idx = rb; }
        }
      }
      continue;
    }
    if (lane != 0) continue;
    if (a.bias) { sa += bf16_to_f32(a.bias[ra]); sb += bf16_to_f32(a.bias[rb]); }
    if (EPI == EPI_QKV_ROPE) {
      const int hh = u / half, p = u - hh * half;
      const int pos = *a.pos;
      if (hh < a.heads + a.kv_heads) {   // q or k head: rotate-half RoPE at absolute position pos
        const float cs = a.rope_cos[(size_t)pos * half + p], sn = a.rope_sin[(size_t)pos * half + p];
        const float na = sa * cs - sb * sn;
        const float nb = sb * cs + sa * sn;
        sa = na; sb = nb;
      }
      if (hh < a.heads) {
        a.q_out[hh * a.hd + p] = sa;
        a.q_out[hh * a.hd + p + half] = sb;
      } else {   // KVCacheManager::append: this position's K / V row, stored in bf16
        bf16_t* dst = (hh < a.heads + a.kv_heads)
                          ? a.k_cache + ((size_t)(hh - a.heads) * a.max_ctx + pos) * a.hd
                          : a.v_cache + ((size_t)(hh - a.heads - a.kv_heads) * a.max_ctx + pos) * a.hd;
        dst[p] = f32_to_bf16(sa);
        dst[p + half] = f32_to_bf16(sb);
      }
    } else if (EPI == EPI_RESIDUAL) {
      a.out[ra] += sa;
      if (rb_valid) a.out[rb] += sb;
    } else if (EPI == EPI_SILU_MUL) {
      a.out[u] = (sa / (1.0f + expf(-sa))) * sb;
    }
  }

  if (EPI == EPI_LOGITS) {
    // workgroup argmax, ties -> lowest index (== argmax(logits, -1), Sampler.cpp:28)
    float* sv = scratch;
    int* si = reinterpret_cast<int*>(scratch + 4);
    __syncthreads();
    if (lane == 0) { sv[wv] = best_val; si[wv] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float bv = sv[0]; int bi = si[0];
      for (int w = 1; w < 4; w++)
        if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
      a.part_val[blockIdx.x] = bv;
      a.part_idx[blockIdx.x] = bi;
    }
  }
}

// ---- greedy finalize: reduce the lm_head partial argmaxes, publish the token, advance the row ---------
// == argmax (Sampler.cpp:28) + tokens = concat(tokens, next) + KV pastLength += 1, and it gathers the next
// step's embedding row (nn::Embedding, GPTModel.h:52) into the residual stream so the decode graph needs
// no host input between steps.
// nn::Embedding row gather: bf16 table row -> fp32 residual stream
__device__ __forceinline__ void gather_embedding(const bf16_t* row, float* x, int H) {
  const u32x4* src = reinterpret_cast<const u32x4*>(row);
  f32x4* dst = reinterpret_cast<f32x4*>(x);
  for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x) {
    const u32x4 v = src[c];
    dst[2 * c] = f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
    dst[2 * c + 1] = f32x4{bf16_lo(v[2]), bf16_hi(v[2]), bf16_lo(v[3]), bf16_hi(v[3])};
  }
}

struct FinalizeArgs {
  const float* part_val;
  const int* part_idx;
  int n_part;
  int* tok;              // this row's current token (device resident)
  int* pos;              // this row's pastLength
  int* step;             // decode steps finalized so far (monotonic; index into the token rings)
  int* tok_log;          // [log_cap][rows] device ring of produced tokens
  volatile int* host_ring;   // [ring_cap][rows] pinned host mirror (AsyncTokenPipeline read-back)
  int log_cap, ring_cap;
  int row, rows;
  int log;               // 1: record the token in the rings (decode steps); 0: tgx_sample after a prefill
  int bump_step;         // 1 on the last row of a step
  const bf16_t* embed;   // [V][H]
  float* x;              // [H] residual stream of this row (fp32)
  int H;
  int advance_pos;       // 1: pos += 1 (the token just consumed is now in the cache)
};

__global__ __launch_bounds__(256) void finalize_greedy_kernel(const FinalizeArgs a) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int s_tok;
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < a.n_part; i += 256) {
    const float v = a.part_val[i]; const int ix = a.part_idx[i];
    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
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
  if (threadIdx.x == 0) {
    const int t = si[0];
    s_tok = t;
    *a.tok = t;
    if (a.advance_pos) *a.pos = *a.pos + 1;
    if (a.log) {
      const int st = *a.step;
      a.tok_log[(st % a.log_cap) * a.rows + a.row] = t;
      a.host_ring[(st % a.ring_cap) * a.rows + a.row] = t;
      if (a.bump_step) *a.step = st + 1;
    }
  }
  __syncthreads();
  gather_embedding(a.embed + (size_t)s_tok * a.H, a.x, a.H);
}

// Prefill-by-steps helper: x <- embed[prompt[pos - pos0]] (nn::Embedding on one prompt position).
struct EmbedArgs {
  const long long* ids;  // [S] this row's prompt on the device
  const int* pos;
  int pos0;
  const bf16_t* embed;
  float* x;
  int H, V;
  int* tok;
};
__global__ __launch_bounds__(256) void embed_prompt_kernel(const EmbedArgs a) {
  const int i = *a.pos - a.pos0;
  long long t = a.ids[i];
  if (threadIdx.x == 0) *a.tok = (int)t;
  gather_embedding(a.embed + (size_t)t * a.H, a.x, a.H);
}

__global__ void advance_pos_kernel(int* pos) { if (threadIdx.x == 0) *pos = *pos + 1; }

}  // namespace tgx
