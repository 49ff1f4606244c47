// attn_decode.h — single-query GQA attention over the pre-allocated KV cache (decode step).
//
// Replaces, for seq == 1:  KVCacheManager::append's concat (CacheManager.h:24-42 — the O(T) realloc+copy
// is gone: the QKV epilogue already stored this step's K/V row in place) and
// function::flashAttention(q, Kall, Vall, isCausal=false) (Attention.h:108-109).
//
// Numerics contract: K/V are in the storage dtype (bf16 / fp16 / fp32) in the cache; scores = (q.k) * hd^-1/2, softmax and P.V are fp32; fp32 output.
//
// Roofline: HBM — 2 * kv_heads * (T+1) * hd * 2 bytes per launch (K and V rows read once; all G = heads/kv_heads
// query heads of a group share one pass over their kv head).
//
// Decomposition (split-K flash-decode): grid = kv_heads * nsplit workgroups of 4 waves.  Workgroup (kvh, sp)
// owns the token range [sp*chunk, (sp+1)*chunk) with chunk derived on the device from the device-resident
// position (so one captured hipGraph serves every step).  Inside a wave, HD/8 adjacent lanes hold one
// token's row as 16-byte slices (a wave-load covers 64/(HD/8) consecutive tokens = 1 KiB contiguous);
// every lane group keeps its own online-softmax stream, merged once at the end (lanes -> waves via LDS ->
// one (m, l, o[HD]) partial per query head per split).  attn_combine_kernel merges the splits.
#pragma once
#include "common.h"

namespace tgx {

struct AttnArgs {
  const float* q;         // [heads][hd] fp32 (RoPE applied)
  const void* k_cache;    // this layer/row: [kv_heads][max_ctx][hd], storage dtype (kernel template DT)
  const void* v_cache;
  const int* pos;         // pastLength BEFORE this step; keys [0, pos] are attended
  float* part;            // [heads][nsplit][hd + 4]  (o[hd], m, l, pad) — 16-byte aligned records
  float* out;             // [heads*hd] fp32 (combine kernel)
  int heads, kv_heads, max_ctx, nsplit;
  float scale;
  // batch rows: blockIdx.y = row; row r uses q + r*q_stride, caches + r*kv_stride, pos[r], part + r*part_stride, out + r*q_stride
  long long q_stride, kv_stride, part_stride;
};

template <int DT, int HD, int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs a) {
  typedef elem_t<DT> E;
  constexpr int LPT = HD / 8;         // lanes per token row
  constexpr int TPW = 64 / LPT;       // tokens per wave-load
  constexpr int NSTREAM = 4 * TPW;    // independent online-softmax streams per workgroup (wave x token slot)
  constexpr int UNR = 4;              // wave-loads of K and of V in flight per iteration
  constexpr float LOG2E = 1.4426950408889634f;
  // per stream and query head: o[HD], m, l  (m in the exp2 domain)
  __shared__ __attribute__((aligned(16))) float red[NSTREAM][G][HD + 4];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* q_row = a.q + blockIdx.y * a.q_stride;
  const E* k_row = static_cast<const E*>(a.k_cache) + blockIdx.y * a.kv_stride;
  const E* v_row = static_cast<const E*>(a.v_cache) + blockIdx.y * a.kv_stride;
  float* part_row = a.part + blockIdx.y * a.part_stride;
  const int kvh = blockIdx.x / a.nsplit, sp = blockIdx.x - kvh * a.nsplit;
  const int part_i = lane % LPT, slot = lane / LPT;
  // Token blocks of STEP = 4 waves x UNR wave-loads are dealt round-robin to the splits: split sp owns blocks sp,
  // sp + nsplit, ...: the active splits are the first ceil(n_keys / STEP) of every kv head and each runs whole blocks;
  // the addresses of a split's first block do not depend on the context length.
  constexpr int STEP = 4 * TPW * UNR;
  const E* kbase = k_row + (size_t)kvh * a.max_ctx * HD + part_i * 8;
  const E* vbase = v_row + (size_t)kvh * a.max_ctx * HD + part_i * 8;
  int t0 = sp * STEP + wv * TPW * UNR;
  Slice8<DT> kv[UNR], vv[UNR];
#pragma unroll
  for (int r = 0; r < UNR; r++) {
    const int tc = min(t0 + r * TPW + slot, a.max_ctx - 1);
    kv[r] = load_slice<DT>(kbase + (size_t)tc * HD, 0);
    vv[r] = load_slice<DT>(vbase + (size_t)tc * HD, 0);
  }
  float qf[G][8];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const f32x4* qp = reinterpret_cast<const f32x4*>(q_row + (size_t)(kvh * G + g) * HD + part_i * 8);
    const f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
    for (int j = 0; j < 4; j++) { qf[g][j] = q0[j]; qf[g][4 + j] = q1[j]; }
  }
  const int n_keys = a.pos[blockIdx.y] + 1;
  const float qscale = a.scale * LOG2E;     // softmax in base 2: exp(x) = exp2(x * log2 e)

  if (sp * STEP >= n_keys) {   // this split has no keys at the current context length (workgroup-uniform): publish "empty"
    // (the compiler sinks the loads above below this branch; running empty splits through the masked path instead keeps
    //  them ahead of the position load but measured 1262 vs 1260 tok/s at context 2.3k and 1267 vs 1289 at 300 — rejected)
    for (int g = threadIdx.x; g < G; g += 256) {
      float* dst = part_row + ((size_t)(kvh * G + g) * a.nsplit + sp) * (HD + 4);
      dst[HD] = -INFINITY; dst[HD + 1] = 0.f;
    }
    return;
  }

#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int j = 0; j < 8; j++) qf[g][j] *= qscale;
  float m[G], l[G], o[G][8];
#pragma unroll
  for (int g = 0; g < G; g++) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) o[g][j] = 0.f;
  }

  while (true) {
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const bool valid = t0 + r * TPW + slot < n_keys;
      float kf[8], vf[8];
      slice_unpack<DT>(kv[r], kf);
      slice_unpack<DT>(vv[r], vf);
#pragma unroll
      for (int g = 0; g < G; g++) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s = fmaf(qf[g][j], kf[j], s);
        s = row_group_sum<LPT>(s);
        if (valid) {
          const float mn = fmaxf(m[g], s);
          const float alpha = exp2f(m[g] - mn);     // exp2(-inf) = 0 on the first key
          const float p = exp2f(s - mn);
          l[g] = l[g] * alpha + p;
#pragma unroll
          for (int j = 0; j < 8; j++) o[g][j] = o[g][j] * alpha + p * vf[j];
          m[g] = mn;
        }
      }
    }
    t0 += a.nsplit * STEP;            // this split's next block (contexts beyond nsplit*STEP tokens)
    if (t0 >= n_keys) break;          // wave-uniform
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const int tc = min(t0 + r * TPW + slot, n_keys - 1);
      kv[r] = load_slice<DT>(kbase + (size_t)tc * HD, 0);
      vv[r] = load_slice<DT>(vbase + (size_t)tc * HD, 0);
    }
  }

  // every (wave, slot) stream parks its state in LDS; the per-stream rescale factors are computed once
  // (NSTREAM*G threads, one exp2 each) and the 256 threads then only multiply-add
  __shared__ float sm_scale[NSTREAM][G];
  __shared__ float sm_M[G];
  const int stream = wv * TPW + slot;
#pragma unroll
  for (int g = 0; g < G; g++) {
    f32x4* dst = reinterpret_cast<f32x4*>(&red[stream][g][part_i * 8]);
    dst[0] = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
    dst[1] = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
    if (part_i == 0) { red[stream][g][HD] = m[g]; red[stream][g][HD + 1] = l[g]; }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NSTREAM * G; idx += 256) {
    const int g = idx / NSTREAM, st = idx - g * NSTREAM;
    float M = -INFINITY;
#pragma unroll 8
    for (int s2 = 0; s2 < NSTREAM; s2++) M = fmaxf(M, red[s2][g][HD]);
    const float ms = red[st][g][HD];
    sm_scale[st][g] = (ms == -INFINITY) ? 0.f : exp2f(ms - M);
    if (st == 0) sm_M[g] = M;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * HD; idx += 256) {
    const int g = idx / HD, d = idx - g * HD;
    float L = 0.f, acc = 0.f;
#pragma unroll 8
    for (int s2 = 0; s2 < NSTREAM; s2++) {
      const float sc = sm_scale[s2][g];
      acc = fmaf(red[s2][g][d], sc, acc);
      if (d == 0) L = fmaf(red[s2][g][HD + 1], sc, L);
    }
    float* dst = part_row + ((size_t)(kvh * G + g) * a.nsplit + sp) * (HD + 4);
    dst[d] = acc;
    if (d == 0) { dst[HD] = sm_M[g]; dst[HD + 1] = L; }
  }
}

// Merges the nsplit partials of every query head and writes the attention output (fp32)
// (== the reshape to [B,S,qDim] that feeds o_proj, Attention.h:111).
// One workgroup per query head; thread (s, dg) owns 8 dims of one split so that all partials are fetched
// with one round of independent 16-byte loads; splits are then summed through LDS in a fixed order.
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnArgs a) {
  constexpr int DG = HD / 8;            // dim groups of 8
  constexpr int SPB = 256 / DG;         // splits handled per pass (32 for hd 64, 16 for hd 128)
  __shared__ float sm_m[32], sm_l[32];
  __shared__ float sm_o[SPB][HD + 4];
  const int h = blockIdx.x, tid = threadIdx.x;
  const float* part_row = a.part + blockIdx.y * a.part_stride;
  float* out_row = a.out + blockIdx.y * a.q_stride;
  const float* p = part_row + (size_t)h * a.nsplit * (HD + 4);
  __shared__ float sm_e[32];
  const int dg = tid % DG, sl = tid / DG;
  constexpr int NPASS = 32 / SPB;        // 1 (hd 64) or 2 (hd 128) passes cover the 32 possible splits
  // all loads of the kernel are issued up front: the (m, l) scalars and this thread's data slices of every split
  f32x4 d0[NPASS], d1[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) {
    const int s = ps * SPB + sl;
    d0[ps] = f32x4{0.f, 0.f, 0.f, 0.f}; d1[ps] = d0[ps];
    if (s < a.nsplit) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p + (size_t)s * (HD + 4) + dg * 8);
      d0[ps] = src[0]; d1[ps] = src[1];
    }
  }
  if (tid < a.nsplit) { sm_m[tid] = p[tid * (HD + 4) + HD]; sm_l[tid] = p[tid * (HD + 4) + HD + 1]; }
  __syncthreads();
  if (tid < a.nsplit) {   // one exp2 per split; empty splits (m = -inf) get weight 0 (their stale data is multiplied away)
    float M0 = -INFINITY;
    for (int s = 0; s < a.nsplit; s++) M0 = fmaxf(M0, sm_m[s]);
    sm_e[tid] = (sm_m[tid] == -INFINITY) ? 0.f : exp2f(sm_m[tid] - M0);   // m is in the exp2 domain
  }
  __syncthreads();
  float L = 0.f;
  for (int s = 0; s < a.nsplit; s++) L = fmaf(sm_l[s], sm_e[s], L);
  float acc = 0.f;                       // threads < HD accumulate dim `tid`
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) {
    const int s = ps * SPB + sl;
    const bool live = s < a.nsplit && sm_e[s] != 0.f;
    const float sc = live ? sm_e[s] : 0.f;
    // empty splits hold stale (possibly non-finite) data: select, do not multiply
    const f32x4 v0 = live ? d0[ps] * sc : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 v1 = live ? d1[ps] * sc : f32x4{0.f, 0.f, 0.f, 0.f};
    float* dst = &sm_o[sl][dg * 8];
    dst[0] = v0[0]; dst[1] = v0[1]; dst[2] = v0[2]; dst[3] = v0[3];
    dst[4] = v1[0]; dst[5] = v1[1]; dst[6] = v1[2]; dst[7] = v1[3];
    __syncthreads();
    if (tid < HD) {
#pragma unroll 8
      for (int k = 0; k < SPB; k++) acc += sm_o[k][tid];
    }
    __syncthreads();
  }
  if (tid < HD) out_row[h * HD + tid] = acc / L;
}

}  // namespace tgx
