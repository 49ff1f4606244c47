// gemv.h — the weight-streaming kernel of the decode path: y = W[N,K] · x (bf16 / fp16 / fp32 weights, fp32
// accumulate), with the ops the reference issues around each nn::Linear fused in as prologue/epilogue.
//
//   reference op sequence (per decode step)                         fused form here
//   RMSNorm -> MergedLinear qkv -> split -> RoPE(q),RoPE(k)          PRO_RMSNORM + EPI_QKV_ROPE
//     -> KVCacheManager::append (Attention.h:94-106)                   (K/V land directly in the cache slot)
//   Linear o_proj -> x + .  (Attention.h:90, DecoderLayer.h:40)      PRO_PLAIN   + EPI_RESIDUAL
//     ... behind split-form attention at batch 1                     kernels/oproj_sliced.h (K-sliced, merges the attention splits itself)
//   RMSNorm -> MergedLinear gate_up -> siluMul (GatedMLP.h:37-39)    PRO_RMSNORM + EPI_SILU_MUL
//   Linear down_proj -> x + .  (GatedMLP.h:40, DecoderLayer.h:41)    PRO_PLAIN   + EPI_RESIDUAL
//   RMSNorm -> Linear lm_head -> argmax (GPTModel.h:56-57,            PRO_RMSNORM + EPI_LOGITS
//     Sampler.cpp:28)
//
// Roofline: HBM.  Algorithmic bytes per launch = 2*N*K (each weight byte is read exactly once, with the
// non-temporal policy).
//
// Work decomposition: a *unit* is the pair of rows whose results one epilogue needs together
// (RoPE partners i / i+hd/2; gate row i / up row i; two adjacent rows otherwise).  KS (1, 2 or 4) waves
// share one unit, each owning a contiguous K range of <= NX*512 elements; a wave keeps ITS slice of the
// activation vector in registers for the whole launch (no LDS staging, no workgroup barrier on the way to
// the first weight load), streams 16-byte weight slices (coalesced 1 KiB per wave-load, 2*NX loads in flight
// per lane, next unit prefetched while the current one is reduced), accumulates in fp32 and finishes each
// dot product with a DPP wave reduction.  With KS > 1 the KS partial sums meet in LDS in a fixed order.
// Norm-fused launches with KS > 1 (the qkv launch: 4 waves per row pair shorten every wave's load -> reduce chain; batch rows at
// hidden sizes whose R x NX slices would not fit one wave) exchange their partial sums of squares through LDS once.
#pragma once
#include "common.h"

namespace tgx {

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1, PRO_LAYERNORM = 2 };   // LAYERNORM: GPT-2's nn::LayerNorm with bias (ModelGPT2.h:120-135)
enum { EPI_QKV_ROPE = 0, EPI_RESIDUAL = 1, EPI_SILU_MUL = 2, EPI_LOGITS = 3, EPI_GELU = 4 };   // GELU: GPT-2's c_fc -> gelu (ModelGPT2.h:96-107)

// ---- greedy finalize: reduce the lm_head partial argmaxes, publish the token, advance the row ---------
// == argmax (Sampler.cpp:28) + tokens = concat(tokens, next) + KV pastLength += 1, and it gathers the next
// step's embedding row (nn::Embedding, GPTModel.h:52) into the residual stream so the decode graph needs
// no host input between steps.
// nn::Embedding row gather: table row `t` (storage dtype) -> fp32 residual stream
// GPT-2 adds the learned position row: wte(ids) + wpe(arange(past, past + S))  (ModelGPT2.h:165-169)
template <int DT>
__device__ __forceinline__ void gather_embedding(const void* table, long long t, float* x, int H, const void* wpe = nullptr, int p = 0) {
  const elem_t<DT>* row = static_cast<const elem_t<DT>*>(table) + (size_t)t * H;
  const elem_t<DT>* prow = static_cast<const elem_t<DT>*>(wpe) + (size_t)p * H;
  f32x4* dst = reinterpret_cast<f32x4*>(x);
  for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x) {
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

struct FinalizeArgs {
  const float* part_val;
  const int* part_idx;
  int n_part;
  int* tok;              // this row's current token (device resident)
  int* pos;              // this row's pastLength
  int* step;             // decode steps finalized so far (monotonic; index into the token rings)
  int* tok_log;          // [log_cap][rows] device ring of produced tokens
  volatile int* host_ring;   // [ring_cap][rows] pinned host mirror (AsyncTokenPipeline read-back); nullptr: not mirrored (multi-step graphs)
  int log_cap, ring_cap;
  int row, rows;
  int log;               // 1: record the token in the rings (decode steps); 0: tgx_sample after a prefill
  int bump_step;         // 1 on the last row of a step (batch 1, or rows finalized by consecutive launches)
  int* done;             // batches: rows of a step run concurrently and must all read the same step value — each counts itself here after reading it, and the
  int done_total;        // row that completes the count (of the whole batch) resets it and moves the step counter.  nullptr: bump_step decides
  const void* embed;     // [V][H], storage dtype
  float* x;              // [H] residual stream of this row (fp32)
  int H, V;
  int advance_pos;       // 1: pos += 1 (the token just consumed is now in the cache)
  const void* wpe;       // GPT-2: [n_pos][H] learned positions (nullptr otherwise); the next token sits at the advanced pos
  int n_pos;
};

// the body of the greedy finalize for one row, run by all 256 threads of a workgroup: finalize_greedy_kernel (tgx_sample after a
// prefill, sampled steps' pick kernel) and the lm_head launch's last-arriving workgroup (greedy decode steps) share it
template <int DT>
__device__ __forceinline__ void finalize_row(const FinalizeArgs& a) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int s_tok, s_pos;
  __syncthreads();                 // the shared arrays may still be read by the previous row's call
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
    const int t = (unsigned)si[0] < (unsigned)a.V ? si[0] : 0;   // all-NaN logits leave the sentinel index: never gather out of the table
    s_tok = t;
    *a.tok = t;
    const int np = *a.pos + (a.advance_pos ? 1 : 0);
    if (a.advance_pos) *a.pos = np;
    s_pos = np < a.n_pos ? np : a.n_pos - 1;    // a full context takes no further step: stay inside wpe
    if (a.log) {
      const int st = *a.step;
      a.tok_log[(st % a.log_cap) * a.rows + a.row] = t;
      if (a.host_ring) a.host_ring[(st % a.ring_cap) * a.rows + a.row] = t;
      if (a.done) { if (atomicAdd(a.done, 1) == a.done_total - 1) { *a.done = 0; *a.step = st + 1; } }
      else if (a.bump_step) *a.step = st + 1;
    }
  }
  __syncthreads();
  gather_embedding<DT>(a.embed, s_tok, a.x, a.H, a.wpe, a.wpe ? s_pos : 0);
}


// Batch rows: R rows (1, 2 or 4) of independent sequences share ONE pass over the weights — every weight slice that
// lands in a register is multiplied into R activation vectors (the reference runs the whole batch through each
// nn::Linear as a [B,1,K] x [K,N] product).  Per-row buffers are slabs with a constant row stride.
struct GemvArgs {
  const void* W;          // [N][K] row-major (torch Linear layout), elements of the storage dtype (kernel template DT)
  const void* bias;       // [N] or nullptr
  const float* x;         // [R][x_stride] input activations (fp32 between ops)
  const void* norm_w;     // [K] RMSNorm / LayerNorm weight
  const void* norm_b;     // [K] LayerNorm bias (PRO_LAYERNORM)
  float eps;
  int N, K;
  int ldw;                // elements between weight rows (== K unless the launch covers a K range of wider rows; 0 = K)
  int units;              // number of row pairs
  int ks;                 // waves per unit (1, 2, 4)
  int act16;              // option act.round16: the Linear's input (after its norm) is rounded to the storage dtype — the reference's bf16 modules see bf16 tensors (ModelLlama.h:62)
  int dbg;                // experiments only ("debug.gemv"): 1 skip the norm arithmetic, 2 skip the weight stream, 4 exit at once, 8 skip the epilogue
  long long x_stride, out_stride, q_stride, kv_stride, logits_stride, part_stride, kraw_stride;   // elements between batch rows
  // EPI_QKV_ROPE
  float* q_out;           // [R][heads*hd] fp32
  void* k_cache;          // this layer: [R][...kv_stride...]: [kv_heads][max_ctx][hd], storage dtype
  void* v_cache;
  const float* rope_cos;  // [max_ctx][hd/2] fp32
  const float* rope_sin;
  const int* pos;         // [R] device-resident pastLength of each row
  int heads, kv_heads, hd, max_ctx;
  const int* blk_tbl;     // paged KV (common.h kv_paged_off): [R][tbl_stride] block tables of the rows, k_cache / v_cache = this layer's pools — or nullptr
  long long tbl_stride;   // entries between batch rows (0: the rows are positions of ONE sequence)
  int raw_qk;             // Qwen3 (q/k RMSNorm before RoPE): emit un-rotated q and k, qk_norm_rope_kernel finishes them
  float* k_raw;           // [R][kv_heads*hd] fp32 staging for k when raw_qk
  // EPI_RESIDUAL: out[n] += acc;  EPI_SILU_MUL: out[i] = silu(g) * u;  EPI_GELU: out[n] = gelu_new(acc)      (fp32)
  float* out;
  // XACC (batch 1 behind kernels/oproj_sliced.h): the residual stream between o_proj and down lives in fixed-point accumulators —
  // PRO_RMSNORM reads its input from x_acc (x = fp32(acc)); EPI_RESIDUAL adds its product to fp32(res_acc), stores out and zeroes res_acc
  const long long* x_acc;   // [K]
  long long* res_acc;       // [N]
  // EPI_LOGITS
  float* logits;          // [R][N] fp32
  float* part_val;        // [R][gridDim.x] best logit of this workgroup
  int* part_idx;
};

// HF "gelu_new" (GPT-2's activation_function): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_new(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

template <int EPI>
__device__ __forceinline__ void unit_rows(const GemvArgs& a, int u, int& ra, int& rb, bool& rb_valid) {
  rb_valid = true;
  if (EPI == EPI_QKV_ROPE) {
    const int half = a.hd >> 1;
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
}

// XACC: see GemvArgs.x_acc / res_acc (R == 1 only)
// (four workgroups per CU — 128 registers — wherever the plain form is within a few registers of it: the lm_head form came out at 130)
template <int DT, int PRO, int EPI, int NX, int R, bool XACC = false>
__global__ __launch_bounds__(256, (NX * R <= 4 && DT != DT_F32) ? 4 : 1) void gemv_kernel(const GemvArgs a) {
  typedef elem_t<DT> E;
  static_assert(!XACC || R == 1, "fixed-point residual: batch 1");
  if (TGX_DBG(a, 4)) return;
  const int n_wg = (int)gridDim.x;
  // double-buffer the weight registers when the activations leave room; with 4 rows also at up to 4 slices per lane (~250 VGPRs, two
  // waves per SIMD): B = 4 Llama-3.2-3B 1577 -> 1621 tok/s, Mistral-7B 821 -> 845, 1B unchanged; two rows at 3 slices lose 5 % with it
  constexpr bool PIPE = NX * R <= 4 || (R == 4 && NX <= 4);
  __shared__ float ps[4][2 * R];
  __shared__ float sv[R][4];
  __shared__ int si[R][4];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int KS = a.ks, UPB = 4 / KS;
  const int slot = wv / KS, kpart = wv - slot * KS;
  const int nchunk = a.K >> 3;                                         // 16-byte weight slices per row
  const int per = ((nchunk + KS * 64 - 1) / (KS * 64)) * 64;           // slices per k-part (multiple of 64)
  const int c_begin = min(kpart * per, nchunk), c_end = min(c_begin + per, nchunk);
  const E* W = static_cast<const E*>(a.W);
  const int stride = n_wg * UPB;

  int cidx[NX];
  bool cok[NX];
#pragma unroll
  for (int j = 0; j < NX; j++) {
    const int c = c_begin + lane + 64 * j;
    cok[j] = c < c_end;
    cidx[j] = cok[j] ? c : max(c_end - 1, 0);    // clamped: always a legal slice of the row
  }

  Slice8<DT> wa[NX], wb[NX], na[PIPE ? NX : 1], nb[PIPE ? NX : 1];
  auto load_unit = [&](int ub, Slice8<DT>* ta, Slice8<DT>* tb) {
    const int u = min(ub + slot, a.units - 1);
    int ra, rb; bool v;
    unit_rows<EPI>(a, u, ra, rb, v);
    const E* pa = W + (size_t)ra * a.ldw;
    const E* pb = W + (size_t)rb * a.ldw;
#pragma unroll
    for (int j = 0; j < NX; j++) { ta[j] = load_slice_nt<DT>(pa, cidx[j]); tb[j] = load_slice_nt<DT>(pb, cidx[j]); }
  };

  float xr[R][NX][8];
  auto load_x = [&]() {
#pragma unroll
    for (int r = 0; r < R; r++) {
      const f32x4* xg = reinterpret_cast<const f32x4*>(a.x + (size_t)r * a.x_stride);
#pragma unroll
      for (int j = 0; j < NX; j++) {
        f32x4 v0 = xg[2 * cidx[j]], v1 = xg[2 * cidx[j] + 1];
        if (!cok[j]) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
#pragma unroll
        for (int t = 0; t < 4; t++) { xr[r][j][t] = v0[t]; xr[r][j][4 + t] = v1[t]; }
      }
    }
  };
  // 0. XACC: this thread's eight accumulators of the residual stream leave FIRST — loads return in order, so behind the weight tile their conversion,
  //    the LDS hand-over and its barrier would all sit between the tile's arrival and the first FMA (gate_up 12.8 -> 13.4 us); ahead of it they are done
  //    by the time the tile lands
  ulonglong2 xacc0[(XACC && PRO == PRO_RMSNORM) ? 4 : 1];
  if constexpr (XACC && PRO == PRO_RMSNORM) {
    const ulonglong2* ag = reinterpret_cast<const ulonglong2*>(a.x_acc + (size_t)min((int)threadIdx.x, nchunk - 1) * 8);
#pragma unroll
    for (int q = 0; q < 4; q++) xacc0[q] = ag[q];
    __builtin_amdgcn_sched_barrier(0);
  }
  // 1. the first unit's weights are in flight before anything else is touched
  int ub = blockIdx.x * UPB;
  if (ub < a.units && !(TGX_DBG(a, 2))) load_unit(ub, wa, wb);
  // (XACC: the norm weights too — behind the hand-over's barrier they would be one more memory round trip)
  Slice8<DT> nw_x[(XACC && PRO == PRO_RMSNORM) ? NX : 1];
  if constexpr (XACC && PRO == PRO_RMSNORM) {
#pragma unroll
    for (int j = 0; j < NX; j++) nw_x[j] = load_slice<DT>(static_cast<const E*>(a.norm_w), cidx[j]);
    __builtin_amdgcn_sched_barrier(0);
  }

  // 2. this wave's slice of every row's activation vector -> registers (zero outside the range)
  if constexpr (XACC && PRO == PRO_RMSNORM) {
    // x = fp32(acc), converted ONCE per workgroup and handed to the waves through LDS (dynamic, K floats): every wave converting its own copy cost
    // 16 KB of accumulator reads and 32 conversions per lane ahead of the first FMA (+2.2 us on the gate_up launch, tools/probes/layer_lab.hip)
    extern __shared__ __attribute__((aligned(16))) float xs[];
    for (int cs = threadIdx.x; cs < nchunk; cs += 256) {
      ulonglong2 t[4];
      if (cs == (int)threadIdx.x) {
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = xacc0[q];
      } else {       // hidden sizes beyond 2048: the later chunks
        const ulonglong2* ag = reinterpret_cast<const ulonglong2*>(a.x_acc + (size_t)cs * 8);
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = ag[q];
      }
      float f[8];
#pragma unroll
      for (int q = 0; q < 4; q++) { f[2 * q] = fix_to_f32((long long)t[q].x); f[2 * q + 1] = fix_to_f32((long long)t[q].y); }
      f32x4* dst = reinterpret_cast<f32x4*>(xs + (size_t)cs * 8);
      dst[0] = f32x4{f[0], f[1], f[2], f[3]}; dst[1] = f32x4{f[4], f[5], f[6], f[7]};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NX; j++) {
      const f32x4* xl = reinterpret_cast<const f32x4*>(xs + (size_t)cidx[j] * 8);
      f32x4 v0 = xl[0], v1 = xl[1];
      if (!cok[j]) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
#pragma unroll
      for (int t = 0; t < 4; t++) { xr[0][j][t] = v0[t]; xr[0][j][4 + t] = v1[t]; }
    }
  } else {
    load_x();          // (activation loads ahead of the weight tile measured no different: tools/probes/README.md round 4)
  }
  // sum of one value per batch row over the KS waves that share a unit, in wave order (every wave of the workgroup takes part)
  auto ks_sum = [&](float* v) {
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; r++) ps[wv][r] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
      float t = ps[slot * KS][r];
      for (int k = 1; k < KS; k++) t += ps[slot * KS + k][r];
      v[r] = t;
    }
    __syncthreads();               // ps is reused (second statistic, unit loop)
  };
  if (PRO == PRO_LAYERNORM) {   // torch LayerNorm: ((x - mean) * rsqrt(var + eps)) * weight + bias, biased variance, two passes
    const E* wg = static_cast<const E*>(a.norm_w);
    const E* bg = static_cast<const E*>(a.norm_b);
    Slice8<DT> nw[NX], nbias[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { nw[j] = load_slice<DT>(wg, cidx[j]); nbias[j] = load_slice<DT>(bg, cidx[j]); }
    float st[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < NX; j++)
#pragma unroll
        for (int t = 0; t < 8; t++) sm += xr[r][j][t];       // lanes outside the range hold zeros
      st[r] = wave_sum(sm);
    }
    if (KS > 1) ks_sum(st);
    float mean[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      mean[r] = st[r] / (float)a.K;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < NX; j++)
#pragma unroll
        for (int t = 0; t < 8; t++) { const float dlt = cok[j] ? xr[r][j][t] - mean[r] : 0.f; sq = fmaf(dlt, dlt, sq); }
      st[r] = wave_sum(sq);
    }
    if (KS > 1) ks_sum(st);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float inv = 1.0f / sqrtf(st[r] / (float)a.K + a.eps);
#pragma unroll
      for (int j = 0; j < NX; j++) {
        float w[8], bb[8];
        slice_unpack<DT>(nw[j], w);
        slice_unpack<DT>(nbias[j], bb);
#pragma unroll
        for (int t = 0; t < 8; t++)
          xr[r][j][t] = cok[j] ? __fadd_rn(__fmul_rn(__fmul_rn(xr[r][j][t] - mean[r], inv), w[t]), bb[t]) : 0.f;
      }
    }
  }
  if (PRO == PRO_RMSNORM && !(TGX_DBG(a, 1))) {   // HF order: weight * (x * rsqrt(mean(x^2)+eps)).  KS == 1: the wave holds all of x;
    // KS > 1 (batch rows at hidden sizes whose R x NX slices would not fit one wave): the KS waves of a unit exchange their
    // partial sums of squares through LDS once, before the weight loop (every wave of the workgroup takes part)
    const E* wg = static_cast<const E*>(a.norm_w);
    Slice8<DT> nw[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { if constexpr (XACC) nw[j] = nw_x[j]; else nw[j] = load_slice<DT>(wg, cidx[j]); }       // in flight together with x
    float ssq[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NX; j++)
#pragma unroll
        for (int t = 0; t < 8; t++) ss = fmaf(xr[r][j][t], xr[r][j][t], ss);
      ssq[r] = wave_sum(ss);
    }
    if (KS > 1) ks_sum(ssq);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float inv = 1.0f / sqrtf(ssq[r] / (float)a.K + a.eps);
#pragma unroll
      for (int j = 0; j < NX; j++) {
        float w[8];
        slice_unpack<DT>(nw[j], w);
#pragma unroll
        for (int t = 0; t < 8; t++) xr[r][j][t] = w[t] * (xr[r][j][t] * inv);
      }
    }
  }

  // option act.round16: a Linear's input is rounded to the storage dtype.  Norm-fused inputs are rounded here, after their norm; the inputs of the PLAIN products
  // (attention output, siluMul / gelu output) are rounded by the kernel that writes them (below, kernels/attn_decode.h): nothing but loads stands between a
  // PLAIN product's launch and its weight stream
  if constexpr (DT != DT_F32 && PRO != PRO_PLAIN) {
    if (a.act16) {          // (kernel-uniform)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int j = 0; j < NX; j++)
#pragma unroll
          for (int t = 0; t < 8; t++) xr[r][j][t] = elem_to_f32<DT>(f32_to_elem<DT>(xr[r][j][t]));
    }
  }

  float best_val[R];
  int best_idx[R];
  int pos[R];
#pragma unroll
  for (int r = 0; r < R; r++) { best_val[r] = -INFINITY; best_idx[r] = 0x7fffffff; pos[r] = (EPI == EPI_QKV_ROPE) ? a.pos[r] : 0; }

  for (; ub < a.units; ub += stride) {   // trip count uniform per workgroup
    const bool has_next = ub + stride < a.units;
    if (PIPE && has_next && !(TGX_DBG(a, 2))) load_unit(ub + stride, na, nb);

    // epilogue operands are fetched now so that their latency hides under the dot products
    const int u = ub + slot;
    const bool writer = u < a.units && kpart == 0 && lane == 0 && !(TGX_DBG(a, 8));
    int ra = 0, rb = 0; bool rb_valid = false;
    float e0[R], e1[R];                // RESIDUAL: x[ra], x[rb];  QKV_ROPE: cos, sin
    int eblk[R];                       // QKV_ROPE, paged KV: the physical block of this position (fetched with cos / sin, under the dot products)
#pragma unroll
    for (int r = 0; r < R; r++) { e0[r] = 0.f; e1[r] = 0.f; eblk[r] = 0; }
    if (writer) {
      unit_rows<EPI>(a, u, ra, rb, rb_valid);
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (EPI == EPI_RESIDUAL) {
          if constexpr (XACC) { e0[r] = fix_to_f32(a.res_acc[ra]); e1[r] = fix_to_f32(a.res_acc[rb]); }
          else { const float* o = a.out + (size_t)r * a.out_stride; e0[r] = o[ra]; e1[r] = o[rb]; }
        }
        if (EPI == EPI_QKV_ROPE) {
          const int half = a.hd >> 1;
          const int p = u % half;
          e0[r] = a.rope_cos[(size_t)pos[r] * half + p]; e1[r] = a.rope_sin[(size_t)pos[r] * half + p];
          if (a.blk_tbl) eblk[r] = a.blk_tbl[(size_t)r * a.tbl_stride + (pos[r] >> KV_BLOCK_SHIFT)];
        }
      }
    }

    float sa[R], sb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      float acc_a0 = 0.f, acc_b0 = 0.f, acc_a1 = 0.f, acc_b1 = 0.f;
#pragma unroll
      for (int j = 0; j < NX; j++) {
        const f32x4 xa = f32x4{xr[r][j][0], xr[r][j][1], xr[r][j][2], xr[r][j][3]};
        const f32x4 xb = f32x4{xr[r][j][4], xr[r][j][5], xr[r][j][6], xr[r][j][7]};
        if (j & 1) { acc_a1 = dot8<DT>(acc_a1, wa[j], xa, xb); acc_b1 = dot8<DT>(acc_b1, wb[j], xa, xb); }
        else       { acc_a0 = dot8<DT>(acc_a0, wa[j], xa, xb); acc_b0 = dot8<DT>(acc_b0, wb[j], xa, xb); }
      }
      sa[r] = wave_sum(acc_a0 + acc_a1);
      sb[r] = wave_sum(acc_b0 + acc_b1);
    }

    if (KS > 1) {   // fixed-order sum of the KS k-part partials through LDS
      __syncthreads();             // previous iteration's readers are done
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) { ps[wv][2 * r] = sa[r]; ps[wv][2 * r + 1] = sb[r]; }
      }
      __syncthreads();
      if (kpart == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          sa[r] = ps[slot * KS][2 * r]; sb[r] = ps[slot * KS][2 * r + 1];
          for (int k = 1; k < KS; k++) { sa[r] += ps[slot * KS + k][2 * r]; sb[r] += ps[slot * KS + k][2 * r + 1]; }
        }
      }
    }

    if (writer) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        float va = sa[r], vb = sb[r];
        if (EPI == EPI_LOGITS) {
          float* lg = a.logits + (size_t)r * a.logits_stride;
          lg[ra] = va;
          if (va > best_val[r]) { best_val[r] = va; best_idx[r] = ra; }     // rows ascend within a wave: '>' keeps the first
          if (rb_valid) {
            lg[rb] = vb;
            if (vb > best_val[r]) { best_val[r] = vb; best_idx[r] = rb; }
          }
          continue;
        }
        if (a.bias) { const E* bias = static_cast<const E*>(a.bias); va += elem_to_f32<DT>(bias[ra]); vb += elem_to_f32<DT>(bias[rb]); }
        if (EPI == EPI_QKV_ROPE) {
          const int half = a.hd >> 1;
          const int hh = u / half, p = u - hh * half;
          const bool is_q = hh < a.heads, is_k = !is_q && hh < a.heads + a.kv_heads;
          if (a.raw_qk && is_k) {   // Qwen3: qk_norm_rope_kernel normalises, rotates and appends k
            float* kr = a.k_raw + (size_t)r * a.kraw_stride + (hh - a.heads) * a.hd;
            kr[p] = va; kr[p + half] = vb;
            continue;
          }
          if (!a.raw_qk && (is_q || is_k)) {   // rotate-half RoPE at absolute position pos
            const float cs = e0[r], sn = e1[r];
            rope_rotate_pair(va, vb, cs, sn);       // one spelling of the rotation in every RoPE + append site (common.h)
          }
          if (is_q) {
            float* q = a.q_out + (size_t)r * a.q_stride + hh * a.hd;
            q[p] = va; q[p + half] = vb;
          } else {   // KVCacheManager::append: this position's K / V row, rounded once into the storage dtype
            const int kh = is_k ? hh - a.heads : hh - a.heads - a.kv_heads;
            const size_t off = a.blk_tbl ? (((size_t)eblk[r] * a.kv_heads + kh) * KV_BLOCK + (pos[r] & (KV_BLOCK - 1))) * (size_t)a.hd
                                         : ((size_t)kh * a.max_ctx + pos[r]) * a.hd + (size_t)r * a.kv_stride;
            E* dst = (is_k ? static_cast<E*>(a.k_cache) : static_cast<E*>(a.v_cache)) + off;
            dst[p] = f32_to_elem<DT>(va);
            dst[p + half] = f32_to_elem<DT>(vb);
          }
        } else if (EPI == EPI_RESIDUAL) {
          float* o = a.out + (size_t)r * a.out_stride;
          o[ra] = e0[r] + va;
          if (rb_valid) o[rb] = e1[r] + vb;
          if constexpr (XACC) { a.res_acc[ra] = 0; if (rb_valid) a.res_acc[rb] = 0; }     // this lane is the only reader / writer of its rows' accumulators
        } else if (EPI == EPI_SILU_MUL) {
          a.out[(size_t)r * a.out_stride + u] = round_storage_if<DT>((va / (1.0f + expf(-va))) * vb, a.act16);        // (the down product's input)
        } else if (EPI == EPI_GELU) {
          float* o = a.out + (size_t)r * a.out_stride;
          o[ra] = round_storage_if<DT>(gelu_new(va), a.act16);                 // (c_proj's input)
          if (rb_valid) o[rb] = round_storage_if<DT>(gelu_new(vb), a.act16);
        }
      }
    }

    if (PIPE) {
#pragma unroll
      for (int j = 0; j < NX; j++) { wa[j] = na[j]; wb[j] = nb[j]; }
    } else if (has_next && !(TGX_DBG(a, 2))) {
      load_unit(ub + stride, wa, wb);
    }
  }

  if (EPI == EPI_LOGITS) {
    // workgroup argmax per batch row, ties -> lowest index (== argmax(logits, -1), Sampler.cpp:28)
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; r++) { sv[r][wv] = best_val[r]; si[r][wv] = best_idx[r]; }
    }
    __syncthreads();
    if (threadIdx.x < R) {
      const int r = threadIdx.x;
      float bv = sv[r][0]; int bi = si[r][0];
      for (int w = 1; w < 4; w++)
        if (sv[r][w] > bv || (sv[r][w] == bv && si[r][w] < bi)) { bv = sv[r][w]; bi = si[r][w]; }
      a.part_val[(size_t)r * a.part_stride + blockIdx.x] = bv;
      a.part_idx[(size_t)r * a.part_stride + blockIdx.x] = bi;
    }
  }
}

// ---- Qwen3: per-head RMSNorm on q and k, then RoPE, then the cache append (AttentionWithQKNorm::projectQKV,
// Attention.h:156-163 + Attention::forward :81-83,:106).  One wave per head; lane p owns the RoPE pair (p, p+hd/2).
struct QkNormArgs {
  float* q;               // [heads][hd] in/out
  const float* k_raw;     // [kv_heads][hd]
  void* k_cache;          // this layer/row: [kv_heads][max_ctx][hd], storage dtype
  const void *q_norm_w, *k_norm_w;     // [hd]
  const float *rope_cos, *rope_sin;
  const int* pos;         // [rows]
  int heads, kv_heads, hd, max_ctx;
  float eps;
  long long q_stride, kraw_stride, kv_stride;   // elements between batch rows (blockIdx.y); kv_stride 0 = the rows are positions of one sequence
  const int* blk_tbl;     // paged KV: [rows][tbl_stride] block tables (k_cache = the layer's pool), or nullptr
  long long tbl_stride;
};
template <int DT>
__global__ __launch_bounds__(64) void qk_norm_rope_kernel(QkNormArgs a) {
  typedef elem_t<DT> E;
  const int hh = blockIdx.x, p = threadIdx.x, half = a.hd >> 1;
  a.q += blockIdx.y * a.q_stride; a.k_raw += blockIdx.y * a.kraw_stride; a.pos += blockIdx.y;
  a.k_cache = static_cast<E*>(a.k_cache) + blockIdx.y * a.kv_stride;       // (paged: kv_stride is 0, the rows differ by their tables)
  const bool is_q = hh < a.heads;
  const float* src = is_q ? a.q + hh * a.hd : a.k_raw + (hh - a.heads) * a.hd;
  const E* w = static_cast<const E*>(is_q ? a.q_norm_w : a.k_norm_w);
  const bool act = p < half;
  float x0 = act ? src[p] : 0.f, x1 = act ? src[p + half] : 0.f;
  const float ss = wave_sum(x0 * x0 + x1 * x1);
  const float inv = 1.0f / sqrtf(ss / (float)a.hd + a.eps);
  if (!act) return;
  x0 = elem_to_f32<DT>(w[p]) * (x0 * inv);
  x1 = elem_to_f32<DT>(w[p + half]) * (x1 * inv);
  const int pos = *a.pos;
  const float cs = a.rope_cos[(size_t)pos * half + p], sn = a.rope_sin[(size_t)pos * half + p];
  float r0 = x0, r1 = x1;
  rope_rotate_pair(r0, r1, cs, sn);
  if (is_q) { a.q[hh * a.hd + p] = r0; a.q[hh * a.hd + p + half] = r1; }
  else {
    E* dst = static_cast<E*>(a.k_cache) + (a.blk_tbl ? kv_paged_off(a.blk_tbl + blockIdx.y * a.tbl_stride, a.kv_heads, hh - a.heads, pos, a.hd)
                                                     : ((size_t)(hh - a.heads) * a.max_ctx + pos) * a.hd);
    dst[p] = f32_to_elem<DT>(r0); dst[p + half] = f32_to_elem<DT>(r1);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void finalize_greedy_kernel(const FinalizeArgs a) { finalize_row<DT>(a); }

// The greedy finalize of every row of a decode batch in ONE launch (blockIdx.x = row; the batched-MFMA decode step): the step counter is
// advanced by the row that completes the batch's count (FinalizeArgs.done), because rows running concurrently must all read the same step value.
struct FinalizeRowsArgs {
  FinalizeArgs f;          // row 0's view
  long long part_stride, x_stride;
};
template <int DT>
__global__ __launch_bounds__(256) void finalize_rows_kernel(const FinalizeRowsArgs a) {
  FinalizeArgs f = a.f;
  const int r = blockIdx.x;
  f.part_val += (size_t)r * a.part_stride; f.part_idx += (size_t)r * a.part_stride;
  f.tok += r; f.pos += r; f.x += (size_t)r * a.x_stride; f.row = a.f.row + r;
  finalize_row<DT>(f);
}

// Prefill-by-steps: chunk row r <- embedding of prompt token r, position pos0 + r (one workgroup per chunk row).
struct EmbedChunkArgs {
  const long long* ids;  // the chunk's first prompt token on the device
  const void* embed;
  float* x;              // [R][H] chunk residual streams
  int* pos;              // [R] chunk positions
  int H, pos0;
  const void* wpe;       // GPT-2 learned positions or nullptr
};
template <int DT>
__global__ __launch_bounds__(256) void embed_chunk_kernel(const EmbedChunkArgs a) {
  const int r = blockIdx.x;
  if (threadIdx.x == 0) a.pos[r] = a.pos0 + r;
  gather_embedding<DT>(a.embed, a.ids[r], a.x + (size_t)r * a.H, a.H, a.wpe, a.pos0 + r);
}

static __global__ void advance_pos_kernel(int* pos) { if (threadIdx.x == 0) *pos = *pos + 1; }
static __global__ void add_pos_kernel(int* pos, int n) { if (threadIdx.x == 0) *pos = *pos + n; }
// Rebuilds the lm_head epilogue's per-workgroup argmax partials from a logits vector (tgx_set_logits: sampler tests).
static __global__ __launch_bounds__(256) void argmax_partials_kernel(const float* logits, int V, float* part_val, int* part_idx) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
    const float v = logits[i];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
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
  if (threadIdx.x == 0) { part_val[blockIdx.x] = sv[0]; part_idx[blockIdx.x] = si[0]; }
}

static __global__ void nop_kernel(int* w) { if (threadIdx.x == 999) *w = 0; }

}  // namespace tgx
