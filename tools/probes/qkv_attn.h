// qkv_attn.h — the QKV product and the split-form attention of a batch-1 decode step as ONE launch (round 6).
//
// Replaces two dependent launches of the step (reference op sequence, Attention.h:71-109):
//   RMSNorm -> MergedLinear qkv -> split -> RoPE(q), RoPE(k) -> KVCacheManager::append      (gemv_kernel<PRO_RMSNORM, EPI_QKV_ROPE>)
//   flashAttention(q, Kall, Vall) over keys [0, pastLength]                                  (attn_decode_kernel, split form)
//
// Why.  At the benchmark's operating point (Llama-3.2-1B, context ~2060) the qkv launch streams 12.6 MB in 5.4 us (HBM two-thirds idle) and the
// attention launch behind it needs 5.0 us for 4.2 MB: position -> K / V wave-loads -> q -> softmax -> LDS meeting, every arrow a memory round trip that
// starts only after the kernel boundary.  Here the attention workgroups ride in the qkv launch: they hold their K / V wave-loads in registers while the
// weight stream runs, and take q (and, in the one split that owns the newest key, this step's k / v rows) from the producers through data-tagged
// 8-byte granules {fp32 value, tag} — one agent-scope store per value, polled by one agent-scope load per lane, no flag, no fence, no atomic RMW
// (MI355X_MICROARCH.md "handoff-1to1").  The tag is a device-resident epoch that the NEXT launch of the layer (o_proj) advances, so a granule of an
// earlier layer or step never matches.
//
// Grid: [0, n_prod) producer workgroups (the GEMV body, never waiting on anything: dispatched first, so every consumer's wait is on workgroups that are
// resident or done — no residency assumption), then heads * nsplit consumer workgroups (one query head x one split of its kv head's keys, 4 waves).
// The K / V cache rows of this step are still stored by the producers (the next steps read them); this step's consumers never read that row from memory.
//
// Roofline: HBM — 2 * (q_dim + 2 kv_dim) * hidden bytes of weights + 2 * kv_heads * (T + 1) * hd * 2 bytes of K / V per launch.
//
// STATUS (round 6): measured, not adopted.  Bit-identical to the two launches; layer_lab (LAB_FUSE=1): the Llama-3.2-1B layer at context 2064 36.3 -> 35.8 us;
// in the model (it was wired in as option decode.fuse_qkv_attn for one A/B): 0.660 vs 0.661 ms per token, no gain at any producer count — the timeline
// (template TIMING) shows why: profiles/r06_decode.txt.  Lives here, outside the product build.
#pragma once
#include "kernels/attn_decode.h"
#include "kernels/gemv.h"

namespace tgx {

struct QkvAttnArgs {
  GemvArgs g;                    // the qkv product: PRO_RMSNORM, EPI_QKV_ROPE, one row; q_out unused
  AttnArgs a;                    // the split form: part records out; q unused
  unsigned long long* gran_q;    // [heads * hd]        granules {value, tag}
  unsigned long long* gran_kv;   // [2 * kv_heads * hd] this step's k row | v row (as the cache holds them: rounded to the storage dtype)
  const unsigned* epoch;         // the tag of this launch (advanced by the layer's next launch)
  int n_prod;                    // producer workgroups (a multiple of 8: consumers keep `id % 8` = their XCD)
  unsigned long long* stamps;    // lab only (template TIMING): [workgroup][8] s_memrealtime stamps (100 MHz)
};
#define QA_STAMP(i) do { if constexpr (TIMING) { if (threadIdx.x == 0) A.stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } } while (0)

__device__ __forceinline__ void granule_store(unsigned long long* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every lane polls ITS granule until the whole wave holds the current tag (wave-uniform exit)
__device__ __forceinline__ float granule_wait(const unsigned long long* p, unsigned tag) {
  unsigned long long gval = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (__builtin_amdgcn_ballot_w64((unsigned)(gval >> 32) == tag) != ~0ull) {
    __builtin_amdgcn_s_sleep(4);
    gval = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)gval);
}

// ---- producer: gemv_kernel<DT, PRO_RMSNORM, EPI_QKV_ROPE, NX, 1> with the q / k / v values leaving as granules --------------------------------
// The producers share the chip with the waiting consumers, so the bytes in flight come from depth, not from the number of resident workgroups (the
// stand-alone launch keeps 1024 workgroups x 2 units in flight): DEPTH units of weights per wave leave with the first instructions.  And the chain behind
// their arrival is ONE pass: the DEPTH dot-product pairs, their wave sums, one LDS meeting of the KS waves, then lane d of the unit's first wave finishes
// unit d (bias, RoPE with cos / sin fetched up front, granules, cache row) — not DEPTH serial {reduce, barrier, epilogue-load, store} rounds.
// COMMUTE: the RMSNorm row scale commutes with the product — (w o x r) W^T = r ((w o x) W^T) — so the dot products run on w o x as the weights land and
// the sum of squares travels with the partial sums to the one LDS meeting; the factor r meets the sums in the epilogue (no LDS round trip and no
// normalisation pass between the activation's arrival and the first FMA).  Not with act.round16 (the rounded quantity is the normalised one).
template <int DT, int NX, int DEPTH, bool TIMING, bool COMMUTE>
__device__ __forceinline__ void qkv_producer(const QkvAttnArgs& A, int bid, int n_wg, unsigned tag) {
  typedef elem_t<DT> E;
  const GemvArgs& a = A.g;
  __shared__ float ps[4][2 * DEPTH + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int KS = a.ks, UPB = 4 / KS;
  const int slot = wv / KS, kpart = wv - slot * KS;
  const int nchunk = a.K >> 3;
  const int per = ((nchunk + KS * 64 - 1) / (KS * 64)) * 64;
  const int c_begin = min(kpart * per, nchunk), c_end = min(c_begin + per, nchunk);
  const E* W = static_cast<const E*>(a.W);
  const int stride = n_wg * UPB;
  const int half = a.hd >> 1;
  int cidx[NX];
  bool cok[NX];
#pragma unroll
  for (int j = 0; j < NX; j++) {
    const int c = c_begin + lane + 64 * j;
    cok[j] = c < c_end;
    cidx[j] = cok[j] ? c : max(c_end - 1, 0);
  }
  Slice8<DT> wa[DEPTH][NX], wb[DEPTH][NX];
  auto load_unit = [&](int ub, Slice8<DT>* ta, Slice8<DT>* tb) {
    const int u = min(ub + slot, a.units - 1);
    int ra, rb; bool v;
    unit_rows<EPI_QKV_ROPE>(a, u, ra, rb, v);
    const E* pa = W + (size_t)ra * a.ldw;
    const E* pb = W + (size_t)rb * a.ldw;
#pragma unroll
    for (int j = 0; j < NX; j++) { ta[j] = load_slice_nt<DT>(pa, cidx[j]); tb[j] = load_slice_nt<DT>(pb, cidx[j]); }
  };
  // x and the norm weights leave FIRST: loads return in order, and behind DEPTH units of weights the activation would arrive when the whole stream has
  // (the norm, its LDS meeting and the normalisation would then all sit between the last weight byte and the first FMA)
  f32x4 xv0[NX], xv1[NX];
  Slice8<DT> nw[NX];
  {
    const f32x4* xg = reinterpret_cast<const f32x4*>(a.x);
    const E* wg = static_cast<const E*>(a.norm_w);
#pragma unroll
    for (int j = 0; j < NX; j++) { xv0[j] = xg[2 * cidx[j]]; xv1[j] = xg[2 * cidx[j] + 1]; nw[j] = load_slice<DT>(wg, cidx[j]); }
  }
  __builtin_amdgcn_sched_barrier(0);
  int ub = bid * UPB;
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    if (ub + d * stride < a.units) load_unit(ub + d * stride, wa[d], wb[d]);      // (workgroup-uniform)
  __builtin_amdgcn_sched_barrier(0);
  float xr[NX][8];
#pragma unroll
  for (int j = 0; j < NX; j++) {
    asm volatile("" : "+v"(xv0[j]), "+v"(xv1[j]));                                // wait for exactly these (the weights may still fly)
    f32x4 v0 = xv0[j], v1 = xv1[j];
    if (!cok[j]) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
#pragma unroll
    for (int t = 0; t < 4; t++) { xr[j][t] = v0[t]; xr[j][4 + t] = v1[t]; }
  }
  // the epilogue operands of lane d's unit (first batch): position -> cos / sin, bias — two dependent loads that run under the weight stream
  const int pos = a.pos[0];
  auto epi_unit = [&](int ub_) { return ub_ + min(lane, DEPTH - 1) * stride + slot; };
  float cs = 0.f, sn = 0.f, bia = 0.f, bib = 0.f;
  auto load_epi = [&](int ub_) {
    const int u = min(epi_unit(ub_), a.units - 1);
    const int p = u % half;
    cs = a.rope_cos[(size_t)pos * half + p]; sn = a.rope_sin[(size_t)pos * half + p];
    int ra, rb; bool rv;
    unit_rows<EPI_QKV_ROPE>(a, u, ra, rb, rv);
    const E* bias = static_cast<const E*>(a.bias ? a.bias : a.norm_w);          // no load under a branch (the join would drain the memory pipeline)
    const float ba = elem_to_f32<DT>(bias[a.bias ? ra : 0]), bb = elem_to_f32<DT>(bias[a.bias ? rb : 0]);
    bia = a.bias ? ba : 0.f; bib = a.bias ? bb : 0.f;
  };
  load_epi(ub);
  float ssq_wave = 0.f;
  {   // HF order: weight * (x * rsqrt(mean(x^2) + eps)); the KS waves of a unit exchange their partial sums of squares through LDS once
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NX; j++)
#pragma unroll
      for (int t = 0; t < 8; t++) ss = fmaf(xr[j][t], xr[j][t], ss);
    ssq_wave = wave_sum(ss);
    float ssq = ssq_wave;
    if (KS > 1 && !COMMUTE) {
      if (lane == 0) ps[wv][0] = ssq;
      __syncthreads();
      float t = ps[slot * KS][0];
      for (int k = 1; k < KS; k++) t += ps[slot * KS + k][0];
      ssq = t;
      __syncthreads();
    }
    const float inv = COMMUTE ? 1.0f : 1.0f / sqrtf(ssq / (float)a.K + a.eps);
#pragma unroll
    for (int j = 0; j < NX; j++) {
      float w[8];
      slice_unpack<DT>(nw[j], w);
#pragma unroll
      for (int t = 0; t < 8; t++) xr[j][t] = COMMUTE ? w[t] * xr[j][t] : w[t] * (xr[j][t] * inv);
    }
    if (a.act16) {
#pragma unroll
      for (int j = 0; j < NX; j++)
#pragma unroll
        for (int t = 0; t < 8; t++) xr[j][t] = elem_to_f32<DT>(f32_to_elem<DT>(xr[j][t]));
    }
  }
  QA_STAMP(1);                                                                    // x normalised
  for (; ub < a.units; ub += DEPTH * stride) {
    float sa[DEPTH], sb[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      float acc_a0 = 0.f, acc_b0 = 0.f, acc_a1 = 0.f, acc_b1 = 0.f;
#pragma unroll
      for (int j = 0; j < NX; j++) {
        const f32x4 xa = f32x4{xr[j][0], xr[j][1], xr[j][2], xr[j][3]};
        const f32x4 xb = f32x4{xr[j][4], xr[j][5], xr[j][6], xr[j][7]};
        if (j & 1) { acc_a1 = dot8<DT>(acc_a1, wa[d][j], xa, xb); acc_b1 = dot8<DT>(acc_b1, wb[d][j], xa, xb); }
        else       { acc_a0 = dot8<DT>(acc_a0, wa[d][j], xa, xb); acc_b0 = dot8<DT>(acc_b0, wb[d][j], xa, xb); }
      }
      sa[d] = wave_sum(acc_a0 + acc_a1); sb[d] = wave_sum(acc_b0 + acc_b1);      // (slots beyond the last unit hold a repeat of it: never written)
    }
    QA_STAMP(2);                                                                  // dot products done = weights landed
    // the next batch's weights leave before the meeting (contexts with more than DEPTH units per wave)
    const bool more = ub + DEPTH * stride < a.units;
    if (more) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
        if (ub + (DEPTH + d) * stride < a.units) load_unit(ub + (DEPTH + d) * stride, wa[d], wb[d]);
    }
    // fixed-order sum of the KS k-part partials; lane d of the unit's first wave owns unit d
    float va = 0.f, vb = 0.f;
    if (KS > 1) {
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { ps[wv][2 * d] = sa[d]; ps[wv][2 * d + 1] = sb[d]; }
        if constexpr (COMMUTE) ps[wv][2 * DEPTH] = ssq_wave;
      }
      __syncthreads();
      if (kpart == 0 && lane < DEPTH) {
        va = ps[slot * KS][2 * lane]; vb = ps[slot * KS][2 * lane + 1];
        for (int k = 1; k < KS; k++) { va += ps[slot * KS + k][2 * lane]; vb += ps[slot * KS + k][2 * lane + 1]; }
        if constexpr (COMMUTE) {
          float t = ps[slot * KS][2 * DEPTH];
          for (int k = 1; k < KS; k++) t += ps[slot * KS + k][2 * DEPTH];
          const float inv = 1.0f / sqrtf(t / (float)a.K + a.eps);
          va *= inv; vb *= inv;
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) if (lane == d) { va = sa[d]; vb = sb[d]; }
      if constexpr (COMMUTE) { const float inv = 1.0f / sqrtf(ssq_wave / (float)a.K + a.eps); va *= inv; vb *= inv; }
    }
    const int u = epi_unit(ub);
    if (kpart == 0 && lane < DEPTH && ub + lane * stride < a.units && u < a.units) {
      va += bia; vb += bib;
      const int hh = u / half, p = u - hh * half;
      const bool is_q = hh < a.heads, is_k = !is_q && hh < a.heads + a.kv_heads;
      if (is_q || is_k) rope_rotate_pair(va, vb, cs, sn);
      if (is_q) {
        unsigned long long* gq = A.gran_q + (size_t)hh * a.hd;
        granule_store(gq + p, va, tag); granule_store(gq + p + half, vb, tag);
      } else {       // KVCacheManager::append (for the steps to come) + the granules this step's attention takes
        const int kh = is_k ? hh - a.heads : hh - a.heads - a.kv_heads;
        E* dst = (is_k ? static_cast<E*>(a.k_cache) : static_cast<E*>(a.v_cache)) + ((size_t)kh * a.max_ctx + pos) * a.hd;
        const E ea = f32_to_elem<DT>(va), eb = f32_to_elem<DT>(vb);
        dst[p] = ea; dst[p + half] = eb;
        unsigned long long* gk = A.gran_kv + (size_t)(is_k ? 0 : a.kv_heads * a.hd) + (size_t)kh * a.hd;
        granule_store(gk + p, elem_to_f32<DT>(ea), tag); granule_store(gk + p + half, elem_to_f32<DT>(eb), tag);
      }
    }
    QA_STAMP(3);                                                                  // published
    if (more) load_epi(ub + DEPTH * stride);
  }
}

// ---- consumer: attn_decode_kernel<DT, HD, 1, 4> (split form), q and this step's k / v from granules ------------------------------------------
template <int DT, int HD, bool TIMING>
__device__ __forceinline__ void attn_consumer(const QkvAttnArgs& A, int cid, unsigned tag) {
  typedef elem_t<DT> E;
  static_assert(HD == 64, "one granule per lane: head_dim 64");
  const AttnArgs& a = A.a;
  constexpr int NW = 4, UNR = 4;
  constexpr int LPT = HD / 8, TPW = 64 / LPT, STEP = NW * TPW * UNR, S = NW * TPW;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) float qs[1][3][HD];           // q, k_new, v_new as they arrive (wave 0 polls)
  __shared__ __attribute__((aligned(16))) float so[S][HD];
  __shared__ float sm[S], sl[S], se[S], sML[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // split-major: the splits that hold keys at this context length (the first ceil(n_keys / STEP)) are dispatched first and wait side by side with the
  // producers; the empty ones come last and leave at once.  Within a split the kv head varies fastest: the gfull query heads of a kv head share
  // `id % 8`, i.e. an XCD and its L2 (the producers are a multiple of 8), so their common K / V block leaves HBM once.
  const int sp = cid / a.heads, hi = cid - sp * a.heads;
  const int zg = hi / a.kv_heads, kvh = hi - zg * a.kv_heads;
  const int head = kvh * a.gfull + zg;
  const int part_i = lane % LPT, slot = lane / LPT;
  const E* kbase = static_cast<const E*>(a.k_cache) + (size_t)kvh * a.max_ctx * HD + part_i * 8;
  const E* vbase = static_cast<const E*>(a.v_cache) + (size_t)kvh * a.max_ctx * HD + part_i * 8;
  int t0 = sp * STEP + wv * TPW * UNR;
  Slice8<DT> kv[UNR], vv[UNR];
#pragma unroll
  for (int r = 0; r < UNR; r++) {
    const int tc = min(t0 + r * TPW + slot, a.max_ctx - 1);
    kv[r] = load_slice<DT>(kbase + (size_t)tc * HD, 0);
    vv[r] = load_slice<DT>(vbase + (size_t)tc * HD, 0);
  }
  const int n_keys = a.pos[0] + 1;
  QA_STAMP(1);
  float* rec = a.part + ((size_t)head * a.nsplit + sp) * (HD + 4);
  if (sp * STEP >= n_keys) {          // no keys for this split at the current context length
    if (threadIdx.x == 0) { rec[HD] = -INFINITY; rec[HD + 1] = 0.f; }
    return;
  }
  // the newest key (position n_keys - 1) lives in block (n_keys - 1) / STEP, dealt to split ((n_keys - 1) / STEP) % nsplit
  const int newest = n_keys - 1;
  const bool own_new = ((newest / STEP) % a.nsplit) == sp;             // workgroup-uniform
  // the first block's K / V rows to fp32 while q is on its way (the loads landed long ago)
  float kf0[UNR][8], vf0[UNR][8];
#pragma unroll
  for (int r = 0; r < UNR; r++) { slice_unpack<DT>(kv[r], kf0[r]); slice_unpack<DT>(vv[r], vf0[r]); }
  // ---- the hand-over: q of this head (and k / v of this step), one granule per lane of wave 0 (one poller per workgroup: the others sleep at the barrier)
  if (wv == 0) {
    const float qv = granule_wait(A.gran_q + (size_t)head * HD + lane, tag);
    qs[0][0][lane] = qv;
    if (own_new) {
      qs[0][1][lane] = granule_wait(A.gran_kv + (size_t)kvh * HD + lane, tag);
      qs[0][2][lane] = granule_wait(A.gran_kv + (size_t)(a.kv_heads + kvh) * HD + lane, tag);
    }
  }
  __syncthreads();
  QA_STAMP(2);                                                                    // q arrived
  const float qscale = a.scale * LOG2E;
  float qf[8], knew[8], vnew[8];
  {
    const f32x4* qp = reinterpret_cast<const f32x4*>(&qs[0][0][part_i * 8]);
    const f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
    for (int j = 0; j < 4; j++) { qf[j] = q0[j] * qscale; qf[4 + j] = q1[j] * qscale; }
    if (own_new) {
      const f32x4* kp = reinterpret_cast<const f32x4*>(&qs[0][1][part_i * 8]);
      const f32x4* vp = reinterpret_cast<const f32x4*>(&qs[0][2][part_i * 8]);
      const f32x4 k0 = kp[0], k1 = kp[1], v0 = vp[0], v1 = vp[1];
#pragma unroll
      for (int j = 0; j < 4; j++) { knew[j] = k0[j]; knew[4 + j] = k1[j]; vnew[j] = v0[j]; vnew[4 + j] = v1[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) { knew[j] = 0.f; vnew[j] = 0.f; }
    }
  }
  float m = -INFINITY, l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool first = true;
  auto block = [&]() {
    if (t0 < n_keys) {
      float kf[UNR][8], vf[UNR][8];
      bool valid[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const int tok = t0 + r * TPW + slot;
        valid[r] = tok < n_keys;
        if (first) {
#pragma unroll
          for (int j = 0; j < 8; j++) { kf[r][j] = kf0[r][j]; vf[r][j] = vf0[r][j]; }
        } else {
          slice_unpack<DT>(kv[r], kf[r]);
          slice_unpack<DT>(vv[r], vf[r]);
        }
        if (tok == newest) {            // this step's own key / value: from the granules (the cache row is being stored by a producer of this launch)
#pragma unroll
          for (int j = 0; j < 8; j++) { kf[r][j] = knew[j]; vf[r][j] = vnew[j]; }
        }
      }
      float sc[UNR];
      float mb = -INFINITY;
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) t = fmaf(qf[j], kf[r][j], t);
        t = row_group_sum<LPT>(t);
        sc[r] = valid[r] ? t : -INFINITY;
        mb = fmaxf(mb, sc[r]);
      }
      if (mb != -INFINITY) {
        const float mn = fmaxf(m, mb);
        const float alpha = exp2f(m - mn);
        float lb = 0.f, ob[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const float p = exp2f(sc[r] - mn);
          lb += p;
#pragma unroll
          for (int j = 0; j < 8; j++) ob[j] = fmaf(p, vf[r][j], ob[j]);
        }
        l = fmaf(l, alpha, lb);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = fmaf(o[j], alpha, ob[j]);
        m = mn;
      }
    }
  };
  while (true) {
    block();
    first = false;
    t0 += a.nsplit * STEP;
    if (t0 >= n_keys) break;
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const int tc = min(t0 + r * TPW + slot, n_keys - 1);
      kv[r] = load_slice<DT>(kbase + (size_t)tc * HD, 0);
      vv[r] = load_slice<DT>(vbase + (size_t)tc * HD, 0);
    }
  }
  QA_STAMP(3);                                                                    // keys done (K / V landed)
  // the NW x TPW token-slot streams of the workgroup meet once in LDS (attn_decode_kernel, NW == 4 form)
  const int st = wv * TPW + slot;
  {
    f32x4* dst = reinterpret_cast<f32x4*>(&so[st][part_i * 8]);
    dst[0] = f32x4{o[0], o[1], o[2], o[3]};
    dst[1] = f32x4{o[4], o[5], o[6], o[7]};
    if (part_i == 0) { sm[st] = m; sl[st] = l; }
  }
  __syncthreads();
  if (threadIdx.x < S) {
    const int i = threadIdx.x;
    const float mi = sm[i], li = sl[i];
    float M = mi;
    M = fmaxf(M, dpp_mov<0xB1, 0xf>(M)); M = fmaxf(M, dpp_mov<0x4E, 0xf>(M)); M = fmaxf(M, dpp_mov<0x141, 0xf>(M)); M = fmaxf(M, dpp_mov<0x140, 0xf>(M));
    M = fmaxf(M, __shfl_xor(M, 16, 64));
    const float e = (mi == -INFINITY) ? 0.f : exp2f(mi - M);
    float L = li * e;
    L = row_group_sum<16>(L);
    L += __shfl_xor(L, 16, 64);
    se[i] = e;
    if (i == 0) { sML[0] = M; sML[1] = L; }
  }
  __syncthreads();
  if (threadIdx.x < HD) {
    const int d = threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < S; i++) acc = fmaf(so[i][d], se[i], acc);
    rec[d] = acc;
    if (d == 0) { rec[HD] = sML[0]; rec[HD + 1] = sML[1]; }
  }
}

template <int DT, int HD, int NX, int DEPTH = 4, bool TIMING = false, bool COMMUTE = false>
__global__ __launch_bounds__(256, 4) void qkv_attn_kernel(const QkvAttnArgs A) {
  QA_STAMP(0);
  const unsigned tag = *A.epoch;
  if ((int)blockIdx.x < A.n_prod) qkv_producer<DT, NX, DEPTH, TIMING, COMMUTE>(A, (int)blockIdx.x, A.n_prod, tag);
  else attn_consumer<DT, HD, TIMING>(A, (int)blockIdx.x - A.n_prod, tag);
  QA_STAMP(7);
}

static __global__ void bump_epoch_kernel(unsigned* e) { if (threadIdx.x == 0) *e = *e + 1; }

}  // namespace tgx
