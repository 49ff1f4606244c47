// sampler.h — Sampler::sample (src/engine/Sampler.cpp:23-79) for one row of fp32 logits, one 1024-thread
// workgroup per batch row, without the reference's full-vocabulary sort.
//
//   reference (TinyTorch ops)                                  here
//   l = logits / temperature              (:34-36)             scale while loading
//   topk -> fill(-inf) -> scatter         (:39-45)             radix select of the k-th largest key (4 x 8-bit levels,
//                                                              integer LDS histograms); ties at the k-th value keep the
//                                                              lowest indices (order: value desc, index asc)
//   sort desc -> softmax -> cumsum ->     (:48-65)             the kept set is a PREFIX of that order, so only its end is
//   keep cum <= topP or first -> scatter                       needed: radix descent over probability-mass histograms
//                                                              (fixed-point u64 LDS atomics: order-independent, hence
//                                                              deterministic), ties resolved by index
//   softmax -> max -> mask p < minP*max   (:68-74)             one pass (p_i = e_i * inv, same expression as the oracle)
//   softmax -> multinomial(probs, 1)      (:77-78)             inverse CDF in index order with a counter-based
//                                                              splitmix64 draw (the reference's RNG stream is unpinnable)
// The kernel also performs the duties of finalize_greedy_kernel (token publish, pastLength+1, token rings, next
// embedding row) so a sampled decode step has the same launch count as a greedy one.
//
// HBM/L2 traffic: ~8 passes over V fp32 logits (512 KB for V = 128k, L2-resident); not bandwidth relevant.
#pragma once
#include "common.h"
#include "gemv.h"

namespace tgx {

struct SampleArgs {
  const float* logits;    // [V] fp32
  float* work;            // [V] scratch: filtered logits (-inf = removed)
  float* probs_out;       // [V] optional: final probabilities (tests); may be nullptr
  int V;
  float temperature; long long top_k; float top_p; float min_p;
  const unsigned long long* seed;   // device word
  FinalizeArgs fin;       // token publish / rings / embedding (part_* unused)
};

constexpr int SAMPLER_THREADS = 1024;

__device__ __forceinline__ unsigned int float_key(float f) {   // ascending order-preserving key
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float block_max_1024(float v, float* sh) {
  v = group_max<64>(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 16; i++) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {   // fixed order: deterministic
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < 16; i++) r += sh[i];
  return r;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Keeps, among the elements whose key equals `key`, the first `keep` in index order; removes the others.
// Elements with key > `key` stay, elements with key < `key` are removed.  Index order needs an ordered count:
// each thread owns a contiguous index range, counts its ties, and an exclusive scan over threads ranks them.
__device__ void apply_cut(float* work, int V, unsigned int key, long long keep, int* sh_cnt) {
  const int per = (V + SAMPLER_THREADS - 1) / SAMPLER_THREADS;
  const int i0 = threadIdx.x * per, i1 = min(V, i0 + per);
  int mine = 0;
  for (int i = i0; i < i1; i++) mine += (float_key(work[i]) == key);
  __syncthreads();
  sh_cnt[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {   // 1024-element exclusive scan; executed once per filter
    int run = 0;
    for (int t = 0; t < SAMPLER_THREADS; t++) { const int c = sh_cnt[t]; sh_cnt[t] = run; run += c; }
  }
  __syncthreads();
  long long rank = sh_cnt[threadIdx.x];
  for (int i = i0; i < i1; i++) {
    const unsigned int k = float_key(work[i]);
    if (k < key) work[i] = -INFINITY;
    else if (k == key) { if (rank >= keep) work[i] = -INFINITY; rank++; }
  }
  __syncthreads();
}

template <int DT>
__global__ __launch_bounds__(SAMPLER_THREADS) void sample_kernel(const SampleArgs a) {
  __shared__ float sh_f[16];
  __shared__ unsigned int hist_cnt[256];
  __shared__ unsigned long long hist_mass[256];
  __shared__ int sh_cnt[SAMPLER_THREADS];
  __shared__ double sh_d[SAMPLER_THREADS];
  __shared__ unsigned int s_sel;
  __shared__ long long s_need;
  __shared__ unsigned long long s_acc;
  __shared__ int s_tok;

  const int V = a.V, tid = threadIdx.x;
  const bool setT = a.temperature > 0.f, setK = a.top_k > 0, setP = a.top_p < 1.f, setM = a.min_p > 0.f;

  // l = logits / T
  float mx = -INFINITY;
  for (int i = tid; i < V; i += SAMPLER_THREADS) {
    float l = a.logits[i];
    if (setT) l = l / a.temperature;
    a.work[i] = l;
    mx = fmaxf(mx, l);
  }
  mx = block_max_1024(mx, sh_f);   // the maximum survives every filter (the first element is always kept)

  // ---- top-k: radix select of the k-th largest key ---------------------------------------------------------
  if (setK) {
    long long need = a.top_k < V ? a.top_k : V;     // how many still to keep inside the current prefix range
    unsigned int prefix = 0;
    for (int level = 0; level < 4; level++) {
      const int shift = 24 - 8 * level;
      if (tid < 256) hist_cnt[tid] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += SAMPLER_THREADS) {
        const unsigned int k = float_key(a.work[i]);
        if ((unsigned int)((unsigned long long)k >> (shift + 8)) == prefix) atomicAdd(&hist_cnt[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        long long n = need; int b = 255;
        for (; b > 0; b--) { if ((long long)hist_cnt[b] >= n) break; n -= hist_cnt[b]; }
        s_sel = (unsigned int)b; s_need = n;
      }
      __syncthreads();
      prefix = (prefix << 8) | s_sel;
      need = s_need;
      __syncthreads();
    }
    apply_cut(a.work, V, prefix, need, sh_cnt);    // `need` ties at the k-th key survive, lowest indices first
  }

  // ---- top-p: end of the kept prefix of the (value desc, index asc) order ---------------------------------
  if (setP) {
    float z = 0.f;
    for (int i = tid; i < V; i += SAMPLER_THREADS) z += expf(a.work[i] - mx);
    z = block_sum_1024(z, sh_f);
    const float inv = 1.0f / z;
    const double FIX = 1099511627776.0;   // 2^40 fixed point
    const unsigned long long p_fix = (unsigned long long)((double)a.top_p * FIX);
    unsigned int prefix = 0;
    unsigned long long acc = 0;           // mass of everything strictly above the current prefix range
    bool all_kept = false;
    for (int level = 0; level < 4 && !all_kept; level++) {
      const int shift = 24 - 8 * level;
      if (tid < 256) { hist_mass[tid] = 0; hist_cnt[tid] = 0; }
      __syncthreads();
      for (int i = tid; i < V; i += SAMPLER_THREADS) {
        const float l = a.work[i];
        const unsigned int k = float_key(l);
        if ((unsigned int)((unsigned long long)k >> (shift + 8)) == prefix) {
          const unsigned long long m = (unsigned long long)((double)(expf(l - mx) * inv) * FIX);
          atomicAdd(&hist_mass[(k >> shift) & 255u], m);
          atomicAdd(&hist_cnt[(k >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long c = acc; int b = 255; bool found = false;
        for (; b >= 0; b--) {
          if (hist_cnt[b] == 0) continue;
          if (c + hist_mass[b] > p_fix) { found = true; break; }
          c += hist_mass[b];
        }
        s_sel = found ? (unsigned int)b : 0xFFFFFFFFu; s_acc = c;
      }
      __syncthreads();
      if (s_sel == 0xFFFFFFFFu) all_kept = true;      // cumulative mass never exceeds top_p inside this range
      else { prefix = (prefix << 8) | s_sel; acc = s_acc; }
      __syncthreads();
    }
    if (!all_kept) {
      // boundary key = prefix: every tie has the same mass; keep floor((P - acc) / mass) of them, lowest index first
      const float lb = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix);
      const unsigned long long m1 = (unsigned long long)((double)(expf(lb - mx) * inv) * FIX);
      long long keep = (m1 > 0 && p_fix > acc) ? (long long)((p_fix - acc) / m1) : 0;
      if (acc == 0 && keep == 0) keep = 1;             // the first token is always kept (Sampler.cpp:54-57)
      apply_cut(a.work, V, prefix, keep, sh_cnt);
    }
  }

  // ---- min-p ---------------------------------------------------------------------------------------------------
  if (setM) {
    float z = 0.f;
    for (int i = tid; i < V; i += SAMPLER_THREADS) z += expf(a.work[i] - mx);
    z = block_sum_1024(z, sh_f);
    const float inv = 1.0f / z;
    const float thr = (1.0f * inv) * a.min_p;          // max prob = exp(0) * inv
    for (int i = tid; i < V; i += SAMPLER_THREADS)
      if (expf(a.work[i] - mx) * inv < thr) a.work[i] = -INFINITY;
    __syncthreads();
  }

  // ---- softmax -> multinomial (inverse CDF in index order) ------------------------------------------------------
  float z = 0.f;
  for (int i = tid; i < V; i += SAMPLER_THREADS) z += expf(a.work[i] - mx);
  z = block_sum_1024(z, sh_f);
  const float inv = 1.0f / z;
  const int per = (V + SAMPLER_THREADS - 1) / SAMPLER_THREADS;
  const int i0 = tid * per, i1 = min(V, i0 + per);
  double mine = 0.0;
  for (int i = i0; i < i1; i++) {
    const float p = expf(a.work[i] - mx) * inv;
    if (a.probs_out) a.probs_out[i] = p;
    mine += (double)p;
  }
  sh_d[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    unsigned long long s = (*a.seed) * 0x9E3779B97F4A7C15ull + (unsigned long long)(*a.fin.pos + a.fin.advance_pos) * 0xD1342543DE82EF95ull +
                           (unsigned long long)a.fin.row;
    const double u = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
    double run = 0.0; int t = 0;
    for (; t < SAMPLER_THREADS - 1; t++) { if (u < run + sh_d[t]) break; run += sh_d[t]; }
    // sequential walk inside thread t's index range (and beyond, if rounding left u above the total)
    int pick = -1; double cum = run;
    for (int i = t * per; i < V; i++) {
      const float p = expf(a.work[i] - mx) * inv;
      if (p > 0.f) { cum += (double)p; pick = i; if (u < cum) break; }
    }
    if (pick < 0) for (int i = 0; i < V; i++) if (a.work[i] != -INFINITY) pick = i;   // u beyond the total: last kept
    if ((unsigned)pick >= (unsigned)a.V) pick = 0;   // all-NaN logits: stay inside the embedding table
    s_tok = pick;
    *a.fin.tok = pick;
    if (a.fin.advance_pos) *a.fin.pos = *a.fin.pos + 1;
    if (a.fin.log) {
      const int st = *a.fin.step;
      a.fin.tok_log[(st % a.fin.log_cap) * a.fin.rows + a.fin.row] = pick;
      a.fin.host_ring[(st % a.fin.ring_cap) * a.fin.rows + a.fin.row] = pick;
      if (a.fin.bump_step) *a.fin.step = st + 1;
    }
  }
  __syncthreads();
  gather_embedding<DT>(a.fin.embed, s_tok, a.fin.x, a.fin.H);
}

}  // namespace tgx
