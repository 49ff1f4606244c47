// sampler.h — Sampler::sample (src/engine/Sampler.cpp:23-79) on fp32 logits without the reference's full-vocabulary
// sort, spread over the whole GPU: ceil(V / 1024) workgroups per batch row, every thread owns 4 consecutive vocabulary
// entries in registers, and the stages of the filter chain are separate launches of one captured graph (a kernel
// boundary is the cheapest grid-wide barrier on this part).
//
//   reference (TinyTorch ops)                                  here
//   l = logits / temperature              (:34-36)             v = l / T while loading (every stage recomputes it)
//   topk -> fill(-inf) -> scatter         (:39-45)             the kept set of every filter is a PREFIX of the order
//   sort desc -> softmax -> cumsum ->     (:48-65)             (value descending, index ascending), so each filter is one
//   keep cum <= topP or first -> scatter                       threshold on the 49-bit composite key (value key, ~index):
//                                                              radix descent over 5 digit levels (11/11/10 value bits,
//                                                              then index bits), one launch per level — top-k descends
//                                                              over COUNT histograms, top-p over probability-MASS
//                                                              histograms in 2^-40 fixed point (integer atomics are
//                                                              order-independent, hence deterministic); the index digits
//                                                              make ties unique, so no separate tie handling exists
//   softmax -> max -> mask p < minP*max   (:68-74)             p_i = e_i * inv with the oracle's expressions; Z from a
//                                                              per-workgroup partial-sum stage
//   softmax -> multinomial(probs, 1)      (:77-78)             inverse CDF in index order (double accumulation) with a
//                                                              counter-based splitmix64 draw (the reference's RNG stream
//                                                              is unpinnable): per-workgroup sums, then one workgroup
//                                                              walks the selected 1024 entries
// The pick kernel also performs the duties of finalize_greedy_kernel (token publish, pastLength+1, token rings, next
// embedding row).  Launches per sampled step: 5 per active top-k/top-p filter + 2 (+1 with min-p) + 1 pick per row;
// T = 0.8 / top-p 0.9 (the CLI defaults): 8 launches, +50 us per decode step for V = 128 256 (the one-workgroup version of
// this round added 600 us; profiles/r01_sampler_cost.txt).
#pragma once
#include "common.h"
#include "gemv.h"

namespace tgx {

constexpr int SAMP_WG = 256;          // threads per workgroup
constexpr int SAMP_EPT = 4;           // vocabulary entries per thread
constexpr int SAMP_TILE = SAMP_WG * SAMP_EPT;
constexpr int SAMP_BINS = 2048;       // 11-bit digits
constexpr int SAMP_LEVELS = 5;
constexpr int SAMP_MAX_WG = 1024;     // V <= 2^20

struct SampLevelState {
  unsigned long long prefix;          // digits chosen so far
  unsigned long long need;            // top-k: entries still to keep inside the prefix range; top-p: the mass threshold
  unsigned long long acc;             // top-p: mass strictly above the prefix range
  int done, pad;                      // top-p: cumulative mass never exceeds the threshold -> everything is kept
};

struct SampScratch {                  // one per batch row; zero at allocation, re-zeroed by the pick kernel
  unsigned int cnt[SAMP_LEVELS][SAMP_BINS];
  unsigned long long mass[SAMP_LEVELS][SAMP_BINS];
  SampLevelState st_k[SAMP_LEVELS], st_p[SAMP_LEVELS];
  unsigned long long thr_k, thr_p;    // kept <=> composite key >= thr
  double wg_z[SAMP_MAX_WG];           // per-workgroup sums of exp(v - max): the set min-p looks at.  The softmax normaliser is
  double wg_z2[SAMP_MAX_WG];          // ... the final kept set.       accumulated in double and rounded once (as the oracle
                                      //                                does): independent of the summation order
  double wg_p[SAMP_MAX_WG];           // per-workgroup sums of the final probabilities
};

struct SampArgs {
  const float* logits; long long logits_stride;       // [rows][V]
  const float* part_val; long long part_stride;       // lm_head per-workgroup maxima (their maximum is max(logits))
  int n_part;
  SampScratch* sc;                                    // [rows]
  float* probs_out; long long probs_stride;           // [rows][V] final probabilities (tgx_read_probs)
  int V, idx_bits, level;
  float temperature; long long top_k; float top_p; float min_p;
  int k_from_hist, p_from_hist;       // 1: this launch is the first after the filter's last level and derives the threshold
};

__device__ __forceinline__ unsigned int float_key(float f) {   // ascending order-preserving key
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ int samp_width(int level, int idx_bits) {
  const int i0 = idx_bits < 11 ? idx_bits : 11;
  return level == 0 ? 11 : level == 1 ? 11 : level == 2 ? 10 : level == 3 ? i0 : idx_bits - i0;
}
__device__ __forceinline__ int samp_shift(int level, int idx_bits) {   // bit position of the digit of `level`
  int s = 32 + idx_bits;
  for (int l = 0; l <= level; l++) s -= samp_width(l, idx_bits);
  return s;
}

// ---- block helpers (256 threads, fixed order) ---------------------------------------------------------------------
__device__ __forceinline__ float samp_block_max(float v, float* sh4) {
  v = group_max<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
}
__device__ __forceinline__ double samp_block_sum_d(double v, double* sh4) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}
// sum of the values held by threads with a HIGHER thread index (exclusive suffix sum) and the block total
__device__ __forceinline__ unsigned long long samp_suffix_excl(unsigned long long v, unsigned long long* sh4, unsigned long long& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_down(inc, off, 64);
    if (lane + off < 64) inc += t;
  }
  __syncthreads();
  if (lane == 0) sh4[w] = inc;
  __syncthreads();
  unsigned long long above = 0;
  for (int k = w + 1; k < 4; k++) above += sh4[k];
  total = sh4[0] + sh4[1] + sh4[2] + sh4[3];
  return inc - v + above;
}
// sum of the values held by threads with a LOWER thread index (exclusive prefix sum), double
__device__ __forceinline__ double samp_prefix_excl_d(double v, double* sh4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) sh4[w] = inc;
  __syncthreads();
  double below = 0.0;
  for (int k = 0; k < w; k++) below += sh4[k];
  return below + (inc - v);
}

// ---- per-thread view of its 4 vocabulary entries ---------------------------------------------------------------------
struct SampElems {
  float v[SAMP_EPT];                  // logit / T
  unsigned long long comp[SAMP_EPT];  // (value key << idx_bits) | (idx_mask - index): larger = earlier in (value desc, index asc)
  bool in[SAMP_EPT];                  // index < V
  int base;                           // first index
};

__device__ __forceinline__ void samp_load(const SampArgs& a, int row, int wg, SampElems& e) {
  const float* lg = a.logits + (size_t)row * a.logits_stride;
  e.base = (wg * SAMP_WG + (int)threadIdx.x) * SAMP_EPT;
  const unsigned long long idx_mask = (1ull << a.idx_bits) - 1ull;
  const bool setT = a.temperature > 0.f;
  float raw[SAMP_EPT];
  if (e.base + SAMP_EPT <= a.V && ((reinterpret_cast<size_t>(lg + e.base) & 15) == 0)) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(lg + e.base);
    raw[0] = q[0]; raw[1] = q[1]; raw[2] = q[2]; raw[3] = q[3];
  } else {
#pragma unroll
    for (int j = 0; j < SAMP_EPT; j++) raw[j] = e.base + j < a.V ? lg[e.base + j] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    e.in[j] = e.base + j < a.V;
    e.v[j] = setT ? raw[j] / a.temperature : raw[j];
    e.comp[j] = ((unsigned long long)float_key(e.v[j]) << a.idx_bits) | (idx_mask - (unsigned long long)(e.base + j));
  }
}

// max over the row of logit / T: the maximum of the lm_head epilogue's per-workgroup maxima, scaled (x -> x / T is monotone)
__device__ __forceinline__ float samp_row_max(const SampArgs& a, int row, float* sh4) {
  const float* pv = a.part_val + (size_t)row * a.part_stride;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < a.n_part; i += SAMP_WG) m = fmaxf(m, pv[i]);
  m = samp_block_max(m, sh4);
  return a.temperature > 0.f ? m / a.temperature : m;
}

__device__ __forceinline__ unsigned long long samp_mass(float v, float mx) {
  return (unsigned long long)((double)expf(v - mx) * 1099511627776.0);   // 2^40 fixed point, e <= 1
}

// ---- digit selection from the previous level's global histogram (every workgroup derives the same answer) -----------
// top-k: walking the bins from the top, the bin in which the cumulative count reaches `need`
__device__ SampLevelState samp_select_count(const unsigned int* hist, int nbins, int width, const SampLevelState& prev,
                                            unsigned long long* sh4, unsigned long long* sh_res) {
  constexpr int PER = SAMP_BINS / SAMP_WG;
  unsigned int own[PER];
  unsigned long long local = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { const int b = threadIdx.x * PER + j; own[j] = b < nbins ? hist[b] : 0u; local += own[j]; }
  unsigned long long total;
  const unsigned long long above = samp_suffix_excl(local, sh4, total);
  if (threadIdx.x == 0) { sh_res[0] = 0; sh_res[1] = prev.need > 0 ? 1 : 0; }   // fallback: lowest bin
  __syncthreads();
  if (above < prev.need && prev.need <= above + local) {
    unsigned long long n = prev.need - above;
    for (int j = PER - 1; j >= 0; j--) {
      if ((unsigned long long)own[j] >= n) { sh_res[0] = (unsigned long long)(threadIdx.x * PER + j); sh_res[1] = n; break; }
      n -= own[j];
    }
  }
  __syncthreads();
  SampLevelState st;
  st.prefix = (prev.prefix << width) | sh_res[0];
  st.need = sh_res[1]; st.acc = 0; st.done = 0; st.pad = 0;
  return st;
}
// top-p: walking from the top with c = mass above, the first bin with c + mass[b] > threshold; none -> everything kept
__device__ SampLevelState samp_select_mass(const unsigned long long* hist, int nbins, int width, const SampLevelState& prev, bool first_level,
                                           float top_p, unsigned long long* sh4, unsigned long long* sh_res) {
  constexpr int PER = SAMP_BINS / SAMP_WG;
  unsigned long long own[PER], local = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { const int b = threadIdx.x * PER + j; own[j] = b < nbins ? hist[b] : 0ull; local += own[j]; }
  unsigned long long total;
  const unsigned long long above = samp_suffix_excl(local, sh4, total);
  // the level-0 histogram holds every entry: its total is the normaliser, which fixes the threshold for the descent
  const unsigned long long thr = first_level ? (unsigned long long)((double)top_p * (double)total) : prev.need;
  if (threadIdx.x == 0) { sh_res[0] = ~0ull; sh_res[1] = 0; }
  __syncthreads();
  const unsigned long long c0 = prev.acc + above;
  if (c0 <= thr && c0 + local > thr) {
    unsigned long long c = c0;
    for (int j = PER - 1; j >= 0; j--) {
      if (c + own[j] > thr) { sh_res[0] = (unsigned long long)(threadIdx.x * PER + j); sh_res[1] = c; break; }
      c += own[j];
    }
  }
  __syncthreads();
  SampLevelState st;
  st.need = thr; st.pad = 0;
  if (prev.done || sh_res[0] == ~0ull) { st.prefix = 0; st.acc = 0; st.done = 1; }
  else { st.prefix = (prev.prefix << width) | sh_res[0]; st.acc = sh_res[1]; st.done = 0; }
  return st;
}

// composite-key threshold of a finished descent (kept <=> comp >= thr)
__device__ unsigned long long samp_threshold_k(const SampArgs& a, SampScratch* sc, unsigned long long* sh4, unsigned long long* sh_res) {
  const int L = SAMP_LEVELS - 1, w = samp_width(L, a.idx_bits);
  const SampLevelState st = samp_select_count(sc->cnt[L], 1 << w, w, sc->st_k[L], sh4, sh_res);
  return st.prefix;                       // the k-th entry itself is kept
}
__device__ unsigned long long samp_threshold_p(const SampArgs& a, SampScratch* sc, unsigned long long* sh4, unsigned long long* sh_res) {
  const int L = SAMP_LEVELS - 1, w = samp_width(L, a.idx_bits);
  const SampLevelState st = samp_select_mass(sc->mass[L], 1 << w, w, sc->st_p[L], false, a.top_p, sh4, sh_res);
  if (st.done) return 0ull;               // cumulative mass never exceeds top_p: nothing is cut
  // st.prefix is the first entry whose inclusive cumulative mass exceeds top_p: it is cut, unless it is the very first entry
  return st.acc == 0 ? st.prefix : st.prefix + 1ull;
}

// thresholds as seen by a stage: derived here (first stage after the filter's last level) or read back
__device__ __forceinline__ void samp_thresholds(const SampArgs& a, SampScratch* sc, unsigned long long& thr_k, unsigned long long& thr_p,
                                                unsigned long long* sh4, unsigned long long* sh_res) {
  thr_k = 0; thr_p = 0;
  if (a.top_k > 0) {
    if (a.k_from_hist) { thr_k = samp_threshold_k(a, sc, sh4, sh_res); if (blockIdx.x == 0 && threadIdx.x == 0) sc->thr_k = thr_k; }
    else thr_k = sc->thr_k;
  }
  if (a.top_p < 1.f) {
    if (a.p_from_hist) { thr_p = samp_threshold_p(a, sc, sh4, sh_res); if (blockIdx.x == 0 && threadIdx.x == 0) sc->thr_p = thr_p; }
    else thr_p = sc->thr_p;
  }
}

// ---- one digit level of a radix descent.  MODE 0: top-k (counts), MODE 1: top-p (masses over the top-k survivors) --------
template <int MODE>
__global__ __launch_bounds__(SAMP_WG) void samp_level_kernel(const SampArgs a) {
  __shared__ unsigned long long sh4[4], sh_res[2];
  __shared__ float shf[4];
  __shared__ unsigned long long lds_hist[SAMP_BINS];       // counts use the low word
  const int row = blockIdx.y, L = a.level, tid = threadIdx.x;
  SampScratch* sc = a.sc + row;

  unsigned long long thr_k = 0;
  SampLevelState st;
  if (L == 0) {
    st.prefix = 0; st.acc = 0; st.done = 0; st.pad = 0;
    st.need = MODE == 0 ? (unsigned long long)(a.top_k < (long long)a.V ? a.top_k : (long long)a.V) : 0ull;
    if (MODE == 1 && a.top_k > 0) {       // the first top-p level runs right after the last top-k level
      thr_k = samp_threshold_k(a, sc, sh4, sh_res);
      if (blockIdx.x == 0 && tid == 0) sc->thr_k = thr_k;
    }
  } else {
    const int wprev = samp_width(L - 1, a.idx_bits);
    if (MODE == 0) st = samp_select_count(sc->cnt[L - 1], 1 << wprev, wprev, sc->st_k[L - 1], sh4, sh_res);
    else st = samp_select_mass(sc->mass[L - 1], 1 << wprev, wprev, sc->st_p[L - 1], L == 1, a.top_p, sh4, sh_res);
    if (MODE == 1 && a.top_k > 0) thr_k = sc->thr_k;
  }
  if (blockIdx.x == 0 && tid == 0) { if (MODE == 0) sc->st_k[L] = st; else sc->st_p[L] = st; }
  if (st.done) return;                     // block-uniform

  const int w = samp_width(L, a.idx_bits), shift = samp_shift(L, a.idx_bits), nbins = 1 << w;
  for (int b = tid; b < nbins; b += SAMP_WG) lds_hist[b] = 0ull;
  float mx = 0.f;
  if (MODE == 1) mx = samp_row_max(a, row, shf);
  __syncthreads();
  SampElems e;
  samp_load(a, row, blockIdx.x, e);
  // runs of equal digits inside a thread are merged before they reach the LDS atomics
  int cur = -1; unsigned long long sum = 0;
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    const bool take = e.in[j] && (e.comp[j] >> (shift + w)) == st.prefix && (MODE == 0 || e.comp[j] >= thr_k);
    if (!take) continue;
    const int d = (int)((e.comp[j] >> shift) & (unsigned long long)(nbins - 1));
    const unsigned long long val = MODE == 0 ? 1ull : samp_mass(e.v[j], mx);
    if (d != cur) { if (cur >= 0 && sum) atomicAdd(&lds_hist[cur], sum); cur = d; sum = 0; }
    sum += val;
  }
  if (cur >= 0 && sum) atomicAdd(&lds_hist[cur], sum);
  __syncthreads();
  for (int b = tid; b < nbins; b += SAMP_WG) {
    const unsigned long long hv = lds_hist[b];
    if (!hv) continue;
    if (MODE == 0) atomicAdd(&sc->cnt[L][b], (unsigned int)hv);
    else atomicAdd(&sc->mass[L][b], hv);
  }
}

// ---- partial sums over the kept set.  STAGE 0: Z of the set min-p looks at; 1: Z of the final set; 2: final probabilities ----
__device__ __forceinline__ float samp_ordered_sum(const double* part, int n, double* shd) {
  // the normaliser: double partial sums in a fixed association, rounded to fp32 once
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += SAMP_WG) s += part[i];
  return (float)samp_block_sum_d(s, shd);
}

template <int STAGE>
__global__ __launch_bounds__(SAMP_WG) void samp_sum_kernel(const SampArgs a) {
  __shared__ unsigned long long sh4[4], sh_res[2];
  __shared__ float shf[4];
  __shared__ double shd[4];
  const int row = blockIdx.y, tid = threadIdx.x, nwg = gridDim.x;
  SampScratch* sc = a.sc + row;
  unsigned long long thr_k, thr_p;
  samp_thresholds(a, sc, thr_k, thr_p, sh4, sh_res);
  const float mx = samp_row_max(a, row, shf);
  SampElems e;
  samp_load(a, row, blockIdx.x, e);
  bool kept[SAMP_EPT];
  float ex[SAMP_EPT];
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    kept[j] = e.in[j] && e.comp[j] >= thr_k && e.comp[j] >= thr_p;
    ex[j] = kept[j] ? expf(e.v[j] - mx) : 0.f;
  }
  if (STAGE >= 1 && a.min_p > 0.f) {        // p_i < minP * p_max with p_max = exp(0) * inv (Sampler.cpp:68-74)
    const float z0 = samp_ordered_sum(sc->wg_z, nwg, shd);
    const float inv0 = 1.0f / z0;
    const float thr = (1.0f * inv0) * a.min_p;
#pragma unroll
    for (int j = 0; j < SAMP_EPT; j++)
      if (kept[j] && ex[j] * inv0 < thr) { kept[j] = false; ex[j] = 0.f; }
  }
  if (STAGE <= 1) {
    double s = (((double)ex[0] + (double)ex[1]) + (double)ex[2]) + (double)ex[3];
    s = samp_block_sum_d(s, shd);
    if (tid == 0) (STAGE == 0 ? sc->wg_z : sc->wg_z2)[blockIdx.x] = s;
    return;
  }
  const float z = samp_ordered_sum(sc->wg_z2, nwg, shd);
  const float inv = 1.0f / z;
  double mine = 0.0;
  float p[SAMP_EPT];
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) { p[j] = kept[j] ? ex[j] * inv : 0.f; mine += (double)p[j]; }
  if (a.probs_out) {
    float* po = a.probs_out + (size_t)row * a.probs_stride;
#pragma unroll
    for (int j = 0; j < SAMP_EPT; j++) if (e.in[j]) po[e.base + j] = p[j];
  }
  // block total in thread order (double)
  const double below = samp_prefix_excl_d(mine, shd);
  if (tid == SAMP_WG - 1) sc->wg_p[blockIdx.x] = below + mine;
}

// ---- the draw: one workgroup per row ----------------------------------------------------------------------------------
struct SampPickArgs {
  SampArgs s;
  int nwg;
  const unsigned long long* seed;   // device word
  FinalizeArgs fin;                 // token publish / rings / embedding (part_* unused)
};

template <int DT>
__global__ __launch_bounds__(SAMP_WG) void samp_pick_kernel(const SampPickArgs pa) {
  __shared__ float shf[4];
  __shared__ double shd[4];
  __shared__ int s_wg, s_pick, s_last, s_pos;
  __shared__ double s_run;
  const SampArgs& a = pa.s;
  const int row = pa.fin.row, tid = threadIdx.x, nwg = pa.nwg;
  SampScratch* sc = a.sc + row;
  const float mx = samp_row_max(a, row, shf);
  // u, and the workgroup whose index range holds the draw (sequential scan of <= 1024 partial sums by one thread)
  if (tid == 0) {
    unsigned long long s = (*pa.seed) * 0x9E3779B97F4A7C15ull + (unsigned long long)(*pa.fin.pos + pa.fin.advance_pos) * 0xD1342543DE82EF95ull +
                           (unsigned long long)pa.fin.row;
    const double u = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
    double run = 0.0; int w = 0, last = 0;
    for (; w < nwg; w++) { const double pw = sc->wg_p[w]; if (pw > 0.0) last = w; if (u < run + pw) break; run += pw; }
    if (w >= nwg) { w = last; run = -1.0; }          // rounding left u above the total: the last kept entry (as the oracle's loop)
    s_wg = w; s_run = run; s_pick = 0x7fffffff; s_last = -1;
    shd[0] = u;
  }
  __syncthreads();
  const double u = shd[0];
  const int wsel = s_wg;
  const double run0 = s_run;
  __syncthreads();
  unsigned long long thr_k = a.top_k > 0 ? sc->thr_k : 0ull, thr_p = a.top_p < 1.f ? sc->thr_p : 0ull;
  SampElems e;
  samp_load(a, row, wsel, e);
  bool kept[SAMP_EPT];
  float ex[SAMP_EPT];
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    kept[j] = e.in[j] && e.comp[j] >= thr_k && e.comp[j] >= thr_p;
    ex[j] = kept[j] ? expf(e.v[j] - mx) : 0.f;
  }
  if (a.min_p > 0.f) {
    const float z0 = samp_ordered_sum(sc->wg_z, nwg, shd);
    const float inv0 = 1.0f / z0;
    const float thr = (1.0f * inv0) * a.min_p;
#pragma unroll
    for (int j = 0; j < SAMP_EPT; j++)
      if (kept[j] && ex[j] * inv0 < thr) { kept[j] = false; ex[j] = 0.f; }
  }
  const float z = samp_ordered_sum(sc->wg_z2, nwg, shd);
  const float inv = 1.0f / z;
  double mine = 0.0;
  float p[SAMP_EPT];
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) { p[j] = kept[j] ? ex[j] * inv : 0.f; mine += (double)p[j]; }
  double cum = (run0 < 0.0 ? 0.0 : run0) + samp_prefix_excl_d(mine, shd);
  int hit = 0x7fffffff, last = -1;
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    if (p[j] > 0.f) {
      cum += (double)p[j];
      last = e.base + j;
      if (run0 >= 0.0 && u < cum && hit == 0x7fffffff) hit = e.base + j;
    }
  }
  if (hit != 0x7fffffff) atomicMin(&s_pick, hit);
  if (last >= 0) atomicMax(&s_last, last);
  // the histograms of this step are spent: zero them for the next one (only the filters that ran touched them)
  if (a.top_k > 0) for (int i = tid; i < SAMP_LEVELS * SAMP_BINS; i += SAMP_WG) (&sc->cnt[0][0])[i] = 0u;
  if (a.top_p < 1.f) for (int i = tid; i < SAMP_LEVELS * SAMP_BINS; i += SAMP_WG) (&sc->mass[0][0])[i] = 0ull;
  __syncthreads();
  if (tid == 0) {
    int pick = s_pick != 0x7fffffff ? s_pick : s_last;   // no hit inside the selected range: its last kept entry
    if ((unsigned)pick >= (unsigned)a.V) pick = 0;       // all-NaN logits: stay inside the embedding table
    s_pick = pick;
    *pa.fin.tok = pick;
    const int np = *pa.fin.pos + (pa.fin.advance_pos ? 1 : 0);
    if (pa.fin.advance_pos) *pa.fin.pos = np;
    s_pos = np < pa.fin.n_pos ? np : pa.fin.n_pos - 1;
    if (pa.fin.log) {
      const int st = *pa.fin.step;
      pa.fin.tok_log[(st % pa.fin.log_cap) * pa.fin.rows + pa.fin.row] = pick;
      if (pa.fin.host_ring) pa.fin.host_ring[(st % pa.fin.ring_cap) * pa.fin.rows + pa.fin.row] = pick;
      if (pa.fin.bump_step) *pa.fin.step = st + 1;
    }
  }
  __syncthreads();
  gather_embedding<DT>(pa.fin.embed, s_pick, pa.fin.x, pa.fin.H, pa.fin.wpe, pa.fin.wpe ? s_pos : 0);
}

}  // namespace tgx
