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
// The draw also performs the duties of finalize_greedy_kernel (token publish, pastLength+1, token rings, next embedding
// row).  It runs at the end of the last filter's tail launch; a chain that ends in min-p (its cut depends on the normaliser
// of the set it looks at) or has no filter ends in one or two partial-sum stages and a pick launch instead.  Launches per
// sampled step: 3 per active top-k / top-p filter (first digit, compaction, tail [+ draw]), or + 2..3 with min-p / without
// filters; T = 0.8 / top-p 0.9 (the CLI defaults): 3 launches (rounds 1-5: 8, +48 us per decode step at V = 128 256; the
// one-workgroup version of round 1 added 600 us; profiles/r06_sampler_cost.txt).  The final probability vector
// (tgx_read_probs) is not part of a step: it is evaluated on demand from the thresholds and normalisers the step left.
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
  unsigned int cnt[1][SAMP_BINS];               // first-digit histograms (the later digits live in the tail's LDS)
  unsigned long long mass[1][SAMP_BINS];
  SampLevelState st_k[SAMP_LEVELS], st_p[SAMP_LEVELS];
  unsigned long long thr_k, thr_p;    // kept <=> composite key >= thr
  double wg_z[SAMP_MAX_WG];           // per-workgroup sums of exp(v - max): the set min-p looks at.  The softmax normaliser is
  double wg_z2[SAMP_MAX_WG];          // ... the final kept set (the draw walks them as tile masses).  The normaliser is accumulated in double and
                                      // rounded once (as the oracle does): independent of the summation order
  double wg_above[SAMP_MAX_WG];       // per-workgroup sums of exp(v - max) over the entries ABOVE the threshold's first-digit bin (kept for sure)
  unsigned int list_n;                // entries of the compacted list (the threshold's first-digit bin); zeroed by the tail
  float zk;                           // normaliser of the final kept set, derived by the last filter's tail (no min-p)
  float mx;                           // max(logits / T) of this step, left by the first sampler launch (SampArgs.mx_ready tells the later ones)
#ifdef TGX_SAMP_TIMELINE              // experiment builds (tools/sampler_timeline.py): wall-clock stamps of the sampler's launches, one line per step
  unsigned long long tl[64][10];
  unsigned int tl_n;
#endif
};
#ifdef TGX_SAMP_TIMELINE
#define SAMP_STAMP(sc_, k_) do { if (blockIdx.x == 0 && threadIdx.x == 0) (sc_)->tl[(sc_)->tl_n & 63u][k_] = wall_clock64(); } while (0)
#define SAMP_STAMP_NEXT(sc_) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned int n_ = (sc_)->tl_n + 1u; for (int k_ = 0; k_ < 10; k_++) (sc_)->tl[n_ & 63u][k_] = 0ull; (sc_)->tl_n = n_; } } while (0)
#else
#define SAMP_STAMP(sc_, k_) do {} while (0)
#define SAMP_STAMP_NEXT(sc_) do {} while (0)
#endif

struct SampArgs {
  const float* logits; long long logits_stride;       // [rows][V]
  const float* part_val; long long part_stride;       // lm_head per-workgroup maxima (their maximum is max(logits))
  int n_part;
  SampScratch* sc;                                    // [rows]
  float* probs_out; long long probs_stride;           // [rows][V] final probabilities (tgx_read_probs)
  int V, idx_bits, level;
  float temperature; long long top_k; float top_p; float min_p;
  int mx_ready;                       // 1: sc->mx holds the row maximum (an earlier launch of this step reduced the lm_head partials)
  int z_from_tail;                    // 1: the kept set's normaliser is sc->zk (the last filter's tail summed it); 0: the ordered sum of wg_z2
  unsigned long long* list_comp;      // [rows][V] compacted entries of the threshold's first-digit bin: composite keys ...
  float* list_v;                      // ... and logit / T
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
// 64-lane inclusive prefix sums of 64-bit values on the DPP crossbar (no LDS traffic: a __shfl of a double is two ds_bpermute round trips per step, and the
// single-workgroup tail of a sampled step is a chain of such scans): shifts by 1 / 2 / 4 / 8 inside the rows of 16 lanes, then the two row broadcasts of the
// GFX9 wave64 scan.  A lane without a source receives 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)v, CTRL, ROW_MASK, 0xf, false);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_scan_incl(unsigned long long v) {
  v += dpp_mov64<0x111, 0xf>(v);   // row_shr:1
  v += dpp_mov64<0x112, 0xf>(v);   // row_shr:2
  v += dpp_mov64<0x114, 0xf>(v);   // row_shr:4
  v += dpp_mov64<0x118, 0xf>(v);   // row_shr:8
  v += dpp_mov64<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
  v += dpp_mov64<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double wave_scan_incl(double v) {
#define TGX_SCAN_STEP(CTRL, RM) v += __builtin_bit_cast(double, dpp_mov64<CTRL, RM>(__builtin_bit_cast(unsigned long long, v)))
  TGX_SCAN_STEP(0x111, 0xf); TGX_SCAN_STEP(0x112, 0xf); TGX_SCAN_STEP(0x114, 0xf); TGX_SCAN_STEP(0x118, 0xf); TGX_SCAN_STEP(0x142, 0xa); TGX_SCAN_STEP(0x143, 0xc);
#undef TGX_SCAN_STEP
  return v;
}
__device__ __forceinline__ double samp_block_sum_d(double v, double* sh4) {
  const double inc = wave_scan_incl(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) sh4[threadIdx.x >> 6] = inc;
  __syncthreads();
  return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}
// sum of the values held by threads with a HIGHER thread index (exclusive suffix sum) and the block total (integers: total - inclusive prefix, exact)
__device__ __forceinline__ unsigned long long samp_suffix_excl(unsigned long long v, unsigned long long* sh4, unsigned long long& total) {
  const int w = threadIdx.x >> 6;
  const unsigned long long inc = wave_scan_incl(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) sh4[w] = inc;
  __syncthreads();
  unsigned long long below = 0;
  for (int k = 0; k < w; k++) below += sh4[k];
  total = sh4[0] + sh4[1] + sh4[2] + sh4[3];
  return total - (below + inc);
}
// sum of the values held by threads with a LOWER thread index (exclusive prefix sum), double
__device__ __forceinline__ double samp_prefix_excl_d(double v, double* sh4) {
  const int w = threadIdx.x >> 6;
  const double inc = wave_scan_incl(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) sh4[w] = inc;
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
  if (a.mx_ready) return a.sc[row].mx;                       // (kernel-uniform)
  const float* pv = a.part_val + (size_t)row * a.part_stride;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < a.n_part; i += SAMP_WG) m = fmaxf(m, pv[i]);
  m = samp_block_max(m, sh4);
  m = a.temperature > 0.f ? m / a.temperature : m;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[row].mx = m;
  return m;
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

// thresholds as seen by the stages behind the filters' tails
__device__ __forceinline__ void samp_thresholds(const SampArgs& a, const SampScratch* sc, unsigned long long& thr_k, unsigned long long& thr_p) {
  thr_k = a.top_k > 0 ? sc->thr_k : 0ull;
  thr_p = a.top_p < 1.f ? sc->thr_p : 0ull;
}

// ---- first digit of a radix descent: the whole chip over the vocabulary.  MODE 0: top-k (counts), MODE 1: top-p (masses over the top-k survivors) ----
template <int MODE>
__global__ __launch_bounds__(SAMP_WG) void samp_level0_kernel(const SampArgs a) {
  __shared__ float shf[4];
  __shared__ unsigned long long lds_hist[SAMP_BINS];       // counts use the low word
  const int row = blockIdx.y, tid = threadIdx.x;
  SampScratch* sc = a.sc + row;
  SAMP_STAMP(sc, 0);
  const unsigned long long thr_k = (MODE == 1 && a.top_k > 0) ? sc->thr_k : 0ull;       // the top-k tail ran before
  const int w = samp_width(0, a.idx_bits), shift = samp_shift(0, a.idx_bits), nbins = 1 << w;
  SampElems e;
  samp_load(a, row, blockIdx.x, e);             // (ahead of the barriers below: the logits travel while the maximum is reduced)
  for (int b = tid; b < nbins; b += SAMP_WG) lds_hist[b] = 0ull;
  float mx = 0.f;
  if (MODE == 1) mx = samp_row_max(a, row, shf);
  __syncthreads();
  // runs of equal digits inside a thread are merged before they reach the LDS atomics
  int cur = -1; unsigned long long sum = 0;
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    const bool take = e.in[j] && (MODE == 0 || e.comp[j] >= thr_k);
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
    if (MODE == 0) atomicAdd(&sc->cnt[0][b], (unsigned int)hv);
    else atomicAdd(&sc->mass[0][b], hv);
  }
  SAMP_STAMP(sc, 1);
}

// ---- second pass of the chip: every workgroup derives the threshold's first-digit bin from the global histogram; entries above it are kept for sure
// (their exp sums per workgroup: the normaliser's bulk), entries inside it go to the compacted list the tail finishes
template <int MODE>
__global__ __launch_bounds__(SAMP_WG) void samp_compact_kernel(const SampArgs a) {
  __shared__ unsigned long long sh4[4], sh_res[2];
  __shared__ float shf[4];
  __shared__ double shd[4];
  const int row = blockIdx.y, tid = threadIdx.x;
  SampScratch* sc = a.sc + row;
  SAMP_STAMP(sc, 2);
  const unsigned long long thr_k = (MODE == 1 && a.top_k > 0) ? sc->thr_k : 0ull;
  const int w = samp_width(0, a.idx_bits), shift = samp_shift(0, a.idx_bits);
  SampLevelState st0;
  st0.prefix = 0; st0.acc = 0; st0.done = 0; st0.pad = 0;
  st0.need = MODE == 0 ? (unsigned long long)(a.top_k < (long long)a.V ? a.top_k : (long long)a.V) : 0ull;
  SampElems e;
  samp_load(a, row, blockIdx.x, e);             // (ahead of the selection's barriers: the logits travel beside the histogram)
  SampLevelState st;
  if (MODE == 0) st = samp_select_count(sc->cnt[0], 1 << w, w, st0, sh4, sh_res);
  else st = samp_select_mass(sc->mass[0], 1 << w, w, st0, true, a.top_p, sh4, sh_res);
  if (blockIdx.x == 0 && tid == 0) { if (MODE == 0) sc->st_k[1] = st; else sc->st_p[1] = st; }
  const float mx = samp_row_max(a, row, shf);
  unsigned long long* lc = a.list_comp + (size_t)row * a.V;
  float* lv = a.list_v + (size_t)row * a.V;
  double above = 0.0;
  bool mine[SAMP_EPT];
  int n_mine = 0;
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) {
    mine[j] = false;
    const bool take = e.in[j] && (MODE == 0 || e.comp[j] >= thr_k);
    if (!take) continue;
    const unsigned long long d = e.comp[j] >> shift;           // the first digit (nothing above it)
    if (st.done || d > st.prefix) above += (double)expf(e.v[j] - mx);
    else if (d == st.prefix) { mine[j] = true; n_mine++; }
  }
  // one reservation per workgroup on the list's counter (list order = arrival order of the workgroups: every use of the list is order-independent)
  unsigned long long wg_total;
  const unsigned long long after = samp_suffix_excl((unsigned long long)n_mine, sh4, wg_total);
  if (tid == 0) sh_res[0] = wg_total ? (unsigned long long)atomicAdd(&sc->list_n, (unsigned int)wg_total) : 0ull;
  __syncthreads();
  unsigned int slot = (unsigned int)(sh_res[0] + wg_total - after - (unsigned long long)n_mine);
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++)
    if (mine[j]) { lc[slot] = e.comp[j]; lv[slot] = e.v[j]; slot++; }
  above = samp_block_sum_d(above, shd);
  if (tid == 0) sc->wg_above[blockIdx.x] = above;
  SAMP_STAMP(sc, 3);
}

// ---- the draw: one workgroup per row ----------------------------------------------------------------------------------
struct SampPickArgs {
  SampArgs s;                       // the launch's first row (rows at the strides inside)
  int nwg;
  const unsigned long long* seed;   // device word
  FinalizeArgs fin;                 // token publish / rings / embedding of the launch's first row (part_* unused); row r of the launch = blockIdx.y:
  long long x_stride;               // tok + r, pos + r, x + r * x_stride (the rows' state lives in slabs)
};

// row r's view of the launch's finalize arguments
__device__ __forceinline__ FinalizeArgs samp_row_fin(const SampPickArgs& pa, int r) {
  FinalizeArgs f = pa.fin;
  f.tok += r; f.pos += r; f.x += (size_t)r * pa.x_stride; f.row = pa.fin.row + r;
  return f;
}

// the words every draw needs; their loads leave at the top of the launch (one workgroup on an idle chip: each dependent round trip costs ~1 us)
struct SampDrawWords {
  unsigned long long seed;
  int pos, step;
};
__device__ __forceinline__ SampDrawWords samp_draw_words(const unsigned long long* seed, const FinalizeArgs& fin) {
  SampDrawWords w;
  w.seed = *seed;
  w.pos = *fin.pos;
  w.step = fin.log ? *fin.step : 0;
  return w;
}

// Inverse CDF in index order + the duties of finalize_greedy_kernel, run by the 256 threads of one workgroup.  pw[j] = the UNNORMALISED kept mass
// (sum of exp(v - max) over the final kept set) of vocabulary tile 4 * tid + j; z = the kept set's normaliser; z0 = the normaliser of the set min-p
// looks at (unused without min-p).  The tile that holds the draw comes from a block-wide prefix sum over the tile masses (x 1 / z in double); INSIDE
// the tile the probabilities are the oracle's floats (e * inv) accumulated in double on top of the tiles below.  The two agree to ~1e-7 of the total:
// a draw that close to a tile boundary takes the boundary's neighbour (the fallbacks below), every other draw is the oracle's.
template <int DT>
__device__ __forceinline__ void samp_draw_and_publish(const SampArgs& a, const FinalizeArgs& fin, int nwg, int row, double (&pw)[4], const SampDrawWords& dw, float mx,
                                                      float z, float z0, unsigned long long thr_k, unsigned long long thr_p, double* shd) {
  __shared__ int s_wg, s_pick, s_last, s_pos;
  __shared__ double s_run, s_u;
  const int tid = threadIdx.x;
  const float inv = 1.0f / z;
  const double invd = (double)inv;
  if (tid == 0) {
    unsigned long long s = dw.seed * 0x9E3779B97F4A7C15ull + (unsigned long long)(dw.pos + fin.advance_pos) * 0xD1342543DE82EF95ull +
                           (unsigned long long)fin.row;
    s_u = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
    s_wg = 0x7fffffff; s_pick = 0x7fffffff; s_last = -1;
  }
  double mine_w = 0.0;
#pragma unroll
  for (int j = 0; j < 4; j++) { const int w = tid * 4 + j; if (w >= nwg) pw[j] = 0.0; mine_w += pw[j]; }
  double run = samp_prefix_excl_d(mine_w, shd);        // sum of the tiles below this thread's four (its barriers publish s_u)
  const double u = s_u;
  int lastw = -1;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int w = tid * 4 + j;
    if (w < nwg && pw[j] > 0.0) lastw = w;
    // the inclusive sums ascend, so the tiles with u < inclusive sum form a suffix: the smallest of them is the walk's stopping point
    if (w < nwg && u < (run + pw[j]) * invd) atomicMin(&s_wg, w);
    run += pw[j];
  }
  if (lastw >= 0) atomicMax(&s_last, lastw);
  __syncthreads();
  const int w0 = s_wg;
  if (w0 != 0x7fffffff && (w0 >> 2) == tid) {          // its owner publishes the mass of the tiles below it
    double r = run - mine_w;
    for (int j = 0; j < (w0 & 3); j++) r += pw[j];
    s_run = r * invd;
  }
  if (w0 == 0x7fffffff && tid == 0) { s_wg = s_last < 0 ? 0 : s_last; s_run = -1.0; }     // rounding left u above the total: the last kept entry (as the oracle's loop)
  __syncthreads();
  const int wsel = s_wg;
  const double run0 = s_run;
  SAMP_STAMP(a.sc + row, 6);
  __syncthreads();
  if (tid == 0) s_last = -1;                           // reused below for the last kept ENTRY of the selected tile
  __syncthreads();
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
    const float inv0 = 1.0f / z0;
    const float thr = (1.0f * inv0) * a.min_p;
#pragma unroll
    for (int j = 0; j < SAMP_EPT; j++)
      if (kept[j] && ex[j] * inv0 < thr) { kept[j] = false; ex[j] = 0.f; }
  }
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
  __syncthreads();
  if (tid == 0) {
    int pick = s_pick != 0x7fffffff ? s_pick : s_last;   // no hit inside the selected tile: its last kept entry
    if ((unsigned)pick >= (unsigned)a.V) pick = 0;       // all-NaN logits: stay inside the embedding table
    s_pick = pick;
    *fin.tok = pick;
    const int np = dw.pos + (fin.advance_pos ? 1 : 0);
    if (fin.advance_pos) *fin.pos = np;
    s_pos = np < fin.n_pos ? np : fin.n_pos - 1;
    if (fin.log) {
      const int st = dw.step;
      fin.tok_log[(st % fin.log_cap) * fin.rows + fin.row] = pick;
      if (fin.host_ring) fin.host_ring[(st % fin.ring_cap) * fin.rows + fin.row] = pick;
      if (fin.done) { if (atomicAdd(fin.done, 1) == fin.done_total - 1) { *fin.done = 0; *fin.step = st + 1; } }
      else if (fin.bump_step) *fin.step = st + 1;
    }
  }
  SAMP_STAMP(a.sc + row, 7);
  __syncthreads();
  gather_embedding<DT>(fin.embed, s_pick, fin.x, fin.H, fin.wpe, fin.wpe ? s_pos : 0);
  __syncthreads();
  SAMP_STAMP(a.sc + row, 8);
  SAMP_STAMP_NEXT(a.sc + row);
}

// ---- the tail: ONE workgroup per row takes the remaining four digits over the compacted list and derives the filter's threshold.  PICK (the last
// filter of a chain without min-p): it also derives the kept set's normaliser and the kept mass of every
// vocabulary tile (bulk sums of the compaction pass + the list's kept entries) and draws — the step's token leaves this launch
template <int MODE, bool PICK, int DT>
__global__ __launch_bounds__(SAMP_WG) void samp_tail_kernel(const SampPickArgs pa) {
  __shared__ unsigned long long sh4[4], sh_res[2];
  __shared__ float shf[4];
  __shared__ double shd[4];
  __shared__ unsigned long long lds_hist[SAMP_BINS];
  __shared__ unsigned int lds_cnt[SAMP_BINS];
  const SampArgs& a = pa.s;
  const int row = blockIdx.y, tid = threadIdx.x, nwg = pa.nwg;
  const FinalizeArgs fin = samp_row_fin(pa, row);
  SampScratch* sc = a.sc + row;
  const unsigned long long* lc = a.list_comp + (size_t)row * a.V;
  const float* lv = a.list_v + (size_t)row * a.V;
  SAMP_STAMP(sc, 4);
  const int n = (int)sc->list_n;
  SampDrawWords dw{};
  double aw[4] = {0.0, 0.0, 0.0, 0.0};    // the compaction pass's bulk sums of this thread's four tiles (in flight beside the list)
  unsigned long long thr_k_prev = 0ull;
  if (PICK) {
    dw = samp_draw_words(pa.seed, fin);
#pragma unroll
    for (int j = 0; j < 4; j++) { const int w = tid * 4 + j; const double t = sc->wg_above[min(w, nwg - 1)]; aw[j] = w < nwg ? t : 0.0; }
    if (MODE == 1 && a.top_k > 0) thr_k_prev = sc->thr_k;
  }
  const float mx = samp_row_max(a, row, shf);
  SampLevelState st = MODE == 0 ? sc->st_k[1] : sc->st_p[1];
  // the list is read ONCE: up to TAIL_CACHE entries per thread stay in registers (keys, and for top-p their masses) for the digits, the lone-entry
  // search and the normaliser; a longer list (every logit in one first-digit bin) streams from memory at each use instead
  constexpr int TAIL_CACHE = 16;
  const bool cached = n <= SAMP_WG * TAIL_CACHE;           // (block-uniform)
  unsigned long long ck[TAIL_CACHE], cm[TAIL_CACHE];
  float cv[TAIL_CACHE];
#pragma unroll
  for (int q = 0; q < TAIL_CACHE; q++) {                   // the loads do not wait for n (the buffers hold V entries): one round trip less in front of the digits
    const int i = tid + q * SAMP_WG;
    const unsigned long long k = i < a.V ? lc[i] : 0ull;
    const float v = i < a.V ? lv[i] : 0.f;
    const bool in = cached && i < n;
    ck[q] = in ? k : 0ull;                                 // key 0 matches no prefix below the first digit of a real entry
    cv[q] = in ? v : 0.f;
  }
  if (MODE == 1 || PICK) {
#pragma unroll
    for (int q = 0; q < TAIL_CACHE; q++) cm[q] = samp_mass(cv[q], mx);
  }
  __shared__ unsigned long long s_only;
  bool unique = false;                    // the chosen bin holds ONE entry: it is the threshold entry, the remaining digits have nothing to decide
  for (int L = 1; L < SAMP_LEVELS && !st.done && !unique; L++) {
    const int w = samp_width(L, a.idx_bits), shift = samp_shift(L, a.idx_bits), nbins = 1 << w;
    __syncthreads();
    for (int b = tid; b < nbins; b += SAMP_WG) { lds_cnt[b] = 0u; if (MODE == 1) lds_hist[b] = 0ull; }
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int q = 0; q < TAIL_CACHE; q++) {
        if (tid + q * SAMP_WG >= n || (ck[q] >> (shift + w)) != st.prefix) continue;
        const int d = (int)((ck[q] >> shift) & (unsigned long long)(nbins - 1));
        atomicAdd(&lds_cnt[d], 1u);
        if (MODE == 1) atomicAdd(&lds_hist[d], cm[q]);
      }
    } else {
      for (int i = tid; i < n; i += SAMP_WG) {
        const unsigned long long cmp = lc[i];
        if ((cmp >> (shift + w)) != st.prefix) continue;
        const int d = (int)((cmp >> shift) & (unsigned long long)(nbins - 1));
        atomicAdd(&lds_cnt[d], 1u);
        if (MODE == 1) atomicAdd(&lds_hist[d], samp_mass(lv[i], mx));
      }
    }
    __syncthreads();
    if (MODE == 0) st = samp_select_count(lds_cnt, nbins, w, st, sh4, sh_res);
    else st = samp_select_mass(lds_hist, nbins, w, st, false, a.top_p, sh4, sh_res);
    if (!st.done && L + 1 < SAMP_LEVELS && lds_cnt[(int)(st.prefix & (unsigned long long)(nbins - 1))] == 1u) {     // (block-uniform)
      if (cached) {
#pragma unroll
        for (int q = 0; q < TAIL_CACHE; q++) if (tid + q * SAMP_WG < n && (ck[q] >> shift) == st.prefix) s_only = ck[q];
      } else {
        for (int i = tid; i < n; i += SAMP_WG) if ((lc[i] >> shift) == st.prefix) s_only = lc[i];
      }
      __syncthreads();
      st.prefix = s_only;                 // the full composite key, as the last digit would have left it
      unique = true;
    }
  }
  // kept <=> composite key >= thr.  top-k: the k-th entry itself is kept; top-p: st.prefix is the first entry whose inclusive cumulative mass exceeds
  // top_p — it is cut, unless it is the very first entry; a cumulative mass that never exceeds top_p cuts nothing
  unsigned long long thr;
  if (MODE == 0) thr = st.prefix;
  else thr = st.done ? 0ull : (st.acc == 0 ? st.prefix : st.prefix + 1ull);
  if (tid == 0) { if (MODE == 0) sc->thr_k = thr; else sc->thr_p = thr; }
  SAMP_STAMP(sc, 5);
  __syncthreads();
  // the first-digit histogram and the list of this filter are spent (PICK: zeroed behind the draw — a barrier waits for the stores in flight)
  if (!PICK) {
    for (int b = tid; b < SAMP_BINS; b += SAMP_WG) { if (MODE == 0) sc->cnt[0][b] = 0u; else sc->mass[0][b] = 0ull; }
    if (tid == 0) sc->list_n = 0u;
    return;
  }
  // the list's kept entries: their exp sums join the normaliser (double, rounded once), their masses their tile's (2^-40 fixed point: integer LDS
  // atomics are order-independent, so the draw is reproducible)
  for (int b = tid; b < SAMP_MAX_WG; b += SAMP_WG) lds_hist[b] = 0ull;
  __syncthreads();
  const unsigned long long idx_mask = (1ull << a.idx_bits) - 1ull;
  double s = ((aw[0] + aw[1]) + aw[2]) + aw[3];
  if (cached) {
#pragma unroll
    for (int q = 0; q < TAIL_CACHE; q++)
      if (tid + q * SAMP_WG < n && ck[q] >= thr) {
        s += (double)expf(cv[q] - mx);
        atomicAdd(&lds_hist[(int)((idx_mask - (ck[q] & idx_mask)) / SAMP_TILE)], cm[q]);
      }
  } else {
    for (int i = tid; i < n; i += SAMP_WG) {
      const unsigned long long cmp = lc[i];
      if (cmp < thr) continue;
      const float v = lv[i];
      s += (double)expf(v - mx);
      atomicAdd(&lds_hist[(int)((idx_mask - (cmp & idx_mask)) / SAMP_TILE)], samp_mass(v, mx));
    }
  }
  s = samp_block_sum_d(s, shd);           // (its barriers also order the LDS atomics before the reads below)
  const float z = (float)s;
  if (tid == 0) sc->zk = z;               // tgx_read_probs evaluates the probabilities from it
  double pw[4];
#pragma unroll
  for (int j = 0; j < 4; j++) pw[j] = aw[j] + (double)lds_hist[tid * 4 + j] * (1.0 / 1099511627776.0);
  __syncthreads();                        // shd is reused by the draw
  samp_draw_and_publish<DT>(a, fin, nwg, row, pw, dw, mx, z, 0.f, MODE == 0 ? thr : thr_k_prev, MODE == 1 ? thr : 0ull, shd);
  for (int b = tid; b < SAMP_BINS; b += SAMP_WG) { if (MODE == 0) sc->cnt[0][b] = 0u; else sc->mass[0][b] = 0ull; }
  if (tid == 0) sc->list_n = 0u;
}

// ---- partial sums over the kept set.  STAGE 0: Z of the set min-p looks at; 1: Z of the final set (and the tile masses the draw walks);
// 2: the final probabilities — not part of a step: tgx_read_probs launches it on demand ----
__device__ __forceinline__ float samp_ordered_sum(const double* part, int n, double* shd) {
  // the normaliser: double partial sums in a fixed association, rounded to fp32 once
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += SAMP_WG) s += part[i];
  return (float)samp_block_sum_d(s, shd);
}

template <int STAGE>
__global__ __launch_bounds__(SAMP_WG) void samp_sum_kernel(const SampArgs a) {
  __shared__ float shf[4];
  __shared__ double shd[4];
  const int row = blockIdx.y, tid = threadIdx.x, nwg = gridDim.x;
  SampScratch* sc = a.sc + row;
  SAMP_STAMP(sc, STAGE == 0 ? 0 : 2);
  SampElems e;
  samp_load(a, row, blockIdx.x, e);
  unsigned long long thr_k, thr_p;
  samp_thresholds(a, sc, thr_k, thr_p);
  const float mx = samp_row_max(a, row, shf);
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
    SAMP_STAMP(sc, STAGE == 0 ? 1 : 3);
    return;
  }
  const float z = a.z_from_tail ? sc->zk : samp_ordered_sum(sc->wg_z2, nwg, shd);
  const float inv = 1.0f / z;
  float* po = a.probs_out + (size_t)row * a.probs_stride;
#pragma unroll
  for (int j = 0; j < SAMP_EPT; j++) if (e.in[j]) po[e.base + j] = kept[j] ? ex[j] * inv : 0.f;
}

// the draw of a chain that ends in partial-sum stages (min-p, or no filter at all): the tile masses are stage 1's sums
template <int DT>
__global__ __launch_bounds__(SAMP_WG) void samp_pick_kernel(const SampPickArgs pa) {
  __shared__ float shf[4];
  __shared__ double shd[4];
  const SampArgs& a = pa.s;
  const int row = blockIdx.y, tid = threadIdx.x, nwg = pa.nwg;
  const FinalizeArgs fin = samp_row_fin(pa, row);
  SampScratch* sc = a.sc + row;
  SAMP_STAMP(sc, 4);
  double pw[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { const int w = tid * 4 + j; pw[j] = sc->wg_z2[min(w, nwg - 1)]; }
  const SampDrawWords dw = samp_draw_words(pa.seed, fin);
  const unsigned long long thr_k = a.top_k > 0 ? sc->thr_k : 0ull, thr_p = a.top_p < 1.f ? sc->thr_p : 0ull;
  const float mx = samp_row_max(a, row, shf);
  const float z0 = a.min_p > 0.f ? samp_ordered_sum(sc->wg_z, nwg, shd) : 0.f;
  __syncthreads();
  const float z = samp_ordered_sum(sc->wg_z2, nwg, shd);
  __syncthreads();
  SAMP_STAMP(sc, 5);
  samp_draw_and_publish<DT>(a, fin, nwg, row, pw, dw, mx, z, z0, thr_k, thr_p, shd);
}

}  // namespace tgx
