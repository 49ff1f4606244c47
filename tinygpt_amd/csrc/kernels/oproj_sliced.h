// oproj_sliced.h — o_proj + residual behind split-form attention, with the merge of the attention splits in its prologue and NO separate
// attn_combine launch (batch 1, contexts beyond the direct form: the benchmark's operating point).
//
// Replaces (reference op sequence, Attention.h:108-112 + :90, DecoderLayer.h:40):
//   flashAttention's split merge (attn_combine_kernel) -> reshape [B,S,qDim] -> Linear o_proj -> x + .
//
// Why a second o_proj form.  Round 2 folded the merge into the row-sliced GEMV's prologue (PRO_ATTNCOMB): every one of its 256 workgroups
// needs the WHOLE attention output, i.e. re-reads all heads' split records (~150 KB each, 38 MB of L2 traffic) — slower than the launch
// it removed.  Here the product is sliced over K instead: workgroup (row block rb, K slice s) owns SW = 128 .. 512 columns = the output of
// 2 .. 4 query heads, so it merges only THOSE heads' records (9-35 KB), multiplies its [RB rows] x [SW columns] weight tile by them and adds
// its partial dot products into per-row accumulators.  The cross-workgroup sum runs on FIXED-POINT 64-bit integer atomics (2^-32 units):
// integer addition commutes, so the result does not depend on arrival order — bit-reproducible like every other reduction of the path
// (an fp32 atomic sum would not be).  Measured (tools/probes/atomic_probe.hip): 16 384 agent-scope int64 adds on 2048 addresses cost
// +0.3 us over plain stores.  The accumulators hold the residual stream between o_proj and down: slice 0 adds x itself, the gate_up launch
// reads them (gemv_kernel XACC: x' = fp32(acc)), the down launch adds its product to fp32(acc), stores x and leaves the accumulators at zero.
//
// Roofline: HBM — 2 * H * qd bytes of weights per launch (+ heads * nsplit * (hd + 4) * 4 bytes of records per row block, from L2).
#pragma once
#include "common.h"

namespace tgx {

#ifdef LAB_DISSECT
#define OPS_DBG(a, bit) ((a).dbg & (bit))
#else
#define OPS_DBG(a, bit) false
#endif

struct OprojSlicedArgs {
  const void* W;        // [H][ldw] storage dtype, row-major (torch Linear layout)
  int ldw;              // = heads * hd
  const float* part;    // [heads][nsplit][hd + 4] split records of attn_decode_kernel: o[hd], m (exp2 domain), l, pad
  int nsplit;           // record slots per head (<= 32); slots of splits without keys hold m = -inf
  const float* x;       // [H] residual stream (fp32), added by K slice 0
  long long* acc;       // [H] fixed-point accumulators, zero when the launch starts
  int H;
  int act16;            // option act.round16: the merged attention output is rounded to the storage dtype before the product
  int dbg;              // experiments only (tools/probes/layer_lab.hip -DLAB_DISSECT): 1 no atomics (plain stores), 2 no record merge, 4 no weight loads
#ifdef LAB_EPOCH
  unsigned* epoch;      // tools/probes/qkv_attn.h (round 6 lab): the granule tag of the layer's fused qkv || attention launch, advanced here by one thread
#endif
};

template <int LPR> constexpr int oproj_sliced_rows() { return 4 * 8 * (64 / LPR); }      // rows per workgroup: 4 waves x 8 wave-loads x rows per wave-load

// LPR = lanes per row slice: a workgroup's K slice is SW = 8 LPR columns = SW / HD query heads; grid = (H / rows, qd / SW)
// Every global load of the launch (split records, weight tile, residual) is issued before the first wait: the compiler otherwise sinks the
// record loads of later passes behind the first pass's arithmetic (a second memory round trip: +1.5 us on an 8 us launch, layer_lab dissect).
template <int DT, int HD, int LPR>
__global__ __launch_bounds__(256) void oproj_sliced_kernel(const OprojSlicedArgs a) {
  typedef elem_t<DT> E;
  constexpr int SW = LPR * 8, HPS = SW / HD, RPL = 64 / LPR, NL = 8, RPW = NL * RPL, RB = 4 * RPW;
  constexpr int DG = HD / 8, SLW = 64 / DG, NPASS = 32 / SLW;
  static_assert(HPS >= 1 && HPS <= 4 && SW % HD == 0, "a K slice holds 1..4 whole heads");
  static_assert(RPW <= 64, "one atomic instruction per wave");
  __shared__ __attribute__((aligned(16))) float xs[SW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slice = blockIdx.y, row0 = blockIdx.x * RB + wv * RPW;
  const int rl = lane / LPR, cl = lane % LPR;
  const int dg = lane % DG, sl = lane / DG;
  // 1. split records of this wave's head (wave wv merges head wv of the slice; slices of fewer than 4 heads leave the other waves without)
  const bool merger = wv < HPS;
  f32x4 d0[NPASS], d1[NPASS];
  float2 ml[NPASS];
  {
    const int h = slice * HPS + min(wv, HPS - 1);
    const float* p = a.part + (size_t)h * a.nsplit * (HD + 4);
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      const float* rec = p + (size_t)min(ps * SLW + sl, a.nsplit - 1) * (HD + 4);      // clamped: a legal record, masked below
      const f32x4* src = reinterpret_cast<const f32x4*>(rec + dg * 8);
      d0[ps] = src[0]; d1[ps] = src[1];
      ml[ps] = *reinterpret_cast<const float2*>(rec + HD);
    }
  }
  __builtin_amdgcn_sched_barrier(0);      // the records leave first: their merge runs while the weight tile is still in flight
  // 2. this wave's weight tile: NL wave-loads of RPL rows x SW columns; the residual of its rows (K slice 0)
  Slice8<DT> w[NL];
  {
    const E* Wp = static_cast<const E*>(a.W) + (size_t)slice * SW;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int row = min(row0 + i * RPL + rl, a.H - 1);
      if (!OPS_DBG(a, 4)) w[i] = load_slice_nt<DT>(Wp + (size_t)row * a.ldw, cl); else w[i].v = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
  }
  const int my_row = min(row0 + lane, a.H - 1);
  const float resid = a.x[my_row];
  __builtin_amdgcn_sched_barrier(0);
  // every load above is issued before the first use below: the empty asm makes the record registers "used" here, so the compiler waits for exactly
  // them (the weight tile and the residual may still fly) and cannot sink a later pass's loads behind the first pass's arithmetic
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) asm volatile("" : "+v"(d0[ps]), "+v"(d1[ps]), "+v"(ml[ps]));
  // 3. merge of the splits (== attn_combine_kernel: out = sum_s o_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)); lane = (split lane sl, dim group dg)
  if (merger && !OPS_DBG(a, 2)) {
    float pm[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) pm[ps] = ps * SLW + sl < a.nsplit ? ml[ps].x : -INFINITY;
    float M = pm[0];
#pragma unroll
    for (int ps = 1; ps < NPASS; ps++) M = fmaxf(M, pm[ps]);
#pragma unroll
    for (int off = DG; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      const bool live = pm[ps] != -INFINITY;          // empty splits hold stale data: select, never multiply
      const float e = live ? exp2f(pm[ps] - M) : 0.f;
      L = fmaf(live ? ml[ps].y : 0.f, e, L);
#pragma unroll
      for (int j = 0; j < 4; j++) { o[j] = fmaf(live ? d0[ps][j] : 0.f, e, o[j]); o[4 + j] = fmaf(live ? d1[ps][j] : 0.f, e, o[4 + j]); }
    }
#pragma unroll
    for (int off = DG; off < 64; off <<= 1) {
      L += __shfl_xor(L, off, 64);
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] += __shfl_xor(o[j], off, 64);
    }
    if (sl == 0) {
      f32x4* dst = reinterpret_cast<f32x4*>(&xs[wv * HD + dg * 8]);
      float on[8];
#pragma unroll
      for (int j = 0; j < 8; j++) on[j] = round_storage_if<DT>(o[j] / L, a.act16);
      dst[0] = f32x4{on[0], on[1], on[2], on[3]};
      dst[1] = f32x4{on[4], on[5], on[6], on[7]};
    }
  }
  if (OPS_DBG(a, 2)) { if (threadIdx.x < SW) xs[threadIdx.x] = 1.0f; }
  __syncthreads();
  // 4. the tile's dot products: every lane holds the 8 activations of its column slice; row sums gathered into lanes 0 .. RPW-1
  const f32x4 xa = *reinterpret_cast<const f32x4*>(&xs[cl * 8]), xb = *reinterpret_cast<const f32x4*>(&xs[cl * 8 + 4]);
  float mine = 0.f;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    float s = dot8<DT>(0.f, w[i], xa, xb);
    if constexpr (LPR == 64) s = wave_sum(s);
    else {
      s = row_group_sum<16>(s);
      if constexpr (LPR == 32) s += __shfl_xor(s, 16, 64);
    }
#pragma unroll
    for (int r = 0; r < RPL; r++) {        // row i * RPL + r of this wave was summed by the lanes r * LPR .. : hand its sum to lane i * RPL + r
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), r * LPR));
      if (lane == i * RPL + r) mine = t;
    }
  }
  // 5. one fixed-point add per row (K slice 0 carries the residual): ONE atomic instruction per wave
  if (lane < RPW && row0 + lane < a.H) {
    long long f = f32_to_fix(mine);
    if (slice == 0) f += f32_to_fix(resid);
    if (OPS_DBG(a, 1)) { if (slice == 0) a.acc[row0 + lane] = f; }
    else __hip_atomic_fetch_add(a.acc + row0 + lane, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef LAB_EPOCH
  if (a.epoch && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.epoch = *a.epoch + 1;
#endif
}

}  // namespace tgx
