// qkv_sliced.h — the QKV projection of a batch-1 decode step as K-sliced weight tiles whose partial sums the attention launch finishes.
//
// Replaces (reference op sequence, DecoderLayer.h:40 + Attention.h:94-106):  RMSNorm -> MergedLinear qkv (+bias) -> split -> RoPE(q), RoPE(k) -> cache append
//   this launch:            part[s][n] = sum over K slice s of W[n][k] * (norm_w[k] * x[k]);   ssq[s] = sum over the slice of x[k]^2
//   the attention launch:   q / k / v = (sum_s part[s][.]) * rsqrt(sum_s ssq[s] / K + eps) [+ bias], RoPE at pos, cache append   (attn_decode.h, template RAW)
// (the RMSNorm factor is one scalar per token, so it commutes with the product: applied to the few hundred outputs a workgroup of the attention launch
// needs instead of to the 2048 inputs in every wave of this one.)
//
// Why a second QKV form (round 4, tools/probes/layer_lab.hip).  The row-sliced GEMV (gemv.h, 4 waves per row pair at hidden 2048) spends 3.4 us of its 5.4 us
// outside the weight stream: every wave loads and squares its quarter of x, the four waves of a row pair meet twice through LDS per unit, and the RoPE
// epilogue waits on a position -> cos / sin chain.  A workgroup here owns a [RB rows] x [SW columns] tile: it needs SW activations (1-2 KB), issues its
// whole tile (8 x 16 bytes per lane) at once, reduces over LPR lanes on the DPP crossbar and stores RB partial sums — no LDS, no barrier, nothing that
// depends on the position.  12.6 MB of Llama-3.2-1B qkv weights: 5.4 -> 4.4 us.  The consumer side costs the attention launch ~0.2 us (it already waits
// a memory round trip for K / V; the slab sums, the rotation and one barrier ride under it).
//
// Roofline: HBM — 2 * N * K bytes of weights per launch.  Deterministic: the slices are summed in slice order by the consumer.
#pragma once
#include "common.h"

namespace tgx {

struct QkvSlicedArgs {
  const void* W;           // [N][K] row-major, storage dtype (q | k | v rows merged like MergedLinear, Linear.h:64-79)
  const float* x;          // [K] residual stream (fp32)
  const void* norm_w;      // [K] RMSNorm weight, storage dtype
  float* part;             // [K / SW][N] partial sums (fp32), slice-major
  float* ssq;              // [K / SW] partial sums of squares of x
  int N, K;
};

template <int LPR, int NL> constexpr int qkv_sliced_rows() { return 4 * NL * (64 / LPR); }      // rows per workgroup

// LPR = lanes per row slice (SW = 8 LPR columns per K slice), NL = wave-loads per wave; grid = (ceil(N / rows), K / SW)
template <int DT, int LPR, int NL>
__global__ __launch_bounds__(256) void qkv_sliced_kernel(const QkvSlicedArgs a) {
  typedef elem_t<DT> E;
  constexpr int SW = LPR * 8, RPL = 64 / LPR, RPW = NL * RPL, RB = 4 * RPW;
  static_assert(RPW <= 64, "one store instruction per wave");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slice = blockIdx.y, row0 = blockIdx.x * RB + wv * RPW;
  const int rl = lane / LPR, cl = lane % LPR;
  // the whole tile and the slice of x / norm_w leave together
  Slice8<DT> w[NL];
  const E* Wp = static_cast<const E*>(a.W) + (size_t)slice * SW;
#pragma unroll
  for (int i = 0; i < NL; i++) w[i] = load_slice_nt<DT>(Wp + (size_t)min(row0 + i * RPL + rl, a.N - 1) * a.K, cl);
  const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + (size_t)slice * SW + cl * 8);
  f32x4 xa = xp[0], xb = xp[1];
  float nw[8];
  slice_unpack<DT>(load_slice<DT>(static_cast<const E*>(a.norm_w) + (size_t)slice * SW, cl), nw);
  if (blockIdx.x == 0 && wv == 0) {          // one wave per K slice leaves the slice's sum of squares
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) { ss = fmaf(xa[t], xa[t], ss); ss = fmaf(xb[t], xb[t], ss); }
    if (rl != 0) ss = 0.f;                   // the lanes of the other row groups hold the same slice
    ss = wave_sum(ss);
    if (lane == 0) a.ssq[slice] = ss;
  }
#pragma unroll
  for (int t = 0; t < 4; t++) { xa[t] *= nw[t]; xb[t] *= nw[4 + t]; }
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
    for (int r = 0; r < RPL; r++) {          // row i * RPL + r of this wave was summed by the lanes r * LPR ..: hand its sum to lane i * RPL + r
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), r * LPR));
      if (lane == i * RPL + r) mine = t;
    }
  }
  if (lane < RPW && row0 + lane < a.N) a.part[(size_t)slice * a.N + row0 + lane] = mine;
}

}  // namespace tgx
