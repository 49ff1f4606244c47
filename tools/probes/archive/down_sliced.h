// down_sliced.h — down_proj + residual of a batch-1 decode step as K-sliced weight tiles adding into the fixed-point residual stream.
//
// Replaces (reference op sequence, GatedMLP.h:40 + DecoderLayer.h:41):  Linear down_proj -> x + .
//
// Why (round 4, tools/probes/layer_lab.hip): the row-sliced GEMV gives every row pair to four waves that each load their quarter of the 8192-wide
// activation vector (as many bytes as the weights), meet through LDS and a barrier, and the launch's 1024 workgroups all read all of h.  A workgroup here
// owns a [RB rows] x [SW columns] tile: SW activations (1-2 KB), the whole tile (8-16 x 16 bytes per lane) requested at once, a DPP reduction over the
// LPR lanes of a row slice, and one fixed-point atomic add per row into the residual accumulators (integer adds commute: bit-reproducible; 32 768
// agent-scope adds per launch on Llama-3.2-1B, +0.5 us, tools/probes/atomic_probe.hip).  Llama-3.2-1B 8.4 -> 7.7 us, Qwen2.5-0.5B 4.9 -> 4.1 us.
// No LDS, no barrier.  The accumulators ARE the residual stream of the step (gemv.h GemvArgs.x_acc): nothing else to write.
//
// Roofline: HBM — 2 * H * I bytes of weights per launch.
#pragma once
#include "common.h"

namespace tgx {

struct DownSlicedArgs {
  const void* W;        // [N][K] row-major, storage dtype
  const float* x;       // [K] activations (fp32: siluMul output)
  long long* acc;       // [N] fixed-point residual stream (2^-32 units)
  int N, K;
};

template <int LPR, int NL> constexpr int down_sliced_rows() { return 4 * NL * (64 / LPR); }      // rows per workgroup

// LPR = lanes per row slice (SW = 8 LPR columns per K slice), NL = wave-loads per wave; grid = (ceil(N / rows), K / SW)
template <int DT, int LPR, int NL>
__global__ __launch_bounds__(256) void down_sliced_kernel(const DownSlicedArgs a) {
  typedef elem_t<DT> E;
  constexpr int SW = LPR * 8, RPL = 64 / LPR, RPW = NL * RPL, RB = 4 * RPW;
  static_assert(RPW <= 64, "one atomic instruction per wave");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slice = blockIdx.y, row0 = blockIdx.x * RB + wv * RPW;
  const int rl = lane / LPR, cl = lane % LPR;
  Slice8<DT> w[NL];
  const E* Wp = static_cast<const E*>(a.W) + (size_t)slice * SW;
#pragma unroll
  for (int i = 0; i < NL; i++) w[i] = load_slice_nt<DT>(Wp + (size_t)min(row0 + i * RPL + rl, a.N - 1) * a.K, cl);
  const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + (size_t)slice * SW + cl * 8);
  const f32x4 xa = xp[0], xb = xp[1];
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
  if (lane < RPW && row0 + lane < a.N) __hip_atomic_fetch_add(a.acc + row0 + lane, f32_to_fix(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tgx
