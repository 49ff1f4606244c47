// l2_prefetch.h — L2 prefetch chaining between the launches of a decode step (round 3).
//
// Measured on MI355X (tools/probes/l2_persist_probe.hip, xcc_map_probe.hip): the per-XCD L2s keep their lines across a kernel boundary
// (a 16 MB region read again by the next launch with the same workgroup -> bytes map: 1.96 us instead of 4.46), and workgroup `linear id`
// always runs on XCD `linear id % 8`.  A decode step is 98 dependent launches whose fabric (HBM / Infinity Cache -> XCD) is idle for half
// of the step — boundaries, ramps, the attention pair's 8.6 us for 4 MB.  So every launch carries a few extra PREFETCH WORKGROUPS (appended
// behind its compute workgroups): they touch, line by line, the weights the NEXT weight-streaming launches will read, on the XCD whose
// workgroups will read them — the consumer's first loads then hit its own L2 instead of crossing the fabric cold.
//
// What a prefetch workgroup needs to know is the consumer GEMV's (gemv.h) unit -> rows map and workgroup -> units map:
//   consumer workgroup b (grid Gc, UPB units per pass) handles units b*UPB + t*Gc*UPB + {0..UPB-1} for t = 0, 1, ...; b runs on XCD b % 8.
// The target list of XCD x is those units in the consumer's own order (pass-major), cut at a byte budget; the prefetch waves on XCD x (read
// from the XCC_ID hardware register) take its touch instructions round-robin.
// A prefetch is a hint: nothing depends on it for correctness — the loaded values are discarded.
#pragma once
#include "common.h"

namespace tgx {

enum { PF_ROWS_PAIR = 0, PF_ROWS_SILU = 1, PF_ROWS_ROPE = 2 };   // unit -> rows map of the consumer (gemv.h unit_rows)

struct PfTarget {
  const void* W;            // consumer's weight matrix (nullptr: no target); rows are contiguous (ldw == K)
  int row_bytes;            // bytes of a row (K * esz)
  int rows_map;             // PF_ROWS_*
  int N, hd;                // rows of W;  PF_ROWS_ROPE: head_dim
  int units, grid, upb;     // consumer launch geometry
  int budget_wp;            // workgroup passes per XCD to prefetch (consumer order)
  int wp_start;             // first workgroup pass of the list to touch (an earlier launch covered [0, wp_start))
};

struct PfArgs {
  PfTarget t[2];            // up to two consumers per launch
  int n_compute;            // x-blocks >= n_compute are prefetch workgroups (0: none in this launch)
  int stride;               // bytes between touched addresses (the L2 line: 128)
  unsigned* sink;           // never written (keeps the touches alive)
};

__device__ __forceinline__ unsigned pf_xcc_id() {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  return id & 7u;
}

// Runs in every thread of a prefetch workgroup (x-block index >= a.n_compute; block size a multiple of 64).
// A "workgroup pass" (wp) = the rows one consumer workgroup reads in one pass of its unit loop: two runs of upb contiguous rows (one run of
// 2 upb rows for the PAIR map).  The target list of XCD x is wp = t * nb_x + kk -> consumer workgroup b = x + 8 kk, pass t.  A touch
// instruction covers 64 lines of one wp; the prefetch waves of an XCD take the list's touch instructions round-robin (static: the rank of a
// prefetch workgroup within its XCD follows from its block index, xcc_map_probe), eight in flight per wave.  All decoding is wave-uniform.
__device__ __forceinline__ unsigned pf_run(const PfArgs& a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);
  const int x = (int)pf_xcc_id();
  // prefetch x-blocks on this XCD: those of [n_compute, gridDim.x) congruent to (this block's x) mod 8, in every (y, z) slab
  const int bx = (int)blockIdx.x, first = a.n_compute + ((bx - a.n_compute) & 7);
  const int rank = (bx - first) >> 3, cnt = ((int)gridDim.x - first + 7) >> 3;
  const int slab = (int)(blockIdx.y + gridDim.y * blockIdx.z), nslab = (int)(gridDim.y * gridDim.z);
  const int W = ((rank * nslab + slab) * nwv + wv), Wtot = cnt * nslab * nwv;
  unsigned acc = 0;
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    const PfTarget& p = a.t[k];
    if (!p.W || p.budget_wp <= 0) continue;
    const int nb_x = (p.grid - x + 7) >> 3;                    // consumer workgroups on this XCD
    if (nb_x <= 0) continue;
    const int lpr = (p.row_bytes + a.stride - 1) / a.stride;   // touches per row
    const int lrun = p.upb * lpr;                              // touches per run of upb rows
    const int ipw = (2 * lrun + 63) >> 6;                      // touch instructions per workgroup pass
    const int total = p.budget_wp * ipw, first_i = p.wp_start * ipw;
    const int half = p.hd >> 1;
    const unsigned char* Wb = static_cast<const unsigned char*>(p.W);
    const size_t last_byte = (size_t)p.N * p.row_bytes - 4;
    auto touch = [&](int ii) -> unsigned {                     // ii wave-uniform
      const int wp = ii / ipw, sub = ii - wp * ipw;
      const int t = wp / nb_x, kk = wp - t * nb_x;
      int unit0 = ((x + 8 * kk) + t * p.grid) * p.upb;
      unit0 = min(unit0, max(p.units - p.upb, 0));             // beyond the consumer's last pass: re-touch its last units
      int ra, rb;                                              // first rows of the two runs
      if (p.rows_map == PF_ROWS_SILU) { ra = unit0; rb = (p.N >> 1) + unit0; }
      else if (p.rows_map == PF_ROWS_ROPE) { const int hh = unit0 / half; ra = hh * p.hd + (unit0 - hh * half); rb = ra + half; }
      else { ra = 2 * unit0; rb = ra + p.upb; }
      const int f = sub * 64 + lane;                           // line within the wp: run A then run B
      const size_t off = f < lrun ? (size_t)ra * p.row_bytes + (size_t)f * a.stride : (size_t)rb * p.row_bytes + (size_t)(f - lrun) * a.stride;
      return *reinterpret_cast<const unsigned*>(Wb + min(off, last_byte));
    };
#pragma unroll 1
    for (int i0 = first_i + W; i0 < total; i0 += 8 * Wtot) {   // eight touch instructions in flight per wave
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = touch(min(i0 + j * Wtot, total - 1));
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= v[j];
    }
  }
  return acc;
}

}  // namespace tgx
