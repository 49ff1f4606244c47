// skinny_dma.h — Y[M][N] = X[M][K] . W[N][K]^T for M <= 64 activation rows given as STORED 16-bit terms (hi / lo), both operands staged by LDS-DMA.
//
// Why a third skinny kernel (round 3): skinny.h stages every 256-k activation panel global -> registers -> LDS with ONE panel of look-ahead (a second
// register set does not fit) and two barriers per panel.  Dissected at four activation blocks (tools/probes/skinny4_probe.hip, profiles/r03_skinny4_probe.txt):
// with the MFMAs, the weight refills and the barriers all compiled out 16.6 of 29.4 us remain for gate_up at 64 rows — eight serial L2 round trips per
// workgroup; the weight stream is not what the kernel waits for.  Here nothing passes through registers: a stage = 64 k of the activation terms
// (all M rows, shared by the four waves) + 64 k of every wave's own 16 (or 32) weight rows, written straight into a ring of D stages by
// global_load_lds_dwordx4 (weights non-temporal); D - 1 stages are in flight while one is consumed, the waits are counted (vmcnt((D - 2) x pieces)), one
// barrier per stage.  LDS image of a stage: rows of 128 bytes (64 k) whose eight 16-byte chunks are XOR-swizzled by (row >> 1) & 7 on the SOURCE side
// (lane l of a 1-KiB piece lands at +16 l), so the 16 lanes of a fragment read (rows r .. r + 15, same k) hit 16 different bank groups.
//
// Math, accumulation order (k ascending in steps of 32, small term first) and epilogues are skinny.h's: the two kernels are bit-identical.
// Roofline: HBM — 2 N K bytes of weights per launch; the activation terms (M K 4 bytes per workgroup) come from L2.
#pragma once
#include "skinny.h"
#include "gemm_dma.h"

namespace tgx {

template <int N_>
__device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
// at most `j` stages of PPW pieces each may still be in flight
template <int PPW, int DMAX>
__device__ __forceinline__ void wait_stages(int j) {
  static_assert(PPW * (DMAX - 2) <= 63, "vmcnt is a 6-bit counter");
  switch (j) {
    case 0: wait_vmcnt_imm<0>(); break;
    case 1: wait_vmcnt_imm<PPW>(); break;
    case 2: wait_vmcnt_imm<(DMAX > 3 ? 2 * PPW : 0)>(); break;
    case 3: wait_vmcnt_imm<(DMAX > 4 ? 3 * PPW : 0)>(); break;
    default: wait_vmcnt_imm<(DMAX > 5 ? 4 * PPW : 0)>(); break;
  }
}
__device__ __forceinline__ void dma_1k_nt(const void* gsrc, unsigned lds_dst) {      // dma_1k (gemm_dma.h) for data read once: the weight rows
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int SKD_KS = 64;                               // k per stage: 128-byte rows
__host__ __device__ constexpr int skd_xrows(int mb, int nt) { return (nt * mb * 16 + 31) / 32 * 32; }      // rows of a stage's activation image: whole 1-KiB pieces per wave
__host__ __device__ constexpr int skd_stage_bytes(int mb, int nbw, int nt = 2) { return (skd_xrows(mb, nt) + 4 * 16 * nbw) * SKD_KS * 2; }
// ring depth, measured (tools/probes/skinny4_probe.hip, -DTGX_SKD_DEPTH_MAX=n; us at depth 3 / 4 / 5 / 6): gate_up 16 rows 17.8 / 15.6 / 15.4 / 15.4, 32 rows
// 18.5 / 16.8 / 17.0 / 17.3, 64 rows 22.1 / 22.1 / 22.3 / 22.7; down (16 K splits of 512 k) 32 rows 11.0 / 11.5 / 12.2 / 13.4, 64 rows 14.6 / 16.5 / 17.7 / 18.3 —
// short K ranges pay the ring's fill, and a shallow ring leaves room for a second workgroup per CU: 4 stages, 3 at four activation blocks
#ifndef TGX_SKD_DEPTH_MAX
#define TGX_SKD_DEPTH_MAX 0
#endif
__host__ __device__ constexpr int skd_depth(int mb, int nbw, int nt = 2) {
  return TGX_SKD_DEPTH_MAX ? ((144 * 1024) / skd_stage_bytes(mb, nbw, nt) > TGX_SKD_DEPTH_MAX ? TGX_SKD_DEPTH_MAX : (144 * 1024) / skd_stage_bytes(mb, nbw, nt))
                           : (3 * skd_stage_bytes(mb, nbw, nt) > 160 * 1024 ? 2 : (mb >= 4 ? 3 : 4));      // (eight blocks x three terms: a double buffer is what fits)
}
__host__ __device__ constexpr size_t skd_lds_bytes(int mb, int nbw, int nt = 2) { return (size_t)skd_depth(mb, nbw, nt) * skd_stage_bytes(mb, nbw, nt); }

// MB = 16-row activation blocks (1, 2, 4; 8 = 128 rows, 64-row workgroups only), NBW = 16-row weight blocks per wave (1: 64-row workgroups, 2: 128-row workgroups); NT = terms per activation
// (2: hi, lo; 3: + A_lo2, the QKV product whose K / V results are rounded to 16 bits again).
// Needs K % 64 == 0 and, for split products, k_per % 64 == 0 (the launcher falls back to skinny.h otherwise).  grid = (N / (64 NBW), K splits).
template <int DT, int EPI, int MB, int NBW, int NT = 2>
__global__ __launch_bounds__(256) void skinny_dma_kernel(const GemmArgs a) {
  constexpr int KS = SKD_KS, WR = 16 * NBW, XROWS = skd_xrows(MB, NT);
  constexpr int XPW = XROWS / 8 / 4;                   // activation pieces per wave and stage (a 1-KiB piece = 8 rows of 128 bytes)
  constexpr int WPW = WR / 8;                          // weight pieces per wave and stage
  constexpr int PPW = XPW + WPW;
  constexpr int D = skd_depth(MB, NBW, NT);
  constexpr int STAGE = skd_stage_bytes(MB, NBW, NT);
  static_assert(XROWS % 32 == 0 && D >= 2, "stage geometry");
  extern __shared__ __attribute__((aligned(1024))) unsigned char skd_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)skd_lds;
  const int n0 = blockIdx.x * (4 * WR) + wv * WR;      // first tile column of this wave
  const int k_begin = EPI == GEMM_PARTIAL ? (int)blockIdx.y * a.k_per : 0;
  const int k_end = EPI == GEMM_PARTIAL ? min(a.K, k_begin + a.k_per) : a.K;
  const bool inter = EPI == GEMM_SILU || (EPI == GEMM_PARTIAL && a.interleave);
  const int nst = (k_end - k_begin) / KS;              // whole stages (checked by the launcher)

  // per-lane sources of this wave's pieces (without the stage's k offset) and their LDS offsets inside a stage
  const int prow = lane >> 3, pchunk = lane & 7;
  const bf16_t* src[PPW];
  unsigned dst[PPW];
#pragma unroll
  for (int j = 0; j < XPW; j++) {
    const int piece = wv + 4 * j, row = piece * 8 + prow;             // row of the stage's activation image: term t, activation row m
    const int t = min(row / (MB * 16), NT - 1), m = row - t * (MB * 16);                  // (rows of the padding beyond NT x MB x 16: copies, never read)
    const int chunk = pchunk ^ ((row >> 1) & 7);
    src[j] = (t == 0 ? a.A_hi : (t == 1 ? a.A_lo : a.A_lo2)) + (size_t)min(m, a.M - 1) * a.K + chunk * 8;      // rows past M: clamped copies, masked at the store
    dst[j] = (unsigned)(piece * 1024);
  }
#pragma unroll
  for (int j = 0; j < WPW; j++) {
    const int row = j * 8 + prow;                                      // row of this wave's weight block
    const int nb = min(n0 + row, a.N - 1);                            // clamped: rows past N reload the last one, masked at the store
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    const int chunk = pchunk ^ ((row >> 1) & 7);
    src[XPW + j] = a.B + brow * a.K + chunk * 8;
    dst[XPW + j] = (unsigned)(XROWS * KS * 2 + (wv * WR + j * 8) * KS * 2);
  }
  // stage s of this workgroup's K range into ring buffer s % D; stages past the end reload the last real one (every wait then counts the same pieces)
  auto issue = [&](int s) {
    const int k0 = k_begin + min(s, nst - 1) * KS;
    const unsigned sb = lds_base + (unsigned)((s % D) * STAGE);
#pragma unroll
    for (int j = 0; j < XPW; j++) dma_1k(src[j] + k0, sb + dst[j]);
#pragma unroll
    for (int j = 0; j < WPW; j++) dma_1k_nt(src[XPW + j] + k0, sb + dst[XPW + j]);
  };

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nst > 0) {
#pragma unroll
    for (int s = 0; s < D - 1; s++) issue(s);
    const int frow = lane & 15, fsw = (frow >> 1) & 7, kg = lane >> 4;
    for (int k = 0; k < nst; k++) {
      wait_stages<PPW, D>(D - 2);                      // stage k landed (this wave's pieces): only the D - 2 stages issued after it may still fly
      __builtin_amdgcn_s_barrier();                    // ... for every wave; and every wave is done reading stage k - 1
      issue(k + D - 1);                                // into the buffer stage k - 1 occupied
      const unsigned char* st = skd_lds + (size_t)(k % D) * STAGE;
      const unsigned char* xw = st + (size_t)XROWS * KS * 2 + (size_t)wv * WR * KS * 2;
#pragma unroll
      for (int ks = 0; ks < KS / 32; ks++) {
        const int chunk = ((ks * 4 + kg) ^ fsw) * 16;  // rows 16 apart share (row >> 1) & 7: one swizzle term per lane
        bf16x8 fb[NBW], fa[MB][NT];
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) fb[nb] = *reinterpret_cast<const bf16x8*>(xw + (16 * nb + frow) * (KS * 2) + chunk);
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
          for (int t = 0; t < NT; t++) fa[mb][t] = *reinterpret_cast<const bf16x8*>(st + ((t * MB + mb) * 16 + frow) * (KS * 2) + chunk);
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
          for (int nb = 0; nb < NBW; nb++) {
#pragma unroll
            for (int t = NT - 1; t >= 0; t--) acc[mb][nb] = mfma16x16<DT>(fa[mb][t], fb[nb], acc[mb][nb]);     // small terms first
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land in this CU's LDS after the workgroup has gone
  }

  // D: col n = lane&15, row m = 4*(lane>>4) + r   (skinny.h's epilogue)
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
      const int col = n0 + 16 * nb + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * mb + 4 * (lane >> 4) + r;
        const float v = acc[mb][nb][r];
        if (EPI == GEMM_SILU) {      // even lanes hold gate_i, odd lanes up_i (i = col / 2): the pair meets over the DPP crossbar
          const float other = dpp_mov<0xB1, 0xf>(v);
          if ((lane & 1) || col >= a.N || row >= a.M) continue;
          const size_t o = (size_t)row * a.inter + (size_t)(col >> 1);
          split16<DT>((v / (1.0f + expf(-v))) * other, a.out_hi[o], a.out_lo[o]);
          continue;
        }
        if (col >= a.N || row >= a.M) continue;
        if (EPI == GEMM_PARTIAL) { a.part[((size_t)blockIdx.y * a.M + row) * a.N + col] = v; continue; }
        float* dstp = a.C + (size_t)row * a.ldc + col;
        const float o = v + (a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f);
        *dstp = (EPI == GEMM_RESIDUAL) ? (*dstp + o) : o;
      }
    }
}

}  // namespace tgx
