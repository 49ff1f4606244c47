// gemm_dma.h — the 16-bit prefill GEMM of prefill.h with its operand tiles delivered by LDS-DMA (global_load_lds_dwordx4: memory ->
// LDS without passing through registers) into a two-stage ring.
//
// Why: in gemm_x2_kernel every operand byte crosses the VGPR file twice (global_load -> ds_write_b128); the ds_write path moves ~79 B/clk
// per CU, so the 48 KB of tiles per K step keep the LDS pipe as busy as the matrix pipe (MfmaUtil 29-38 %), and the register-staged
// prefetch forces two workgroup barriers per K step.  Here a K step is: issue the next stage's DMA, MFMAs on the current stage, one
// counted wait + ONE barrier.
//
// LDS image: a tile is rows x 64 elements (128-byte rows, no padding — a DMA piece is 1 KiB of consecutive LDS bytes = 8 rows); the 16-byte
// chunk c of row r lives in slot c ^ ((r >> 1) & 7): the rows of a ds_read_b128 lane group then fall on distinct bank quads.  The swizzle is
// applied on the GLOBAL side (each lane fetches the chunk that belongs in its slot) and again in the fragment read address.
#pragma once
#include <type_traits>

#include "prefill.h"

namespace tgx {

__host__ __device__ constexpr size_t gemm_dma_lds_bytes(int mi, bool three, int dbk, int ns) {
  return (size_t)ns * ((three ? 3 : 2) * 64 * mi + 128) * dbk * 2;
}

// s_waitcnt vmcnt(n) for the piece counts the rings below produce (the count must be an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// one 1-KiB LDS-DMA piece: lane l's 16 bytes land at lds_dst + 16 l (cdna_hip_programming.md §5.7: M0 carries the LDS base and is
// compiler-reserved, so it is saved, set and restored inside ONE statement)
__device__ __forceinline__ void dma_1k(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);       // wave-uniform by construction; make it provably scalar for the "s" operand
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// DBK = k per stage: 64 (128-byte rows, 8 chunk slots, swizzle by (row >> 1) & 7) or 32 (64-byte rows, 4 slots, (row >> 2) & 3: half the LDS per
// stage, so that the 128-row tile keeps three workgroups per CU)
// NS = ring stages: NS - 1 stages are in flight while one is consumed; the waits are counted (this wave's pieces of the later stages may still fly)
// The tile body: rows [m0, m0 + 64 MI) x columns [n0, n0 + 128) of C; `three` = the third split term of A is multiplied in as well.
// [k_begin, k_begin + k_len) = this workgroup's share of K (the whole of it unless EPI == GEMM_PARTIAL: split-K slab `zslab` of a short prompt).
template <int DT, int EPI, int MI, int DBK, int NS>
__device__ __forceinline__ void gemm_dma_tile(const GemmArgs& a, const int m0, const int n0, const bool three, bf16_t* dma_lds, const int k_begin, const int k_len, const int zslab) {
  constexpr int TM = 64 * MI;
  constexpr int CPR = DBK / 8;                 // 16-byte chunks per tile row
  constexpr int RPP = 64 / CPR;                // tile rows per 1-KiB piece
  constexpr int SW_SH = DBK == 64 ? 1 : 2, SW_MASK = CPR - 1;
  const int NA = three ? 3 : 2;
  const int stage_elems = (NA * TM + GBN) * DBK;                      // A_hi | A_lo | [A_lo2] | B
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;                // LDS byte offset of the ring (low 32 bits of the generic address)

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // DMA map: a piece = 8 tile rows; lane l -> row (l >> 3) of the piece, LDS slot (l & 7) -> global chunk slot ^ ((row >> 1) & 7)
  const int prow = lane / CPR, pslot = lane % CPR;
  const bool inter = EPI == GEMM_SILU || (EPI == GEMM_PARTIAL && a.interleave);
  auto issue_stage = [&](int k0, int stage) {
    const unsigned sbase = lds_base + (unsigned)(stage * stage_elems * 2);
    // A tiles: TM / 8 pieces each, dealt to the 4 waves
#pragma unroll
    for (int p = 0; p < TM / RPP / 4; p++) {
      const int piece = wv + 4 * p, row = piece * RPP + prow;
      const int chunk = pslot ^ ((row >> SW_SH) & SW_MASK);
      const size_t g = (size_t)min(m0 + row, a.M - 1) * a.K + k0 + chunk * 8;       // rows past M are computed on a clamped row, never stored
      const unsigned dst = sbase + (unsigned)(piece * 1024);
      dma_1k(a.A_hi + g, dst);
      dma_1k(a.A_lo + g, dst + (unsigned)(TM * DBK * 2));
      if (three) dma_1k(a.A_lo2 + g, dst + (unsigned)(2 * TM * DBK * 2));
    }
#pragma unroll
    for (int p = 0; p < GBN / RPP / 4; p++) {
      const int piece = wv + 4 * p, row = piece * RPP + prow;
      const int chunk = pslot ^ ((row >> SW_SH) & SW_MASK);
      const int nb = min(n0 + row, a.N - 1);
      const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
      dma_1k(a.B + brow * a.K + k0 + chunk * 8, sbase + (unsigned)(NA * TM * DBK * 2) + (unsigned)(piece * 1024));
    }
  };
  auto frag = [&](const bf16_t* tile, int row, int kchunk) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(tile + row * DBK + ((kchunk ^ ((row >> SW_SH) & SW_MASK)) << 3));
  };

  const int nk = k_len / DBK;
  const int ppw = (NA * TM + GBN) / RPP / 4;                         // pieces per wave and stage
#pragma unroll
  for (int sg = 0; sg < NS - 1; sg++)
    if (sg < nk) issue_stage(k_begin + sg * DBK, sg);
  for (int k = 0; k < nk; k++) {
    // stage k has landed for this wave once only the pieces of the stages issued after it are outstanding (LDS-DMA completes in order)
    wait_vmcnt(min(NS - 2, nk - 1 - k) * ppw);
    __builtin_amdgcn_s_barrier();                                    // ... for every wave; and every wave is done reading stage k-1
    if (k + NS - 1 < nk) issue_stage(k_begin + (k + NS - 1) * DBK, (k + NS - 1) % NS);    // into the buffer stage k-1 occupied
    const bf16_t* st = dma_lds + (size_t)(k % NS) * stage_elems;
    const bf16_t *tAh = st, *tAl = st + TM * DBK, *tAl2 = st + 2 * TM * DBK, *tB = st + NA * TM * DBK;
#pragma unroll
    for (int kk = 0; kk < DBK / 16; kk++) {
      const int kchunk = kk * 2 + (lane >> 5);
      bf16x8 fah[MI], fal[MI], fal2[MI], fb[2];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int row = wm * (32 * MI) + i * 32 + (lane & 31);
        fah[i] = frag(tAh, row, kchunk);
        fal[i] = frag(tAl, row, kchunk);
        if (three) fal2[i] = frag(tAl2, row, kchunk);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) fb[j] = frag(tB, wn * 64 + j * 32 + (lane & 31), kchunk);
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (three) acc[i][j] = mfma16<DT>(fal2[i], fb[j], acc[i][j]);
          acc[i][j] = mfma16<DT>(fal[i], fb[j], acc[i][j]);   // small terms first
          acc[i][j] = mfma16<DT>(fah[i], fb[j], acc[i][j]);
        }
    }
  }

  // epilogue: as gemm_x2_kernel (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (EPI == GEMM_SILU) {
        silu_block_store<DT>(acc[i][j], lane, col, m0 + wm * (32 * MI) + i * 32, a);
        continue;
      }
      if (col >= a.N) continue;
      if (EPI == GEMM_PARTIAL) {          // split-K slab: summed in z order by gemm_splitk_reduce_kernel or the next row-wise kernel
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < a.M) a.part[((size_t)zslab * a.M + row) * a.N + col] = acc[i][j][r];
        }
        continue;
      }
      const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const float v = acc[i][j][r] + bv;
        if (EPI == GEMM_GELU) {
          const size_t o = (size_t)row * a.N + col;
          split16<DT>(gelu_new_fast(v), a.out_hi[o], a.out_lo[o]);
          continue;
        }
        float* dst = a.C + (size_t)row * a.ldc + col;
        *dst = (EPI == GEMM_RESIDUAL) ? (*dst + v) : v;
      }
    }
}


template <int DT, int EPI, int MI, int DBK, int NS>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const bool three = a.A_lo2 != nullptr && (int)(blockIdx.x + 1) * GBN > a.three_from;      // workgroup-uniform
  if constexpr (EPI == GEMM_PARTIAL) {
    const int k_begin = (int)blockIdx.z * a.k_per, k_len = min(a.K, k_begin + a.k_per) - k_begin;
    gemm_dma_tile<DT, EPI, MI, DBK, NS>(a, (int)blockIdx.y * 64 * MI, (int)blockIdx.x * GBN, three, dma_lds, k_begin, k_len, (int)blockIdx.z);
  } else {
    gemm_dma_tile<DT, EPI, MI, DBK, NS>(a, (int)blockIdx.y * 64 * MI, (int)blockIdx.x * GBN, three, dma_lds, 0, a.K, 0);
  }
}

// XCD-aware tile order (round 4).  Workgroup `linear id` runs on XCD `linear id % 8` (tools/probes/xcc_map_probe.hip) and each XCD has its own 4 MB L2.  With
// blockIdx.x = N tile fastest, XCD j owned the N tiles j (mod 8) of EVERY row block: all eight L2s streamed their own copy of the whole activation
// matrix (gate_up at S = 2048: 373 MB fetched for 84 MB of operands, profiles/r03_prefill_mfma.txt).  Here the tile grid is cut into 2 x 4 rectangles, one
// per XCD, walked M-fastest inside: an XCD's co-resident workgroups share 1/2 of the row blocks and 1/4 of the weight tiles.  Falls back to the plain order
// when the grid does not divide.  Performance only: any bijection is correct.
__device__ __forceinline__ void xcd_tile(const GemmArgs& a, int& tm, int& tn) {
  const int gn = (int)gridDim.x, gm = (int)gridDim.y;
  tm = (int)blockIdx.y; tn = (int)blockIdx.x;
  if ((gm & 1) || (gn & 3)) return;
  const int id = tn + gn * tm, xcd = id & 7, l = id >> 3;
  const int sm = gm >> 1, sn = gn >> 2;          // the XCD's rectangle: sm x sn tiles
  const int xm = xcd >> 2, xn = xcd & 3;
  tm = xm * sm + l % sm; tn = xn * sn + l / sm;
}

// The siluMul epilogue of a 256 x 256 tile through LDS (round 5; waves own 64 x 128 blocks: WI 2, WJ 4).  In the MFMA C layout a lane ends up with ONE (gate, up) result per row
// pair, so the direct epilogue stores 2 bytes per lane, 32 contiguous bytes per row and instruction — 128 store instructions per wave and term, a quarter of a line each.  Here
// every wave writes its 64 x 64 results (hi and lo) into its own 18-KB slice of the idle ring ([64 rows][72]: 144-byte rows keep the 16-byte reads aligned and the row pairs off
// each other's banks), reads them back eight outputs per lane and stores whole 128-byte rows: 16 store instructions per wave.  Same values, same rounding.
template <int DT>
__device__ __forceinline__ void silu_epilogue_transposed(const f32x16 (&acc)[2][4], bf16_t* dma_lds, const GemmArgs& a, int wv, int lane, int m0, int n0, int wm, int wn) {
  constexpr int WI = 2, WJ = 4;
  constexpr int RS = 72;                                   // 16-bit elements per staged row (64 results + 8 of padding)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the trailing fragment reads of the ring
  __builtin_amdgcn_s_barrier();                            // ... by every wave: the ring is free
  bf16_t* sh = dma_lds + (size_t)wv * (2 * 64 * RS);       // this wave's slice: hi rows, then lo rows
  bf16_t* sl = sh + 64 * RS;
  const bool odd = lane & 1;
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {                     // as silu_block_store: the even lane finishes row r, the odd lane row r + 1 of the column pair
        const float a0 = acc[i][j][r], a1 = acc[i][j][r + 1];
        const float recv = dpp_mov<0xB1, 0xf>(odd ? a0 : a1);
        const float g = odd ? recv : a0, u = odd ? a1 : recv;
        const int rl = i * 32 + (r & 3) + (odd ? 1 : 0) + 8 * (r >> 2) + 4 * (lane >> 5);
        bf16_t hi, lo;
        split16<DT>(silu_mul_fast(g, u), hi, lo);
        const int o = rl * RS + j * 16 + ((lane & 31) >> 1);
        sh[o] = hi; sl[o] = lo;
      }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave reads only what it wrote itself: no workgroup barrier
  const int oc0 = (n0 + wn * 128) / 2 + (lane & 7) * 8;     // first of this lane's eight outputs
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const int rl = p * 8 + (lane >> 3), row = m0 + wm * 64 + rl;
    const u32x4 vh = *reinterpret_cast<const u32x4*>(sh + rl * RS + (lane & 7) * 8);
    const u32x4 vl = *reinterpret_cast<const u32x4*>(sl + rl * RS + (lane & 7) * 8);
    if (row < a.M && 2 * oc0 + 15 < a.N) {
      *reinterpret_cast<u32x4*>(a.out_hi + (size_t)row * a.inter + oc0) = vh;
      *reinterpret_cast<u32x4*>(a.out_lo + (size_t)row * a.inter + oc0) = vl;
    } else if (row < a.M) {                                  // a ragged last tile column: element by element
      const bf16_t* eh = reinterpret_cast<const bf16_t*>(&vh); const bf16_t* el = reinterpret_cast<const bf16_t*>(&vl);
      for (int e = 0; e < 8; e++)
        if (2 * (oc0 + e) + 1 < a.N) { a.out_hi[(size_t)row * a.inter + oc0 + e] = eh[e]; a.out_lo[(size_t)row * a.inter + oc0 + e] = el[e]; }
    }
  }
}

// ---- 256 x 256 tile, 8 waves, three-stage LDS-DMA ring (the wide products: gate_up / c_fc) -------------------------------------------
// Why a bigger tile and a deeper ring: one stage of the 128² kernel holds 0.2-0.4 µs of MFMA work per wave, an LDS-DMA piece takes 1-2 µs
// to land — the two-stage ring stalls on every stage and only co-resident workgroups hide it.  Here a stage (k = 32) is 48 KB
// (A_hi | A_lo | B, 256 rows x 64 bytes each) for 32 MFMAs per wave with two waves per SIMD (~0.85 µs of matrix work per SIMD), and TWO
// stages are in flight while the third is consumed: the waits are counted (vmcnt(6): this wave's six pieces of the NEXT stage may still
// fly), never a drain.  The six DMA instructions of a wave and stage are issued ONE AT A TIME between groups of MFMAs so that their issue
// cost (~100 cycles each) hides under the matrix pipe instead of opening every stage.
// The wave's output block (round 5): the A operand comes as TWO 16-bit terms (hi, lo), B as one, and an MFMA pair (lo, hi) shares its B fragment — per k16
// step a wave that owns WI x WJ blocks of 32 x 32 reads 2 WI + WJ fragments for 2 WI WJ MFMAs.  128 x 64 per wave (WI 4, WJ 2: rounds 2-4) = 10 reads per 16
// MFMAs; 64 x 128 (WI 2, WJ 4) = 8 per 16 and 15 fewer registers: gate_up at S = 2048 235 -> 230 us, same MFMAs per output element in the same order
// (bit-identical; tools/probes/gemm_lab.hip, profiles/r05_prefill.txt).
// LO = false (option act.round16: the Linear's input is rounded to the storage dtype, so A_hi IS the activation): the A_lo tile is neither staged nor
// multiplied — 4 DMA pieces per stage instead of 6, half the MFMAs (gate_up at S = 2048: 232 -> 130 us, profiles/r04_act16_cost.txt)
// (Round 5's lab-only template switches of this kernel — parts compiled out, the ping-pong schedule, the free-running operand stream — live in
// tools/probes/archive/gemm_dma_r05_lab.h, the copy tools/probes/gemm_lab.hip builds against; measured, not adopted: profiles/r05_prefill.txt section 5.)
// The siluMul epilogue goes through LDS.  In the MFMA C layout a lane ends up with ONE (gate, up) result per row pair, so the
// direct epilogue stores 2 bytes per lane, 32 contiguous bytes per row and instruction — 128 store instructions per wave and term, a quarter of a line each.  Here every wave
// writes its 64 x 64 results (hi and lo) into its own 18-KB slice of the idle ring ([64 rows][72]: 144-byte rows keep the 16-byte reads aligned and the row pairs off each
// other's banks), reads them back eight outputs per lane and stores whole 128-byte rows: 16 store instructions per wave.  Same values, same rounding.
template <int DT, int EPI, bool LO = true>
__global__ __launch_bounds__(512) void gemm_dma8_kernel(const GemmArgs a) {
  constexpr int DBK = 32, CPR = 4, RPP = 16, TMN = 256, NS = 3, WJ = 4;      // a wave owns 64 x 128 of the tile: WI x WJ = 2 x 4 blocks of 32 x 32
  constexpr int WI = 8 / WJ, NWN = TMN / (32 * WJ);      // 32 x 32 blocks per wave along M / N, waves along N
  constexpr int STAGE = 3 * TMN * DBK;                    // 16-bit elements per stage
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / NWN, wn = wv % NWN;
  int tile_m, tile_n;
  xcd_tile(a, tile_m, tile_n);
  const int m0 = tile_m * TMN, n0 = tile_n * TMN;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;
  const bool inter = EPI == GEMM_SILU;

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int prow = lane / CPR, pslot = lane % CPR;
  const int k_begin = 0, k_end = a.K;
  const bf16_t* gsrc[6];
  unsigned ldst[6];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int piece = wv + 8 * p, row = piece * RPP + prow;
    const int chunk = pslot ^ ((row >> 2) & 3);
    const size_t g = (size_t)min(m0 + row, a.M - 1) * a.K + chunk * 8;
    const int nb = min(n0 + row, a.N - 1);
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    gsrc[3 * p] = a.A_hi + g + k_begin; gsrc[3 * p + 1] = a.A_lo + g + k_begin; gsrc[3 * p + 2] = a.B + brow * a.K + chunk * 8 + k_begin;
    ldst[3 * p] = (unsigned)(piece * 1024); ldst[3 * p + 1] = ldst[3 * p] + (unsigned)(TMN * DBK * 2); ldst[3 * p + 2] = ldst[3 * p] + (unsigned)(2 * TMN * DBK * 2);
  }
  const int nk = (k_end - k_begin) / DBK;
  const int klast = (nk - 1) * DBK;
  auto issue_piece = [&](int q, int s) {
    if (!LO && q % 3 == 1) return;
    dma_1k(gsrc[q] + min(s * DBK, klast), lds_base + (unsigned)((s % NS) * STAGE * 2) + ldst[q]);
  };
  const int swz = ((lane & 31) >> 2) & 3;
  const int arow = (wm * 32 * WI + (lane & 31)) * DBK, brow_l = (wn * 32 * WJ + (lane & 31)) * DBK;
  auto read_frags = [&](int s, int kk, bf16x8* fa, bf16x8* fb) {
    const bf16_t* st = dma_lds + (size_t)(s % NS) * STAGE;
    const int ko = ((kk * 2 + (lane >> 5)) ^ swz) << 3;
#pragma unroll
    for (int j = 0; j < WJ; j++) fb[j] = *reinterpret_cast<const bf16x8*>(st + 2 * TMN * DBK + brow_l + j * 32 * DBK + ko);
#pragma unroll
    for (int i = 0; i < WI; i++) {
      fa[2 * i] = *reinterpret_cast<const bf16x8*>(st + arow + i * 32 * DBK + ko);
      if (LO) fa[2 * i + 1] = *reinterpret_cast<const bf16x8*>(st + TMN * DBK + arow + i * 32 * DBK + ko);
    }
  };
  // the 16 MFMAs of one k16 step in four groups of four (two blocks x two terms, the B fragment shared by a pair), one DMA piece behind each of the first three
  auto mfma_step = [&](const bf16x8* fa, const bf16x8* fb, int q0, int s) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const int idx = 2 * g + p, i = idx / WJ, j = idx % WJ;
        if (LO) acc[i][j] = mfma16<DT>(fa[2 * i + 1], fb[j], acc[i][j]);   // small term first
        acc[i][j] = mfma16<DT>(fa[2 * i], fb[j], acc[i][j]);
      }
      if (g < 3) issue_piece(q0 + g, s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

#pragma unroll
  for (int q = 0; q < 6; q++) issue_piece(q, 0);
#pragma unroll
  for (int q = 0; q < 6; q++) issue_piece(q, 1);
#pragma unroll
  for (int q = 0; q < 3; q++) issue_piece(q, 2);
  bf16x8 fa0[2 * WI], fb0[WJ], fa1[2 * WI], fb1[WJ];
  if (LO) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");          // stage 0 landed (this wave's pieces); stage 1 and half of stage 2 may fly
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, fa0, fb0);
  for (int k = 0; k < nk; k++) {
    read_frags(k, 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa0, fb0, 3, k + 2);
    // stage k+1 landed = only what was issued after its last piece may fly (three pieces of the previous half step, three of this one); this wave's reads of stage k are complete
    if (LO) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(k + 1, 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa1, fb1, 0, k + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (EPI == GEMM_SILU) {
    silu_epilogue_transposed<DT>(acc, dma_lds, a, wv, lane, m0, n0, wm, wn);
    return;
  }
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++) {
      const int col = n0 + wn * 32 * WJ + j * 32 + (lane & 31);
      if (EPI == GEMM_SILU) {
        silu_block_store<DT>(acc[i][j], lane, col, m0 + wm * 32 * WI + i * 32, a);
        continue;
      }
      if (col >= a.N) continue;
      const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const float v = acc[i][j][r] + bv;
        if (EPI == GEMM_GELU) {
          const size_t o = (size_t)row * a.N + col;
          split16<DT>(gelu_new_fast(v), a.out_hi[o], a.out_lo[o]);
          continue;
        }
        float* dst = a.C + (size_t)row * a.ldc + col;
        *dst = (EPI == GEMM_RESIDUAL) ? (*dst + v) : v;
      }
    }
}

// ---- 256 x 256 tile on FULL 128-byte lines (round 5) ----------------------------------------------------------------------------------------
// What bounds the eight-wave kernels above is largely operand delivery, and what bounds delivery is the number of cache LINES asked of the XCD's L2, not bytes
// (tools/probes/gemm_lab.hip, profiles/r05_prefill.txt: the same 1.6 GB lands in 146 us as 64-byte half-line pieces — k32 stages of row-major operands — and in
// 84 us as full lines; an extra 4-byte touch per line costs as much as the line).  A k32 stage of a row-major [M][K] operand is half a line per row.  Here
//   * the A operand arrives INTERLEAVED: Ai[m][K/32][hi 32 | lo 32] — the two terms of one k32 block of a row are one 128-byte line (written that way by the
//     producing row-wise kernel: same bytes, another address);
//   * B is staged per k64 block (one full line per weight row) and serves two k32 steps.
// The ring holds five 32-KB units (160 KB), issued in the order they are freed: block b = [A step 2b | B block b | A step 2b+1]; three units are in flight while
// two are consumed, as before (96 KB).  Same MFMAs in the same order as gemm_dma8_kernel: bit-identical results.  Wave = 64 x 128 of the output; the siluMul
// epilogue goes through LDS (silu_epilogue_transposed).  In the product for gate_up of prompts that fill the chip with 256 x 256 tiles (prefill.hip, option
// prefill.full_lines): 226 -> 213-216 us per launch in the lab (profiles/r05_prefill.txt).
template <int DT, int EPI>
__global__ __launch_bounds__(512) void gemm_dma8i_kernel(const GemmArgs a) {
  constexpr int TMN = 256, NSLOT = 5, UNIT = TMN * 64;     // 16-bit elements per unit: 256 rows x 128 bytes
  constexpr int WI = 2, WJ = 4;
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  int tile_m, tile_n;
  xcd_tile(a, tile_m, tile_n);
  const int m0 = tile_m * TMN, n0 = tile_n * TMN;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;
  const bool inter = EPI == GEMM_SILU;

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < WJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // output of the 128 x 128 kernel's)
  const int k_begin = 0, k_end = a.K;
  // DMA map: a unit = 32 pieces of 1 KiB = 8 rows x 128 bytes each; wave w takes pieces w, w + 8, w + 16, w + 24; chunk c of row r sits in slot c ^ ((r >> 1) & 7)
  const int prow = lane >> 3, pslot = lane & 7;
  const bf16_t *srcA[4], *srcB[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int row = (wv + 8 * p) * 8 + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    srcA[p] = a.A_hi + (size_t)min(m0 + row, a.M - 1) * (2 * (size_t)a.K) + 2 * (size_t)k_begin + chunk * 8;        // interleaved rows are 2 K elements long
    const int nb = min(n0 + row, a.N - 1);
    const size_t brow = inter ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    srcB[p] = a.B + brow * a.K + k_begin + chunk * 8;
  }
  const int nblk = (k_end - k_begin) / 64;
  // unit u = 3 b + j: j = 0 the A lines of step 2b, 1 the B lines of block b, 2 the A lines of step 2b + 1 (units past the end reload the last block)
  auto issue = [&](int b, int j, int p) {
    const int bb = min(b, nblk - 1);
    const unsigned dst = lds_base + (unsigned)(((3 * b + j) % NSLOT) * UNIT * 2) + (unsigned)((wv + 8 * p) * 1024);
    if (j == 1) dma_1k(srcB[p] + (size_t)bb * 64, dst);
    else dma_1k(srcA[p] + (size_t)(2 * bb + (j >> 1)) * 64, dst);
  };
  const int swz = ((lane & 31) >> 1) & 7;
  const int arow = (wm * 64 + (lane & 31)) * 64, brow_l = (wn * 128 + (lane & 31)) * 64;
  // fragments of k16 step kk of k32 step s: A chunks (term * 4 + kk * 2 + half), B chunks ((s & 1) * 4 + kk * 2 + half) of the block's lines
  auto read_frags = [&](int s, int kk, bf16x8* fa, bf16x8* fb) {
    const int b = s >> 1;
    const bf16_t* ua = dma_lds + (size_t)((3 * b + ((s & 1) << 1)) % NSLOT) * UNIT;
    const bf16_t* ub = dma_lds + (size_t)((3 * b + 1) % NSLOT) * UNIT;
    const int ca = kk * 2 + (lane >> 5), cb = (s & 1) * 4 + ca;
#pragma unroll
    for (int j = 0; j < WJ; j++) fb[j] = *reinterpret_cast<const bf16x8*>(ub + brow_l + j * 32 * 64 + ((cb ^ swz) << 3));
#pragma unroll
    for (int i = 0; i < WI; i++) {
      fa[2 * i] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + ((ca ^ swz) << 3));
      fa[2 * i + 1] = *reinterpret_cast<const bf16x8*>(ua + arow + i * 32 * 64 + (((4 + ca) ^ swz) << 3));
    }
  };
  // the 16 MFMAs of one k16 step in four groups of four, one DMA piece behind each of the first `np` groups: pieces p0 .. of unit (b, j)
  auto mfma_step = [&](const bf16x8* fa, const bf16x8* fb, int b, int j, int p0, int np) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const int idx = 2 * g + p, i = idx / WJ, jj = idx % WJ;
        acc[i][jj] = mfma16<DT>(fa[2 * i + 1], fb[jj], acc[i][jj]);   // small term first
        acc[i][jj] = mfma16<DT>(fa[2 * i], fb[jj], acc[i][jj]);
      }
      if (g < np) issue(b, j, p0 + g);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

#pragma unroll
  for (int u = 0; u < 5; u++)
#pragma unroll
    for (int p = 0; p < 4; p++) issue(u / 3, u % 3, p);
  bf16x8 fa0[2 * WI], fb0[WJ], fa1[2 * WI], fb1[WJ];
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // units 0, 1 landed (this wave's pieces); units 2, 3, 4 may fly
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, fa0, fb0);
  for (int b = 0; b < nblk; b++) {
    const int s = 2 * b;
    // ---- step 2b
    read_frags(s, 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa0, fb0, b + 1, 1, 0, b > 0 ? 4 : 0);             // unit 3b+4 = B of block b+1 (the prologue issued block 1's)
    // step 2b+1 needs unit 3b+2: behind its last piece only units 3b+3, 3b+4 were issued (8 pieces)
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // every wave has read the A lines of step 2b: unit 3b is free
    read_frags(s + 1, 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa1, fb1, b + 1, 2, 0, 3);                         // unit 3b+5 = A of step 2b+3, pieces 0..2
    // ---- step 2b+1
    read_frags(s + 1, 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa0, fb0, b + 1, 2, 3, 1);                         // ... piece 3
    // step 2b+2 needs units 3b+3, 3b+4: behind them only unit 3b+5 (4 pieces)
    asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // units 3b+1 (B) and 3b+2 are free
    read_frags(s + 2, 0, fa0, fb0);                             // (after the last block: a harmless read of a reloaded unit)
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(fa1, fb1, b + 2, 0, 0, 4);                         // unit 3b+6 = A of step 2b+4
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  static_assert(EPI == GEMM_SILU, "the whole-line kernel serves the gate_up product (its activation terms arrive interleaved from the norm launch)");
  silu_epilogue_transposed<DT>(acc, dma_lds, a, wv, lane, m0, n0, wm, wn);
}

// ---- 128 x 256 tile, 8 waves = 2 row halves x 2 column halves x 2 k halves, split-K slabs (the N = hidden products with K >> N: `down`; round 5) -----------------
// The 128 x 128 kernel below waits for its operands: 48 KB of lines per k64 step and tile, ~1920 cycles of the CU's memory pipeline for 1024 of its matrix pipe
// (DESIGN.md section 5).  A tile twice as wide shares its activation lines between two column tiles — 64 KB per k64 step for twice the outputs, a third fewer lines per
// output — at 128 accumulator registers per wave (64 x 128 of the output, as the 256 x 256 kernel's waves); the four wave tiles exist twice, for the first and the
// second half of every stage's k16 steps, and the two partial accumulators meet once through LDS.  128 tiles at S = 2048: blockIdx.z halves K (two fp32 slabs, summed
// in z order by the next row-wise launch).  Another fp32 summation order than the 128 x 128 kernel: results differ by ~1e-6.
template <int DT>
__global__ __launch_bounds__(512) void gemm_dma8n_kernel(const GemmArgs a) {
  constexpr int DBK = 64, NS = 2, TM = 128, TN = 256;
  constexpr int T_A = TM * DBK, T_B = TN * DBK;                           // 16-bit elements of a term tile / the weight tile
  constexpr int STAGE = 2 * T_A + T_B;                                    // 64 KB
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wv >> 2, wm = (wv >> 1) & 1, wn = wv & 1;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;
  int tile_m, tile_n;
  xcd_tile(a, tile_m, tile_n);
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  const int k_begin = (int)blockIdx.z * a.k_per, k_end = min(a.K, k_begin + a.k_per);

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // DMA map: a piece = 8 rows of 128 bytes; lane l -> row (l >> 3), LDS slot l & 7 <- global chunk slot ^ ((row >> 1) & 7).  Wave w: pieces w, w + 8 of each term tile,
  // pieces w, w + 8, w + 16, w + 24 of the weight tile: eight per stage
  const int prow = lane >> 3, pslot = lane & 7;
  const bf16_t* gsrc[8];
  unsigned ldst[8];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int piece = wv + 8 * p, row = piece * 8 + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    const size_t ga = (size_t)min(m0 + row, a.M - 1) * a.K + k_begin + chunk * 8;
    gsrc[2 * p] = a.A_hi + ga; gsrc[2 * p + 1] = a.A_lo + ga;
    ldst[2 * p] = (unsigned)(piece * 1024); ldst[2 * p + 1] = ldst[2 * p] + (unsigned)(T_A * 2);
  }
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int piece = wv + 8 * p, row = piece * 8 + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    gsrc[4 + p] = a.B + (size_t)min(n0 + row, a.N - 1) * a.K + k_begin + chunk * 8;
    ldst[4 + p] = (unsigned)(2 * T_A * 2 + piece * 1024);
  }
  auto frag = [&](const bf16_t* tile, int row, int kchunk) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(tile + row * DBK + ((kchunk ^ ((row >> 1) & 7)) << 3));
  };

  const int nk = (k_end - k_begin) / DBK;
#pragma unroll
  for (int q = 0; q < 8; q++) dma_1k(gsrc[q], lds_base + ldst[q]);
  for (int k = 0; k < nk; k++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // stage k landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                      // ... every wave's; and every wave is done reading stage k - 1
    const bool more = k + 1 < nk;
    const int nk0 = (k + 1) * DBK;
    const unsigned nsb = lds_base + (unsigned)(((k + 1) % NS) * STAGE * 2);
    const bf16_t* st = dma_lds + (size_t)(k % NS) * STAGE;
    const bf16_t *tAh = st, *tAl = st + T_A, *tB = st + 2 * T_A;
#pragma unroll
    for (int h = 0; h < 2; h++) {                                      // this wave's two k16 steps of the stage
      const int kchunk = (2 * kh + h) * 2 + (lane >> 5);
      bf16x8 fah[2], fal[2], fb[4];
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = frag(tB, wn * 128 + j * 32 + (lane & 31), kchunk);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int row = wm * 64 + i * 32 + (lane & 31);
        fah[i] = frag(tAh, row, kchunk);
        fal[i] = frag(tAl, row, kchunk);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[i][j] = mfma16<DT>(fal[i], fb[j], acc[i][j]);   // small term first
          acc[i][j] = mfma16<DT>(fah[i], fb[j], acc[i][j]);
          if (more && ((i * 4 + j) & 1)) { const int q = 4 * h + ((i * 4 + j) >> 1); dma_1k(gsrc[q] + nk0, nsb + ldst[q]); }      // one piece of the next stage per two blocks
        }
    }
  }

  // the second k half joins the first through LDS (the ring is idle: 4 x 32 KB = its 128 KB), lane to lane; the first half stores the slab
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* red = reinterpret_cast<float*>(dma_lds) + (size_t)(wv & 3) * 8 * 16 * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[((i * 4 + j) * 16 + r) * 64] = acc[i][j][r];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = n0 + wn * 128 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = acc[i][j][r] + red[((i * 4 + j) * 16 + r) * 64];
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.M && col < a.N) a.part[((size_t)blockIdx.z * a.M + row) * a.N + col] = v;
      }
    }
}

// ---- 128 x 128 tile, 8 waves with the K step split between them, three-stage ring (the N = hidden products: o_proj, down) ---------
// At S = 2048 these products have only 256 tiles of 128 x 128 — one per CU.  Four waves per tile leave one wave per SIMD (nothing hides a
// wave's fragment-read latency); 64-row tiles double the workgroups but read a fragment per MFMA.  Here the 128 x 128 tile gets EIGHT waves
// = 2 row halves x FOUR k quarters (round 5; rounds 2-4: 4 quadrants x 2 k halves, 12 fragment reads per 16 MFMAs): a wave owns 64 x 128 of the
// output (8 reads per 16 MFMAs) and takes ONE of the four k16 steps of every k64 stage, so a SIMD holds two waves, a stage is 48 KB with three
// stages in the ring, and the four partial accumulators meet through LDS after the K loop in a fixed order, (q0 + q2) + (q1 + q3); the last
// exchange hands each of the two surviving wave pairs one column half, so four waves run the epilogue as before.  down at S = 2048 (K = 8192): 161 -> 140 us;
// o_proj (K = 2048) 52 -> 52 (tools/probes/gemm_lab.hip, profiles/r05_prefill.txt).  Another fp32 summation order than the 4-wave kernel: results differ by ~1e-6.
template <int DT, int EPI, bool LO = true>
__global__ __launch_bounds__(512) void gemm_dma8k_kernel(const GemmArgs a) {
  constexpr int DBK = 64, CPR = 8, RPP = 8, TMN = 128, NS = 3;
  constexpr int STAGE = 3 * TMN * DBK;
  extern __shared__ __attribute__((aligned(1024))) bf16_t dma_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = wv >> 1, wm = wv & 1;
  int tile_m, tile_n;
  xcd_tile(a, tile_m, tile_n);
  const int m0 = tile_m * TMN, n0 = tile_n * TMN;
  const unsigned lds_base = (unsigned)(size_t)dma_lds;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int prow = lane / CPR, pslot = lane % CPR;
  const int k_begin = EPI == GEMM_PARTIAL ? (int)blockIdx.z * a.k_per : 0;
  const int k_end = EPI == GEMM_PARTIAL ? min(a.K, k_begin + a.k_per) : a.K;
  const bf16_t* gsrc[6];
  unsigned ldst[6];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int piece = wv + 8 * p, row = piece * RPP + prow;
    const int chunk = pslot ^ ((row >> 1) & 7);
    const size_t g = (size_t)min(m0 + row, a.M - 1) * a.K + chunk * 8;
    const int nb = min(n0 + row, a.N - 1);
    const size_t brow = EPI == GEMM_SILU ? (size_t)((nb & 1) ? a.inter : 0) + (size_t)(nb >> 1) : (size_t)nb;
    gsrc[3 * p] = a.A_hi + g + k_begin; gsrc[3 * p + 1] = a.A_lo + g + k_begin; gsrc[3 * p + 2] = a.B + brow * a.K + chunk * 8 + k_begin;
    ldst[3 * p] = (unsigned)(piece * 1024); ldst[3 * p + 1] = ldst[3 * p] + (unsigned)(TMN * DBK * 2); ldst[3 * p + 2] = ldst[3 * p] + (unsigned)(2 * TMN * DBK * 2);
  }
  auto issue_piece = [&](int q, int k0, int stage) { if (!LO && q % 3 == 1) return; dma_1k(gsrc[q] + k0, lds_base + (unsigned)(stage * STAGE * 2) + ldst[q]); };
  auto frag = [&](const bf16_t* tile, int row, int kchunk) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(tile + row * DBK + ((kchunk ^ ((row >> 1) & 7)) << 3));
  };

  const int nk = (k_end - k_begin) / DBK;
#pragma unroll
  for (int q = 0; q < 6; q++) issue_piece(q, 0, 0);
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < 6; q++) issue_piece(q, DBK, 1);
  }
  bf16x8 fah[2], fal[2], fb[4];
  for (int k = 0; k < nk; k++) {
    // stage k landed = only this wave's pieces of stage k+1 may fly
    if (k + 1 < nk) { if (LO) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = k + 2 < nk;
    const int nk0 = (k + 2) * DBK, nst = (k + 2) % NS;
    const bf16_t* st = dma_lds + (size_t)(k % NS) * STAGE;
    const bf16_t *tAh = st, *tAl = st + TMN * DBK, *tB = st + 2 * TMN * DBK;
    const int kchunk = kq * 2 + (lane >> 5);               // this wave's k16 step of the stage
#pragma unroll
    for (int j = 0; j < 4; j++) fb[j] = frag(tB, j * 32 + (lane & 31), kchunk);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = wm * 64 + i * 32 + (lane & 31);
      fah[i] = frag(tAh, row, kchunk);
      if (LO) fal[i] = frag(tAl, row, kchunk);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (LO) acc[i][j] = mfma16<DT>(fal[i], fb[j], acc[i][j]);   // small term first
        acc[i][j] = mfma16<DT>(fah[i], fb[j], acc[i][j]);
        if (more && (i * 4 + j) < 6) issue_piece(i * 4 + j, nk0, nst);     // one DMA per two MFMAs
      }
  }

  // (q0 + q2) + (q1 + q3): quarters 2, 3 -> LDS -> added by quarters 0, 1 (same row half); then q0 / q1 exchange column halves
  __builtin_amdgcn_s_barrier();
  float* red = reinterpret_cast<float*>(dma_lds);
  if (kq >= 2) {
    float* dst = red + (size_t)(wv - 4) * 8 * 16 * 64;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((i * 4 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (kq >= 2) return;
  {
    const float* src = red + (size_t)wv * 8 * 16 * 64;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] += src[((i * 4 + j) * 16 + r) * 64 + lane];
  }
  __syncthreads();                                         // (the four surviving waves) quarters 2, 3 have been read: the buffer is free
  // q0 finishes columns 0..63 (j 0, 1), q1 columns 64..127 (j 2, 3): each hands the other the half it does not finish.  (Written as two branches with
  // constant block indices: a run-time index into the accumulator array would send it to scratch memory.)
  auto put_half = [&](auto J0) {
    constexpr int j0 = decltype(J0)::value;
    float* dst = red + (size_t)wv * 4 * 16 * 64;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++)
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((i * 2 + jj) * 16 + r) * 64 + lane] = acc[i][j0 + jj][r];
  };
  if (kq == 0) put_half(std::integral_constant<int, 2>{}); else put_half(std::integral_constant<int, 0>{});
  __syncthreads();
  const float* other = red + (size_t)(wv ^ 2) * 4 * 16 * 64;
  f32x16 fin[2][2];                                          // the wave's final 64 x 64 half
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float o = other[((i * 2 + jj) * 16 + r) * 64 + lane];
        fin[i][jj][r] = kq == 0 ? acc[i][jj][r] + o : o + acc[i][2 + jj][r];      // (q0 + q2) + (q1 + q3) on both sides
      }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int col = n0 + kq * 64 + jj * 32 + (lane & 31);
      if (EPI == GEMM_SILU) {
        silu_block_store<DT>(fin[i][jj], lane, col, m0 + wm * 64 + i * 32, a);
        continue;
      }
      if (col >= a.N) continue;
      if (EPI == GEMM_PARTIAL) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < a.M) a.part[((size_t)blockIdx.z * a.M + row) * a.N + col] = fin[i][jj][r];
        }
        continue;
      }
      const float bv = a.bias ? elem_to_f32<DT>(a.bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const float v = fin[i][jj][r] + bv;
        float* dst = a.C + (size_t)row * a.ldc + col;
        *dst = (EPI == GEMM_RESIDUAL) ? (*dst + v) : v;
      }
    }
}

}  // namespace tgx
