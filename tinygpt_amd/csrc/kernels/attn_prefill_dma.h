// attn_prefill_dma.h — causal GQA flash attention of a prompt at head_dim 64 with its K / V tiles delivered by LDS-DMA and the next tile's scores
// in flight under the current tile's softmax (round 5).  Same math and the same operand layouts in the matrix instructions as attn_prefill_kernel
// (kernels/prefill.h: S^T = K.Q^T so that a lane owns one query column, O^T += V^T.P^T with V^T through ds_read_b64_tr_b16, Q and P as two 16-bit terms,
// fp32 online softmax in base 2); what changes is the schedule around them.
//
// Why.  attn_prefill_kernel walks a tile as  barrier - ds_write K, V - barrier - fetch - QK^T - softmax - PV, every part waiting for the one before: ~3900
// cycles per 64-key tile and wave, of which 1024 are matrix instructions (MfmaUtil 24 %), and a launch lasts as long as its heaviest workgroup's chain of
// tiles (32 at S = 2048).  Here
//   * K and V tiles go memory -> LDS without passing through registers (global_load_lds_dwordx4 into rings of three tiles each): no ds_write, no second
//     barrier, ONE counted wait per tile.  Rows are unpadded 128-byte lines (a DMA piece is 1 KiB = 8 rows); bank conflicts are avoided by an XOR of the
//     16-byte chunk index, applied on the global side of the DMA and again in the fragment addresses: K by (row >> 1) & 7 (the GEMMs' swizzle: the 16 rows
//     of a ds_read_b128 lane group fall on 16 bank quads), V by ((row >> 1) & 1) << 2 (the four key rows of a transposing read fall on four bank quarters);
//   * the K ring runs one tile ahead of the V ring: while the wave works on the softmax of tile t (VALU) its QK^T of tile t + 1 is already in the matrix
//     pipe, and PV of tile t follows.  Issue group G(s) = {K(s + 1), V(s)}; at step t the wave waits for G(t) (issued two steps earlier), meets the others,
//     issues G(t + 2) into the slots that K(t) and V(t - 1) have left, and computes.
//   * tiles wholly below the wave's diagonal take a branch-free body (no masks, no visibility tests): one basic block per step, which the scheduler can
//     interleave; the last one or two tiles of a wave go through the general body.
// Reference semantics: Attention.h:103-112 (flashAttention, isCausal) over the cache offsets of CacheManager.h:24-51.
#pragma once
#include "gemm_dma.h"

namespace tgx {

// PAGED (round 6): the workgroup's slice of the sequence's block table goes to LDS first; a DMA piece's source row then takes one LDS read (a tile of 64
// keys lies inside one 128-token page, the clamped rows of the last tile in that page or an earlier one).
template <int DT, bool PAGED = false>
__global__ __launch_bounds__(256, 2) void attn_prefill_dma_kernel(const AttnPrefillArgs a) {
  constexpr int HD = 64, KS = HD / 16, NB = HD / 32, CH = HD / 8, NS = 3;
  constexpr int TILE = 64 * HD;                 // 16-bit elements of one K or V tile
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(1024))) bf16_t smem[];       // K ring (NS tiles) | V ring (NS tiles); the output rows pass through it at the end
  const bf16_t* const sKr = smem;
  const bf16_t* const sVr = smem + NS * TILE;
  const unsigned lds_k = (unsigned)(size_t)smem, lds_v = lds_k + (unsigned)(NS * TILE * 2);

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5, ql = lane & 31;
  const int h = a.heavy_first ? blockIdx.x : blockIdx.y, G = a.heads / a.kv_heads, kvh = h / G;
  const int qd = a.heads * HD;
  const int qblk = a.heavy_first ? (int)gridDim.y - 1 - (int)blockIdx.y
                                 : ((a.qblk_mirror && h >= a.heads / 2) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
  const int q0 = qblk * 128 + wv * 32;                 // first query of this wave
  const int qi = q0 + ql;                              // this lane's query
  const bool qvalid = qi < a.S;
  const int qpos = a.past + qi;

  bf16x8 qh[KS], qlo[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    u32x4 vh = u32x4{0u, 0u, 0u, 0u}, vl = vh;
    if (qvalid) {
      const size_t o = (size_t)qi * qd + (size_t)h * HD + kk * 16 + 8 * hh;
      vh = *reinterpret_cast<const u32x4*>(a.q_hi + o);
      vl = *reinterpret_cast<const u32x4*>(a.q_lo + o);
    }
    qh[kk] = __builtin_bit_cast(bf16x8, vh);
    qlo[kk] = __builtin_bit_cast(bf16x8, vl);
  }
  f32x16 oacc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[b][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* kbase = a.k_cache + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD);
  const bf16_t* vbase = a.v_cache + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD);
  const int wg_last_pos = a.past + min(qblk * 128 + 127, a.S - 1);   // keys beyond it are never attended by this workgroup
  __shared__ int stbl[PAGED ? 1024 : 1];
  if constexpr (PAGED) {
    for (int i = tid; i <= (wg_last_pos >> KV_BLOCK_SHIFT); i += 256) stbl[i] = a.blk_tbl[i];
    __syncthreads();
  }
  auto key_off = [&](int key) -> size_t {      // element offset of a key's row from kbase / vbase
    if constexpr (PAGED) return (((size_t)stbl[key >> KV_BLOCK_SHIFT] * a.kv_heads + kvh) * KV_BLOCK + (key & (KV_BLOCK - 1))) * (size_t)HD;
    else return (size_t)key * HD;
  };
  const int n_kt = wg_last_pos / 64 + 1;
  const bool wave_live = q0 < a.S;
  const int wave_last_pos = a.past + min(q0 + 31, a.S - 1);
  const int t_last = wave_live ? min(wave_last_pos / 64, n_kt - 1) : -1;               // last tile with a key visible to one of the wave's queries
  const int t_full = (wave_live && q0 + 31 < a.S) ? (a.past + q0 + 1) / 64 : 0;        // tiles [0, t_full) lie wholly below the diagonal of every query of the wave
  const float qs = a.scale * LOG2E;

  // DMA map: piece pc of a tile = its rows 8 pc .. 8 pc + 7; wave w moves pieces w and w + 4 of the K tile and of the V tile of a group.  Lane l -> row (l >> 3) of the
  // piece, LDS slot l & 7, global chunk = slot ^ swizzle(row).  Keys past the workgroup's range are clamped to its last key (a written row; masked by index below).
  const int prow = lane >> 3, pslot = lane & 7;
  int prw[2], kof[2], vof[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    prw[j] = 8 * (wv + 4 * j) + prow;
    kof[j] = (pslot ^ ((prw[j] >> 1) & 7)) * 8;
    vof[j] = (pslot ^ (((prw[j] >> 1) & 1) << 2)) * 8;
  }
  auto issue_k = [&](int s, int slot) __attribute__((always_inline)) {
    const unsigned dst = lds_k + (unsigned)(slot * TILE * 2);
#pragma unroll
    for (int j = 0; j < 2; j++) dma_1k(kbase + key_off(min(s * 64 + prw[j], wg_last_pos)) + kof[j], dst + (unsigned)((wv + 4 * j) * 1024));
  };
  auto issue_v = [&](int s, int slot) __attribute__((always_inline)) {
    const unsigned dst = lds_v + (unsigned)(slot * TILE * 2);
#pragma unroll
    for (int j = 0; j < 2; j++) dma_1k(vbase + key_off(min(s * 64 + prw[j], wg_last_pos)) + vof[j], dst + (unsigned)((wv + 4 * j) * 1024));
  };

  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; r++) zero16[r] = 0.f;
  const int kswz = (ql >> 1) & 7;                      // the K fragment rows are 32 sub + ql: (row >> 1) & 7 does not depend on sub
  const int vb_flip = (lane >> 3) & 1;                 // V: rows kloc + ((lane & 15) >> 2), kloc a multiple of 4 -> ((row >> 1) & 1) = bit 3 of the lane

  // S^T sub-tiles of tile s: sacc[sub][r] = raw score (q.k) of key 64 s + 32 sub + (r&3) + 8 (r>>2) + 4 hh for this lane's query.  The two sub-tiles' chains alternate.
  auto qk_tile = [&](int slot, f32x16& s0, f32x16& s1) __attribute__((always_inline)) {
    const bf16_t* sK = sKr + slot * TILE;
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
      const int co = ((kk * 2 + hh) ^ kswz) << 3;
      const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(sK + ql * HD + co);
      const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(sK + (32 + ql) * HD + co);
      s0 = mfma16<DT>(fk0, qlo[kk], kk == 0 ? zero16 : s0);
      s1 = mfma16<DT>(fk1, qlo[kk], kk == 0 ? zero16 : s1);
      s0 = mfma16<DT>(fk0, qh[kk], s0);
      s1 = mfma16<DT>(fk1, qh[kk], s1);
    }
  };
  // online softmax of tile t on its scores (in place: s0, s1 become the probabilities); MASKED: the causal mask / padded queries / keys past the wave's range
  auto softmax_tile = [&](int t, f32x16& s0, f32x16& s1, auto MASKED) __attribute__((always_inline)) {
    if constexpr (decltype(MASKED)::value) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int key = t * 64 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (!(qvalid && key <= qpos)) s0[r] = -INFINITY;     // isCausal (Attention.h:108) with the cache offset
        if (!(qvalid && key + 32 <= qpos)) s1[r] = -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;          // qs > 0: scaling commutes with the maximum
    const float m_new = fmaxf(m_run, mx);
    const bool dead = m_new == -INFINITY;               // padded query: nothing attended yet
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
    const float neg_m = dead ? 0.f : -m_new;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s0[r], qs, neg_m));   // exp2(-inf) = 0 for masked entries
      s0[r] = p;
      sum += p;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s1[r], qs, neg_m));
      s1[r] = p;
      sum += p;
    }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (__any(alpha != 1.f)) {                   // the running maximum settles after the first tiles: most steps leave O as it is
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[b][r] *= alpha;
    }
  };
  // O^T += V^T . P^T over the 16-key step (sub, s2) of a tile; B-operand element j is register 8 s2 + j of the sub-tile's probabilities
  auto pv_step = [&](const bf16_t* sV, const f32x16& p, int sub, int s2) __attribute__((always_inline)) {
    unsigned int wh[4], wl[4];
    if constexpr (DT == DT_BF16) {          // pairs: one packed convert per term, the residual from the packed hi word itself
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float p0 = p[8 * s2 + 2 * j], p1 = p[8 * s2 + 2 * j + 1];
        wh[j] = pack_bf16(p0, p1);
        wl[j] = pack_bf16(p0 - bf16_lo(wh[j]), p1 - bf16_hi(wh[j]));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        bf16_t ph, pl;
        split16<DT>(p[8 * s2 + j], ph, pl);
        if (j & 1) { wh[j >> 1] |= (unsigned int)ph << 16; wl[j >> 1] |= (unsigned int)pl << 16; }
        else { wh[j >> 1] = ph; wl[j >> 1] = pl; }
      }
    }
    const bf16x8 fph = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    const bf16x8 fpl = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
    const int kloc = 32 * sub + 16 * s2 + 4 * hh;     // tile-local key of element 0; elements 4..7 are 8 keys further
    bf16x8 fv[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      // ds_read_b64_tr_b16: the 16 lanes of a group point at the [4 keys][16 dims] block (lane i: key i >> 2, dims 4 (i & 3)..+3) and lane i receives column i.
      // dims 32 b + (lane & 16) + 4 (lane & 3): chunk 4 b + 2 [lane & 16] + ((lane & 3) >> 1), swizzled by flipping b for the key rows 2, 3 of the four
      const bf16_t* vblk = sV + (kloc + ((lane & 15) >> 2)) * HD + 32 * (b ^ vb_flip) + (lane & 16) + 4 * (lane & 3);
      const u32x2 v0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk)));
      const u32x2 v1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 8 * HD)));
      fv[b] = __builtin_bit_cast(bf16x8, u32x4{v0[0], v0[1], v1[0], v1[1]});
    }
#pragma unroll
    for (int b = 0; b < NB; b++) oacc[b] = mfma16<DT>(fv[b], fpl, oacc[b]);
#pragma unroll
    for (int b = 0; b < NB; b++) oacc[b] = mfma16<DT>(fv[b], fph, oacc[b]);
  };
  auto pv_tile = [&](int slot, const f32x16& s0, const f32x16& s1) __attribute__((always_inline)) {
    const bf16_t* sV = sVr + slot * TILE;
    pv_step(sV, s0, 0, 0); pv_step(sV, s0, 0, 1); pv_step(sV, s1, 1, 0); pv_step(sV, s1, 1, 1);
  };

  // one step: tile t's softmax and PV, tile t + 1's scores.  SLOT = t % 3 at compile time (the loop is unrolled over the ring): every LDS offset is an immediate
  f32x16 c0, c1, n0, n1;
  auto step = [&](int t, auto SLOT) __attribute__((always_inline)) {
    constexpr int sl = decltype(SLOT)::value, sl1 = (sl + 1) % NS, sl2 = (sl + 2) % NS;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // G(t) = {K(t + 1), V(t)} landed (this wave's pieces); G(t + 1) may fly
    __builtin_amdgcn_s_barrier();                         // ... every wave's; and every wave is past QK(t) and PV(t - 1)
    issue_k(t + 3, sl);
    issue_v(t + 2, sl2);
    if (t + 1 < t_full) {                                 // tiles t and t + 1 wholly visible to every query of the wave: no masks, no tests
      qk_tile(sl1, n0, n1);
      softmax_tile(t, c0, c1, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);                  // the V fragments are not fetched across the softmax (registers)
      pv_tile(sl, c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (t + 1 <= t_last) qk_tile(sl1, n0, n1);
      if (t <= t_last) {
        if (t < t_full) softmax_tile(t, c0, c1, std::false_type{}); else softmax_tile(t, c0, c1, std::true_type{});
        pv_tile(sl, c0, c1);
      }
    }
    c0 = n0; c1 = n1;
  };

  issue_k(0, 0);
  issue_k(1, 1); issue_v(0, 0);
  issue_k(2, 2); issue_v(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // K(0)
  __builtin_amdgcn_s_barrier();
  qk_tile(0, c0, c1);          // unconditional: the compiler's wait for the q registers (it does not see the DMA's vmcnt traffic) must sit HERE, not at their first use inside the loop, where it would drain the rings at every step
  for (int t = 0; t < n_kt; t += 3) {
    step(t, std::integral_constant<int, 0>{});
    if (t + 1 < n_kt) step(t + 1, std::integral_constant<int, 1>{});      // (workgroup-uniform)
    if (t + 2 < n_kt) step(t + 2, std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped groups past the last tile have landed too: the rings are quiet

  // normalise and emit as hi / lo 16-bit pairs (the o_proj GEMM's A operand) through an LDS transpose, whole rows per store (as attn_prefill_kernel)
  __syncthreads();                              // every wave is done with the last K / V tile
  constexpr int RS = HD + 4;                    // staged row stride in 16-bit elements
  static_assert(4 * 32 * RS <= 2 * NS * TILE, "output staging exceeds the rings");
  bf16_t* const wrow = smem + wv * 32 * RS;
  const float inv_l = qvalid ? 1.0f / l_run : 0.f;
  unsigned int whi[NB * 8], wlo[NB * 8];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float v0 = oacc[b][r] * inv_l, v1 = oacc[b][r + 1] * inv_l;
      bf16_t h0, l0, h1, l1;
      split16<DT>(v0, h0, l0);
      split16<DT>(v1, h1, l1);
      whi[b * 8 + (r >> 1)] = (unsigned int)h0 | ((unsigned int)h1 << 16);
      wlo[b * 8 + (r >> 1)] = (unsigned int)l0 | ((unsigned int)l1 << 16);
    }
#pragma unroll
  for (int term = 0; term < 2; term++) {
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int d = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hh;                 // dims d, d + 1
        *reinterpret_cast<unsigned int*>(wrow + ql * RS + d) = term ? wlo[b * 8 + (r >> 1)] : whi[b * 8 + (r >> 1)];
      }
    bf16_t* const dst = term ? a.o_lo : a.o_hi;
#pragma unroll
    for (int i = 0; i < HD / 16; i++) {          // 32 rows x HD / 8 chunks of 16 bytes over 64 lanes
      const int c = lane + 64 * i, row = c / CH, cc = c - row * CH;
      const u32x2 p0 = *reinterpret_cast<const u32x2*>(wrow + row * RS + cc * 8);
      const u32x2 p1 = *reinterpret_cast<const u32x2*>(wrow + row * RS + cc * 8 + 4);
      if (q0 + row < a.S)
        *reinterpret_cast<u32x4*>(dst + (size_t)(q0 + row) * qd + (size_t)h * HD + cc * 8) = u32x4{p0[0], p0[1], p1[0], p1[1]};
    }
  }
}

}  // namespace tgx
