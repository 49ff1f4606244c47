// attn_decode_mfma.h — single-query GQA attention over long contexts on the matrix cores.
//
// The VALU kernel of attn_decode.h spends ~120 instructions per 2 KB of K+V and wave (dot product, two exp2, P.V update per query head):
// from a few thousand keys on it is bound by that arithmetic, not by HBM (Mistral-7B at 24k keys: 3.0 TB/s).  Here the G query heads of a kv
// group are the narrow operand of the prefill attention's MFMA scheme (prefill.h attn_prefill_kernel): a wave computes S^T = K . Q^T for 64
// keys x 32 "query columns" (the G heads of the group, zero-padded), keeps the online softmax in registers (lane = query column) and
// accumulates O^T += V^T . P^T.  What differs from the prefill kernel: every wave walks its OWN key blocks (blocks of 64 keys are dealt
// round-robin to splits x waves), so K fragments come straight from global memory (a fragment row is 16 contiguous bytes of a key row) and
// the V^T tile is wave-private in LDS — the main loop has no workgroup barrier.  The four waves' partials meet once in LDS and leave as one
// (o[hd], m, l) record per query head and split, merged by attn_combine_kernel exactly like the VALU kernel's.
//
// STATUS (round 2): the first version (V^T staged through LDS with lane exchanges and 4-byte stores) was correct but slower than the VALU
// kernel (Mistral-7B at 24k keys: 40 vs 33 us per layer).  With V left [key][d] in the wave's tile and read by ds_read_b64_tr_b16, paired
// probability packing, chains started from a constant zero and O rescaled only when a maximum moved, it wins from ~6k keys at head_dim 64 with
// 8 kv heads (Llama-3.2-1B: 23.3 -> 18.3 us per layer at 30k) and from ~14k at head_dim 128 or 2 kv heads (Mistral-7B 39.5 -> 31.4 at 30k,
// Qwen2.5-0.5B 18.1 -> 14.4): the shim switches at those contexts (attn_mfma_threshold; option attn.mfma_min overrides;
// profiles/r02_attn_long.txt).  Logits within 3e-5 of the VALU kernel, ids equal (tools/attn_long.py, tests).
//
// Roofline: HBM — 2 * kv_heads * (T+1) * hd * 2 bytes per launch.  MFMA work: 2 x (QK^T + PV) x 32 / G of the useful flops — negligible.
#pragma once
#include "attn_decode.h"
#include "prefill.h"

namespace tgx {

template <int HD, int NW = 4>
__host__ __device__ constexpr size_t attn_mfma_lds_bytes() { return (size_t)NW * 64 * (HD + 32) * 2; }

// RAW form: room for the finished q heads (fp32, at most 8 per kv head) and this step's k / v rows (storage dtype) behind the waves' V tiles
constexpr int ATTN_RAW_GMAX = 8;
template <int HD, int NW = 4>
__host__ __device__ constexpr size_t attn_mfma_raw_lds_bytes() { return attn_mfma_lds_bytes<HD, NW>() + (size_t)ATTN_RAW_GMAX * HD * 4 + (size_t)2 * HD * 2; }

// NW = waves per workgroup.  a.direct (round 3, batches): ONE workgroup holds all the keys of its (row, kv head) — no split records, no combine
// launch: the merged rows are normalised and written straight into the o_proj input.
// RAW (with a.direct): the workgroup first finishes its own slice of the QKV product (AttnArgs.raw_*: slab sums + bias, q / k norm at head_dim 128, RoPE,
// cache append) — the G q heads go to LDS as fp32 and feed the Q^T fragments, the new k / v rows go to the cache AND to LDS, from where the lanes that
// hold key `pos` take them (the global copy was stored by this same workgroup a moment ago: never read back in this launch).  Same arithmetic, in the
// same order, as rope_kv_rows_kernel (skinny.h): the two forms are bit-identical.
// LOOKAHEAD: the next block's K / V loads are issued before the current block's arithmetic (a second register set: 309 registers at head_dim 64 = one
// wave per SIMD).  Without it the RAW form fits 256: two workgroups per CU, for batches with more (row, kv head) workgroups than CUs.
// PAGED (round 6): K / V through the row's block table (common.h kv_paged_off) — a block of 64 keys lies inside one 128-token page, so a block's loads
// share ONE table entry, read through the scalar cache (the block index is wave-uniform).
template <int DT, int HD, int NW = 4, bool RAW = false, bool LOOKAHEAD = true, bool PAGED = false>
__global__ __launch_bounds__(64 * NW) void attn_decode_mfma_kernel(const AttnArgs a) {
  typedef elem_t<DT> E;
  constexpr int LV = HD + 32;                 // 16-bit row stride of the wave's V tile ([key][d]; read with the transposing LDS read, as attn_prefill_kernel)
  constexpr int KS = HD / 16;                 // MFMA k-steps over the head dimension
  constexpr int NB = HD / 32;                 // 32-row output-dim blocks
  constexpr int CH = HD / 8;                  // 16-byte chunks per key row
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) bf16_t amf_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5, ql = lane & 31;
  bf16_t* sV = amf_lds + (size_t)wv * 64 * LV;         // this wave's V tile [64 keys][LV]
  const int nsp = a.direct ? 1 : a.nsplit;
  const int kvh = blockIdx.x / nsp, sp = blockIdx.x - kvh * nsp;
  const int G = a.gfull;
  const float* q_row = a.q + blockIdx.y * a.q_stride;
  const E* kbase = static_cast<const E*>(a.k_cache) + (PAGED ? (size_t)0 : blockIdx.y * a.kv_stride + (size_t)kvh * a.max_ctx * HD);
  const E* vbase = static_cast<const E*>(a.v_cache) + (PAGED ? (size_t)0 : blockIdx.y * a.kv_stride + (size_t)kvh * a.max_ctx * HD);
  const int* tbl = PAGED ? a.blk_tbl + blockIdx.y * a.tbl_stride : nullptr;
  // element offset of key `t` (wave-uniform) from kbase / vbase
  auto key_off = [&](int t) -> size_t {
    if constexpr (PAGED) return kv_paged_off(tbl, a.kv_heads, kvh, __builtin_amdgcn_readfirstlane(t), HD);
    else return (size_t)t * HD;
  };
  float* part_row = a.part + blockIdx.y * a.part_stride;
  const int n_keys = a.pos[blockIdx.y] + 1;
  const bool qvalid = ql < G;
  const int nblk = (n_keys + 63) >> 6;
  // register sets of one block's loads: V rows (chunk c = lane + 64 i -> key row c / CH, 16-byte column c % CH) and the K fragments
  // (row = key 32 sub + ql, 8 d at 16 kk + 8 hh); the NEXT block's loads are issued before the current block's arithmetic
  constexpr bool PREF = HD == 64 && LOOKAHEAD;   // head_dim 128: a second register set spills (measured 40 -> 68 us per layer at 24k keys)
  u32x4 vvr[CH], kfr[2][KS], vnx[PREF ? CH : 1], knx[2][PREF ? KS : 1];
  auto load_block = [&](int blk, u32x4* vdst, u32x4 (*kdst)[KS]) {
    const int key0 = blk * 64, last = n_keys - 1 - key0;      // (>= 0: a block that is walked holds a key of the context)
    const size_t o0 = key_off(key0);
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const int c = lane + 64 * i, row = c / CH, kc = c - row * CH;
      vdst[i] = *reinterpret_cast<const u32x4*>(vbase + o0 + (size_t)min(row, last) * HD + kc * 8);     // clamped inside the context; masked by P = 0
    }
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const E* krow = kbase + o0 + (size_t)min(32 * sub + ql, last) * HD + 8 * hh;
#pragma unroll
      for (int kk = 0; kk < KS; kk++) kdst[sub][kk] = *reinterpret_cast<const u32x4*>(krow + kk * 16);
    }
  };
  const int blk0 = sp * NW + wv, bstep = nsp * NW;
  if (PREF && blk0 < nblk) load_block(blk0, vvr, kfr);     // in flight during the Q fragments' (RAW: the QKV finish's) loads
  float* sQ = reinterpret_cast<float*>(amf_lds + (size_t)NW * 64 * LV);      // RAW: [G][HD] fp32
  bf16_t* sKn = reinterpret_cast<bf16_t*>(sQ + ATTN_RAW_GMAX * HD);           // RAW: this step's key row [HD], storage dtype
  bf16_t* sVn = sKn + HD;                                                     //      and value row
  if constexpr (RAW) {
    constexpr int half = HD / 2;
    const int pos = n_keys - 1, row = blockIdx.y;
    const int qd = a.heads * HD, kvd = a.kv_heads * HD, N = qd + 2 * kvd;
    const size_t slab = (size_t)a.raw_rows * N;
    auto value = [&](int idx) -> float {
      if (a.raw_qkv) return a.raw_qkv[(size_t)row * N + idx];
      const float* src = a.raw_part + (size_t)row * N + idx;
      float t[16];
#pragma unroll
      for (int z = 0; z < 16; z++) t[z] = z < a.raw_nsplit ? src[z * slab] : 0.f;    // all slabs in flight together; summed in z order
      float v = t[0];
#pragma unroll
      for (int z = 1; z < 16; z++) v += t[z];
      for (int z = 16; z < a.raw_nsplit; z++) v += src[z * slab];
      if (a.raw_bias) v += elem_to_f32<DT>(static_cast<const E*>(a.raw_bias)[idx]);
      return v;
    };
    // G + 2 head vectors (the group's q heads, k, v) x HD / 2 rotation pairs; a vector's pairs are consecutive threads (half a wave at head_dim 64, a wave at 128)
    for (int it = tid; it < (G + 2) * half; it += 64 * NW) {
      const int vec = it / half, p = it - vec * half;
      const int col = vec < G ? (kvh * G + vec) * HD : (vec == G ? qd + kvh * HD : qd + kvd + kvh * HD);
      float x0 = value(col + p), x1 = value(col + p + half);
      if constexpr (HD == 128) {       // Qwen3: per-head RMSNorm of q and k over head_dim (one wave = one vector here)
        if (a.q_norm_w != nullptr && vec <= G) {
          const float ss = wave_sum(head_sq_pair(x0, x1));
          const float inv = head_rms_inv(ss, HD, a.eps);
          const E* w = static_cast<const E*>(vec < G ? a.q_norm_w : a.k_norm_w);
          x0 = elem_to_f32<DT>(w[p]) * (x0 * inv);
          x1 = elem_to_f32<DT>(w[p + half]) * (x1 * inv);
        }
      }
      if (vec <= G) {
        const float cs = a.rope_cos[(size_t)pos * half + p], sn = a.rope_sin[(size_t)pos * half + p];
        rope_rotate_pair(x0, x1, cs, sn);
      }
      if (vec < G) { sQ[vec * HD + p] = x0; sQ[vec * HD + p + half] = x1; }
      else {
        E* cache = const_cast<E*>(vec == G ? kbase : vbase) + key_off(pos);         // KVCacheManager::append
        E* stage = reinterpret_cast<E*>(vec == G ? sKn : sVn);
        const E e0 = f32_to_elem<DT>(x0), e1 = f32_to_elem<DT>(x1);
        cache[p] = e0; cache[p + half] = e1;
        stage[p] = e0; stage[p + half] = e1;
      }
    }
    __syncthreads();
  }

  // Q^T fragments (B operand: column = query head ql, 8 consecutive d at 16 kk + 8 hh), fp32 -> hi / lo terms
  bf16x8 qh[KS], qlo[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    unsigned int wh[4] = {0u, 0u, 0u, 0u}, wl[4] = {0u, 0u, 0u, 0u};
    if (qvalid) {
      const f32x4* qp = RAW ? reinterpret_cast<const f32x4*>(sQ + ql * HD + kk * 16 + 8 * hh)
                            : reinterpret_cast<const f32x4*>(q_row + (size_t)(kvh * G + ql) * HD + kk * 16 + 8 * hh);
      const f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        bf16_t h, l;
        split16<DT>(j < 4 ? q0[j] : q1[j - 4], h, l);
        if (j & 1) { wh[j >> 1] |= (unsigned int)h << 16; wl[j >> 1] |= (unsigned int)l << 16; }
        else { wh[j >> 1] = h; wl[j >> 1] = l; }
      }
    }
    qh[kk] = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    qlo[kk] = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
  }
  f32x16 oacc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[b][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float qs = a.scale * LOG2E;

  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; r++) zero16[r] = 0.f;
  for (int blk = blk0; blk < nblk; blk += bstep) {     // wave-uniform trip count
    const int key0 = blk * 64;
    const bool has_next = PREF && blk + bstep < nblk;
    if constexpr (PREF) { if (has_next) load_block(blk + bstep, vnx, knx); }
    else load_block(blk, vvr, kfr);
    if constexpr (RAW) {
      if (key0 + 64 >= n_keys) {      // the block that holds this step's own key: rows from it on (clamped copies, masked below) come from LDS
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
          if (key0 + 32 * sub + ql >= n_keys - 1) {
#pragma unroll
            for (int kk = 0; kk < KS; kk++) kfr[sub][kk] = *reinterpret_cast<const u32x4*>(sKn + kk * 16 + 8 * hh);
          }
#pragma unroll
        for (int i = 0; i < CH; i++) {
          const int c = lane + 64 * i, row = c / CH, kc = c - row * CH;
          if (key0 + row >= n_keys - 1) vvr[i] = *reinterpret_cast<const u32x4*>(sVn + kc * 8);
        }
      }
    }
    f32x16 sacc[2];
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const int kb = key0 + 32 * sub;
#pragma unroll
      for (int kk = 0; kk < KS; kk++) {        // always both sub-tiles (rows past the context are clamped loads, masked below)
        const bf16x8 fk = __builtin_bit_cast(bf16x8, kfr[sub][kk]);
        sacc[sub] = mfma16<DT>(fk, qlo[kk], kk == 0 ? zero16 : sacc[sub]);
        sacc[sub] = mfma16<DT>(fk, qh[kk], sacc[sub]);
      }
      if (kb + 32 > n_keys) {      // the context ends inside (or before) this sub-tile
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= n_keys) sacc[sub][r] = -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[sub][r]);
    }
    // V into the wave's LDS tile as it is ([key][d]); the PV step reads it transposed (ds_read_b64_tr_b16)
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const int c = lane + 64 * i, row = c / CH, kc = c - row * CH;
      *reinterpret_cast<u32x4*>(&sV[row * LV + kc * 8]) = vvr[i];
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
    const float m_new = fmaxf(m_run, mx);                 // finite: every block holds at least one key of the context
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float sum = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[sub][r], qs, -m_new));   // exp2(-inf) = 0 for masked keys
        sacc[sub][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[b][r] *= alpha;
    }
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        unsigned int wh[4], wl[4];
        if constexpr (DT == DT_BF16) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float p0 = sacc[sub][8 * s2 + 2 * j], p1 = sacc[sub][8 * s2 + 2 * j + 1];
            wh[j] = pack_bf16(p0, p1);
            wl[j] = pack_bf16(p0 - bf16_lo(wh[j]), p1 - bf16_hi(wh[j]));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            bf16_t ph, pl;
            split16<DT>(sacc[sub][8 * s2 + j], ph, pl);
            if (j & 1) { wh[j >> 1] |= (unsigned int)ph << 16; wl[j >> 1] |= (unsigned int)pl << 16; }
            else { wh[j >> 1] = ph; wl[j >> 1] = pl; }
          }
        }
        const bf16x8 fph = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
        const bf16x8 fpl = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
        const int kloc = 32 * sub + 16 * s2 + 4 * hh;
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const bf16_t* vblk = &sV[(kloc + ((lane & 15) >> 2)) * LV + 32 * b + (lane & 16) + 4 * (lane & 3)];
          const u32x2 v0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk)));
          const u32x2 v1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 8 * LV)));
          const bf16x8 fv = __builtin_bit_cast(bf16x8, u32x4{v0[0], v0[1], v1[0], v1[1]});
          oacc[b] = mfma16<DT>(fv, fpl, oacc[b]);
          oacc[b] = mfma16<DT>(fv, fph, oacc[b]);
        }
      }
    }
    if constexpr (PREF) {
      if (has_next) {
#pragma unroll
        for (int i = 0; i < CH; i++) vvr[i] = vnx[i];
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int kk = 0; kk < KS; kk++) kfr[sub][kk] = knx[sub][kk];
      }
    }
  }

  // the four waves meet in LDS (one (o[HD], m, l) record per wave and query head), merged in wave order — the VALU kernel's step 2
  __syncthreads();                                       // every wave is done with its V tile: the space is reused
  float* red = reinterpret_cast<float*>(amf_lds);        // [NW][G][HD + 4]
  if (qvalid) {
    float* dst = red + ((size_t)wv * G + ql) * (HD + 4);
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) dst[32 * b + (r & 3) + 8 * (r >> 2) + 4 * hh] = oacc[b][r];
    if (hh == 0) { dst[HD] = m_run; dst[HD + 1] = l_run; }
  }
  __syncthreads();
  for (int idx = tid; idx < G * HD; idx += 64 * NW) {
    const int g = idx / HD, d = idx - g * HD;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) M = fmaxf(M, red[((size_t)w * G + g) * (HD + 4) + HD]);
    float acc = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float* rec = red + ((size_t)w * G + g) * (HD + 4);
      const float sw = (rec[HD] == -INFINITY) ? 0.f : exp2f(rec[HD] - M);
      acc = fmaf(rec[d], sw, acc);
      L = fmaf(rec[HD + 1], sw, L);
    }
    if (a.direct) {      // the only "split": softmax normalisation here (a row always holds its own key: L > 0)
      const size_t o = blockIdx.y * a.q_stride + (size_t)(kvh * G + g) * HD + d;
      if (a.out_hi) split16<DT>(acc / L, a.out_hi[o], a.out_lo[o]);
      else a.out[o] = round_storage_if<DT>(acc / L, a.act16);
      continue;
    }
    float* out = part_row + ((size_t)(kvh * G + g) * a.nsplit + sp) * (HD + 4);
    out[d] = acc;
    if (d == 0) { out[HD] = M; out[HD + 1] = L; }       // a split without keys publishes m = -inf, l = 0
  }
}

}  // namespace tgx
