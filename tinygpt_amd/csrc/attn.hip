// attn.hip — the decode attention launches of the MI355X shim (kernels/attn_decode.h, attn_decode_mfma.h): the form is chosen per decode call from the
// context (direct / split + combine / matrix cores, update_attn_modes in abi.hip) and per launch from the batch rows.
// == flashAttention(q, Kall, Vall) over keys [0, pos[row]]  (Attention.h:103-112)
#include "ctx.h"
#include "kernels/attn_decode.h"
#include "kernels/attn_decode_mfma.h"

// the direct-form attention of a batched step runs on the matrix cores from attn.batch_mfma rows (17) when a kv head serves 3+ query heads (the VALU form's
// cost grows with the heads per workgroup, the MFMA form's does not: Qwen3-1.7B, 2 heads per kv head, B = 32 2.29 (VALU) vs 2.39 ms/step)
bool attn_batch_on_mfma(const tgx_ctx* c, int R) {
  return c->attn_batch_mfma > 0 && R >= c->attn_batch_mfma && (c->d.heads / c->d.kv_heads >= 3 || c->attn_batch_mfma == 1);
}

// P: paged KV (option kv.budget_tokens; kernels/common.h kv_paged_off) — every form exists in a paged instantiation (16-bit storage): same launch shapes, K / V
// through the rows' block tables
template <int DT, int HD, bool QKN = false, bool P = false>
static void launch_attn_g(tgx_ctx* c, tgx::AttnArgs a, int R, bool combine) {
  // the query heads of a kv group go to workgroups two at a time (blockIdx.z): the per-head state (8 output registers, the
  // merges) is what a workgroup's time grows with, while the K/V tile the groups re-read is small and mostly L2-resident.
  // Measured (option attn.gmax; tok/s at 4 / 2 / 1 heads per workgroup): Llama-3.2-1B ctx 2.3k 1364 / 1391 / 1388, ctx 8k
  // 1282 / 1312 / 1295; Qwen2.5-0.5B (7 heads per kv head) 1512 / 1610 / 1621; Mistral-7B 337 / 340 / 339
  // Round 5 (after round 4's single LDS meeting of the token-slot streams; tools/sweep.py --grid attn.gmax=1,2, batch 1, ms per token at 1 / 2 heads per workgroup):
  // Llama-3.2-1B context 0.9k 0.6559 / 0.6591, 1.2k 0.6546 / 0.6574, 2.2k 0.6525 / 0.6570 (three runs each, spread 0.0002), 4.3k 0.6767 / 0.6847; Qwen2.5-0.5B 2.2k
  // 0.5935 / 0.5967; Mistral-7B 2.9312 / 2.9370; Llama-3.2-3B 1.6084 / 1.6322 — one head per workgroup for a batch-1 step, two when rows share the launch
  const int gmax = c->attn_gmax > 0 ? c->attn_gmax : (R == 1 ? 1 : 2);
  const int gfull = a.heads / a.kv_heads, ngroups = gfull > gmax ? (gfull + gmax - 1) / gmax : 1, G = (gfull + ngroups - 1) / ngroups;
  a.gfull = gfull;
  a.direct = c->attn_direct ? 1 : 0;
  auto launch_combine = [&]() {      // the merge of the split records (combine = false: the caller's K-sliced o_proj merges them, kernels/oproj_sliced.h)
    if ((c->debug_skip & 2) || !combine) return;
    hipLaunchKernelGGL((tgx::attn_combine_kernel<HD>), dim3(a.heads, R), dim3(256), 0, c->stream, a);
  };
  if (a.direct) {   // short context: one 16-wave workgroup per query head, no combine launch.  Measured (tok/s, direct vs split at context
    // ~120 / ~300 / ~430): see DESIGN.md §5; 1 head per workgroup beats 2 and 4 here (the K/V block is L2-resident, the softmax chain is not)
    // Batches (round 3): with R rows the K/V working set (R x kv_heads x T rows) no longer fits the L2s, and one workgroup per QUERY head reads each kv
    // head gfull times — Llama-3.2-1B B = 32 at context ~600: 24 us per layer, a quarter of the step.  From `attn.direct_rows` rows on, a workgroup takes
    // two query heads of a kv head (option attn.direct_g: 1, 2 or 4 heads).
    // measured ms/step by heads per workgroup (1 / 2 / 4): Llama-3.2-1B context 600 B = 16 1.105 / 1.058 / 1.132, B = 32 1.551 / 1.421 / 1.382; context 2k
    // B = 16 1.354 / 1.208 / 1.384, B = 32 2.458 / 1.906 / 1.680; Mistral-7B context 600 B = 16 4.11 / 3.94 / 4.35, B = 32 6.08 / 5.47 / 5.53
    // Batches on the matrix cores (round 3, option attn.batch_mfma = rows from which): the VALU form's arithmetic grows with the heads per workgroup
    // (softmax chain + P.V update per key and head: B = 32 at context ~600 is VALU-bound at 16 us per layer for 39 MB of K / V), the MFMA form's does
    // not — one workgroup per (row, kv head), all the group's query heads as the narrow operand, no split, no combine launch.  Measured ms/step (VALU /
    // MFMA, 4 waves; 8 waves the same within 0.5 %): Llama-3.2-1B context 600 B = 16 1.047 / 1.068, B = 24 1.308 / 1.259, B = 32 1.356 / 1.307; context 2k
    // B = 16 1.210 / 1.224, B = 24 1.608 / 1.453, B = 32 1.708 / 1.562; Mistral-7B B = 16 3.92 / 4.10, B = 32 5.38 / 5.12: from 24 rows (below, rows x kv heads
    // workgroups leave CUs empty).  Closing build (no look-ahead set, QKV finish in the prologue), VALU / MFMA: B = 12 1.005 / 1.007, B = 16 1.052 / 1.072,
    // B = 17 1.247 / 1.159, B = 20 1.293 / 1.200, context 2k B = 17 1.553 / 1.340; Mistral-7B B = 16 3.92 / 4.07, B = 17 4.89 / 4.78: from 17 rows.
    // With AttnArgs.raw_* set the launch also finishes the QKV product (attn.raw_fuse: B = 32 1.335 -> 1.326, B = 8 1.000 -> 0.978)
    if constexpr (!QKN && DT != tgx::DT_F32 && HD == 64) {
      if (a.oj_w) {     // batch-1 step: + the o_proj product and the residual add (AttnArgs.oj_*, template OPJ); 4 waves per head; 4 wave-loads per softmax block up to attn_fused_nw4 keys, 8 beyond
        // workgroups per head: as many as keep heads x parts within one round of CUs (each repeats the head's attention and takes hidden / parts rows of the
        // strip) — layer_lab: Llama-3.2-1B at context 30 31.4 / 30.9 us per layer with 4 / 8 parts, 34.9 with 2; Qwen2.5-0.5B 18.7 / 18.8
        int rs = 8;
        while (rs > 1 && (a.heads * rs > c->num_cus || a.oj_H % rs || (a.oj_H / rs) % 8)) rs >>= 1;
        a.oj_rsplit = rs;
        const dim3 grid(a.kv_heads, 1, gfull * rs);
        if (c->debug_skip & 1) return;
        if (c->attn_nw4) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, false, false, 4, true, P>), grid, dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, false, false, 8, true, P>), grid, dim3(256), 0, c->stream, a);      // beyond: four waves x EIGHT wave-loads per block (layer_lab, 600 keys: 7.33 vs 7.57 us for eight waves x four on Llama-3.2-1B, 6.60 vs 6.85 on Qwen2.5-0.5B)
        return;
      }
    }
    if constexpr (!QKN && DT != tgx::DT_F32) {
      if (attn_batch_on_mfma(c, R)) {
        const dim3 gm(a.kv_heads, R);
        if (!(c->debug_skip & 1)) {
          constexpr size_t lds4 = tgx::attn_mfma_lds_bytes<HD, 4>(), ldsr = tgx::attn_mfma_raw_lds_bytes<HD, 4>();
          // head_dim 64: the form without the second K / V register set — 208 instead of 309 registers, two workgroups per CU.  Measured ms/step with /
          // without (Llama-3.2-1B, context 600): B = 32 1.317 / 1.314, B = 48 1.814 / 1.713, B = 64 1.902 / 1.796 — never behind: the default
          // (option attn.batch_la: 1 = look-ahead, -1 = only while the workgroups number at most one per CU)
          const bool la = HD != 64 || (c->attn_batch_la >= 0 ? c->attn_batch_la != 0 : (int)(gm.x * gm.y) <= c->num_cus);
          // eight waves per workgroup (head_dim 64, option attn.batch_nw8: 1 = while the workgroups number at most one per CU, 2 = always): the blocks of 64 keys and the
          // QKV finish's slab sums spread over twice the waves
          if constexpr (HD == 64) {
            if ((a.raw_part || a.raw_qkv) && (c->attn_batch_nw8 >= 2 || (c->attn_batch_nw8 == 1 && (int)(gm.x * gm.y) <= c->num_cus))) {
              constexpr size_t lds8 = tgx::attn_mfma_raw_lds_bytes<HD, 8>();
              hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 8, true, false, P>), gm, dim3(512), lds8, c->stream, a);
              return;
            }
          }
          if (a.raw_part || a.raw_qkv) {     // + the QKV product's finish
            if (la) hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, true, true, P>), gm, dim3(256), ldsr, c->stream, a);
            else hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, true, false, P>), gm, dim3(256), ldsr, c->stream, a);
          } else {
            if (la) hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, false, true, P>), gm, dim3(256), lds4, c->stream, a);
            else hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, false, false, P>), gm, dim3(256), lds4, c->stream, a);
          }
        }
        return;
      }
    }
    int dg = 1;
    // (two heads per workgroup as soon as one workgroup per query head would exceed one round of CUs: Llama-3.2-1B B = 9 0.937 -> 0.905 ms/step, B = 10 at context 2k
    //  1.194 -> 1.100; at 8 rows and fewer one head per workgroup stays ahead: B = 8 0.870 vs 0.895)
    if (!QKN && c->attn_direct_g > 0) dg = R >= 24 ? (HD == 64 ? 4 : 2) : ((R >= 12 || R * a.heads > c->num_cus) ? 2 : 1);
    if (!QKN && c->attn_direct_g < 0) dg = -c->attn_direct_g;          // experiments: force
    dg = std::min(dg, gfull);
    const bool raw = a.raw_part || a.raw_qkv;       // + the QKV product's finish in the prologue (batched step)
    if (dg >= 2 && gfull % dg == 0) {
      const dim3 gridg(a.kv_heads, R, gfull / dg), blkg(1024);
      if constexpr (!QKN && DT != tgx::DT_F32) {
        if (raw && !(c->debug_skip & 1)) {
          if (dg == 2) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 16, false, true, 2, false, P>), gridg, blkg, 0, c->stream, a);
          else hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 16, false, true, 1, false, P>), gridg, blkg, 0, c->stream, a);
          return;
        }
      }
      if constexpr (!QKN) {
        if (!(c->debug_skip & 1)) {
          if (dg == 2) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 16, false, false, 2, false, P>), gridg, blkg, 0, c->stream, a);
          else hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 16, false, false, 1, false, P>), gridg, blkg, 0, c->stream, a);
        }
      }
      return;
    }
    if constexpr (!QKN && DT != tgx::DT_F32) {
      if (raw && !(c->debug_skip & 1)) {          // (every remaining direct form of a batched step is one head per workgroup: also a group size dg does not divide)
        hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 16, false, true, 4, false, P>), dim3(a.kv_heads, R, gfull), dim3(1024), 0, c->stream, a);
        return;
      }
    }
    // very short contexts (option attn.direct_nw4: keys up to which the direct form runs FOUR waves per head instead of sixteen): one pass of a 4-wave
    // workgroup covers 128 keys at head_dim 64 (64 at 128), and four records merge faster than sixteen
    if (c->attn_nw4 && dg == 1) {
      const dim3 grid4(a.kv_heads, R, gfull), blk4(256);
      if (!(c->debug_skip & 1)) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, QKN, false, 4, false, P>), grid4, blk4, 0, c->stream, a);
      return;
    }
    const dim3 grid(a.kv_heads, R, gfull), blk(1024);
    if (!(c->debug_skip & 1)) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 16, QKN, false, 4, false, P>), grid, blk, 0, c->stream, a);
    return;
  }
  if (c->attn_mfma && !QKN && DT != tgx::DT_F32) {   // long context: QK^T and PV on the matrix cores, the kv group's query heads as the narrow operand
    if constexpr (!QKN && DT != tgx::DT_F32) {
      const dim3 gm(a.kv_heads * a.nsplit, R), bm(256);
      hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, false, true, P>), gm, bm, tgx::attn_mfma_lds_bytes<HD>(), c->stream, a);
    }
    launch_combine();
    return;
  }
  const int gx = a.kv_heads * a.nsplit;
  const dim3 grid(gx, R, ngroups), blk(256);
  if (!(c->debug_skip & 1)) switch (G) {
    case 1: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, QKN, false, 4, false, P>), grid, blk, 0, c->stream, a); break;
    case 2: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 4, QKN, false, 4, false, P>), grid, blk, 0, c->stream, a); break;
    case 3: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 3, 4, QKN, false, 4, false, P>), grid, blk, 0, c->stream, a); break;
    default: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 4, QKN, false, 4, false, P>), grid, blk, 0, c->stream, a); break;
  }
  launch_combine();
}

void launch_attn(tgx_ctx* c, const tgx::AttnArgs& a, int R, bool combine) {
  const bool qkn = a.k_raw && c->d.head_dim == 128;       // Qwen3's q/k norm + RoPE + cache append inside the attention launch (head_dim 128: every released Qwen3 size)
  if (a.blk_tbl) {
    TGX_DT16_SWITCH(c->dt,
      if (c->d.head_dim == 64) launch_attn_g<DT, 64, false, true>(c, a, R, combine);
      else if (qkn) launch_attn_g<DT, 128, true, true>(c, a, R, combine);
      else launch_attn_g<DT, 128, false, true>(c, a, R, combine);)
    return;
  }
  if (qkn) { TGX_DT_SWITCH(c->dt, (launch_attn_g<DT, 128, true>(c, a, R, combine))) return; }
  TGX_DT_SWITCH(c->dt, if (c->d.head_dim == 64) launch_attn_g<DT, 64>(c, a, R, combine); else launch_attn_g<DT, 128>(c, a, R, combine))
}

// dynamic LDS sizes of the matrix-core forms
template <bool P>
static int attn_set_attrs_p(tgx_ctx* c) {
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 64, 8, true, false, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<64, 8>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 64, 8, true, false, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<64, 8>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 128, 4, true, true, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 128, 4, true, true, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 128, 4, false, true, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 128, 4, false, true, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_lds_bytes<128>()));
  return TGX_OK;
}
int attn_set_attrs(tgx_ctx* c) {
  int rc = attn_set_attrs_p<false>(c);
  return rc ? rc : attn_set_attrs_p<true>(c);
}
