// attn_decode.h — single-query GQA attention over the pre-allocated KV cache (decode step).
//
// Replaces, for seq == 1:  KVCacheManager::append's concat (CacheManager.h:24-42 — the O(T) realloc+copy
// is gone: the QKV epilogue already stored this step's K/V row in place) and
// function::flashAttention(q, Kall, Vall, isCausal=false) (Attention.h:108-109).
//
// Numerics contract: K/V are in the storage dtype (bf16 / fp16 / fp32) in the cache; scores = (q.k) * hd^-1/2, softmax and P.V are fp32; fp32 output.
//
// Roofline: HBM — 2 * kv_heads * (T+1) * hd * 2 bytes per launch (K and V rows read once; all G = heads/kv_heads
// query heads of a group share one pass over their kv head).
//
// Decomposition (split-K flash-decode): grid = kv_heads * nsplit workgroups of 4 waves.  Workgroup (kvh, sp)
// owns the token range [sp*chunk, (sp+1)*chunk) with chunk derived on the device from the device-resident
// position (so one captured hipGraph serves every step).  Inside a wave, HD/8 adjacent lanes hold one
// token's row as 16-byte slices (a wave-load covers 64/(HD/8) consecutive tokens = 1 KiB contiguous);
// every lane group keeps its own online-softmax stream, merged once at the end (lanes -> waves via LDS ->
// one (m, l, o[HD]) partial per query head per split).  attn_combine_kernel merges the splits.
#pragma once
#include "common.h"

namespace tgx {

struct AttnArgs {
  const float* q;         // [heads][hd] fp32 (RoPE applied)
  const void* k_cache;    // this layer/row: [kv_heads][max_ctx][hd], storage dtype (kernel template DT)
  const void* v_cache;
  const int* pos;         // pastLength BEFORE this step; keys [0, pos] are attended
  float* part;            // [heads][nsplit][hd + 4]  (o[hd], m, l, pad) — 16-byte aligned records
  float* out;             // [heads*hd] fp32 (combine kernel)
  int heads, kv_heads, max_ctx, nsplit;
  float scale;
  // batch rows: blockIdx.y = row; row r uses q + r*q_stride, caches + r*kv_stride, pos[r], part + r*part_stride, out + r*q_stride
  long long q_stride, kv_stride, part_stride;
  int gfull;  // query heads per kv head; the G heads of a workgroup are gfull-group blockIdx.z * G .. +G-1 (the last group may be short)
  int direct;  // 1: short context — one workgroup per (kv head, head group) walks every block and writes the normalised output itself
               //    (grid.x = kv_heads, no partials, no combine launch); chosen on the host from pastLength (attn.direct_max)
  // Qwen3 (kernel template QKN): q arrives un-normalised and this position's k sits in k_raw (fp32) — the kernel applies the per-head
  // RMSNorm and RoPE to its q heads and to k itself (AttentionWithQKNorm, Attention.h:156-163), uses that k for key `pos`, and the
  // first head group of a kv head appends it to the cache.  One launch per layer less than a separate norm kernel.
  const float* k_raw;     // [kv_heads][hd] fp32, rows kraw_stride apart
  const void *q_norm_w, *k_norm_w;   // [hd], storage dtype
  const float *rope_cos, *rope_sin;  // [max_ctx][hd/2]
  float eps;
  long long kraw_stride;
  int dbg;   // experiments only (tgx_set_option "debug.attn"): 1 skip K/V work, 2 skip the LDS merge, 4 exit at once — results invalid
  // batched step (attn_decode_mfma_kernel template RAW): the finish of the QKV product — sum of its split-K slabs + bias, [Qwen3 q / k RMSNorm,] RoPE at
  // pos[row], KVCacheManager::append — runs in the attention launch's prologue for the workgroup's own (row, kv head): q never touches memory and
  // the row-wise rope_kv_rows launch disappears.  Columns of a QKV row: [q: heads x hd | k: kv_heads x hd | v: kv_heads x hd]
  const float* raw_qkv;   // [raw_rows][qd + 2 kvd] fp32 with the bias added, or nullptr: the sums of raw_part
  const float* raw_part;  // [raw_nsplit][raw_rows][qd + 2 kvd]
  const void* raw_bias;   // [qd + 2 kvd] storage dtype or nullptr (with raw_part only)
  int raw_nsplit, raw_rows;
  // direct forms: when set, the normalised rows leave as exact 16-bit split terms (row stride q_stride) — the o_proj product of a batched step then takes
  // stored terms (kernels/skinny_dma.h) instead of splitting fp32 rows while staging
  unsigned short *out_hi, *out_lo;
  // direct form with the o_proj product in its epilogue (kernel template OPJ, batch 1, short contexts; round 4): the workgroup of query head h multiplies
  // its normalised output by W_o[:, h * hd .. + hd) and adds the H partial sums to the fixed-point residual accumulators (kernels/oproj_sliced.h has the
  // long-context twin).  blockIdx.z = head-in-group * oj_rsplit + row part: oj_rsplit workgroups share a head, each repeats its attention and takes
  // H / oj_rsplit rows of the product.  No attention output touches memory and the o_proj launch disappears.
  const void* oj_w;       // [H][heads * hd] storage dtype
  const float* oj_x;      // [H] residual input (added once, by head 0)
  long long* oj_acc;      // [H] fixed-point accumulators, zero on entry
  int oj_H, oj_ldw, oj_rsplit;
  // paged KV (kernel template PAGED; common.h kv_paged_off): k_cache / v_cache = this layer's pools, row r's keys through blk_tbl + r * tbl_stride
  const int* blk_tbl;
  long long tbl_stride;
  int act16;              // option act.round16 (0 off, 1 bf16, 2 fp16): the normalised output — the o_proj input, in this launch (OPJ) or the next — is rounded to the
                          // storage dtype.  (ONE field for both uses: with a second one the split form's kernel, which executes neither, measured 5.02 -> 5.26 us)
};

template <int HD>
__device__ __forceinline__ void attn_combine_head(const float* p, int nsplit, float* out_head, float (*sm_o)[HD + 4], int round_mode = 0);   // below

// the normalised output value of a direct form; option act.round16 rounds it to the storage dtype (the o_proj input).  A branch, not a select: these stores sit
// at the end of the kernel (no loads in flight), and the select form made the SPLIT form's kernel 0.24 us slower although it never executes this code
template <int DT>
__device__ __forceinline__ float attn_out_value(float acc, float L, int round16) {
  float v = acc / L;
  if constexpr (DT != DT_F32) { if (round16) v = elem_to_f32<DT>(f32_to_elem<DT>(v)); }
  return v;
}

// NW = waves per workgroup: 4 for the split form; 16 for the direct form (short contexts), where ONE workgroup covers a block of
// NW * TPW * UNR tokens (512 at head_dim 64, 256 at 128) per pass over the load -> softmax chain.
// RAW (direct forms of a batched step): the workgroup first finishes its own slice of the QKV product (AttnArgs.raw_*: slab sums + bias, q / k norm at
// head_dim 128, RoPE, cache append by the kv head's first head group) — q, k and v of this position go through LDS, the row-wise rope_kv_rows launch
// disappears; the same arithmetic in the same order (common.h rope_rotate_pair / head_rms_inv): bit-identical to it.  See attn_decode_mfma.h for the MFMA twin.
// PAGED (round 6): the keys of a row are found through its block table — every wave-load (TPW consecutive tokens at a multiple of TPW) lies inside one
// KV_BLOCK-token page, so the lookup is one table entry per wave-load; the arithmetic is the unpaged kernel's.
template <int DT, int HD, int G, int NW = 4, bool QKN = false, bool RAW = false, int UNR_ = 4, bool OPJ = false, bool PAGED = false>
__global__ __launch_bounds__(64 * NW) void attn_decode_kernel(const AttnArgs a) {
  static_assert(!OPJ || (G == 1 && !QKN && !RAW && DT != DT_F32), "OPJ: one head per workgroup, 16-bit storage");
  static_assert(!PAGED || DT != DT_F32, "paged KV: 16-bit storage");
  typedef elem_t<DT> E;
  constexpr int LPT = HD / 8;         // lanes per token row
  constexpr int TPW = 64 / LPT;       // tokens per wave-load
  constexpr int UNR = UNR_;           // wave-loads of K and of V in flight per iteration (4; experiments: tools/probes/layer_lab.hip)
  constexpr float LOG2E = 1.4426950408889634f;
  // per wave and query head: o[HD], m, l  (m in the exp2 domain)
  __shared__ __attribute__((aligned(16))) float red[NW][G][HD + 4];
  __shared__ __attribute__((aligned(16))) float raw_q[RAW ? G : 1][RAW ? HD : 4], raw_kv[2][RAW ? HD : 4];      // RAW: this position's q heads (fp32) and k / v rows (as the cache holds them)

  if (TGX_DBG(a, 4)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* q_row = a.q + blockIdx.y * a.q_stride;
  const E* k_row = static_cast<const E*>(a.k_cache) + blockIdx.y * a.kv_stride;
  const E* v_row = static_cast<const E*>(a.v_cache) + blockIdx.y * a.kv_stride;
  float* part_row = a.part + blockIdx.y * a.part_stride;
  const int nsp = a.direct ? 1 : a.nsplit;
  const int kvh = blockIdx.x / nsp, sp = blockIdx.x - kvh * nsp;
  const int part_i = lane % LPT, slot = lane / LPT;
  const int g_base = (OPJ ? (int)blockIdx.z / a.oj_rsplit : (int)blockIdx.z) * G;      // first head (within the kv group) of this workgroup
  auto head_of = [&](int g) { return kvh * a.gfull + min(g_base + g, a.gfull - 1); };   // clamped: a short last group reloads its last head
  auto head_live = [&](int g) { return g_base + g < a.gfull; };
  // Token blocks of STEP = NW waves x UNR wave-loads are dealt round-robin to the splits: split sp owns blocks sp,
  // sp + nsplit, ...: the active splits are the first ceil(n_keys / STEP) of every kv head and each runs whole blocks;
  // the addresses of a split's first block do not depend on the context length.
  constexpr int STEP = NW * TPW * UNR;
  const E* kbase = k_row + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD) + part_i * 8;
  const E* vbase = v_row + (PAGED ? (size_t)0 : (size_t)kvh * a.max_ctx * HD) + part_i * 8;
  const int* tbl = PAGED ? a.blk_tbl + blockIdx.y * a.tbl_stride : nullptr;
  // element offset of token t's row from kbase / vbase
  auto tok_off = [&](int t) -> size_t {
    if constexpr (PAGED) return kv_paged_off(tbl, a.kv_heads, kvh, t, HD);
    else return (size_t)t * HD;
  };
  int t0 = sp * STEP + wv * TPW * UNR;
  Slice8<DT> kv[UNR], vv[UNR];
#pragma unroll
  for (int r = 0; r < UNR; r++) {
    const int tc = min(t0 + r * TPW + slot, a.max_ctx - 1);
    const size_t o = tok_off(tc);
    kv[r] = load_slice<DT>(kbase + o, 0);
    vv[r] = load_slice<DT>(vbase + o, 0);
  }
  const int n_keys = a.pos[blockIdx.y] + 1;
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
    // G + 2 head vectors (this workgroup's q heads, k, v) x HD / 2 rotation pairs; a vector's pairs are consecutive threads (half a wave at head_dim 64, a wave at 128)
    for (int it = threadIdx.x; it < (G + 2) * half; it += 64 * NW) {
      const int vec = it / half, p = it - vec * half;
      const int col = vec < G ? head_of(vec) * HD : (vec == G ? qd + kvh * HD : qd + kvd + kvh * HD);
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
      if (vec < G) { raw_q[vec][p] = x0; raw_q[vec][p + half] = x1; }
      else {
        const E e0 = f32_to_elem<DT>(x0), e1 = f32_to_elem<DT>(x1);
        if (g_base == 0) {                                                                // KVCacheManager::append, once per kv head
          E* cache = const_cast<E*>(vec == G ? kbase : vbase) - part_i * 8 + tok_off(pos);
          cache[p] = e0; cache[p + half] = e1;
        }
        raw_kv[vec - G][p] = elem_to_f32<DT>(e0); raw_kv[vec - G][p + half] = elem_to_f32<DT>(e1);
      }
    }
    __syncthreads();
  }
  float qf[G][8];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const f32x4* qp;
    if constexpr (RAW) qp = reinterpret_cast<const f32x4*>(&raw_q[g][part_i * 8]);
    else qp = reinterpret_cast<const f32x4*>(q_row + (size_t)head_of(g) * HD + part_i * 8);
    const f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
    for (int j = 0; j < 4; j++) { qf[g][j] = q0[j]; qf[g][4 + j] = q1[j]; }
  }
  // OPJ: this wave's rows of the o_proj strip W_o[rows, head columns]: LPT lanes per row (one 16-byte slice each), TPW rows per wave-load, OJC wave-loads
  // per chunk.  Chunk 0 leaves now, behind the first K / V block (loads return in order: the attention's waits are not held up by it) and behind q, and lands while the
  // attention runs; chunk 1 leaves when the key loop is done.
  constexpr int OJC = LPT;        // wave-loads per chunk: a chunk's OJC row sums go to the LPT lanes of a row group (8 at head_dim 64, 16 at 128)
  Slice8<DT> ow0[OPJ ? OJC : 1], ow1[OPJ ? OJC : 1];
  int oj_row0 = 0, oj_rend = 0, oj_rounds = 0;
  const E* oj_wp = nullptr;
  float oj_res[2] = {0.f, 0.f};
  if constexpr (OPJ) {
    const int rs = (int)blockIdx.z % a.oj_rsplit, rwg = a.oj_H / a.oj_rsplit;
    const int rw = ((rwg + NW * TPW - 1) / (NW * TPW)) * TPW;            // rows per wave (whole wave-loads)
    oj_rounds = rw / TPW;
    oj_rend = (rs + 1) * rwg;
    oj_row0 = rs * rwg + wv * rw;
    oj_wp = static_cast<const E*>(a.oj_w) + (size_t)head_of(0) * HD;
#pragma unroll
    for (int r = 0; r < OJC; r++) ow0[r] = load_slice_nt<DT>(oj_wp + (size_t)min(oj_row0 + r * TPW + slot, a.oj_H - 1) * a.oj_ldw, part_i);
#pragma unroll
    for (int c = 0; c < 2; c++) oj_res[c] = a.oj_x[min(oj_row0 + (c * OJC + part_i) * TPW + slot, a.oj_H - 1)];
  }
  if constexpr (OPJ) __builtin_amdgcn_sched_barrier(0);
  const float qscale = a.scale * LOG2E;     // softmax in base 2: exp(x) = exp2(x * log2 e)
  float knew[8];                            // QKN / RAW: this lane group's slice of the current position's key, as the cache will hold it
  float vnew[8];                            // RAW: and of its value row
  if constexpr (RAW) {
#pragma unroll
    for (int j = 0; j < 8; j++) { knew[j] = raw_kv[0][part_i * 8 + j]; vnew[j] = raw_kv[1][part_i * 8 + j]; }
  }
  if constexpr (QKN) {
    // a head row is spread over LPT lanes (8 dims each): lanes 0..LPT/2-1 hold the first half, their partners (lane ^ LPT/2) the second;
    // RoPE pairs (p, p + hd/2) therefore meet over one lane exchange
    constexpr int HL = LPT / 2;
    const int pos = n_keys - 1, half = HD / 2;
    const bool second = part_i >= HL;
    const int p0 = (part_i - (second ? HL : 0)) * 8;         // pair index of this lane's first element
    float cs[8], sn[8];
    {
      const f32x4* cp = reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos * half + p0);
      const f32x4* sp_ = reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos * half + p0);
      const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp_[0], s1 = sp_[1];
#pragma unroll
      for (int j = 0; j < 4; j++) { cs[j] = c0[j]; cs[4 + j] = c1[j]; sn[j] = s0[j]; sn[4 + j] = s1[j]; }
    }
    auto norm_rope = [&](float* x, const void* wv_) {      // x[8] in place: w * (x * inv_rms), then the rotation
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) ss = fmaf(x[j], x[j], ss);
      ss = row_group_sum<LPT>(ss);
      const float inv = 1.0f / sqrtf(ss / (float)HD + a.eps);
      float w[8];
      slice_unpack<DT>(load_slice<DT>(static_cast<const E*>(wv_), part_i), w);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float v = w[j] * (x[j] * inv);
        const float other = __shfl_xor(v, HL, 64);
        // == rope_rotate_pair (common.h) from the view of one partner: first half v cs - other sn, second half v cs + other sn
        x[j] = second ? fmaf(v, cs[j], __fmul_rn(other, sn[j])) : fmaf(v, cs[j], -__fmul_rn(other, sn[j]));
      }
    };
#pragma unroll
    for (int g = 0; g < G; g++) norm_rope(qf[g], a.q_norm_w);
    {
      const f32x4* kp = reinterpret_cast<const f32x4*>(a.k_raw + blockIdx.y * a.kraw_stride + (size_t)kvh * HD + part_i * 8);
      const f32x4 k0 = kp[0], k1 = kp[1];
#pragma unroll
      for (int j = 0; j < 4; j++) { knew[j] = k0[j]; knew[4 + j] = k1[j]; }
      norm_rope(knew, a.k_norm_w);
#pragma unroll
      for (int j = 0; j < 8; j++) knew[j] = elem_to_f32<DT>(f32_to_elem<DT>(knew[j]));     // what the cache holds from now on
    }
  }

  if (sp * STEP >= n_keys) {   // this split has no keys at the current context length (workgroup-uniform): publish "empty"
    // (the compiler sinks the loads above below this branch; running empty splits through the masked path instead keeps
    //  them ahead of the position load but measured 1262 vs 1260 tok/s at context 2.3k and 1267 vs 1289 at 300 — rejected)
    for (int g = threadIdx.x; g < G; g += 64 * NW) {
      if (!head_live(g)) continue;
      float* dst = part_row + ((size_t)head_of(g) * nsp + sp) * (HD + 4);
      dst[HD] = -INFINITY; dst[HD + 1] = 0.f;
    }
    return;
  }

#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int j = 0; j < 8; j++) qf[g][j] *= qscale;
  float m[G], l[G], o[G][8];
#pragma unroll
  for (int g = 0; g < G; g++) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) o[g][j] = 0.f;
  }

  auto block = [&]() {
    if (t0 < n_keys) {          // wave-uniform: a wave whose whole range lies beyond the context skips the arithmetic
    // the block's UNR scores first (independent dot products), then ONE softmax update per block and head: a maximum over the block, UNR + 1
    // exponentials and one rescale of the running output instead of 2 UNR exponentials and UNR rescales in a serial chain (round 4)
    float kf[UNR][8], vf[UNR][8];
    bool valid[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      valid[r] = t0 + r * TPW + slot < n_keys;
      slice_unpack<DT>(kv[r], kf[r]);
      slice_unpack<DT>(vv[r], vf[r]);
      if constexpr (RAW) {
        if (t0 + r * TPW + slot == n_keys - 1) {      // this step's own key / value: from the prologue (the cache row was stored a moment ago, possibly by another workgroup)
#pragma unroll
          for (int j = 0; j < 8; j++) { kf[r][j] = knew[j]; vf[r][j] = vnew[j]; }
        }
      }
      if constexpr (QKN) {
        const int tok = t0 + r * TPW + slot;
        if (tok == n_keys - 1) {            // the key of this step: computed above, not yet in the cache
#pragma unroll
          for (int j = 0; j < 8; j++) kf[r][j] = knew[j];
          if (g_base == 0) store_slice<DT>(const_cast<E*>(kbase + tok_off(tok)), 0, knew);   // KVCacheManager::append, once per kv head
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      float sc[UNR];
      float mb = -INFINITY;
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) t = fmaf(qf[g][j], kf[r][j], t);
        t = row_group_sum<LPT>(t);
        sc[r] = valid[r] ? t : -INFINITY;
        mb = fmaxf(mb, sc[r]);
      }
      if (mb != -INFINITY) {                        // (lane-group uniform) some key of this block is valid
        const float mn = fmaxf(m[g], mb);
        const float alpha = exp2f(m[g] - mn);       // exp2(-inf) = 0 on the first block
        float lb = 0.f, ob[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const float p = exp2f(sc[r] - mn);        // 0 for a masked key
          lb += p;
#pragma unroll
          for (int j = 0; j < 8; j++) ob[j] = fmaf(p, vf[r][j], ob[j]);
        }
        l[g] = fmaf(l[g], alpha, lb);
#pragma unroll
        for (int j = 0; j < 8; j++) o[g][j] = fmaf(o[g][j], alpha, ob[j]);
        m[g] = mn;
      }
    }
    }
  };
  auto reload = [&]() {
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const int tc = min(t0 + r * TPW + slot, n_keys - 1);
      const size_t o = tok_off(tc);
      kv[r] = load_slice<DT>(kbase + o, 0);
      vv[r] = load_slice<DT>(vbase + o, 0);
    }
  };
  if constexpr (OPJ) {
    // the first block outside the loop: a loop header would make its waits the conservative ones of the back edge (all but 3 loads landed), i.e. the
    // attention would wait for the o_proj strip that was issued behind its K / V block
    block();
    t0 += nsp * STEP;
    while (t0 < n_keys) { reload(); block(); t0 += nsp * STEP; }
  } else {
    while (!(TGX_DBG(a, 1))) {
      block();
      t0 += nsp * STEP;                 // this split's next block (contexts beyond nsplit*STEP tokens)
      if (t0 >= n_keys) break;          // wave-uniform
      reload();
    }
  }

  // 1. merge the TPW token-slot streams of a wave in registers: lanes with the same part_i exchange (m, l, o) over the lane
  //    bits above LPT (butterfly: every lane ends with the same sums, in the same association order)
  if (TGX_DBG(a, 2)) {
    if (wv == 0 && slot == 0)
      for (int g = 0; g < G; g++) {
        float* dst = part_row + ((size_t)head_of(g) * nsp + sp) * (HD + 4);
        for (int j = 0; j < 8; j++) dst[part_i * 8 + j] = o[g][j];
        if (part_i == 0) { dst[HD] = m[g]; dst[HD + 1] = l[g]; }
      }
    return;
  }
  if constexpr (NW == 4 && !OPJ) {
    // Split form and four-wave direct form (round 4): the NW x TPW token-slot streams of the workgroup meet ONCE in LDS.  Every lane group writes its
    // (m, l, o[HD]) stream; G x S threads derive each stream's weight 2^(m_i - M) and the sums M, L (one DPP row reduction + one cross-row exchange);
    // G x HD threads then take their output dim over the S weighted streams.  Replaces a 3-step register butterfly per wave (~30 ds_bpermute per
    // head) plus a 4-record LDS merge: attention 6.0 -> see profiles/r04_attn_pair.txt.
    constexpr int S = NW * TPW;                           // streams per head: 32 at head_dim 64, 16 at 128
    __shared__ __attribute__((aligned(16))) float so[S][G][HD];
    __shared__ float sm[G][S], sl[G][S], se[G][S], sML[G][2];
    const int st = wv * TPW + slot;
#pragma unroll
    for (int g = 0; g < G; g++) {
      f32x4* dst = reinterpret_cast<f32x4*>(&so[st][g][part_i * 8]);
      dst[0] = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
      dst[1] = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
      if (part_i == 0) { sm[g][st] = m[g]; sl[g][st] = l[g]; }
    }
    __syncthreads();
    if (threadIdx.x < G * S) {                            // (G x S <= 128 threads: whole rows of 16 lanes per head)
      const int g = threadIdx.x / S, i = threadIdx.x - g * S;
      const float mi = sm[g][i], li = sl[g][i];
      float M = mi;
      M = fmaxf(M, dpp_mov<0xB1, 0xf>(M)); M = fmaxf(M, dpp_mov<0x4E, 0xf>(M)); M = fmaxf(M, dpp_mov<0x141, 0xf>(M)); M = fmaxf(M, dpp_mov<0x140, 0xf>(M));
      if constexpr (S == 32) M = fmaxf(M, __shfl_xor(M, 16, 64));
      const float e = (mi == -INFINITY) ? 0.f : exp2f(mi - M);      // a stream without keys (m = -inf) weighs nothing
      float L = li * e;
      L = row_group_sum<16>(L);
      if constexpr (S == 32) L += __shfl_xor(L, 16, 64);
      se[g][i] = e;
      if (i == 0) { sML[g][0] = M; sML[g][1] = L; }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * HD; idx += 64 * NW) {
      const int g = idx / HD, d = idx - g * HD;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < S; i++) acc = fmaf(so[i][g][d], se[g][i], acc);
      if (!head_live(g)) continue;
      const float M = sML[g][0], L = sML[g][1];
      if (a.direct) {   // the only split: softmax normalisation here, straight into the o_proj input (fp32 rows, or 16-bit split terms for a batched step)
        const size_t oi = blockIdx.y * a.q_stride + (size_t)head_of(g) * HD + d;
        if constexpr (DT != DT_F32) {
          if (a.out_hi) {          // x = hi + lo, hi = round16(x), lo = round16(x - hi)  (prefill.h split16)
            const float v = acc / L;
            const E h = f32_to_elem<DT>(v);
            a.out_hi[oi] = h; a.out_lo[oi] = f32_to_elem<DT>(v - elem_to_f32<DT>(h));
            continue;
          }
        }
        a.out[oi] = attn_out_value<DT>(acc, L, a.act16);
        continue;
      }
      float* dst = part_row + ((size_t)head_of(g) * a.nsplit + sp) * (HD + 4);
      dst[d] = acc;
      if (d == 0) { dst[HD] = M; dst[HD + 1] = L; }
    }
    return;
  }
  if constexpr ((NW > 4 || OPJ) && G == 1) {
    if constexpr (OPJ) {
      if (oj_rounds > OJC) {
#pragma unroll
        for (int r = 0; r < OJC; r++) ow1[r] = load_slice_nt<DT>(oj_wp + (size_t)min(oj_row0 + (OJC + r) * TPW + slot, a.oj_H - 1) * a.oj_ldw, part_i);
      }
    }
    // Direct forms with 8 / 16 waves and one head per workgroup (batch 1, contexts of a few hundred keys): the same single LDS meeting of all NW x TPW
    // streams as above instead of a register butterfly per wave plus an NW-record merge whose cost grows with NW (16 records: ~2 us).  Wave 0 derives
    // the weights (1-2 streams per lane, wave reductions); HD x PARTS threads sum their quarter of the streams, HD threads add the quarters in order.
    constexpr int S = NW * TPW, KPL = (S + 63) / 64, PARTS = NW / 4, SPP = S / PARTS;
    __shared__ __attribute__((aligned(16))) float so[S][HD];
    __shared__ float sm[S], sl[S], se[S], sML[2], spart[PARTS][HD];
    const int st = wv * TPW + slot;
    {
      f32x4* dst = reinterpret_cast<f32x4*>(&so[st][part_i * 8]);
      dst[0] = f32x4{o[0][0], o[0][1], o[0][2], o[0][3]};
      dst[1] = f32x4{o[0][4], o[0][5], o[0][6], o[0][7]};
      if (part_i == 0) { sm[st] = m[0]; sl[st] = l[0]; }
    }
    __syncthreads();
    if (wv == 0) {
      float mi[KPL], li[KPL], M = -INFINITY;
#pragma unroll
      for (int k = 0; k < KPL; k++) {
        const int i = lane + 64 * k;
        mi[k] = i < S ? sm[i] : -INFINITY; li[k] = i < S ? sl[i] : 0.f;
        M = fmaxf(M, mi[k]);
      }
      M = group_max<64>(M);
      float L = 0.f;
#pragma unroll
      for (int k = 0; k < KPL; k++) {
        const float e = (mi[k] == -INFINITY) ? 0.f : exp2f(mi[k] - M);      // a stream without keys weighs nothing
        L = fmaf(li[k], e, L);
        if (lane + 64 * k < S) se[lane + 64 * k] = e;
      }
      L = wave_sum(L);
      if (lane == 0) { sML[0] = M; sML[1] = L; }
    }
    __syncthreads();
    if (threadIdx.x < HD * PARTS) {
      const int d = threadIdx.x % HD, pt = threadIdx.x / HD;
      float acc = 0.f;
#pragma unroll 8
      for (int i = pt * SPP; i < (pt + 1) * SPP; i++) acc = fmaf(so[i][d], se[i], acc);
      spart[pt][d] = acc;
    }
    __syncthreads();
    if (threadIdx.x < HD && head_live(0)) {
      const int d = threadIdx.x;
      float acc = spart[0][d];
#pragma unroll
      for (int pt = 1; pt < PARTS; pt++) acc += spart[pt][d];
      const float M = sML[0], L = sML[1];
      if constexpr (OPJ) spart[0][d] = round_storage_if<DT>(acc / L, a.act16);     // the normalised head output stays in LDS: the o_proj strip's activation
      else if (a.direct) {
        const size_t oi = blockIdx.y * a.q_stride + (size_t)head_of(0) * HD + d;
        bool done = false;
        if constexpr (DT != DT_F32) {
          if (a.out_hi) {
            const float v = acc / L;
            const E h = f32_to_elem<DT>(v);
            a.out_hi[oi] = h; a.out_lo[oi] = f32_to_elem<DT>(v - elem_to_f32<DT>(h));
            done = true;
          }
        }
        if (!done) a.out[oi] = attn_out_value<DT>(acc, L, a.act16);
      } else {
        float* dst = part_row + ((size_t)head_of(0) * a.nsplit + sp) * (HD + 4);
        dst[d] = acc;
        if (d == 0) { dst[HD] = M; dst[HD + 1] = L; }
      }
    }
    if constexpr (OPJ) {
      // == Attention.h:111-112 o_proj(attn) + the residual add of DecoderLayer.h:40, for this head's HD columns of every row of the part: lane (slot, part_i)
      // holds 8 activations; a wave-load covers TPW rows; the row sums of a chunk are handed to distinct lanes (round r -> the lanes with part_i == r), so a
      // chunk ends in ONE atomic instruction per wave over 64 consecutive accumulators.  64-bit fixed point: the sum over heads does not depend on the order
      // in which the workgroups arrive (common.h f32_to_fix).
      __syncthreads();
      const f32x4 xa = *reinterpret_cast<const f32x4*>(&spart[0][part_i * 8]), xb = *reinterpret_cast<const f32x4*>(&spart[0][part_i * 8 + 4]);
      const bool first_head = head_of(0) == 0;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        if (c * OJC >= oj_rounds) break;
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < OJC; r++) {
          float sdot = dot8<DT>(0.f, c == 0 ? ow0[r] : ow1[r], xa, xb);
          sdot = row_group_sum<LPT>(sdot);
          if (part_i == r) mine = sdot;
        }
        const int row = oj_row0 + (c * OJC + part_i) * TPW + slot;
        if (c * OJC + part_i < oj_rounds && row < oj_rend) {
          long long f = f32_to_fix(mine);
          if (first_head) f += f32_to_fix(oj_res[c]);
          __hip_atomic_fetch_add(a.oj_acc + row, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      for (int c = 2; c * OJC < oj_rounds; c++) {        // strips taller than 2 chunks x NW waves (hidden > 2048 per part): streamed, latency exposed
#pragma unroll
        for (int r = 0; r < OJC; r++) ow0[r] = load_slice_nt<DT>(oj_wp + (size_t)min(oj_row0 + (c * OJC + r) * TPW + slot, a.oj_H - 1) * a.oj_ldw, part_i);
        const int row = oj_row0 + (c * OJC + part_i) * TPW + slot;
        const float res = a.oj_x[min(row, a.oj_H - 1)];
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < OJC; r++) {
          float sdot = dot8<DT>(0.f, ow0[r], xa, xb);
          sdot = row_group_sum<LPT>(sdot);
          if (part_i == r) mine = sdot;
        }
        if (c * OJC + part_i < oj_rounds && row < oj_rend) {
          long long f = f32_to_fix(mine);
          if (first_head) f += f32_to_fix(res);
          __hip_atomic_fetch_add(a.oj_acc + row, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
    float M = m[g];
#pragma unroll
    for (int off = LPT; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    const float sc = (m[g] == -INFINITY) ? 0.f : exp2f(m[g] - M);
    l[g] *= sc;
#pragma unroll
    for (int j = 0; j < 8; j++) o[g][j] *= sc;
#pragma unroll
    for (int off = LPT; off < 64; off <<= 1) {
      l[g] += __shfl_xor(l[g], off, 64);
#pragma unroll
      for (int j = 0; j < 8; j++) o[g][j] += __shfl_xor(o[g][j], off, 64);
    }
    m[g] = M;
  }
  // 2. the waves meet in LDS (one record per wave and query head), merged in wave order by G*HD threads
  if (slot == 0) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      f32x4* dst = reinterpret_cast<f32x4*>(&red[wv][g][part_i * 8]);
      dst[0] = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
      dst[1] = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
      if (part_i == 0) { red[wv][g][HD] = m[g]; red[wv][g][HD + 1] = l[g]; }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * HD; idx += 64 * NW) {
    const int g = idx / HD, d = idx - g * HD;
    float M = red[0][g][HD];
#pragma unroll
    for (int w = 1; w < NW; w++) M = fmaxf(M, red[w][g][HD]);
    float acc = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float mw = red[w][g][HD];
      const float sw = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
      acc = w == 0 ? red[0][g][d] * sw : fmaf(red[w][g][d], sw, acc);
      L = w == 0 ? red[0][g][HD + 1] * sw : fmaf(red[w][g][HD + 1], sw, L);
    }
    if (!head_live(g)) continue;
    if (a.direct) {   // the only split: softmax normalisation here, straight into the o_proj input (fp32 rows, or 16-bit split terms for a batched step)
      const size_t o = blockIdx.y * a.q_stride + (size_t)head_of(g) * HD + d;
      if constexpr (DT != DT_F32) {
        if (a.out_hi) {          // x = hi + lo, hi = round16(x), lo = round16(x - hi)  (prefill.h split16)
          const float v = acc / L;
          const E h = f32_to_elem<DT>(v);
          a.out_hi[o] = h; a.out_lo[o] = f32_to_elem<DT>(v - elem_to_f32<DT>(h));
          continue;
        }
      }
      a.out[o] = attn_out_value<DT>(acc, L, a.act16);
      continue;
    }
    float* dst = part_row + ((size_t)head_of(g) * a.nsplit + sp) * (HD + 4);
    dst[d] = acc;
    if (d == 0) { dst[HD] = M; dst[HD + 1] = L; }
  }
}

// Merges the nsplit partials of every query head and writes the attention output (fp32)
// (== the reshape to [B,S,qDim] that feeds o_proj, Attention.h:111).
// One workgroup per query head; thread (s, dg) owns 8 dims of one split, so all partials arrive with one round of
// independent 16-byte loads.  Each wave folds its splits in registers (butterfly over the lane bits above DG, fixed
// order), the four waves meet once in LDS.
// The merge of one query head's split records (256 threads): thread (s, dg) owns 8 dims of one split, so all partials arrive with one round
// of independent 16-byte loads; each wave folds its splits in registers (butterfly over the lane bits above DG, fixed order), the four waves
// meet once in LDS.  `nsplit` records are read; sm_o is [4][HD + 4] floats of LDS.  Shared by attn_combine_kernel and the in-kernel fold.
template <int HD>
__device__ __forceinline__ void attn_combine_head(const float* p, int nsplit, float* out_head, float (*sm_o)[HD + 4], int round_mode) {
  constexpr int DG = HD / 8;            // lanes per split (dim groups of 8)
  constexpr int SPB = 256 / DG;         // splits per pass (32 for hd 64, 16 for hd 128)
  constexpr int NPASS = 32 / SPB;       // 1 or 2 passes cover the 32 possible splits
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int dg = tid % DG, sl = tid / DG;
  f32x4 d0[NPASS], d1[NPASS];
  float pm[NPASS], pl[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) {   // every load of the merge is issued here
    const int s = ps * SPB + sl;
    d0[ps] = f32x4{0.f, 0.f, 0.f, 0.f}; d1[ps] = d0[ps]; pm[ps] = -INFINITY; pl[ps] = 0.f;
    if (s < nsplit) {
      const float* rec = p + (size_t)s * (HD + 4);
      const f32x4* src = reinterpret_cast<const f32x4*>(rec + dg * 8);
      d0[ps] = src[0]; d1[ps] = src[1];
      pm[ps] = rec[HD]; pl[ps] = rec[HD + 1];
    }
  }
  // this thread's splits -> one (M, L, o[8]); empty splits (m = -inf) hold stale data: select, never multiply
  float M = pm[0];
#pragma unroll
  for (int ps = 1; ps < NPASS; ps++) M = fmaxf(M, pm[ps]);
#pragma unroll
  for (int off = DG; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));     // wave maximum (m is in the exp2 domain)
  float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) {
    if (pm[ps] != -INFINITY) {
      const float e = exp2f(pm[ps] - M);
      L = fmaf(pl[ps], e, L);
#pragma unroll
      for (int j = 0; j < 4; j++) { o[j] = fmaf(d0[ps][j], e, o[j]); o[4 + j] = fmaf(d1[ps][j], e, o[4 + j]); }
    }
  }
#pragma unroll
  for (int off = DG; off < 64; off <<= 1) {
    L += __shfl_xor(L, off, 64);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] += __shfl_xor(o[j], off, 64);
  }
  if (lane < DG) {
    f32x4* dst = reinterpret_cast<f32x4*>(&sm_o[wv][dg * 8]);
    dst[0] = f32x4{o[0], o[1], o[2], o[3]};
    dst[1] = f32x4{o[4], o[5], o[6], o[7]};
    if (lane == 0) { sm_o[wv][HD] = M; sm_o[wv][HD + 1] = L; }
  }
  __syncthreads();
  if (tid < HD) {
    const float m0 = sm_o[0][HD], m1 = sm_o[1][HD], m2 = sm_o[2][HD], m3 = sm_o[3][HD];
    const float MM = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float s0 = (m0 == -INFINITY) ? 0.f : exp2f(m0 - MM), s1 = (m1 == -INFINITY) ? 0.f : exp2f(m1 - MM);
    const float s2 = (m2 == -INFINITY) ? 0.f : exp2f(m2 - MM), s3 = (m3 == -INFINITY) ? 0.f : exp2f(m3 - MM);
    float acc = sm_o[0][tid] * s0;
    acc = fmaf(sm_o[1][tid], s1, acc); acc = fmaf(sm_o[2][tid], s2, acc); acc = fmaf(sm_o[3][tid], s3, acc);
    float LL = sm_o[0][HD + 1] * s0;
    LL = fmaf(sm_o[1][HD + 1], s1, LL); LL = fmaf(sm_o[2][HD + 1], s2, LL); LL = fmaf(sm_o[3][HD + 1], s3, LL);
    out_head[tid] = round_storage_mode(acc / LL, round_mode);       // option act.round16 (AttnArgs.act16: 1 bf16, 2 fp16): the o_proj input
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sm_o[4][HD + 4];   // per wave: o[HD], M, L
  const int h = blockIdx.x;
  const float* p = a.part + blockIdx.y * a.part_stride + (size_t)h * a.nsplit * (HD + 4);
  attn_combine_head<HD>(p, a.nsplit, a.out + blockIdx.y * a.q_stride + (size_t)h * HD, sm_o, a.act16);
}

}  // namespace tgx
