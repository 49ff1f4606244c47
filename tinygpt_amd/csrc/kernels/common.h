// common.h — device helpers shared by the gfx950 kernels of the decode path.
// Wavefront = 64 lanes everywhere (CDNA4); all reductions run in a fixed order (deterministic output).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgx {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;   // bf16 bit pattern

constexpr int WAVE = 64;

// Experiment hooks (tools/gemv_dissect.py, tools/attn_dissect.py): parts of a kernel can be switched off through the `dbg`
// field of its argument block — compiled in only with -DTGX_DISSECT=1 (TGX_DISSECT=1 python tinygpt_amd/build.py -f);
// in the product build the tests fold to false (even a uniform branch on a kernel argument costs ~0.3 us per launch).
#ifndef TGX_DISSECT
#define TGX_DISSECT 0
#endif
#define TGX_DBG(args, bit) (TGX_DISSECT && ((args).dbg & (bit)))

// ---- bf16 <-> fp32 (round-to-nearest-even; the R() of the numerics contract) ----
__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((unsigned int)b) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // v_cvt_pk_bf16_f32: round-to-nearest-even in hardware (gfx950)
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float rbf(float f) { return bf16_to_f32(f32_to_bf16(f)); }
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// ---- storage dtype of parameters and of the KV cache (tgx_model_desc.compute_dtype; activations are fp32 in every
// mode).  Kernels take the dtype as a template parameter; a *slice* is 8 consecutive elements — one
// 16-byte load for the 16-bit types, two for fp32.
enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };
template <int DT> struct Elem { typedef unsigned short type; };
template <> struct Elem<DT_F32> { typedef float type; };
template <int DT> using elem_t = typename Elem<DT>::type;

__device__ __forceinline__ float f16_bits_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // v_cvt_f16_f32, RNE

template <int DT> __device__ __forceinline__ float elem_to_f32(elem_t<DT> v) {
  if constexpr (DT == DT_BF16) return bf16_to_f32(v);
  else if constexpr (DT == DT_F16) return f16_bits_to_f32(v);
  else return v;
}
template <int DT> __device__ __forceinline__ elem_t<DT> f32_to_elem(float f) {   // round-to-nearest-even into the storage dtype
  if constexpr (DT == DT_BF16) return f32_to_bf16(f);
  else if constexpr (DT == DT_F16) return f32_to_f16_bits(f);
  else return f;
}
// two adjacent 16-bit elements packed in one dword -> fp32
template <int DT> __device__ __forceinline__ float pair_lo(unsigned int u) {
  if constexpr (DT == DT_BF16) return bf16_lo(u); else return f16_bits_to_f32((unsigned short)(u & 0xffffu));
}
template <int DT> __device__ __forceinline__ float pair_hi(unsigned int u) {
  if constexpr (DT == DT_BF16) return bf16_hi(u); else return f16_bits_to_f32((unsigned short)(u >> 16));
}

template <int DT> struct Slice8 { u32x4 v; };
template <> struct Slice8<DT_F32> { u32x4 v, w; };

// ---- streaming (read-once) 16-byte load: global_load_dwordx4 ... nt -------------------------------
__device__ __forceinline__ u32x4 load_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

// slice `i` (elements 8i .. 8i+7) of a row that starts at `row` (16-byte aligned); nt = read-once weight stream
template <int DT> __device__ __forceinline__ Slice8<DT> load_slice_nt(const elem_t<DT>* row, size_t i) {
  Slice8<DT> s;
  if constexpr (DT == DT_F32) {
    const u32x4* p = reinterpret_cast<const u32x4*>(row) + 2 * i;
    s.v = load_nt(p); s.w = load_nt(p + 1);
  } else {
    s.v = load_nt(reinterpret_cast<const u32x4*>(row) + i);
  }
  return s;
}
template <int DT> __device__ __forceinline__ Slice8<DT> load_slice(const elem_t<DT>* row, size_t i) {
  Slice8<DT> s;
  if constexpr (DT == DT_F32) {
    const u32x4* p = reinterpret_cast<const u32x4*>(row) + 2 * i;
    s.v = p[0]; s.w = p[1];
  } else {
    s.v = reinterpret_cast<const u32x4*>(row)[i];
  }
  return s;
}
template <int DT> __device__ __forceinline__ void slice_unpack(const Slice8<DT>& s, float f[8]) {
  if constexpr (DT == DT_F32) {
#pragma unroll
    for (int t = 0; t < 4; t++) { f[t] = __uint_as_float(s.v[t]); f[4 + t] = __uint_as_float(s.w[t]); }
  } else {
#pragma unroll
    for (int t = 0; t < 4; t++) { f[2 * t] = pair_lo<DT>(s.v[t]); f[2 * t + 1] = pair_hi<DT>(s.v[t]); }
  }
}
template <int DT> __device__ __forceinline__ void store_slice(elem_t<DT>* row, size_t i, const float f[8]) {
  if constexpr (DT == DT_F32) {
    f32x4* p = reinterpret_cast<f32x4*>(row) + 2 * i;
    p[0] = f32x4{f[0], f[1], f[2], f[3]}; p[1] = f32x4{f[4], f[5], f[6], f[7]};
  } else {
    u32x4 o;
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = (unsigned int)f32_to_elem<DT>(f[2 * t]) | ((unsigned int)f32_to_elem<DT>(f[2 * t + 1]) << 16);
    reinterpret_cast<u32x4*>(row)[i] = o;
  }
}

// acc += dot(8 stored weights in w, 8 fp32 activations in xa/xb) — exact conversions, fp32 FMA chain in element order
template <int DT>
__device__ __forceinline__ float dot8(float acc, const Slice8<DT>& w, const f32x4 xa, const f32x4 xb) {
  float f[8];
  slice_unpack<DT>(w, f);
  acc = fmaf(f[0], xa[0], acc);
  acc = fmaf(f[1], xa[1], acc);
  acc = fmaf(f[2], xa[2], acc);
  acc = fmaf(f[3], xa[3], acc);
  acc = fmaf(f[4], xb[0], acc);
  acc = fmaf(f[5], xb[1], acc);
  acc = fmaf(f[6], xb[2], acc);
  acc = fmaf(f[7], xb[3], acc);
  return acc;
}

// RoPE rotation of one (p, p + hd/2) pair and the per-head RMSNorm factor with the contractions written out: the batched step computes them either in
// rope_kv_rows_kernel (skinny.h) or in the attention launch's prologue (attn_decode_mfma.h RAW) and the two must agree bit for bit
__device__ __forceinline__ void rope_rotate_pair(float& x0, float& x1, float cs, float sn) {
  const float r0 = fmaf(x0, cs, -__fmul_rn(x1, sn)), r1 = fmaf(x1, cs, __fmul_rn(x0, sn));
  x0 = r0; x1 = r1;
}
__device__ __forceinline__ float head_sq_pair(float x0, float x1) { return fmaf(x0, x0, __fmul_rn(x1, x1)); }
__device__ __forceinline__ float head_rms_inv(float ss, int hd, float eps) { return 1.0f / sqrtf(__fadd_rn(__fdiv_rn(ss, (float)hd), eps)); }

// ---- the residual stream in fixed point (kernels/oproj_sliced.h): 2^-32 units in a 64-bit integer — range +-2^31, resolution 2.3e-10 (below the
// fp32 ulp of any |x| > 4e-3).  Integer adds commute, so a sum of contributions from many workgroups does not depend on their arrival order.
// f32 -> fixed: v * 2^32 is exact in fp32 (a power-of-two scale), the conversion rounds to nearest.  fixed -> f32: the correctly rounded int64 -> float
// conversion (sign, leading-zero count, shift, round: ~10 VALU instructions) and an exact scale.  (A cheaper hi / lo split was tried first: its error is
// ABSOLUTE, 2^-25, and small negative values — -1 + 0.9999 — lost four digits: logits 3e-5 off at a one-token context.)
// Range and non-finite values: |v| >= 2^31 saturates and NaN converts to 0 (v_cvt semantics via __float2ll_rn), so a DIVERGED activation does
// not propagate as Inf / NaN through the batch-1 fixed-point path the way it does through the fp32 paths (batch >= 2, fp32 storage, prefill).  A residual
// stream anywhere near 2^31 = 2.1e9 is a broken model (trained checkpoints: 1e1 .. 1e4), the logits of such a step are garbage on every path, and the
// greedy argmax over garbage is what the caller gets either way; the price of a sticky non-finite flag (a second atomic per contribution) is not paid.
// Callers that must detect divergence read the logits (tgx_read_logits): a saturated stream shows as logits of ~1e9.
__device__ __forceinline__ long long f32_to_fix(float v) { return __float2ll_rn(v * 4294967296.0f); }
// widest hidden size the fixed-point forms serve: gemv_kernel<.., XACC> requests hidden * 4 bytes of dynamic LDS for the hand-over (32 KB here, inside the
// 64 KB a launch gets without raising the kernel's dynamic-LDS attribute); decode.hip's capability predicates send wider models to the fp32-residual forms
constexpr int XACC_HIDDEN_MAX = 8192;
__device__ __forceinline__ float fix_to_f32(long long a) { return (float)a * (1.0f / 4294967296.0f); }

// ---- paged KV cache (round 6; option kv.budget_tokens) --------------------------------------------------------------------------------------------
// The reference's KVCacheManager grows a row's cache by concat (CacheManager.h:24-42); the unpaged layout here gives every row a max_ctx slab
// [layer][kv_head][max_ctx][hd].  Paged: one pool per layer, [block][kv_head][KV_BLOCK tokens][hd], and a per-row block table on the device (entry b = the
// physical block of tokens [b KV_BLOCK, (b + 1) KV_BLOCK); entry 0 of the pool is a scratch block that unassigned table entries point to, so a speculative or
// clamped access is always legal).  KV_BLOCK = 128 = the key block of a decode-attention split at head_dim 64 (two of them at 128): a wave-load, a split's
// block and a prefill attention tile never straddle two pages.
constexpr int KV_BLOCK = 128, KV_BLOCK_SHIFT = 7;
// element offset of token t of kv head kvh inside a layer's pool
__device__ __forceinline__ size_t kv_paged_off(const int* tbl, int kv_heads, int kvh, int t, int hd) {
  return (((size_t)tbl[t >> KV_BLOCK_SHIFT] * kv_heads + kvh) * KV_BLOCK + (t & (KV_BLOCK - 1))) * (size_t)hd;
}

// ---- cross-lane reductions ----------------------------------------------------------------------
// 64-lane sum on the DPP crossbar (no LDS traffic): quad butterflies, half-row / row mirrors, then the two
// row broadcasts of the GFX9 wave64 reduction; the total lands in lane 63 and is returned wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);    // quad_perm:[2,3,0,1]
  v += dpp_mov<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_mov<0x140, 0xf>(v);   // row_mirror
  v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
  v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int WIDTH = 64>
__device__ __forceinline__ float group_sum(float v) {   // butterfly: every lane of the group ends with the sum
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH = 64>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over aligned groups of 8 or 16 adjacent lanes on the DPP crossbar; every lane of the group gets the sum.
template <int WIDTH>
__device__ __forceinline__ float row_group_sum(float v) {
  static_assert(WIDTH == 8 || WIDTH == 16, "row_group_sum: 8 or 16 lanes");
  v += dpp_mov<0xB1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);    // quad_perm:[2,3,0,1]
  v += dpp_mov<0x141, 0xf>(v);   // row_half_mirror
  if (WIDTH == 16) v += dpp_mov<0x140, 0xf>(v);   // row_mirror
  return v;
}

// option act.round16: v rounded to the storage dtype when `on` (kernel-uniform), as a SELECT — a branch between a kernel's loads and their uses makes hipcc's
// s_waitcnt insertion drain the memory pipeline at the join (measured: down 8.7 -> 9.1 us with the branch form)
template <int DT>
__device__ __forceinline__ float round_storage_if(float v, int on) {
  if constexpr (DT == DT_F32) return v;
  else {
    const float r = elem_to_f32<DT>(f32_to_elem<DT>(v));
    return on ? r : v;
  }
}

// the same for kernels without a storage-dtype template: mode 0 = off, 1 = bf16, 2 = fp16
__device__ __forceinline__ float round_storage_mode(float v, int mode) {
  const float rb = elem_to_f32<DT_BF16>(f32_to_elem<DT_BF16>(v)), rh = elem_to_f32<DT_F16>(f32_to_elem<DT_F16>(v));
  return mode == 1 ? rb : (mode == 2 ? rh : v);
}

// Block-wide sum over 256 threads (4 waves) through a 4-float LDS scratch; every thread gets the result.
__device__ __forceinline__ float block_sum_256(float v, float* scratch4) {
  v = group_sum<64>(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) scratch4[wv] = v;
  __syncthreads();
  float r = scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
  __syncthreads();
  return r;
}

}  // namespace tgx
