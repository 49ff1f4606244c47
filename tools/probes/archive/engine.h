// engine.h — the persistent weight-streaming engine of the batch-1 decode step (round 3).
//
// ONE launch runs a chain of dependent Linears of a decoder layer —  o_proj (+residual) -> RMSNorm -> gate_up -> siluMul ->
// down (+residual) -> [next layer's RMSNorm -> qkv -> RoPE -> cache append]  (Attention.h:90, DecoderLayer.h:38-43, GatedMLP.h:37-41,
// Attention.h:94-106) — on 256 resident workgroups, one per CU, instead of four dependent GEMV launches.  What it buys over the launches
// (gemv.h) is not the kernel boundaries themselves (an in-launch all-to-all hand-off costs as much, tools/probes/hop_probe.hip) but that the
// WEIGHT STREAM NEVER STOPS: a loader wave per CU walks the static tile schedule of all ops and lands the next op's weights in LDS
// (global_load_lds, nt) while the current op's output vector is still being exchanged.
//
// MEASURED (profiles/r03_engine.txt): parity-green and deterministic, and SLOWER than the launches — 33.6 vs 29.8 us per layer for the four ops, 22.3 vs
// 21.3 for the MLP pair; the stream runs at 6.1 TB/s, but every all-to-all edge costs 3.4-4.5 us against ~2.65 us for a launch boundary plus the
// dependent fetch, and the ring (4.8 us of stream) cannot hide the chain start -> o_proj -> x' edge.  Hence option engine.mode, default 0.
//
// Roles of the 5 waves of a workgroup (MI355X_MICROARCH.md "ldsdma-fill", "prefetch-credit", "gather-pass"):
//   wave 0    loader     16 x 1-KiB global_load_lds_dwordx4 per 16-KiB tile into a ring of NS slots; counted vmcnt -> `landed`
//   wave 1-3  consumers  tile t goes to consumer t mod 3: 8 weight rows x 1024 k from LDS (ds_read_b128), fp32 FMA against the activation
//                        slice of the tile's k range (LDS), one 8-value cross-lane reduction, partial sums to LDS; the wave that completes a
//                        row group (all k chunks) sums the partials in k order, runs the epilogue and publishes the results
//   wave 4    gatherer   collects the NEXT op's input vector from the other CUs' granules (8-byte {fp32, tag}, one sc1 store each, swept with
//                        sc1 loads until every tag is current), applies RMSNorm, stages it in LDS for the consumers
//
// Tile = 8 rows x 1024 k = 16 pieces of 1 KiB; piece (r, hf) holds row r, k = kc*1024 + hf*512 + 8*lane .. +8 for lane 0..63, so a consumer
// lane reads 16 contiguous bytes per piece (conflict-free) and always multiplies against the same 16 activation values per k chunk.
// Row groups: the 8 rows whose results one epilogue needs together (gate rows u..u+3 with up rows u..u+3; RoPE partners p..p+3 with
// p+hd/2..; 8 adjacent rows for the residual products).  The h edge (gate_up -> down) is PIPELINED: CU c's i-th row group produces
// h[1024 i + 4c .. +4), so k chunk i of h is complete when every CU has finished its i-th group, and the down product's tile (.., kc) only needs
// chunk kc — the 64 KB all-to-all that a launch boundary (or a monolithic gather) would serialise overlaps the gate_up stream.
//
// Numerics: same contract as gemv.h: weights exact to fp32, fp32 FMA, fp32 activations end to end; x_hat = w * (x * inv_rms)
// as in gemv.h.  Summation order differs from the launches (another lane -> k map, k-chunk partials), so results agree with them to
// rounding (~1e-6), not bit for bit; the order is fixed, so the engine is deterministic.
//
// Every spin is bounded: a wait that gives up raises `abort` in LDS (every other wait of the workgroup then falls through) and records a
// code in EngArgs.err; the kernel still terminates.
#pragma once
#include "common.h"

namespace tgx {

typedef unsigned long long u64;

constexpr int ENG_THREADS = 320;
constexpr int ENG_NCONS = 3;
constexpr int ENG_SLOT = 16384;
constexpr int ENG_MAX_OPS = 4;
constexpr int ENG_RGS = 8;             // row-group partial buffers in flight (bounded by the ring: a slot is released after its epilogue)
constexpr int ENG_KCMAX = 16;          // k chunks per row (K <= 16384)
constexpr int ENG_RES_MAX = 32;        // residual values a CU owns (hidden <= 8192 on 256 CUs)
constexpr unsigned ENG_SPIN_LIMIT = 1u << 22;

enum { EOP_RESID = 0, EOP_SILU = 1, EOP_QKV = 2 };

struct EngOp {
  const void* W;          // [N][K] storage dtype
  const void* bias;       // [N] or nullptr
  const void* norm_w;     // RMSNorm weight over the input [K] (nullptr: plain input)
  const float* in_plain;  // input vector in global memory (written by an earlier launch) or nullptr
  const u64* in_gran;     // input vector as granules written inside this launch (in_plain == nullptr)
  u64* out_gran;          // output vector as granules (the next op's input) or nullptr
  float* out_plain;       // output vector as plain fp32 (read by later launches) or nullptr
  int N, K;
  int epi;
  int in_tag, out_tag;    // edge numbers within the launch (tag = epoch base + number)
  // schedule, computed on the host (eng_plan_op): CU c owns nrg_lo + (c * cmul < nrg_rem) row groups — no division on the device
  int nrg_lo, nrg_rem, cmul;
};

struct EngArgs {
  EngOp op[ENG_MAX_OPS];
  int nops;
  int ns;                 // ring slots
  int xb_bytes[2];        // input staging buffers (ops alternate)
  int thin;               // while this CU's gatherer runs an urgent sweep (gather-pass): 1 = the loader keeps one fill in flight, 2 = it pauses
  int depth;              // fills in flight per CU before the loader waits for the oldest (2..4)
  const float* x_in;      // residual stream at launch start [H]
  // EOP_QKV epilogue
  float* q_out;
  void* k_cache;          // this layer's [kv_heads][max_ctx][hd]
  void* v_cache;
  const float* rope_cos;  // [max_ctx][hd/2]
  const float* rope_sin;
  const int* pos;
  int heads, kv_heads, hd, max_ctx;
  float eps;
  unsigned* epoch;        // device counter: base of this launch's tags
  unsigned* err;          // first give-up code (0 = none)
  unsigned long long* stats;   // [gridDim.x][ENG_NSTAT] wall-clock (100 MHz) stamps of the STATS instantiation, or nullptr
};

__host__ __device__ inline size_t eng_lds_bytes(int ns, int xb0, int xb1) {
  return (size_t)ns * ENG_SLOT + (size_t)xb0 + (size_t)xb1 + (size_t)ENG_RGS * ENG_KCMAX * 32 + 1024;
}

// control block (LDS)
struct EngCtl {
  unsigned landed;                 // tiles whose DMA has landed
  unsigned slot_done[16];          // slot s: T + 1 once tile T (T % ns == s) is consumed
  unsigned in_ready[ENG_MAX_OPS];  // k chunks of op o's input staged
  unsigned wave_op[ENG_NCONS];     // ops each consumer wave has left behind
  unsigned cnt[ENG_RGS];           // arrivals per row-group buffer
  unsigned gathering;              // 1 while the gatherer sweeps (the loader thins itself)
  unsigned abort;
  unsigned res_ready;              // residual rows staged
  unsigned rope_ready;             // rotation row staged
  unsigned rgdone[ENG_MAX_OPS];    // row groups of op o this CU has finished (its own share of op o's output is published)
  unsigned base;                   // epoch base of this launch
  float res[ENG_RES_MAX];          // this CU's rows of the residual stream
  float rope[2][64];               // cos / sin row of the current position
};

__device__ __forceinline__ unsigned eng_tag(unsigned base, int edge) { return (base + (unsigned)edge) | 0x80000000u; }

// ---- schedule: which row groups CU c owns in an op, and their rows --------------------------------------------------------------
__host__ __device__ inline void eng_plan_op(EngOp& o, int G) {
  if (o.epi == EOP_SILU) {                       // round i: units i*4G + 4c .. +4  (I % 4 == 0)
    const int I = o.N >> 1, per = 4 * G;
    o.nrg_lo = I / per; o.nrg_rem = I % per; o.cmul = 4;
  } else {                                       // strided: row groups c, c + G, ...
    const int nrg = o.N >> 3;
    o.nrg_lo = nrg / G; o.nrg_rem = nrg % G; o.cmul = 1;
  }
}
__device__ __forceinline__ int eng_nrg(const EngOp& o, int c, int G) { return o.nrg_lo + (c * o.cmul < o.nrg_rem ? 1 : 0); }
// row r (0..7) of CU c's i-th row group
__device__ __forceinline__ int eng_row(const EngOp& o, const EngArgs& a, int c, int G, int i, int r) {
  if (o.epi == EOP_SILU) {
    const int u0 = i * 4 * G + 4 * c;
    return r < 4 ? u0 + r : (o.N >> 1) + u0 + (r - 4);
  }
  const int rg = c + G * i;
  if (o.epi == EOP_QKV) {
    const int sh = a.hd == 64 ? 3 : 4, half = a.hd >> 1;      // head_dim / 8 row groups per head (head_dim 64 or 128)
    const int hh = rg >> sh, p0 = (rg - (hh << sh)) * 4;
    return hh * a.hd + (r < 4 ? p0 + r : half + p0 + (r - 4));
  }
  return 8 * rg + r;
}

// ---- LDS accesses by 32-bit LDS byte address through address_space(3) pointers: every access is a ds_ instruction (pointers kept in arrays or
// read through `volatile` lose their address space and turn into flat_load ... sc0 sc1 — measured: the first build's tile loop waited on vmcnt).
// Control words are volatile (ds_read / ds_write, no fences: a workgroup-scope release would drain the loader's DMA queue).
#define ENG_AS3 __attribute__((address_space(3)))
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned eng_lds_int;       // LDS pointers are 32 bits wide in the device pass
#else
typedef size_t eng_lds_int;         // the host pass only parses these helpers
#endif
__device__ __forceinline__ unsigned ctl_ld(unsigned a) { return *(volatile ENG_AS3 unsigned*)(eng_lds_int)a; }
__device__ __forceinline__ void ctl_st(unsigned a, unsigned v) { *(volatile ENG_AS3 unsigned*)(eng_lds_int)a = v; }
__device__ __forceinline__ float lds_ldf(unsigned a) { return *(ENG_AS3 float*)(eng_lds_int)a; }
__device__ __forceinline__ void lds_stf(unsigned a, float v) { *(ENG_AS3 float*)(eng_lds_int)a = v; }
__device__ __forceinline__ f32x4 lds_ld4f(unsigned a) { return *(ENG_AS3 f32x4*)(eng_lds_int)a; }
__device__ __forceinline__ void lds_st4f(unsigned a, f32x4 v) { *(ENG_AS3 f32x4*)(eng_lds_int)a = v; }
__device__ __forceinline__ u32x4 lds_ld4u(unsigned a) { return *(ENG_AS3 u32x4*)(eng_lds_int)a; }
__device__ __forceinline__ unsigned lds_add(unsigned a, unsigned v) { return __hip_atomic_fetch_add((ENG_AS3 unsigned*)(eng_lds_int)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define ENG_CBAR() asm volatile("" ::: "memory")
#define ENG_CTL(field) (ctl + (unsigned)offsetof(EngCtl, field))

// waits until the LDS word at `p` is >= want; false on abort / give-up
__device__ __forceinline__ bool eng_wait_ge(unsigned ctl, unsigned p, unsigned want, unsigned* err, unsigned code) {
  unsigned spins = 0;
  for (;;) {
    if (ctl_ld(p) >= want) { ENG_CBAR(); return true; }
    if (ctl_ld(ENG_CTL(abort))) return false;
    if (++spins > ENG_SPIN_LIMIT) {
      ctl_st(ENG_CTL(abort), 1u);
      if ((threadIdx.x & 63) == 0) atomicCAS(err, 0u, code);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// one 1-KiB piece, non-temporal (the weights are read once): lane l's 16 bytes land at lds_dst + 16 l
__device__ __forceinline__ void eng_dma_1k(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// 8 per-lane values -> their 64-lane sums.  v[0..7] in; returns, in every lane, the total of value index 4*b3 + 2*b4 + b5 (b_k = bit k of
// the lane number).  Three halving exchanges (lane ^ 32: v_permlane32_swap, lane ^ 16: v_permlane16_swap, lane ^ 8: DPP row_ror:8), then
// the 8 lanes that share bits 3-5 are summed (quad butterflies + row_half_mirror).  Fixed order: deterministic.
// (inline asm: with the builtins hipcc 7.2 folded `r[0] + r[1]` of a swap into `2 * r[0]` — measured wrong sums; the two wait states a swap needs
// after a VALU write of its operands are the s_nop inside the string, cdna_hip_programming.md §5.7 item 2)
__device__ __forceinline__ void eng_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void eng_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float eng_reduce8(const float v[8], int lane) {
  float s[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {      // lanes 0-31 keep v[2i], lanes 32-63 keep v[2i+1]
    float p = v[2 * i], q = v[2 * i + 1];
    eng_swap32(p, q);                // p = [v[2i] lanes 0-31 | v[2i+1] lanes 0-31], q = [v[2i] lanes 32-63 | v[2i+1] lanes 32-63]
    s[i] = p + q;
  }
  float u[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {      // even 16-lane rows keep s[2j], odd rows keep s[2j+1]
    float p = s[2 * j], q = s[2 * j + 1];
    eng_swap16(p, q);
    u[j] = p + q;
  }
  const bool b3 = (lane & 8) != 0;
  const float give = b3 ? u[0] : u[1], keep = b3 ? u[1] : u[0];
  float t = keep + dpp_mov<0x128, 0xf>(give);     // row_ror:8 == lane ^ 8 within the 16-lane row
  t += dpp_mov<0xB1, 0xf>(t);                     // lane ^ 1
  t += dpp_mov<0x4E, 0xf>(t);                     // lane ^ 2
  t += dpp_mov<0x141, 0xf>(t);                    // row_half_mirror: the other quad of the 8-lane group
  return t;
}
__device__ __forceinline__ int eng_reduce8_index(int lane) { return ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1); }

// one row of a tile for one lane: 16 stored weights (two 16-byte pieces) against the lane's 16 activation values, as packed fp32 FMAs
// (v_pk_fma_f32: two FMAs per instruction) — even and odd elements accumulate separately and meet once; fixed order.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int DT>
__device__ __forceinline__ float eng_dot16(const u32x4 w0, const u32x4 w1, const f32x4 xa[2], const f32x4 xc[2]) {
  f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const f32x2 w = f32x2{pair_lo<DT>(w0[t]), pair_hi<DT>(w0[t])};
    const f32x2 x = t < 2 ? f32x2{xa[0][2 * t], xa[0][2 * t + 1]} : f32x2{xc[0][2 * (t - 2)], xc[0][2 * (t - 2) + 1]};
    acc = __builtin_elementwise_fma(w, x, acc);
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const f32x2 w = f32x2{pair_lo<DT>(w1[t]), pair_hi<DT>(w1[t])};
    const f32x2 x = t < 2 ? f32x2{xa[1][2 * t], xa[1][2 * t + 1]} : f32x2{xc[1][2 * (t - 2)], xc[1][2 * (t - 2) + 1]};
    acc = __builtin_elementwise_fma(w, x, acc);
  }
  return acc[0] + acc[1];
}

// stats layout (STATS instantiation), per CU, 100 MHz ticks since the kernel's first instruction unless noted:
//   [0..3]   loader: op k's last tile issued          [4] loader: all landed      [5] loader: ticks blocked on a full ring
//   [6..9]   gatherer: op k's input staged            [10] gatherer: ticks inside sweeps
//   [12..15] consumer 0: op k finished   [16..19] consumer 1   [20..23] consumer 2
//   [24..26] consumer w: ticks waiting for input      [27..29] consumer w: ticks waiting for tiles   [30] consumer 0: ticks in tile work
constexpr int ENG_NSTAT = 32;

template <int DT, bool STATS = false>
__global__ __launch_bounds__(ENG_THREADS, 1) void engine_kernel(const EngArgs a) {
  static_assert(DT != DT_F32, "the engine streams 16-bit weights");
  typedef elem_t<DT> E;
  extern __shared__ __attribute__((aligned(1024))) unsigned char eng_lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x, G = gridDim.x;
  const int NS = a.ns;
  // LDS map (byte addresses): ring | xb[0] | xb[1] | part [RGS][KCMAX][8] f32 | EngCtl
  const unsigned ring = (unsigned)(size_t)eng_lds;
  const unsigned xb0 = ring + (unsigned)NS * ENG_SLOT, xb1 = xb0 + (unsigned)a.xb_bytes[0];
  const unsigned part = xb1 + (unsigned)a.xb_bytes[1];
  const unsigned ctl = part + ENG_RGS * ENG_KCMAX * 32;

  // ONE workgroup barrier, executed by every wave at the top of its role (s_barrier counts arrivals; the roles are wave-uniform branches):
  // ahead of it the gatherer issues its first loads and zeroes the control block, behind it the loader starts its burst.
  unsigned long long t_begin = 0;
  if (STATS) t_begin = wall_clock64();
  unsigned long long* st = (STATS && a.stats) ? a.stats + (size_t)c * ENG_NSTAT : nullptr;
  auto stamp = [&](int k) { if (STATS && st && lane == 0) st[k] = wall_clock64() - t_begin; };

  if (wv == 0) {
    // =============================================== loader ===============================================================
    __builtin_amdgcn_s_barrier();
    unsigned T = 0, oldest = 0;       // tiles issued / tiles published as landed
    unsigned slot = 0;                // T mod NS
    unsigned long long t_stall = 0;
    auto publish_to = [&](unsigned upto) { oldest = upto; ctl_st(ENG_CTL(landed), upto); };
    for (int oi = 0; oi < a.nops; oi++) {
      const EngOp& o = a.op[oi];
      const int nrg = eng_nrg(o, c, G), nch = o.K >> 10;
      const unsigned char* Wb = static_cast<const unsigned char*>(o.W);
      const size_t row_bytes = (size_t)o.K * 2;
      for (int i = 0; i < nrg; i++) {
        const unsigned char* rowp[8];
#pragma unroll
        for (int r = 0; r < 8; r++) rowp[r] = Wb + (size_t)eng_row(o, a, c, G, i, r) * row_bytes + (size_t)lane * 16;
        for (int kc = 0; kc < nch; kc++) {
          if (T >= (unsigned)NS) {
            const unsigned want = T - (unsigned)NS + 1u;
            if (ctl_ld(ENG_CTL(slot_done) + 4u * slot) < want) {
              // ring full: everything in flight lands and is published before the loader blocks (the consumers may be waiting for it)
              unsigned long long t0 = 0;
              if (STATS) t0 = wall_clock64();
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              publish_to(T);
              eng_wait_ge(ctl, ENG_CTL(slot_done) + 4u * slot, want, a.err, 0x100u + (unsigned)oi);
              if (STATS) t_stall += wall_clock64() - t0;
            }
          }
          if (a.thin && ctl_ld(ENG_CTL(gathering))) {    // this CU sweeps granules: its loads queue behind this wave's fills (gather-pass)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish_to(T);
            if (a.thin == 2) {                           // pause until the sweep is over
              unsigned spins = 0;
              while (ctl_ld(ENG_CTL(gathering)) && !ctl_ld(ENG_CTL(abort)) && ++spins < ENG_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
            }
          }
          const unsigned dst = ring + slot * (unsigned)ENG_SLOT;
          const size_t kofs = (size_t)kc * 2048;
#pragma unroll
          for (int r = 0; r < 8; r++) {
            eng_dma_1k(rowp[r] + kofs, dst + (unsigned)(r * 2) * 1024u);
            eng_dma_1k(rowp[r] + kofs + 1024, dst + (unsigned)(r * 2 + 1) * 1024u);
          }
          T++;
          slot = slot + 1u == (unsigned)NS ? 0u : slot + 1u;
          if (T - oldest >= (unsigned)a.depth) {      // the oldest fill in flight has landed once at most depth - 1 younger ones remain
            if (a.depth >= 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            else if (a.depth == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            publish_to(oldest + 1u);
          }
        }
      }
      stamp(oi);
    }
    // drain
    if (T - oldest == 3u) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); publish_to(oldest + 1u); }
    if (T - oldest == 2u) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); publish_to(oldest + 1u); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_to(T);
    stamp(4);
    if (STATS && st && lane == 0) st[5] = t_stall;
    return;
  }

  if (wv == 4) {
    // =============================================== gatherer ============================================================
    // The gatherer's first loads go out before anything else in the workgroup touches memory — the epoch base, this CU's rows of the residual
    // stream, the position, and op 0's input vector when an earlier launch wrote it: they are ahead of the loader's first burst in this CU's
    // memory queue (behind it they waited 2-4 us).  Then it zeroes the control block and meets the other waves at the barrier.
    unsigned long long t_sweep = 0;
    int nres = 0;
    bool has_qkv = false;
    for (int oi = a.nops - 1; oi >= 0; oi--) { if (a.op[oi].epi == EOP_RESID) nres = eng_nrg(a.op[oi], c, G) * 8; has_qkv |= a.op[oi].epi == EOP_QKV; }
    f32x4 g_pv[8][2];
    const bool g_early = a.op[0].in_plain != nullptr && (a.op[0].K >> 9) <= 8;
    if (g_early) {
      const int n512 = a.op[0].K >> 9;
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j < n512) {
          const f32x4* src = reinterpret_cast<const f32x4*>(a.op[0].in_plain + (size_t)j * 512 + lane * 8);
          g_pv[j][0] = src[0]; g_pv[j][1] = src[1];
        }
    }
    // (vector loads: a scalar load of the uniform words would sit in lgkmcnt and hold the barrier below behind a memory round trip)
    unsigned vzero = 0;
    asm volatile("" : "+v"(vzero));               // a VGPR index keeps the loads on the vector path (global_load, vmcnt)
    const unsigned base = a.epoch[vzero];
    const float resv = lane < nres ? a.x_in[8 * (c + G * (lane >> 3)) + (lane & 7)] : 0.f;
    const int gpos = has_qkv ? a.pos[vzero] : 0;
    for (int i = lane; i < (int)(sizeof(EngCtl) / 4); i += 64) ctl_st(ctl + 4u * (unsigned)i, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // op 0's early input: staged raw now (its registers die here), the norm / publish happens in the op loop
    float ss_early = 0.f;
    if (g_early) {
      const int n512 = a.op[0].K >> 9;
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j < n512) {
          const f32x4 v0 = g_pv[j][0], v1 = g_pv[j][1];
#pragma unroll
          for (int t = 0; t < 4; t++) { ss_early = fmaf(v0[t], v0[t], ss_early); }
#pragma unroll
          for (int t = 0; t < 4; t++) { ss_early = fmaf(v1[t], v1[t], ss_early); }
          lds_st4f(xb0 + (unsigned)j * 2048u + (unsigned)lane * 16u, v0);
          lds_st4f(xb0 + (unsigned)j * 2048u + 1024u + (unsigned)lane * 16u, v1);
        }
    }
    bool prologue_done = false;
    auto finish_prologue = [&]() {
      if (prologue_done) return;
      prologue_done = true;
      if (lane == 0) ctl_st(ENG_CTL(base), base);
      if (lane < nres) lds_stf(ENG_CTL(res) + 4u * (unsigned)lane, resv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ctl_st(ENG_CTL(res_ready), 1u);
      if (has_qkv) {
        const int half = a.hd >> 1;
        if (lane < half) {
          lds_stf(ENG_CTL(rope) + 4u * (unsigned)lane, a.rope_cos[(size_t)gpos * half + lane]);
          lds_stf(ENG_CTL(rope) + 256u + 4u * (unsigned)lane, a.rope_sin[(size_t)gpos * half + lane]);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ctl_st(ENG_CTL(rope_ready), 1u);
      }
    };
    for (int oi = 0; oi < a.nops; oi++) {
      const EngOp& o = a.op[oi];
      const unsigned xb = (oi & 1) ? xb1 : xb0;
      const int nch = o.K >> 10, n512 = o.K >> 9;
      if (oi >= 2) {      // the buffer was op oi-2's input: every consumer must have left that op
        for (int w = 0; w < ENG_NCONS; w++) eng_wait_ge(ctl, ENG_CTL(wave_op) + 4u * (unsigned)w, (unsigned)(oi - 1), a.err, 0x200u + (unsigned)oi);
      }
      // norm weight slices of this lane: in flight before the sweep (K <= 4096 for normed inputs)
      Slice8<DT> nw[8];
      if (o.norm_w) {
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n512) nw[j] = load_slice<DT>(static_cast<const E*>(o.norm_w), (size_t)j * 64 + lane);
      }
      float ss = 0.f;
      if (o.in_plain) {
        // written by an earlier launch: plain loads, eight 512-element slices (one round trip) at a time, all in flight before the first use
        if (oi == 0 && g_early) ss = ss_early;      // fetched ahead of the barrier, staged right behind it
        else for (int j0 = 0; j0 < n512; j0 += 8) {
          f32x4 pv[8][2];
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (j0 + j < n512) {
              const f32x4* src = reinterpret_cast<const f32x4*>(o.in_plain + (size_t)(j0 + j) * 512 + lane * 8);
              pv[j][0] = src[0]; pv[j][1] = src[1];
            }
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (j0 + j < n512) {
              const f32x4 v0 = pv[j][0], v1 = pv[j][1];
#pragma unroll
              for (int t = 0; t < 4; t++) { ss = fmaf(v0[t], v0[t], ss); }
#pragma unroll
              for (int t = 0; t < 4; t++) { ss = fmaf(v1[t], v1[t], ss); }
              lds_st4f(xb + (unsigned)(j0 + j) * 2048u + (unsigned)lane * 16u, v0);
              lds_st4f(xb + (unsigned)(j0 + j) * 2048u + 1024u + (unsigned)lane * 16u, v1);
            }
        }
        if (!o.norm_w) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ctl_st(ENG_CTL(in_ready) + 4u * (unsigned)oi, (unsigned)nch); finish_prologue(); }
      } else {
        finish_prologue();
        const unsigned tag = eng_tag(base, o.in_tag);
        const int nrg_prev = oi > 0 ? eng_nrg(a.op[oi - 1], c, G) : 0;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(o.in_gran), 0, 0x7fffffff, 0x00020000);
        // a normed input is needed whole (the norm's scale): its chunks are swept two per pass; a plain input (h) is staged chunk by chunk
        const int cpp = (o.norm_w && (nch & 1) == 0) ? 2 : 1;
        for (int ch = 0; ch < nch; ch += cpp) {
          u32x4 d[2][2][4];
          unsigned spins = 0;
          unsigned long long t0 = 0;
          if (STATS) t0 = wall_clock64();
          // sweep only once this CU's own share of the chunk is out (the CUs run in step: the others' shares are then out or about to be);
          // the last sweep of an edge is the urgent one — consumers are idle behind it — and thins this CU's loader
          const bool urgent = ch + cpp >= nch;
          if (oi > 0) eng_wait_ge(ctl, ENG_CTL(rgdone) + 4u * (unsigned)(oi - 1), (unsigned)min(o.norm_w ? nrg_prev : ch + cpp, nrg_prev), a.err, 0x700u + (unsigned)oi);
          if (a.thin && urgent) ctl_st(ENG_CTL(gathering), 1u);
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 2; q++)
              if (q < cpp) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                  for (int t = 0; t < 4; t++)
                    d[q][hf][t] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((((ch + q) * 2 + hf) * 512 + lane * 8 + t * 2) * 8), 0, 16 /*sc1*/);
              }
#pragma unroll
            for (int q = 0; q < 2; q++)
              if (q < cpp) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                  for (int t = 0; t < 4; t++) ok &= (d[q][hf][t][1] == tag) & (d[q][hf][t][3] == tag);
              }
            if (__all(ok)) break;
            if (ctl_ld(ENG_CTL(abort))) break;
            if (++spins > ENG_SPIN_LIMIT / 4) { ctl_st(ENG_CTL(abort), 1u); if (lane == 0) atomicCAS(a.err, 0u, 0x300u + (unsigned)oi); break; }
            __builtin_amdgcn_s_sleep(2);
          }
          if (a.thin && urgent) ctl_st(ENG_CTL(gathering), 0u);
          if (STATS) t_sweep += wall_clock64() - t0;
#pragma unroll
          for (int q = 0; q < 2; q++)
            if (q < cpp) {
#pragma unroll
              for (int hf = 0; hf < 2; hf++) {
                const f32x4 v0 = f32x4{__uint_as_float(d[q][hf][0][0]), __uint_as_float(d[q][hf][0][2]), __uint_as_float(d[q][hf][1][0]), __uint_as_float(d[q][hf][1][2])};
                const f32x4 v1 = f32x4{__uint_as_float(d[q][hf][2][0]), __uint_as_float(d[q][hf][2][2]), __uint_as_float(d[q][hf][3][0]), __uint_as_float(d[q][hf][3][2])};
#pragma unroll
                for (int t = 0; t < 4; t++) { ss = fmaf(v0[t], v0[t], ss); }
#pragma unroll
                for (int t = 0; t < 4; t++) { ss = fmaf(v1[t], v1[t], ss); }
                lds_st4f(xb + (unsigned)((ch + q) * 2 + hf) * 2048u + (unsigned)lane * 16u, v0);
                lds_st4f(xb + (unsigned)((ch + q) * 2 + hf) * 2048u + 1024u + (unsigned)lane * 16u, v1);
              }
            }
          if (!o.norm_w) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ctl_st(ENG_CTL(in_ready) + 4u * (unsigned)oi, (unsigned)(ch + cpp)); }
        }
      }
      if (o.norm_w) {     // x_hat = w * (x * inv_rms), in place (every lane rewrites exactly the values it staged)
        const float tot = wave_sum(ss);
        const float inv = 1.0f / sqrtf(tot / (float)o.K + a.eps);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (j < n512) {
            float w[8];
            slice_unpack<DT>(nw[j], w);
            const unsigned p0 = xb + (unsigned)j * 2048u + (unsigned)lane * 16u, p1 = p0 + 1024u;
            f32x4 v0 = lds_ld4f(p0), v1 = lds_ld4f(p1);
#pragma unroll
            for (int t = 0; t < 4; t++) { v0[t] = w[t] * (v0[t] * inv); v1[t] = w[4 + t] * (v1[t] * inv); }
            lds_st4f(p0, v0); lds_st4f(p1, v1);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ctl_st(ENG_CTL(in_ready) + 4u * (unsigned)oi, (unsigned)nch);
      }
      finish_prologue();
      stamp(6 + oi);
    }
    // the next launch's tags: bumped by CU 0 once its last gather is complete (every CU has read the base by then: it published granules)
    if (c == 0 && lane == 0) {
      int nedge = 0;
      for (int oi = 0; oi < a.nops; oi++) if (a.op[oi].out_gran) nedge++;
      if (nedge) *reinterpret_cast<volatile unsigned*>(a.epoch) = base + (unsigned)nedge;
    }
    if (STATS && st && lane == 0) st[10] = t_sweep;
    return;
  }

  // ================================================= consumers ===========================================================
  __builtin_amdgcn_s_barrier();
  const int cw = wv - 1;
  const int pos = *a.pos;
  unsigned T = 0, slot = 0, rgslot = 0;
  int turn = 0;                       // T mod ENG_NCONS
  unsigned long long t_wait_in = 0, t_wait_land = 0, t_busy = 0;
  for (int oi = 0; oi < a.nops; oi++) {
    const EngOp& o = a.op[oi];
    const int nrg = eng_nrg(o, c, G), nch = o.K >> 10;
    const unsigned xb = (oi & 1) ? xb1 : xb0;
    for (int i = 0; i < nrg; i++, rgslot = (rgslot + 1u) & (unsigned)(ENG_RGS - 1)) {
      const unsigned prg = part + rgslot * (unsigned)(ENG_KCMAX * 32);
      for (int kc = 0; kc < nch; kc++, T++, slot = (slot + 1u == (unsigned)NS ? 0u : slot + 1u), turn = (turn + 1 == ENG_NCONS ? 0 : turn + 1)) {
        if (turn != cw) continue;
        unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if (STATS) t0 = wall_clock64();
        eng_wait_ge(ctl, ENG_CTL(in_ready) + 4u * (unsigned)oi, o.norm_w ? (unsigned)nch : (unsigned)(kc + 1), a.err, 0x400u + (unsigned)oi);
        if (STATS) t1 = wall_clock64();
        eng_wait_ge(ctl, ENG_CTL(landed), T + 1u, a.err, 0x500u + (unsigned)oi);
        if (STATS) t2 = wall_clock64();
        const unsigned tb = ring + slot * (unsigned)ENG_SLOT + (unsigned)lane * 16u;
        f32x4 xa[2], xc[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          xa[hf] = lds_ld4f(xb + (unsigned)(kc * 2 + hf) * 2048u + (unsigned)lane * 16u);
          xc[hf] = lds_ld4f(xb + (unsigned)(kc * 2 + hf) * 2048u + 1024u + (unsigned)lane * 16u);
        }
        // all 16 weight reads of the tile are issued before the first FMA: the consumer is alone on its SIMD, nothing else hides LDS latency
        u32x4 wr[16];
#pragma unroll
        for (int p = 0; p < 16; p++) wr[p] = lds_ld4u(tb + (unsigned)p * 1024u);
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = eng_dot16<DT>(wr[2 * r], wr[2 * r + 1], xa, xc);
        const float tot = eng_reduce8(acc, lane);
        if ((lane & 7) == 0) lds_stf(prg + (unsigned)(kc * 8 + eng_reduce8_index(lane)) * 4u, tot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned old = 0;
        if (lane == 0) old = lds_add(ENG_CTL(cnt) + 4u * rgslot, 1u);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == (unsigned)(nch - 1)) {
          // ---- this wave completed the row group: sum the k-chunk partials in order, epilogue, publish -------------------------
          if (lane == 0) ctl_st(ENG_CTL(cnt) + 4u * rgslot, 0u);
          ENG_CBAR();
          float y = 0.f, y2 = 0.f;
          if (lane < 8) {
            for (int k = 0; k < nch; k++) y += lds_ldf(prg + (unsigned)(k * 8 + lane) * 4u);
          }
          if (o.epi != EOP_RESID && lane < 4) {
            for (int k = 0; k < nch; k++) y2 += lds_ldf(prg + (unsigned)(k * 8 + lane + 4) * 4u);
          }
          eng_wait_ge(ctl, ENG_CTL(res_ready), 1u, a.err, 0x600u);      // the gatherer's prologue (epoch base, residual rows) is in LDS
          const unsigned base = ctl_ld(ENG_CTL(base));
          if (o.epi == EOP_RESID) {
            if (lane < 8) {
              const int n = 8 * (c + G * i) + lane;
              if (o.bias) y += elem_to_f32<DT>(static_cast<const E*>(o.bias)[n]);
              const unsigned ra = ENG_CTL(res) + 4u * (unsigned)(i * 8 + lane);
              const float xn = lds_ldf(ra) + y;
              lds_stf(ra, xn);
              if (o.out_gran) __hip_atomic_store(o.out_gran + n, (u64)__float_as_uint(xn) | ((u64)eng_tag(base, o.out_tag) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (o.out_plain) o.out_plain[n] = xn;
            }
          } else if (o.epi == EOP_SILU) {
            if (lane < 4) {
              const int u = i * 4 * G + 4 * c + lane;
              const float hv = (y / (1.0f + expf(-y))) * y2;
              if (o.out_gran) __hip_atomic_store(o.out_gran + u, (u64)__float_as_uint(hv) | ((u64)eng_tag(base, o.out_tag) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (o.out_plain) o.out_plain[u] = hv;
            }
          } else {   // EOP_QKV: bias, rotate-half RoPE at `pos`, q -> q_out, k / v -> this position's cache row
            if (lane < 4) {
              const int rg = c + G * i, sh = a.hd == 64 ? 3 : 4, half = a.hd >> 1;
              const int hh = rg >> sh, p = (rg - (hh << sh)) * 4 + lane;
              float va = y, vb = y2;
              if (o.bias) { const E* b = static_cast<const E*>(o.bias); va += elem_to_f32<DT>(b[hh * a.hd + p]); vb += elem_to_f32<DT>(b[hh * a.hd + half + p]); }
              const bool is_q = hh < a.heads, is_k = !is_q && hh < a.heads + a.kv_heads;
              if (is_q || is_k) {
                eng_wait_ge(ctl, ENG_CTL(rope_ready), 1u, a.err, 0x601u);
                const float cs = lds_ldf(ENG_CTL(rope) + 4u * (unsigned)p), sn = lds_ldf(ENG_CTL(rope) + 256u + 4u * (unsigned)p);
                const float ra = va * cs - vb * sn, rb = vb * cs + va * sn;
                va = ra; vb = rb;
              }
              if (is_q) { a.q_out[hh * a.hd + p] = va; a.q_out[hh * a.hd + half + p] = vb; }
              else {
                E* dst = is_k ? static_cast<E*>(a.k_cache) + ((size_t)(hh - a.heads) * a.max_ctx + pos) * a.hd
                              : static_cast<E*>(a.v_cache) + ((size_t)(hh - a.heads - a.kv_heads) * a.max_ctx + pos) * a.hd;
                dst[p] = f32_to_elem<DT>(va); dst[half + p] = f32_to_elem<DT>(vb);
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) lds_add(ENG_CTL(rgdone) + 4u * (unsigned)oi, 1u);
        }
        ctl_st(ENG_CTL(slot_done) + 4u * slot, T + 1u);
        if (STATS) { const unsigned long long t3 = wall_clock64(); t_wait_in += t1 - t0; t_wait_land += t2 - t1; t_busy += t3 - t2; }
      }
    }
    ctl_st(ENG_CTL(wave_op) + 4u * (unsigned)cw, (unsigned)(oi + 1));
    stamp(12 + 4 * cw + oi);
  }
  if (STATS && st && lane == 0) {
    st[24 + cw] = t_wait_in; st[27 + cw] = t_wait_land;
    if (cw == 0) st[30] = t_busy;
  }
}

}  // namespace tgx
