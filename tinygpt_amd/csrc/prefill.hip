// prefill.hip — the batched prefill of the MI355X shim for 16-bit storage (kernels/prefill.h, gemm_dma.h): every product of S prompt positions on the
// matrix cores, causal flash attention, RoPE + cache append.  == CausalLM::forward on [B,S] ids with an empty cache (GPTModel.h:51-56)
#include "ctx.h"
#include <type_traits>
#include "kernels/prefill.h"
#include "kernels/gemm_dma.h"
#include "kernels/gemm_dma_qkv.h"
#include "kernels/attn_prefill_dma.h"
#include "kernels/gemm_f32.h"

bool prefill_shapes_ok(const tgx_model_desc& d) {
  return d.hidden % 64 == 0 && (d.heads * d.head_dim) % 64 == 0 && d.inter % 64 == 0;
}

int ensure_prefill_ws(tgx_ctx* c, int S) {
  if (S <= c->ws_rows && (!c->act16 || c->ws_zero || c->dt == tgx::DT_F32)) return TGX_OK;
  S = std::max(S, c->ws_rows);
  const tgx_model_desc& d = c->d;
  const size_t H = (size_t)d.hidden, qd = (size_t)d.heads * d.head_dim, kvd = (size_t)d.kv_heads * d.head_dim, I = (size_t)d.inter;
  const size_t wout = qd + 2 * kvd, wa = std::max(H, qd);   // the gate_up product leaves no fp32 intermediate (GEMM_SILU)
  drop_step_graphs(c);                                       // a captured batched decode step holds pointers into the old workspace
  HIP_OK(c, hipStreamSynchronize(c->stream));
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(c->ws_x); fr(c->ws_out); fr(c->ws_ah); fr(c->ws_al); fr(c->ws_al2); fr(c->ws_qh); fr(c->ws_ql); fr(c->ws_hh); fr(c->ws_hl); fr(c->ws_zero);
  c->ws_al2 = nullptr; c->ws_zero = nullptr; c->ws_zero_elems = 0;
  c->ws_x = nullptr; c->ws_out = nullptr; c->ws_ah = c->ws_al = c->ws_qh = c->ws_ql = c->ws_hh = c->ws_hl = nullptr; c->ws_rows = 0;
  const size_t rows = (size_t)S;
  // fp32 storage: ws_ah / ws_qh / ws_hh hold fp32 rows (the fp32 GEMM's A operands, the rotated queries); the lo terms are unused
  const size_t te = c->dt == tgx::DT_F32 ? 4 : 2, lo = c->dt == tgx::DT_F32 ? 0 : 1;
  HIP_OK(c, hipMalloc((void**)&c->ws_x, rows * H * 4));
  HIP_OK(c, hipMalloc((void**)&c->ws_out, rows * wout * 4));
  HIP_OK(c, hipMalloc((void**)&c->ws_ah, rows * wa * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_al, rows * wa * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_al2, rows * H * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_qh, rows * qd * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_ql, rows * qd * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_hh, rows * I * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_hl, rows * I * 2 * lo + 16));
  if (c->act16 && lo) {      // option act.round16: the all-zero "lo term" of every stored-term product (any A operand fits: rows x max(H, qd, I))
    c->ws_zero_elems = rows * std::max(wa, I) + 8;
    HIP_OK(c, hipMalloc((void**)&c->ws_zero, c->ws_zero_elems * 2));
    HIP_OK(c, hipMemset(c->ws_zero, 0, c->ws_zero_elems * 2));
  }
  if (c->dt == tgx::DT_F32) {
    if (c->ws_pos) (void)hipFree(c->ws_pos);
    c->ws_pos = nullptr;
    HIP_OK(c, hipMalloc((void**)&c->ws_pos, rows * 4));
  }
  c->ws_rows = S;
  return TGX_OK;
}

// defer (optional, RESIDUAL / STORE products): when the product is split over K, leave the slabs in ws_part for the consumer kernel to sum
// (rmsnorm_split_kernel / rope_kv_split_kernel: same z order, one launch and one pass over the rows less) and report the slab count; 1 = done here.
// -1 (round 5): the UNSPLIT eight-wave N = hidden product stored its result as ONE slab instead of adding it to the residual
// stream in its epilogue — a read-modify-write of M x N floats by four waves per CU at the end of a launch that has one tile per CU (nothing left to overlap
// it: 18-19 of o_proj's 52 us, tools/probes/gemm_lab.hip); the row-wise norm kernel that reads the stream next adds the slab while it streams.
// The gate_up product on full 128-byte lines (kernels/gemm_dma.h gemm_dma8i_kernel): taken where the 256 x 256 kernel would be, on the
// default contract; the producing norm launch then writes the two terms interleaved per k32 block (rmsnorm_split_kernel `inter`) into ws_out, which is idle
// between the RoPE / cache-append launch and the next layer's QKV product.
static bool gemm_full_lines(const tgx_ctx* c, int epi, int M, int N, int K) {
  if (epi != tgx::GEMM_SILU || c->gpt2 || (c->act16 && c->ws_zero) || !(c->gemm_dma & 4) || K % 64 != 0 || N % 256 != 0) return false;
  const tgx_model_desc& d = c->d;
  if ((size_t)d.heads * d.head_dim + 2 * (size_t)d.kv_heads * d.head_dim < (size_t)K) return false;      // ws_out holds [M][2 K] 16-bit terms
  const int t256 = ((N + 255) / 256) * ((M + 255) / 256), r256 = (t256 + c->num_cus - 1) / c->num_cus;
  const bool ragged256 = c->wide_8k_eff > 0 && (c->gemm_dma & 8) && t256 >= c->num_cus && 100 * t256 < c->wide_8k_eff * r256 * c->num_cus;
  return t256 >= c->num_cus && !ragged256;
}

static void launch_gemm(tgx_ctx* c, int epi, const ebyte* B_, const ebyte* bias_, float* C, int M, int N, int K, int ldc, bool three_terms = false,
                 const bf16_t* a_hi = nullptr, const bf16_t* a_lo = nullptr, int three_from = 0, int* defer = nullptr, bool a_inter = false) {
  if (defer) *defer = 1;
  if (a_inter) {       // A arrives interleaved in a_hi (the caller asked gemm_full_lines first)
    tgx::GemmArgs g{};
    g.A_hi = a_hi; g.A_lo = nullptr; g.A_lo2 = nullptr; g.inter = N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
    g.B = reinterpret_cast<const bf16_t*>(B_); g.bias = nullptr; g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.three_from = 1 << 30;
    const dim3 g8((N + 255) / 256, (M + 255) / 256), b8(512);
    TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma8i_kernel<DT, tgx::GEMM_SILU>), g8, b8, (size_t)5 * 256 * 128, c->stream, g))
    return;
  }
  const bool one = c->act16 && c->ws_zero;       // option act.round16: A_hi is the (rounded) activation; every other term reads zeros, the eight-wave kernels skip it
  if (one) { three_terms = false; three_from = 0; }
  const bool one_k = one && c->act16_kernels;    // ... in kernels that have a one-term form
  const bf16_t* B = reinterpret_cast<const bf16_t*>(B_);   // 16-bit storage (bf16 or fp16 bit patterns); fp32 storage never gets here
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(bias_);
  tgx::GemmArgs g{};
  g.A_hi = a_hi ? a_hi : c->ws_ah; g.A_lo = one ? c->ws_zero : (a_lo ? a_lo : c->ws_al); g.A_lo2 = three_terms ? c->ws_al2 : nullptr;
  g.inter = N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  g.B = B; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.three_from = three_from;
  // few column tiles (N = hidden) -> 64-row tiles, so that at least two workgroups share a CU
  const bool few = ((N + tgx::GBN - 1) / tgx::GBN) * ((M + tgx::GBM - 1) / tgx::GBM) < 2 * c->num_cus;
  // measured (tools/prefill_bench.py --gemm-tm, Llama-3.2-1B, S = 2048): this policy 15.0 ms, 64-row tiles also for the three-term
  // QKV product 15.3, 128-row tiles everywhere 15.85, 64-row tiles everywhere 15.9
  const bool small = (epi == tgx::GEMM_SILU || epi == tgx::GEMM_GELU) ? false : (c->gemm_tm ? c->gemm_tm == 64 : (few && !three_terms));
  const int tm = small ? 64 : tgx::GBM;
  const dim3 grid((N + tgx::GBN - 1) / tgx::GBN, (M + tm - 1) / tm), blk(256);
  const size_t dyn = three_terms ? (size_t)tm * tgx::GLD * 2 : 0;      // LDS tile of the third term
  // few row tiles (a short prompt): the tiles alone cannot stream the weights at rate (S <= 96 cost a flat 4.7 ms on Llama-3.2-1B) —
  // split K over blockIdx.z until ~2 workgroups per CU exist; the slabs are summed in z order by a second launch (deterministic)
  const int ntiles = (int)(grid.x * grid.y), ktiles = K / tgx::GBK;
  int nsplit = 1;
  // (the balanced QKV launch below fills the chip by itself from 1024 rows on at hidden 2048: 128 + 128 workgroups of equal work; Llama-3.2-1B S = 1024 5.16 -> 5.06 ms — no slabs then; option prefill.qkv_nosplit)
  const bool qkv_bal = (c->gemm_dma & 3) && three_terms && epi == tgx::GEMM_STORE && three_from > 0 && three_from % tgx::GBN == 0 && N > three_from && K % 64 == 0 && M >= 128;
  const int qkv_wgs = qkv_bal ? (three_from / tgx::GBN) * ((M + 127) / 128) + ((N - three_from + tgx::GBN - 1) / tgx::GBN) * ((M + 63) / 64) : 0;
  const bool qkv_nosplit = qkv_bal && c->qkv_nosplit && qkv_wgs >= c->num_cus;
  if (c->gemm_splitk && ntiles < c->num_cus && K % tgx::GBK == 0 && !qkv_nosplit) nsplit = std::min(std::min(16, ktiles), (2 * c->num_cus + ntiles - 1) / ntiles);
  // N = hidden products of a 129-1500-row prompt (round 4, option prefill.splitk_8k): their 128 x 128 tiles number less than a chip (Llama-3.2-1B S = 1024: 128 tiles on 256
  // CUs, 84 us per product against 104 at twice the rows) — the eight-wave LDS-DMA kernel over 2-4 K slabs instead of 64-row register-staged slabs or a half-empty chip
  bool part_8k = false;
  if (c->gemm_splitk && c->splitk_8k && (c->gemm_dma & 8) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE) && M > 128) {
    const int t128 = ((N + 127) / 128) * ((M + 127) / 128);
    // slabs z = 2..4 that shorten the critical path: rounds of workgroups x 1 / z of the K loop against one round of whole tiles (160 tiles on 256 CUs: z = 3 -> 2 rounds of a
    // third = 0.67; 128 tiles: z = 2 -> 0.5); taken from 0.8 down (the slabs cost a pass over z x M x N floats)
    int z = 1; double best = 0.8001;
    for (int zz = 2; zz <= 4; zz++) {
      const double cost = (double)((zz * t128 + c->num_cus - 1) / c->num_cus) / zz;
      if (t128 < c->num_cus && cost < best - 1e-9 && K / 64 >= 4 * zz) { best = cost; z = zz; }
    }
    if (z >= 2) { part_8k = true; nsplit = z; }
  }
  // K >> N (`down`) on 128 x 256 tiles x 2 K slabs: a third fewer operand lines per output than the 128 x 128 kernel, which waits for them (kernels/gemm_dma.h
  // gemm_dma8n_kernel).  Needs a consumer that sums pending slabs (`defer`) and two rounds of workgroups (prefill.wide_n_min, in chips): at one
  // round (Llama-3.2-1B S = 2048: 256 workgroups) it ties with the one-slab 128 x 128 launch (141 vs 140-146 us) and costs a second slab; Mistral-7B S = 2048 56.1 -> 54.1 ms
  if (nsplit == 1 && defer && c->defer_reduce && epi == tgx::GEMM_RESIDUAL && (c->gemm_dma & 8) && !three_terms && !one && K >= 2 * N && K % 128 == 0 && N % 256 == 0 &&
      2 * (N / 256) * ((M + 127) / 128) >= c->wide_n_min * c->num_cus) {
    const size_t need = (size_t)2 * M * N * 4;
    if (need > c->ws_part_bytes) {
      drop_step_graphs(c);
      (void)hipStreamSynchronize(c->stream);
      if (c->ws_part) (void)hipFree(c->ws_part);
      c->ws_part = nullptr; c->ws_part_bytes = 0;
      if (hipMalloc((void**)&c->ws_part, need) == hipSuccess) c->ws_part_bytes = need;
    }
    if (c->ws_part_bytes >= need) {
      g.part = c->ws_part; g.nsplit = 2; g.interleave = 0; g.k_per = K / 2;
      const dim3 g8(N / 256, (M + 127) / 128, 2), b8(512);
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma8n_kernel<DT>), g8, b8, (size_t)2 * (2 * 128 + 256) * 64 * 2, c->stream, g))
      *defer = 2;
      return;
    }
  }
  // one tile per CU on the eight-wave 128 x 128 kernel, a consumer that can take a pending slab: store the product, let the consumer add it (see `defer` above)
  const int t128u = ((N + 127) / 128) * ((M + 127) / 128);
  const bool store_slab = nsplit == 1 && defer && c->defer_reduce && epi == tgx::GEMM_RESIDUAL && (c->gemm_dma & 8) && K % 64 == 0 && !three_terms &&
                          2 * t128u >= c->num_cus && 2 * t128u <= 3 * c->num_cus &&
                          !((c->gemm_dma & 4) && ((N + 255) / 256) * ((M + 255) / 256) >= c->num_cus);
  if (nsplit > 1 || store_slab) {
    const size_t need = (size_t)nsplit * M * N * 4;
    if (need > c->ws_part_bytes) {
      drop_step_graphs(c);              // a captured batched decode step points into the old slab buffer
      (void)hipStreamSynchronize(c->stream);
      if (c->ws_part) (void)hipFree(c->ws_part);
      c->ws_part = nullptr; c->ws_part_bytes = 0;
      if (hipMalloc((void**)&c->ws_part, need) == hipSuccess) c->ws_part_bytes = need; else nsplit = 1;
    }
  }
  if (store_slab && c->ws_part_bytes >= (size_t)M * N * 4) {
    g.part = c->ws_part; g.nsplit = 1; g.interleave = 0; g.k_per = K;
    const dim3 g8((N + 127) / 128, (M + 127) / 128, 1), b8(512);
    const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
    TGX_DT16_SWITCH(c->dt, if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_PARTIAL, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_PARTIAL>), g8, b8, lds8, c->stream, g);)
    *defer = -1;
    return;
  }
  if (nsplit > 1) {
    g.part = c->ws_part; g.nsplit = nsplit; g.interleave = epi == tgx::GEMM_SILU ? 1 : 0;
    g.k_per = part_8k ? ((K / 64 + nsplit - 1) / nsplit) * 64 : ((ktiles + nsplit - 1) / nsplit) * tgx::GBK;
    const dim3 gz(grid.x, grid.y, nsplit);
    const size_t nout = (size_t)M * (epi == tgx::GEMM_SILU ? N / 2 : N);
    const dim3 rg((unsigned)((nout + 255) / 256));
    // the slabs' GEMM: operand tiles by LDS-DMA (round 3) — the register-staged kernel streamed a short prompt's weights at
    // 1-2 TB/s (S = 48: gate_up 32 us for 67 MB); 64-row tiles whenever the prompt fits them, k = 64 per stage (32 for the 128-row three-term tile)
    // measured (Llama-3.2-1B, ms per prompt, DMA vs register-staged slabs): S = 40 1.61 / 1.79, 48 1.65 / 1.76, 64 1.71 / 1.87; 96 2.08 / 1.99, 128 2.12 / 2.09,
    // 256 2.65 / 2.68; Mistral-7B S = 48 5.40 / 6.23 — the 64-row tile wins, the 128-row one does not: prompts of <= 64 rows only (value 2 = always)
    if (part_8k) {
      const dim3 g8((N + 127) / 128, (M + 127) / 128, nsplit), b8(512);
      const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
      TGX_DT16_SWITCH(c->dt, if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_PARTIAL, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_PARTIAL>), g8, b8, lds8, c->stream, g);)
    }
    const bool dma_part = !part_8k && M <= 64 && (c->gemm_dma & 3) && g.k_per % 64 == 0 && K % 64 == 0;
    if (dma_part) {
      const int mi = (small || M <= 64) ? 1 : 2;
      const int dbk = (mi == 2 && three_terms) ? 32 : 64;
      const dim3 gd((N + tgx::GBN - 1) / tgx::GBN, (M + 64 * mi - 1) / (64 * mi), nsplit);
      const size_t lds = tgx::gemm_dma_lds_bytes(mi, three_terms, dbk, 2);
      TGX_DT16_SWITCH(c->dt,
        if (mi == 1) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 1, 64, 2>), gd, blk, lds, c->stream, g);
        else if (dbk == 64) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 2, 64, 2>), gd, blk, lds, c->stream, g);
        else hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 2, 32, 2>), gd, blk, lds, c->stream, g);)
    }
    TGX_DT16_SWITCH(c->dt,
      if (dma_part || part_8k) {}
      else if (small) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_PARTIAL, 1>), gz, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_PARTIAL, 2>), gz, blk, dyn, c->stream, g);
      if (defer && c->defer_reduce && M >= c->defer_min_rows && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE)) { *defer = nsplit; }   // with few rows the row-wise consumers are too few workgroups to sum 16 slabs quickly (round 2, S = 64: 1.82 -> 1.87 ms; S = 256: 2.80 -> 2.70)
      else if (epi == tgx::GEMM_SILU) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_SILU>), rg, blk, 0, c->stream, g);
      else if (epi == tgx::GEMM_GELU) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_GELU>), rg, blk, 0, c->stream, g);
      else if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_RESIDUAL>), rg, blk, 0, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_STORE>), rg, blk, 0, c->stream, g);)
    return;
  }
  // (a prompt whose 256 x 256 tiles leave the last round of workgroups mostly empty — 1152 rows at intermediate 8192: 320 tiles = 1.25 rounds, the time of 2048 rows — takes the
  //  128 x 128 kernel below instead: 1152 tiles = 4.5 rounds; option prefill.wide_8k_eff = per cent of the last round's fill below which that happens)
  const int t256 = ((N + 255) / 256) * ((M + 255) / 256), r256 = (t256 + c->num_cus - 1) / c->num_cus;
  const bool ragged256 = c->wide_8k_eff > 0 && epi == tgx::GEMM_SILU && (c->gemm_dma & 8) && t256 >= c->num_cus && 100 * t256 < c->wide_8k_eff * r256 * c->num_cus;
  if ((c->gemm_dma & 4) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_SILU || epi == tgx::GEMM_GELU) && t256 >= c->num_cus && !ragged256) {
    // the wide product (gate_up / c_fc) with enough 256 x 256 tiles to fill the chip: 8 waves, three-stage LDS-DMA ring
    const dim3 g8((N + 255) / 256, (M + 255) / 256), b8(512);
    const size_t lds8 = (size_t)3 * 3 * 256 * 32 * 2;
    TGX_DT16_SWITCH(c->dt,
      if (epi == tgx::GEMM_SILU) { if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_SILU, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_SILU>), g8, b8, lds8, c->stream, g); }
      else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_GELU>), g8, b8, lds8, c->stream, g);)
    return;
  }
  if ((c->gemm_dma & 4) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE) &&
      ((N + 255) / 256) * ((M + 255) / 256) >= c->num_cus) {
    // the N = hidden products of a prompt long enough to give every CU a 256 x 256 tile (Llama-3.2-1B from 8192 rows, Mistral-7B from 4096): the wide
    // product's kernel with the plain fp32 epilogue 
    const dim3 g8((N + 255) / 256, (M + 255) / 256), b8(512);
    const size_t lds8 = (size_t)3 * 3 * 256 * 32 * 2;
    TGX_DT16_SWITCH(c->dt,
      if (epi == tgx::GEMM_RESIDUAL) { if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_RESIDUAL, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_RESIDUAL>), g8, b8, lds8, c->stream, g); }
      else { if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_STORE, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_STORE>), g8, b8, lds8, c->stream, g); })
    return;
  }
  if ((c->gemm_dma & 8) && K % 64 == 0 && !three_terms && epi == tgx::GEMM_SILU) {
    // the wide product of a prompt too short for 256 x 256 tiles (129-384 rows: 128-384 tiles of 128 x 128): the eight-wave kernel with the K step split between
    // wave pairs instead of the four-wave one (option prefill.wide_8k: Llama-3.2-1B S = 256 gate_up 61 us per layer)
    const int t128 = ((N + 127) / 128) * ((M + 127) / 128);
    if (2 * t128 >= c->num_cus && (2 * t128 <= c->wide_8k_max * c->num_cus || ragged256)) {
      const dim3 g8((N + 127) / 128, (M + 127) / 128), b8(512);
      const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
      TGX_DT16_SWITCH(c->dt, if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_SILU, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_SILU>), g8, b8, lds8, c->stream, g);)
      return;
    }
  }
  if ((c->gemm_dma & 8) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE)) {
    // N = hidden products whose 128 x 128 tiles number between half a chip and a chip and a half: eight waves per tile, K step split between wave pairs
    const int t128 = ((N + 127) / 128) * ((M + 127) / 128);
    if (2 * t128 >= c->num_cus && 2 * t128 <= 3 * c->num_cus) {
      const dim3 g8((N + 127) / 128, (M + 127) / 128), b8(512);
      const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
      TGX_DT16_SWITCH(c->dt,
        if (epi == tgx::GEMM_RESIDUAL) { if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_RESIDUAL, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_RESIDUAL>), g8, b8, lds8, c->stream, g); }
        else { if (one_k) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_STORE, false>), g8, b8, lds8, c->stream, g); else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_STORE>), g8, b8, lds8, c->stream, g); })
      return;
    }
  }
  if ((c->gemm_dma & 3) && three_terms && epi == tgx::GEMM_STORE && three_from > 0 && three_from % tgx::GBN == 0 && N > three_from && K % 64 == 0 &&
      (N - three_from) % 64 == 0 && three_from / tgx::GBN == (N - three_from) / 64 && 2 * (three_from / tgx::GBN) * ((M + 127) / 128) >= c->num_cus) {
    // ... with as many 128-column Q tiles as 64-column K | V tiles (q_dim = 4 kv_dim) and at least half a chip of workgroups: eight waves per workgroup, the Q tile and the
    // K | V tile of a row block on ONE staging of the activation lines (kernels/gemm_dma.h gemm_dma_qkv8_kernel)
    const int nwg = (three_from / tgx::GBN) * ((M + 127) / 128);
    if (c->qkv_epi.q_hi && c->d.head_dim == 64 && !c->d.qk_norm && defer) {
      // ... and RoPE + cache append + the q split in its epilogue (one sequence, head_dim 64): no fp32 QKV matrix, no rope_kv_split launch
      g.rope_q_hi = c->qkv_epi.q_hi; g.rope_q_lo = c->qkv_epi.q_lo; g.rope_k = c->qkv_epi.k; g.rope_v = c->qkv_epi.v;
      g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.rope_past = c->qkv_epi.past; g.rope_max_ctx = c->d.max_ctx; g.rope_kv_heads = c->d.kv_heads; g.rope_tbl = c->qkv_epi.tbl;
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma_qkv8_kernel<DT, true>), dim3(nwg), dim3(512), (size_t)2 * (3 * 128 + 128 + 64) * 64 * 2, c->stream, g))
      *defer = 0;         // the rows are finished: the caller skips its RoPE / cache-append launch
      return;
    }
    TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma_qkv8_kernel<DT>), dim3(nwg), dim3(512), (size_t)2 * (3 * 128 + 128 + 64) * 64 * 2, c->stream, g))
    return;
  }
  if ((c->gemm_dma & 3) && three_terms && epi == tgx::GEMM_STORE && three_from > 0 && three_from % tgx::GBN == 0 && N > three_from && K % 64 == 0 && M >= 128) {
    // the QKV product of a bf16 prompt: Q columns as two-term 128-row tiles, K / V columns as three-term 64-row tiles, ONE launch with
    // equal work per workgroup pair (kernels/gemm_dma.h gemm_dma_qkv_kernel)
    const int nq = (three_from / tgx::GBN) * ((M + 127) / 128), nkv = ((N - three_from + tgx::GBN - 1) / tgx::GBN) * ((M + 63) / 64);
    const size_t ldsq = std::max(tgx::gemm_dma_lds_bytes(2, false, 32, 2), tgx::gemm_dma_lds_bytes(1, true, 32, 2));
    TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma_qkv_kernel<DT, 32, 2>), dim3(nq + nkv), blk, ldsq, c->stream, g))
    return;
  }
  if ((c->gemm_dma & 3) && K % 64 == 0) {     // operand tiles by LDS-DMA into a two-stage ring (kernels/gemm_dma.h): one barrier per K step
    // geometry per tile height: 128-row tiles k = 32, 64-row tiles k = 64 (round 2's sweep, profiles/r02_prefill_dma_sweep.txt: deeper rings lose 4-20 % to occupancy
    // here — the eight-wave kernels are where they pay; the other geometries went with the option bits that selected them in round 6)
    const size_t lds = small ? tgx::gemm_dma_lds_bytes(1, three_terms, 64, 2) : tgx::gemm_dma_lds_bytes(2, three_terms, 32, 2);
#define TGX_DMA(EPI_, MI_) do { if ((MI_) == 1) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, EPI_, 1, 64, 2>), grid, blk, lds, c->stream, g); \
                                else hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, EPI_, 2, 32, 2>), grid, blk, lds, c->stream, g); } while (0)
    if (lds <= 160 * 1024) {
      TGX_DT16_SWITCH(c->dt,
        if (epi == tgx::GEMM_SILU) TGX_DMA(tgx::GEMM_SILU, 2);
        else if (epi == tgx::GEMM_GELU) TGX_DMA(tgx::GEMM_GELU, 2);
        else if (epi == tgx::GEMM_RESIDUAL) { if (small) TGX_DMA(tgx::GEMM_RESIDUAL, 1); else TGX_DMA(tgx::GEMM_RESIDUAL, 2); }
        else { if (small) TGX_DMA(tgx::GEMM_STORE, 1); else TGX_DMA(tgx::GEMM_STORE, 2); })
      return;
    }
#undef TGX_DMA
  }
  TGX_DT16_SWITCH(c->dt,
    if (epi == tgx::GEMM_SILU) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_SILU, 2>), grid, blk, dyn, c->stream, g);
    else if (epi == tgx::GEMM_GELU) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_GELU, 2>), grid, blk, dyn, c->stream, g);
    else if (small) {
      if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_RESIDUAL, 1>), grid, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_STORE, 1>), grid, blk, dyn, c->stream, g);
    } else {
      if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_RESIDUAL, 2>), grid, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_STORE, 2>), grid, blk, dyn, c->stream, g);
    })
}

// causal GQA flash attention of S prompt positions of one batch row (grid = ceil(S / 128) query blocks x heads)
void launch_attn_prefill(tgx_ctx* c, const tgx::AttnPrefillArgs& a_, bool allow_lean) {
  tgx::AttnPrefillArgs a = a_;
  const int hd = c->d.head_dim;
  const int nqb = (a.S + 127) / 128, nwg = nqb * a.heads;
  const size_t lds1 = (size_t)(64 * (hd + 8) + 64 * (hd + 32)) * 2;      // one K tile | V tile pair (kernels/prefill.h)
  // head_dim 64, three or more workgroups per CU (prompts from ~3k tokens at 32 heads): K / V tiles by LDS-DMA, the next tile's scores under the current tile's
  // softmax (kernels/attn_prefill_dma.h; bit-identical to attn_prefill_kernel): S = 4096 203 -> 177 us per layer, 8192 730 -> 632; at S = 2048 (one round of 512
  // workgroups) the launch lasts as long as its heaviest workgroup's chain of tiles in either form (62-63 us).  Option prefill.attn_dma: 0 never, 1 auto, 2 always
  if (hd == 64 && (c->attn_dma == 2 || (c->attn_dma == 1 && allow_lean && nwg >= 3 * c->num_cus))) {
    a.heavy_first = 1;
    const dim3 grid(a.heads, nqb), blk(256);
    TGX_DT16_SWITCH(c->dt, if (a.blk_tbl) hipLaunchKernelGGL((tgx::attn_prefill_dma_kernel<DT, true>), grid, blk, (size_t)2 * 3 * 64 * 64 * 2, c->stream, a);
                           else hipLaunchKernelGGL((tgx::attn_prefill_dma_kernel<DT, false>), grid, blk, (size_t)2 * 3 * 64 * 64 * 2, c->stream, a))
    return;
  }
  // key split inside the workgroup (attn_prefill_kernel KP = 2; option prefill.attn_ksplit: 0 never, 1 auto, 2 always): eight waves, the odd tiles on waves 4-7,
  // one merge at the end — half the chain of tiles per wave.  head_dim 128: S = 2048 94 -> 91 us per layer, 4096 401 -> 305, 8192 1284 -> 1086 (always);
  // head_dim 64 below three workgroups per CU: S = 2048 62 -> 55-56 us in isolation, 68.1 -> 61.3 in the model's trace (profiles/r05_prefill.txt section 5)
  const bool ksplit = nqb >= 2 && (c->attn_ksplit == 2 || (c->attn_ksplit == 1 && (hd == 128 || nwg < 3 * c->num_cus)));
  // head_dim 64 with three or more workgroups per CU: the one-tile look-ahead form at three waves per SIMD (prefill.h)
  const bool lean = allow_lean && hd == 64 && nwg >= 3 * c->num_cus;
  a.heavy_first = ksplit ? 1 : 0;
  const dim3 gridk(a.heads, nqb), blkk(512), grid(nqb, a.heads), blk(256);
  auto go = [&](auto paged) {        // (paged KV: the same forms through the sequence's block table)
    constexpr bool P = decltype(paged)::value;
    TGX_DT16_SWITCH(c->dt,
      if (ksplit) {
        if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64, 2, 2, P>), gridk, blkk, 2 * lds1, c->stream, a);
        else hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 128, 1, 2, P>), gridk, blkk, 2 * lds1, c->stream, a);
      } else if (lean) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64, 1, 1, P>), grid, blk, lds1, c->stream, a);
      else if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64, 2, 1, P>), grid, blk, lds1, c->stream, a);
      else hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 128, 1, 1, P>), grid, blk, lds1, c->stream, a))     // head_dim 128: two waves per SIMD only in this form (95 vs 138 µs per layer at S = 2048)
  };
  if (a.blk_tbl) go(std::true_type{}); else go(std::false_type{});
}
// RoPE + cache append + q split of S prompt rows of one batch row
void launch_rope_kv_split(tgx_ctx* c, const tgx::RopeKvArgs& a, int S) {
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rope_kv_split_kernel<DT>, dim3(S), dim3(256), 0, c->stream, a))
}
// RMSNorm of the rows of x into 16-bit terms (ws_ah / ws_al), first adding a pending split-K residual (nsplit > 1: the slabs in ws_part)
void launch_norm_terms(tgx_ctx* c, float* x, const ebyte* norm_w, int M, int H, int nsplit, bool third) {
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, x, reinterpret_cast<const bf16_t*>(norm_w), c->d.norm_eps, H,
                                          c->ws_ah, c->ws_al, third ? c->ws_al2 : (bf16_t*)nullptr, (const float*)(nsplit > 1 ? c->ws_part : nullptr), nsplit, (long long)M * H, (const bf16_t*)nullptr))
}
// split-K slabs of a gate_up product -> siluMul -> 16-bit terms (ws_hh / ws_hl), z-ordered sums
void launch_silu_slab_reduce(tgx_ctx* c, int M, int I, int nsplit) {
  tgx::GemmArgs g{};
  g.part = c->ws_part; g.nsplit = nsplit; g.M = M; g.N = 2 * I; g.inter = I; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  const dim3 rg((unsigned)(((size_t)M * I + 255) / 256));
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_SILU>), rg, dim3(256), 0, c->stream, g))
}
void launch_embed_rows(tgx_ctx* c, const long long* ids, float* X, int M, int S) {
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_rows_kernel<DT>, dim3(M), dim3(256), 0, c->stream, ids, (const bf16_t*)c->embed, X, c->d.hidden, S, (long long)c->d.max_ctx))
}

// All layers for S prompt positions of one row at once; leaves the last position's hidden state in row.x.
// == CausalLM::forward on [1,S] ids with an empty cache (GPTModel.h:51-56)
// NB batch rows [row0, row0 + NB) are stacked into ONE [NB*S] row block for the row-wise kernels and the GEMMs (the weights stream
// once for all of them); RoPE / cache append and attention run per batch row on its slice and its own cache.
void launch_prefill(tgx_ctx* c, int row0, int NB, int S, int past) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const size_t kv_layer = c->kv_paged ? (size_t)c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK * hd : (size_t)d.kv_heads * d.max_ctx * hd;      // elements (paged KV: a layer's pool)
  const int M = NB * S;
  const size_t wout = (size_t)qd + 2 * kvd;
  // GPT-2 (ModelGPT2.h:23-208): wte + wpe rows, LayerNorm with bias ahead of both products, a bias on every Conv1D, c_fc -> gelu_new;
  // its rotation tables are the identity, so the RoPE / cache-append kernel and the attention are the Llama family's
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_rows_any_kernel<DT>, dim3(M), dim3(256), 0, c->stream, (const long long*)c->rows[(size_t)row0].prompt, (const void*)c->embed, (const void*)(c->gpt2 ? c->wpe : nullptr), c->ws_x, H, S, (long long)d.max_ctx, past))
  int pend = 1;                     // slabs of the previous layer's down product still to be added to ws_x (1: none; -1: one whole-K slab, see launch_gemm)
  const bf16_t* pend_bias = nullptr;
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    // the QKV product feeds a second rounding (the KV cache): bf16 needs three split terms to reproduce the step path's cache
    // entries (two leave 1-8 % of them one ulp off); fp16's two terms already carry 22 bits
    const bool three = c->dt == tgx::DT_BF16;
    if (c->gpt2) { TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::norm_rows_kernel<DT, 1, 1>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w.in_norm, (const void*)w.in_norm_b, d.norm_eps, H, (float*)nullptr, c->ws_ah, c->ws_al, three ? c->ws_al2 : (bf16_t*)nullptr)) }
    else { TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, c->ws_x, (const bf16_t*)w.in_norm, d.norm_eps, H, c->ws_ah, c->ws_al, three ? c->ws_al2 : (bf16_t*)nullptr,
                                                  (const float*)(pend != 1 ? c->ws_part : nullptr), std::abs(pend), (long long)M * H, pend_bias)) }
    pend = 1;
    int qsl = 1;
    c->qkv_epi = QkvEpi{};
    if (NB == 1) {      // one sequence: the QKV product may finish its rows itself (launch_gemm: gemm_dma_qkv8_kernel<.., ROPE>)
      RowState& r0 = c->rows[(size_t)row0];
      c->qkv_epi.q_hi = c->ws_qh; c->qkv_epi.q_lo = c->ws_ql; c->qkv_epi.past = past; c->qkv_epi.tbl = r0.tbl;
      c->qkv_epi.k = reinterpret_cast<bf16_t*>(r0.kcache) + (size_t)l * kv_layer; c->qkv_epi.v = reinterpret_cast<bf16_t*>(r0.vcache) + (size_t)l * kv_layer;
    }
    launch_gemm(c, tgx::GEMM_STORE, w.wqkv, w.bqkv, c->ws_out, M, qd + 2 * kvd, H, qd + 2 * kvd, /*three_terms=*/three, nullptr, nullptr, /*three_from=*/qd, &qsl);   // Q columns: two terms
    c->qkv_epi = QkvEpi{};
    for (int b = 0; b < NB && qsl != 0; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      bf16_t* kc = reinterpret_cast<bf16_t*>(r.kcache);
      bf16_t* vc = reinterpret_cast<bf16_t*>(r.vcache);
      const size_t ro = (size_t)b * S;             // first workspace row of this batch row
      tgx::RopeKvArgs a{};
      a.QKV = c->ws_out + ro * wout; a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      if (qsl > 1) { a.QKV = nullptr; a.part = c->ws_part + ro * wout; a.nsplit = qsl; a.slab = (long long)M * (long long)wout; a.bias = reinterpret_cast<const bf16_t*>(w.bqkv); }
      a.k_cache = kc + (size_t)l * kv_layer; a.v_cache = vc + (size_t)l * kv_layer;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx; a.past = past; a.blk_tbl = r.tbl;
      a.q_norm_w = d.qk_norm ? (const bf16_t*)w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? (const bf16_t*)w.k_norm : nullptr; a.eps = d.norm_eps;
      launch_rope_kv_split(c, a, S);
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      bf16_t* kc = reinterpret_cast<bf16_t*>(r.kcache);
      bf16_t* vc = reinterpret_cast<bf16_t*>(r.vcache);
      const size_t ro = (size_t)b * S;
      tgx::AttnPrefillArgs a{};
      a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd; a.k_cache = kc + (size_t)l * kv_layer; a.v_cache = vc + (size_t)l * kv_layer;
      a.o_hi = c->ws_ah + ro * qd; a.o_lo = c->ws_al + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = past;
      a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = 1; a.blk_tbl = r.tbl;
      launch_attn_prefill(c, a, /*allow_lean=*/true);
    }
    int osl = 1;
    launch_gemm(c, tgx::GEMM_RESIDUAL, w.wo, w.bo, c->ws_x, M, H, qd, H, false, nullptr, nullptr, 0, c->gpt2 ? nullptr : &osl);
    if (c->gpt2) {
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::norm_rows_kernel<DT, 1, 1>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w.post_norm, (const void*)w.post_norm_b, d.norm_eps, H, (float*)nullptr, c->ws_ah, c->ws_al, (bf16_t*)nullptr))
      launch_gemm(c, tgx::GEMM_GELU, w.wgu, w.bfc, nullptr, M, I, H, I);             // c_fc + bias + gelu_new -> ws_hh / ws_hl
    } else {
      const bool fl = gemm_full_lines(c, tgx::GEMM_SILU, M, 2 * I, H);
      bf16_t* const ai = reinterpret_cast<bf16_t*>(c->ws_out);
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, c->ws_x, (const bf16_t*)w.post_norm, d.norm_eps, H, fl ? ai : c->ws_ah, c->ws_al, (bf16_t*)nullptr,
                                                (const float*)(osl != 1 ? c->ws_part : nullptr), std::abs(osl), (long long)M * H, reinterpret_cast<const bf16_t*>(w.bo), fl ? 1 : 0))
      if (fl) launch_gemm(c, tgx::GEMM_SILU, w.wgu, nullptr, nullptr, M, 2 * I, H, 2 * I, false, ai, nullptr, 0, nullptr, true);
      else launch_gemm(c, tgx::GEMM_SILU, w.wgu, nullptr, nullptr, M, 2 * I, H, 2 * I);      // gate_up + siluMul -> ws_hh / ws_hl
    }
    // the down product's slabs wait for the next layer's input norm (the last layer, and GPT-2's LayerNorm path, finish them here)
    const bool can_defer = !c->gpt2 && l + 1 < d.layers;
    launch_gemm(c, tgx::GEMM_RESIDUAL, w.wdown, w.bdown, c->ws_x, M, H, I, H, false, c->ws_hh, c->ws_hl, 0, can_defer ? &pend : nullptr);
    pend_bias = reinterpret_cast<const bf16_t*>(w.bdown);
  }
  for (int b = 0; b < NB; b++)     // the last position of every batch row feeds lm_head
    (void)hipMemcpyAsync(c->rows[(size_t)(row0 + b)].x, c->ws_x + ((size_t)(b + 1) * S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, c->stream);
}

// dynamic LDS sizes of the tiled GEMMs
int prefill_set_attrs(tgx_ctx* c) {
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_STORE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_STORE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_PARTIAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_PARTIAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
#define TGX_DMA_ATTR1(DT_, EPI_, MI_, BK_, NS_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_kernel<DT_, EPI_, MI_, BK_, NS_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>(160 * 1024, tgx::gemm_dma_lds_bytes(MI_, true, BK_, NS_))));
#define TGX_DMA_ATTR(DT_, EPI_, MI_) TGX_DMA_ATTR1(DT_, EPI_, MI_, ((MI_) == 1 ? 64 : 32), 2)
#define TGX_DMA_ATTR_D(DT_) TGX_DMA_ATTR(DT_, tgx::GEMM_SILU, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_GELU, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_RESIDUAL, 1) TGX_DMA_ATTR(DT_, tgx::GEMM_RESIDUAL, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_STORE, 1) TGX_DMA_ATTR(DT_, tgx::GEMM_STORE, 2)
  TGX_DMA_ATTR_D(tgx::DT_BF16) TGX_DMA_ATTR_D(tgx::DT_F16)
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8n_kernel<tgx::DT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 128 + 256) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8n_kernel<tgx::DT_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 128 + 256) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_qkv8_kernel<tgx::DT_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (3 * 128 + 128 + 64) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_qkv8_kernel<tgx::DT_F16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (3 * 128 + 128 + 64) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_qkv8_kernel<tgx::DT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (3 * 128 + 128 + 64) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_qkv8_kernel<tgx::DT_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (3 * 128 + 128 + 64) * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8i_kernel<tgx::DT_BF16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 256 * 128));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8i_kernel<tgx::DT_F16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 256 * 128));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_prefill_kernel<tgx::DT_BF16, 128, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 136 + 64 * 160) * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_prefill_kernel<tgx::DT_BF16, 128, 1, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 136 + 64 * 160) * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_prefill_kernel<tgx::DT_F16, 128, 1, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 136 + 64 * 160) * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_prefill_kernel<tgx::DT_F16, 128, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 136 + 64 * 160) * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  // split-K slabs on the eight-wave kernel
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_PARTIAL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_PARTIAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_PARTIAL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_PARTIAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  // option act.round16: the one-term forms
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_SILU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_SILU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_STORE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_STORE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_SILU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_SILU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_STORE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_STORE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
#define TGX_DMA_ATTR_P(DT_) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 1, 64, 2) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 2, 64, 2) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 2, 32, 2)
  TGX_DMA_ATTR_P(tgx::DT_BF16) TGX_DMA_ATTR_P(tgx::DT_F16)
#undef TGX_DMA_ATTR_P
#undef TGX_DMA_ATTR_D
#undef TGX_DMA_ATTR
#undef TGX_DMA_ATTR1
  return TGX_OK;
}
