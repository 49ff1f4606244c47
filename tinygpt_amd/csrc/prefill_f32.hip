// prefill_f32.hip — batched prefill for fp32 storage (kernels/gemm_f32.h): every product on v_mfma_f32_32x32x2_f32 (exact fp32 products), the
// row-wise ops in fp32, flash attention on the same instruction.  All families incl. GPT-2 — BASELINE.json configs[0] is GPT-2 fp32.
#include "ctx.h"
#include "kernels/attn_decode.h"
#include "kernels/gemm_f32.h"
#include "kernels/skinny.h"

static void launch_gemm_f32(tgx_ctx* c, int epi, const ebyte* B, const ebyte* bias, const float* A, float* C, int M, int N, int K, int ldc) {
  tgx::GemmF32Args g{};
  g.A = A; g.B = reinterpret_cast<const float*>(B); g.bias = reinterpret_cast<const float*>(bias); g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.inter = N / 2;
  const bool few = ((N + tgx::FBN - 1) / tgx::FBN) * ((M + 127) / 128) < 2 * c->num_cus;
  const bool small = epi == tgx::F32_SILU ? false : few;
  const int tm = small ? 64 : 128;
  dim3 grid((N + tgx::FBN - 1) / tgx::FBN, (M + tm - 1) / tm), blk(256);
  // few tiles (a short prompt, or N = hidden): split K over blockIdx.z until ~2 workgroups per CU exist; slabs summed in z order
  const int ntiles = (int)(grid.x * grid.y), ksteps = (K + tgx::FBK - 1) / tgx::FBK;
  int nsplit = 1;
  if (c->gemm_splitk && ntiles < 2 * c->num_cus) nsplit = std::max(1, std::min(std::min(16, ksteps / 4), (2 * c->num_cus + ntiles - 1) / ntiles));
  if (nsplit > 1 && (size_t)nsplit * M * N * 4 > c->ws_part_bytes) nsplit = 1;      // sized in tgx_forward (ensure_f32_part)
  if (nsplit > 1) {
    g.k_per = ((ksteps + nsplit - 1) / nsplit) * tgx::FBK;
    nsplit = (K + g.k_per - 1) / g.k_per;
  }
  g.part = c->ws_part; g.nsplit = nsplit;
  grid.z = nsplit > 1 ? nsplit : 1;
  switch (epi) {
    case tgx::F32_SILU: hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_SILU, 2>), grid, blk, 0, c->stream, g); break;
    case tgx::F32_GELU: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_GELU, 1>), grid, blk, 0, c->stream, g);
                        else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_GELU, 2>), grid, blk, 0, c->stream, g); break;
    case tgx::F32_RESIDUAL: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_RESIDUAL, 1>), grid, blk, 0, c->stream, g);
                            else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_RESIDUAL, 2>), grid, blk, 0, c->stream, g); break;
    default: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_STORE, 1>), grid, blk, 0, c->stream, g);
             else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_STORE, 2>), grid, blk, 0, c->stream, g); break;
  }
  if (nsplit > 1) {
    const size_t nout = (size_t)M * (epi == tgx::F32_SILU ? N / 2 : N);
    const dim3 rg((unsigned)((nout + 255) / 256));
    switch (epi) {
      case tgx::F32_SILU: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_SILU>), rg, blk, 0, c->stream, g); break;
      case tgx::F32_GELU: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_GELU>), rg, blk, 0, c->stream, g); break;
      case tgx::F32_RESIDUAL: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_RESIDUAL>), rg, blk, 0, c->stream, g); break;
      default: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_STORE>), rg, blk, 0, c->stream, g); break;
    }
  }
}

// split-K slabs of the fp32 products: up to 16 splits of the widest [rows][N] output
int ensure_f32_part(tgx_ctx* c, int rows) {
  const tgx_model_desc& d = c->d;
  const size_t widest = std::max<size_t>(std::max<size_t>((size_t)d.heads * d.head_dim + 2 * (size_t)d.kv_heads * d.head_dim, (size_t)(c->gpt2 ? 1 : 2) * d.inter), (size_t)d.hidden);
  // splits shrink as the tile count grows: nsplit * tiles stays near two per CU, so nsplit * rows * N is bounded by ~2 CUs x one 128 x 128 tile x 16
  const size_t need = std::min<size_t>((size_t)16 * rows * widest * 4, (size_t)64 << 20);
  if (need > c->ws_part_bytes) {
    drop_step_graphs(c);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (c->ws_part) (void)hipFree(c->ws_part);
    c->ws_part = nullptr; c->ws_part_bytes = 0;
    HIP_OK(c, hipMalloc((void**)&c->ws_part, need));
    c->ws_part_bytes = need;
  }
  return TGX_OK;
}

void launch_prefill_f32(tgx_ctx* c, int row0, int NB, int S) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd, nq = qd + 2 * kvd;
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd * c->esz;
  const int M = NB * S;
  float* xn = reinterpret_cast<float*>(c->ws_ah);      // [M][max(H, qd)]: normalised rows, later the attention output
  float* qrows = reinterpret_cast<float*>(c->ws_qh);   // [M][qd] rotated queries
  float* hrows = reinterpret_cast<float*>(c->ws_hh);   // [M][I]
  hipLaunchKernelGGL((tgx::embed_rows_any_kernel<tgx::DT_F32>), dim3(M), dim3(256), 0, c->stream, (const long long*)c->rows[(size_t)row0].prompt, (const void*)c->embed, (const void*)(c->gpt2 ? c->wpe : nullptr), c->ws_x, H, S, (long long)d.max_ctx, (int)c->past);
  hipLaunchKernelGGL(tgx::iota_pos_kernel, dim3((S + 255) / 256), dim3(256), 0, c->stream, c->ws_pos, (int)c->past, S);
  auto norm = [&](const ebyte* w, const ebyte* b) {
    if (c->gpt2) hipLaunchKernelGGL((tgx::norm_rows_kernel<tgx::DT_F32, 1, 0>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w, (const void*)b, d.norm_eps, H, xn, (bf16_t*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr);
    else hipLaunchKernelGGL((tgx::norm_rows_kernel<tgx::DT_F32, 0, 0>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w, (const void*)nullptr, d.norm_eps, H, xn, (bf16_t*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr);
  };
  c->attn_direct = c->past + S <= c->attn_direct_max;
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    norm(w.in_norm, w.in_norm_b);
    launch_gemm_f32(c, tgx::F32_STORE, w.wqkv, w.bqkv, xn, c->ws_out, M, nq, H, nq);
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::RopeRowsArgs a{};
      a.QKV = c->ws_out + ro * nq; a.rows = S; a.q_out = qrows + ro * qd; a.q_stride = qd;
      a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer; a.kv_stride = 0;     // the rows are positions of ONE sequence
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = c->ws_pos;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx;
      a.q_norm_w = d.qk_norm ? w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? w.k_norm : nullptr; a.eps = d.norm_eps;
      hipLaunchKernelGGL((tgx::rope_kv_rows_kernel<tgx::DT_F32>), dim3(S, d.heads + 2 * d.kv_heads), dim3(64), 0, c->stream, a);
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      {                          // causal flash attention on the f32-input MFMA (K / V tiles shared by 128 queries)
        tgx::AttnPrefillF32Args a{};
        a.q = qrows + ro * qd; a.k_cache = reinterpret_cast<const float*>(r.kcache + (size_t)l * kv_layer); a.v_cache = reinterpret_cast<const float*>(r.vcache + (size_t)l * kv_layer);
        a.out = xn + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = (int)c->past;
        a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = 1;
        const dim3 grid((S + 127) / 128, d.heads), blk(256);
        if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_f32_kernel<64>), grid, blk, 0, c->stream, a);
        else hipLaunchKernelGGL((tgx::attn_prefill_f32_kernel<128>), grid, blk, 0, c->stream, a);
      }
    }
    launch_gemm_f32(c, tgx::F32_RESIDUAL, w.wo, w.bo, xn, c->ws_x, M, H, qd, H);
    norm(w.post_norm, w.post_norm_b);
    if (c->gpt2) launch_gemm_f32(c, tgx::F32_GELU, w.wgu, w.bfc, xn, hrows, M, I, H, I);
    else launch_gemm_f32(c, tgx::F32_SILU, w.wgu, nullptr, xn, hrows, M, 2 * I, H, I);
    launch_gemm_f32(c, tgx::F32_RESIDUAL, w.wdown, w.bdown, hrows, c->ws_x, M, H, I, H);
  }
  for (int b = 0; b < NB; b++)
    (void)hipMemcpyAsync(c->rows[(size_t)(row0 + b)].x, c->ws_x + ((size_t)(b + 1) * S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, c->stream);
}
