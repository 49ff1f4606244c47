// sampler.hip — Sampler::sample on the logits of a step (Sampler.cpp:23-79) + token publish / pastLength / next embedding.
// Greedy: one finalize launch per row (decode.hip).  Otherwise the staged sampler of kernels/sampler.h: ceil(V/1024) workgroups per row
// (rows on blockIdx.y), one launch per digit level of each active filter, the partial-sum stages, then one pick per row.
#include "ctx.h"
#include "kernels/sampler.h"

void launch_sample(tgx_ctx* c, int row0, int R, const tgx_sampler_cfg& cfg, bool advance_pos, bool log_step) {
  if (is_greedy(&cfg)) { launch_finalize_greedy(c, row0, R, advance_pos, log_step); return; }
  const int V = c->d.vocab;
  RowState& r = c->rows[(size_t)row0];
  tgx::SampArgs a{};
  a.logits = r.logits; a.logits_stride = V;
  a.part_val = r.part_val; a.part_stride = c->lm_grid; a.n_part = c->lm_grid;
  a.sc = c->samp_scratch + row0;
  a.probs_out = r.probs; a.probs_stride = V;
  a.V = V; a.idx_bits = 1;
  while ((1 << a.idx_bits) < V) a.idx_bits++;
  a.temperature = cfg.temperature; a.top_k = cfg.top_k; a.top_p = cfg.top_p; a.min_p = cfg.min_p;
  const bool setK = cfg.top_k > 0, setP = cfg.top_p < 1.f, setM = cfg.min_p > 0.f;
  const int nwg = (V + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE;
  const dim3 grid(nwg, R), blk(tgx::SAMP_WG);
  if (setK) for (int l = 0; l < tgx::SAMP_LEVELS; l++) { a.level = l; hipLaunchKernelGGL(tgx::samp_level_kernel<0>, grid, blk, 0, c->stream, a); }
  if (setP) for (int l = 0; l < tgx::SAMP_LEVELS; l++) { a.level = l; hipLaunchKernelGGL(tgx::samp_level_kernel<1>, grid, blk, 0, c->stream, a); }
  // the first stage after a filter's last level derives that filter's threshold from the level-4 histogram; later stages read it
  a.k_from_hist = (setK && !setP) ? 1 : 0;      // with top-p on, its first level already derived the top-k threshold
  a.p_from_hist = setP ? 1 : 0;
  if (setM) { hipLaunchKernelGGL(tgx::samp_sum_kernel<0>, grid, blk, 0, c->stream, a); a.k_from_hist = 0; a.p_from_hist = 0; }
  hipLaunchKernelGGL(tgx::samp_sum_kernel<1>, grid, blk, 0, c->stream, a);
  a.k_from_hist = 0; a.p_from_hist = 0;
  hipLaunchKernelGGL(tgx::samp_sum_kernel<2>, grid, blk, 0, c->stream, a);
  for (int b = row0; b < row0 + R; b++) {
    tgx::SampPickArgs pa{};
    pa.s = a;
    pa.s.logits = c->rows[(size_t)b].logits; pa.s.part_val = c->rows[(size_t)b].part_val; pa.s.sc = c->samp_scratch;   // the pick kernel indexes sc by fin.row
    pa.s.logits_stride = 0; pa.s.part_stride = 0;
    pa.nwg = nwg; pa.seed = c->seed_dev;
    pa.fin = make_finalize_args(c, b, advance_pos, log_step);
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::samp_pick_kernel<DT>, dim3(1), dim3(tgx::SAMP_WG), 0, c->stream, pa))
  }
}

int sampler_alloc(tgx_ctx* c) {
  const tgx_model_desc& d = c->d;
  int rc;
  if ((rc = dev_alloc(c, &c->samp_scratch, (size_t)d.max_batch))) return rc;
  HIP_OK(c, hipMemset(c->samp_scratch, 0, sizeof(tgx::SampScratch) * (size_t)d.max_batch));
  if ((d.vocab + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE > tgx::SAMP_MAX_WG) return set_err(c, TGX_ERR_UNSUPPORTED, "vocabulary %d exceeds the sampler's %d entries", d.vocab, tgx::SAMP_MAX_WG * tgx::SAMP_TILE);
  return TGX_OK;
}
