// sampler.hip — Sampler::sample on the logits of a step (Sampler.cpp:23-79) + token publish / pastLength / next embedding.
// Greedy: one finalize launch per row (decode.hip).  Otherwise the staged sampler of kernels/sampler.h: ceil(V/1024) workgroups per row
// (rows on blockIdx.y) for the first digit of each active filter and for the compaction of its threshold bin, one workgroup per row for the
// filter's tail, the probability stage, then one pick per row.
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
  a.list_comp = c->samp_list_comp + (size_t)row0 * V; a.list_v = c->samp_list_v + (size_t)row0 * V;
  const dim3 tail(1, R);
  // a filter = first digit over the vocabulary, compaction of the threshold's bin, the tail (four digits, threshold[, normaliser]) in one workgroup
  a.mx_ready = 0;                 // the first launch that needs max(logits / T) reduces the lm_head partials and leaves it in sc->mx for the others
  if (setK) {
    hipLaunchKernelGGL(tgx::samp_level0_kernel<0>, grid, blk, 0, c->stream, a);       // (counts: no maximum needed)
    hipLaunchKernelGGL(tgx::samp_compact_kernel<0>, grid, blk, 0, c->stream, a);
    a.mx_ready = 1;
    hipLaunchKernelGGL(tgx::samp_tail_kernel<0>, tail, blk, 0, c->stream, a, nwg, (!setP && !setM) ? 1 : 0);
  }
  if (setP) {
    hipLaunchKernelGGL(tgx::samp_level0_kernel<1>, grid, blk, 0, c->stream, a);
    a.mx_ready = 1;
    hipLaunchKernelGGL(tgx::samp_compact_kernel<1>, grid, blk, 0, c->stream, a);
    hipLaunchKernelGGL(tgx::samp_tail_kernel<1>, tail, blk, 0, c->stream, a, nwg, !setM ? 1 : 0);
  }
  // the normaliser: from the last filter's tail; with min-p (its cut depends on the normaliser of the set it looks at) or without any filter, two
  // partial-sum stages over the vocabulary
  a.z_from_tail = ((setK || setP) && !setM) ? 1 : 0;
  if (setM) { hipLaunchKernelGGL(tgx::samp_sum_kernel<0>, grid, blk, 0, c->stream, a); a.mx_ready = 1; }
  if (!a.z_from_tail) { hipLaunchKernelGGL(tgx::samp_sum_kernel<1>, grid, blk, 0, c->stream, a); a.mx_ready = 1; }
  hipLaunchKernelGGL(tgx::samp_sum_kernel<2>, grid, blk, 0, c->stream, a);
  for (int b = row0; b < row0 + R; b++) {
    tgx::SampPickArgs pa{};
    pa.s = a;
    pa.s.logits = c->rows[(size_t)b].logits; pa.s.part_val = c->rows[(size_t)b].part_val; pa.s.sc = c->samp_scratch;   // the pick kernel indexes sc by fin.row
    pa.s.logits_stride = 0; pa.s.part_stride = 0; pa.s.list_comp = nullptr; pa.s.list_v = nullptr;
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
  // the compacted list of a filter's threshold bin: a few thousand entries on real logits, the whole vocabulary when every logit is equal
  if ((rc = dev_alloc(c, &c->samp_list_comp, (size_t)d.max_batch * d.vocab)) || (rc = dev_alloc(c, &c->samp_list_v, (size_t)d.max_batch * d.vocab))) return rc;
  if ((d.vocab + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE > tgx::SAMP_MAX_WG) return set_err(c, TGX_ERR_UNSUPPORTED, "vocabulary %d exceeds the sampler's %d entries", d.vocab, tgx::SAMP_MAX_WG * tgx::SAMP_TILE);
  return TGX_OK;
}
