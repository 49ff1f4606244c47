// sampler.hip — Sampler::sample on the logits of a step (Sampler.cpp:23-79) + token publish / pastLength / next embedding.
// Greedy: one finalize launch per row (decode.hip).  Otherwise the staged sampler of kernels/sampler.h: ceil(V/1024) workgroups per row
// (rows on blockIdx.y) for the first digit of each active filter and for the compaction of its threshold bin, one workgroup per row for the
// filter's tail — which also draws the token when the chain ends there.
#include "ctx.h"
#include "kernels/sampler.h"
#include <type_traits>

static tgx::SampArgs samp_args(tgx_ctx* c, int row0, const tgx_sampler_cfg& cfg) {
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
  a.list_comp = c->samp_list_comp + (size_t)row0 * V; a.list_v = c->samp_list_v + (size_t)row0 * V;
  // the normaliser of the final kept set: left by the last filter's tail, or — with min-p, or without any filter — the ordered sum of stage 1's tile sums
  a.z_from_tail = ((cfg.top_k > 0 || cfg.top_p < 1.f) && !(cfg.min_p > 0.f)) ? 1 : 0;
  return a;
}

void launch_sample(tgx_ctx* c, int row0, int R, const tgx_sampler_cfg& cfg, bool advance_pos, bool log_step) {
  if (is_greedy(&cfg)) { launch_finalize_greedy(c, row0, R, advance_pos, log_step); return; }
  tgx::SampArgs a = samp_args(c, row0, cfg);
  const bool setK = cfg.top_k > 0, setP = cfg.top_p < 1.f, setM = cfg.min_p > 0.f;
  const int nwg = (a.V + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE;
  const dim3 grid(nwg, R), blk(tgx::SAMP_WG);
  // the draw's arguments — the last launch of the chain, one workgroup per row (blockIdx.y; the row that completes the batch's count moves the step counter)
  auto pick_args = [&](const tgx::SampArgs& now) {
    tgx::SampPickArgs pa{};
    pa.s = now;
    pa.nwg = nwg; pa.seed = c->seed_dev;
    pa.fin = make_finalize_args(c, row0, advance_pos, log_step);
    pa.x_stride = c->d.hidden;
    return pa;
  };
  // a filter = first digit over the vocabulary, compaction of the threshold's bin, the tail (four digits, threshold) in one workgroup; the tail of the
  // chain's LAST filter also draws when no min-p follows
  a.mx_ready = 0;                 // the first launch that needs max(logits / T) reduces the lm_head partials and leaves it in sc->mx for the others
  auto tail = [&](auto mode, bool draws) {
    constexpr int MODE = decltype(mode)::value;
    const tgx::SampPickArgs pa = pick_args(a);
    if (!draws) { hipLaunchKernelGGL((tgx::samp_tail_kernel<MODE, false, 0>), dim3(1, R), blk, 0, c->stream, pa); return; }
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL((tgx::samp_tail_kernel<MODE, true, DT>), dim3(1, R), blk, 0, c->stream, pa))
  };
  if (setK) {
    hipLaunchKernelGGL(tgx::samp_level0_kernel<0>, grid, blk, 0, c->stream, a);       // (counts: no maximum needed)
    hipLaunchKernelGGL(tgx::samp_compact_kernel<0>, grid, blk, 0, c->stream, a);
    a.mx_ready = 1;
    tail(std::integral_constant<int, 0>{}, !setP && !setM);
  }
  if (setP) {
    hipLaunchKernelGGL(tgx::samp_level0_kernel<1>, grid, blk, 0, c->stream, a);
    a.mx_ready = 1;
    hipLaunchKernelGGL(tgx::samp_compact_kernel<1>, grid, blk, 0, c->stream, a);
    tail(std::integral_constant<int, 1>{}, !setM);
  }
  if (a.z_from_tail) return;      // the last tail drew
  // with min-p (its cut depends on the normaliser of the set it looks at) or without any filter: partial-sum stages over the vocabulary, then the pick
  if (setM) { hipLaunchKernelGGL(tgx::samp_sum_kernel<0>, grid, blk, 0, c->stream, a); a.mx_ready = 1; }
  hipLaunchKernelGGL(tgx::samp_sum_kernel<1>, grid, blk, 0, c->stream, a);
  a.mx_ready = 1;
  {
    const tgx::SampPickArgs pa = pick_args(a);
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::samp_pick_kernel<DT>, dim3(1, R), blk, 0, c->stream, pa))
  }
}

// tgx_read_probs: the final probability vector of row `row`'s last sampled step, evaluated from what that step left on the device (its logits, the
// filters' thresholds, the normalisers) — the step itself never needs the vector
void launch_probs(tgx_ctx* c, int row, const tgx_sampler_cfg& cfg) {
  tgx::SampArgs a = samp_args(c, row, cfg);
  a.mx_ready = 0;
  const int nwg = (a.V + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE;
  hipLaunchKernelGGL(tgx::samp_sum_kernel<2>, dim3(nwg, 1), dim3(tgx::SAMP_WG), 0, c->stream, a);
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

#ifdef TGX_SAMP_TIMELINE
// experiment builds only (tools/sampler_timeline.py): row 0's stamps, [64][10] ticks of the 100 MHz wall clock + the number of steps stamped
extern "C" __attribute__((visibility("default"))) int tgx_debug_samp_timeline(tgx_ctx* c, unsigned long long* out, unsigned int* n) {
  if (hipStreamSynchronize(c->stream) != hipSuccess) return 1;
  tgx::SampScratch h;
  if (hipMemcpy(&h, c->samp_scratch, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  memcpy(out, h.tl, sizeof(h.tl)); *n = h.tl_n;
  return 0;
}
#endif
