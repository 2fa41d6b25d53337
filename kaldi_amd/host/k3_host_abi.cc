// k3_host_abi.cc -- the C ABI of include/k3host.h over k3_lattice.cc / k3_host.cc (-> kaldi_amd/lib/libk3host.so).  No GPU code.
#include <sstream>
#include <cstring>
#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <memory>
#include <thread>
#include "../../include/k3host.h"
#include "k3_host.h"
using namespace k3host;

struct k3h_transitions { TransitionInfo info; };
struct k3h_clat { CompactLattice c; };
struct k3h_ivector_config { IvectorExtractionInfo info; };
namespace {
thread_local std::string g_err;
template <class F> int Guard(F f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown error";
    return -2;
  }
}
Lattice FromArrays(int32_t ns, int32_t start, const float *fin, int64_t na, const int32_t *src, const int32_t *dst, const int32_t *il, const int32_t *ol,
    const float *g, const float *ac) {
  if (ns < 0 || na < 0 || (ns > 0 && (start < 0 || start >= ns))) K3H_ERR << "bad lattice: " << ns << " states, start " << start << ", " << na << " arcs";
  Lattice l; l.start = ns ? start : -1; l.st_frame.assign(ns, 0); l.st_state.assign(ns, 0); l.st_final.assign(fin, fin + ns);
  for (float &f : l.st_final) if (!std::isfinite(f)) f = std::numeric_limits<float>::infinity();
  l.arc_src.assign(src, src + na);
  l.arc_dst.assign(dst, dst + na);
  l.arc_ilabel.assign(il, il + na);
  l.arc_olabel.assign(ol, ol + na);
  l.arc_graph.assign(g, g + na);
  l.arc_ac.assign(ac, ac + na);
  for (int64_t a = 0; a < na; a++) if (src[a] < 0 || src[a] >= ns || dst[a] < 0 || dst[a] >= ns) K3H_ERR << "bad lattice: arc " << a << " goes " << src[a] <<
      " -> " << dst[a];
  return l;
}
}  // namespace

extern "C" {
const char *k3h_last_error(void) { return g_err.c_str(); }
void k3h_det_opts_default(k3h_det_opts *o) {
  const DeterminizeLatticePhonePrunedOptions d;
  o->delta = d.delta;
  o->max_mem = d.max_mem;
  o->phone_determinize = d.phone_determinize;
  o->word_determinize = d.word_determinize;
  o->minimize = d.minimize;
}

int k3h_transitions_read(const char *rx, k3h_transitions **out) {
  return Guard([&] { auto *t = new k3h_transitions; try { t->info = ReadTransitionModel(rx); } catch (...) { delete t; throw; } *out = t; });
}
int32_t k3h_transitions_num_ids(const k3h_transitions *t) { return (int32_t)t->info.id2pdf.size() - 1; }
void k3h_transitions_free(k3h_transitions *t) { delete t; }

int k3h_determinize_lattice(const k3h_transitions *trans, int32_t ns, int32_t start, const float *fin, int64_t na, const int32_t *src, const int32_t *dst,
    const int32_t *il,
                            const int32_t *ol, const float *g, const float *ac, double beam, const k3h_det_opts *o, k3h_clat **out, int32_t *complete) {
  return Guard([&] {
    const Lattice lat = FromArrays(ns, start, fin, na, src, dst, il, ol, g, ac);
    DeterminizeLatticePhonePrunedOptions po;
    if (o) {
      po.delta = o->delta;
      po.max_mem = o->max_mem;
      po.phone_determinize = o->phone_determinize != 0;
      po.word_determinize = o->word_determinize != 0;
      po.minimize = o->minimize != 0;
    }
    auto *c = new k3h_clat; bool ok;
    try {
      if (trans) ok = DeterminizeLatticePhonePruned(lat, trans->info, beam, &c->c, po);
      else {
        DeterminizeLatticePrunedOptions d; d.delta = po.delta; d.max_mem = po.max_mem;
        ok = DeterminizeLatticePruned(lat, beam, &c->c, d);
        if (po.minimize) { PushCompactLatticeStrings(&c->c); PushCompactLatticeWeights(&c->c); MinimizeCompactLattice(&c->c); }
      }
    } catch (...) { delete c; throw; }
    *out = c; if (complete) *complete = ok ? 1 : 0;
  });
}
int k3h_convert_lattice(int32_t ns, int32_t start, const float *fin, int64_t na, const int32_t *src, const int32_t *dst, const int32_t *il, const int32_t *ol,
    const float *g, const float *ac, k3h_clat **out) {
  return Guard([&] { const Lattice lat = FromArrays(ns, start, fin, na, src, dst, il, ol, g, ac); auto *c =
      new k3h_clat; try { ConvertLattice(lat, &c->c); } catch (...) { delete c; throw; } *out = c; });
}
// LatticePostprocessor::GetCTM on every lattice of a table: lattice -> CompactLattice (ConvertLattice) -> scales / word insertion penalty of the config file ->
// MBR -> CTM lines
// (MergeSegmentsToCTMOutput's layout).  Returns the number of bytes written to `out` (0-terminated), or -1 (k3h_last_error).
int64_t k3h_lattice_table_to_ctm(const char *lattice_rspecifier, const char *postprocessor_config_rxfilename, float decoder_frame_shift_seconds, char *out,
    int64_t out_cap) {
  int64_t n = -1;
  const int rc = Guard([&] {
    auto pp = LoadLatticePostprocessor(postprocessor_config_rxfilename); pp->SetDecoderFrameShift(decoder_frame_shift_seconds);
    std::ostringstream os;
    for (auto &kv : ReadLatticeTable(lattice_rspecifier)) {
      Connect(&kv.second);
      CompactLattice clat;
      if (kv.second.NumStates() > 0) ConvertLattice(kv.second, &clat);
      CtmResult ctm;
      pp->GetCTM(clat, &ctm);
      WriteCtm(ctm, kv.first, os);
    }
    const std::string s = os.str();
    if ((int64_t)s.size() + 1 > out_cap) K3H_ERR << "k3h_lattice_table_to_ctm: output buffer too small (" << s.size() + 1 << " bytes needed)";
    memcpy(out, s.c_str(), s.size() + 1); n = (int64_t)s.size();
  });
  return rc == 0 ? n : -1;
}
// ... with the model (final.mdl) the post-processor needs when its config names a --word-boundary-rxfilename: the lattice is word-aligned (WordAlignLattice)
// before MBR, so that the
// CTM times are word boundaries (cudadecoder/lattice-postprocessor.cc:66-85)
int64_t k3h_lattice_table_to_ctm_model(const char *lattice_rspecifier, const char *postprocessor_config_rxfilename, float decoder_frame_shift_seconds,
    const char *model_rxfilename, char *out, int64_t out_cap) {
  int64_t n = -1;
  const int rc = Guard([&] {
    auto pp = LoadLatticePostprocessor(postprocessor_config_rxfilename); pp->SetDecoderFrameShift(decoder_frame_shift_seconds);
    TransitionInfo ti; if (model_rxfilename && *model_rxfilename) { ti = ReadTransitionModel(model_rxfilename); pp->SetTransitionInformation(&ti); }
    std::ostringstream os;
    for (auto &kv : ReadLatticeTable(lattice_rspecifier)) {
      Connect(&kv.second);
      CompactLattice clat;
      if (kv.second.NumStates() > 0) ConvertLattice(kv.second, &clat);
      CtmResult ctm;
      pp->GetCTM(clat, &ctm);
      WriteCtm(ctm, kv.first, os);
    }
    const std::string s = os.str();
    if ((int64_t)s.size() + 1 > out_cap) K3H_ERR << "k3h_lattice_table_to_ctm_model: output buffer too small (" << s.size() + 1 << " bytes needed)";
    memcpy(out, s.c_str(), s.size() + 1); n = (int64_t)s.size();
  });
  return rc == 0 ? n : -1;
}
int k3h_clat_sizes(const k3h_clat *c, int32_t *ns, int64_t *na, int64_t *nl) {
  return Guard([&] { int64_t n = 0; for (const auto &s : c->c.fin_str) n += (int64_t)s.size(); for (const auto &s : c->c.arc_str) n += (int64_t)s.size(); *ns =
      c->c.NumStates(); *na = (int64_t)c->c.arc_src.size(); *nl = n; });
}
int k3h_clat_get(const k3h_clat *h, int32_t *start, uint8_t *is_final, float *fg, float *fa, int64_t *foff, int32_t *src, int32_t *dst, int32_t *label,
    float *g, float *a, int64_t *aoff, int32_t *strings) {
  return Guard([&] {
    const CompactLattice &c = h->c; const int32_t ns = c.NumStates(); const size_t na = c.arc_src.size(); int64_t p = 0;
    *start = c.start;
    for (int32_t s = 0; s < ns; s++) {
      is_final[s] = (uint8_t)c.is_final[s];
      fg[s] = c.fin_graph[s];
      fa[s] = c.fin_ac[s];
      foff[s] = p;
      for (int32_t t : c.fin_str[s]) strings[p++] = t;
    }
    foff[ns] = p;
    for (size_t k = 0; k < na; k++) {
      src[k] = c.arc_src[k];
      dst[k] = c.arc_dst[k];
      label[k] = c.arc_label[k];
      g[k] = c.arc_graph[k];
      a[k] = c.arc_ac[k];
      aoff[k] = p;
      for (int32_t t : c.arc_str[k]) strings[p++] = t;
    }
    aoff[na] = p;
  });
}
// The host tail for a whole batch as k3_decoder_get_raw_lattices hands it over (concatenated arrays + per-utterance offsets): per utterance
// Connect (decoder-wrappers.cc:353) + DeterminizeLatticePhonePrunedWrapper / DeterminizeLatticePruned, on `num_threads` worker threads
// (cf. the CPU worker pool of the reference pipeline, batched-threaded-nnet3-cuda-pipeline2.h:170-177).  h_out (nullable): one handle per
// utterance (NULL for an utterance without a surviving path), to be freed by the caller; without it the lattices are dropped after counting.
int k3h_postprocess_batch(const k3h_transitions *trans, int32_t num_utts, const int64_t *state_offsets, const int64_t *arc_offsets, int32_t graph_start,
                          const int32_t *st_frame, const int32_t *st_state, const float *st_final, const int32_t *arc_src, const int32_t *arc_dst,
                              const int32_t *arc_ilabel,
                          const int32_t *arc_olabel, const float *arc_graph, const float *arc_ac, double beam, const k3h_det_opts *o, int32_t num_threads,
                          k3h_clat **h_out, int32_t *h_clat_states, int64_t *h_clat_arcs, int32_t *h_complete) {
  return Guard([&] {
    if (num_utts < 0 || !state_offsets || !arc_offsets) K3H_ERR << "k3h_postprocess_batch: bad argument";
    DeterminizeLatticePhonePrunedOptions po;
    if (o) {
      po.delta = o->delta;
      po.max_mem = o->max_mem;
      po.phone_determinize = o->phone_determinize != 0;
      po.word_determinize = o->word_determinize != 0;
      po.minimize = o->minimize != 0;
    }
    std::atomic<int32_t> next(0); std::atomic<int> failed(0); std::string first_err; std::mutex mu;
    auto work = [&]() {
      for (;;) {
        const int32_t u = next.fetch_add(1); if (u >= num_utts) return;
        try {
          const int64_t s0 = state_offsets[u], ns = state_offsets[u + 1] - s0, a0 = arc_offsets[u], na = arc_offsets[u + 1] - a0;
          if (h_out) h_out[u] = nullptr;
          if (h_clat_states) h_clat_states[u] = 0;
          if (h_clat_arcs) h_clat_arcs[u] = 0;
          if (h_complete) h_complete[u] = 0;
          if (ns == 0) continue;
          Lattice lat;
          lat.st_frame.assign(st_frame + s0, st_frame + s0 + ns);
          lat.st_state.assign(st_state + s0, st_state + s0 + ns);
          lat.st_final.assign(st_final + s0, st_final + s0 + ns);
          lat.arc_src.assign(arc_src + a0, arc_src + a0 + na);
          lat.arc_dst.assign(arc_dst + a0, arc_dst + a0 + na);
          lat.arc_ilabel.assign(arc_ilabel + a0, arc_ilabel + a0 + na);
          lat.arc_olabel.assign(arc_olabel + a0, arc_olabel + a0 + na);
          lat.arc_graph.assign(arc_graph + a0, arc_graph + a0 + na);
          lat.arc_ac.assign(arc_ac + a0, arc_ac + a0 + na);
          lat.start = -1; for (int64_t s = 0; s < ns; s++) if (lat.st_frame[s] == 0 && lat.st_state[s] == graph_start) lat.start = (int32_t)s;
          if (lat.start < 0) continue;
          Connect(&lat);
          if (lat.NumStates() == 0) continue;
          std::unique_ptr<k3h_clat> c(new k3h_clat); bool ok;
          if (trans) ok = DeterminizeLatticePhonePruned(lat, trans->info, beam, &c->c, po);
          else { DeterminizeLatticePrunedOptions d; d.delta = po.delta; d.max_mem = po.max_mem; ok = DeterminizeLatticePruned(lat, beam, &c->c, d); }
          if (h_clat_states) h_clat_states[u] = c->c.NumStates();
          if (h_clat_arcs) h_clat_arcs[u] = (int64_t)c->c.arc_src.size();
          if (h_complete) h_complete[u] = ok ? 1 : 0;
          if (h_out) h_out[u] = c.release();
        } catch (const std::exception &e) { std::lock_guard<std::mutex> g(mu); if (!failed.fetch_add(1)) first_err = e.what(); }
      }
    };
    std::vector<std::thread> th; const int nt = std::max(1, std::min<int>(num_threads, num_utts));
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed.load()) K3H_ERR << "k3h_postprocess_batch: " << failed.load() << " utterances failed, first: " << first_err;
  });
}
int k3h_clat_scale_acoustic(k3h_clat *c, double scale) { return Guard([&] { ScaleAcoustic(&c->c, scale); }); }
int k3h_clat_write(const k3h_clat *c, const char *key, const char *wspecifier) {
  return Guard([&] { TableWriter w(wspecifier); w.WriteCompactLattice(key, c->c); w.Flush(); });
}
void k3h_clat_free(k3h_clat *c) { delete c; }
int k3h_ivector_config_read(const char *rx, k3h_ivector_config **out) {
  return Guard([&] { auto *c = new k3h_ivector_config; try { c->info = ReadIvectorExtractionConfig(rx); } catch (...) { delete c; throw; } *out = c; });
}
int k3h_ivector_config_get(const k3h_ivector_config *c, int32_t *ints, double *reals, const float **lda, const double **gstats, const double **gconsts,
    const double **miv, const double **iv,
                           const double **M, const double **sigma_inv) {
  return Guard([&] {
    const IvectorExtractionInfo &i = c->info;
    const int32_t v[16] = {
      i.global_cmvn_stats.cols - 1, i.lda_rows, i.lda_cols, i.ubm.num_gauss, i.ie.ivector_dim, i.left_context, i.right_context, i.ivector_period,
          i.num_gselect, i.num_cg_iters,
                           i.cmn_window, i.speaker_frames, i.global_frames, i.normalize_mean, i.normalize_variance, i.online_cmvn_iextractor};
    if (ints) memcpy(ints, v, sizeof v);
    if (reals) {
      reals[0] = i.min_post;
      reals[1] = i.posterior_scale;
      reals[2] = i.max_count;
      reals[3] = i.ie.prior_offset;
      reals[4] = i.max_remembered_frames;
    }
    if (lda) *lda = i.lda.data();
    if (gstats) *gstats = i.global_cmvn_stats.data.data();
    if (gconsts) *gconsts = i.ubm.gconsts.data();
    if (miv) *miv = i.ubm.means_invvars.data();
    if (iv) *iv = i.ubm.inv_vars.data();
    if (M) *M = i.ie.M.data();
    if (sigma_inv) *sigma_inv = i.ie.sigma_inv.data();
  });
}
void k3h_ivector_config_free(k3h_ivector_config *c) { delete c; }
}
