// k3_mbr.cc -- word-level Minimum Bayes Risk decoding of a CompactLattice and the lattice post-processor of the reference's CUDA pipeline, restated
// over k3host::CompactLattice (host side of the path: SURVEY 8f row 1 / row P).
//   MinimumBayesRisk      lat/sausages.{h,cc} (Xu, Povey, Mangu, Zhu: "Minimum Bayes Risk decoding and system combination based on a recursion for edit
//                         distance"): Figures 4-6 of the paper as the reference codes them -- EditDistance :130-166, AccStats :169-318, MbrDecode :28-108,
//                         PrepareLatticeAndInitStats :320-363 (CreateSuperFinal fstext/pre-determinize-inl.h:689-724, CompactLatticeStateTimes
//                         lat/lattice-functions.cc:109-147), first hypothesis = the lattice's best path (:377-395)
//   LatticePostprocessor  cudadecoder/lattice-postprocessor.{h,cc}: ScaleLattice (fstext/lattice-utils-inl.h:197-219), AddWordInsPenToCompactLattice
//                         (lat/lattice-functions.cc:1342-1363), MBR -> CTMResult {words, (begin, end) in seconds, confidences}
//   WriteCtm              MergeSegmentsToCTMOutput of one un-segmented utterance (cudadecoder/cuda-pipeline-common.cc:67-142)
// Pinned to the reference's own lat/sausages.cc compiled unmodified (oracle/_ref/bin/ref-mbr, tests/test_lattice_det.py).
#include "k3_host.h"
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <limits>
#include <map>
#include <sstream>

namespace k3host {
namespace {
const double kLogZero = -std::numeric_limits<double>::infinity();
inline double LogAdd(double x, double y) {      // base/kaldi-math.h:177-193
  double diff;
  if (x < y) { diff = x - y; x = y; } else diff = y - x;
  if (diff >= -36.04365338911715 /* kMinLogDiffDouble = log(DBL_EPSILON) */) return x + std::log1p(std::exp(diff));
  return x;
}
inline void AddToMap(int32_t i, double d, std::map<int32_t, double> *m) { if (d == 0) return; auto r = m->insert({i, d}); if (!r.second) r.first->second += d; }
struct Dense { int32_t cols = 0; std::vector<double> v; Dense(int32_t r, int32_t c) : cols(c), v((size_t)r * c, 0.0) {} double &operator()(int32_t r, int32_t c) { return v[(size_t)r * cols + c]; } };
}  // namespace

struct MinimumBayesRisk::Impl {
  struct Arc { int32_t word, start_node, end_node; float loglike; };
  MinimumBayesRiskOptions opts; std::vector<Arc> arcs; std::vector<std::vector<int32_t>> pre; std::vector<int32_t> state_times; std::vector<int32_t> R; double L = 0.0;
  std::vector<std::vector<std::pair<int32_t, float>>> gamma; std::vector<std::vector<std::pair<float, float>>> times; std::vector<std::pair<float, float>> sausage_times, one_best_times;
  std::vector<float> one_best_conf;
  static double delta() { return 1.0e-05f; }      // (BaseFloat in the reference)
  static double l(int32_t a, int32_t b, bool penalize = false) { return a == b ? 0.0 : (penalize ? 1.0 + delta() : 1.0); }
  int32_t r(int32_t q) const { return R[q - 1]; }
  static void RemoveEps(std::vector<int32_t> *v) { v->erase(std::remove(v->begin(), v->end(), 0), v->end()); }
  static void NormalizeEps(std::vector<int32_t> *v) {
    RemoveEps(v); v->resize(1 + v->size() * 2); const int32_t s = (int32_t)v->size();
    for (int32_t i = s / 2 - 1; i >= 0; i--) { (*v)[i * 2 + 1] = (*v)[i]; (*v)[i * 2 + 2] = 0; }
    (*v)[0] = 0;
  }
  double EditDistance(int32_t N, int32_t Q, std::vector<double> &alpha, Dense &alpha_dash, std::vector<double> &alpha_dash_arc) {
    alpha[1] = 0.0; alpha_dash(1, 0) = 0.0;
    for (int32_t q = 1; q <= Q; q++) alpha_dash(1, q) = alpha_dash(1, q - 1) + l(0, r(q));
    for (int32_t n = 2; n <= N; n++) {
      double alpha_n = kLogZero;
      for (int32_t ai : pre[n]) { const Arc &arc = arcs[ai]; alpha_n = LogAdd(alpha_n, alpha[arc.start_node] + arc.loglike); }
      alpha[n] = alpha_n;
      for (int32_t ai : pre[n]) {
        const Arc &arc = arcs[ai]; const int32_t s_a = arc.start_node, w_a = arc.word; const float p_a = arc.loglike;
        for (int32_t q = 0; q <= Q; q++) {
          if (q == 0) alpha_dash_arc[q] = alpha_dash(s_a, q) + l(w_a, 0, true);
          else {
            const int32_t r_q = r(q);
            const double a1 = alpha_dash(s_a, q - 1) + l(w_a, r_q), a2 = alpha_dash(s_a, q) + l(w_a, 0, true), a3 = alpha_dash_arc[q - 1] + l(0, r_q);
            alpha_dash_arc[q] = std::min(a1, std::min(a2, a3));
          }
          alpha_dash(n, q) += std::exp(alpha[s_a] + p_a - alpha[n]) * alpha_dash_arc[q];
        }
      }
    }
    return alpha_dash(N, Q);
  }
  void AccStats() {
    const int32_t N = (int32_t)pre.size() - 1, Q = (int32_t)R.size();
    std::vector<double> alpha(N + 1, 0.0), alpha_dash_arc(Q + 1, 0.0), beta_dash_arc(Q + 1, 0.0); Dense alpha_dash(N + 1, Q + 1), beta_dash(N + 1, Q + 1);
    std::vector<char> b_arc(Q + 1, 0); std::vector<std::map<int32_t, double>> gam(Q + 1), tau_b(Q + 1), tau_e(Q + 1);
    L = EditDistance(N, Q, alpha, alpha_dash, alpha_dash_arc);
    beta_dash(N, Q) = 1.0;
    for (int32_t n = N; n >= 2; n--) {
      for (int32_t ai : pre[n]) {
        const Arc &arc = arcs[ai]; const int32_t s_a = arc.start_node, w_a = arc.word; const float p_a = arc.loglike;
        alpha_dash_arc[0] = alpha_dash(s_a, 0) + l(w_a, 0, true);
        for (int32_t q = 1; q <= Q; q++) {
          const int32_t r_q = r(q);
          const double a1 = alpha_dash(s_a, q - 1) + l(w_a, r_q), a2 = alpha_dash(s_a, q) + l(w_a, 0, true), a3 = alpha_dash_arc[q - 1] + l(0, r_q);
          if (a1 <= a2) { if (a1 <= a3) { b_arc[q] = 1; alpha_dash_arc[q] = a1; } else { b_arc[q] = 3; alpha_dash_arc[q] = a3; } }
          else { if (a2 <= a3) { b_arc[q] = 2; alpha_dash_arc[q] = a2; } else { b_arc[q] = 3; alpha_dash_arc[q] = a3; } }
        }
        std::fill(beta_dash_arc.begin(), beta_dash_arc.end(), 0.0);
        const double w = std::exp(alpha[s_a] + p_a - alpha[n]);
        for (int32_t q = Q; q >= 1; q--) {
          beta_dash_arc[q] += w * beta_dash(n, q);
          switch (b_arc[q]) {
            case 1: beta_dash(s_a, q - 1) += beta_dash_arc[q]; AddToMap(w_a, beta_dash_arc[q], &gam[q]); AddToMap(w_a, state_times[s_a] * beta_dash_arc[q], &tau_b[q]); AddToMap(w_a, state_times[n] * beta_dash_arc[q], &tau_e[q]); break;
            case 2: beta_dash(s_a, q) += beta_dash_arc[q]; break;
            default: beta_dash_arc[q - 1] += beta_dash_arc[q]; AddToMap(0, beta_dash_arc[q], &gam[q]); AddToMap(0, state_times[n] * beta_dash_arc[q], &tau_b[q]); AddToMap(0, state_times[n] * beta_dash_arc[q], &tau_e[q]); break;
          }
        }
        beta_dash_arc[0] += w * beta_dash(n, 0);
        beta_dash(s_a, 0) += beta_dash_arc[0];
      }
    }
    std::fill(beta_dash_arc.begin(), beta_dash_arc.end(), 0.0);
    for (int32_t q = Q; q >= 1; q--) {
      beta_dash_arc[q] += beta_dash(1, q); beta_dash_arc[q - 1] += beta_dash_arc[q];
      AddToMap(0, beta_dash_arc[q], &gam[q]); AddToMap(0, state_times[1] * beta_dash_arc[q], &tau_b[q]); AddToMap(0, state_times[1] * beta_dash_arc[q], &tau_e[q]);
    }
    gamma.assign(Q, {}); times.assign(Q, {}); sausage_times.assign(Q, {0.0f, 0.0f});
    for (int32_t q = 1; q <= Q; q++) {
      for (const auto &kv : gam[q]) gamma[q - 1].push_back({kv.first, (float)kv.second});
      std::sort(gamma[q - 1].begin(), gamma[q - 1].end(), [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.second > b.second || (a.second == b.second && a.first > b.first); });
      double t_b = 0.0, t_e = 0.0;
      for (const auto &g : gamma[q - 1]) {
        const double w_b = tau_b[q][g.first], w_e = tau_e[q][g.first];
        times[q - 1].push_back({(float)(w_b / g.second), (float)(w_e / g.second)}); t_b += w_b; t_e += w_e;
      }
      sausage_times[q - 1] = {(float)t_b, (float)t_e};
      if (q > 1 && sausage_times[q - 2].second > sausage_times[q - 1].first) sausage_times[q - 2].second = sausage_times[q - 1].first = 0.5f * (sausage_times[q - 2].second + sausage_times[q - 1].first);
    }
  }
  void MbrDecode() {
    for (size_t counter = 0;; counter++) {
      NormalizeEps(&R); AccStats();
      double delta_Q = 0.0; one_best_times.clear(); one_best_conf.clear();
      for (size_t q = 0; q < R.size(); q++) {
        if (opts.decode_mbr) {
          const auto &g = gamma[q]; double old_gamma = 0, new_gamma = g[0].second; const int32_t rq = R[q], rhat = g[0].first;
          for (const auto &e : g) if (e.first == rq) old_gamma = e.second;
          delta_Q += (old_gamma - new_gamma); R[q] = rhat;
        }
        if (R[q] != 0 || opts.print_silence) {
          int32_t s = 0;
          for (size_t j = 0; j < gamma[q].size(); j++) if (gamma[q][j].first == R[q]) { s = (int32_t)j; break; }
          one_best_times.push_back(times[q][s]);
          const size_t i = one_best_times.size();
          if (i > 1 && one_best_times[i - 2].second > one_best_times[i - 1].first) {      // overlapping words: the available interval is shared out
            const float prev_right = i > 2 ? one_best_times[i - 3].second : 0.0f;
            const float left = std::max(prev_right, std::min(one_best_times[i - 2].first, one_best_times[i - 1].first)), right = std::max(one_best_times[i - 2].second, one_best_times[i - 1].second);
            const float first_dur = one_best_times[i - 2].second - one_best_times[i - 2].first, second_dur = one_best_times[i - 1].second - one_best_times[i - 1].first;
            const float mid = first_dur > 0 ? left + (right - left) * first_dur / (first_dur + second_dur) : left;
            one_best_times[i - 2].first = left; one_best_times[i - 2].second = one_best_times[i - 1].first = mid; one_best_times[i - 1].second = right;
          }
          float conf = 0.0f;
          for (const auto &e : gamma[q]) if (e.first == R[q]) { conf = e.second; break; }
          one_best_conf.push_back(conf);
        }
      }
      if (delta_Q == 0) break;
      if (counter > 100) { K3H_WARN << "Iterating too many times in MbrDecode; stopping."; break; }
    }
    if (!opts.print_silence) RemoveEps(&R);
  }
};

MinimumBayesRisk::MinimumBayesRisk(const CompactLattice &clat_in, MinimumBayesRiskOptions opts) : impl_(new Impl) {
  Impl &m = *impl_; m.opts = opts;
  CompactLattice clat(clat_in);
  // CreateSuperFinal: one final state with weight One and no arcs out
  {
    std::vector<int32_t> finals; for (int32_t s = 0; s < clat.NumStates(); s++) if (clat.is_final[s]) finals.push_back(s);
    bool done = false;
    if (finals.size() == 1) {
      const int32_t f = finals[0]; const bool one = clat.fin_graph[f] == 0.0f && clat.fin_ac[f] == 0.0f && clat.fin_str[f].empty();
      if (one && std::find(clat.arc_src.begin(), clat.arc_src.end(), f) == clat.arc_src.end()) done = true;
    }
    if (!done) {
      const int32_t fs = clat.AddState(); clat.is_final[fs] = 1;
      for (int32_t s : finals) {
        clat.arc_src.push_back(s); clat.arc_dst.push_back(fs); clat.arc_label.push_back(0); clat.arc_graph.push_back(clat.fin_graph[s]); clat.arc_ac.push_back(clat.fin_ac[s]); clat.arc_str.push_back(clat.fin_str[s]);
        clat.is_final[s] = 0; clat.fin_graph[s] = 0; clat.fin_ac[s] = 0; clat.fin_str[s].clear();
      }
    }
  }
  {      // fst::TopSort when the lattice is not known to be sorted (fst/topsort.h: depth-first from the start state, then from every state not reached yet; new numbers = reverse finishing order)
    bool sorted = clat.start == 0;
    for (size_t a = 0; a < clat.arc_src.size() && sorted; a++) sorted = clat.arc_dst[a] > clat.arc_src[a];
    if (!sorted) {
      const int32_t n = clat.NumStates(); std::vector<std::vector<int32_t>> nx(n); for (size_t a = 0; a < clat.arc_src.size(); a++) nx[clat.arc_src[a]].push_back(clat.arc_dst[a]);
      std::vector<char> color(n, 0); std::vector<size_t> pos(n, 0); std::vector<int32_t> stack, finish;
      auto visit = [&](int32_t root) {
        stack.push_back(root); color[root] = 1;
        while (!stack.empty()) {
          const int32_t u = stack.back();
          if (pos[u] < nx[u].size()) { const int32_t d = nx[u][pos[u]++]; if (color[d] == 1) K3H_ERR << "Cycles detected in lattice."; if (color[d] == 0) { color[d] = 1; stack.push_back(d); } }
          else { color[u] = 2; finish.push_back(u); stack.pop_back(); }
        }
      };
      if (clat.start >= 0) visit(clat.start);
      for (int32_t u = 0; u < n; u++) if (color[u] == 0) visit(u);
      std::vector<int32_t> newid(n); for (int32_t i = 0; i < n; i++) newid[finish[n - 1 - i]] = i;
      CompactLattice o; o.start = newid[clat.start]; o.is_final.assign(n, 0); o.fin_graph.assign(n, 0); o.fin_ac.assign(n, 0); o.fin_str.assign(n, {});
      for (int32_t u = 0; u < n; u++) { const int32_t v = newid[u]; o.is_final[v] = clat.is_final[u]; o.fin_graph[v] = clat.fin_graph[u]; o.fin_ac[v] = clat.fin_ac[u]; o.fin_str[v] = clat.fin_str[u]; }
      // arcs keep their order inside a state; states in new order
      std::vector<std::vector<int32_t>> by(n); for (size_t a = 0; a < clat.arc_src.size(); a++) by[newid[clat.arc_src[a]]].push_back((int32_t)a);
      for (int32_t v = 0; v < n; v++) for (int32_t a : by[v]) { o.arc_src.push_back(v); o.arc_dst.push_back(newid[clat.arc_dst[a]]); o.arc_label.push_back(clat.arc_label[a]); o.arc_graph.push_back(clat.arc_graph[a]); o.arc_ac.push_back(clat.arc_ac[a]); o.arc_str.push_back(clat.arc_str[a]); }
      clat = std::move(o);
    }
  }
  const int32_t N = clat.NumStates();
  if (N == 0 || clat.start != 0) K3H_ERR << "MinimumBayesRisk: empty lattice";
  // arcs grouped by source state in stored order; CompactLatticeStateTimes
  std::vector<std::vector<int32_t>> out(N); for (size_t a = 0; a < clat.arc_src.size(); a++) out[clat.arc_src[a]].push_back((int32_t)a);
  std::vector<int32_t> t(N, -1); t[0] = 0;
  for (int32_t s = 0; s < N; s++) for (int32_t a : out[s]) { const int32_t d = clat.arc_dst[a], len = (int32_t)clat.arc_str[a].size(); if (t[d] == -1) t[d] = t[s] + len; else if (t[d] != t[s] + len) K3H_ERR << "MinimumBayesRisk: inconsistent state times in the lattice"; }
  m.state_times.assign(N + 1, 0); for (int32_t s = 0; s < N; s++) m.state_times[s + 1] = t[s];
  m.pre.assign(N + 1, {});
  for (int32_t n = 1; n <= N; n++) for (int32_t a : out[n - 1]) {
    Impl::Arc arc; arc.word = clat.arc_label[a]; arc.start_node = n; arc.end_node = clat.arc_dst[a] + 1; arc.loglike = -(clat.arc_graph[a] + clat.arc_ac[a]);
    m.pre[arc.end_node].push_back((int32_t)m.arcs.size()); m.arcs.push_back(arc);
  }
  // first hypothesis: the words of the best path (ShortestPath over the lattice as a tropical FST, weights graph + acoustic in float)
  {
    std::vector<float> best(N, std::numeric_limits<float>::infinity()); std::vector<int32_t> back(N, -1); best[0] = 0.0f;
    for (int32_t s = 0; s < N; s++) if (best[s] < std::numeric_limits<float>::infinity()) for (int32_t a : out[s]) { const float c = best[s] + (clat.arc_graph[a] + clat.arc_ac[a]); const int32_t d = clat.arc_dst[a]; if (c < best[d]) { best[d] = c; back[d] = a; } }
    std::vector<int32_t> words; int32_t s = N - 1;      // the super-final state is the last one of the sorted lattice
    for (int32_t f = 0; f < N; f++) if (clat.is_final[f]) s = f;
    while (s != 0 && back[s] >= 0) { const int32_t a = back[s]; if (clat.arc_label[a] != 0) words.push_back(clat.arc_label[a]); s = clat.arc_src[a]; }
    std::reverse(words.begin(), words.end()); m.R = words; m.L = 0.0;
  }
  m.MbrDecode();
}
MinimumBayesRisk::~MinimumBayesRisk() {}
const std::vector<int32_t> &MinimumBayesRisk::GetOneBest() const { return impl_->R; }
const std::vector<std::pair<float, float>> &MinimumBayesRisk::GetOneBestTimes() const { return impl_->one_best_times; }
const std::vector<float> &MinimumBayesRisk::GetOneBestConfidences() const { return impl_->one_best_conf; }
const std::vector<std::vector<std::pair<int32_t, float>>> &MinimumBayesRisk::GetSausageStats() const { return impl_->gamma; }
const std::vector<std::pair<float, float>> &MinimumBayesRisk::GetSausageTimes() const { return impl_->sausage_times; }
double MinimumBayesRisk::GetBayesRisk() const { return impl_->L; }

// ---- LatticePostprocessor -----------------------------------------------------------------------------------------------------
void LatticePostprocessorConfig::Register(ParseOptions *po) {
  po->Register("max-expand", &max_expand, "If >0, the maximum amount by which this program will expand lattices before refusing to continue.");
  po->Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods"); po->Register("lm-scale", &lm_scale, "Scaling factor for graph/lm costs");
  po->Register("acoustic2lm-scale", &acoustic2lm_scale, "Add this times original acoustic costs to LM costs"); po->Register("lm2acoustic-scale", &lm2acoustic_scale, "Add this times original LM costs to acoustic costs");
  po->Register("word-ins-penalty", &word_ins_penalty, "Word insertion penalty."); po->Register("word-boundary-rxfilename", &word_boundary_rxfilename, "Word boundary file");
  po->Register("decode-mbr", &mbr_opts.decode_mbr, "If true, do Minimum Bayes Risk decoding (else, Maximum a Posteriori)");
  po->Register("print-silence", &mbr_opts.print_silence, "Keep the inter-word '<eps>' bins in the 1-best output (ctm, <eps> can be a 'silence' or a 'deleted' word)");
  po->Register("silence-label", &silence_label, "Numeric id of word symbol that is to be used for silence arcs in the word-aligned lattice (zero is OK)");
  po->Register("partial-word-label", &partial_word_label, "Numeric id of word symbol that is to be used for arcs in the word-aligned lattice corresponding to partial words at the end of forced-out utterances (zero is OK)");
  po->Register("reorder", &reorder, "True if the lattices were generated from graphs that had the --reorder option true");
}
LatticePostprocessor::LatticePostprocessor(const LatticePostprocessorConfig &config) : config_(config) {
  use_lattice_scale_ = config_.lm_scale != 1.0f || config_.acoustic2lm_scale != 0.0f || config_.lm2acoustic_scale != 0.0f || config_.acoustic_scale != 1.0f;
  if (!config_.word_boundary_rxfilename.empty())
    K3H_ERR << "LatticePostprocessor: --word-boundary-rxfilename (WordAlignLattice, lat/word-align-lattice.cc) is not part of this build; without it the words' times come from the "
               "MBR statistics over the lattice as it is (what the reference does when no word-boundary file is configured)";
}
bool LatticePostprocessor::GetPostprocessedLattice(CompactLattice &clat, CompactLattice *out) const {
  if (clat.NumStates() == 0) return true;
  if (use_lattice_scale_) {      // ScaleTupleWeight (fstext/lattice-utils-inl.h:175-195): both new values from the OLD pair, in double
    const double s00 = config_.lm_scale, s01 = config_.acoustic2lm_scale, s10 = config_.lm2acoustic_scale, s11 = config_.acoustic_scale;
    auto sc = [&](float &g, float &a) { const double v1 = g, v2 = a; g = (float)(s00 * v1 + s01 * v2); a = (float)(s10 * v1 + s11 * v2); };
    for (size_t k = 0; k < clat.arc_src.size(); k++) sc(clat.arc_graph[k], clat.arc_ac[k]);
    for (int32_t s = 0; s < clat.NumStates(); s++) if (clat.is_final[s]) sc(clat.fin_graph[s], clat.fin_ac[s]);
  }
  if (config_.word_ins_penalty > 0.0f) for (size_t k = 0; k < clat.arc_src.size(); k++) if (clat.arc_label[k] != 0) clat.arc_graph[k] += config_.word_ins_penalty;
  if (decoder_frame_shift_ == 0.0f) K3H_ERR << "SetDecoderFrameShift() must be called (typically by pipeline)";
  *out = clat; return true;
}
bool LatticePostprocessor::GetCTM(CompactLattice &clat, CtmResult *ctm) const {
  if (clat.NumStates() == 0) return true;
  CompactLattice pp; GetPostprocessedLattice(clat, &pp);
  MinimumBayesRisk mbr(pp, config_.mbr_opts);
  ctm->conf = mbr.GetOneBestConfidences(); ctm->words = mbr.GetOneBest(); ctm->times_seconds = mbr.GetOneBestTimes();
  for (auto &p : ctm->times_seconds) { p.first *= decoder_frame_shift_; p.second *= decoder_frame_shift_; }
  return true;
}
std::shared_ptr<LatticePostprocessor> LoadLatticePostprocessor(const std::string &config_rxfilename) {
  ParseOptions po(""); LatticePostprocessorConfig c; c.Register(&po); po.ReadConfigFile(config_rxfilename);
  return std::make_shared<LatticePostprocessor>(c);
}
void WriteCtm(const CtmResult &ctm, const std::string &key, std::ostream &os, const std::vector<std::string> *word_syms) {
  os << std::fixed; os.precision(2);      // KALDI_CUDA_DECODER_BIN_FLOAT_PRINT_PRECISION
  for (size_t i = 0; i < ctm.times_seconds.size(); i++) {
    const float from = ctm.times_seconds[i].first, to = ctm.times_seconds[i].second;
    os << key << " " << 0 << "  " << from << ' ' << (to - from) << ' ';
    const int32_t w = ctm.words[i];
    if (word_syms && w >= 0 && (size_t)w < word_syms->size() && !(*word_syms)[w].empty()) os << (*word_syms)[w]; else os << w;
    os << ' ' << ctm.conf[i] << '\n';
  }
}
}  // namespace k3host
