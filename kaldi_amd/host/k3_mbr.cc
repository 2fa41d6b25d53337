// k3_mbr.cc -- word-level Minimum Bayes Risk decoding of a CompactLattice and the lattice post-processor of the reference's CUDA pipeline, restated
// over k3host::CompactLattice (host side of the path: SURVEY 8f row 1 / row P).
//   MinimumBayesRisk      lat/sausages.{h,cc} (Xu, Povey, Mangu, Zhu: "Minimum Bayes Risk decoding and system combination based on a recursion for edit
//                         distance"): Figures 4-6 of the paper as the reference codes them -- EditDistance :130-166, AccStats :169-318, MbrDecode :28-108,
//                         PrepareLatticeAndInitStats :320-363 (CreateSuperFinal fstext/pre-determinize-inl.h:689-724, CompactLatticeStateTimes
//                         lat/lattice-functions.cc:109-147), first hypothesis = the lattice's best path (:377-395)
//   WordAlignLattice      lat/word-align-lattice.{h,cc} (round 5): the lattice re-cut so that every arc is one word / silence with exactly its transition-ids -- what the post-processor
// does in front of MBR when its config names --word-boundary-rxfilename; pinned to the reference's own source (oracle/_ref/bin/ref-word-align,
// tests/test_word_align.py)
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
#include <unordered_map>

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
struct Dense {
  int32_t cols = 0;
  std::vector<double> v;
  Dense(int32_t r, int32_t c) : cols(c), v((size_t)r * c, 0.0) {} double &operator()(int32_t r, int32_t c) {
    return v[(size_t)r * cols + c];
  }
};
}  // namespace

struct MinimumBayesRisk::Impl {
  struct Arc { int32_t word, start_node, end_node; float loglike; };
  MinimumBayesRiskOptions opts; std::vector<Arc> arcs; std::vector<std::vector<int32_t>> pre; std::vector<int32_t> state_times; std::vector<int32_t> R; double L = 0.0;
  std::vector<std::vector<std::pair<int32_t, float>>> gamma;
  std::vector<std::vector<std::pair<float, float>>> times;
  std::vector<std::pair<float, float>> sausage_times, one_best_times;
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
            case 1: beta_dash(s_a, q - 1) += beta_dash_arc[q];
            AddToMap(w_a, beta_dash_arc[q], &gam[q]);
            AddToMap(w_a, state_times[s_a] * beta_dash_arc[q], &tau_b[q]);
            AddToMap(w_a, state_times[n] * beta_dash_arc[q], &tau_e[q]);
            break;
            case 2: beta_dash(s_a, q) += beta_dash_arc[q]; break;
            default: beta_dash_arc[q - 1] += beta_dash_arc[q];
            AddToMap(0, beta_dash_arc[q], &gam[q]);
            AddToMap(0, state_times[n] * beta_dash_arc[q], &tau_b[q]);
            AddToMap(0, state_times[n] * beta_dash_arc[q], &tau_e[q]);
            break;
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
      std::sort(gamma[q - 1].begin(), gamma[q - 1].end(),
          [](const std::pair<int32_t, float> &a, const std::pair<int32_t,
          float> &b) { return a.second > b.second || (a.second == b.second && a.first > b.first); });
      double t_b = 0.0, t_e = 0.0;
      for (const auto &g : gamma[q - 1]) {
        const double w_b = tau_b[q][g.first], w_e = tau_e[q][g.first];
        times[q - 1].push_back({(float)(w_b / g.second), (float)(w_e / g.second)}); t_b += w_b; t_e += w_e;
      }
      sausage_times[q - 1] = {(float)t_b, (float)t_e};
      if (q > 1 && sausage_times[q - 2].second > sausage_times[q - 1].first) sausage_times[q - 2].second = sausage_times[q - 1].first =
          0.5f * (sausage_times[q - 2].second + sausage_times[q - 1].first);
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
            const float left = std::max(prev_right, std::min(one_best_times[i - 2].first, one_best_times[i - 1].first)),
                right = std::max(one_best_times[i - 2].second, one_best_times[i - 1].second);
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
        clat.arc_src.push_back(s);
        clat.arc_dst.push_back(fs);
        clat.arc_label.push_back(0);
        clat.arc_graph.push_back(clat.fin_graph[s]);
        clat.arc_ac.push_back(clat.fin_ac[s]);
        clat.arc_str.push_back(clat.fin_str[s]);
        clat.is_final[s] = 0; clat.fin_graph[s] = 0; clat.fin_ac[s] = 0; clat.fin_str[s].clear();
      }
    }
  }
  // fst::TopSort when the lattice is not known to be sorted (fst/topsort.h: depth-first from the start state, then from every state not reached yet; new
  // numbers = reverse finishing order)
  {
    bool sorted = clat.start == 0;
    for (size_t a = 0; a < clat.arc_src.size() && sorted; a++) sorted = clat.arc_dst[a] > clat.arc_src[a];
    if (!sorted) {
      const int32_t n = clat.NumStates(); std::vector<std::vector<int32_t>> nx(n); for (size_t a = 0; a < clat.arc_src.size(); a++) nx[clat.arc_src[a]].push_back(clat.arc_dst[a]);
      std::vector<char> color(n, 0); std::vector<size_t> pos(n, 0); std::vector<int32_t> stack, finish;
      auto visit = [&](int32_t root) {
        stack.push_back(root); color[root] = 1;
        while (!stack.empty()) {
          const int32_t u = stack.back();
          if (pos[u] < nx[u].size()) {
            const int32_t d = nx[u][pos[u]++];
            if (color[d] == 1) K3H_ERR << "Cycles detected in lattice.";
            if (color[d] == 0) {
              color[d] = 1;
              stack.push_back(d);
            }
          }
          else { color[u] = 2; finish.push_back(u); stack.pop_back(); }
        }
      };
      if (clat.start >= 0) visit(clat.start);
      for (int32_t u = 0; u < n; u++) if (color[u] == 0) visit(u);
      std::vector<int32_t> newid(n); for (int32_t i = 0; i < n; i++) newid[finish[n - 1 - i]] = i;
      CompactLattice o; o.start = newid[clat.start]; o.is_final.assign(n, 0); o.fin_graph.assign(n, 0); o.fin_ac.assign(n, 0); o.fin_str.assign(n, {});
      for (int32_t u = 0; u < n; u++) {
        const int32_t v = newid[u];
        o.is_final[v] = clat.is_final[u];
        o.fin_graph[v] = clat.fin_graph[u];
        o.fin_ac[v] = clat.fin_ac[u];
        o.fin_str[v] = clat.fin_str[u];
      }
      // arcs keep their order inside a state; states in new order
      std::vector<std::vector<int32_t>> by(n); for (size_t a = 0; a < clat.arc_src.size(); a++) by[newid[clat.arc_src[a]]].push_back((int32_t)a);
      for (int32_t v = 0; v < n; v++) for (int32_t a : by[v]) {
        o.arc_src.push_back(v);
        o.arc_dst.push_back(newid[clat.arc_dst[a]]);
        o.arc_label.push_back(clat.arc_label[a]);
        o.arc_graph.push_back(clat.arc_graph[a]);
        o.arc_ac.push_back(clat.arc_ac[a]);
        o.arc_str.push_back(clat.arc_str[a]);
      }
      clat = std::move(o);
    }
  }
  const int32_t N = clat.NumStates();
  if (N == 0 || clat.start != 0) K3H_ERR << "MinimumBayesRisk: empty lattice";
  // arcs grouped by source state in stored order; CompactLatticeStateTimes
  std::vector<std::vector<int32_t>> out(N); for (size_t a = 0; a < clat.arc_src.size(); a++) out[clat.arc_src[a]].push_back((int32_t)a);
  std::vector<int32_t> t(N, -1); t[0] = 0;
  for (int32_t s = 0; s < N; s++) for (int32_t a : out[s]) {
    const int32_t d = clat.arc_dst[a], len = (int32_t)clat.arc_str[a].size();
    if (t[d] == -1) t[d] = t[s] + len;
    else if (t[d] != t[s] + len) K3H_ERR << "MinimumBayesRisk: inconsistent state times in the lattice";
  }
  m.state_times.assign(N + 1, 0); for (int32_t s = 0; s < N; s++) m.state_times[s + 1] = t[s];
  m.pre.assign(N + 1, {});
  for (int32_t n = 1; n <= N; n++) for (int32_t a : out[n - 1]) {
    Impl::Arc arc; arc.word = clat.arc_label[a]; arc.start_node = n; arc.end_node = clat.arc_dst[a] + 1; arc.loglike = -(clat.arc_graph[a] + clat.arc_ac[a]);
    m.pre[arc.end_node].push_back((int32_t)m.arcs.size()); m.arcs.push_back(arc);
  }
  // first hypothesis: the words of the best path (ShortestPath over the lattice as a tropical FST, weights graph + acoustic in float)
  {
    std::vector<float> best(N, std::numeric_limits<float>::infinity()); std::vector<int32_t> back(N, -1); best[0] = 0.0f;
    for (int32_t s = 0; s < N; s++) if (best[s] < std::numeric_limits<float>::infinity()) for (int32_t a : out[s]) {
      const float c = best[s] + (clat.arc_graph[a] + clat.arc_ac[a]);
      const int32_t d = clat.arc_dst[a];
      if (c < best[d]) {
        best[d] = c;
        back[d] = a;
      }
    }
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

// ---- WordAlignLattice (lat/word-align-lattice.cc) -------------------------------------------------------------------------------
// The reference builds a new lattice whose states are (input state, computation state) pairs -- the computation state holds the transition-ids and word labels seen but not yet
// put on an output arc, and the weight that goes with them -- expands them from a LIFO queue (:204-248), lets a computation state emit an arc as soon as it holds a complete word /
// silence (OutputNormalWordArc / OutputSilenceArc / OutputOnePhoneWordArc, :349-533; here: OutputCompleteUnit) before any more input is read, forces out what is left at the final state (:591-667), and
// then removes the epsilon arcs that carried the input (fst::RmEpsilon + Connect) and, when the silence / partial-word labels were asked to be 0, maps the stand-in labels back
// (RemoveSomeInputSymbols + Project, :292-306).  Restated over plain vectors; the order in which states and arcs are created is the reference's (the pin compares whole lattices).
namespace {
struct LW { float g = 0.0f, a = 0.0f; };      // LatticeWeight: (graph, acoustic)
const float kInfF = std::numeric_limits<float>::infinity();
inline bool IsZero(const LW &w) { return w.g == kInfF && w.a == kInfF; }
inline LW TimesLW(const LW &x, const LW &y) { return LW{x.g + y.g, x.a + y.a}; }
inline int CompareLW(const LW &x, const LW &y) {      // fstext/lattice-weight.h:299-315: > 0 when x is better (lower cost)
  const float f1 = x.g + x.a, f2 = y.g + y.a;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (x.g < y.g) return 1;
  if (x.g > y.g) return -1;
  return 0;
}
struct CW { LW w; std::vector<int32_t> str; };      // CompactLatticeWeight
inline CW ZeroCW() { return CW{LW{kInfF, kInfF}, {}}; }
inline int CompareCW(const CW &x, const CW &y) {      // :674-698: weight first, then the SHORTER string is better, then lexicographic
  const int c = CompareLW(x.w, y.w); if (c != 0) return c;
  const size_t l1 = x.str.size(), l2 = y.str.size();
  if (l1 > l2) return -1;
  if (l1 < l2) return 1;
  for (size_t i = 0; i < l1; i++) {
    if (x.str[i] < y.str[i]) return -1;
    if (x.str[i] > y.str[i]) return 1;
  }
  return 0;
}
inline CW PlusCW(const CW &x, const CW &y) { return CompareCW(x, y) >= 0 ? x : y; }
inline CW TimesCW(const CW &x, const CW &y) {      // :740-764 (Zero absorbs)
  if (IsZero(x.w) || IsZero(y.w)) return ZeroCW();
  CW r; r.w = TimesLW(x.w, y.w); r.str = x.str; r.str.insert(r.str.end(), y.str.begin(), y.str.end()); return r;
}
struct WArc { int32_t label, next; CW w; };
struct WFst { int32_t start = -1; std::vector<std::vector<WArc>> arcs; std::vector<CW> fin;
  int32_t AddState() { arcs.emplace_back(); fin.push_back(ZeroCW()); return (int32_t)fin.size() - 1; } int32_t NumStates() const { return (int32_t)fin.size(); } };

struct CompState {      // LatticeWordAligner::ComputationState (:30-128)
  std::vector<int32_t> tids, words; LW weight;
  bool operator==(const CompState &o) const { return tids == o.tids && words == o.words && weight.g == o.weight.g && weight.a == o.weight.a; }
  bool IsEmpty() const { return tids.empty() && words.empty(); }
};
struct AlignCtx { const TransitionInfo &tm; const WordBoundaryInfo &info; bool *error;
  int32_t Phone(int32_t tid) const {
    if (tid <= 0 || (size_t)tid >= tm.id2phone.size()) K3H_ERR << "WordAlignLattice: transition-id " << tid << " is not in the model";
    return tm.id2phone[tid];
  }
  // (every id goes through Phone()'s range check first: the unit walkers ask Final() before they ask Phone())
  bool Final(int32_t tid) const { Phone(tid); return tm.is_final[tid] != 0; } bool SelfLoop(int32_t tid) const { Phone(tid); return tm.self_loop[tid] != 0; } };

// What a computation state can emit before more input is read (:349-545) is ONE complete unit at the front of its transition-ids: a silence, a one-phone word or a word of
// several phones.  The class of the first phone says which (the three are exclusive), and a unit is a run of PHONE SPANS -- a span = the transition-ids of one phone up to and
// including its final transition-id and, with --reorder, the self-loops behind it -- so one routine walks them from a table instead of one function per unit.  What differs
// between the units is in the table: whether a word label goes with it, and, because the lattices this runs on may be broken, which of the reference's consistency checks
// warn and which of them also raise the aligner's error flag (the flag silences every later warning and decides WordAlignLattice's return value, so the pin needs it exact).
struct SpanRule { bool check_phone, raise_on_change;        // while looking for the final transition-id: compare every id's phone with the span's / also raise the flag
                  bool raise_after; const char *after_msg; };      // behind the self-loops: the last id's phone is compared; flag; the warning's text
struct UnitRule { WordBoundaryInfo::PhoneType opens; bool takes_word, has_inner_phones; SpanRule first, last; };
const char *const kMsgChanged = "Phone changed before final transition-id found [broken lattice or mismatched model or wrong --reorder option?]";
const char *const kMsgUnexpected = "Phone changed unexpectedly in lattice [broken lattice or mismatched model?]";
const UnitRule kUnitRules[3] = {
    {WordBoundaryInfo::kNonWordPhone, false, false, {true, true, false, kMsgUnexpected}, {}},                  // silence (:349-393)
    {WordBoundaryInfo::kWordBeginAndEndPhone, true, false, {true, false, true, kMsgUnexpected}, {}},         // one-phone word (:396-446)
    {WordBoundaryInfo::kWordBeginPhone, true, true, {false, false, true, kMsgUnexpected},                     // word of at least two phones (:451-545): begin phone ...
     {true, true, true, "Phone changed while following final self-loop [broken lattice or mismatched model or wrong --reorder option?]"}}};      // ... and end phone
// one span starting at *pos; false when the ids run out before the span is known to be over (the unit cannot be emitted yet)
bool WalkSpan(const CompState &c, const AlignCtx &x, const SpanRule &r, int32_t phone, size_t *pos) {
  const size_t len = c.tids.size(); size_t i = *pos;
  for (; i < len; i++) {
    if (r.check_phone && x.Phone(c.tids[i]) != phone && !*x.error) { if (r.raise_on_change) *x.error = true; K3H_WARN << kMsgChanged; }
    if (x.Final(c.tids[i])) break;
  }
  if (i == len) return false;
  i++;
  if (x.info.reorder) while (i < len && x.SelfLoop(c.tids[i])) i++;
  if (i == len) return false;
  if (x.Phone(c.tids[i - 1]) != phone && !*x.error) { if (r.raise_after) *x.error = true; K3H_WARN << r.after_msg; }
  *pos = i;
  return true;
}
bool OutputCompleteUnit(CompState &c, const AlignCtx &x, WArc *out) {
  if (c.tids.empty()) return false;
  const int32_t phone0 = x.Phone(c.tids[0]); const WordBoundaryInfo::PhoneType type0 = x.info.TypeOfPhone(phone0);
  const UnitRule *u = nullptr; for (const UnitRule &r : kUnitRules) if (r.opens == type0) u = &r;
  if (!u || (u->takes_word && c.words.empty())) return false;
  size_t i = 0;
  if (!WalkSpan(c, x, u->first, phone0, &i)) return false;
  if (u->has_inner_phones) {
    const size_t len = c.tids.size();
    for (; i < len; i++) {      // the word-internal phones, up to the first id of a word-end phone
      const int32_t ph = x.Phone(c.tids[i]); const WordBoundaryInfo::PhoneType t = x.info.TypeOfPhone(ph);
      if (t == WordBoundaryInfo::kWordEndPhone) break;
      if (t != WordBoundaryInfo::kWordInternalPhone && !*x.error) { K3H_WARN << "Unexpected phone " << ph << " found inside a word."; *x.error = true; }
    }
    if (i == len) return false;
    if (!WalkSpan(c, x, u->last, x.Phone(c.tids[i]), &i)) return false;
  }
  *out = WArc{u->takes_word ? c.words[0] : x.info.silence_label, -1, CW{c.weight, std::vector<int32_t>(c.tids.begin(), c.tids.begin() + i)}};
  c.tids.erase(c.tids.begin(), c.tids.begin() + i); if (u->takes_word) c.words.erase(c.words.begin()); c.weight = LW();
  return true;
}
bool IsPlausibleWord(const AlignCtx &x, const std::vector<int32_t> &tids) {      // :549-569
  if (tids.empty()) return false;
  const int32_t first_phone = x.Phone(tids.front()), last_phone = x.Phone(tids.back());
  if ((x.info.TypeOfPhone(first_phone) == WordBoundaryInfo::kWordBeginAndEndPhone && first_phone == last_phone) ||
      (x.info.TypeOfPhone(first_phone) == WordBoundaryInfo::kWordBeginPhone && x.info.TypeOfPhone(last_phone) == WordBoundaryInfo::kWordEndPhone)) {
    if (!x.info.reorder) return x.Final(tids.back());
    int32_t i = (int32_t)tids.size() - 1; while (i > 0 && x.SelfLoop(tids[i])) i--;
    return x.Final(tids[i]);
  }
  return false;
}
void OutputArcForce(CompState &c, const AlignCtx &x, WArc *out) {      // :572-667
  if (!c.words.empty() && !c.tids.empty()) {
    const int32_t word = c.words[0];
    if (!*x.error && !IsPlausibleWord(x, c.tids)) { *x.error = true; K3H_WARN << "Invalid word at end of lattice [partial lattice, forced out?]"; }
    *out = WArc{word, -1, CW{c.weight, c.tids}}; c.weight = LW(); c.tids.clear(); c.words.erase(c.words.begin());
  } else if (!c.words.empty() && c.tids.empty()) {
    if (!*x.error) { *x.error = true; K3H_WARN << "Discarding word-ids at the end of a sentence, that don't have alignments."; }
    *out = WArc{0, -1, CW{c.weight, c.tids}}; c.weight = LW(); c.words.clear();
  } else if (!c.tids.empty() && c.words.empty()) {
    const int32_t first_phone = x.Phone(c.tids[0]);
    if (x.info.TypeOfPhone(first_phone) == WordBoundaryInfo::kNonWordPhone) {
      if (first_phone != x.Phone(c.tids.back()) && !*x.error) { *x.error = true; K3H_ERR << "Broken silence arc at end of utterance (the phone changed); code error"; }
      if (!*x.error) {
        int32_t i = (int32_t)c.tids.size() - 1;
        if (x.info.reorder) while (x.SelfLoop(c.tids[i]) && i > 0) i--;
        if (!x.Final(c.tids[i])) { *x.error = true; K3H_WARN << "Broken silence arc at end of utterance (does not reach end of silence)"; }
      }
      *out = WArc{x.info.silence_label, -1, CW{c.weight, c.tids}};
    } else {
      if (!*x.error) { *x.error = true; K3H_WARN << "Partial word detected at end of utterance"; }
      *out = WArc{x.info.partial_word_label, -1, CW{c.weight, c.tids}};
    }
    c.tids.clear(); c.weight = LW();
  } else K3H_ERR << "Code error, word-aligning lattice";
}

// fst::RmEpsilon(fst, connect = true) over the compact-lattice semiring, for the aligner's output (its epsilon arcs -- label 0 -- form no cycle): see
// third_party/minifst/fst/fstlib.h
// for the algorithm's statement; states that only epsilon arcs reach are dropped, every other state gets the non-epsilon arcs of its epsilon closure
void RmEpsilonAndConnect(WFst *f) {
  const int32_t n = f->NumStates(); if (f->start < 0) return;
  std::vector<char> noneps_in(n, 0); noneps_in[f->start] = 1;
  for (int32_t s = 0; s < n; s++) for (const WArc &a : f->arcs[s]) if (a.label != 0) noneps_in[a.next] = 1;
  std::vector<int32_t> order; bool top = true;
  for (int32_t s = 0; s < n && top; s++) for (const WArc &a : f->arcs[s]) if (a.next <= s) { top = false; break; }
  if (top) for (int32_t s = 0; s < n; s++) order.push_back(s);
  else {
    std::vector<char> color(n, 0); std::vector<size_t> pos(n, 0); std::vector<int32_t> stack, finish;
    auto visit = [&](int32_t root) {
      stack.push_back(root);
      color[root] = 1;
      while (!stack.empty()) {
        const int32_t s = stack.back();
        if (pos[s] < f->arcs[s].size()) {
          const int32_t d = f->arcs[s][pos[s]++].next;
          if (color[d] == 1) K3H_ERR << "WordAlignLattice: cycle in the aligned lattice";
          if (color[d] == 0) {
            color[d] = 1;
            stack.push_back(d);
          }
        } else {
          color[s] = 2;
          finish.push_back(s);
          stack.pop_back();
        }
      }
    };
    visit(f->start); for (int32_t s = 0; s < n; s++) if (color[s] == 0) visit(s);
    order.assign(finish.rbegin(), finish.rend());
  }
  std::vector<CW> dist(n); std::vector<char> has(n, 0), visited(n, 0);
  while (!order.empty()) {
    const int32_t source = order.back(); order.pop_back();
    if (!noneps_in[source]) continue;
    std::vector<int32_t> q; std::vector<int32_t> touched;
    dist[source] = CW(); has[source] = 1; q.push_back(source); touched.push_back(source);
    for (size_t h = 0; h < q.size(); h++) {      // shortest distances over epsilon arcs (label-correcting: exact for a path semiring)
      const int32_t s = q[h];
      for (const WArc &a : f->arcs[s]) {
        if (a.label != 0) continue;
        CW w = TimesCW(dist[s], a.w);
        if (!has[a.next] || CompareCW(w, dist[a.next]) > 0) { if (!has[a.next]) touched.push_back(a.next); dist[a.next] = std::move(w); has[a.next] = 1; q.push_back(a.next); }
      }
    }
    std::vector<WArc> arcs; CW final_weight = ZeroCW(); std::vector<int32_t> eps_stack, seen; eps_stack.push_back(source);
    while (!eps_stack.empty()) {
      const int32_t s = eps_stack.back(); eps_stack.pop_back();
      if (visited[s]) continue;
      visited[s] = 1; seen.push_back(s);
      for (const WArc &a0 : f->arcs[s]) {
        if (a0.label == 0) { if (!visited[a0.next]) eps_stack.push_back(a0.next); continue; }
        WArc arc{a0.label, a0.next, TimesCW(dist[s], a0.w)}; bool merged = false;
        for (WArc &b : arcs) if (b.label == arc.label && b.next == arc.next) { b.w = PlusCW(b.w, arc.w); merged = true; break; }
        if (!merged) arcs.push_back(std::move(arc));
      }
      final_weight = PlusCW(final_weight, TimesCW(dist[s], f->fin[s]));
    }
    for (int32_t s : touched) has[s] = 0;
    for (int32_t s : seen) visited[s] = 0;
    f->fin[source] = final_weight; f->arcs[source].clear();
    while (!arcs.empty()) { f->arcs[source].push_back(std::move(arcs.back())); arcs.pop_back(); }
  }
  for (int32_t s = 0; s < n; s++) if (!noneps_in[s]) f->arcs[s].clear();
  // fst::Connect: accessible and co-accessible states, in their old relative order
  std::vector<char> acc(n, 0), co(n, 0); std::vector<int32_t> st; std::vector<std::vector<int32_t>> rev(n);
  for (int32_t s = 0; s < n; s++) for (const WArc &a : f->arcs[s]) rev[a.next].push_back(s);
  acc[f->start] = 1; st.push_back(f->start);
  while (!st.empty()) { const int32_t s = st.back(); st.pop_back(); for (const WArc &a : f->arcs[s]) if (!acc[a.next]) { acc[a.next] = 1; st.push_back(a.next); } }
  for (int32_t s = 0; s < n; s++) if (!IsZero(f->fin[s].w)) { co[s] = 1; st.push_back(s); }
  while (!st.empty()) { const int32_t s = st.back(); st.pop_back(); for (int32_t p_ : rev[s]) if (!co[p_]) { co[p_] = 1; st.push_back(p_); } }
  std::vector<int32_t> newid(n, -1); int32_t m = 0; for (int32_t s = 0; s < n; s++) if (acc[s] && co[s]) newid[s] = m++;
  if (m == n) return;
  WFst g; if (m == 0 || newid[f->start] < 0) { *f = g; return; }
  g.arcs.resize(m); g.fin.resize(m); g.start = newid[f->start];
  for (int32_t s = 0; s < n; s++) if (newid[s] >= 0) {
    g.fin[newid[s]] = f->fin[s];
    for (WArc &a : f->arcs[s]) if (newid[a.next] >= 0) {
      a.next = newid[a.next];
      g.arcs[newid[s]].push_back(std::move(a));
    }
  }
  *f = std::move(g);
}
}  // namespace

WordBoundaryInfo::PhoneType WordBoundaryInfo::TypeOfPhone(int32_t p) const {
  if (p < 0 || (size_t)p >= phone_to_type.size()) K3H_ERR << "Phone " << p << " was not specified in word-boundary file (or options)";
  return phone_to_type[p];
}
WordBoundaryInfo ReadWordBoundaryInfo(const std::string &rxfilename, bool reorder, int32_t silence_label, int32_t partial_word_label) {      // WordBoundaryInfo::Init (:696-722)
  WordBoundaryInfo info; info.reorder = reorder; info.silence_label = silence_label; info.partial_word_label = partial_word_label;
  std::istringstream is(ReadWholeInput(rxfilename)); std::string line;
  while (std::getline(is, line)) {
    std::istringstream ls(line); std::vector<std::string> f; std::string t; while (ls >> t) f.push_back(t);
    char *end = nullptr; const long p = f.size() == 2 ? strtol(f[0].c_str(), &end, 10) : 0;
    if (f.size() != 2 || *end != '\0' || f[0].empty()) K3H_ERR << "Invalid line in word-boundary file: " << line;
    if (p <= 0) K3H_ERR << "Invalid line in word-boundary file (phone ids are positive): " << line;
    if (info.phone_to_type.size() <= (size_t)p) info.phone_to_type.resize(p + 1, WordBoundaryInfo::kNoPhone);
    if (f[1] == "nonword") info.phone_to_type[p] = WordBoundaryInfo::kNonWordPhone; else if (f[1] == "begin") info.phone_to_type[p] = WordBoundaryInfo::kWordBeginPhone;
    else if (f[1] == "singleton") info.phone_to_type[p] = WordBoundaryInfo::kWordBeginAndEndPhone; else if (f[1] == "end") info.phone_to_type[p] = WordBoundaryInfo::kWordEndPhone;
    else if (f[1] == "internal") info.phone_to_type[p] = WordBoundaryInfo::kWordInternalPhone; else K3H_ERR << "Invalid line in word-boundary file: " << line;
  }
  if (info.phone_to_type.empty()) K3H_ERR << "Empty word-boundary file";
  return info;
}

bool WordAlignLattice(const CompactLattice &lat, const TransitionInfo &tmodel, const WordBoundaryInfo &info_in, int32_t max_states, CompactLattice *lat_out) {
  *lat_out = CompactLattice();
  // ---- the input as per-state arc lists + CreateSuperFinal (fstext/pre-determinize-inl.h:689-724): one final state, weight One, reached by epsilon arcs with
  // the old final weights
  WFst in; in.start = lat.start; const int32_t n0 = lat.NumStates();
  for (int32_t s = 0; s < n0; s++) { in.AddState(); if (lat.is_final[s]) in.fin[s] = CW{LW{lat.fin_graph[s], lat.fin_ac[s]}, lat.fin_str[s]}; }
  int32_t highest_label = 0;
  for (size_t k = 0; k < lat.arc_src.size(); k++) {
    in.arcs[lat.arc_src[k]].push_back(WArc{lat.arc_label[k], lat.arc_dst[k], CW{LW{lat.arc_graph[k], lat.arc_ac[k]}, lat.arc_str[k]}});
    highest_label = std::max(highest_label, lat.arc_label[k]);
  }
  {
    bool idet = true, ieps = false;
    for (int32_t s = 0; s < n0; s++) {
      std::vector<int32_t> l;
      for (const WArc &a : in.arcs[s]) {
        l.push_back(a.label);
        if (a.label == 0) ieps = true;
      }
      std::sort(l.begin(), l.end());
      if (std::adjacent_find(l.begin(), l.end()) != l.end()) idet = false;
    }
    if (!idet || ieps) K3H_WARN <<
        "[Lattice has input epsilons and/or is not input-deterministic (in Mohri sense)]-- i.e. lattice is not deterministic.  Word-alignment may be slow and-or blow up in memory.";
  }
  {
    std::vector<int32_t> finals; for (int32_t s = 0; s < n0; s++) if (!IsZero(in.fin[s].w)) finals.push_back(s);
    const bool single = finals.size() == 1 && in.fin[finals[0]].w.g == 0.0f && in.fin[finals[0]].w.a == 0.0f && in.fin[finals[0]].str.empty() && in.arcs[finals[0]].empty();
    if (!single) { const int32_t fs = in.AddState(); in.fin[fs] = CW(); for (int32_t s : finals) { in.arcs[s].push_back(WArc{0, fs, in.fin[s]}); in.fin[s] = ZeroCW(); } }
  }
  WordBoundaryInfo info = info_in;      // zero silence / partial-word labels are replaced by unused ones while the epsilons are removed (:268-283)
  if (info.partial_word_label == 0 || info.silence_label == 0) {
    int32_t unused = 1 + highest_label;
    if (info.partial_word_label >= unused) unused = info.partial_word_label + 1;
    if (info.silence_label >= unused) unused = info.silence_label + 1;
    if (info.partial_word_label == 0) info.partial_word_label = unused++;
    if (info.silence_label == 0) info.silence_label = unused;
  }
  bool error = false; const AlignCtx ctx{tmodel, info, &error};
  WFst out;
  struct Tuple { int32_t state; CompState c; };
  struct TupleHash {
    size_t operator()(const Tuple &t) const {
      size_t h = (size_t)t.state;
      for (int32_t v : t.c.tids) h = h * 7853 + (size_t)v;
      for (int32_t v : t.c.words) h = h * 90647 + (size_t)v;
      return h;
    }
  };
  struct TupleEq { bool operator()(const Tuple &x, const Tuple &y) const { return x.state == y.state && x.c == y.c; } };
  std::unordered_map<Tuple, int32_t, TupleHash, TupleEq> map; std::vector<std::pair<Tuple, int32_t>> queue;
  // GetStateForTuple(.., true)
  auto state_for = [&](const Tuple &t) {
    auto it = map.find(t);
    if (it != map.end()) return it->second;
    const int32_t o = out.AddState();
    map.emplace(t, o);
    queue.push_back({t, o});
    return o;
  };
  bool ok = true;
  if (in.start < 0) { K3H_WARN << "Trying to word-align empty lattice."; return false; }
  out.start = state_for(Tuple{in.start, CompState()});
  while (!queue.empty()) {
    if (max_states > 0 && out.NumStates() > max_states) {
      K3H_WARN << "Number of states in lattice exceeded max-states of " << max_states << ", original lattice had " << in.NumStates() <<
          " states.  Returning what we have.";
      ok = false;
      break;
    }
    Tuple tuple = std::move(queue.back().first); const int32_t ostate = queue.back().second; queue.pop_back();      // ProcessQueueElement (:204-248)
    WArc arc;
    // something complete is pending: emit it before reading on
    if (OutputCompleteUnit(tuple.c, ctx, &arc)) {
      arc.next = state_for(tuple); out.arcs[ostate].push_back(std::move(arc));
      continue;
    }
    if (!IsZero(in.fin[tuple.state].w)) {      // ProcessFinal (:169-201): the super-final state (weight One, no arcs)
      if (tuple.c.IsEmpty()) out.fin[ostate] = PlusCW(out.fin[ostate], CW{tuple.c.weight, {}});
      else { Tuple t2 = tuple; OutputArcForce(t2.c, ctx, &arc); arc.next = state_for(t2); out.arcs[ostate].push_back(std::move(arc)); }
    }
    for (const WArc &a : in.arcs[tuple.state]) {      // read one input arc: its labels join the computation state, its weight goes out on an epsilon arc (Advance, :38-47)
      Tuple nt = tuple; nt.c.tids.insert(nt.c.tids.end(), a.w.str.begin(), a.w.str.end()); if (a.label != 0) nt.c.words.push_back(a.label);
      const LW w = TimesLW(nt.c.weight, a.w.w); nt.c.weight = LW(); nt.state = a.next;
      const int32_t no = state_for(nt);
      out.arcs[ostate].push_back(WArc{0, no, CW{w, {}}});
    }
  }
  RmEpsilonAndConnect(&out);
  for (int32_t s = 0; s < out.NumStates(); s++) for (WArc &a : out.arcs[s]) {      // RemoveSomeInputSymbols + Project (:292-306)
    if (info_in.partial_word_label == 0 && a.label == info.partial_word_label) a.label = 0;
    if (info_in.silence_label == 0 && a.label == info.silence_label) a.label = 0;
  }
  lat_out->start = out.start;
  for (int32_t s = 0; s < out.NumStates(); s++) {
    lat_out->AddState();
    if (!IsZero(out.fin[s].w)) {
      lat_out->is_final[s] = 1;
      lat_out->fin_graph[s] = out.fin[s].w.g;
      lat_out->fin_ac[s] = out.fin[s].w.a;
      lat_out->fin_str[s] = out.fin[s].str;
    }
  }
  for (int32_t s = 0; s < out.NumStates(); s++) for (WArc &a : out.arcs[s]) {
    lat_out->arc_src.push_back(s);
    lat_out->arc_dst.push_back(a.next);
    lat_out->arc_label.push_back(a.label);
    lat_out->arc_graph.push_back(a.w.w.g);
    lat_out->arc_ac.push_back(a.w.w.a);
    lat_out->arc_str.push_back(std::move(a.w.str));
  }
  return ok && !error;
}

// ---- LatticePostprocessor -----------------------------------------------------------------------------------------------------
void LatticePostprocessorConfig::Register(ParseOptions *po) {
  po->Register("max-expand", &max_expand, "If >0, the maximum amount by which this program will expand lattices before refusing to continue.");
  po->Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods"); po->Register("lm-scale", &lm_scale, "Scaling factor for graph/lm costs");
  po->Register("acoustic2lm-scale", &acoustic2lm_scale, "Add this times original acoustic costs to LM costs");
  po->Register("lm2acoustic-scale", &lm2acoustic_scale, "Add this times original LM costs to acoustic costs");
  po->Register("word-ins-penalty", &word_ins_penalty, "Word insertion penalty."); po->Register("word-boundary-rxfilename", &word_boundary_rxfilename, "Word boundary file");
  po->Register("decode-mbr", &mbr_opts.decode_mbr, "If true, do Minimum Bayes Risk decoding (else, Maximum a Posteriori)");
  po->Register("print-silence", &mbr_opts.print_silence, "Keep the inter-word '<eps>' bins in the 1-best output (ctm, <eps> can be a 'silence' or a 'deleted' word)");
  po->Register("silence-label", &silence_label, "Numeric id of word symbol that is to be used for silence arcs in the word-aligned lattice (zero is OK)");
  po->Register("partial-word-label", &partial_word_label,
      "Numeric id of word symbol that is to be used for arcs in the word-aligned lattice corresponding to partial words at the end of forced-out utterances (zero is OK)");
  po->Register("reorder", &reorder, "True if the lattices were generated from graphs that had the --reorder option true");
}
LatticePostprocessor::LatticePostprocessor(const LatticePostprocessorConfig &config) : config_(config) {
  use_lattice_scale_ = config_.lm_scale != 1.0f || config_.acoustic2lm_scale != 0.0f || config_.lm2acoustic_scale != 0.0f || config_.acoustic_scale != 1.0f;
  if (!config_.word_boundary_rxfilename.empty())      // LoadWordBoundaryInfo (lattice-postprocessor.h:100-103)
    word_info_ = std::make_shared<WordBoundaryInfo>(ReadWordBoundaryInfo(config_.word_boundary_rxfilename, config_.reorder, config_.silence_label, config_.partial_word_label));
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
  if (word_info_) {      // :66-85: the return value is ignored as in the reference (a warning has been printed; it happens with end-pointing)
    if (!tmodel_) K3H_ERR << "SetTransitionInformation() must be called (typically by pipeline)";
    const int32_t max_states = config_.max_expand > 0 ? (int32_t)(1000 + config_.max_expand * clat.NumStates()) : 0;
    WordAlignLattice(clat, *tmodel_, *word_info_, max_states, out);
  } else *out = clat;
  return true;
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
