// k3_lattice.cc -- host-side lattice post-processing above the decoder's raw lattices: word-level pruned determinization to a
// CompactLattice, PruneLattice, topological sorting, the CompactLattice writers and a reader for state-level lattice tables.
//
// This restates the algorithm of the reference's lat/determinize-lattice-pruned.cc (LatticeDeterminizerPruned, cited per function)
// in this library's own data structures: a CSR input automaton, a string trie addressed by int32 ids instead of Entry pointers,
// std containers for the subset hashes.  PINNED to the reference's own source: OpenFst is not vendored in /root/reference, but
// lat/determinize-lattice-pruned.cc compiles unmodified against a stand-in for the part of OpenFst it touches (third_party/minifst
// -> oracle/_ref/bin/ref-lattice-determinize), and the programs built on this file print the same CompactLattices as that binary,
// character for character, on random lattices, exact cost ties, decoder lattices and the --max-mem prune-and-retry path, for the
// word-level and the phone+word entry points (tests/test_lattice_det.py; digests of the reference output in tests/golden/).
// The property tests of the same file (deterministic on word labels, per word sequence the cost and alignment of the best raw path,
// every word sequence within the beam kept) stay: they are what the reference's determinize-lattice-pruned-test.cc checks.
//
// Both entry points of the reference are here: DeterminizeLatticePruned (one word-level pass; lattice-determinize-pruned) and
// DeterminizeLatticePhonePruned (phone-level pass first, then the word-level pass; what the decoders and
// lattice-determinize-phone-pruned call), with --minimize (PushCompactLatticeStrings / Weights + MinimizeCompactLattice).
// --word-determinize=false gives the first pass's result (or, with both passes off, the lattice itself) re-packed by ConvertLattice.
#include "k3_host.h"
#include <sstream>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <queue>
#include <unordered_map>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace k3host {
namespace {
const float kInfF = std::numeric_limits<float>::infinity();
const double kInfD = std::numeric_limits<double>::infinity();

// ---- LatticeWeightTpl<float> (fstext/lattice-weight.h): value1 = graph cost, value2 = acoustic cost -----------------------------
struct LatW { float g, a; };
inline LatW One() { return {0.0f, 0.0f}; }
inline LatW Zero() { return {kInfF, kInfF}; }
inline bool operator==(LatW x, LatW y) { return x.g == y.g && x.a == y.a; }
inline bool operator!=(LatW x, LatW y) { return !(x == y); }
inline LatW Times(LatW x, LatW y) { return {x.g + y.g, x.a + y.a}; }
inline int Compare(LatW x, LatW y) {           // :295-308: +1 when x is the better (cheaper) one; ties broken on the graph part
  const float fx = x.g + x.a, fy = y.g + y.a;
  if (fx < fy) return 1;
  if (fx > fy) return -1;
  if (x.g < y.g) return 1;
  if (x.g > y.g) return -1;
  return 0;
}
inline LatW Plus(LatW x, LatW y) { return Compare(x, y) >= 0 ? x : y; }      // :312-315
inline LatW Divide(LatW x, LatW y) {           // :371-386
  const float g = x.g - y.g, a = x.a - y.a;
  if (g != g || a != a || g == -kInfF || a == -kInfF) { K3H_WARN << "LatticeWeight Divide: NaN or invalid number produced, returning zero"; return Zero(); }
  if (g == kInfF || a == kInfF) return Zero();
  return {g, a};
}
inline bool ApproxEqual(LatW x, LatW y, float delta) {    // :390-395
  if (x.g == y.g && x.a == y.a) return true;
  return std::fabs((x.g + x.a) - (y.g + y.a)) <= delta;
}
inline double Cost(LatW w) { return (double)w.g + (double)w.a; }   // ConvertToCost :846

// ---- the determinizer's input: topologically sorted, arcs sorted on the word label, word = input side ----------------------------
struct InputFst {
  int32_t start = -1;
  std::vector<int32_t> off, word, tid, next; std::vector<LatW> w, fin;
  int32_t NumStates() const { return (int32_t)fin.size(); }
};

// depth-first topological order like fst::TopSort: the search starts at the start state, then at every state not reached yet in
// numeric order (states that cannot be reached from the start state stay in the automaton, as in the reference); arcs are visited in
// stored order; new numbering = reverse finishing order.  false on a cycle.  order[i] = old state id of new state i.
bool TopOrder(int32_t n, int32_t start, const std::vector<int32_t> &off, const std::vector<int32_t> &next, std::vector<int32_t> *order) {
  order->clear();
  if (start < 0) return true;
  std::vector<char> color(n, 0); std::vector<int32_t> stack, pos(n, 0), finish;
  auto visit = [&](int32_t root) {
    stack.push_back(root); color[root] = 1; pos[root] = off[root];
    while (!stack.empty()) {
      const int32_t s = stack.back();
      if (pos[s] < off[s + 1]) {
        const int32_t d = next[pos[s]++];
        if (color[d] == 1) return false;
        if (color[d] == 0) { color[d] = 1; pos[d] = off[d]; stack.push_back(d); }
      } else { color[s] = 2; finish.push_back(s); stack.pop_back(); }
    }
    return true;
  };
  if (!visit(start)) return false;
  for (int32_t s = 0; s < n; s++) if (color[s] == 0 && !visit(s)) return false;
  order->assign(finish.rbegin(), finish.rend());
  return true;
}

// mutable edge-list form of the same automaton (word = input side, transition-id = output side), for the steps that add states
struct EdgeFst {
  int32_t start = -1; std::vector<LatW> fin; std::vector<int32_t> src, dst, word, tid; std::vector<LatW> w;
  int32_t AddState() { fin.push_back(Zero()); return (int32_t)fin.size() - 1; }
  void AddArc(int32_t s, int32_t d, int32_t wd, int32_t t, LatW wt) { src.push_back(s); dst.push_back(d); word.push_back(wd); tid.push_back(t); w.push_back(wt); }
};

// Invert (lattice-determinize-pruned.cc:104).  The lattice is NOT trimmed first: states that lead nowhere still count as members of
// the subsets (they change which weight / string is common to a subset), exactly as in the reference.  The decoders trim before
// they determinize (decoder-wrappers.cc:353), and so do the programs here.
EdgeFst InvertedEdges(const Lattice &lat) {
  EdgeFst e; e.start = lat.NumStates() ? lat.start : -1;
  for (int32_t s = 0; s < lat.NumStates(); s++) e.fin.push_back(std::isfinite(lat.st_final[s]) ?
      LatW{lat.st_final[s], lat.st_final_ac.empty() ? 0.0f : lat.st_final_ac[s]} : Zero());
  for (size_t a = 0; a < lat.arc_src.size(); a++) e.AddArc(lat.arc_src[a], lat.arc_dst[a], lat.arc_olabel[a], lat.arc_ilabel[a], {lat.arc_graph[a], lat.arc_ac[a]});
  return e;
}

// TopSort + ArcSort(ILabelCompare) (lattice-determinize-pruned.cc:106-112): states reachable from the start state in depth-first
// topological order, the arcs of a state sorted on (word label, transition-id) (std::sort with OpenFst's ILabelCompare; sort_arcs = false: TopSort only)
InputFst SortedInput(const EdgeFst &e, bool sort_arcs = true) {
  InputFst f; const int32_t n = (int32_t)e.fin.size(); const size_t na = e.src.size();
  if (n == 0 || e.start < 0) return f;
  std::vector<int32_t> off(n + 1, 0), idx(na), nx(na);
  for (size_t a = 0; a < na; a++) off[e.src[a] + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < na; a++) { const int32_t k = p[e.src[a]]++; idx[k] = (int32_t)a; nx[k] = e.dst[a]; } }
  std::vector<int32_t> order;
  if (!TopOrder(n, e.start, off, nx, &order)) K3H_ERR << "Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles).";
  std::vector<int32_t> newid(n, -1); for (size_t i = 0; i < order.size(); i++) newid[order[i]] = (int32_t)i;
  const int32_t m = (int32_t)order.size();
  f.start = newid[e.start]; f.fin.resize(m); f.off.assign(m + 1, 0);
  for (int32_t i = 0; i < m; i++) {
    const int32_t s = order[i];
    f.fin[i] = e.fin[s];
    std::vector<int32_t> arcs(idx.begin() + off[s], idx.begin() + off[s + 1]);
    // (OpenFst's ILabelCompare orders on the pair (input label, output label): arcs with the same word are ordered by their transition-id)
    if (sort_arcs) std::sort(arcs.begin(), arcs.end(), [&](int32_t x, int32_t y) { return e.word[x] != e.word[y] ? e.word[x] < e.word[y] : e.tid[x] < e.tid[y]; });
    for (int32_t a : arcs) { f.word.push_back(e.word[a]); f.tid.push_back(e.tid[a]); f.next.push_back(newid[e.dst[a]]); f.w.push_back(e.w[a]); }
    f.off[i + 1] = (int32_t)f.word.size();
  }
  return f;
}

// kaldi::PruneLattice (lat/lattice-functions.cc:233-318) on the prepared automaton; state order (hence sortedness) is kept
bool PruneInput(double beam, InputFst *f) {
  const int32_t n = f->NumStates();
  if (n == 0) return false;
  std::vector<double> fwd(n, kInfD); fwd[f->start] = 0.0; double best_final = kInfD;
  for (int32_t s = 0; s < n; s++) {
    for (int32_t k = f->off[s]; k < f->off[s + 1]; k++) { const double c = fwd[s] + Cost(f->w[k]); if (fwd[f->next[k]] > c) fwd[f->next[k]] = c; }
    const double fc = fwd[s] + Cost(f->fin[s]); if (fc < best_final) best_final = fc;
  }
  const double cutoff = best_final + beam;
  std::vector<double> bwd(n, kInfD); std::vector<char> keep_arc(f->word.size(), 0);
  for (int32_t s = n - 1; s >= 0; s--) {
    double b = Cost(f->fin[s]);
    if (b + fwd[s] > cutoff && b != kInfD) f->fin[s] = Zero();
    for (int32_t k = f->off[s]; k < f->off[s + 1]; k++) {
      const double ab = Cost(f->w[k]) + bwd[f->next[k]];
      if (ab < b) b = ab;
      keep_arc[k] = !(fwd[s] + ab > cutoff);
    }
    bwd[s] = b;
  }
  // trim: accessible through kept arcs and co-accessible to a final weight that is still there
  std::vector<char> acc(n, 0), co(n, 0); acc[f->start] = 1;
  for (int32_t s = 0; s < n; s++) if (acc[s]) for (int32_t k = f->off[s]; k < f->off[s + 1]; k++) if (keep_arc[k]) acc[f->next[k]] = 1;
  for (int32_t s = n - 1; s >= 0; s--) {
    if (f->fin[s] != Zero()) co[s] = 1;
    for (int32_t k = f->off[s]; k < f->off[s + 1] && !co[s]; k++) if (keep_arc[k] && co[f->next[k]]) co[s] = 1;
  }
  std::vector<int32_t> newid(n, -1); int32_t m = 0; for (int32_t s = 0; s < n; s++) if (acc[s] && co[s]) newid[s] = m++;
  InputFst o; o.start = newid[f->start]; o.fin.resize(m); o.off.assign(m + 1, 0);
  for (int32_t s = 0; s < n; s++) {
    if (newid[s] < 0) continue;
    o.fin[newid[s]] = f->fin[s];
    for (int32_t k = f->off[s]; k < f->off[s + 1]; k++) if (keep_arc[k] && newid[f->next[k]] >= 0) {
      o.word.push_back(f->word[k]);
      o.tid.push_back(f->tid[k]);
      o.next.push_back(newid[f->next[k]]);
      o.w.push_back(f->w[k]);
    }
    o.off[newid[s] + 1] = (int32_t)o.word.size();
  }
  if (o.start < 0) o = InputFst();
  *f = std::move(o);
  return true;
}

// ---- LatticeStringRepository (fstext/determinize-lattice.h): strings of transition-ids as nodes of a trie; id 0 is the empty string.
// Equal sequences always get the same id, which is what lets subsets be hashed and compared on (state, string id).
// uint64 -> int32 hash table with linear probing (the trie's child index and its memo of prefix removals: millions of look-ups per lattice)
class FlatMap64 {
 public:
  FlatMap64() : keys_(1024), vals_(1024, -1), mask_(1023) {}
  int32_t *Find(uint64_t key) { for (size_t i = Hash(key) & mask_;; i = (i + 1) & mask_) { if (vals_[i] < 0) return nullptr; if (keys_[i] == key) return &vals_[i]; } }
  void Insert(uint64_t key, int32_t val) {           // key must not be present; val >= 0
    if (2 * (count_ + 1) > keys_.size()) Grow();
    for (size_t i = Hash(key) & mask_;; i = (i + 1) & mask_) if (vals_[i] < 0) { keys_[i] = key; vals_[i] = val; count_++; return; }
  }
  void Clear() { std::fill(vals_.begin(), vals_.end(), -1); count_ = 0; }
 private:
  static size_t Hash(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; return (size_t)k; }
  void Grow() {
    std::vector<uint64_t> ok; std::vector<int32_t> ov; ok.swap(keys_); ov.swap(vals_);
    keys_.assign(2 * ok.size(), 0); vals_.assign(2 * ov.size(), -1); mask_ = keys_.size() - 1; count_ = 0;
    for (size_t i = 0; i < ok.size(); i++) if (ov[i] >= 0) Insert(ok[i], ov[i]);
  }
  std::vector<uint64_t> keys_; std::vector<int32_t> vals_; size_t mask_, count_ = 0;
};

class StringTrie {
 public:
  StringTrie() : parent_(1, -1), label_(1, 0), depth_(1, 0), held_(1, 1) {}
  int32_t Successor(int32_t s, int32_t label) {
    const uint64_t key = ((uint64_t)(uint32_t)s << 32) | (uint32_t)label;
    if (int32_t *hit = succ_.Find(key)) { if (!held_[*hit]) { held_[*hit] = 1; num_held_++; } return *hit; }
    const int32_t id = (int32_t)parent_.size();
    parent_.push_back(s); label_.push_back(label); depth_.push_back(depth_[s] + 1); held_.push_back(1); num_held_++; succ_.Insert(key, id);
    return id;
  }
  // The reference frees the entries nobody refers to when memory runs short (LatticeStringRepository::Rebuild) and creates them
  // again on demand; ids here stay valid for ever, so only the book-keeping is mirrored: which entries the reference would hold now.
  size_t NumHeld() const { return num_held_; }
  void KeepOnly(const std::vector<char> &needed) { num_held_ = 0; for (size_t i = 1; i < held_.size(); i++) { held_[i] = needed[i]; num_held_ += needed[i]; } removed_.Clear(); }
  void ToVector(int32_t s, std::vector<int32_t> *v) const { v->resize(depth_[s]); for (int32_t i = depth_[s] - 1; i >= 0; i--, s = parent_[s]) (*v)[i] = label_[s]; }
  int32_t FromVector(const std::vector<int32_t> &v, size_t from = 0) { int32_t s = 0; for (size_t i = from; i < v.size(); i++) s = Successor(s, v[i]); return s; }
  int32_t Concatenate(int32_t a, int32_t b) {
    if (b == 0) return a;
    if (a == 0) return b;
    std::vector<int32_t> v;
    ToVector(b, &v);
    for (int32_t l : v) a = Successor(a, l);
    return a;
  }
  void ReduceToCommonPrefix(int32_t s, std::vector<int32_t> *prefix) const {
    while (depth_[s] > (int32_t)prefix->size()) s = parent_[s];
    prefix->resize(depth_[s]);
    for (int32_t i = depth_[s] - 1; i >= 0; i--, s = parent_[s]) if (label_[s] != (*prefix)[i]) prefix->resize(i);
  }
  // the string without its first n labels.  RemovePrefix(s, n) = Successor(RemovePrefix(parent(s), n), label(s)): results are remembered
  // per (node, n), so strings that share an ancestor share the work.  (The memo is dropped whenever KeepOnly changes what is held: a
  // remembered answer skips the Successor calls that would mark its entries as held again.)
  int32_t RemovePrefix(int32_t s, size_t n) {
    if (n == 0) return s;
    chain_.clear(); int32_t base = 0;
    for (int32_t cur = s; depth_[cur] > (int32_t)n; cur = parent_[cur]) {
      if (int32_t *hit = removed_.Find(((uint64_t)(uint32_t)cur << 32) | (uint32_t)n)) { base = *hit; break; }
      chain_.push_back(cur);
    }
    for (size_t i = chain_.size(); i-- > 0;) { base = Successor(base, label_[chain_[i]]); removed_.Insert(((uint64_t)(uint32_t)chain_[i] << 32) | (uint32_t)n, base); }
    return base;
  }
  int32_t CommonPrefix(int32_t a, int32_t b) const {
    while (depth_[a] > depth_[b]) a = parent_[a];
    while (depth_[b] > depth_[a]) b = parent_[b];
    while (a != b) { a = parent_[a]; b = parent_[b]; }
    return a;
  }
  int32_t Depth(int32_t s) const { return depth_[s]; }
  int32_t Parent(int32_t s) const { return parent_[s]; }
  size_t NumEntries() const { return parent_.size() - 1; }
 private:
  std::vector<int32_t> parent_, label_, depth_, chain_; std::vector<char> held_; size_t num_held_ = 0;
  FlatMap64 succ_, removed_;
};

// ---- LatticeDeterminizerPruned (lat/determinize-lattice-pruned.cc:47-1190) -------------------------------------------------------
class Determinizer {
 public:
  Determinizer(const InputFst &f, double beam, const DeterminizeLatticePrunedOptions &opts) : f_(f), beam_(beam), opts_(opts),
      minimal_hash_(16, SubsetHash(), SubsetEqual{opts.delta}), initial_hash_(16, SubsetHash(), SubsetEqual{opts.delta}) {}

  bool Determinize(double *effective_beam) {                 // :329-375
    Initialize();
    while (!queue_.empty()) {
      const size_t num_states = out_.size();
      if ((opts_.max_states > 0 && (int64_t)num_states > opts_.max_states) || (opts_.max_arcs > 0 && num_arcs_ > opts_.max_arcs) ||
          (num_states % 10 == 0 && !CheckMemoryUsage())) break;
      Task *task = queue_.top(); queue_.pop();
      ProcessTransition(task->state, task->label, &task->subset);
      delete task;
    }
    *effective_beam = queue_.empty() ? beam_ : queue_.top()->priority_cost - backward_[f_.start];
    const bool done = queue_.empty();
    while (!queue_.empty()) { delete queue_.top(); queue_.pop(); }
    return done;
  }

  // :112-186: the same automaton as an ordinary FST; a string of k transition-ids becomes a chain of k arcs (weight and word on the
  // first), a final weight with a string a chain into a new final state
  void Output(EdgeFst *o) const {
    *o = EdgeFst();
    if (out_.empty()) return;
    for (size_t s = 0; s < out_.size(); s++) o->AddState();
    o->start = 0;
    std::vector<int32_t> seq;
    for (size_t s = 0; s < out_.size(); s++)
      for (const TempArc &t : out_[s].arcs) {
        trie_.ToVector(t.string, &seq);
        int32_t cur = (int32_t)s;
        if (t.next < 0) {
          for (size_t i = 0; i < seq.size(); i++) { const int32_t nx = o->AddState(); o->AddArc(cur, nx, 0, seq[i], i == 0 ? t.w : One()); cur = nx; }
          o->fin[cur] = seq.empty() ? t.w : One();
        } else {
          for (size_t i = 0; i + 1 < seq.size(); i++) { const int32_t nx = o->AddState(); o->AddArc(cur, nx, i == 0 ? t.label : 0, seq[i], i == 0 ? t.w : One()); cur = nx; }
          o->AddArc(cur, t.next, seq.size() <= 1 ? t.label : 0, seq.empty() ? 0 : seq.back(), seq.size() <= 1 ? t.w : One());
        }
      }
  }

  void Output(CompactLattice *o) const {                     // :61-107: one output state per determinized state, final weights from the kNoState arcs
    *o = CompactLattice();
    if (out_.empty()) return;
    for (size_t s = 0; s < out_.size(); s++) o->AddState();
    o->start = 0;
    std::vector<int32_t> str;
    for (size_t s = 0; s < out_.size(); s++)
      for (const TempArc &t : out_[s].arcs) {
        trie_.ToVector(t.string, &str);
        if (t.next < 0) { o->is_final[s] = 1; o->fin_graph[s] = t.w.g; o->fin_ac[s] = t.w.a; o->fin_str[s] = str; }
        else {
          o->arc_src.push_back((int32_t)s);
          o->arc_dst.push_back(t.next);
          o->arc_label.push_back(t.label);
          o->arc_graph.push_back(t.w.g);
          o->arc_ac.push_back(t.w.a);
          o->arc_str.push_back(str);
        }
      }
  }

 private:
  struct Element {                                           // :390-409; `state` is an input state except inside initial_hash_ values
    int32_t state, string; LatW w;
    bool operator!=(const Element &o) const { return state != o.state || string != o.string || w != o.w; }
  };
  struct TempArc { int32_t label, string, next; LatW w; };    // :413-418; next < 0: really a final weight
  struct OutputState { std::vector<Element> minimal_subset; std::vector<TempArc> arcs; double forward_cost; };
  struct Task { int32_t state, label; std::vector<Element> subset; double priority_cost; };
  struct TaskCompare { bool operator()(const Task *a, const Task *b) const { return a->priority_cost > b->priority_cost; } };     // :1156-1162: cheapest first
  struct SubsetHash {                                        // :432-443: state and string only, not the weight
    size_t operator()(const std::vector<Element> *v) const {
      size_t h = 0, factor = 1;
      for (const Element &e : *v) {
        h *= factor;
        h += (size_t)e.state + 103333u * (size_t)e.string;
        factor *= 23531;
      }
      return h;
    }
  };
  struct SubsetEqual {                                       // :447-463: exact on state and string, within delta on the weight
    float delta;
    bool operator()(const std::vector<Element> *a, const std::vector<Element> *b) const {
      if (a->size() != b->size()) return false;
      for (size_t i = 0; i < a->size(); i++) if ((*a)[i].state != (*b)[i].state || (*a)[i].string != (*b)[i].string || !ApproxEqual((*a)[i].w, (*b)[i].w, delta)) return false;
      return true;
    }
  };

  // order on (weight, string) pairs of CompactLatticeWeight (:601-625): the better weight wins; on a tie the SHORTER string is the
  // greater one (opposite order on lengths), then the lexicographically larger
  int Compare(LatW aw, int32_t as, LatW bw, int32_t bs) const {
    const int wc = k3host::Compare(aw, bw);
    if (wc != 0) return wc;
    if (as == bs) return 0;
    const int32_t al = trie_.Depth(as), bl = trie_.Depth(bs);
    if (al > bl) return -1;
    if (al < bl) return 1;
    std::vector<int32_t> av, bv; trie_.ToVector(as, &av); trie_.ToVector(bs, &bv);
    for (int32_t i = 0; i < al; i++) { if (av[i] < bv[i]) return -1; if (av[i] > bv[i]) return 1; }
    return 0;
  }

  bool IsIsymbolOrFinal(int32_t s) const {                   // :1018-1042 (no cache: the arcs are sorted, one look at the last arc decides)
    if (f_.fin[s] != Zero()) return true;
    for (int32_t k = f_.off[s + 1] - 1; k >= f_.off[s] && f_.word[k] != 0; k--) if (f_.w[k] != Zero()) return true;
    return false;
  }
  void ConvertToMinimal(std::vector<Element> *subset) const {  // :501-513
    size_t o = 0; for (const Element &e : *subset) if (IsIsymbolOrFinal(e.state)) (*subset)[o++] = e;
    subset->resize(o);
  }

  void EpsilonClosure(std::vector<Element> *subset) {        // :632-724: best (weight, string) per state reachable over word-epsilon arcs
    struct ByState { bool operator()(const Element &a, const Element &b) const { return a.state > b.state; } };
    std::priority_queue<Element, std::vector<Element>, ByState> queue;
    // the reference's map state -> Element, as a slot table over the input states (slots are handed back at the end)
    if (slot_.size() != (size_t)f_.NumStates()) slot_.assign(f_.NumStates(), -1);
    std::vector<Element> &cur = closure_; cur.clear();
    auto find = [&](int32_t state) -> Element * { return slot_[state] < 0 ? nullptr : &cur[slot_[state]]; };
    auto insert = [&](const Element &e) { slot_[e.state] = (int32_t)cur.size(); cur.push_back(e); };
    for (const Element &e : *subset) { queue.push(e); if (Element *p = find(e.state)) *p = e; else insert(e); }
    bool replaced = false; int32_t counter = 0;
    while (!queue.empty()) {
      const Element elem = queue.top(); queue.pop();
      if (replaced && *find(elem.state) != elem) continue;     // a stale copy of an element that was improved later
      if (opts_.max_loop > 0 && counter++ > opts_.max_loop) K3H_ERR << "Lattice determinization aborted since looped more than " << opts_.max_loop <<
          " times during epsilon closure.";
      for (int32_t k = f_.off[elem.state]; k < f_.off[elem.state + 1] && f_.word[k] == 0; k++) {
        if (f_.w[k] == Zero()) continue;
        Element nx; nx.state = f_.next[k]; nx.w = Times(elem.w, f_.w[k]); nx.string = -1;
        auto string_of = [&]() { return f_.tid[k] == 0 ? elem.string : trie_.Successor(elem.string, f_.tid[k]); };
        Element *old = find(nx.state);
        if (old == nullptr) { nx.string = string_of(); insert(nx); queue.push(nx); continue; }
        int comp = k3host::Compare(nx.w, old->w);
        if (comp == 0) { nx.string = string_of(); comp = Compare(nx.w, nx.string, old->w, old->string); }
        if (comp == 1) { if (nx.string < 0) nx.string = string_of(); old->string = nx.string; old->w = nx.w; queue.push(nx); replaced = true; }
      }
    }
    for (const Element &e : cur) slot_[e.state] = -1;
    subset->assign(cur.begin(), cur.end());
    std::sort(subset->begin(), subset->end(), [](const Element &a, const Element &b) { return a.state < b.state; });
  }

  void ProcessFinal(int32_t id) {                            // :731-769
    OutputState &st = out_[id];
    int32_t fstr = 0; LatW fw = Zero(); bool is_final = false;
    for (const Element &e : st.minimal_subset) {
      const LatW w = Times(e.w, f_.fin[e.state]);
      if (w != Zero() && (!is_final || Compare(w, e.string, fw, fstr) == 1)) { is_final = true; fw = w; fstr = e.string; }
    }
    if (is_final && Cost(fw) + st.forward_cost <= cutoff_) { st.arcs.push_back({0, fstr, -1, fw}); num_arcs_++; }
  }

  void NormalizeSubset(std::vector<Element> *elems, LatW *tot, int32_t *common) {    // :774-803
    if (elems->empty()) { K3H_WARN << "empty subset"; *common = 0; *tot = Zero(); return; }
    // the longest common prefix of the strings is their lowest common ancestor in the trie (a node, no vectors needed; it and its
    // ancestors are prefixes of strings in use, so the reference's ConvertFromVector would find them all in place)
    int32_t prefix = (*elems)[0].string;
    LatW w = (*elems)[0].w;
    for (size_t i = 1; i < elems->size(); i++) { w = Plus(w, (*elems)[i].w); prefix = trie_.CommonPrefix(prefix, (*elems)[i].string); }
    const size_t prefix_len = (size_t)trie_.Depth(prefix);
    for (Element &e : *elems) { e.w = Divide(e.w, w); e.string = trie_.RemovePrefix(e.string, prefix_len); }
    *common = prefix; *tot = w;
  }

  void MakeSubsetUnique(std::vector<Element> *subset) const {  // :808-835: sorted on state; keep the best element per state
    size_t o = 0;
    for (size_t i = 0; i < subset->size();) {
      Element best = (*subset)[i++];
      for (; i < subset->size() && (*subset)[i].state == best.state; i++)
        if (Compare((*subset)[i].w, (*subset)[i].string, best.w, best.string) == 1) { best.string = (*subset)[i].string; best.w = (*subset)[i].w; }
      (*subset)[o++] = best;
    }
    subset->resize(o);
  }

  int32_t MinimalToStateId(const std::vector<Element> &subset, double forward_cost) {   // :520-547
    auto it = minimal_hash_.find(&subset);
    if (it != minimal_hash_.end()) {
      if (forward_cost < out_[it->second].forward_cost - 0.1) K3H_WARN << "New cost is less (check the difference is small) " << forward_cost << ", " <<
          out_[it->second].forward_cost;
      return it->second;
    }
    const int32_t id = (int32_t)out_.size();
    out_.push_back(OutputState{subset, {}, forward_cost});
    minimal_hash_[&out_.back().minimal_subset] = id;
    num_elems_ += (int64_t)subset.size();
    ProcessFinal(id);
    ProcessTransitions(id);
    return id;
  }

  // a normalized subset before epsilon closure -> its determinized state plus what normalisation after the closure took out (:552-593)
  int32_t InitialToStateId(const std::vector<Element> &subset_in, double forward_cost, LatW *remaining, int32_t *common) {
    auto it = initial_hash_.find(&subset_in);
    if (it != initial_hash_.end()) { *remaining = it->second.w; *common = it->second.string; return it->second.state; }
    std::vector<Element> subset(subset_in);
    EpsilonClosure(&subset);
    ConvertToMinimal(&subset);
    Element elem;
    NormalizeSubset(&subset, &elem.w, &elem.string);
    forward_cost += Cost(elem.w);
    elem.state = MinimalToStateId(subset, forward_cost);
    *remaining = elem.w; *common = elem.string;
    if (elem.w == Zero()) K3H_WARN << "Zero weight!";
    initial_keys_.push_back(subset_in);
    initial_hash_[&initial_keys_.back()] = elem;
    num_elems_ += (int64_t)subset_in.size();
    return elem.state;
  }

  void ProcessTransition(int32_t ostate, int32_t label, std::vector<Element> *subset) {   // :845-874
    double forward_cost = out_[ostate].forward_cost;
    int32_t common; LatW tot;
    NormalizeSubset(subset, &tot, &common);
    forward_cost += Cost(tot);
    LatW next_tot; int32_t next_common;
    const int32_t next = InitialToStateId(*subset, forward_cost, &next_tot, &next_common);
    out_[ostate].arcs.push_back({label, trie_.Concatenate(common, next_common), next, Times(tot, next_tot)});
    num_arcs_++;
  }

  void ProcessTransitions(int32_t id) {                      // :903-1014: one queued task per word label leaving the subset
    std::vector<std::pair<int32_t, Element>> all;
    for (const Element &e : out_[id].minimal_subset)
      for (int32_t k = f_.off[e.state]; k < f_.off[e.state + 1]; k++) {
        if (f_.word[k] == 0 || f_.w[k] == Zero()) continue;
        all.push_back({f_.word[k], Element{f_.next[k], f_.tid[k] == 0 ? e.string : trie_.Successor(e.string, f_.tid[k]), Times(e.w, f_.w[k])}});
      }
    std::sort(all.begin(), all.end(), [](const std::pair<int32_t, Element> &a, const std::pair<int32_t,
        Element> &b) { return a.first != b.first ? a.first < b.first : a.second.state < b.second.state; });
    const double forward_cost = out_[id].forward_cost;
    for (size_t i = 0; i < all.size();) {
      Task *task = new Task; task->state = id; task->label = all[i].first; task->priority_cost = kInfD;
      for (; i < all.size() && all[i].first == task->label; i++) {
        task->subset.push_back(all[i].second);
        task->priority_cost = std::min(task->priority_cost, Cost(all[i].second.w) + backward_[all[i].second.state]);
      }
      task->priority_cost += forward_cost;
      if (task->priority_cost > cutoff_) { delete task; continue; }
      MakeSubsetUnique(&task->subset);
      queue_.push(task);
      const double best = backward_[f_.start];
      if (task->priority_cost < best - (0.01 + 1.0e-04 * std::fabs(best))) K3H_WARN << "Cost below best cost was encountered:" << task->priority_cost << " < " << best;
    }
  }

  void Initialize() {                                        // ComputeBackwardWeight :1044-1068 + InitializeDeterminization :1070-1118
    const int32_t n = f_.NumStates();
    backward_.assign(n, kInfD);
    for (int32_t s = n - 1; s >= 0; s--) {
      double c = Cost(f_.fin[s]);
      for (int32_t k = f_.off[s]; k < f_.off[s + 1]; k++) c = std::min(c, Cost(f_.w[k]) + backward_[f_.next[k]]);
      backward_[s] = c;
    }
    if (f_.start < 0) return;
    if (backward_[f_.start] == kInfD) K3H_WARN << "Total weight of input lattice is zero.";
    cutoff_ = backward_[f_.start] + beam_;
    // the start state's subset is not normalized: there is nowhere to put a common weight / string in front of it
    std::vector<Element> subset(1, Element{f_.start, 0, One()});
    EpsilonClosure(&subset);
    ConvertToMinimal(&subset);
    out_.push_back(OutputState{subset, {}, 0.0});
    num_elems_ += (int64_t)subset.size();
    minimal_hash_[&out_.back().minimal_subset] = 0;
    ProcessFinal(0);
    ProcessTransitions(0);
  }

  // :286-327.  The reference measures (2 x 16 B per trie entry) + 32 B per arc + 24 B per subset element and, above --max-mem,
  // rebuilds the trie from the strings still referenced (:218-264).  Here nothing needs freeing; the trie only notes which entries
  // the reference would be holding, so that the early exit happens at the same point.
  bool CheckMemoryUsage() {
    if (opts_.max_mem <= 0) return true;
    const int64_t arcs = num_arcs_ * 32, elems = num_elems_ * 24;        // sizeof(TempArc) = 32 (4 + pad + 8 + 4 + 8, 8-aligned), sizeof(Element) = 24
    const int64_t repo = (int64_t)trie_.NumHeld() * 32;
    if (repo + arcs + elems <= opts_.max_mem) return true;
    std::vector<char> live(trie_.NumEntries() + 1, 0);
    auto mark = [&](int32_t s) { for (; s > 0 && !live[s]; s = trie_.Parent(s)) live[s] = 1; };
    for (const OutputState &st : out_) { for (const Element &e : st.minimal_subset) mark(e.string); for (const TempArc &t : st.arcs) mark(t.string); }
    for (const auto &kv : initial_hash_) { for (const Element &e : *kv.first) mark(e.string); mark(kv.second.string); }
    {
      std::vector<Task *> tasks;
      while (!queue_.empty()) {
        tasks.push_back(queue_.top());
        queue_.pop();
      }
      for (Task *t : tasks) {
        for (const Element &e : t->subset) mark(e.string);
        queue_.push(t);
      }
    }
    trie_.KeepOnly(live);
    const int64_t new_repo = (int64_t)trie_.NumHeld() * 32;
    if (new_repo + arcs + elems > (int64_t)(opts_.max_mem * 0.8)) {
      double eff = beam_; if (!queue_.empty()) eff = queue_.top()->priority_cost - backward_[f_.start];
      K3H_WARN << "Did not reach requested beam in determinize-lattice: size exceeds maximum " << opts_.max_mem << " bytes; (repo,arcs,elems) = (" << repo <<
          "," << arcs << "," << elems
               << "), after rebuilding, repo size was " << new_repo << ", effective beam was " << eff << " vs. requested beam " << beam_;
      return false;
    }
    return true;
  }

  const InputFst &f_; double beam_; DeterminizeLatticePrunedOptions opts_;
  double cutoff_ = kInfD; std::vector<double> backward_;
  StringTrie trie_; int64_t num_arcs_ = 0, num_elems_ = 0;
  std::vector<int32_t> slot_; std::vector<Element> closure_;       // scratch of EpsilonClosure
  std::deque<OutputState> out_;                              // deque: the hashes keep pointers to the subsets
  std::deque<std::vector<Element>> initial_keys_;
  std::unordered_map<const std::vector<Element> *, int32_t, SubsetHash, SubsetEqual> minimal_hash_;
  std::unordered_map<const std::vector<Element> *, Element, SubsetHash, SubsetEqual> initial_hash_;
  std::priority_queue<Task *, std::vector<Task *>, TaskCompare> queue_;
};
}  // namespace

namespace {
// lat/determinize-lattice-pruned.cc:1190-1236 / :1243-1287: determinize; when a limit cut it short at less than retry_cutoff x beam,
// prune the raw lattice to a narrower beam and start over (at most 10 times).  `emit` receives the determinizer that is kept.
template <class Emit> bool DeterminizeWithRetries(InputFst f, double beam, const DeterminizeLatticePrunedOptions &opts, Emit emit) {
  if (!(beam > 0.0)) K3H_ERR << "DeterminizeLatticePruned: beam must be positive, got " << beam;
  if (!(opts.retry_cutoff >= 0.0f && opts.retry_cutoff < 1.0f)) K3H_ERR << "DeterminizeLatticePruned: retry-cutoff must be in [0, 1)";
  for (int iter = 0;; iter++) {
    Determinizer det(f, beam, opts);
    double effective_beam;
    const bool ans = det.Determinize(&effective_beam);
    if (effective_beam >= beam * opts.retry_cutoff || beam == kInfD || iter + 1 == 10) { emit(det); return ans; }
    if (effective_beam < 0.0) effective_beam = 0.0;
    double new_beam = beam * std::sqrt(effective_beam / beam);
    if (new_beam < 0.5 * beam) new_beam = 0.5 * beam;
    beam = new_beam;
    PruneInput(beam, &f);
    K3H_LOG << "Pruned state-level lattice with beam " << beam << " and retrying determinization with that beam.";
    if (f.NumStates() == 0) return false;
  }
}

// DeterminizeLatticeInsertPhones (:1291-1343): a phone label on the word side of every arc that starts a phone (a non-self-loop
// transition out of HMM state 0), on an extra arc when the arc already carries a word; arcs leaving the start state are left alone.
// Returns the first label used for phones.
int32_t InsertPhones(const TransitionInfo &ti, EdgeFst *e) {
  int32_t first = 1; for (int32_t wd : e->word) first = std::max(first, wd + 1);
  const size_t na = e->src.size();
  for (size_t a = 0; a < na; a++) {
    const int32_t t = e->tid[a];
    if (e->src[a] == e->start || t == 0) continue;
    if (t < 0 || t >= (int32_t)ti.id2phone.size()) K3H_ERR << "Lattice has transition-id " << t << " but the model has " << ti.id2phone.size() - 1;
    if (!ti.phone_start[t] || ti.self_loop[t]) continue;
    const int32_t label = first + ti.id2phone[t];
    if (e->word[a] == 0) e->word[a] = label;
    else { const int32_t x = e->AddState(); e->AddArc(x, e->dst[a], label, 0, One()); e->dst[a] = x; }
  }
  return first;
}
}  // namespace

bool DeterminizeLatticePruned(const Lattice &lat, double beam, CompactLattice *clat, const DeterminizeLatticePrunedOptions &opts) {
  *clat = CompactLattice();
  InputFst f = SortedInput(InvertedEdges(lat));
  if (f.NumStates() == 0) return true;
  return DeterminizeWithRetries(std::move(f), beam, opts, [&](const Determinizer &det) { det.Output(clat); Connect(clat); });
}

// DeterminizeLatticePhonePrunedWrapper (:1479-1499) -> DeterminizeLatticePhonePruned (:1410-1462): first a pass over the lattice with
// phone labels inserted at the phone boundaries (DeterminizeLatticePhonePrunedFirstPass :1388-1407; its output is an ordinary FST,
// no longer deterministic once the phone labels are deleted again), then the word-level pass.
bool DeterminizeLatticePhonePruned(const Lattice &lat, const TransitionInfo &trans, double beam, CompactLattice *clat, const DeterminizeLatticePhonePrunedOptions &opts) {
  // ConvertLattice(..., invert = false) of an automaton that has the words on its input side (:1422-1425, :1443-1446)
  auto convert = [&](const EdgeFst &e) {
    Lattice l; l.start = e.start; const size_t ns = e.fin.size();
    l.st_frame.assign(ns, 0); l.st_state.assign(ns, 0); l.st_final.resize(ns); l.st_final_ac.resize(ns);
    for (size_t s = 0; s < ns; s++) { const bool f = e.fin[s] != Zero(); l.st_final[s] = f ? e.fin[s].g : kInfF; l.st_final_ac[s] = f ? e.fin[s].a : 0.0f; }
    l.arc_src = e.src; l.arc_dst = e.dst; l.arc_ilabel = e.tid; l.arc_olabel = e.word;
    for (const LatW &w : e.w) { l.arc_graph.push_back(w.g); l.arc_ac.push_back(w.a); }
    ConvertLattice(l, clat); Connect(clat);          // the wrapper's Connect (:1497)
  };
  auto sorted_edges = [&](const InputFst &f) {           // back to an edge list, in the order the determinizer's input had (TopSort + ArcSort)
    EdgeFst e; e.start = f.start; e.fin = f.fin;
    for (int32_t s = 0; s < f.NumStates(); s++) for (int32_t k = f.off[s]; k < f.off[s + 1]; k++) e.AddArc(s, f.next[k], f.word[k], f.tid[k], f.w[k]);
    return e;
  };
  if (!opts.phone_determinize && !opts.word_determinize) {
    K3H_WARN << "Both --phone-determinize and --word-determinize are set to false, copying lattice without determinization.";
    convert(sorted_edges(SortedInput(InvertedEdges(lat)))); return true;
  }
  DeterminizeLatticePrunedOptions det_opts; det_opts.delta = opts.delta; det_opts.max_mem = opts.max_mem;
  // :1455-1460, then the wrapper's Connect (:1497): pushing and minimizing see the determinizer's output before it is trimmed
  auto finish = [&](bool ans) {
    if (opts.minimize) { ans = PushCompactLatticeStrings(clat) && ans; ans = PushCompactLatticeWeights(clat) && ans; ans = MinimizeCompactLattice(clat) && ans; }
    Connect(clat);
    return ans;
  };
  if (!opts.phone_determinize) {
    *clat = CompactLattice();
    InputFst f = SortedInput(InvertedEdges(lat));
    if (f.NumStates() == 0) return true;
    return finish(DeterminizeWithRetries(std::move(f), beam, det_opts, [&](const Determinizer &det) { det.Output(clat); }));
  }
  *clat = CompactLattice();
  EdgeFst e = InvertedEdges(lat);
  if (e.fin.empty()) return true;
  const int32_t first_phone_label = InsertPhones(trans, &e);
  InputFst f = SortedInput(e);
  EdgeFst pass1;
  bool ans = DeterminizeWithRetries(std::move(f), beam, det_opts, [&](const Determinizer &det) { det.Output(&pass1); });
  for (int32_t &wd : pass1.word) if (wd >= first_phone_label) wd = 0;          // DeterminizeLatticeDeletePhones :1346-1368
  if (!opts.word_determinize) { convert(sorted_edges(SortedInput(pass1, false))); return ans; }      // TopSort :1405, then ConvertLattice :1443-1446
  InputFst g = SortedInput(pass1);
  if (g.NumStates() == 0) return ans;
  return finish(DeterminizeWithRetries(std::move(g), beam, det_opts, [&](const Determinizer &det) { det.Output(clat); }) && ans);
}

bool PruneLattice(double beam, Lattice *lat) {
  if (!(beam > 0.0)) K3H_ERR << "PruneLattice: beam must be positive";
  Connect(lat);
  const int32_t n = lat->NumStates(); const size_t na = lat->arc_src.size();
  if (n == 0) return false;
  std::vector<int32_t> off(n + 1, 0), idx(na), nx(na);
  for (size_t a = 0; a < na; a++) off[lat->arc_src[a] + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < na; a++) { const int32_t k = p[lat->arc_src[a]]++; idx[k] = (int32_t)a; nx[k] = lat->arc_dst[a]; } }
  std::vector<int32_t> order;
  if (!TopOrder(n, lat->start, off, nx, &order)) { K3H_WARN << "Cycles detected in lattice"; return false; }
  auto fin = [&](int32_t s) { return std::isfinite(lat->st_final[s]) ? (double)lat->st_final[s] + (lat->st_final_ac.empty() ? 0.0 : (double)lat->st_final_ac[s]) : kInfD; };
  auto cost = [&](int32_t a) { return (double)lat->arc_graph[a] + (double)lat->arc_ac[a]; };
  std::vector<double> fwd(n, kInfD), bwd(n, kInfD); fwd[lat->start] = 0.0; double best = kInfD;
  for (int32_t s : order) {
    for (int32_t k = off[s]; k < off[s + 1]; k++) {
      const double c = fwd[s] + cost(idx[k]);
      if (c < fwd[nx[k]]) fwd[nx[k]] = c;
    }
    best = std::min(best, fwd[s] + fin(s));
  }
  const double cutoff = best + beam;
  std::vector<char> drop(na, 0);
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    const int32_t s = *it; double b = fin(s);
    if (b + fwd[s] > cutoff && b != kInfD) lat->st_final[s] = kInfF;
    for (int32_t k = off[s]; k < off[s + 1]; k++) { const double ab = cost(idx[k]) + bwd[nx[k]]; if (ab < b) b = ab; if (fwd[s] + ab > cutoff) drop[idx[k]] = 1; }
    bwd[s] = b;
  }
  size_t o = 0;
  for (size_t a = 0; a < na; a++) if (!drop[a]) {
    lat->arc_src[o] = lat->arc_src[a];
    lat->arc_dst[o] = lat->arc_dst[a];
    lat->arc_ilabel[o] = lat->arc_ilabel[a];
    lat->arc_olabel[o] = lat->arc_olabel[a];
    lat->arc_graph[o] = lat->arc_graph[a];
    lat->arc_ac[o] = lat->arc_ac[a];
    o++;
  }
  for (auto *v : {&lat->arc_src, &lat->arc_dst, &lat->arc_ilabel, &lat->arc_olabel}) v->resize(o);
  lat->arc_graph.resize(o); lat->arc_ac.resize(o);
  Connect(lat);
  return true;
}

// ------------------------------------------------------------------------------------------------ CompactLattice ----
namespace {
void Renumber(CompactLattice *c, const std::vector<int32_t> &newid, int32_t m) {     // newid[s] < 0: state removed
  CompactLattice o; for (int32_t i = 0; i < m; i++) o.AddState();
  o.start = c->start >= 0 ? newid[c->start] : -1;
  for (int32_t s = 0; s < c->NumStates(); s++) if (newid[s] >= 0) {
    const int32_t t = newid[s];
    o.is_final[t] = c->is_final[s];
    o.fin_graph[t] = c->fin_graph[s];
    o.fin_ac[t] = c->fin_ac[s];
    o.fin_str[t] = std::move(c->fin_str[s]);
  }
  // arcs stay grouped by (new) source state, in their old relative order
  std::vector<int32_t> keep; for (size_t a = 0; a < c->arc_src.size(); a++) if (newid[c->arc_src[a]] >= 0 && newid[c->arc_dst[a]] >= 0) keep.push_back((int32_t)a);
  std::stable_sort(keep.begin(), keep.end(), [&](int32_t x, int32_t y) { return newid[c->arc_src[x]] < newid[c->arc_src[y]]; });
  for (int32_t a : keep) {
    o.arc_src.push_back(newid[c->arc_src[a]]);
    o.arc_dst.push_back(newid[c->arc_dst[a]]);
    o.arc_label.push_back(c->arc_label[a]);
    o.arc_graph.push_back(c->arc_graph[a]);
    o.arc_ac.push_back(c->arc_ac[a]);
    o.arc_str.push_back(std::move(c->arc_str[a]));
  }
  if (o.start < 0) o = CompactLattice();
  *c = std::move(o);
}
void Csr(const CompactLattice &c, std::vector<int32_t> *off, std::vector<int32_t> *nx) {
  const int32_t n = c.NumStates(); off->assign(n + 1, 0); nx->resize(c.arc_src.size());
  for (int32_t s : c.arc_src) (*off)[s + 1]++;
  for (int32_t s = 0; s < n; s++) (*off)[s + 1] += (*off)[s];
  std::vector<int32_t> p(off->begin(), off->end() - 1); for (size_t a = 0; a < c.arc_src.size(); a++) (*nx)[p[c.arc_src[a]]++] = c.arc_dst[a];
}
}  // namespace

void Connect(CompactLattice *c) {
  const int32_t n = c->NumStates();
  if (n == 0) return;
  std::vector<int32_t> off, nx; Csr(*c, &off, &nx);
  std::vector<int32_t> roff(n + 1, 0), radj(c->arc_src.size());
  for (int32_t d : c->arc_dst) roff[d + 1]++;
  for (int32_t s = 0; s < n; s++) roff[s + 1] += roff[s];
  { std::vector<int32_t> p(roff.begin(), roff.end() - 1); for (size_t a = 0; a < c->arc_src.size(); a++) radj[p[c->arc_dst[a]]++] = c->arc_src[a]; }
  std::vector<char> acc(n, 0), co(n, 0); std::vector<int32_t> st;
  if (c->start >= 0) { acc[c->start] = 1; st.push_back(c->start); }
  while (!st.empty()) { const int32_t s = st.back(); st.pop_back(); for (int32_t k = off[s]; k < off[s + 1]; k++) if (!acc[nx[k]]) { acc[nx[k]] = 1; st.push_back(nx[k]); } }
  for (int32_t s = 0; s < n; s++) if (c->is_final[s]) { co[s] = 1; st.push_back(s); }
  while (!st.empty()) { const int32_t s = st.back(); st.pop_back(); for (int32_t k = roff[s]; k < roff[s + 1]; k++) if (!co[radj[k]]) { co[radj[k]] = 1; st.push_back(radj[k]); } }
  std::vector<int32_t> newid(n, -1); int32_t m = 0; for (int32_t s = 0; s < n; s++) if (acc[s] && co[s]) newid[s] = m++;
  if (m == n) return;
  Renumber(c, newid, m);
}

void ConvertLattice(const Lattice &lat, CompactLattice *out) {
  *out = CompactLattice();
  const int32_t n = lat.NumStates(); const size_t na = lat.arc_src.size();
  if (n == 0 || lat.start < 0) return;
  std::vector<int32_t> off(n + 1, 0), idx(na);
  for (size_t a = 0; a < na; a++) off[lat.arc_src[a] + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < na; a++) idx[p[lat.arc_src[a]]++] = (int32_t)a; }
  // depth-first discovery order (DfsOrderVisitor), arcs in stored order
  std::vector<int32_t> order, stack, pos(n); std::vector<char> seen(n, 0);
  auto visit = [&](int32_t root) {
    stack.push_back(root); seen[root] = 1; pos[root] = off[root]; order.push_back(root);
    while (!stack.empty()) {
      const int32_t s = stack.back();
      if (pos[s] < off[s + 1]) { const int32_t d = lat.arc_dst[idx[pos[s]++]]; if (!seen[d]) { seen[d] = 1; pos[d] = off[d]; order.push_back(d); stack.push_back(d); } }
      else stack.pop_back();
    }
  };
  visit(lat.start);
  for (int32_t s = 0; s < n; s++) if (!seen[s]) visit(s);           // DfsVisit goes on with the states the start state does not reach
  // GetStateProperties: a state is the middle of a chain when it has exactly one arc in and one out, is not final, not the start
  // state, and its arc out carries no word
  std::vector<int32_t> nin(n, 0); for (size_t a = 0; a < na; a++) nin[lat.arc_dst[a]]++;
  std::vector<char> remove(n, 0);
  for (int32_t s = 0; s < n; s++)
    remove[s] = s != lat.start && nin[s] == 1 && off[s + 1] - off[s] == 1 && !std::isfinite(lat.st_final[s]) && lat.arc_olabel[idx[off[s]]] == 0;
  EdgeFst f; std::vector<int32_t> map(n, -1); std::vector<std::vector<int32_t>> strings;
  auto state_of = [&](int32_t s) { if (map[s] < 0) map[s] = f.AddState(); return map[s]; };
  for (int32_t s : order) {
    if (remove[s]) continue;
    const int32_t ns = state_of(s);
    for (int32_t k = off[s]; k < off[s + 1]; k++) {
      int32_t a = idx[k]; LatW w{lat.arc_graph[a], lat.arc_ac[a]}; const int32_t word = lat.arc_olabel[a];
      std::vector<int32_t> str; if (lat.arc_ilabel[a] != 0) str.push_back(lat.arc_ilabel[a]);
      int32_t d = lat.arc_dst[a];
      while (remove[d]) { a = idx[off[d]]; w = Times(w, LatW{lat.arc_graph[a], lat.arc_ac[a]}); if (lat.arc_ilabel[a] != 0) str.push_back(lat.arc_ilabel[a]); d = lat.arc_dst[a]; }
      f.AddArc(ns, state_of(d), word, (int32_t)strings.size(), w); strings.push_back(std::move(str));       // the tid slot holds the index of the arc's string
    }
    if (std::isfinite(lat.st_final[s])) f.fin[ns] = LatW{lat.st_final[s], lat.st_final_ac.empty() ? 0.0f : lat.st_final_ac[s]};
  }
  f.start = map[lat.start];
  // TopSort(&ffst), then one compact arc per factored arc
  const int32_t m = (int32_t)f.fin.size();
  std::vector<int32_t> foff(m + 1, 0), fnx(f.src.size()), fidx(f.src.size());
  for (int32_t x : f.src) foff[x + 1]++;
  for (int32_t x = 0; x < m; x++) foff[x + 1] += foff[x];
  { std::vector<int32_t> p(foff.begin(), foff.end() - 1); for (size_t a = 0; a < f.src.size(); a++) { const int32_t k = p[f.src[a]]++; fidx[k] = (int32_t)a; fnx[k] = f.dst[a]; } }
  std::vector<int32_t> torder;
  if (!TopOrder(m, f.start, foff, fnx, &torder)) K3H_ERR << "ConvertLattice: the lattice has a cycle";
  std::vector<int32_t> newid(m, -1); for (size_t i = 0; i < torder.size(); i++) newid[torder[i]] = (int32_t)i;
  for (size_t i = 0; i < torder.size(); i++) out->AddState();
  out->start = newid[f.start];
  for (int32_t x : torder) {
    const int32_t t = newid[x];
    if (f.fin[x] != Zero()) { out->is_final[t] = 1; out->fin_graph[t] = f.fin[x].g; out->fin_ac[t] = f.fin[x].a; }
    for (int32_t k = foff[x]; k < foff[x + 1]; k++) {
      const int32_t a = fidx[k];
      if (newid[f.dst[a]] < 0) continue;
      out->arc_src.push_back(t);
      out->arc_dst.push_back(newid[f.dst[a]]);
      out->arc_label.push_back(f.word[a]);
      out->arc_graph.push_back(f.w[a].g);
      out->arc_ac.push_back(f.w[a].a);
      out->arc_str.push_back(std::move(strings[f.tid[a]]));
    }
  }
}

void ConvertLattice(const CompactLattice &c, Lattice *out) {
  Lattice l; const int32_t n = c.NumStates();
  l.start = c.start; l.st_final.assign(n, kInfF); l.st_final_ac.assign(n, 0.0f);
  auto add_state = [&]() { l.st_final.push_back(kInfF); l.st_final_ac.push_back(0.0f); return (int32_t)l.st_final.size() - 1; };
  auto add_arc = [&](int32_t s, int32_t d, int32_t tid, int32_t word, float g, float a) {
    l.arc_src.push_back(s);
    l.arc_dst.push_back(d);
    l.arc_ilabel.push_back(tid);
    l.arc_olabel.push_back(word);
    l.arc_graph.push_back(g);
    l.arc_ac.push_back(a);
  };
  std::vector<int32_t> off(n + 1, 0), idx(c.arc_src.size());
  for (int32_t s : c.arc_src) off[s + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < c.arc_src.size(); a++) idx[p[c.arc_src[a]]++] = (int32_t)a; }
  for (int32_t s = 0; s < n; s++) {
    if (c.is_final[s]) {
      int32_t cur = s; const std::vector<int32_t> &str = c.fin_str[s];
      for (size_t k = 0; k < str.size(); k++) {
        const int32_t nx = add_state();
        add_arc(cur, nx, str[k], 0, k == 0 ? c.fin_graph[s] : 0.0f, k == 0 ? c.fin_ac[s] : 0.0f);
        cur = nx;
      }
      l.st_final[cur] = str.empty() ? c.fin_graph[s] : 0.0f; l.st_final_ac[cur] = str.empty() ? c.fin_ac[s] : 0.0f;
    }
    for (int32_t k = off[s]; k < off[s + 1]; k++) {
      const int32_t a = idx[k]; const std::vector<int32_t> &str = c.arc_str[a]; int32_t cur = s;
      for (size_t i = 0; i + 1 < str.size(); i++) {
        const int32_t nx = add_state();
        add_arc(cur, nx, str[i], i == 0 ? c.arc_label[a] : 0, i == 0 ? c.arc_graph[a] : 0.0f, i == 0 ? c.arc_ac[a] : 0.0f);
        cur = nx;
      }
      const bool single = str.size() <= 1;
      add_arc(cur, c.arc_dst[a], str.empty() ? 0 : str.back(), single ? c.arc_label[a] : 0, single ? c.arc_graph[a] : 0.0f, single ? c.arc_ac[a] : 0.0f);
    }
  }
  l.st_frame.assign(l.st_final.size(), 0); l.st_state.assign(l.st_final.size(), 0);
  *out = std::move(l);
}

void ScaleAcoustic(CompactLattice *c, double scale) {
  for (float &a : c->arc_ac) a = (float)(a * scale);
  for (int32_t s = 0; s < c->NumStates(); s++) if (c->is_final[s]) c->fin_ac[s] = (float)(c->fin_ac[s] * scale);
}

bool TopSortIfNeeded(CompactLattice *c) {
  bool sorted = true;
  for (size_t a = 0; a < c->arc_src.size() && sorted; a++) sorted = c->arc_dst[a] > c->arc_src[a];
  if (sorted) return true;
  std::vector<int32_t> off, nx, order; Csr(*c, &off, &nx);
  if (!TopOrder(c->NumStates(), c->start, off, nx, &order)) return false;
  std::vector<int32_t> newid(c->NumStates(), -1); for (size_t i = 0; i < order.size(); i++) newid[order[i]] = (int32_t)i;
  int32_t m = (int32_t)order.size(); for (int32_t s = 0; s < c->NumStates(); s++) if (newid[s] < 0) newid[s] = m++;     // unreachable states go last
  Renumber(c, newid, m);
  return true;
}

// ------------------------------------------------------------------------------------------------ push + minimize ----
namespace {
struct ArcsByState {                 // arc indices grouped by source state, in stored order
  std::vector<int32_t> off, idx;
  explicit ArcsByState(const CompactLattice &c) : off(c.NumStates() + 1, 0), idx(c.arc_src.size()) {
    for (int32_t s : c.arc_src) off[s + 1]++;
    for (int32_t s = 0; s < c.NumStates(); s++) off[s + 1] += off[s];
    std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < c.arc_src.size(); a++) idx[p[c.arc_src[a]]++] = (int32_t)a;
  }
  int32_t Num(int32_t s) const { return off[s + 1] - off[s]; }
  int32_t Arc(int32_t s, int32_t k) const { return idx[off[s] + k]; }
};
// CompactLatticePusher::GetString (push-lattice.cc:53-84): the first `len` transition-ids met from `state`, through arc `arc_k`
// (-1: any path -- the final string if the state is final, else the first arc)
void GetString(const CompactLattice &c, const ArcsByState &by, int32_t state, int32_t arc_k, int32_t *out, size_t len) {
  if (len == 0) return;
  if (arc_k < 0 && c.is_final[state]) {
    if (c.fin_str[state].size() < len) K3H_ERR << "PushCompactLatticeStrings: paths in lattice have inconsistent lengths";
    std::copy(c.fin_str[state].begin(), c.fin_str[state].begin() + len, out); return;
  }
  if (by.Num(state) == 0 || arc_k >= by.Num(state)) K3H_ERR << "PushCompactLatticeStrings: paths in lattice are inconsistent in length";
  const int32_t a = by.Arc(state, arc_k < 0 ? 0 : arc_k); const std::vector<int32_t> &str = c.arc_str[a];
  if (str.size() >= len) std::copy(str.begin(), str.begin() + len, out);
  else { std::copy(str.begin(), str.end(), out); GetString(c, by, c.arc_dst[a], -1, out + str.size(), len - str.size()); }
}
}  // namespace

bool PushCompactLatticeStrings(CompactLattice *c) {      // push-lattice.cc:30-227
  if (!TopSortIfNeeded(c)) {
    K3H_WARN << "Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles; this  is a bad idea.)";
    return false;
  }
  const int32_t n = c->NumStates(); const ArcsByState by(*c);
  std::vector<int32_t> shift(n, 0);
  for (int32_t s = n - 1; s > c->start; s--) {           // ComputeShifts :134-164; the start state keeps shift 0
    const int32_t narcs = by.Num(s);
    if (narcs == 0) { shift[s] = c->is_final[s] ? (int32_t)c->fin_str[s].size() : 0; continue; }
    int32_t sh = std::numeric_limits<int32_t>::max();
    if (c->is_final[s]) sh = std::min(sh, (int32_t)c->fin_str[s].size());
    for (int32_t k = 0; k < narcs; k++) { const int32_t a = by.Arc(s, k); sh = std::min(sh, shift[c->arc_dst[a]] + (int32_t)c->arc_str[a].size()); }
    if (narcs + (c->is_final[s] ? 1 : 0) > 1 && sh > 0) {     // CheckForConflict :86-131: the strings moved back must agree on every way out
      std::vector<int32_t> str(sh), cmp(sh); int32_t k;
      if (c->is_final[s]) { std::copy(c->fin_str[s].begin(), c->fin_str[s].begin() + sh, str.begin()); k = 0; }
      else { GetString(*c, by, s, 0, str.data(), str.size()); k = 1; }
      for (; k < narcs; k++) {
        GetString(*c, by, s, k, cmp.data(), cmp.size());
        const auto pr = std::mismatch(str.begin(), str.end(), cmp.begin());
        if (pr.first != str.end()) { sh = (int32_t)(pr.first - str.begin()); str.resize(sh); cmp.resize(sh); }
      }
    }
    shift[s] = sh;
  }
  // ApplyShifts :166-200 (in state order; GetString reads strings of later states, which are still unmodified)
  CompactLattice o = *c;
  for (int32_t s = 0; s < n; s++) {
    for (int32_t k = 0; k < by.Num(s); k++) {
      const int32_t a = by.Arc(s, k); std::vector<int32_t> str = c->arc_str[a]; const size_t orig = str.size(), next_shift = (size_t)shift[c->arc_dst[a]];
      str.resize(orig + next_shift);
      GetString(o, by, c->arc_dst[a], -1, str.data() + orig, next_shift);
      o.arc_str[a].assign(str.begin() + shift[s], str.end());
    }
    if (c->is_final[s]) o.fin_str[s].assign(c->fin_str[s].begin() + shift[s], c->fin_str[s].end());
  }
  *c = std::move(o);
  return true;
}

bool PushCompactLatticeWeights(CompactLattice *c) {      // push-lattice.cc:236-289
  if (!TopSortIfNeeded(c)) {
    K3H_WARN << "Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles; this  is a bad idea.)";
    return false;
  }
  const int32_t n = c->NumStates();
  if (n == 0) { K3H_WARN << "Pushing weights of empty compact lattice"; return true; }
  const ArcsByState by(*c);
  std::vector<LatW> to_end(n);
  for (int32_t s = n - 1; s >= 0; s--) {
    LatW w = c->is_final[s] ? LatW{c->fin_graph[s], c->fin_ac[s]} : Zero();
    for (int32_t k = 0; k < by.Num(s); k++) { const int32_t a = by.Arc(s, k); w = Plus(w, Times(LatW{c->arc_graph[a], c->arc_ac[a]}, to_end[c->arc_dst[a]])); }
    if (w == Zero()) K3H_WARN << "Lattice has non-coaccessible states.";
    to_end[s] = w;
  }
  to_end[0] = One();                 // the leftover weight stays on the start state
  for (int32_t s = 0; s < n; s++) {
    const LatW here = to_end[s];
    if (here == Zero()) continue;
    for (int32_t k = 0; k < by.Num(s); k++) {
      const int32_t a = by.Arc(s, k); const LatW next = to_end[c->arc_dst[a]];
      if (next != Zero()) { const LatW w = Times(LatW{c->arc_graph[a], c->arc_ac[a]}, Divide(next, here)); c->arc_graph[a] = w.g; c->arc_ac[a] = w.a; }
    }
    if (c->is_final[s]) { const LatW w = Divide(LatW{c->fin_graph[s], c->fin_ac[s]}, here); c->fin_graph[s] = w.g; c->fin_ac[s] = w.a; }
  }
  return true;
}

bool MinimizeCompactLattice(CompactLattice *c, float delta) {      // minimize-lattice.cc:37-275
  if (!TopSortIfNeeded(c)) {
    K3H_WARN << "Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles; this  is a bad idea.)";
    return false;
  }
  const int32_t n = c->NumStates(); const ArcsByState by(*c);
  // VectorHasher, never 0
  auto string_hash = [](const std::vector<int32_t> &v) {
    size_t h = 0;
    for (int32_t x : v) {
      h *= 7853;
      h += (size_t)x;
    }
    return h == 0 ? (size_t)53281 : h;
  };
  std::vector<size_t> hash(n);
  for (int32_t s = n - 1; s >= 0; s--) {                 // ComputeStateHashValues :98-123: order-insensitive over the arcs
    size_t h = c->is_final[s] ? (size_t)607 * string_hash(c->fin_str[s]) : (size_t)33317;
    for (int32_t k = 0; k < by.Num(s); k++) {
      const int32_t a = by.Arc(s, k); size_t label = (size_t)c->arc_label[a]; if (label == 0) label = 51907;
      h += (size_t)1447 * label * (1 + string_hash(c->arc_str[a]) * hash[c->arc_dst[a]]);
    }
    hash[s] = h;
  }
  std::unordered_map<size_t, std::vector<int32_t>> groups;
  for (int32_t s = 0; s < n; s++) groups[hash[s]].push_back(s);
  std::vector<int32_t> map(n); for (int32_t s = 0; s < n; s++) map[s] = s;
  struct A { int32_t label, next; LatW w; const std::vector<int32_t> *str; };
  auto arcs_of = [&](int32_t s) {
    std::vector<A> v;
    for (int32_t k = 0; k < by.Num(s); k++) {
      const int32_t a = by.Arc(s, k);
      v.push_back({c->arc_label[a], map[c->arc_dst[a]], LatW{c->arc_graph[a], c->arc_ac[a]}, &c->arc_str[a]});
    }
    std::sort(v.begin(), v.end(), [](const A &x, const A &y) { return x.label != y.label ? x.label < y.label : x.next < y.next; });
    return v;
  };
  auto equivalent = [&](int32_t s, int32_t t) {          // :143-186
    if (c->is_final[s] != c->is_final[t]) return false;
    if (c->is_final[s] && !(ApproxEqual(LatW{c->fin_graph[s], c->fin_ac[s]}, LatW{c->fin_graph[t], c->fin_ac[t]}, delta) && c->fin_str[s] == c->fin_str[t])) return false;
    if (by.Num(s) != by.Num(t)) return false;
    const std::vector<A> x = arcs_of(s), y = arcs_of(t);
    for (size_t i = 0; i < x.size(); i++) if (x[i].next != y[i].next || x[i].label != y[i].label ||
        !(ApproxEqual(x[i].w, y[i].w, 1.0f / 1024.0f) && *x[i].str == *y[i].str)) return false;
    return true;
  };
  for (int32_t s = n - 1; s >= 0; s--)                   // ComputeStateMap :188-229
    for (int32_t t : groups[hash[s]]) if (t > s && map[t] == t && equivalent(s, t)) { map[s] = t; break; }
  bool any = false; for (int32_t s = 0; s < n; s++) any |= map[s] != s;
  if (!any) return true;
  c->start = map[c->start];                              // ModifyModel :231-262
  for (size_t a = 0; a < c->arc_src.size(); a++) if (map[c->arc_src[a]] == c->arc_src[a]) c->arc_dst[a] = map[c->arc_dst[a]];
  Connect(c);
  return true;
}

namespace {
template <class T> void Put(std::string *o, T v) { o->append((const char *)&v, sizeof(T)); }
void PutStr(std::string *o, const std::string &s) { Put<int32_t>(o, (int32_t)s.size()); o->append(s); }
std::string Num(float v) { if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity"; char buf[64]; snprintf(buf, sizeof buf, "%g", (double)v); return buf; }
void PrintCompactWeight(std::string *o, float g, float a, const std::vector<int32_t> &str) {     // operator<< of CompactLatticeWeightTpl (fstext/lattice-weight.h:728-738)
  *o += Num(g); *o += ","; *o += Num(a); *o += ",";
  for (size_t i = 0; i < str.size(); i++) { if (i) *o += "_"; *o += std::to_string(str[i]); }
}
}  // namespace

void TableWriter::WriteCompactLattice(const std::string &key, const CompactLattice &c) {      // WriteCompactLattice (lat/kaldi-lattice.cc:65-94)
  const int32_t n = c.NumStates(); const size_t na = c.arc_src.size();
  std::vector<int32_t> off(n + 1, 0), order(na);
  for (size_t a = 0; a < na; a++) off[c.arc_src[a] + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < na; a++) order[p[c.arc_src[a]]++] = (int32_t)a; }
  std::string o = key + " ";
  if (!binary_) {          // FstPrinter in acceptor mode: "src dst label weight", weight left out when it is One (0,0,empty string)
    o += "\n";
    auto print_state = [&](int32_t s) {
      for (int32_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t a = order[k];
        o += std::to_string(s) + "\t" + std::to_string(c.arc_dst[a]) + "\t" + std::to_string(c.arc_label[a]);
        if (!(c.arc_graph[a] == 0.0f && c.arc_ac[a] == 0.0f && c.arc_str[a].empty())) { o += "\t"; PrintCompactWeight(&o, c.arc_graph[a], c.arc_ac[a], c.arc_str[a]); }
        o += "\n";
      }
      if (c.is_final[s]) {
        o += std::to_string(s);
        if (!(c.fin_graph[s] == 0.0f && c.fin_ac[s] == 0.0f && c.fin_str[s].empty())) {
          o += "\t";
          PrintCompactWeight(&o, c.fin_graph[s], c.fin_ac[s], c.fin_str[s]);
        }
        o += "\n";
      }
    };
    if (c.start >= 0) print_state(c.start);
    for (int32_t s = 0; s < n; s++) if (s != c.start) print_state(s);
    o += "\n";
  } else {                  // VectorFst<CompactLatticeArc>::Write; the weight is {2 floats, int32 length, int32 labels} (lattice-weight.h:529-539)
    Put<int32_t>(&o, 2125659606); PutStr(&o, "vector"); PutStr(&o, "compactlattice44");
    Put<int32_t>(&o, 2); Put<int32_t>(&o, 0); Put<uint64_t>(&o, 3); Put<int64_t>(&o, c.start); Put<int64_t>(&o, n); Put<int64_t>(&o, (int64_t)na);
    auto put_w = [&](float g, float a, const std::vector<int32_t> &str) { Put(&o, g); Put(&o, a); Put<int32_t>(&o, (int32_t)str.size()); for (int32_t t : str) Put(&o, t); };
    static const std::vector<int32_t> kEmpty;
    for (int32_t s = 0; s < n; s++) {
      if (c.is_final[s]) put_w(c.fin_graph[s], c.fin_ac[s], c.fin_str[s]); else put_w(kInfF, kInfF, kEmpty);
      Put<int64_t>(&o, off[s + 1] - off[s]);
      for (int32_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t a = order[k];
        Put(&o, c.arc_label[a]);
        Put(&o, c.arc_label[a]);
        put_w(c.arc_graph[a], c.arc_ac[a], c.arc_str[a]);
        Put(&o, c.arc_dst[a]);
      }
    }
  }
  if (fwrite(o.data(), 1, o.size(), f_.get()) != o.size()) K3H_ERR << "Write failure on compact lattice " << key;
}

// ------------------------------------------------------------------------------------------------ DeterminizeSequencer ----
struct DeterminizeSequencer::Impl {
  Config cfg; TableWriter *writer;
  std::mutex m; std::condition_variable cv_work, cv_done;
  struct Job { int64_t seq; std::string key; Lattice lat; };
  std::deque<Job> queue;                                                   // submitted, not yet picked up
  struct Done { std::string key; CompactLattice clat; std::string ctm; };
  std::map<int64_t, Done> finished;      // determinized, waiting for their turn to be written
  int64_t submitted = 0, written = 0; int32_t num_warn = 0; bool stop = false; std::string error;
  std::vector<std::thread> threads;

  void Work() {
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        job = std::move(queue.front());
        queue.pop_front();
      }
      CompactLattice clat; bool warn = false; std::string err, ctm_text;
      try {
        if (cfg.pre_scale != 1.0) ScaleAcoustic(&job.lat, cfg.pre_scale);
        bool ok = true;
        if (!cfg.determinize) ConvertLattice(job.lat, &clat);      // (--determinize-lattice=false with a post-processor: the lattice re-packed as it is)
        else ok = cfg.trans ? DeterminizeLatticePhonePruned(job.lat, *cfg.trans, cfg.beam, &clat, cfg.phone_det) : DeterminizeLatticePruned(job.lat, cfg.beam, &clat, cfg.det);
        if (!ok) { K3H_WARN << "For key " << job.key << ", determinization did not succeed(partial output will be pruned tighter than the specified beam.)"; warn = true; }
        if (clat.NumStates() == 0) { K3H_WARN << "For key " << job.key << ", determinized and trimmed lattice was empty."; warn = true; }
        // with cfg.trans: phone_det.minimize, inside
        if (cfg.minimize && !cfg.trans) {
          PushCompactLatticeStrings(&clat);
          PushCompactLatticeWeights(&clat);
          MinimizeCompactLattice(&clat);
        }
        if (cfg.topsort && !TopSortIfNeeded(&clat)) K3H_WARN << "Topological sorting of the determinized lattice failed for key " << job.key;
        if (cfg.post_scale != 1.0) ScaleAcoustic(&clat, cfg.post_scale);
        if (cfg.postprocessor) {      // SetResultUsingLattice (cudadecoder/lattice-postprocessor.cc:112-137)
          if (cfg.ctm_out) { CtmResult ctm; cfg.postprocessor->GetCTM(clat, &ctm); std::ostringstream os; WriteCtm(ctm, job.key, os, cfg.word_syms); ctm_text = os.str(); }
          else { CompactLattice pp; cfg.postprocessor->GetPostprocessedLattice(clat, &pp); clat = std::move(pp); }
        }
      } catch (const std::exception &e) { err = e.what(); }
      job.lat = Lattice();
      if (cfg.on_done) cfg.on_done(job.key);
      std::unique_lock<std::mutex> lk(m);
      if (!err.empty() && error.empty()) error = err;
      num_warn += warn;
      finished.emplace(job.seq, Done{std::move(job.key), std::move(clat), std::move(ctm_text)});
      // whoever completes the next lattice in line writes it and everything behind it that is already there
      for (auto it = finished.find(written); it != finished.end(); it = finished.find(written)) {
        if (error.empty()) {
          try {
            if (cfg.ctm_out) {
              *cfg.ctm_out << it->second.ctm;
              cfg.ctm_out->flush();
            } else writer->WriteCompactLattice(it->second.key, it->second.clat);
          } catch (const std::exception &e) {
            error = e.what();
          }
        }
        finished.erase(it); written++;
      }
      cv_done.notify_all();
    }
  }
};

DeterminizeSequencer::DeterminizeSequencer(const Config &config, TableWriter *writer) : impl_(new Impl) {
  impl_->cfg = config; impl_->writer = writer;
  const int32_t n = std::max<int32_t>(1, config.num_threads);
  for (int32_t i = 0; i < n; i++) impl_->threads.emplace_back([this] { impl_->Work(); });
}
DeterminizeSequencer::~DeterminizeSequencer() {
  { std::unique_lock<std::mutex> lk(impl_->m); impl_->stop = true; }
  impl_->cv_work.notify_all();
  for (std::thread &t : impl_->threads) t.join();      // workers drain the queue before they leave
}
void DeterminizeSequencer::Run(std::string key, Lattice &&lat) {
  std::unique_lock<std::mutex> lk(impl_->m);
  const int64_t max_in_flight = (int64_t)impl_->threads.size() + 20;
  impl_->cv_done.wait(lk, [&] { return impl_->submitted - impl_->written < max_in_flight; });
  if (!impl_->error.empty()) { const std::string e = impl_->error; lk.unlock(); throw FatalError(e); }
  impl_->queue.push_back(Impl::Job{impl_->submitted++, std::move(key), std::move(lat)});
  lk.unlock(); impl_->cv_work.notify_one();
}
void DeterminizeSequencer::Wait() {
  std::unique_lock<std::mutex> lk(impl_->m);
  impl_->cv_done.wait(lk, [&] { return impl_->written == impl_->submitted; });
  if (!impl_->error.empty()) { const std::string e = impl_->error; lk.unlock(); throw FatalError(e); }
}
int32_t DeterminizeSequencer::NumDone() const { std::unique_lock<std::mutex> lk(impl_->m); return (int32_t)impl_->written; }
int32_t DeterminizeSequencer::NumWarn() const { std::unique_lock<std::mutex> lk(impl_->m); return impl_->num_warn; }

// ------------------------------------------------------------------------------------------------ lattice table reader ----
std::vector<std::pair<std::string, Lattice>> ReadLatticeTable(const std::string &rspecifier) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos || rspecifier.compare(0, 3, "ark") != 0) K3H_ERR << "Invalid lattice rspecifier " << rspecifier <<
      " (supported: ark:<rxfilename>, ark,t:<rxfilename>)";
  const std::string b = ReadWholeInput(rspecifier.substr(colon + 1));
  std::vector<std::pair<std::string, Lattice>> out;
  size_t p = 0;
  auto parse_float = [&](const std::string &t) {
    if (t == "Infinity") return kInfF;
    if (t == "-Infinity") return -kInfF;
    char *e;
    const float v = strtof(t.c_str(), &e);
    if (*e || t.empty()) K3H_ERR << "Bad number \"" << t << "\" in lattice";
    return v;
  };
  while (true) {
    while (p < b.size() && (b[p] == '\n' || b[p] == ' ' || b[p] == '\t' || b[p] == '\r')) p++;
    if (p >= b.size()) break;
    const size_t k0 = p; while (p < b.size() && !isspace((unsigned char)b[p])) p++;
    const std::string key = b.substr(k0, p - k0);
    if (p >= b.size() || b[p] != ' ') K3H_ERR << "Lattice table: expected a space after key " << key;
    p++;
    Lattice lat;
    auto add_state = [&](int64_t s) {
      while ((int64_t)lat.st_final.size() <= s) {
        lat.st_final.push_back(kInfF);
        lat.st_final_ac.push_back(kInfF);
        lat.st_frame.push_back(0);
        lat.st_state.push_back(0);
      }
    };
    if (p < b.size() && (unsigned char)b[p] == 214) {       // first byte of the FST magic number: binary
      auto get = [&](auto *v) { if (p + sizeof(*v) > b.size()) K3H_ERR << "unexpected end of lattice " << key; memcpy(v, b.data() + p, sizeof(*v)); p += sizeof(*v); };
      auto str = [&]() { int32_t n; get(&n); if (n < 0 || p + n > b.size()) K3H_ERR << "corrupt FST header in lattice " << key; std::string s = b.substr(p, n); p += n; return s; };
      int32_t magic, version, flags; uint64_t props; int64_t start, ns, na; get(&magic);
      const std::string ftype = str(), atype = str(); get(&version); get(&flags); get(&props); get(&start); get(&ns); get(&na);
      if (ftype != "vector" || (atype != "lattice4" && atype != "compactlattice44")) K3H_ERR << "Lattice " << key <<
          ": expected a vector FST with arc type lattice4 or compactlattice44, got " << ftype << " / " << atype;
      if (flags & 3) K3H_ERR << "Lattice " << key << " has embedded symbol tables";
      if (atype == "compactlattice44") {
        CompactLattice c; c.start = (int32_t)start; for (int64_t s = 0; s < ns; s++) c.AddState();
        auto weight = [&](float *g, float *a, std::vector<int32_t> *str) {
          int32_t n;
          get(g);
          get(a);
          get(&n);
          if (n < 0) K3H_ERR << "corrupt compact lattice " << key;
          str->resize(n);
          for (int32_t &t : *str) get(&t);
        };
        for (int64_t s = 0; s < ns; s++) {
          float g, a; std::vector<int32_t> str; int64_t n; weight(&g, &a, &str); get(&n);
          if (std::isfinite(g) && std::isfinite(a)) { c.is_final[s] = 1; c.fin_graph[s] = g; c.fin_ac[s] = a; c.fin_str[s] = str; }
          for (int64_t i = 0; i < n; i++) {
            int32_t il, ol, nx;
            get(&il);
            get(&ol);
            weight(&g, &a, &str);
            get(&nx);
            c.arc_src.push_back((int32_t)s);
            c.arc_dst.push_back(nx);
            c.arc_label.push_back(il);
            c.arc_graph.push_back(g);
            c.arc_ac.push_back(a);
            c.arc_str.push_back(str);
          }
        }
        ConvertLattice(c, &lat);
        out.emplace_back(key, std::move(lat)); continue;
      }
      lat.start = (int32_t)start; if (ns > 0) add_state(ns - 1);
      for (int64_t s = 0; s < ns; s++) {
        float g, a; int64_t n; get(&g); get(&a); get(&n);
        lat.st_final[s] = (std::isfinite(g) && std::isfinite(a)) ? g : kInfF; lat.st_final_ac[s] = a;
        for (int64_t i = 0; i < n; i++) {
          int32_t il, ol, nx;
          get(&il);
          get(&ol);
          get(&g);
          get(&a);
          get(&nx);
          lat.arc_src.push_back((int32_t)s);
          lat.arc_dst.push_back(nx);
          lat.arc_ilabel.push_back(il);
          lat.arc_olabel.push_back(ol);
          lat.arc_graph.push_back(g);
          lat.arc_ac.push_back(a);
        }
      }
    } else {                                                  // text: lines until an empty line (lat/kaldi-lattice.cc:204-300 reads the FstPrinter format)
      while (p < b.size() && b[p] != '\n') p++;
      p++;
      bool first = true, compact = false; CompactLattice c;
      auto cadd = [&](int64_t s) { while (c.NumStates() <= s) c.AddState(); };
      auto cweight = [&](const std::string &t, float *g, float *a, std::vector<int32_t> *str) {      // "graph,acoustic,t1_t2_..."
        const size_t c1 = t.find(','), c2 = t.find(',', c1 + 1); *g = parse_float(t.substr(0, c1)); *a = parse_float(t.substr(c1 + 1, c2 - c1 - 1)); str->clear();
        for (size_t q = c2 + 1; q < t.size();) {
          size_t e = t.find('_', q);
          if (e == std::string::npos) e = t.size();
          str->push_back((int32_t)strtol(t.substr(q, e - q).c_str(), nullptr, 10));
          q = e + 1;
        }
      };
      while (p < b.size()) {
        const size_t l0 = p; while (p < b.size() && b[p] != '\n') p++;
        std::string line = b.substr(l0, p - l0); p++;
        std::vector<std::string> col; { std::istringstream ss(line); std::string t; while (ss >> t) col.push_back(t); }
        if (col.empty()) break;
        // a CompactLattice record is an acceptor (3 or 4 columns per arc) whose weights have three comma-separated fields
        if (first && ((col.size() == 3) || (col.size() == 4 && std::count(col[3].begin(), col[3].end(), ',') == 2) ||
            (col.size() == 2 && std::count(col[1].begin(), col[1].end(), ',') == 2))) compact = true;
        if (compact) {
          const int64_t s = strtoll(col[0].c_str(), nullptr, 10); cadd(s);
          if (first) { c.start = (int32_t)s; first = false; }
          float g = 0, a = 0; std::vector<int32_t> str;
          if (col.size() <= 2) { if (col.size() == 2) cweight(col[1], &g, &a, &str); c.is_final[s] = 1; c.fin_graph[s] = g; c.fin_ac[s] = a; c.fin_str[s] = str; }
          else if (col.size() <= 4) { if (col.size() == 4) cweight(col[3], &g, &a, &str); const int64_t d = strtoll(col[1].c_str(), nullptr, 10); cadd(d);
            c.arc_src.push_back((int32_t)s);
            c.arc_dst.push_back((int32_t)d);
            c.arc_label.push_back((int32_t)strtol(col[2].c_str(), nullptr, 10));
            c.arc_graph.push_back(g);
            c.arc_ac.push_back(a);
            c.arc_str.push_back(str);
            }
          else K3H_ERR << "Lattice " << key << ": bad line \"" << line << "\"";
          continue;
        }
        auto weight = [&](const std::string &t, float *g, float *a) {
          const size_t c = t.find(',');
          if (c == std::string::npos || t.find(',', c + 1) != std::string::npos) K3H_ERR << "Lattice " << key << ": bad weight \"" << t << "\"";
          *g = parse_float(t.substr(0, c));
          *a = parse_float(t.substr(c + 1));
        };
        const int64_t s = strtoll(col[0].c_str(), nullptr, 10); add_state(s);
        if (first) { lat.start = (int32_t)s; first = false; }
        if (col.size() <= 2) { float g = 0, a = 0; if (col.size() == 2) weight(col[1], &g, &a); lat.st_final[s] = g; lat.st_final_ac[s] = a; }
        else if (col.size() == 4 || col.size() == 5) {
          float g = 0, a = 0; if (col.size() == 5) weight(col[4], &g, &a);
          const int64_t d = strtoll(col[1].c_str(), nullptr, 10); add_state(d);
          lat.arc_src.push_back((int32_t)s);
          lat.arc_dst.push_back((int32_t)d);
          lat.arc_ilabel.push_back((int32_t)strtol(col[2].c_str(), nullptr, 10));
          lat.arc_olabel.push_back((int32_t)strtol(col[3].c_str(), nullptr, 10));
          lat.arc_graph.push_back(g);
          lat.arc_ac.push_back(a);
        } else K3H_ERR << "Lattice " << key << ": bad line \"" << line << "\"";
      }
      if (compact) ConvertLattice(c, &lat);
    }
    for (size_t s = 0; s < lat.st_final.size(); s++) if (!std::isfinite(lat.st_final[s])) lat.st_final_ac[s] = 0.0f;
    out.emplace_back(key, std::move(lat));
  }
  return out;
}

}  // namespace k3host
