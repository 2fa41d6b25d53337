// oracle/ref_tools/minifst/fst/fstlib.h -- TEST INFRASTRUCTURE.  A stand-in for the part of OpenFst's public interface that the
// reference's decoder/lattice-faster-decoder.{h,cc} and fstext/lattice-weight.h touch, so that those reference sources can be
// compiled UNMODIFIED into oracle/_ref without OpenFst (which /root/reference does not vendor); lat/determinize-lattice-pruned.cc
// compiles against it as well.  Written from OpenFst's documented API (names, signatures, semantics); containers are plain
// std::vector.  TopSort, ArcSort, Invert and Connect are implemented (depth-first topological order / std::sort / label swap / trim,
// as documented); ShortestPath (single best path, label-correcting) for the lattice tools.
#ifndef K3_MINIFST_FSTLIB_H_
#define K3_MINIFST_FSTLIB_H_
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <list>
#include <map>
#include <queue>
#include <set>
#include <sstream>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define CHECK(x) do { if (!(x)) { std::cerr << "CHECK failed: " #x "\n"; std::abort(); } } while (0)

namespace fst {
using int32 = int32_t; using int64 = int64_t; using uint32 = uint32_t; using uint64 = uint64_t;
constexpr int kNoStateId = -1;
constexpr int kNoLabel = -1;
constexpr float kDelta = 1.0F / 1024.0F;
constexpr char kStringSeparator = '_';
constexpr uint64 kLeftSemiring = 0x1, kRightSemiring = 0x2, kSemiring = 0x3, kCommutative = 0x4, kIdempotent = 0x8, kPath = 0x10;
constexpr uint64 kExpanded = 0x1, kMutable = 0x2, kILabelSorted = 0x10000000ULL, kOLabelSorted = 0x40000000ULL, kTopSorted = 0x4000000000ULL, kIEpsilons = 0x0000000001000000ULL, kFstProperties = 0x0000ffffffff0007ULL;
inline std::string FST_FLAGS_fst_weight_separator = ",";
enum DivideType { DIVIDE_LEFT, DIVIDE_RIGHT, DIVIDE_ANY };

template <class T> struct FloatLimits {
  static constexpr T PosInfinity() { return std::numeric_limits<T>::infinity(); }
  static constexpr T NegInfinity() { return -std::numeric_limits<T>::infinity(); }
  static constexpr T NumberBad() { return std::numeric_limits<T>::quiet_NaN(); }
};
template <class T> std::istream &ReadType(std::istream &strm, T *t) { return strm.read(reinterpret_cast<char *>(t), sizeof(T)); }
template <class T> std::ostream &WriteType(std::ostream &strm, const T t) { return strm.write(reinterpret_cast<const char *>(&t), sizeof(T)); }

template <class W> class NaturalLess { public: bool operator()(const W &w1, const W &w2) const { return (Plus(w1, w2) == w1) && w1 != w2; } };
template <class W1, class W2> struct WeightConvert;

template <class T> class TropicalWeightTpl {
 public:
  using ValueType = T;
  TropicalWeightTpl() : value_(0) {}
  TropicalWeightTpl(T f) : value_(f) {}
  static TropicalWeightTpl Zero() { return TropicalWeightTpl(FloatLimits<T>::PosInfinity()); }
  static TropicalWeightTpl One() { return TropicalWeightTpl(0); }
  static const std::string &Type() { static const std::string t = "tropical"; return t; }
  const T &Value() const { return value_; }
 private:
  T value_;
};
template <class T> bool operator==(const TropicalWeightTpl<T> &a, const TropicalWeightTpl<T> &b) { return a.Value() == b.Value(); }
template <class T> bool operator!=(const TropicalWeightTpl<T> &a, const TropicalWeightTpl<T> &b) { return a.Value() != b.Value(); }
template <class T> TropicalWeightTpl<T> Plus(const TropicalWeightTpl<T> &a, const TropicalWeightTpl<T> &b) { return a.Value() < b.Value() ? a : b; }
template <class T> TropicalWeightTpl<T> Times(const TropicalWeightTpl<T> &a, const TropicalWeightTpl<T> &b) { return TropicalWeightTpl<T>(a.Value() + b.Value()); }
using TropicalWeight = TropicalWeightTpl<float>;

template <class W> struct ArcTpl {
  using Weight = W; using Label = int; using StateId = int;
  ArcTpl() {}
  ArcTpl(Label i, Label o, Weight w, StateId s) : ilabel(i), olabel(o), weight(w), nextstate(s) {}
  static const std::string &Type() { static const std::string t = W::Type() == "tropical" ? "standard" : W::Type(); return t; }
  Label ilabel; Label olabel; Weight weight; StateId nextstate;
};
using StdArc = ArcTpl<TropicalWeight>;

// read interface: what the decoder asks of a decoding graph
class SymbolTable;
template <class A> class Fst {
 public:
  using Arc = A; using StateId = typename A::StateId; using Weight = typename A::Weight;
  virtual ~Fst() {}
  virtual StateId Start() const = 0;
  virtual Weight Final(StateId s) const = 0;
  virtual size_t NumArcs(StateId s) const = 0;
  virtual size_t NumInputEpsilons(StateId s) const = 0;
  virtual const std::string &Type() const = 0;
  virtual uint64 Properties(uint64 mask, bool /*test*/) const { return mask & kExpanded; }
  virtual const SymbolTable *InputSymbols() const { return nullptr; }
  virtual const SymbolTable *OutputSymbols() const { return nullptr; }
  virtual const A *ArcsOf(StateId s) const = 0;          // (not OpenFst API: what this stand-in's ArcIterator reads)
  virtual Fst *Copy(bool safe = false) const = 0;
};
template <class A> class ExpandedFst : public Fst<A> {
 public:
  virtual typename A::StateId NumStates() const = 0;
  ExpandedFst *Copy(bool safe = false) const override = 0;
};
template <class A> class MutableFst : public ExpandedFst<A> {
 public:
  virtual typename A::StateId AddState() = 0;
  virtual void AddArc(typename A::StateId s, const A &arc) = 0;
  virtual void SetStart(typename A::StateId s) = 0;
  virtual void SetFinal(typename A::StateId s, typename A::Weight w) = 0;
  virtual void DeleteStates() = 0;
  virtual void SetInputSymbols(const SymbolTable *) {}
  virtual void SetOutputSymbols(const SymbolTable *) {}
  virtual void SetProperties(uint64 /*props*/, uint64 /*mask*/) {}
  virtual A *MutableArcsOf(typename A::StateId s) = 0;   // (not OpenFst API: behind MutableArcIterator and the algorithms below)
  virtual void ArcsChanged(typename A::StateId s, bool sorted_on_ilabel) = 0;
  virtual void ReplaceStates(const std::vector<typename A::StateId> &order) = 0;      // new state i = old state order[i]; states not listed are deleted
  MutableFst *Copy(bool safe = false) const override = 0;
};

template <class A> class VectorFst : public MutableFst<A> {
 public:
  using Arc = A; using StateId = typename A::StateId; using Weight = typename A::Weight;
  VectorFst() {}
  explicit VectorFst(const Fst<A> &f) { CopyFrom(f); }
  VectorFst &operator=(const Fst<A> &f) { if (&f != this) { states_.clear(); start_ = kNoStateId; ilabel_sorted_ = false; CopyFrom(f); } return *this; }
  VectorFst *Copy(bool = false) const override { return new VectorFst(*this); }
  StateId Start() const override { return start_; }
  Weight Final(StateId s) const override { return states_[s].final; }
  size_t NumArcs(StateId s) const override { return states_[s].arcs.size(); }
  size_t NumInputEpsilons(StateId s) const override { return states_[s].niepsilons; }
  const std::string &Type() const override { static const std::string t = "vector"; return t; }
  const A *ArcsOf(StateId s) const override { return states_[s].arcs.data(); }
  StateId NumStates() const override { return (StateId)states_.size(); }
  StateId AddState() override { states_.emplace_back(); return (StateId)states_.size() - 1; }
  void AddArc(StateId s, const A &arc) override { states_[s].arcs.push_back(arc); if (arc.ilabel == 0) states_[s].niepsilons++; ilabel_sorted_ = false; }
  A *MutableArcsOf(StateId s) override { return states_[s].arcs.data(); }
  void ArcsChanged(StateId s, bool sorted_on_ilabel) override {
    states_[s].niepsilons = 0; for (const A &a : states_[s].arcs) if (a.ilabel == 0) states_[s].niepsilons++;
    if (!sorted_on_ilabel) ilabel_sorted_ = false;
  }
  void MarkILabelSorted() { ilabel_sorted_ = true; }
  void ReplaceStates(const std::vector<StateId> &order) override {
    std::vector<StateId> newid(states_.size(), kNoStateId);
    for (size_t i = 0; i < order.size(); i++) newid[order[i]] = (StateId)i;
    std::vector<State> ns(order.size());
    for (size_t i = 0; i < order.size(); i++) {
      State &o = states_[order[i]]; ns[i].final = o.final;
      for (A &a : o.arcs) if (newid[a.nextstate] != kNoStateId) { a.nextstate = newid[a.nextstate]; ns[i].arcs.push_back(a); if (a.ilabel == 0) ns[i].niepsilons++; }
    }
    start_ = start_ == kNoStateId ? kNoStateId : newid[start_];
    states_.swap(ns);
  }
  // kExpanded / kMutable always; kILabelSorted as last established by ArcSort (or computed when `test`); kTopSorted only computed
  uint64 Properties(uint64 mask, bool test) const override {
    uint64 p = kExpanded | kMutable;
    bool ils = ilabel_sorted_;
    if (test) {
      bool top = true, ieps = false; ils = true;
      for (size_t s = 0; s < states_.size(); s++) for (size_t k = 0; k < states_[s].arcs.size(); k++) {
        if (states_[s].arcs[k].nextstate <= (StateId)s) top = false;
        if (states_[s].arcs[k].ilabel == 0) ieps = true;
        if (k && states_[s].arcs[k].ilabel < states_[s].arcs[k - 1].ilabel) ils = false;
      }
      if (top) p |= kTopSorted;
      if (ieps) p |= kIEpsilons;
    }
    if (ils) p |= kILabelSorted;
    return p & mask;
  }
  void SetStart(StateId s) override { start_ = s; }
  void SetFinal(StateId s, Weight w) override { states_[s].final = w; }
  void DeleteStates() override { states_.clear(); start_ = kNoStateId; }
  void ReserveStates(size_t n) { states_.reserve(n); }
  void ReserveArcs(StateId s, size_t n) { states_[s].arcs.reserve(n); }
 protected:
  void CopyFrom(const Fst<A> &f);
  struct State { Weight final = Weight::Zero(); std::vector<A> arcs; size_t niepsilons = 0; };
  std::vector<State> states_; StateId start_ = kNoStateId; bool ilabel_sorted_ = false;
};
template <class A> void VectorFst<A>::CopyFrom(const Fst<A> &f) {
  const auto *e = dynamic_cast<const ExpandedFst<A> *>(&f); CHECK(e != nullptr);
  for (StateId s = 0; s < e->NumStates(); s++) { AddState(); SetFinal(s, e->Final(s)); const A *a = e->ArcsOf(s); for (size_t k = 0; k < e->NumArcs(s); k++) AddArc(s, a[k]); }
  start_ = e->Start();
}
template <class A> class ConstFst : public VectorFst<A> {
 public:
  ConstFst() {}
  ConstFst *Copy(bool = false) const override { return new ConstFst(*this); }
  const std::string &Type() const override { static const std::string t = "const"; return t; }
};
using StdFst = Fst<StdArc>; using StdVectorFst = VectorFst<StdArc>; using StdConstFst = ConstFst<StdArc>;

template <class F> class ArcIterator {
 public:
  using Arc = typename F::Arc; using StateId = typename Arc::StateId;
  ArcIterator(const F &fst, StateId s) : arcs_(fst.ArcsOf(s)), n_(fst.NumArcs(s)), i_(0) {}
  bool Done() const { return i_ >= n_; }
  const Arc &Value() const { return arcs_[i_]; }
  void Next() { ++i_; }
  void Reset() { i_ = 0; }
  void Seek(size_t a) { i_ = a; }
  size_t Position() const { return i_; }
 private:
  const Arc *arcs_; size_t n_, i_;
};

template <class A> typename A::StateId CountStates(const Fst<A> &fst) {      // (OpenFst's StateIterator works on the base class too)
  const ExpandedFst<A> *e = dynamic_cast<const ExpandedFst<A> *>(&fst); CHECK(e != nullptr); return e->NumStates();
}
template <class F> class StateIterator {
 public:
  using StateId = typename F::Arc::StateId;
  explicit StateIterator(const F &fst) : n_(CountStates<typename F::Arc>(fst)), s_(0) {}
  bool Done() const { return s_ >= n_; }
  StateId Value() const { return s_; }
  void Next() { ++s_; }
 private:
  StateId n_, s_;
};
template <class F> class MutableArcIterator {
 public:
  using Arc = typename F::Arc; using StateId = typename Arc::StateId;
  MutableArcIterator(F *fst, StateId s) : fst_(fst), s_(s), n_(fst->NumArcs(s)), i_(0) {}
  bool Done() const { return i_ >= n_; }
  const Arc &Value() const { return fst_->ArcsOf(s_)[i_]; }
  void Next() { ++i_; }
  void SetValue(const Arc &a) { fst_->MutableArcsOf(s_)[i_] = a; fst_->ArcsChanged(s_, false); }
 private:
  F *fst_; StateId s_; size_t n_, i_;
};
template <class To, class From> To down_cast(From *f) { return static_cast<To>(f); }

template <class T> class MemoryPool {          // fst/memory.h: fixed-size object pool; here straight to the heap
 public:
  explicit MemoryPool(size_t /*pool_size*/ = 0) {}
  void *Allocate() { return ::operator new(sizeof(T)); }
  void Free(void *p) { ::operator delete(p); }
};

// names the reference's fstext/openfst_compat.h and lattice-weight.h mention in declarations that the decoder never uses
enum MapFinalAction { MAP_NO_SUPERFINAL, MAP_ALLOW_SUPERFINAL, MAP_REQUIRE_SUPERFINAL };
enum MapSymbolsAction { MAP_CLEAR_SYMBOLS, MAP_COPY_SYMBOLS, MAP_NOOP_SYMBOLS };
struct CacheOptions { CacheOptions(bool = false, size_t = 0) {} };
struct ArcMapFstOptions { ArcMapFstOptions() {} explicit ArcMapFstOptions(const CacheOptions &) {} };
template <class A, class B, class C> class ArcMapFst;
template <class W1, class W2> class PairWeight {
 public:
  PairWeight() {}
  PairWeight(W1 w1, W2 w2) : value1_(w1), value2_(w2) {}
  const W1 &Value1() const { return value1_; }
  const W2 &Value2() const { return value2_; }
 private:
  W1 value1_; W2 value2_;
};

template <class A> struct ILabelCompare { bool operator()(const A &a, const A &b) const { return a.ilabel < b.ilabel; } };
[[noreturn]] inline void NotInStandIn(const char *what) { std::cerr << what << " is not part of the OpenFst stand-in (oracle/ref_tools/minifst)\n"; std::abort(); }
// single best path (fst/shortest-path.h with n = 1) for weights with a natural order (Plus(a, b) is a or b): label-correcting search from the start state,
// result = a linear FST from its start state to one final state carrying the path's arcs and the final weight
template <class A> void ShortestPath(const Fst<A> &fst, MutableFst<A> *out) {
  using StateId = typename A::StateId; using W = typename A::Weight;
  out->DeleteStates();
  const auto *e = dynamic_cast<const ExpandedFst<A> *>(&fst); CHECK(e != nullptr);
  const StateId n = e->NumStates(); if (n == 0 || fst.Start() == kNoStateId) return;
  NaturalLess<W> less; std::vector<W> dist(n, W::Zero()); std::vector<StateId> prev(n, kNoStateId); std::vector<size_t> parc(n, 0); std::vector<char> inq(n, 0); std::vector<StateId> q;
  dist[fst.Start()] = W::One(); q.push_back(fst.Start()); inq[fst.Start()] = 1;
  for (size_t h = 0; h < q.size(); h++) {
    const StateId s = q[h]; inq[s] = 0;
    for (size_t k = 0; k < fst.NumArcs(s); k++) {
      const A &a = fst.ArcsOf(s)[k]; const W w = Times(dist[s], a.weight);
      if (less(w, dist[a.nextstate])) { dist[a.nextstate] = w; prev[a.nextstate] = s; parc[a.nextstate] = k; if (!inq[a.nextstate]) { inq[a.nextstate] = 1; q.push_back(a.nextstate); } }
    }
  }
  StateId best = kNoStateId; W bw = W::Zero();
  for (StateId s = 0; s < n; s++) if (fst.Final(s) != W::Zero() && dist[s] != W::Zero()) { const W w = Times(dist[s], fst.Final(s)); if (best == kNoStateId || less(w, bw)) { best = s; bw = w; } }
  if (best == kNoStateId) return;
  std::vector<StateId> path; for (StateId s = best; s != kNoStateId; s = prev[s]) path.push_back(s);
  std::reverse(path.begin(), path.end());
  for (size_t i = 0; i < path.size(); i++) out->AddState();
  out->SetStart(0);
  for (size_t i = 0; i + 1 < path.size(); i++) { A a = fst.ArcsOf(path[i])[parc[path[i + 1]]]; a.nextstate = (StateId)i + 1; out->AddArc((StateId)i, a); }
  out->SetFinal((StateId)path.size() - 1, fst.Final(best));
}
template <class A> void Invert(MutableFst<A> *f) {
  for (typename A::StateId s = 0; s < f->NumStates(); s++) { A *a = f->MutableArcsOf(s); for (size_t k = 0; k < f->NumArcs(s); k++) std::swap(a[k].ilabel, a[k].olabel); f->ArcsChanged(s, false); }
}
template <class A, class C> void ArcSort(MutableFst<A> *f, C comp) {
  for (typename A::StateId s = 0; s < f->NumStates(); s++) { A *a = f->MutableArcsOf(s); std::sort(a, a + f->NumArcs(s), comp); f->ArcsChanged(s, true); }
  if (auto *v = dynamic_cast<VectorFst<A> *>(f)) v->MarkILabelSorted();
}
// fst/dfs-visit.h: depth-first traversal driven by a visitor (InitVisit, InitState(s, root), TreeArc / BackArc / ForwardOrCrossArc(s, arc),
// FinishState(s, parent, parent_arc), FinishVisit); roots are the start state, then every state not reached yet in numeric order.
template <class F, class Visitor> void DfsVisit(const F &fst, Visitor *visitor) {
  using Arc = typename F::Arc; using StateId = typename Arc::StateId;
  visitor->InitVisit(fst);
  const auto *e = dynamic_cast<const ExpandedFst<Arc> *>(&fst); CHECK(e != nullptr);
  const StateId n = e->NumStates();
  if (fst.Start() == kNoStateId) { visitor->FinishVisit(); return; }
  std::vector<char> color(n, 0); std::vector<size_t> pos(n, 0); std::vector<StateId> stack;
  bool go = true;
  auto run = [&](StateId root) {
    color[root] = 1; stack.push_back(root); go = visitor->InitState(root, root);
    while (!stack.empty()) {
      const StateId s = stack.back();
      if (!go || pos[s] >= fst.NumArcs(s)) {
        color[s] = 2; stack.pop_back();
        if (!stack.empty()) { const StateId p = stack.back(); visitor->FinishState(s, p, &fst.ArcsOf(p)[pos[p] - 1]); } else visitor->FinishState(s, kNoStateId, nullptr);
        continue;
      }
      const Arc &arc = fst.ArcsOf(s)[pos[s]++];
      const StateId d = arc.nextstate;
      if (color[d] == 0) { go = visitor->TreeArc(s, arc); if (!go) continue; color[d] = 1; stack.push_back(d); go = visitor->InitState(d, root); }
      else if (color[d] == 1) go = visitor->BackArc(s, arc);
      else go = visitor->ForwardOrCrossArc(s, arc);
    }
  };
  run(fst.Start());
  for (StateId s = 0; s < n && go; s++) if (color[s] == 0) run(s);
  visitor->FinishVisit();
}
// depth-first search from the start state, then from every state not reached yet in numeric order; new numbering = reverse finishing
// order (fst/topsort.h).  false (and the FST untouched) when there is a cycle.
template <class A> bool TopSort(MutableFst<A> *f) {
  using StateId = typename A::StateId;
  const StateId n = f->NumStates();
  if (n == 0) return true;
  std::vector<char> color(n, 0); std::vector<size_t> pos(n, 0); std::vector<StateId> stack, finish;
  auto visit = [&](StateId root) {
    stack.push_back(root); color[root] = 1;
    while (!stack.empty()) {
      const StateId s = stack.back();
      if (pos[s] < f->NumArcs(s)) {
        const StateId d = f->ArcsOf(s)[pos[s]++].nextstate;
        if (color[d] == 1) return false;
        if (color[d] == 0) { color[d] = 1; stack.push_back(d); }
      } else { color[s] = 2; finish.push_back(s); stack.pop_back(); }
    }
    return true;
  };
  if (f->Start() != kNoStateId && !visit(f->Start())) return false;
  for (StateId s = 0; s < n; s++) if (color[s] == 0 && !visit(s)) return false;
  std::vector<StateId> order(finish.rbegin(), finish.rend());
  f->ReplaceStates(order);
  return true;
}
// keeps the states that are reachable from the start state and from which a final state is reachable, in their old relative order
template <class A> void Connect(MutableFst<A> *f) {
  using StateId = typename A::StateId; using Weight = typename A::Weight;
  const StateId n = f->NumStates();
  if (n == 0) return;
  std::vector<char> acc(n, 0), co(n, 0); std::vector<StateId> st; std::vector<std::vector<StateId>> rev(n);
  for (StateId s = 0; s < n; s++) for (size_t k = 0; k < f->NumArcs(s); k++) rev[f->ArcsOf(s)[k].nextstate].push_back(s);
  if (f->Start() != kNoStateId) { acc[f->Start()] = 1; st.push_back(f->Start()); }
  while (!st.empty()) { const StateId s = st.back(); st.pop_back(); for (size_t k = 0; k < f->NumArcs(s); k++) { const StateId d = f->ArcsOf(s)[k].nextstate; if (!acc[d]) { acc[d] = 1; st.push_back(d); } } }
  for (StateId s = 0; s < n; s++) if (f->Final(s) != Weight::Zero()) { co[s] = 1; st.push_back(s); }
  while (!st.empty()) { const StateId s = st.back(); st.pop_back(); for (StateId p : rev[s]) if (!co[p]) { co[p] = 1; st.push_back(p); } }
  std::vector<StateId> order; for (StateId s = 0; s < n; s++) if (acc[s] && co[s]) order.push_back(s);
  if ((StateId)order.size() == n) return;
  if (order.empty() || !(acc[f->Start()] && co[f->Start()])) { f->DeleteStates(); return; }
  f->ReplaceStates(order);
}
}  // namespace fst
#endif
