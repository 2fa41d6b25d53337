// oracle/ref_tools/minifst/fst/fstlib.h -- TEST INFRASTRUCTURE.  A stand-in for the part of OpenFst's public interface that the
// reference's decoder/lattice-faster-decoder.{h,cc} and fstext/lattice-weight.h touch, so that those reference sources can be
// compiled UNMODIFIED into oracle/_ref without OpenFst (which /root/reference does not vendor).  Written from OpenFst's documented
// API (names, signatures, semantics); containers are plain std::vector.  Only what the decoder needs is real; the FST algorithms its
// other member functions mention (ShortestPath, Invert, ArcSort, Connect) are declared and abort when called.
#ifndef K3_MINIFST_FSTLIB_H_
#define K3_MINIFST_FSTLIB_H_
#include <cstdint>
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
constexpr uint64 kExpanded = 0x1, kMutable = 0x2, kILabelSorted = 0x10000000ULL, kTopSorted = 0x4000000000ULL;
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
  virtual const A *ArcsOf(StateId s) const = 0;          // (not OpenFst API: what this stand-in's ArcIterator reads)
};
template <class A> class ExpandedFst : public Fst<A> { public: virtual typename A::StateId NumStates() const = 0; };
template <class A> class MutableFst : public ExpandedFst<A> {
 public:
  virtual typename A::StateId AddState() = 0;
  virtual void AddArc(typename A::StateId s, const A &arc) = 0;
  virtual void SetStart(typename A::StateId s) = 0;
  virtual void SetFinal(typename A::StateId s, typename A::Weight w) = 0;
  virtual void DeleteStates() = 0;
};

template <class A> class VectorFst : public MutableFst<A> {
 public:
  using Arc = A; using StateId = typename A::StateId; using Weight = typename A::Weight;
  VectorFst() {}
  explicit VectorFst(const Fst<A> &f) { CopyFrom(f); }
  StateId Start() const override { return start_; }
  Weight Final(StateId s) const override { return states_[s].final; }
  size_t NumArcs(StateId s) const override { return states_[s].arcs.size(); }
  size_t NumInputEpsilons(StateId s) const override { return states_[s].niepsilons; }
  const std::string &Type() const override { static const std::string t = "vector"; return t; }
  const A *ArcsOf(StateId s) const override { return states_[s].arcs.data(); }
  StateId NumStates() const override { return (StateId)states_.size(); }
  StateId AddState() override { states_.emplace_back(); return (StateId)states_.size() - 1; }
  void AddArc(StateId s, const A &arc) override { states_[s].arcs.push_back(arc); if (arc.ilabel == 0) states_[s].niepsilons++; }
  void SetStart(StateId s) override { start_ = s; }
  void SetFinal(StateId s, Weight w) override { states_[s].final = w; }
  void DeleteStates() override { states_.clear(); start_ = kNoStateId; }
  void ReserveStates(size_t n) { states_.reserve(n); }
  void ReserveArcs(StateId s, size_t n) { states_[s].arcs.reserve(n); }
 protected:
  void CopyFrom(const Fst<A> &f);
  struct State { Weight final = Weight::Zero(); std::vector<A> arcs; size_t niepsilons = 0; };
  std::vector<State> states_; StateId start_ = kNoStateId;
};
template <class A> class ConstFst : public VectorFst<A> {
 public:
  ConstFst() {}
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

template <class T> class MemoryPool {          // fst/memory.h: fixed-size object pool; here straight to the heap
 public:
  explicit MemoryPool(size_t /*pool_size*/ = 0) {}
  void *Allocate() { return ::operator new(sizeof(T)); }
  void Free(void *p) { ::operator delete(p); }
};

// names the reference's fstext/openfst_compat.h and lattice-weight.h mention in declarations that the decoder never uses
struct ArcMapFstOptions {};
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
template <class A> void ShortestPath(const Fst<A> &, MutableFst<A> *) { NotInStandIn("ShortestPath"); }
template <class A> void Invert(MutableFst<A> *) { NotInStandIn("Invert"); }
template <class A, class C> void ArcSort(MutableFst<A> *, C) { NotInStandIn("ArcSort"); }
template <class A> void Connect(MutableFst<A> *) { NotInStandIn("Connect"); }
}  // namespace fst
#endif
