// TEST INFRASTRUCTURE: declarations only, so that chain/chain-den-graph.cc compiles UNMODIFIED against the OpenFst stand-in.  The oracle tool
// uses DenominatorGraph's constructor (SetTransitions / SetInitialProbs, which only iterate over a vector FST); the graph-compilation
// functions of that file (CreateDenominatorFst ...) need real OpenFst, are never called, and their callees below are never defined
// (the tools link with --unresolved-symbols=ignore-all).
#ifndef K3_MINIFST_HMM_UTILS_H_
#define K3_MINIFST_HMM_UTILS_H_
#include <vector>
#include "fst/fstlib.h"
#include "fstext/deterministic-fst.h"
#include "hmm/transition-model.h"
#include "tree/context-dep.h"
namespace fst {
template <class F> void RmEpsilon(F *);
template <class A> struct QuantizeMapper { explicit QuantizeMapper(float); };
template <class F, class M> void ArcMap(F *, M);
enum EncodeType { ENCODE = 1, DECODE = 2 };
const unsigned kEncodeLabels = 1, kEncodeWeights = 2;
template <class A> struct EncodeMapper { EncodeMapper(unsigned, EncodeType); };
template <class F, class A> void Encode(F *, EncodeMapper<A> *);
template <class F, class A> void Decode(F *, const EncodeMapper<A> &);
namespace internal { template <class F> void AcceptorMinimize(F *); }
template <class F, class V> void StateSort(F *, const V &);
template <class F1, class F2> void Reverse(const F1 &, F2 *);
enum ProjectType { PROJECT_INPUT = 1, PROJECT_OUTPUT = 2 };
template <class F> void Project(F *, ProjectType);
template <class F> void RemoveEpsLocal(F *);
template <class F> long NumArcs(const F &);
template <class F> void AddSubsequentialLoop(int, F *);
template <class A, class B, class C> void TableCompose(const A &, const B &, C *);
class InverseContextFst : public DeterministicOnDemandFst<StdArc> {
 public:
  InverseContextFst(int, const std::vector<int> &, const std::vector<int> &, int, int);
  const std::vector<std::vector<int> > &IlabelInfo() const;
  StateId Start(); Weight Final(StateId); bool GetArc(StateId, Label, StdArc *);
};
template <class A, class B, class C> void ComposeDeterministicOnDemandInverse(const A &, B *, C *);
}
namespace kaldi {
struct HTransducerConfig { float transition_scale; HTransducerConfig(): transition_scale(1.0f) {} };
fst::VectorFst<fst::StdArc> *GetHTransducer(const std::vector<std::vector<int> > &, const ContextDependencyInterface &, const TransitionModel &, const HTransducerConfig &, std::vector<int> *);
void AddSelfLoops(const TransitionModel &, const std::vector<int> &, float, bool, bool, fst::VectorFst<fst::StdArc> *);
}
#endif
