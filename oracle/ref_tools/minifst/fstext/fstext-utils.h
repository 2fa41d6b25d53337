// TEST INFRASTRUCTURE: stands in for the reference's fstext/fstext-utils.h (a large collection of OpenFst extensions) with the
// functions lat/determinize-lattice-pruned.cc and lat/sausages.cc take from it; ConvertLattice comes from the reference's own fstext/lattice-utils.h.
#ifndef K3_MINIFST_FSTEXT_UTILS_H_
#define K3_MINIFST_FSTEXT_UTILS_H_
#include "fst/fstlib.h"
#include "fstext/lattice-weight.h"
#include "util/stl-utils.h"      // the reference's: brings unordered_map / unordered_set into scope like the real fstext-utils.h does
namespace fst {
template <class Arc> typename Arc::Label HighestNumberedInputSymbol(const Fst<Arc> &f) {       // fstext/fstext-utils-inl.h: max ilabel over all arcs, 0 if none
  typename Arc::Label ans = 0;
  const auto *e = dynamic_cast<const ExpandedFst<Arc> *>(&f); CHECK(e != nullptr);
  for (typename Arc::StateId s = 0; s < e->NumStates(); s++) for (size_t k = 0; k < e->NumArcs(s); k++) ans = std::max(ans, e->ArcsOf(s)[k].ilabel);
  return ans;
}
// fstext/fstext-utils-inl.h:221-260: the labels and total weight along a linear FST (false when the FST is not linear)
template <class Arc, class I> bool GetLinearSymbolSequence(const Fst<Arc> &f, std::vector<I> *isymbols_out, std::vector<I> *osymbols_out, typename Arc::Weight *tot_weight_out) {
  typedef typename Arc::Weight Weight;
  Weight tot = Weight::One(); std::vector<I> il, ol; typename Arc::StateId cur = f.Start();
  if (cur == kNoStateId) { if (isymbols_out) isymbols_out->clear(); if (osymbols_out) osymbols_out->clear(); if (tot_weight_out) *tot_weight_out = Weight::Zero(); return true; }
  while (1) {
    const Weight w = f.Final(cur);
    if (w != Weight::Zero()) {
      tot = Times(tot, w); if (f.NumArcs(cur) != 0) return false;
      if (isymbols_out) *isymbols_out = il; if (osymbols_out) *osymbols_out = ol; if (tot_weight_out) *tot_weight_out = tot;
      return true;
    }
    if (f.NumArcs(cur) != 1) return false;
    const Arc &arc = f.ArcsOf(cur)[0]; tot = Times(tot, arc.weight);
    if (arc.ilabel != 0) il.push_back(arc.ilabel); if (arc.olabel != 0) ol.push_back(arc.olabel);
    cur = arc.nextstate;
  }
}
// fstext/pre-determinize-inl.h:689-724: one final state (weight One, no arcs out), reached by epsilon arcs that carry the old final weights
template <class Arc> typename Arc::StateId CreateSuperFinal(MutableFst<Arc> *fst) {
  typedef typename Arc::StateId StateId; typedef typename Arc::Weight Weight;
  const StateId n = fst->NumStates(); std::vector<StateId> finals;
  for (StateId s = 0; s < n; s++) if (fst->Final(s) != Weight::Zero()) finals.push_back(s);
  if (finals.size() == 1 && fst->Final(finals[0]) == Weight::One() && fst->NumArcs(finals[0]) == 0) return finals[0];
  const StateId fs = fst->AddState(); fst->SetFinal(fs, Weight::One());
  for (StateId s : finals) { const Weight w = fst->Final(s); fst->SetFinal(s, Weight::Zero()); Arc arc; arc.ilabel = 0; arc.olabel = 0; arc.nextstate = fs; arc.weight = w; fst->AddArc(s, arc); }
  return fs;
}
}  // namespace fst
#include "fstext/lattice-utils.h"     // the reference's: ConvertLattice (+ Factor), used by DeterminizeLatticePhonePruned when --word-determinize=false
#endif
