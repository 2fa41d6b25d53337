// TEST INFRASTRUCTURE: stands in for the reference's fstext/fstext-utils.h (a large collection of OpenFst extensions) with the one
// function lat/determinize-lattice-pruned.cc takes from it; ConvertLattice comes from the reference's own fstext/lattice-utils.h.
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
}  // namespace fst
#include "fstext/lattice-utils.h"     // the reference's: ConvertLattice (+ Factor), used by DeterminizeLatticePhonePruned when --word-determinize=false
#endif
