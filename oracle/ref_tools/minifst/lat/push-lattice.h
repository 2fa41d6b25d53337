// TEST INFRASTRUCTURE: declarations only (used by determinize-lattice-pruned.cc when --minimize=true, which the oracle driver never sets).
#ifndef K3_MINIFST_PUSH_LATTICE_H_
#define K3_MINIFST_PUSH_LATTICE_H_
#include "lat/kaldi-lattice.h"
namespace fst {
template <class Weight, class IntType> bool PushCompactLatticeStrings(MutableFst<ArcTpl<CompactLatticeWeightTpl<Weight, IntType>>> *) { NotInStandIn("PushCompactLatticeStrings"); }
template <class Weight, class IntType> bool PushCompactLatticeWeights(MutableFst<ArcTpl<CompactLatticeWeightTpl<Weight, IntType>>> *) { NotInStandIn("PushCompactLatticeWeights"); }
}
#endif
