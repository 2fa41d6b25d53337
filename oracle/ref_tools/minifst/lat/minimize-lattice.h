// TEST INFRASTRUCTURE: declaration only (used by determinize-lattice-pruned.cc when --minimize=true, which the oracle driver never sets).
#ifndef K3_MINIFST_MINIMIZE_LATTICE_H_
#define K3_MINIFST_MINIMIZE_LATTICE_H_
#include "lat/kaldi-lattice.h"
namespace fst {
template <class Weight, class IntType> bool MinimizeCompactLattice(MutableFst<ArcTpl<CompactLatticeWeightTpl<Weight, IntType>>> *, float = 1.0e-04) { NotInStandIn("MinimizeCompactLattice"); }
}
#endif
