// TEST INFRASTRUCTURE: the type definitions of the reference's lat/kaldi-lattice.h:40-52 over the stand-in containers (the
// real header also declares the table I/O classes, which need the rest of OpenFst).
#ifndef K3_MINIFST_KALDI_LATTICE_H_
#define K3_MINIFST_KALDI_LATTICE_H_
#include "fstext/fstext-lib.h"
#include "base/kaldi-common.h"
namespace kaldi {
typedef fst::LatticeWeightTpl<BaseFloat> LatticeWeight;
typedef fst::CompactLatticeWeightTpl<LatticeWeight, int32> CompactLatticeWeight;
typedef fst::ArcTpl<LatticeWeight> LatticeArc;
typedef fst::ArcTpl<CompactLatticeWeight> CompactLatticeArc;
typedef fst::VectorFst<LatticeArc> Lattice;
typedef fst::VectorFst<CompactLatticeArc> CompactLattice;
}
#endif
