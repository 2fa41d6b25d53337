// TEST INFRASTRUCTURE: the lattice type names of the reference's lat/kaldi-lattice.h (:40-52) bound to the stand-in containers.
// (The real header also declares the table I/O classes, which need the rest of OpenFst.)
#ifndef K3_MINIFST_KALDI_LATTICE_H_
#define K3_MINIFST_KALDI_LATTICE_H_
#include "fstext/fstext-lib.h"
#include "base/kaldi-common.h"
namespace kaldi {
using LatticeWeight = fst::LatticeWeightTpl<BaseFloat>;                       // (graph cost, acoustic cost)
using CompactLatticeWeight = fst::CompactLatticeWeightTpl<LatticeWeight, int32>;   // + transition-id string
using LatticeArc = fst::ArcTpl<LatticeWeight>;
using CompactLatticeArc = fst::ArcTpl<CompactLatticeWeight>;
using Lattice = fst::VectorFst<LatticeArc>;
using CompactLattice = fst::VectorFst<CompactLatticeArc>;
}  // namespace kaldi
#endif
