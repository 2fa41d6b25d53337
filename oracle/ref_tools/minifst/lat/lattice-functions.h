// TEST INFRASTRUCTURE: lattice-faster-decoder.cc includes lat/lattice-functions.h and uses nothing from it;
// determinize-lattice-pruned.cc takes PruneLattice from it (only on its retry path, after --max-mem stopped a pass early).
// The real header drags in the whole of fstext; this PruneLattice follows the documented behaviour (lat/lattice-functions.h:
// "prunes a lattice or compact lattice", forward-backward over a topologically sorted lattice, then trims).
#ifndef K3_MINIFST_LATTICE_FUNCTIONS_H_
#define K3_MINIFST_LATTICE_FUNCTIONS_H_
#include <limits>
#include <vector>
#include "lat/kaldi-lattice.h"
#include "fstext/fstext-utils.h"
namespace kaldi {
using fst::ConvertLattice; using fst::RemoveAlignmentsFromCompactLattice; using fst::GetLinearSymbolSequence; using fst::CreateSuperFinal;
int32 CompactLatticeStateTimes(const CompactLattice &clat, std::vector<int32> *times);      // lat/lattice-functions.cc:109 (defined by the tool that links lat/sausages.cc)
template <class LatType> bool PruneLattice(BaseFloat beam, LatType *lat) {
  typedef typename LatType::Arc Arc; typedef typename Arc::Weight Weight;
  if (!lat->Properties(fst::kTopSorted, true) && !fst::TopSort(lat)) return false;
  const int32 n = lat->NumStates(); if (n == 0) return false;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> fwd(n, inf), bwd(n, inf); fwd[lat->Start()] = 0.0; double best = inf;
  for (int32 s = 0; s < n; s++) {
    for (fst::ArcIterator<LatType> it(*lat, s); !it.Done(); it.Next()) { const double c = fwd[s] + ConvertToCost(it.Value().weight); if (c < fwd[it.Value().nextstate]) fwd[it.Value().nextstate] = c; }
    best = std::min(best, fwd[s] + ConvertToCost(lat->Final(s)));
  }
  const double cutoff = best + beam;
  const int32 bad = lat->AddState();
  for (int32 s = n - 1; s >= 0; s--) {
    double b = ConvertToCost(lat->Final(s));
    if (b + fwd[s] > cutoff && b != inf) lat->SetFinal(s, Weight::Zero());
    for (fst::MutableArcIterator<LatType> it(lat, s); !it.Done(); it.Next()) {
      Arc arc = it.Value(); const double ab = ConvertToCost(arc.weight) + bwd[arc.nextstate];
      if (ab < b) b = ab;
      if (fwd[s] + ab > cutoff) { arc.nextstate = bad; it.SetValue(arc); }
    }
    bwd[s] = b;
  }
  fst::Connect(lat);
  return lat->NumStates() > 0;
}
}  // namespace kaldi
#endif
