// TEST INFRASTRUCTURE: the option structs of the reference's lat/determinize-lattice-pruned.h (LatticeFasterDecoderConfig embeds
// one and registers its options) and declarations of the two functions LatticeFasterDecoderTpl::GetLattice mentions.  The
// determinizer itself needs all of OpenFst and is not built; GetLattice is never called by the oracle driver.
#ifndef K3_MINIFST_DET_LATTICE_PRUNED_H_
#define K3_MINIFST_DET_LATTICE_PRUNED_H_
#include "fst/fstlib.h"
#include "itf/options-itf.h"
#include "lat/kaldi-lattice.h"
namespace fst {
struct DeterminizeLatticePrunedOptions {
  float delta = kDelta; int max_mem = -1, max_loop = -1, max_states = -1, max_arcs = -1; float retry_cutoff = 0.5;
  void Register(kaldi::OptionsItf *) {}
};
struct DeterminizeLatticePhonePrunedOptions {
  float delta = kDelta; int max_mem = 50000000; bool phone_determinize = true, word_determinize = true, minimize = false;
  void Register(kaldi::OptionsItf *opts) { opts->Register("delta", &delta, ""); opts->Register("max-mem", &max_mem, ""); opts->Register("phone-determinize", &phone_determinize, "");
                                           opts->Register("word-determinize", &word_determinize, ""); opts->Register("minimize", &minimize, ""); }
};
template <class Weight> bool DeterminizeLatticePruned(const ExpandedFst<ArcTpl<Weight>> &, double, MutableFst<ArcTpl<CompactLatticeWeightTpl<Weight, int>>> *, DeterminizeLatticePrunedOptions = DeterminizeLatticePrunedOptions()) {
  NotInStandIn("DeterminizeLatticePruned");
}
}  // namespace fst
#endif
