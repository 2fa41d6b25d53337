// oracle/ref_tools/dump_tidinfo.cc -- TEST INFRASTRUCTURE.  Links the REFERENCE's own TransitionModel class (oracle/_ref/libref.a) and
// prints per transition-id: TransitionIdToPhone, IsSelfLoop, TransitionIdIsStartOfPhone -- what DeterminizeLatticeInsertPhones
// (lat/determinize-lattice-pruned.cc:1291-1343) asks of it; kaldi_amd/host's parser is tested against this.
#include <iostream>
#include "hmm/transition-model.h"
#include "util/common-utils.h"
int main(int argc, char **argv) {
  using namespace kaldi;
  if (argc != 2) { std::cerr << "usage: dump-tidinfo <model.mdl>\n"; return 1; }
  try {
    TransitionModel tm; bool binary; Input ki(argv[1], &binary); tm.Read(ki.Stream(), binary);
    for (int32 t = 1; t <= tm.NumTransitionIds(); t++)
      std::cout << t << " " << tm.TransitionIdToPhone(t) << " " << (int)tm.IsSelfLoop(t) << " " << (int)tm.TransitionIdIsStartOfPhone(t) << "\n";
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
  return 0;
}
