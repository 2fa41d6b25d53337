// oracle/ref_tools/dump_tid2pdf.cc -- TEST INFRASTRUCTURE.  Links the REFERENCE's own TransitionModel class (oracle/_ref/libref.a,
// compiled from /root/reference/src/hmm/transition-model.cc) and prints, for a .mdl file, NumPdfs, NumTransitionIds and
// TransitionIdToPdf(t) for every transition-id: the truth kaldi_amd/host's own TransitionModel parser is tested against.
#include <iostream>
#include "hmm/transition-model.h"
#include "util/common-utils.h"
int main(int argc, char **argv) {
  using namespace kaldi;
  if (argc != 2) { std::cerr << "usage: dump-tid2pdf <model.mdl>\n"; return 1; }
  try {
    TransitionModel tm; bool binary; Input ki(argv[1], &binary); tm.Read(ki.Stream(), binary);
    std::cout << tm.NumPdfs() << " " << tm.NumTransitionIds() << "\n";
    for (int32 t = 1; t <= tm.NumTransitionIds(); t++) std::cout << tm.TransitionIdToPdf(t) << (t % 32 ? " " : "\n");
    std::cout << "\n";
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
  return 0;
}
