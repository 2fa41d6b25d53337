// lattice-determinize-phone-pruned -- same command line as the reference's latbin/lattice-determinize-phone-pruned.cc:28-160:
//   lattice-determinize-phone-pruned [options] <model> <lattice-rspecifier> <lattice-wspecifier>
// The determinization the decoders apply (DeterminizeLatticePhonePrunedWrapper): a first pass with phone labels inserted at the phone
// boundaries, found with the model's transition-id -> phone map, then the word-level pass.  Host-only.
// Not implemented: --write-compact=false (rejected, not ignored).
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "lattice-determinize-phone-pruned";
    const char *usage =
        "Determinize lattices, keeping only the best path (sequence of\n"
        "acoustic states) for each input-symbol sequence. This version does\n"
        "phone insertion when doing a first pass determinization, it then\n"
        "removes the inserted symbols and does a second pass determinization.\n"
        "It also does pruning as part of the determinization algorithm, which\n"
        "is more efficient and prevents blowup.\n"
        "\n"
        "Usage: lattice-determinize-phone-pruned [options] <model> \\\n"
        "                  <lattice-rspecifier> <lattice-wspecifier>\n"
        " e.g.: lattice-determinize-phone-pruned --acoustic-scale=0.1 \\\n"
        "                            final.mdl ark:in.lats ark:det.lats\n";
    ParseOptions po(usage);
    bool write_compact = true; float acoustic_scale = 1.0f, beam = 10.0f;
    DeterminizeLatticePhonePrunedOptions opts; opts.max_mem = 50000000;
    po.Register("write-compact", &write_compact, "If true, write in normal (compact) form (only true is supported by this build)");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods.");
    po.Register("beam", &beam, "Pruning beam [applied after acoustic scaling].");
    po.Register("delta", &opts.delta, "Tolerance used in determinization");
    po.Register("max-mem", &opts.max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &opts.phone_determinize, "If true, do an initial pass of determinization on both phones and words (see also --word-determinize)");
    po.Register("word-determinize", &opts.word_determinize, "If true, do a second pass of determinization on words only (see also --phone-determinize)");
    po.Register("minimize", &opts.minimize, "If true, push and minimize after determinization.");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    if (!write_compact) K3H_ERR << "--write-compact=false is not supported";
    if (acoustic_scale == 0.0f) K3H_ERR << "Do not use a zero acoustic scale (cannot be inverted)";
    const TransitionInfo trans = ReadTransitionModel(po.GetArg(1));
    auto lats = ReadLatticeTable(po.GetArg(2));
    TableWriter writer(po.GetArg(3));
    int32_t n_done = 0, n_warn = 0;
    for (auto &kv : lats) {
      Lattice &lat = kv.second;
      ScaleAcoustic(&lat, acoustic_scale);
      CompactLattice clat;
      if (!DeterminizeLatticePhonePruned(lat, trans, beam, &clat, opts)) {
        K3H_WARN << "For key " << kv.first << ", determinization did not succeed(partial output will be pruned tighter than the specified beam.)";
        n_warn++;
      }
      if (!TopSortIfNeeded(&clat)) K3H_WARN << "Topological sorting of the determinized lattice failed for key " << kv.first;
      ScaleAcoustic(&clat, 1.0 / acoustic_scale);
      writer.WriteCompactLattice(kv.first, clat);
      n_done++;
    }
    writer.Flush();
    K3H_LOG << "Done " << n_done << " lattices, determinization finished earlier than specified by the beam on " << n_warn << " of these.";
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
