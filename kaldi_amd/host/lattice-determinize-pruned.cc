// lattice-determinize-pruned -- same command line as the reference's latbin/lattice-determinize-pruned.cc:28-170: reads state-level
// lattices (what the decoders here write with --determinize-lattice=false), scales the acoustic costs, determinizes on the word
// labels with pruning, writes CompactLattices with the acoustic scale undone.  Host-only (no GPU work on this path).
// Not implemented: --write-compact=false (rejected, not ignored).
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "lattice-determinize-pruned";
    const char *usage =
        "Determinize lattices, keeping only the best path (sequence of acoustic states)\n"
        "for each input-symbol sequence.  This version does pruning as part of the\n"
        "determinization algorithm, which is more efficient and prevents blowup.\n"
        "\n"
        "Usage: lattice-determinize-pruned [options] lattice-rspecifier lattice-wspecifier\n"
        " e.g.: lattice-determinize-pruned --acoustic-scale=0.1 --beam=6.0 ark:in.lats ark:det.lats\n";
    ParseOptions po(usage);
    bool write_compact = true, minimize = false; float acoustic_scale = 1.0f, beam = 10.0f;
    DeterminizeLatticePrunedOptions opts; opts.max_mem = 50000000; opts.max_loop = 0;
    po.Register("write-compact", &write_compact, "If true, write in normal (compact) form (only true is supported by this build)");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("beam", &beam, "Pruning beam [applied after acoustic scaling].");
    po.Register("minimize", &minimize, "If true, push and minimize after determinization");
    po.Register("delta", &opts.delta, "Tolerance used in determinization");
    po.Register("max-mem", &opts.max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this)");
    po.Register("max-arcs", &opts.max_arcs, "Maximum number of arcs in output FST (total, not per state");
    po.Register("max-states", &opts.max_states, "Maximum number of arcs in output FST (total, not per state");
    po.Register("max-loop", &opts.max_loop, "Option used to detect a particular type of determinization failure, typically due to invalid input (e.g., negative-cost loops)");
    po.Register("retry-cutoff", &opts.retry_cutoff,
        "Controls pruning un-determinized lattice and retrying determinization: if effective-beam < retry-cutoff * beam, we prune the raw lattice and retry.");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    if (!write_compact) K3H_ERR << "--write-compact=false is not supported";
    if (acoustic_scale == 0.0f) K3H_ERR << "Do not use a zero acoustic scale (cannot be inverted)";
    auto lats = ReadLatticeTable(po.GetArg(1));
    TableWriter writer(po.GetArg(2));
    int32_t n_done = 0, n_warn = 0; double states_in = 0, arcs_out = 0;
    for (auto &kv : lats) {
      Lattice &lat = kv.second;
      ScaleAcoustic(&lat, acoustic_scale);
      CompactLattice clat;
      if (!DeterminizeLatticePruned(lat, beam, &clat, opts)) {
        K3H_WARN << "For key " << kv.first << ", determinization did not succeed(partial output will be pruned tighter than the specified beam.)";
        n_warn++;
      }
      if (clat.NumStates() == 0) { K3H_WARN << "For key " << kv.first << ", determinized and trimmed lattice was empty."; n_warn++; }
      if (minimize) { PushCompactLatticeStrings(&clat); PushCompactLatticeWeights(&clat); MinimizeCompactLattice(&clat); }
      if (!TopSortIfNeeded(&clat)) K3H_WARN << "Topological sorting of the determinized lattice failed for key " << kv.first;
      states_in += lat.NumStates(); arcs_out += (double)clat.arc_src.size();
      ScaleAcoustic(&clat, 1.0 / acoustic_scale);
      writer.WriteCompactLattice(kv.first, clat);
      n_done++;
    }
    writer.Flush();
    K3H_LOG << "Done " << n_done << " lattices, determinization finished earlier than specified by the beam (or output was empty) on " << n_warn << " of these.";
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
