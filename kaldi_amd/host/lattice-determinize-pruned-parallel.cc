// lattice-determinize-pruned-parallel -- same command line as the reference's latbin/lattice-determinize-pruned-parallel.cc:104-190:
// lattice-determinize-pruned with --num-threads; lattices are determinized on worker threads and written in input order
// (DeterminizeSequencer = the reference's TaskSequencer).  Host-only.
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "lattice-determinize-pruned-parallel";
    const char *usage =
        "Determinize lattices, keeping only the best path (sequence of acoustic states)\n"
        "for each input-symbol sequence.  This is a version of lattice-determnize-pruned\n"
        "that accepts the --num-threads option.  These programs do pruning as part of the\n"
        "determinization algorithm, which is more efficient and prevents blowup.\n"
        "\n"
        "Usage: lattice-determinize-pruned-parallel [options] lattice-rspecifier lattice-wspecifier\n"
        " e.g.: lattice-determinize-pruned-parallel --acoustic-scale=0.1 --beam=6.0 ark:in.lats ark:det.lats\n";
    ParseOptions po(usage);
    bool minimize = false; float acoustic_scale = 1.0f, beam = 10.0f; int32_t num_threads = 1, num_threads_total = 0;
    DeterminizeLatticePrunedOptions opts; opts.max_mem = 50000000; opts.max_loop = 0;
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
    po.Register("num-threads", &num_threads, "Number of actively processing threads to run in parallel");
    po.Register("num-threads-total", &num_threads_total, "(accepted; the number of lattices in flight is num-threads + 20)");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    if (acoustic_scale == 0.0f) K3H_ERR << "Do not use a zero acoustic scale (cannot be inverted)";
    if (num_threads < 1) K3H_ERR << "--num-threads must be at least 1";
    auto lats = ReadLatticeTable(po.GetArg(1));
    TableWriter writer(po.GetArg(2));
    int32_t n_done = 0, n_warn = 0;
    {
      DeterminizeSequencer::Config cfg;
      cfg.num_threads = num_threads;
      cfg.beam = beam;
      cfg.pre_scale = acoustic_scale;
      cfg.post_scale = 1.0 / acoustic_scale;
      cfg.det = opts;
      cfg.minimize = minimize;
      DeterminizeSequencer seq(cfg, &writer);
      for (auto &kv : lats) seq.Run(kv.first, std::move(kv.second));
      seq.Wait(); n_done = seq.NumDone(); n_warn = seq.NumWarn();
    }
    writer.Flush();
    K3H_LOG << "Done " << n_done << " lattices, had warnings on " << n_warn << " of these.";
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
