// lattice-scale / lattice-add-penalty / lattice-prune -- one source, three programs (the name it is invoked under selects the mode), with
// the command lines of the reference's latbin/lattice-scale.cc, lattice-add-penalty.cc, lattice-prune.cc: the three filters
// local/score.sh puts between the decoder's lattices and lattice-best-path.  Host-only.  Input: Lattice or CompactLattice tables;
// output: CompactLattice (ConvertLattice) like the reference, or the state-level Lattice with --write-compact=false (lattice-scale only
// in the reference; accepted by all three here).
#include <cmath>
#include <cstring>
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    const char *base = strrchr(argv[0], '/'); const std::string prog = base ? base + 1 : argv[0]; g_program = prog;
    const bool scale_mode = prog == "lattice-scale", penalty_mode = prog == "lattice-add-penalty", prune_mode = prog == "lattice-prune";
    if (!scale_mode && !penalty_mode && !prune_mode) { std::cerr << "k3-lattice-tool: invoke it as lattice-scale, lattice-add-penalty or lattice-prune\n"; return 1; }
    const std::string usage = scale_mode ?
        "Apply scaling to lattice weights\nUsage: lattice-scale [options] lattice-rspecifier lattice-wspecifier\n e.g.: lattice-scale --lm-scale=0.0 ark:1.lats ark:scaled.lats\n"
                            : penalty_mode ?
                                "Add word insertion penalty to the lattice.\nUsage: lattice-add-penalty [options] <lattice-rspecifier> <lattice-wspecifier>\n e.g.: lattice-add-penalty --word-ins-penalty=1.0 ark:- ark:-\n"
                                           : "Apply beam pruning to lattices\nUsage: lattice-prune [options] lattice-rspecifier lattice-wspecifier\n e.g.: lattice-prune --acoustic-scale=0.1 --beam=4.0 ark:1.lats ark:pruned.lats\n";
    ParseOptions po(usage.c_str());
    bool write_compact = true; float acoustic_scale = 1.0f, inv_acoustic_scale = 1.0f, lm_scale = 1.0f, a2l = 0.0f, l2a = 0.0f, word_ins_penalty = 0.0f, beam = 10.0f;
    po.Register("write-compact", &write_compact, "If true, write in normal (compact) form.");
    if (scale_mode || prune_mode) {
      po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
      po.Register("inv-acoustic-scale", &inv_acoustic_scale, "An alternative way of setting the acoustic scale: you can set its inverse.");
    }
    if (scale_mode) {
      po.Register("lm-scale", &lm_scale, "Scaling factor for graph/lm costs");
      po.Register("acoustic2lm-scale", &a2l, "Add this times original acoustic costs to LM costs");
      po.Register("lm2acoustic-scale", &l2a, "Add this times original LM costs to acoustic costs");
    }
    if (penalty_mode) po.Register("word-ins-penalty", &word_ins_penalty, "Word insertion penalty");
    if (prune_mode) po.Register("beam", &beam, "Pruning beam [applied after acoustic scaling]");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    if (!(acoustic_scale == 1.0f || inv_acoustic_scale == 1.0f)) K3H_ERR << "You cannot specify both --acoustic-scale and --inv-acoustic-scale";
    if (inv_acoustic_scale != 1.0f) acoustic_scale = 1.0f / inv_acoustic_scale;
    if (prune_mode && acoustic_scale == 0.0f) K3H_ERR << "Do not use a zero acoustic scale (cannot be inverted)";
    TableWriter writer(po.GetArg(2)); int32_t n_done = 0, n_err = 0; int64_t arcs_in = 0, arcs_out = 0;
    // (graph, acoustic) -> (s00 g + s01 a, s10 g + s11 a), in double like fst::ScaleLattice
    auto scale = [](Lattice *l, double s00, double s01, double s10, double s11) {
      for (size_t k = 0; k < l->arc_graph.size(); k++) {
        const double g = l->arc_graph[k], a = l->arc_ac[k];
        l->arc_graph[k] = (float)(s00 * g + s01 * a);
        l->arc_ac[k] = (float)(s10 * g + s11 * a);
      }
      if (l->st_final_ac.empty()) l->st_final_ac.assign(l->st_final.size(), 0.0f);
      for (size_t s = 0; s < l->st_final.size(); s++) if (std::isfinite(l->st_final[s])) {
        const double g = l->st_final[s], a = l->st_final_ac[s];
        l->st_final[s] = (float)(s00 * g + s01 * a);
        l->st_final_ac[s] = (float)(s10 * g + s11 * a);
      }
    };
    for (auto &kv : ReadLatticeTable(po.GetArg(1))) {
      Lattice &lat = kv.second; arcs_in += (int64_t)lat.arc_src.size();
      if (scale_mode) scale(&lat, lm_scale, a2l, l2a, acoustic_scale);
      // AddWordInsPenToCompactLattice: on every arc that carries a word
      else if (penalty_mode) {
        for (size_t k = 0; k < lat.arc_olabel.size(); k++) if (lat.arc_olabel[k] != 0) lat.arc_graph[k] += word_ins_penalty;
      }
      else {
        scale(&lat, 1.0, 0.0, 0.0, acoustic_scale);
        if (!PruneLattice(beam, &lat)) { K3H_WARN << "Error pruning lattice for utterance " << kv.first; n_err++; }
        scale(&lat, 1.0, 0.0, 0.0, 1.0 / acoustic_scale);
      }
      arcs_out += (int64_t)lat.arc_src.size();
      if (write_compact) { Connect(&lat); CompactLattice c; ConvertLattice(lat, &c); writer.WriteCompactLattice(kv.first, c); } else writer.WriteLattice(kv.first, lat);
      n_done++;
    }
    writer.Flush();
    if (prune_mode) K3H_LOG << "Overall, pruned from on average " << (arcs_in / std::max(1, n_done)) << " to " << (arcs_out / std::max(1, n_done)) <<
        " arcs, over " << n_done << " utterances.";
    K3H_LOG << "Done " << n_done << " lattices" << (prune_mode ? ", errors on " + std::to_string(n_err) : std::string());
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
