// k3-host-tool -- developer/test aid for the host-side format readers (no GPU needed):
//   k3-host-tool tid2pdf <model.mdl>          NumPdfs, NumTransitionIds, then TransitionIdToPdf(1..N)  (same output as oracle's dump-tid2pdf)
//   k3-host-tool tidinfo <model.mdl>          per transition-id: phone, self-loop flag, start-of-phone flag
//   k3-host-tool gmm-dump <diag-gmm> | ie-dump <ivector-extractor>      the numbers of an i-vector extractor's model files
//   k3-host-tool fstinfo <fst>                states arcs start, FNV-1a checksum of the CSR
//   k3-host-tool copy-fst <fst-in> <fst-out>  read (vector|const) and write as vector
//   k3-host-tool parse-options [options] [args]   the option values ParseOptions::Read ends up with (config files vs command line)
//   k3-host-tool convert-lattice <lattice-rspecifier> <lattice-wspecifier>   state-level lattices re-packed as CompactLattices
#include <iostream>
#include <fstream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "k3-host-tool";
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "tid2pdf" && argc == 3) {
      TransitionInfo ti = ReadTransitionModel(argv[2]);
      std::cout << ti.num_pdfs << " " << ti.id2pdf.size() - 1 << "\n";
      for (size_t t = 1; t < ti.id2pdf.size(); t++) std::cout << ti.id2pdf[t] << (t % 32 ? " " : "\n");
      std::cout << "\n"; return 0;
    }
    if (cmd == "tidinfo" && argc == 3) {       // per transition-id: phone, IsSelfLoop, TransitionIdIsStartOfPhone (same output as oracle's dump-tidinfo)
      TransitionInfo ti = ReadTransitionModel(argv[2]);
      for (size_t t = 1; t < ti.id2pdf.size(); t++) std::cout << t << " " << ti.id2phone[t] << " " << (int)ti.self_loop[t] << " " << (int)ti.phone_start[t] << "\n";
      return 0;
    }
    if (cmd == "gmm-dump" && argc == 3) {               // every number of a DiagGmm file, one block per member (compared with the reference's text dump)
      const DiagGmmModel g = ReadDiagGmm(argv[2]); std::cout.precision(9);
      std::cout << "num_gauss " << g.num_gauss << " dim " << g.dim << "\n";
      auto dump = [&](const char *name, const std::vector<double> &v) { std::cout << name; for (double x : v) std::cout << " " << x; std::cout << "\n"; };
      dump("gconsts", g.gconsts); dump("weights", g.weights); dump("means_invvars", g.means_invvars); dump("inv_vars", g.inv_vars); return 0;
    }
    if (cmd == "ie-dump" && argc == 3) {
      const IvectorExtractorModel m = ReadIvectorExtractor(argv[2]); std::cout.precision(12);
      std::cout << "num_gauss " << m.num_gauss << " feat_dim " << m.feat_dim << " ivector_dim " << m.ivector_dim << " w " << m.w_rows << "x" << m.w_cols <<
          " prior_offset " << m.prior_offset << "\n";
      auto dump = [&](const char *name, const std::vector<double> &v) { std::cout << name; for (double x : v) std::cout << " " << x; std::cout << "\n"; };
      dump("w", m.w); dump("w_vec", m.w_vec); dump("M", m.M); dump("sigma_inv", m.sigma_inv); return 0;
    }
    if (cmd == "fstinfo" && argc == 3) {
      HostFst f = ReadFstKaldiGeneric(argv[2]);
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) {
          h ^= b[i];
          h *= 1099511628211ull;
        }
      };
      mix(f.arc_offsets.data(), 4 * f.arc_offsets.size()); mix(f.ilabel.data(), 4 * f.ilabel.size()); mix(f.olabel.data(), 4 * f.olabel.size());
      mix(f.weight.data(), 4 * f.weight.size()); mix(f.nextstate.data(), 4 * f.nextstate.size()); mix(f.final_cost.data(), 4 * f.final_cost.size());
      std::cout << f.NumStates() << " " << f.ilabel.size() << " " << f.start << " " << h << "\n"; return 0;
    }
    if (cmd == "convert-lattice" && argc == 4) {       // Lattice table -> CompactLattice table without determinization (ConvertLattice)
      TableWriter w(argv[3]);
      for (auto &kv : ReadLatticeTable(argv[2])) { Connect(&kv.second); CompactLattice c; ConvertLattice(kv.second, &c); w.WriteCompactLattice(kv.first, c); }
      w.Flush(); return 0;
    }
    if (cmd == "parse-options") {                      // what an option ends up as after ParseOptions::Read (util/parse-options.cc:329-371: config files first, command line wins)
      ParseOptions po("test"); float beam = 16.0f; int32_t max_active = 7; bool flag = false; std::string name = "dflt";
      po.Register("beam", &beam, "beam"); po.Register("max-active", &max_active, "max active"); po.Register("flag", &flag, "flag"); po.Register("name", &name, "name");
      po.Read(argc - 1, argv + 1);
      std::cout << "beam=" << beam << " max-active=" << max_active << " flag=" << (flag ? "true" : "false") << " name=" << name << " nargs=" << po.NumArgs() << "\n"; return 0;
    }
    if (cmd == "word-align" && argc >= 6) {      // WordAlignLattice + MBR on every lattice of a table, in the format of oracle/ref_tools/ref_word_align.cc:
      // k3-host-tool word-align <mdl> <word_boundary.int> <lattice-rspecifier> <out.txt> [reorder [silence-label [partial-word-label [max-expand]]]]
      const TransitionInfo ti = ReadTransitionModel(argv[2]);
      const WordBoundaryInfo info = ReadWordBoundaryInfo(argv[3], argc > 6 ? atoi(argv[6]) != 0 : true, argc > 7 ? atoi(argv[7]) : 0, argc > 8 ? atoi(argv[8]) : 0);
      const float max_expand = argc > 9 ? (float)atof(argv[9]) : 0.0f;
      std::ofstream out(argv[5]); out.precision(9);
      for (auto &kv : ReadLatticeTable(argv[4])) {
        Connect(&kv.second); CompactLattice c; if (kv.second.NumStates() > 0) ConvertLattice(kv.second, &c);
        const int32_t max_states = max_expand > 0 ? (int32_t)(1000 + max_expand * c.NumStates()) : 0;
        CompactLattice al; const bool ok = c.NumStates() == 0 ? true : WordAlignLattice(c, ti, info, max_states, &al);
        out << kv.first << "\nok " << (ok ? 1 : 0) << "\nstates " << al.NumStates() << " start " << (al.NumStates() ? al.start : -1) << "\n";
        size_t k = 0;
        for (int32_t s = 0; s < al.NumStates(); s++) {
          for (; k < al.arc_src.size() && al.arc_src[k] == s; k++) {
            out << "a " << s << " " << al.arc_dst[k] << " " << al.arc_label[k] << " " << al.arc_graph[k] << " " << al.arc_ac[k] << " ";
            for (size_t j = 0; j < al.arc_str[k].size(); j++) out << (j ? "_" : "") << al.arc_str[k][j];
            out << "\n";
          }
          if (al.is_final[s]) out << "f " << s << " " << al.fin_graph[s] << " " << al.fin_ac[s] << "\n";
        }
        if (al.NumStates() > 0) {
          MinimumBayesRisk mbr(al, MinimumBayesRiskOptions());
          out << "words"; for (int32_t w : mbr.GetOneBest()) out << " " << w;
          out << "\ntimes"; for (const auto &t : mbr.GetOneBestTimes()) out << " " << t.first << " " << t.second;
          out << "\nconf"; for (float x : mbr.GetOneBestConfidences()) out << " " << x;
          out << "\n";
        }
        out << "end\n";
      }
      return 0;
    }
    if (cmd == "copy-fst" && argc == 4) { WriteFstVector(ReadFstKaldiGeneric(argv[2]), argv[3]); return 0; }
    std::cerr << "usage: k3-host-tool tid2pdf <mdl> | tidinfo <mdl> | fstinfo <fst> | copy-fst <in> <out> | convert-lattice <rspecifier> <wspecifier>\n"; return 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
