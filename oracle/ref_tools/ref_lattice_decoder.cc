// oracle/ref_tools/ref_lattice_decoder.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's LatticeFasterDecoder -- decoder/
// lattice-faster-decoder.cc compiled unmodified from /root/reference against the OpenFst stand-in in third_party/minifst --
// on a graph, a log-likelihood matrix and a transition-id -> pdf map read from one binary file, and writes the raw lattice
// (GetRawLattice, before fst::Connect) to another.  oracle/lattice_faster_oracle.cc (the restatement every GPU test is checked
// against) is pinned to this program's output in tests/test_oracle_decoder.py.
//   ref-lattice-decoder <in.bin> <out.bin>
// in.bin : int32 {magic 0x4b33, num_states, start, num_arcs, T, num_pdfs, num_tids_plus_1, max_active, min_active, prune_interval}
//          float {beam, lattice_beam, beam_delta, hash_ratio, prune_scale}
//          int32 arc_offsets[num_states+1], ilabel[A], olabel[A], nextstate[A]; float weight[A], final[num_states];
//          int32 tid2pdf[num_tids_plus_1]; float loglikes[T*num_pdfs]
// out.bin: int64 {num_states, num_arcs, start, reached_final, num_frames_decoded}; int32 frame[S]; float final_graph[S], final_ac[S];
//          int32 src[A], dst[A], ilabel[A], olabel[A]; float graph[A], acoustic[A]; double seconds spent in Decode()
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>
#include "decoder/lattice-faster-decoder.h"

namespace {
struct Reader {
  FILE *f;
  template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "ref-lattice-decoder: short read\n"; exit(2); } }
};
class MatrixDecodable : public kaldi::DecodableInterface {          // like DecodableMatrixScaledMapped (decoder/decodable-matrix.h), scale 1
 public:
  MatrixDecodable(const std::vector<float> &ll, int32_t T, int32_t P, const std::vector<int32_t> &t2p) : ll_(ll), T_(T), P_(P), t2p_(t2p) {}
  kaldi::BaseFloat LogLikelihood(kaldi::int32 frame, kaldi::int32 tid) override { return ll_[(size_t)frame * P_ + t2p_[tid]]; }
  kaldi::int32 NumFramesReady() const override { return T_; }
  bool IsLastFrame(kaldi::int32 frame) const override { return frame == T_ - 1; }
  kaldi::int32 NumIndices() const override { return (kaldi::int32)t2p_.size() - 1; }
 private:
  const std::vector<float> &ll_; int32_t T_, P_; const std::vector<int32_t> &t2p_;
};
// the decoder keeps its per-frame token lists protected; a subclass may count them (that gives each lattice state its frame:
// GetRawLattice numbers the states frame by frame in list order, lattice-faster-decoder.cc:148-157)
class Decoder : public kaldi::LatticeFasterDecoderTpl<fst::VectorFst<fst::StdArc>, kaldi::decoder::StdToken> {
 public:
  using Base = kaldi::LatticeFasterDecoderTpl<fst::VectorFst<fst::StdArc>, kaldi::decoder::StdToken>;
  Decoder(const fst::VectorFst<fst::StdArc> &f, const kaldi::LatticeFasterDecoderConfig &c) : Base(f, c) {}
  std::vector<int32_t> TokensPerFrame() const {
    std::vector<int32_t> n;
    for (const auto &tl : active_toks_) { int32_t k = 0; for (auto *t = tl.toks; t != NULL; t = t->next) k++; n.push_back(k); }
    return n;
  }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc != 3) { std::cerr << "usage: ref-lattice-decoder <in.bin> <out.bin>\n"; return 1; }
  try {
    Reader in{fopen(argv[1], "rb")};
    if (!in.f) { std::cerr << "cannot open " << argv[1] << "\n"; return 2; }
    int32_t h[10]; float c[5]; in.get(h, 10); in.get(c, 5);
    if (h[0] != 0x4b33) { std::cerr << "bad magic\n"; return 2; }
    const int32_t S = h[1], start = h[2], A = h[3], T = h[4], P = h[5], NT = h[6];
    std::vector<int32_t> off(S + 1), il(A), ol(A), nx(A), t2p(NT); std::vector<float> w(A), fin(S), ll((size_t)T * P);
    in.get(off.data(), S + 1); in.get(il.data(), A); in.get(ol.data(), A); in.get(nx.data(), A); in.get(w.data(), A); in.get(fin.data(), S); in.get(t2p.data(), NT); in.get(ll.data(), ll.size());
    fclose(in.f);
    fst::VectorFst<fst::StdArc> graph;
    for (int32_t s = 0; s < S; s++) graph.AddState();
    graph.SetStart(start);
    for (int32_t s = 0; s < S; s++) {
      graph.SetFinal(s, fst::TropicalWeight(fin[s]));
      for (int32_t a = off[s]; a < off[s + 1]; a++) graph.AddArc(s, fst::StdArc(il[a], ol[a], fst::TropicalWeight(w[a]), nx[a]));
    }
    kaldi::LatticeFasterDecoderConfig cfg;
    cfg.beam = c[0]; cfg.lattice_beam = c[1]; cfg.beam_delta = c[2]; cfg.hash_ratio = c[3]; cfg.prune_scale = c[4];
    cfg.max_active = h[7]; cfg.min_active = h[8]; cfg.prune_interval = h[9];
    Decoder dec(graph, cfg);
    MatrixDecodable decodable(ll, T, P, t2p);
    const auto t0 = std::chrono::steady_clock::now();
    dec.Decode(&decodable);
    const double decode_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    kaldi::Lattice lat;
    dec.GetRawLattice(&lat, true);
    const int64_t ns = lat.NumStates(); int64_t na = 0; for (int64_t s = 0; s < ns; s++) na += (int64_t)lat.NumArcs((int)s);
    std::vector<int32_t> frame; { const auto per = dec.TokensPerFrame(); for (size_t f = 0; f < per.size(); f++) frame.insert(frame.end(), per[f], (int32_t)f); }
    if ((int64_t)frame.size() != ns) { std::cerr << "ref-lattice-decoder: " << frame.size() << " tokens but " << ns << " lattice states\n"; return 3; }
    std::vector<float> fg(ns), fa(ns), g, ac; std::vector<int32_t> src, dst, oi, oo;
    for (int64_t s = 0; s < ns; s++) {
      const kaldi::LatticeWeight f = lat.Final((int)s); fg[s] = f.Value1(); fa[s] = f.Value2();
      for (fst::ArcIterator<kaldi::Lattice> it(lat, (int)s); !it.Done(); it.Next()) {
        const kaldi::LatticeArc &arc = it.Value();
        src.push_back((int32_t)s); dst.push_back(arc.nextstate); oi.push_back(arc.ilabel); oo.push_back(arc.olabel); g.push_back(arc.weight.Value1()); ac.push_back(arc.weight.Value2());
      }
    }
    FILE *o = fopen(argv[2], "wb");
    if (!o) { std::cerr << "cannot open " << argv[2] << "\n"; return 2; }
    const int64_t hdr[5] = {ns, na, lat.Start(), dec.ReachedFinal() ? 1 : 0, dec.NumFramesDecoded()};
    fwrite(hdr, 8, 5, o); fwrite(frame.data(), 4, ns, o); fwrite(fg.data(), 4, ns, o); fwrite(fa.data(), 4, ns, o);
    fwrite(src.data(), 4, na, o); fwrite(dst.data(), 4, na, o); fwrite(oi.data(), 4, na, o); fwrite(oo.data(), 4, na, o); fwrite(g.data(), 4, na, o); fwrite(ac.data(), 4, na, o);
    fwrite(&decode_seconds, 8, 1, o);
    fclose(o);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
