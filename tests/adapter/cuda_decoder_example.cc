// tests/adapter/cuda_decoder_example.cc -- a caller of kaldi::cuda_decoder::CudaFst / CudaDecoder written against the REFERENCE's signatures
// (cudadecoder/cuda-fst.h:62-149, cuda-decoder.h:224-345) and compiled against include/k3_cuda_decoder.h + the reference's lattice types.
// Same file protocol as oracle/ref_tools/ref_lattice_decoder.cc (the reference's CPU LatticeFasterDecoder), so that the test can compare the
// two outputs directly:   cuda-decoder-example <in.bin> <out.bin> [frames-per-call]
#include <tuple>
#include <algorithm>
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>
#include "k3_cuda_decoder.h"
using namespace kaldi; using namespace kaldi::cuda_decoder;
namespace {
struct Reader { FILE *f; template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "short read\n"; exit(2); } } };
struct IdentityTransitions : public TransitionInformation {      // the test graphs carry their own transition-id -> pdf-id table
  explicit IdentityTransitions(const std::vector<int32> &t2p) : t2p_(t2p) {}
  bool TransitionIdsEquivalent(int32 a, int32 b) const override { return a == b; }
  bool TransitionIdIsStartOfPhone(int32) const override { return true; }
  int32 TransitionIdToPhone(int32) const override { return 1; }
  bool IsFinal(int32) const override { return true; }
  bool IsSelfLoop(int32) const override { return false; }
  const std::vector<int32> &TransitionIdToPdfArray() const override { return t2p_; }
  int32 NumPdfs() const override { int32 m = 0; for (int32 p : t2p_) m = std::max(m, p + 1); return m; }
  std::vector<int32> t2p_;
};
}
int main(int argc, char **argv) {
  if (argc < 3) { std::cerr << "usage: cuda-decoder-example <in.bin> <out.bin> [frames-per-call]\n"; return 1; }
  try {
    Reader in{fopen(argv[1], "rb")}; if (!in.f) return 2;
    int32_t h[10]; float c[5]; in.get(h, 10); in.get(c, 5);
    const int32_t S = h[1], start = h[2], A = h[3], T = h[4], P = h[5], NT = h[6];
    std::vector<int32_t> off(S + 1), il(A), ol(A), nx(A), t2p(NT); std::vector<float> w(A), fin(S), ll((size_t)T * P);
    in.get(off.data(), S + 1); in.get(il.data(), A); in.get(ol.data(), A); in.get(nx.data(), A); in.get(w.data(), A); in.get(fin.data(), S); in.get(t2p.data(), NT); in.get(ll.data(), ll.size());
    fclose(in.f);
    fst::VectorFst<fst::StdArc> graph;
    for (int32_t s = 0; s < S; s++) graph.AddState();
    graph.SetStart(start);
    for (int32_t s = 0; s < S; s++) { graph.SetFinal(s, fst::TropicalWeight(fin[s])); for (int32_t a = off[s]; a < off[s + 1]; a++) graph.AddArc(s, fst::StdArc(il[a], ol[a], fst::TropicalWeight(w[a]), nx[a])); }
    IdentityTransitions trans(t2p);
    CudaFst cuda_fst(graph, &trans);
    CudaDecoderConfig cfg; cfg.default_beam = c[0]; cfg.lattice_beam = c[1]; cfg.beam_delta = c[2]; cfg.hash_ratio = c[3]; cfg.max_active = h[7]; cfg.min_active = h[8];
    cfg.main_q_capacity = 65536; cfg.aux_q_capacity = 262144; cfg.ntokens_pre_allocated = 2500000;
    const int32 max_batch_size = 1, num_channels = 2;
    cfg.max_frames_per_channel = T + 8;
    // constructed the way the reference's online pipeline does (cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.cc:125-136): (fst, config, nlanes, nchannels)
    std::unique_ptr<CudaDecoder> cuda_decoder;
    cuda_decoder.reset(new CudaDecoder(cuda_fst, cfg, max_batch_size, num_channels));
    cuda_decoder->SetThreadPoolAndStartCPUWorkers((void *)NULL, 2);
    cuda_decoder->SetOutputFrameShiftInSeconds(0.03f);
    CudaDecoder &decoder = *cuda_decoder;
    if (cuda_fst.NumPdfs() > P) { std::cerr << "graph needs " << cuda_fst.NumPdfs() << " pdfs, rows have " << P << "\n"; return 4; }
    { CudaDecoder three_args(cuda_fst, cfg, num_channels); (void)three_args; }      // cuda-decoder.h:230-232
    decoder.AllowPartialHypotheses();
    float *d_ll = NULL;
    if (hipMalloc((void **)&d_ll, sizeof(float) * ll.size()) != hipSuccess || hipMemcpy(d_ll, ll.data(), sizeof(float) * ll.size(), hipMemcpyHostToDevice) != hipSuccess) { std::cerr << "hip alloc/copy failed\n"; return 3; }
    const int32 step = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<ChannelId> channels = {1};
    // two channels share the ONE lane (nlanes = 1 < nchannels = 2, cuda-decoder.h:224-229): the same utterance is decoded on both, their AdvanceDecoding calls alternating --
    // every call carries a single (channel, frame pointer) pair; a channel's state stays resident between its calls
    decoder.InitDecoding(std::vector<ChannelId>{0, 1});
    // channel 0 goes through the DEPRECATED overload (cuda-decoder.h:267-270): a CudaDecodableInterface that hands out device pointers frame by frame, at most n frames a call
    struct RowsDecodable : public CudaDecodableInterface {
      float *rows; int32 T, P;
      RowsDecodable(float *rows, int32 T, int32 P) : rows(rows), T(T), P(P) {}
      BaseFloat LogLikelihood(int32, int32) override { return 0.0f; }
      bool IsLastFrame(int32 frame) const override { return frame == T - 1; }
      int32 NumFramesReady() const override { return T; }
      int32 NumIndices() const override { return P; }
      BaseFloat *GetLogLikelihoodsCudaPointer(int32 subsampled_frame) override { return rows + (size_t)subsampled_frame * P; }
    } decodable0(d_ll, T, P);
    for (int32 t = 0; t < T; t += step) {
      const int32 n = std::min(step, T - t);
      for (ChannelId ch : {1, 0}) {
        if (ch == 0) { std::vector<ChannelId> c0 = {0}; std::vector<CudaDecodableInterface *> d0 = {&decodable0}; decoder.AdvanceDecoding(c0, d0, n); continue; }
        std::vector<std::pair<ChannelId, const BaseFloat *>> lanes = {{ch, d_ll + (size_t)t * P}};
        if (n == 1) decoder.AdvanceDecoding(lanes); else decoder.AdvanceDecoding(lanes, n, P);
      }
    }
    {
      Lattice l0, l1; std::vector<Lattice *> o0 = {&l0}, o1 = {&l1};
      decoder.GetRawLattice(std::vector<ChannelId>{0}, o0, true); decoder.GetRawLattice(std::vector<ChannelId>{1}, o1, true);
      // (state numbers and arc order follow the order tokens / links were allocated in, which without literal_order depends on timing: compare the arcs as a multiset of
      //  (ilabel, olabel, graph cost, acoustic cost) and the final costs as a multiset; channel 1's lattice is held to the reference decoder's arc by arc by the calling test)
      typedef std::tuple<int32, int32, float, float> ArcKey;
      auto arcs_of = [](const Lattice &l) { std::vector<ArcKey> v; for (int32 st = 0; st < l.NumStates(); st++) for (fst::ArcIterator<Lattice> it(l, st); !it.Done(); it.Next()) { const LatticeArc &x = it.Value(); v.push_back(ArcKey(x.ilabel, x.olabel, x.weight.Value1(), x.weight.Value2())); } std::sort(v.begin(), v.end()); return v; };
      auto finals_of = [](const Lattice &l) { std::vector<float> v; for (int32 st = 0; st < l.NumStates(); st++) v.push_back(l.Final(st).Value1()); std::sort(v.begin(), v.end()); return v; };
      const bool same = l0.NumStates() == l1.NumStates() && decoder.NumFramesDecoded(0) == decoder.NumFramesDecoded(1) && arcs_of(l0) == arcs_of(l1) && finals_of(l0) == finals_of(l1);
      if (!same) { int64_t a0 = 0, a1 = 0; for (int32 st = 0; st < l0.NumStates(); st++) a0 += l0.NumArcs(st); for (int32 st = 0; st < l1.NumStates(); st++) a1 += l1.NumArcs(st);
        std::cerr << "channel 0: " << l0.NumStates() << " states " << a0 << " arcs " << decoder.NumFramesDecoded(0) << " frames; channel 1: " << l1.NumStates() << " states " << a1 << " arcs " << decoder.NumFramesDecoded(1) << " frames\n"; }
      std::cerr << "two channels interleaved on one lane: lattices " << (same ? "identical" : "DIFFER") << "\n";
      if (!same) return 5;
    }
    PartialHypothesis *ph; decoder.GetPartialHypothesis(1, &ph);
    Lattice best, lat; std::vector<Lattice *> outs = {&best};
    decoder.GetBestPath(channels, outs, true);
    outs[0] = &lat; decoder.GetRawLattice(channels, outs, true);
    std::cerr << "frames decoded " << decoder.NumFramesDecoded(1) << ", partial hypothesis: " << ph->out_str << ", best path arcs " << best.NumStates() - 1 << "\n";
    const int64_t ns = lat.NumStates(); int64_t na = 0; for (int64_t s = 0; s < ns; s++) na += (int64_t)lat.NumArcs((int)s);
    // (states of the raw lattice are numbered frame by frame by the C ABI: recover the frames from the arcs)
    std::vector<int32_t> frame(ns, 0); std::vector<float> fg(ns), fa(ns), g, ac; std::vector<int32_t> src, dst, oi, oo;
    for (int64_t s = 0; s < ns; s++) {
      const LatticeWeight f = lat.Final((int)s); fg[s] = f.Value1(); fa[s] = f.Value2();
      for (fst::ArcIterator<Lattice> it(lat, (int)s); !it.Done(); it.Next()) { const LatticeArc &arc = it.Value(); src.push_back((int32_t)s); dst.push_back(arc.nextstate); oi.push_back(arc.ilabel); oo.push_back(arc.olabel); g.push_back(arc.weight.Value1()); ac.push_back(arc.weight.Value2()); }
    }
    for (bool changed = true; changed;) { changed = false; for (size_t a = 0; a < src.size(); a++) { const int32_t fr = frame[src[a]] + (oi[a] != 0 ? 1 : 0); if (fr > frame[dst[a]]) { frame[dst[a]] = fr; changed = true; } } }
    FILE *o = fopen(argv[2], "wb"); if (!o) return 2;
    const int64_t hdr[5] = {ns, na, lat.Start(), 1, decoder.NumFramesDecoded(1)}; const double secs = 0.0;
    fwrite(hdr, 8, 5, o); fwrite(frame.data(), 4, ns, o); fwrite(fg.data(), 4, ns, o); fwrite(fa.data(), 4, ns, o);
    fwrite(src.data(), 4, na, o); fwrite(dst.data(), 4, na, o); fwrite(oi.data(), 4, na, o); fwrite(oo.data(), 4, na, o); fwrite(g.data(), 4, na, o); fwrite(ac.data(), 4, na, o); fwrite(&secs, 8, 1, o);
    fclose(o); (void)hipFree(d_ll);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
