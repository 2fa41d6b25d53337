// kaldi_amd/adapter/chain-k3.cc -- the chain (LF-MMI) library surface that the reference's TRAINER calls, on the MI355X: with it the reference's own, unmodified
//   chainbin/nnet3-chain-train.cc  (main: <raw-nnet-in> <denominator-fst-in> <chain-training-examples-in> <raw-nnet-out>, NnetChainTrainingOptions)
//   nnet3/nnet-chain-training.cc   (NnetChainTrainer::Train / TrainInternal / ProcessOutputs: xent regularisation, deriv weights, backstitch, max-change, ...)
//   nnet3/nnet-chain-example.cc, nnet-example.cc, nnet-example-utils.cc  (NnetChainExample archives, GetChainComputationRequest)
// link over the CuMatrix adapter (cu-k3.cc) and train on the GPU -- kaldi_amd/adapter/_build/nnet3-chain-train-egs (kaldi_amd/adapter/build.sh).
// What is defined here instead of compiling the reference's chain/*.cc:
//   chain::ComputeChainObjfAndDeriv   chain/chain-training.cc:242-337 (+ the end-to-end branch :86-215)  -> k3_chain_objf_and_deriv on the adapter's device pointers
//   chain::DenominatorGraph           chain/chain-den-graph.cc:29-143: constructor -> k3_chain_den_create (transitions by source / destination and the initial probabilities are
//                                     built on the GPU side as that constructor builds them); NumStates / InitialProbs
//   chain::Supervision                chain/chain-supervision.cc:611-661,708-713: Read (text and binary), copy, Swap -- that file as a whole needs real OpenFst (graph
//                                     compilation), which /root/reference does not vendor
//   fst::ReadFstKaldi(name, VectorFst*)  fstext/kaldi-fst-io.cc:109-115, through the host layer's OpenFst-binary reader (k3host::ReadFstKaldiGeneric)
// The merged supervision FST of a minibatch (nnet3-chain-merge-egs: MergeSupervision = fst::Concat + RmEpsilon + SortBreadthFirstSearch, chain-supervision.cc:738-777) is cut
// back into its sequences here, because the numerator kernel walks every sequence with its own wavefront (k3_chain_supervision_create takes UNMERGED FSTs): SplitMergedSupervision.
//
// PARITY NOTE (binary egs): Supervision::Read's binary branch reads an OpenFst StdCompactAcceptorFst (fst::CompactFst with the acceptor compactor).  OpenFst 1.8.4 is not in
// /root/reference (tools/Makefile downloads it), so that file layout is restated from OpenFst's published format (FstHeader; Unsigned states[nstates + 1];
// {label, weight, nextstate}
// compacts[], a final weight being an element with label kNoLabel) and checked only against files written by tests/adapter/write_chain_egs.py from the same description: UNPINNED.
// The text form (nnet3-chain-copy-egs ark,t:) uses the OpenFst text format Kaldi's own fstext/kaldi-fst-io-inl.h:76-166 parses; that parser is restated from the reference.
#include <map>
#include <mutex>
#include "chain/chain-training.h"
#include "chain/chain-den-graph.h"
#include "chain/chain-supervision.h"
#include "cudamatrix/cu-allocator.h"
#include "k3hip.h"
extern "C" void *k3_adapter_stream();      // kaldi_amd/adapter/cu-k3.cc: the calling thread's stream (every CuMatrix operation of this thread is queued on it)
#include "k3_host.h"

// cudamatrix/cu-allocator.cc:49 (RegisterCuAllocatorOptions registers its fields; the allocator itself is the adapter's)
namespace kaldi {
  CuAllocatorOptions g_allocator_options;
}

namespace {
struct Csr { std::vector<int64_t> off; std::vector<int32_t> il, nx; std::vector<float> w, fin; };
void ToCsr(const fst::StdVectorFst &f, Csr *c) {
  const int32_t S = f.NumStates(); c->off.assign(1, 0); c->il.clear(); c->nx.clear(); c->w.clear(); c->fin.resize(S);
  for (int32_t s = 0; s < S; s++) {
    for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) {
      const fst::StdArc &a = it.Value();
      c->il.push_back(a.ilabel);
      c->nx.push_back(a.nextstate);
      c->w.push_back(a.weight.Value());
    }
    c->off.push_back((int64_t)c->il.size()); c->fin[s] = f.Final(s).Value();
  }
}
std::mutex g_mu; std::map<const kaldi::chain::DenominatorGraph *, k3_chain_den *> g_den;
}  // namespace

namespace fst {
// fstext/kaldi-fst-io.cc:109-115
void ReadFstKaldi(std::string rxfilename, VectorFst<StdArc> *ofst) {
  const k3host::HostFst h = k3host::ReadFstKaldiGeneric(rxfilename);
  *ofst = VectorFst<StdArc>();
  for (int32_t s = 0; s < h.NumStates(); s++) ofst->AddState();
  if (h.NumStates() > 0) ofst->SetStart(h.start);
  for (int32_t s = 0; s < h.NumStates(); s++) {
    if (h.final_cost[s] != std::numeric_limits<float>::infinity()) ofst->SetFinal(s, TropicalWeight(h.final_cost[s]));
    for (int32_t a = h.arc_offsets[s]; a < h.arc_offsets[s + 1]; a++) ofst->AddArc(s, StdArc(h.ilabel[a], h.olabel[a], TropicalWeight(h.weight[a]), h.nextstate[a]));
  }
}
}  // namespace fst

namespace kaldi {
namespace chain {

// ---- DenominatorGraph (chain-den-graph.cc:29-45, :47-60).  The members of the class stay empty but for the initial probabilities: the graph lives behind k3_chain_den.
DenominatorGraph::DenominatorGraph(const fst::StdVectorFst &fst, int32 num_pdfs): num_pdfs_(num_pdfs) {
  if (GetVerboseLevel() > 2) KALDI_LOG << "Before initialization, transition-probs=" << fst.NumStates();
  Csr c; ToCsr(fst, &c);
  for (int32_t l : c.il) if (l < 1 || l > num_pdfs) KALDI_ERR << "Denominator FST has the label " << l << " (labels are pdf-ids + 1 in [1, " << num_pdfs << "])";
  k3_chain_den *den = NULL;
  if (k3_chain_den_create(fst.NumStates(), fst.Start(), num_pdfs, c.off.data(), c.il.data(), c.nx.data(), c.w.data(), c.fin.data(), &den) != K3_OK) KALDI_ERR << k3_last_error();
  Vector<BaseFloat> ip(fst.NumStates()); if (k3_chain_den_initial_probs(den, ip.Data()) != K3_OK) KALDI_ERR << k3_last_error();
  initial_probs_ = ip;
  std::lock_guard<std::mutex> g(g_mu); g_den[this] = den;
}
int32 DenominatorGraph::NumStates() const { return initial_probs_.Dim(); }
const CuVector<BaseFloat> &DenominatorGraph::InitialProbs() const { return initial_probs_; }

// ---- Supervision (chain-supervision.cc:708-713, :596-609 Swap, :611-661 Read)
Supervision::Supervision(const Supervision &other): weight(other.weight), num_sequences(other.num_sequences), frames_per_sequence(other.frames_per_sequence),
    label_dim(other.label_dim), fst(other.fst),
                                                    e2e_fsts(other.e2e_fsts), alignment_pdfs(other.alignment_pdfs) { }
void Supervision::Swap(Supervision *other) {
  std::swap(weight, other->weight);
  std::swap(num_sequences, other->num_sequences);
  std::swap(frames_per_sequence, other->frames_per_sequence);
  std::swap(label_dim, other->label_dim);
  std::swap(fst, other->fst); std::swap(e2e_fsts, other->e2e_fsts); std::swap(alignment_pdfs, other->alignment_pdfs);
}
namespace {
// OpenFst text format as Kaldi reads it inside a table (fstext/kaldi-fst-io-inl.h:76-166, ReadFstKaldi(std::istream&, binary = false, ...)): the newline the form starts with, then
// lines "src dst ilabel olabel [weight]" / "final [weight]"; the first line's source state is the start state; an empty line ends the FST.
void ReadFstText(std::istream &is, fst::StdVectorFst *ofst) {
  while (std::isspace(is.peek()) && is.peek() != '\n') is.get();
  if (is.peek() == '\n') is.get(); else KALDI_ERR << "Reading FST: unexpected sequence of spaces  at file position " << is.tellg();
  *ofst = fst::StdVectorFst(); std::string line; size_t nline = 0;
  auto weight = [&](const std::string &t, float *w) { if (t == "Infinity") { *w = std::numeric_limits<float>::infinity(); return true; } return ConvertStringToReal(t, w); };
  while (std::getline(is, line)) {
    nline++; std::vector<std::string> col; SplitStringToVector(line, " \t\r\n", true, &col);
    if (col.empty()) break;      // the empty line that terminates the FST in the archive format
    int32 s = 0; if (col.size() > 5 || !ConvertStringToInteger(col[0], &s)) KALDI_ERR << "Bad line in FST: " << line;
    while (s >= ofst->NumStates()) ofst->AddState();
    if (nline == 1) ofst->SetStart(s);
    bool ok = true; int32 d = s, il = 0, ol = 0; float w = 0.0f;
    switch (col.size()) {
      case 1: ofst->SetFinal(s, fst::TropicalWeight::One()); break;
      case 2: ok = weight(col[1], &w); if (ok) ofst->SetFinal(s, fst::TropicalWeight(w)); break;
      case 4: ok = ConvertStringToInteger(col[1], &d) && ConvertStringToInteger(col[2], &il) && ConvertStringToInteger(col[3], &ol);
        if (ok) { while (d >= ofst->NumStates()) ofst->AddState(); ofst->AddArc(s, fst::StdArc(il, ol, fst::TropicalWeight::One(), d)); } break;
      case 5: ok = ConvertStringToInteger(col[1], &d) && ConvertStringToInteger(col[2], &il) && ConvertStringToInteger(col[3], &ol) && weight(col[4], &w);
        if (ok) { while (d >= ofst->NumStates()) ofst->AddState(); ofst->AddArc(s, fst::StdArc(il, ol, fst::TropicalWeight(w), d)); } break;
      default: ok = false;      // (3 columns: not an acceptor in this format)
    }
    if (!ok) KALDI_ERR << "Bad line in FST: " << line;
  }
}
// OpenFst binary StdCompactAcceptorFst (see the parity note at the top): FstHeader {int32 magic 2125659606; string fsttype "compact_acceptor"; string arctype
// "standard"; int32 version;
// int32 flags; uint64 properties; int64 start, numstates, numarcs}, [symbol tables when flagged], then the compact store: uint32 states[numstates + 1] (element
// offsets; read when the
// compactor has variable out-degree, as the acceptor compactor has), element {int32 label, float weight, int32 nextstate} x states[numstates]; an element with
// label -1 is the state's
// final weight.  In an aligned file (flag bit 2) both arrays start at a multiple of 16 bytes of the stream.
void ReadCompactAcceptor(std::istream &is, fst::StdVectorFst *ofst) {
  auto rd = [&](void *p, size_t n) { is.read(reinterpret_cast<char *>(p), n); if (!is) KALDI_ERR << "Error reading compact FST from disk"; };
  auto rstr = [&]() {
    int32 n = 0;
    rd(&n, 4);
    if (n < 0 || n > 1024) KALDI_ERR << "Error reading compact FST from disk (header)";
    std::string s(n, '\0');
    if (n) rd(&s[0], n);
    return s;
  };
  const std::streampos p0 = is.tellg();
  int32 magic = 0; rd(&magic, 4); if (magic != 2125659606) KALDI_ERR << "Error reading compact FST from disk (bad magic number " << magic << ")";
  const std::string fsttype = rstr(), arctype = rstr();
  if (fsttype != "compact_acceptor" || arctype != "standard") KALDI_ERR << "Expected a compact_acceptor FST over standard arcs, got " << fsttype << " / " << arctype;
  int32 version = 0, flags = 0; uint64 props = 0; int64 start = 0, nstates = 0, narcs = 0;
  rd(&version, 4); rd(&flags, 4); rd(&props, 8); rd(&start, 8); rd(&nstates, 8); rd(&narcs, 8);
  if ((flags & 3) != 0) KALDI_ERR << "compact FST with symbol tables: not supported (Kaldi writes none)";
  const bool aligned = (flags & 4) != 0;
  auto align = [&]() { if (!aligned) return; const std::streamoff pos = is.tellg() - p0; const int pad = (int)((16 - pos % 16) % 16); char tmp[16]; if (pad) rd(tmp, pad); };
  if (nstates < 0 || nstates > (1ll << 31)) KALDI_ERR << "Error reading compact FST from disk (states)";
  std::vector<uint32> st((size_t)nstates + 1); align(); rd(st.data(), st.size() * 4);
  const uint32 ncompacts = st[nstates];
  struct Elem { int32 label; float weight; int32 nextstate; }; static_assert(sizeof(Elem) == 12, "compact element");
  std::vector<Elem> el(ncompacts); align(); if (ncompacts) rd(el.data(), (size_t)ncompacts * 12);
  *ofst = fst::StdVectorFst();
  for (int64 s = 0; s < nstates; s++) ofst->AddState();
  if (nstates > 0 && start >= 0) ofst->SetStart((int32)start);
  for (int64 s = 0; s < nstates; s++)
    for (uint32 k = st[s]; k < st[s + 1]; k++) {
      if (el[k].label == fst::kNoLabel) ofst->SetFinal((int32)s, fst::TropicalWeight(el[k].weight));
      else ofst->AddArc((int32)s, fst::StdArc(el[k].label, el[k].label, fst::TropicalWeight(el[k].weight), el[k].nextstate));
    }
}
void ReadSupFst(std::istream &is, bool binary, fst::StdVectorFst *f) { if (binary) ReadCompactAcceptor(is, f); else ReadFstText(is, f); }
}  // namespace
void Supervision::Read(std::istream &is, bool binary) {
  ExpectToken(is, binary, "<Supervision>"); ExpectToken(is, binary, "<Weight>"); ReadBasicType(is, binary, &weight);
  ExpectToken(is, binary, "<NumSequences>"); ReadBasicType(is, binary, &num_sequences); ExpectToken(is, binary, "<FramesPerSeq>"); ReadBasicType(is, binary, &frames_per_sequence);
  ExpectToken(is, binary, "<LabelDim>"); ReadBasicType(is, binary, &label_dim);
  bool e2e; ExpectToken(is, binary, "<End2End>"); ReadBasicType(is, binary, &e2e);
  if (!e2e) ReadSupFst(is, binary, &fst);
  else {
    e2e_fsts.resize(num_sequences); ExpectToken(is, binary, "<Fsts>");
    for (int i = 0; i < num_sequences; i++) ReadSupFst(is, binary, &e2e_fsts[i]);
    ExpectToken(is, binary, "</Fsts>");
  }
  if (PeekToken(is, binary) == 'A') { ExpectToken(is, binary, "<AlignmentPdfs>"); ReadIntegerVector(is, binary, &alignment_pdfs); } else alignment_pdfs.clear();
  ExpectToken(is, binary, "</Supervision>");
}

// chain-supervision.cc:549-609 (Write) and operator==
namespace {
void WriteSupFst(std::ostream &os, bool binary, const fst::StdVectorFst &f) {
  // WriteFstKaldi(os, false, fst): a newline, "src dst ilabel olabel [weight]" lines in state order with the start state first, finals, an empty line
  // (fstext/kaldi-fst-io-inl.h:34-72)
  if (!binary) {
    os << '\n';
    auto state = [&](int32 s) {
      for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) {
        const fst::StdArc &a = it.Value();
        os << s << '\t' << a.nextstate << '\t' << a.ilabel << '\t' << a.olabel;
        if (a.weight != fst::TropicalWeight::One()) os << '\t' << a.weight.Value();
        os << '\n';
      }
      if (f.Final(s) != fst::TropicalWeight::Zero()) { os << s; if (f.Final(s) != fst::TropicalWeight::One()) os << '\t' << f.Final(s).Value(); os << '\n'; }
    };
    if (f.Start() != fst::kNoStateId) { state(f.Start()); for (int32 s = 0; s < f.NumStates(); s++) if (s != f.Start()) state(s); }
    os << '\n'; return;
  }
  // StdCompactAcceptorFst (see ReadCompactAcceptor): unaligned, no symbol tables
  auto wr = [&](const void *p, size_t n) { os.write(reinterpret_cast<const char *>(p), n); };
  auto wstr = [&](const std::string &s) { const int32 n = (int32)s.size(); wr(&n, 4); wr(s.data(), s.size()); };
  struct Elem { int32 label; float weight; int32 nextstate; };
  std::vector<uint32> st; std::vector<Elem> el; int64 narcs = 0;
  for (int32 s = 0; s < f.NumStates(); s++) {
    st.push_back((uint32)el.size());
    if (f.Final(s) != fst::TropicalWeight::Zero()) el.push_back(Elem{fst::kNoLabel, f.Final(s).Value(), fst::kNoStateId});
    for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) { el.push_back(Elem{it.Value().ilabel, it.Value().weight.Value(), it.Value().nextstate}); narcs++; }
  }
  st.push_back((uint32)el.size());
  const int32 magic = 2125659606, version = 2, flags = 0; const uint64 props = 0x0000000000010003ULL /* kExpanded | kMutable-free: acceptor */; const int64 start = f.Start(), nstates = f.NumStates();
  wr(&magic, 4); wstr("compact_acceptor"); wstr("standard"); wr(&version, 4); wr(&flags, 4); wr(&props, 8); wr(&start, 8); wr(&nstates, 8); wr(&narcs, 8);
  wr(st.data(), st.size() * 4); if (!el.empty()) wr(el.data(), el.size() * 12);
}
bool SameFst(const fst::StdVectorFst &a, const fst::StdVectorFst &b) {
  if (a.NumStates() != b.NumStates() || a.Start() != b.Start()) return false;
  for (int32 s = 0; s < a.NumStates(); s++) {
    if (a.Final(s) != b.Final(s) || a.NumArcs(s) != b.NumArcs(s)) return false;
    fst::ArcIterator<fst::StdVectorFst> i(a, s), j(b, s);
    for (; !i.Done(); i.Next(), j.Next()) if (i.Value().ilabel != j.Value().ilabel || i.Value().olabel != j.Value().olabel ||
        i.Value().nextstate != j.Value().nextstate || i.Value().weight != j.Value().weight) return false;
  }
  return true;
}
}  // namespace
void Supervision::Write(std::ostream &os, bool binary) const {
  WriteToken(os, binary, "<Supervision>");
  WriteToken(os, binary, "<Weight>");
  WriteBasicType(os, binary, weight);
  WriteToken(os, binary, "<NumSequences>");
  WriteBasicType(os, binary, num_sequences);
  WriteToken(os, binary, "<FramesPerSeq>"); WriteBasicType(os, binary, frames_per_sequence); WriteToken(os, binary, "<LabelDim>"); WriteBasicType(os, binary, label_dim);
  KALDI_ASSERT(frames_per_sequence > 0 && label_dim > 0 && num_sequences > 0);
  const bool e2e = !e2e_fsts.empty(); WriteToken(os, binary, "<End2End>"); WriteBasicType(os, binary, e2e);
  if (!e2e) WriteSupFst(os, binary, fst);
  else {
    KALDI_ASSERT((int32)e2e_fsts.size() == num_sequences);
    WriteToken(os, binary, "<Fsts>");
    for (int32 i = 0; i < num_sequences; i++) WriteSupFst(os, binary, e2e_fsts[i]);
    WriteToken(os, binary, "</Fsts>");
  }
  if (!alignment_pdfs.empty()) { WriteToken(os, binary, "<AlignmentPdfs>"); WriteIntegerVector(os, binary, alignment_pdfs); }
  WriteToken(os, binary, "</Supervision>");
}
bool Supervision::operator == (const Supervision &other) const {
  if (!(weight == other.weight && num_sequences == other.num_sequences && frames_per_sequence == other.frames_per_sequence && label_dim == other.label_dim &&
      SameFst(fst, other.fst))) return false;
  if (e2e_fsts.size() != other.e2e_fsts.size() || alignment_pdfs != other.alignment_pdfs) return false;
  for (size_t i = 0; i < e2e_fsts.size(); i++) if (!SameFst(e2e_fsts[i], other.e2e_fsts[i])) return false;
  return true;
}

// ---- a merged supervision FST back into its sequences
namespace {
// chain-supervision.cc:663-700 (ComputeFstStateTimes): a state's time = the length of every path to it; returns the path length of the FST
int32 StateTimes(const fst::StdVectorFst &f, std::vector<int32> *t) {
  if (f.Start() != 0) KALDI_ERR << "Expecting input FST start state to be zero";
  const int32 n = f.NumStates(); int32 total = -1; t->assign(n, -1); (*t)[0] = 0;
  for (int32 s = 0; s < n; s++) {
    const int32 nt = (*t)[s] + 1; if (nt <= 0) KALDI_ERR << "Input FST does not have required properties.";
    for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) {
      int32 &r = (*t)[it.Value().nextstate];
      if (r == -1) r = nt;
      else if (r != nt) KALDI_ERR << "Input FST does not have required properties.";
    }
    if (f.Final(s) != fst::TropicalWeight::Zero()) { if (total == -1) total = nt - 1; else if (total != nt - 1) KALDI_ERR << "Input FST does not have required properties."; }
  }
  if (total < 0) KALDI_ERR << "Input FST does not have required properties.";
  return total;
}
// MergeSupervision concatenates the sequences' FSTs and removes the epsilons of the concatenation: a final state f of sequence n (final weight w_f, no arcs of its own) ends up
// with a copy of the arcs of sequence n + 1's start state, every weight raised by w_f.  So at time (n + 1) T the merged FST has the states B = {former finals of n}, all with the
// same arcs up to an additive constant.  Cut there: sequence n + 1 starts in a new state with the arcs of B's first state b0 (as they are: w_b0 rides on them), and b in B is
// final in sequence n with cost w_b - w_b0, read off any pair of corresponding arcs.  Every path of the merged FST has its weight split between the two pieces exactly.
void SplitMergedSupervision(const Supervision &sup, std::vector<int32_t> *state_off, Csr *c) {
  const fst::StdVectorFst &f = sup.fst; const int32 B = sup.num_sequences, T = sup.frames_per_sequence;
  std::vector<int32> t; const int32 len = StateTimes(f, &t);
  if (len != B * T) KALDI_ERR << "Supervision FST has paths of " << len << " arcs, expected num-sequences * frames-per-sequence = " << B * T;
  const int32 n = f.NumStates();
  for (int32 s = 1; s < n; s++) if (t[s] < t[s - 1]) KALDI_ERR << "Supervision FST is not sorted on time (SortBreadthFirstSearch)";
  std::vector<int32> first_at((size_t)B * T + 2, n);      // first state of every time (states are sorted on time)
  for (int32 s = n - 1; s >= 0; s--) first_at[t[s]] = s;
  for (int32 k = B * T; k >= 0; k--) if (first_at[k] == n) first_at[k] = first_at[k + 1];
  state_off->assign(1, 0); c->off.assign(1, 0); c->il.clear(); c->nx.clear(); c->w.clear(); c->fin.clear();
  const float inf = std::numeric_limits<float>::infinity();
  for (int32 q = 0; q < B; q++) {
    // local numbering: 0 = the start; merged states of times (qT, (q + 1) T] follow in their order
    const int32 b0 = first_at[q * T], lo = first_at[q * T + 1], hi = q + 1 < B ? first_at[(q + 1) * T + 1] : n, fin_lo = first_at[(q + 1) * T];
    auto local = [&](int32 s) { return s - lo + 1; };
    auto emit_arcs = [&](int32 s) {
      for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) {
        const fst::StdArc &a = it.Value();
        if (a.nextstate < lo || a.nextstate >= hi) KALDI_ERR << "Supervision FST: an arc leaves its sequence";
                                    c->il.push_back(a.ilabel); c->nx.push_back(local(a.nextstate)); c->w.push_back(a.weight.Value()); } c->off.push_back((int64_t)c->il.size()); };
    emit_arcs(b0); c->fin.push_back(inf);
    // the next sequence's reference state and one of its arcs, to read the final costs off
    const int32 nb0 = fin_lo; fst::StdArc ref_arc; bool have_ref = false;
    if (q + 1 < B) {
      fst::ArcIterator<fst::StdVectorFst> it(f, nb0);
      if (it.Done()) KALDI_ERR << "Supervision FST: a sequence boundary without arcs";
      ref_arc = it.Value();
      have_ref = true;
    }
    for (int32 s = lo; s < hi; s++) {
      if (s < fin_lo) { emit_arcs(s); c->fin.push_back(inf); continue; }
      c->off.push_back((int64_t)c->il.size());      // a final state of this sequence: its arcs belong to the next one
      if (!have_ref) { c->fin.push_back(f.Final(s).Value()); continue; }
      float cost = inf;
      // the corresponding arc: by POSITION first (the boundary states carry copies of the same arc list, so arc 0 of b is arc 0 of b0 -- unambiguous also when two parallel arcs
      // share label and destination), by (label, destination) only if the lists are ordered differently
      {
        fst::ArcIterator<fst::StdVectorFst> it(f, s);
        if (!it.Done() && it.Value().ilabel == ref_arc.ilabel && it.Value().nextstate == ref_arc.nextstate) cost =
            it.Value().weight.Value() - ref_arc.weight.Value();
      }
      if (cost == inf) for (fst::ArcIterator<fst::StdVectorFst> it(f, s); !it.Done(); it.Next()) if (it.Value().ilabel == ref_arc.ilabel &&
          it.Value().nextstate == ref_arc.nextstate) {
        cost = it.Value().weight.Value() - ref_arc.weight.Value();
        break;
      }
      if (cost == inf) KALDI_ERR << "Supervision FST: the states of a sequence boundary do not share their arcs (not the output of MergeSupervision?)";
      c->fin.push_back(cost);
    }
    state_off->push_back((int32_t)c->fin.size());
  }
}
}  // namespace

// ---- the objective (chain-training.cc:242-337 / :86-215)
void ComputeChainObjfAndDeriv(const ChainTrainingOptions &opts, const DenominatorGraph &den_graph, const Supervision &supervision, const CuMatrixBase<BaseFloat> &nnet_output,
                              BaseFloat *objf, BaseFloat *l2_term, BaseFloat *weight, CuMatrixBase<BaseFloat> *nnet_output_deriv, CuMatrix<BaseFloat> *xent_output_deriv) {
  k3_chain_den *den = NULL;
  { std::lock_guard<std::mutex> g(g_mu); auto it = g_den.find(&den_graph); if (it != g_den.end()) den = it->second; }
  if (!den) KALDI_ERR << "ComputeChainObjfAndDeriv: a DenominatorGraph that was not made by its (FST, num-pdfs) constructor";
  const int32 B = supervision.num_sequences, T = supervision.frames_per_sequence, P = supervision.label_dim;
  if (nnet_output.NumRows() != B * T || nnet_output.NumCols() != P) KALDI_ERR << "Network output is " << nnet_output.NumRows() << " x " <<
      nnet_output.NumCols() << ", the supervision wants " << B * T << " x " << P;
  if (nnet_output_deriv && (nnet_output_deriv->NumRows() != nnet_output.NumRows() || nnet_output_deriv->NumCols() != nnet_output.NumCols())) KALDI_ERR <<
      "Derivative matrix of the wrong size";
  k3_chain_supervision *ks = NULL; std::vector<int32_t> so; Csr c;
  const bool e2e = !supervision.e2e_fsts.empty();
  if (!e2e) SplitMergedSupervision(supervision, &so, &c);
  else {
    if ((int32)supervision.e2e_fsts.size() != B) KALDI_ERR << "End-to-end supervision with " << supervision.e2e_fsts.size() << " FSTs for " << B << " sequences";
    so.assign(1, 0); c.off.assign(1, 0);
    for (const fst::StdVectorFst &f : supervision.e2e_fsts) {
      if (f.Start() != 0) KALDI_ERR << "Expecting input FST start state to be zero";
      Csr one; ToCsr(f, &one); const int64_t a0 = c.off.back();
      for (size_t s = 1; s < one.off.size(); s++) c.off.push_back(a0 + one.off[s]);
      c.il.insert(c.il.end(), one.il.begin(), one.il.end());
      c.nx.insert(c.nx.end(), one.nx.begin(), one.nx.end());
      c.w.insert(c.w.end(), one.w.begin(), one.w.end());
      c.fin.insert(c.fin.end(), one.fin.begin(), one.fin.end());
      so.push_back((int32_t)c.fin.size());
    }
  }
  if ((e2e ? k3_chain_supervision_create_e2e : k3_chain_supervision_create)(B, T, P, supervision.weight, so.data(), c.off.data(), c.il.data(), c.nx.data(),
      c.w.data(), c.fin.data(), &ks) != K3_OK) KALDI_ERR << k3_last_error();
  if (xent_output_deriv) xent_output_deriv->Resize(nnet_output.NumRows(), nnet_output.NumCols(), kUndefined);      // (zeroed by the kernel side)
  // the reference applies the out-of-range penalty on every other minibatch, by a coin flip on the host's rand() (chain-training.cc:273-277)
  // -- RandInt(0, 1) is drawn if and only if a derivative is asked for (:107, :249), whatever the penalty's scale: the host's rand() stream then stays in step
  // with the reference's.
  // The end-to-end branch scales the denominator derivative by 1 + opts.lwf_den_scale (:124-128); k3_chain_objf_and_deriv has no such factor: refuse instead of ignoring it.
  // (GenericNumeratorComputation::ForwardBackward's own `ok` only ever turns false under --verbose >= 1 (CheckValues, chain-generic-numerator.cc:281); at the default level the
  // reference's numerator_ok is the finiteness test the kernel side applies.)
  if (e2e && opts.lwf_den_scale != 0.0) KALDI_ERR << "--lwf-den-scale=" << opts.lwf_den_scale << " (end-to-end supervision) is not supported by the MI355X objective";
  k3_chain_training_opts o = {opts.l2_regularize, opts.out_of_range_regularize, opts.leaky_hmm_coefficient, (nnet_output_deriv != NULL && RandInt(0, 1) == 0) ? 1 : 0};
  float fo = 0, fl = 0, fw = 0;
  const int rc = k3_chain_objf_and_deriv(den, ks, &o, nnet_output.Data(), nnet_output.Stride(), nnet_output_deriv ? nnet_output_deriv->Data() : NULL,
      nnet_output_deriv ? nnet_output_deriv->Stride() : 0,
                                         xent_output_deriv ? xent_output_deriv->Data() : NULL, xent_output_deriv ? xent_output_deriv->Stride() : 0, &fo, &fl,
                                             &fw, k3_adapter_stream());
  k3_chain_supervision_destroy(ks);
  if (rc != K3_OK) KALDI_ERR << k3_last_error();
  *objf = fo; *l2_term = fl; *weight = fw;
}

}  // namespace chain
}  // namespace kaldi
