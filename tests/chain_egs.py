"""TEST INFRASTRUCTURE: writes NnetChainExample archives in Kaldi's TEXT form (nnet3-chain-copy-egs ark,t:) -- what nnet3-chain-get-egs | nnet3-chain-merge-egs hand to nnet3-chain-train.
The reference's tools that make such examples need OpenFst (not vendored); the archive layout is restated from the reference's Write functions:
  NnetChainExample::Write  nnet3/nnet-chain-example.cc:141-155   <Nnet3ChainEg> <NumInputs> n  NnetIo...  <NumOutputs> m  NnetChainSupervision...  </Nnet3ChainEg>
  NnetIo::Write            nnet3/nnet-example.cc:31-39            <NnetIo> name  indexes  features  </NnetIo>
  WriteIndexVector / Index::Write  nnet3/nnet-common.cc:126-140, :25-33   <I1V> size  (<I1> n t x)...
  NnetChainSupervision::Write      nnet3/nnet-chain-example.cc:28-37      <NnetChainSup> name  indexes  supervision  <DW2> deriv-weights  </NnetChainSup>
  Supervision::Write       chain/chain-supervision.cc:549-609     <Supervision> <Weight> w <NumSequences> B <FramesPerSeq> T <LabelDim> P <End2End> F|T  fst | <Fsts> fst... </Fsts>  </Supervision>
  WriteFstKaldi (text)     fstext/kaldi-fst-io-inl.h:34-72        a newline, "src dst ilabel olabel [weight]" / "final [weight]" lines (start state first), an empty line
The reading side is the reference's own code (NnetChainExample::Read & co compiled unmodified into kaldi_amd/adapter/_build/nnet3-chain-train-egs / nnet3-chain-copy-egs) except
Supervision::Read (kaldi_amd/adapter/chain-k3.cc)."""
import numpy as np

def _matrix_text(m):
    m = np.asarray(m, np.float32)
    if m.shape[0] == 0: return " [ ]\n"
    return " [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m[:-1]) + ("\n" if m.shape[0] > 1 else "") + "  " + " ".join(repr(float(x)) for x in m[-1]) + " ]\n"

def _vector_text(v): return " [ " + " ".join(repr(float(x)) for x in np.asarray(v, np.float32)) + " ]\n"

def _indexes_text(idx): return "<I1V> %d " % len(idx) + "".join("<I1> %d %d %d " % (n, t, x) for n, t, x in idx)

def _fst_text(f):
    """kaldi_amd.fst.Fst (acceptor: olabel = ilabel) -> WriteFstKaldi's text form; the start state's lines come first"""
    def state(s):
        out = []
        for a in range(int(f.arc_offsets[s]), int(f.arc_offsets[s + 1])):
            w = float(f.weight[a]); out.append("%d\t%d\t%d\t%d" % (s, int(f.nextstate[a]), int(f.ilabel[a]), int(f.ilabel[a])) + ("" if w == 0.0 else "\t" + repr(w)))
        if np.isfinite(f.final[s]): out.append("%d" % s + ("" if float(f.final[s]) == 0.0 else "\t" + repr(float(f.final[s]))))
        return out
    lines = state(f.start)
    for s in range(f.num_states):
        if s != f.start: lines += state(s)
    return "\n" + "\n".join(lines) + "\n\n"

def supervision_text(weight, num_sequences, frames_per_sequence, label_dim, fst=None, e2e_fsts=None):
    s = "<Supervision> <Weight> %r <NumSequences> %d <FramesPerSeq> %d <LabelDim> %d <End2End> %s " % (float(weight), num_sequences, frames_per_sequence, label_dim, "T" if e2e_fsts else "F")
    if e2e_fsts: s += "<Fsts> " + "".join(_fst_text(f) for f in e2e_fsts) + "</Fsts> "
    else: s += _fst_text(fst)
    return s + "</Supervision> "

def write_chain_egs_text(path, egs):
    """egs: [(key, [(input name, [(n, t, x)...], matrix)...], [(output name, [(n, t, x)...], supervision text (supervision_text above), deriv weights)...])]"""
    with open(path, "w") as fh:
        for key, inputs, outputs in egs:
            fh.write(key + " <Nnet3ChainEg> <NumInputs> %d " % len(inputs))
            for name, idx, mat in inputs: fh.write("<NnetIo> " + name + " " + _indexes_text(idx) + _matrix_text(mat) + "</NnetIo> ")
            fh.write("<NumOutputs> %d " % len(outputs))
            for name, idx, sup, dw in outputs: fh.write("<NnetChainSup> " + name + " " + _indexes_text(idx) + sup + "<DW2> " + _vector_text(dw) + "</NnetChainSup> ")
            fh.write("</Nnet3ChainEg> \n")

def minibatch(key, feats, num_sequences, frames_per_sequence, subsampling, left, right, label_dim, merged_fst=None, e2e_fsts=None, weight=1.0, deriv_weights=None):
    """one merged example as nnet3-chain-merge-egs makes it: input rows (n, t) with t = -left .. (T - 1) s + right, sequence-minor (all n of a t together, the order a merged example has);
    output rows (n, t = f s), frame-major as NnetChainSupervision::CheckDim demands"""
    B, T, s = num_sequences, frames_per_sequence, subsampling
    in_idx = [(n, t, 0) for t in range(-left, (T - 1) * s + right + 1) for n in range(B)]; out_idx = [(n, f * s, 0) for f in range(T) for n in range(B)]
    assert np.asarray(feats).shape[0] == len(in_idx)
    dw = np.ones(B * T, np.float32) if deriv_weights is None else deriv_weights
    return (key, [("input", in_idx, feats)], [("output", out_idx, supervision_text(weight, B, T, label_dim, fst=merged_fst, e2e_fsts=e2e_fsts), dw)])
