"""NnetChainExample archives through the reference's own reader / writer code over the adapter's Supervision I/O (kaldi_amd/adapter/chain-k3.cc), on the host: text -> binary -> text
is a fixed point and keeps every arc of the supervision FST.  The binary form of the FST is OpenFst's compact_acceptor container as chain-k3.cc restates it (UNPINNED against OpenFst
itself: /root/reference does not vendor it); what this test pins is that the reader and the writer agree with each other and with the text form, and the header fields the format documents."""
import os, re, struct, subprocess, numpy as np, pytest
from kaldi_amd import synth
from tests import chain_egs as ce
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CP = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-copy-egs")

def _arcs_of_text(txt):
    """(src, dst, label, weight) lines and final states of every FST printed in an ark,t archive, in file order"""
    arcs, finals = [], []
    for line in txt.splitlines():
        f = line.split("\t")
        if len(f) >= 4 and all(re.fullmatch(r"-?\d+", x) for x in f[:4]): arcs.append((int(f[0]), int(f[1]), int(f[2]), float(f[4]) if len(f) > 4 else 0.0))
        elif 1 <= len(f) <= 2 and re.fullmatch(r"\d+", f[0]) and (len(f) == 1 or re.fullmatch(r"-?[0-9.eE+-]+", f[1])): finals.append((int(f[0]), float(f[1]) if len(f) > 1 else 0.0))
    return arcs, finals

@pytest.mark.skipif(not os.path.exists(CP), reason="kaldi_amd/adapter/_build/nnet3-chain-copy-egs is built by kaldi_amd/adapter/build.sh where /root/reference exists")
def test_chain_example_archives_round_trip_text_binary_text(tmp_path):
    td = str(tmp_path); B, T, P, s = 4, 6, 30, 3; lc = rc = 5; Tin = (T - 1) * s + 1 + lc + rc
    egs, merged_all = [], []
    for m in range(3):
        rng = np.random.default_rng(m); x = rng.standard_normal((Tin * B, 10)).astype(np.float32)
        fsts = [synth.make_supervision_fst(T, P, seed=10 + i + 50 * m) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts); merged_all.append(merged)
        egs.append(ce.minibatch(f"mb{m}", x, B, T, s, lc, rc, P, merged_fst=merged, deriv_weights=(rng.uniform(0, 1, B * T) > 0.2).astype(np.float32) if m == 1 else None))
    ce.write_chain_egs_text(f"{td}/a.txt", egs)
    run = lambda a, b: subprocess.run([CP, a, b], capture_output=True, text=True)
    r = run(f"ark,t:{td}/a.txt", f"ark:{td}/b.egs"); assert r.returncode == 0 and "wrote 3" in r.stderr, r.stderr[-800:]
    r = run(f"ark:{td}/b.egs", f"ark,t:{td}/c.txt"); assert r.returncode == 0 and "wrote 3" in r.stderr, r.stderr[-800:]
    r = run(f"ark,t:{td}/c.txt", f"ark,t:{td}/d.txt"); assert r.returncode == 0
    c, d = open(f"{td}/c.txt").read(), open(f"{td}/d.txt").read()
    assert c == d                                                          # text is a fixed point
    r = run(f"ark:{td}/b.egs", f"ark:{td}/e.egs"); assert r.returncode == 0
    assert open(f"{td}/b.egs", "rb").read() == open(f"{td}/e.egs", "rb").read()      # and so is binary
    # every arc and final weight of the merged supervision FSTs survives the compact-acceptor container (float32 weights)
    arcs, finals = _arcs_of_text(c); want_arcs, want_finals = [], []
    for f in merged_all:
        order = [f.start] + [q for q in range(f.num_states) if q != f.start]
        for q in order:
            for a in range(int(f.arc_offsets[q]), int(f.arc_offsets[q + 1])): want_arcs.append((q, int(f.nextstate[a]), int(f.ilabel[a]), float(np.float32(f.weight[a]))))
            if np.isfinite(f.final[q]): want_finals.append((q, float(np.float32(f.final[q]))))
    assert len(arcs) == len(want_arcs) and len(finals) == len(want_finals)
    assert sorted((a[0], a[1], a[2]) for a in arcs) == sorted((a[0], a[1], a[2]) for a in want_arcs)
    assert np.allclose(sorted(a[3] for a in arcs), sorted(a[3] for a in want_arcs), rtol=0, atol=1e-6) and np.allclose(sorted(x[1] for x in finals), sorted(x[1] for x in want_finals), rtol=0, atol=1e-6)
    # the container's header: OpenFst's magic number and the type strings of a compact acceptor over the tropical semiring
    b = open(f"{td}/b.egs", "rb").read(); i = b.find(struct.pack("<i", 2125659606)); assert i >= 0
    n = struct.unpack_from("<i", b, i + 4)[0]; assert b[i + 8:i + 8 + n] == b"compact_acceptor"
    j = i + 8 + n; n2 = struct.unpack_from("<i", b, j)[0]; assert b[j + 4:j + 4 + n2] == b"standard"
