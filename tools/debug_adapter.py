import os, sys, subprocess, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import decoder_cases as dcases, lattice_sig as lsig
from oracle import ref_decoder as rd, lattice_oracle as lo
EXE = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "cuda-decoder-example")
for name in ("default", "long", "max_active"):
    f, t2p, ll, kw = dcases.make(name); cfg = lo.Config(**kw)
    ref = rd.decode(f, ll, t2p, cfg)
    for step in (1, 7, 1000):
        with tempfile.TemporaryDirectory() as td:
            a, b = td + "/in.bin", td + "/out.bin"
            with open(a, "wb") as fh:
                np.array([0x4b33, f.num_states, f.start, f.ilabel.size, ll.shape[0], ll.shape[1], t2p.size, cfg.max_active, cfg.min_active, cfg.prune_interval], np.int32).tofile(fh)
                np.array([cfg.beam, cfg.lattice_beam, cfg.beam_delta, cfg.hash_ratio, cfg.prune_scale], np.float32).tofile(fh)
                for x, dt in ((f.arc_offsets, np.int32), (f.ilabel, np.int32), (f.olabel, np.int32), (f.nextstate, np.int32), (f.weight, np.float32), (f.final, np.float32), (t2p, np.int32), (ll, np.float32)):
                    np.ascontiguousarray(x, dt).tofile(fh)
            r = subprocess.run([EXE, a, b, str(step)], capture_output=True, text=True)
            with open(b, "rb") as fh:
                ns, na, start, reached, nframes = np.fromfile(fh, np.int64, 5)
                frame = np.fromfile(fh, np.int32, ns); fg = np.fromfile(fh, np.float32, ns); fa = np.fromfile(fh, np.float32, ns)
                src, dst, il, ol = (np.fromfile(fh, np.int32, na) for _ in range(4)); g = np.fromfile(fh, np.float32, na); ac = np.fromfile(fh, np.float32, na)
            got = dict(frame=frame, final_graph=fg, final_ac=fa, src=src, dst=dst, ilabel=il, olabel=ol, graph=g, ac=ac, start=int(start))
            same = lsig.canonical_of_reference(got) == lsig.canonical_of_reference(ref)
            print(name, "step", step, "states", ns, ref["frame"].size, "arcs", na, ref["src"].size, "finals", int(np.isfinite(fg).sum()), int(np.isfinite(ref["final_graph"]).sum()),
                  "frame hist equal", np.array_equal(np.bincount(frame), np.bincount(ref["frame"])), "SAME" if same else "DIFF", r.stderr.strip()[-80:])
