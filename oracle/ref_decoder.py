"""oracle/ref_decoder.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Front-end of oracle/_ref/bin/ref-lattice-decoder: the REFERENCE's own decoder/lattice-faster-decoder.cc, compiled unmodified
against the OpenFst stand-in of third_party/minifst (build: oracle/build_ref.sh).  Used only to pin the restated decoder
oracle (oracle/lattice_faster_oracle.cc) in tests/test_oracle_decoder.py."""
import os, subprocess, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "_ref", "bin", "ref-lattice-decoder")

def available(): return os.path.exists(EXE)

def decode(fst, loglikes, tid2pdf, cfg):
    """fst: kaldi_amd.fst.Fst; cfg: oracle.lattice_oracle.Config.  Returns dict(frame, final_graph, final_ac, src, dst, ilabel, olabel,
    graph, ac, start, reached_final, num_frames): the raw lattice before Connect; states carry their frame only (the reference's
    tokens do not remember their graph state)."""
    ll = np.ascontiguousarray(loglikes, np.float32); t2p = np.ascontiguousarray(tid2pdf, np.int32)
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(a, "wb") as f:
            np.array([0x4b33, fst.num_states, fst.start, fst.ilabel.size, ll.shape[0], ll.shape[1], t2p.size, cfg.max_active, cfg.min_active, cfg.prune_interval], np.int32).tofile(f)
            np.array([cfg.beam, cfg.lattice_beam, cfg.beam_delta, cfg.hash_ratio, cfg.prune_scale], np.float32).tofile(f)
            for x, dt in ((fst.arc_offsets, np.int32), (fst.ilabel, np.int32), (fst.olabel, np.int32), (fst.nextstate, np.int32), (fst.weight, np.float32), (fst.final, np.float32), (t2p, np.int32), (ll, np.float32)):
                np.ascontiguousarray(x, dt).tofile(f)
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([EXE, a, b], capture_output=True, text=True, env=env)
        if r.returncode != 0: raise RuntimeError("ref-lattice-decoder failed: " + r.stderr[-2000:])
        with open(b, "rb") as f:
            ns, na, start, reached, nframes = np.fromfile(f, np.int64, 5)
            frame = np.fromfile(f, np.int32, ns); fg = np.fromfile(f, np.float32, ns); fa = np.fromfile(f, np.float32, ns)
            src, dst, il, ol = (np.fromfile(f, np.int32, na) for _ in range(4)); g = np.fromfile(f, np.float32, na); ac = np.fromfile(f, np.float32, na)
            secs = float(np.fromfile(f, np.float64, 1)[0])
    return dict(frame=frame, final_graph=fg, final_ac=fa, src=src, dst=dst, ilabel=il, olabel=ol, graph=g, ac=ac, start=int(start), reached_final=bool(reached), num_frames=int(nframes), decode_seconds=secs, log=r.stderr)
