"""GPU: include/k3_cuda_decoder.h -- kaldi::cuda_decoder::CudaFst / CudaDecoder with the reference's signatures (cudadecoder/cuda-fst.h:62-149,
cuda-decoder.h:224-345) over the C ABI -- compiled into a C++ caller (tests/adapter/cuda_decoder_example.cc, built by kaldi_amd/adapter/build.sh)
that goes InitDecoding -> AdvanceDecoding(lanes_assignements) frame by frame (or 7 frames per call) -> GetBestPath / GetPartialHypothesis ->
GetRawLattice.  Its raw lattice must equal the one of the REFERENCE's CPU LatticeFasterDecoder (oracle/_ref/bin/ref-lattice-decoder, same file
protocol) bit for bit, up to the numbering of the states."""
import os, subprocess, numpy as np, pytest
from tests import decoder_cases as dcases, lattice_sig as lsig
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "cuda-decoder-example")

@pytest.mark.parametrize("name,step", [("default", 1), ("max_active", 7), ("hash_ratio", 1), ("long", 50)])
def test_cuda_decoder_adapter_equals_the_reference_cpu_decoder(name, step, tmp_path):
    from oracle import ref_decoder as rd, lattice_oracle as lo
    if not os.path.exists(EXE): pytest.fail("kaldi_amd/adapter/_build/cuda-decoder-example is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    f, t2p, ll, kw = dcases.make(name); cfg = lo.Config(**kw)
    a, b = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(a, "wb") as fh:      # the input file of oracle/ref_tools/ref_lattice_decoder.cc
        np.array([0x4b33, f.num_states, f.start, f.ilabel.size, ll.shape[0], ll.shape[1], t2p.size, cfg.max_active, cfg.min_active, cfg.prune_interval], np.int32).tofile(fh)
        np.array([cfg.beam, cfg.lattice_beam, cfg.beam_delta, cfg.hash_ratio, cfg.prune_scale], np.float32).tofile(fh)
        for x, dt in ((f.arc_offsets, np.int32), (f.ilabel, np.int32), (f.olabel, np.int32), (f.nextstate, np.int32), (f.weight, np.float32), (f.final, np.float32), (t2p, np.int32), (ll, np.float32)):
            np.ascontiguousarray(x, dt).tofile(fh)
    r = subprocess.run([EXE, a, b, str(step)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"frames decoded {ll.shape[0]}" in r.stderr
    with open(b, "rb") as fh:
        ns, na, start, reached, nframes = np.fromfile(fh, np.int64, 5)
        frame = np.fromfile(fh, np.int32, ns); fg = np.fromfile(fh, np.float32, ns); fa = np.fromfile(fh, np.float32, ns)
        src, dst, il, ol = (np.fromfile(fh, np.int32, na) for _ in range(4)); g = np.fromfile(fh, np.float32, na); ac = np.fromfile(fh, np.float32, na)
    got = dict(frame=frame, final_graph=fg, final_ac=fa, src=src, dst=dst, ilabel=il, olabel=ol, graph=g, ac=ac, start=int(start))
    if rd.available():
        ref = rd.decode(f, ll, t2p, cfg)
        assert lsig.canonical_of_reference(got) == lsig.canonical_of_reference(ref)
    else:
        lat, _ = lo.decode(f, ll, t2p, cfg, 0)
        assert lsig.canonical_of_reference(got) == lsig.canonical_of_raw(lat)
