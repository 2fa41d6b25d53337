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
    assert "two channels interleaved on one lane: lattices identical" in r.stderr      # nlanes = 1 < nchannels = 2
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


def test_pipeline_class_with_the_references_constructor_types(tmp_path):
    """include/k3_batched_pipeline.h: kaldi::cuda_decoder::BatchedThreadedNnet3CudaPipeline2(config, fst::Fst<fst::StdArc>, nnet3::AmNnetSimple, TransitionModel) and
    DecodeWithCallback(std::shared_ptr<WaveData>, void(CompactLattice &)) -- the reference's signatures (batched-threaded-nnet3-cuda-pipeline2.h:153-200) -- in a caller that reads
    final.mdl with the reference's own TransitionModel / AmNnetSimple readers (tests/adapter/cuda_pipeline_example.cc).  The lattices its callbacks receive must be the ones
    the batched-wav-nnet3-cuda2 program writes for the same files."""
    import struct
    from kaldi_amd import synth
    from oracle import kaldi_io as kio
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "cuda-pipeline-example")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/cuda-pipeline-example is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 12000]
    for i, n in enumerate(lens): kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(n, 40 + i))
    open(f"{td}/wav.scp", "w").write("".join(f"utt{i} {td}/u{i}.wav\n" for i in range(len(lens))))
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    g = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); g.write_openfst(f"{td}/HCLG.fst")
    with open(f"{td}/graph.bin", "wb") as fh:
        fh.write(struct.pack("<3i", g.num_states, g.start, int(g.ilabel.size)))
        for x, dt in ((g.arc_offsets, np.int32), (g.ilabel, np.int32), (g.olabel, np.int32), (g.nextstate, np.int32), (g.weight, np.float32), (g.final, np.float32)): np.ascontiguousarray(x, dt).tofile(fh)
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([exe, f"{td}/final.mdl", f"{td}/graph.bin", f"{td}/wav.scp", f"{td}/fbank.conf", f"ark,t:{td}/cls.txt"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "Decoded 4 utterances, 0 with errors." in r.stderr, r.stderr[-3000:]
    p = subprocess.run([os.path.join(ROOT, "kaldi_amd", "bin", "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0",
                        "--lattice-beam=8.0", "--max-active=10000", "--max-batch-size=2", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/prog.txt"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    # (the model went through the reference's reader and writer on its way into the class: the last digit of a cost may differ; everything else must be the same)
    from tests import lattice_cases as lc
    a, b = lc.parse_compact_text(open(f"{td}/cls.txt").read()), lc.parse_compact_text(open(f"{td}/prog.txt").read())
    assert list(a) == list(b) == ["utt0", "utt1", "utt2", "utt3"]
    for k in a:
        assert len(a[k]["arcs"]) == len(b[k]["arcs"]) > 0 and sorted(a[k]["finals"]) == sorted(b[k]["finals"]), k
        for x, y in zip(a[k]["arcs"], b[k]["arcs"]):
            assert x[:3] == y[:3] and list(x[5]) == list(y[5]) and abs(x[3] - y[3]) <= 2e-3 and abs(x[4] - y[4]) <= 2e-3, (k, x, y)


def test_online_pipeline_class_with_the_references_constructor_types(tmp_path):
    """include/k3_batched_online_pipeline.h: kaldi::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline(config, fst::Fst<fst::StdArc>, nnet3::AmNnetSimple, TransitionModel),
    TryInitCorrID / SetLatticeCallback(void(CompactLattice &)) / DecodeBatch(corr_ids, std::vector<SubVector<BaseFloat>>, is_first_chunk, is_last_chunk, partial hypotheses) -- the
    reference's signatures (batched-threaded-nnet3-cuda-online-pipeline.h:119-330) -- in a caller that plays wave files as concurrent streams, more streams than channels
    (tests/adapter/cuda_online_pipeline_example.cc).  The lattices its callbacks receive must be the ones the offline program writes for the same files (streaming == offline)."""
    import struct
    from kaldi_amd import synth
    from oracle import kaldi_io as kio
    from tests import lattice_cases as lc
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "cuda-online-pipeline-example")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/cuda-online-pipeline-example is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 12000, 5000]
    for i, n in enumerate(lens): kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(n, 40 + i))
    open(f"{td}/wav.scp", "w").write("".join(f"utt{i} {td}/u{i}.wav\n" for i in range(len(lens))))
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    g = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); g.write_openfst(f"{td}/HCLG.fst")
    with open(f"{td}/graph.bin", "wb") as fh:
        fh.write(struct.pack("<3i", g.num_states, g.start, int(g.ilabel.size)))
        for x, dt in ((g.arc_offsets, np.int32), (g.ilabel, np.int32), (g.olabel, np.int32), (g.nextstate, np.int32), (g.weight, np.float32), (g.final, np.float32)): np.ascontiguousarray(x, dt).tofile(fh)
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    r = subprocess.run([exe, f"{td}/final.mdl", f"{td}/graph.bin", f"{td}/wav.scp", f"{td}/fbank.conf", f"ark,t:{td}/cls.txt"], capture_output=True, text=True, env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"))
    assert r.returncode == 0 and "Decoded 5 utterances, 0 with errors" in r.stderr, r.stderr[-3000:]
    assert int(r.stderr.split(" non-empty partial hypotheses")[0].split()[-1]) > 0
    p = subprocess.run([os.path.join(ROOT, "kaldi_amd", "bin", "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0",
                        "--lattice-beam=8.0", "--max-active=10000", "--max-batch-size=2", "--determinize-lattice=false", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/prog.txt"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    a, b = lc.parse_compact_text(open(f"{td}/cls.txt").read()), lc.parse_compact_text(open(f"{td}/prog.txt").read())
    assert sorted(a) == sorted(b) == [f"utt{i}" for i in range(5)]
    for k in a:      # (raw lattices as CompactLattices: state numbers follow the order the GPU emitted the arcs in; the model went through the reference's reader and writer: last digit of a cost)
        key = lambda x: (x[2], tuple(x[5]), round(float(x[3]), 2), round(float(x[4]), 2))
        assert len(a[k]["arcs"]) == len(b[k]["arcs"]) > 0 and sorted(map(key, a[k]["arcs"])) == sorted(map(key, b[k]["arcs"])), k
