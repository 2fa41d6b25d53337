"""GPU tests of the drop-in command-line programs (kaldi_amd/bin, built from kaldi_amd/host):
compute-fbank-feats-cuda / compute-mfcc-feats-cuda against the reference's own compute-fbank-feats / compute-mfcc-feats
binaries (oracle/_ref) on the same wav.scp, and batched-wav-nnet3-cuda2 end to end (final.mdl + OpenFst HCLG + wav.scp ->
lattice archive) against the Python-API pipeline over the same C ABI and against the oracle chain."""
import os, subprocess, numpy as np, pytest, torch
from kaldi_amd import synth
from kaldi_amd.lattice import RawLattice
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi_amd", "bin"); REF = os.path.join(ROOT, "oracle", "_ref", "bin")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")

def _wavs(td, lens, seed=50):
    from oracle import kaldi_io as kio
    lines = []
    for i, n in enumerate(lens):
        kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(n, seed + i)); lines.append(f"utt{i} {td}/u{i}.wav")
    open(f"{td}/wav.scp", "w").write("\n".join(lines) + "\n")

@pytest.mark.parametrize("kind,flags", [("fbank", ["--dither=0", "--num-mel-bins=40"]), ("mfcc", ["--dither=0", "--num-mel-bins=40", "--num-ceps=40", "--low-freq=20", "--high-freq=-400"]),
                                        ("fbank", ["--dither=0", "--snip-edges=false", "--use-energy=true", "--window-type=hamming"])])
def test_feature_programs_match_the_reference_binaries(tmp_path, kind, flags):
    from oracle import kaldi_io as kio
    if not os.path.exists(os.path.join(REF, f"compute-{kind}-feats")): pytest.skip("oracle/_ref not present")
    td = str(tmp_path); _wavs(td, [16000, 4001, 23001, 399 + 160 * 3])
    r = subprocess.run([os.path.join(REF, f"compute-{kind}-feats")] + flags + [f"scp:{td}/wav.scp", f"ark:{td}/ref.ark"], env=ENV, capture_output=True, text=True); assert r.returncode == 0, r.stderr
    g = subprocess.run([os.path.join(BIN, f"compute-{kind}-feats-cuda")] + flags + [f"scp:{td}/wav.scp", f"ark:{td}/gpu.ark"], capture_output=True, text=True); assert g.returncode == 0, g.stderr
    a, b = kio.read_ark(f"{td}/ref.ark"), kio.read_ark(f"{td}/gpu.ark")
    assert list(a) == list(b)
    # log-mel outputs: 1e-4.  Hi-res MFCC (40 cepstra, lifter 22): the liftering multiplies high cepstra by up to 12, and with
    # it the float32 differences between the two FFT algorithms (16x16 Stockham here, split-radix there) -> 1e-4 * 6
    tol = 1e-4 if kind == "fbank" else 6e-4
    for k in a:
        assert a[k].shape == b[k].shape and np.abs(a[k] - b[k]).max() <= tol, (k, np.abs(a[k] - b[k]).max())
    assert "Done 4 out of 4 utterances" in g.stderr

def _parse_text_lattices(path, start_state_of=None):
    lats, key, arcs, fins = {}, None, [], {}
    for line in open(path):
        line = line.rstrip("\n")
        if key is None:
            if line.strip(): key = line.strip(); arcs, fins = [], {}
            continue
        if line == "":
            lats[key] = (arcs, fins); key = None; continue
        f = line.split("\t")
        if len(f) >= 4:
            g, a = (0.0, 0.0) if len(f) == 4 else map(float, f[4].split(","))
            arcs.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), np.float32(g), np.float32(a)))
        else:
            fins[int(f[0])] = np.float32(0.0) if len(f) == 1 else np.float32(float(f[1].split(",")[0]))
    return lats

def _compact_best_words(c, want="words"):
    """cheapest path of a parsed CompactLattice (tests/lattice_cases.parse_compact_text): its word labels or its transition-ids"""
    by_src = {}
    for a in c["arcs"]: by_src.setdefault(a[0], []).append(a)
    memo = {}
    def best(s):
        if s not in memo:
            cand = [(c["finals"][s][0] + c["finals"][s][1], (), tuple(c["finals"][s][2]))] if s in c["finals"] else []
            for a in by_src.get(s, []):
                bc, bw, bt = best(a[1]); cand.append((a[3] + a[4] + bc, (a[2],) + bw, tuple(a[5]) + bt))
            memo[s] = min(cand) if cand else (float("inf"), (), ())
        return memo[s]
    import sys; sys.setrecursionlimit(20000)
    r = best(c["start"])
    return list(r[1] if want == "words" else r[2])

def test_batched_wav_nnet3_cuda2_end_to_end(tmp_path):
    from kaldi_amd import feat, nnet3, decoder
    from oracle import kaldi_io as kio
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40   # comment\n--dither=0\n")
    cmd = [os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0",
           "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", "--max-batch-size=2", "--determinize-lattice=false", "--write-compact=false", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/lat.txt"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Decoded 3 utterances, 0 with errors." in r.stderr and "RealTimeX:" in r.stderr
    got = _parse_text_lattices(f"{td}/lat.txt")
    assert list(got) == ["utt0", "utt1", "utt2"]
    # the same path through the Python API
    dev = torch.device("cuda:0"); t2p = synth.tid2pdf(N)
    sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    waves = [kio.read_wav(f"{td}/u{i}.wav")[0].astype(np.float32) for i in range(3)]
    wo, fo, total, fo_h = sf.offsets([len(w) for w in waves], dev)
    feats = sf.ComputeFeatures(torch.from_numpy(np.concatenate(waves)).to(dev), wo, fo, total)
    nn = nnet3.Nnet(f"{td}/final.mdl"); nb = nnet3.NnetBatch(nn, [fo_h[i + 1] - fo_h[i] for i in range(3)], 3)
    ll = nb.forward(feats)
    dec = decoder.CudaDecoder(decoder.CudaFst(graph, t2p), decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, literal_order=1), 3, N)      # the program's default: --literal-order=true
    dec.DecodeBatch(ll, nb.out_offsets); lats = dec.GetRawLattices()
    for u, key in enumerate(got):
        ref = lats[u].connect()
        arcs, fins = got[key]
        assert len(arcs) == ref.num_arcs and len(arcs) > 0, (key, len(arcs), ref.num_arcs)
        # the text format prints 6 significant digits: compare label multisets exactly and costs loosely
        assert sorted((a[2], a[3]) for a in arcs) == sorted(zip(ref.arc_ilabel.tolist(), ref.arc_olabel.tolist()))
        assert abs(sum(float(a[4]) for a in arcs) - float(ref.arc_graph.astype(np.float64).sum())) < 1e-3 * len(arcs)
        assert abs(sum(float(a[5]) for a in arcs) - float(ref.arc_ac.astype(np.float64).sum())) < 1e-3 * len(arcs)
        assert len(fins) == int(np.isfinite(ref.st_final).sum())
    # binary archive: same lattices, exact float bits
    cmd[-1] = f"ark:{td}/lat.ark"
    assert subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    assert os.path.getsize(f"{td}/lat.ark") > 1000 and open(f"{td}/lat.ark", "rb").read(9) == b"utt0 \xd6\xfd\xb2\x7e"
    # default: determinized CompactLattices.  Must equal what lattice-determinize-phone-pruned (same host code, CPU program) makes of the
    # raw binary archive above, and its best path must spell the raw lattice's best word sequence.
    from tests import lattice_cases as lc
    cmd_det = [c for c in cmd if not c.startswith("--determinize") and not c.startswith("--write-compact")]; cmd_det[-1] = f"ark,t:{td}/det.txt"
    r = subprocess.run(cmd_det, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r2 = subprocess.run([os.path.join(BIN, "lattice-determinize-phone-pruned"), "--beam=8.0", "--acoustic-scale=1.0", f"{td}/final.mdl", f"ark:{td}/lat.ark", f"ark,t:{td}/det2.txt"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    det, det2 = lc.parse_compact_text(open(f"{td}/det.txt").read()), lc.parse_compact_text(open(f"{td}/det2.txt").read())
    assert list(det) == list(det2) == ["utt0", "utt1", "utt2"]
    for u, key in enumerate(det):
        # state numbers differ (the CPU program sorts topologically); the arcs without them must agree
        canon = lambda c: sorted((a[2], a[5], round(a[3], 2), round(a[4], 2)) for a in c["arcs"])
        assert canon(det[key]) == canon(det2[key]) and len(det[key]["arcs"]) > 0, key
        by_src = {}
        for a_ in det[key]["arcs"]:
            assert (a_[0], a_[2]) not in by_src and a_[2] != 0          # deterministic on words, no epsilons
            by_src[(a_[0], a_[2])] = a_
        assert _compact_best_words(det[key]) == lats[u].connect().best_path()[1], key
    # --determinize-lattice=false alone: the raw lattice re-packed as a CompactLattice (ConvertLattice), = k3-host-tool convert-lattice
    cmd_cv = [c for c in cmd if not c.startswith("--write-compact")]; cmd_cv[-1] = f"ark,t:{td}/conv.txt"
    assert subprocess.run(cmd_cv, capture_output=True, text=True).returncode == 0
    assert subprocess.run([os.path.join(BIN, "k3-host-tool"), "convert-lattice", f"ark:{td}/lat.ark", f"ark,t:{td}/conv2.txt"], capture_output=True).returncode == 0
    cv, cv2 = lc.parse_compact_text(open(f"{td}/conv.txt").read()), lc.parse_compact_text(open(f"{td}/conv2.txt").read())
    for u, key in enumerate(cv):
        canon = lambda c: sorted((a[2], a[5], round(a[3], 2), round(a[4], 2)) for a in c["arcs"])
        assert canon(cv[key]) == canon(cv2[key]) and 0 < len(cv[key]["arcs"]) <= lats[u].connect().num_arcs, key
        assert sum(len(a[5]) for a in cv[key]["arcs"]) == int((lats[u].connect().arc_ilabel != 0).sum())          # every transition-id is on exactly one arc

def test_nnet3_compute_matches_the_reference_binary_incl_compressed_archives(tmp_path):
    """same model file, same feature archive (also as a COMPRESSED archive and an scp with byte offsets written by the
    reference's copy-feats) -> our nnet3-compute vs the reference's nnet3-compute"""
    from oracle import kaldi_io as kio
    if not os.path.exists(os.path.join(REF, "nnet3-compute")): pytest.skip("oracle/_ref not present")
    td = str(tmp_path); rng = np.random.default_rng(3)
    feats = {f"utt{i}": (rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32) for i, T in enumerate([83, 7, 250])}
    kio.write_ark(f"{td}/f.ark", feats)
    mdl = os.path.join(ROOT, "tests", "golden", "nnet_small.raw")
    for s in (1, 3):
        assert subprocess.run([os.path.join(REF, "nnet3-compute"), "--use-gpu=no", f"--frame-subsampling-factor={s}", mdl, f"ark:{td}/f.ark", f"ark:{td}/ref{s}.ark"], env=ENV, capture_output=True).returncode == 0
        g = subprocess.run([os.path.join(BIN, "nnet3-compute"), f"--frame-subsampling-factor={s}", mdl, f"ark:{td}/f.ark", f"ark:{td}/gpu{s}.ark"], capture_output=True, text=True); assert g.returncode == 0, g.stderr
        a, b = kio.read_ark(f"{td}/ref{s}.ark"), kio.read_ark(f"{td}/gpu{s}.ark")
        assert list(a) == list(b)
        for k in a: assert a[k].shape == b[k].shape and np.abs(a[k] - b[k]).max() <= 1e-4, (s, k, np.abs(a[k] - b[k]).max())
    # compressed archive + scp with offsets, produced by the reference's copy-feats: both programs see the same (lossy) features
    assert subprocess.run([os.path.join(REF, "copy-feats"), "--compress=true", f"ark:{td}/f.ark", f"ark,scp:{td}/c.ark,{td}/c.scp"], env=ENV, capture_output=True).returncode == 0
    assert subprocess.run([os.path.join(REF, "nnet3-compute"), "--use-gpu=no", mdl, f"scp:{td}/c.scp", f"ark:{td}/refc.ark"], env=ENV, capture_output=True).returncode == 0
    g = subprocess.run([os.path.join(BIN, "nnet3-compute"), mdl, f"scp:{td}/c.scp", f"ark,t:{td}/gpuc.txt"], capture_output=True, text=True); assert g.returncode == 0, g.stderr
    a = kio.read_ark(f"{td}/refc.ark")
    txt = open(f"{td}/gpuc.txt").read().replace("[", " ").replace("]", " ").split()
    vals = [t for t in txt if not t.startswith("utt")]; got = np.array(vals, np.float32)
    ref = np.concatenate([a[k].ravel() for k in a])
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4      # text output keeps 6 significant digits

def test_nnet3_latgen_faster_end_to_end(tmp_path):
    """features archive -> lattices, words, alignments; the best path must equal the oracle chain's (oracle nnet3 + oracle decoder)"""
    from oracle import kaldi_io as kio, nnet3_oracle as no, lattice_oracle as lo
    td = str(tmp_path); N = 120; rng = np.random.default_rng(4)
    feats = {f"utt{i}": (rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32) for i, T in enumerate([120, 45])}
    kio.write_ark(f"{td}/f.ark", feats)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=feats["utt0"], out_std=1.5)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N); net.write(f"{td}/final.raw")
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    cmd = [os.path.join(BIN, "nnet3-latgen-faster"), "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=7000",
           "--determinize-lattice=false", "--allow-partial=true", f"{td}/final.mdl", f"{td}/HCLG.fst", f"ark:{td}/f.ark", f"ark,t:{td}/lat.txt", f"ark,t:{td}/words.txt", f"ark,t:{td}/ali.txt"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Done 2 utterances, failed for 0" in r.stderr and "Overall log-likelihood per frame is" in r.stderr
    words = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(f"{td}/words.txt")}
    ali = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(f"{td}/ali.txt")}
    onet = no.read_nnet(f"{td}/final.raw"); t2p = synth.tid2pdf(N)
    for k, f in feats.items():
        ll = no.compute(onet, f, 3)
        ref, _ = lo.decode(graph, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, max_active=7000), mode=0)
        bp = ref.connect().best_path()
        assert len(ali[k]) == ll.shape[0]
        assert words[k] == bp[1] and ali[k] == bp[0], k
    # default --determinize-lattice=true: CompactLattices whose best path is the same word sequence and alignment
    from tests import lattice_cases as lc
    cmd_det = [c for c in cmd if not c.startswith("--determinize")]; cmd_det[-3] = f"ark,t:{td}/clat.txt"
    r = subprocess.run(cmd_det, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    clats = lc.parse_compact_text(open(f"{td}/clat.txt").read())
    assert sorted(clats) == sorted(feats)
    for k in feats:
        assert _compact_best_words(clats[k]) == words[k], k
        assert _compact_best_words(clats[k], want="tids") == ali[k], k


@pytest.mark.parametrize("flags,spk", [([], False), (["--cmn-window=100", "--speaker-frames=100", "--global-frames=10", "--norm-vars=true", "--skip-dims=0:5"], True),
                                       (["--cmn-window=100", "--speaker-frames=60", "--global-frames=25"], True)])
def test_apply_cmvn_online_cuda_matches_the_reference_binary(tmp_path, cmvn_online_golden, flags, spk):
    """apply-cmvn-online-cuda (and its --spk2utt extension) vs the reference's online2bin/apply-cmvn-online run here on the same archive
    (oracle/_ref), and vs the committed fixtures where the option set is one of theirs."""
    from oracle import kaldi_io as kio
    g = cmvn_online_golden; td = str(tmp_path)
    kio.write_ark(f"{td}/ab.ark", {"utt_a": g["feats_a"], "utt_b": g["feats_b"]})
    with open(f"{td}/g.txt", "w") as f:
        f.write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in g["global"]) + " ]\n")
    open(f"{td}/spk2utt", "w").write("spk1 utt_a utt_b\n")
    extra = [f"--spk2utt=ark:{td}/spk2utt"] if spk else []
    r = subprocess.run([os.path.join(BIN, "apply-cmvn-online-cuda")] + flags + extra + [f"{td}/g.txt", f"ark:{td}/ab.ark", f"ark:{td}/gpu.ark"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Applied online CMVN to 2 files, or 1141 frames." in r.stderr
    got = kio.read_ark(f"{td}/gpu.ark")
    if os.path.exists(os.path.join(REF, "apply-cmvn-online")):
        q = subprocess.run([os.path.join(REF, "apply-cmvn-online")] + flags + extra + [f"{td}/g.txt", f"ark:{td}/ab.ark", f"ark:{td}/ref.ark"], env=ENV, capture_output=True, text=True)
        assert q.returncode == 0, q.stderr
        ref = kio.read_ark(f"{td}/ref.ark")
        assert list(ref) == list(got)
        for k in ref: assert np.array_equal(ref[k], got[k]), (k, np.abs(ref[k] - got[k]).max())
    name = {(): "default", ("--cmn-window=100", "--speaker-frames=100", "--global-frames=10", "--norm-vars=true", "--skip-dims=0:5"): "w100_spk_vars_skip",
            ("--cmn-window=100", "--speaker-frames=60", "--global-frames=25"): "w100_spk"}[tuple(flags)]
    for k in got: assert np.array_equal(got[k], g[f"ref_{name}_{k}"])
    # usage / error behaviour of the reference program: wrong argument count -> usage + exit 1; bad stats -> message + exit 255 (-1)
    assert subprocess.run([os.path.join(BIN, "apply-cmvn-online-cuda"), f"{td}/g.txt"], capture_output=True).returncode == 1
    open(f"{td}/bad.txt", "w").write(" [\n 1 2 3 ]\n")
    b = subprocess.run([os.path.join(BIN, "apply-cmvn-online-cuda"), f"{td}/bad.txt", f"ark:{td}/ab.ark", f"ark:{td}/o.ark"], capture_output=True, text=True)
    assert b.returncode == 255 and "stats" in b.stderr


def test_batched_wav_nnet3_cuda_online_equals_offline_program(tmp_path):
    """the streaming program (chunks of audio over a few channels, channels reused as files end) writes the same lattices as
    batched-wav-nnet3-cuda2: same keys, identical text records"""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 31000, 4100]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", "--determinize-lattice=false", "--write-compact=false"]
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=5", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/off.txt"], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    off = _parse_text_lattices(f"{td}/off.txt")
    for fpc, nch in ((150, 2), (30, 3)):
        import time; t0 = time.time()
        b = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + ["--write-lattice=true", f"--num-channels={nch}", f"--frames-per-chunk={fpc}", "--max-utterance-frames=400",
                                                                                            f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/on.txt"], capture_output=True, text=True)
        assert b.returncode == 0, b.stderr
        print("online program", fpc, nch, "%.1f s" % (time.time() - t0), b.stderr.strip().splitlines()[-1])
        assert "Decoded 5 utterances, 0 with errors." in b.stderr and "RealTimeX:" in b.stderr
        on = _parse_text_lattices(f"{td}/on.txt")
        assert sorted(on) == sorted(off) == [f"utt{i}" for i in range(5)]
        for k in off:
            # state numbers depend on the order the GPU happened to emit the arcs in: compare the arcs without them
            canon = lambda lat: (sorted((a[2], a[3], float(a[4]), float(a[5])) for a in lat[0]), sorted(float(v) for v in lat[1].values()))
            assert canon(on[k]) == canon(off[k]), (fpc, k)
    assert subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online"), "x"], capture_output=True).returncode == 1


def test_batched_wav_nnet3_cuda2_rank_sharding_and_job_naming(tmp_path):
    """--rank / --world-size: rank r decodes the utterances i with i % world == r and writes lat.<r+1> (the lat.JOB convention of
    decode.sh); the union of the ranks' archives is the single-process archive, record for record.  (The ranks run one after the other here: the
    test box has one GPU; with --nccl-id-file they would also share one RCCL broadcast of the graph, tests/test_parallel_gpu.py.)"""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 12000, 8000]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    synth.make_hclg(3000, 8000, N, seed=11, start_degree=50).write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    base = [os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0",
            "--max-batch-size=4", "--determinize-lattice=false", "--write-compact=false"]
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    r = subprocess.run(base + tail + [f"ark,t:{td}/all.txt"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    for rank in (0, 1):
        r = subprocess.run(base + [f"--rank={rank}", "--world-size=2", "--device=0"] + tail + [f"ark,t:{td}/lat.JOB.txt"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
        assert f"Decoded {3 - rank} utterances, 0 with errors." in r.stderr
    whole, r1, r2 = _parse_text_lattices(f"{td}/all.txt"), _parse_text_lattices(f"{td}/lat.1.txt"), _parse_text_lattices(f"{td}/lat.2.txt")
    assert list(r1) == ["utt0", "utt2", "utt4"] and list(r2) == ["utt1", "utt3"] and sorted(whole) == sorted(list(r1) + list(r2))
    canon = lambda lat: (sorted((a[2], a[3], float(a[4]), float(a[5])) for a in lat[0]), sorted(float(v) for v in lat[1].values()))      # state numbers depend on the batch a lattice was decoded in
    for part in (r1, r2):
        for k, lat in part.items(): assert canon(lat) == canon(whole[k]) and len(lat[0]) > 0, k


def test_ivector_extract_online2_and_nnet3_compute_with_ivectors(tmp_path):
    """the recipe's i-vector chain on the command line: ivector-extract-online2 (two utterances per speaker: the adaptation state is carried) against the
    REFERENCE binary's output (tests/golden/ivector), --repeat, then nnet3-compute --online-ivectors / --ivectors against the reference's nnet3-compute"""
    from oracle import kaldi_io as kio
    td = str(tmp_path); IV = os.path.join(ROOT, "tests", "golden", "ivector")
    g = np.load(os.path.join(IV, "ivector_golden.npz")); ref = np.load(os.path.join(IV, "ivector_adapt_golden.npz")); utts = ["utt0", "utt1", "utt2", "utt3"]
    kio.write_ark(f"{td}/feats.ark", {u: g["feat_" + u] for u in utts}); open(f"{td}/spk2utt", "w").write("spkA utt0 utt1\nspkB utt2 utt3 utt_missing\n")
    exe = os.path.join(BIN, "ivector-extract-online2")
    r = subprocess.run([exe, "--config=ivector_extractor.conf", f"ark:{td}/spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/iv.ark"], capture_output=True, text=True, cwd=IV); assert r.returncode == 0, r.stderr
    assert "Did not find audio for utterance utt_missing" in r.stderr and "Estimated iVectors for 4 files, 1 with errors." in r.stderr
    iv = kio.read_ark(f"{td}/iv.ark")
    for u in utts: assert iv[u].shape == ref["iv_" + u].shape and np.abs(iv[u] - ref["iv_" + u]).max() <= 2e-5, (u, np.abs(iv[u] - ref["iv_" + u]).max())
    open(f"{td}/spk2utt1", "w").write("".join(f"{u} {u}\n" for u in utts))
    r = subprocess.run([exe, "--config=ivector_extractor.conf", "--repeat=true", "--max-batch-size=3", f"ark:{td}/spk2utt1", f"ark:{td}/feats.ark", f"ark:{td}/ivr.ark"], capture_output=True, text=True, cwd=IV); assert r.returncode == 0, r.stderr
    ivr = kio.read_ark(f"{td}/ivr.ark")
    for u in utts: assert ivr[u].shape == g["iv_repeat_" + u].shape and np.abs(ivr[u] - g["iv_repeat_" + u]).max() <= 2e-5
    # ---- silence weighting: --frame-weights-rspecifier / --length-tolerance against the reference binary (tests/golden/make_golden_ivector_weighted.py)
    wg = np.load(os.path.join(IV, "ivector_weighted_golden.npz"))
    open(f"{td}/w.txt", "w").write("".join(u + "  [ " + " ".join(repr(float(x)) for x in wg["w_" + u]) + " ]\n" for u in utts))
    open(f"{td}/spk2utt2", "w").write("spkA utt0 utt1\nspkB utt2 utt3\n")
    for tag, extra in (("iv", []), ("ivrep", ["--repeat=true"])):
        r = subprocess.run([exe, "--config=ivector_extractor.conf", "--length-tolerance=2", f"--frame-weights-rspecifier=ark,t:{td}/w.txt"] + extra + [f"ark:{td}/spk2utt2", f"ark:{td}/feats.ark", f"ark:{td}/ivw.ark"],
                           capture_output=True, text=True, cwd=IV); assert r.returncode == 0, r.stderr
        ivw = kio.read_ark(f"{td}/ivw.ark")
        for u in utts:
            got = ivw[u] if tag == "iv" else ivw[u][::10]
            assert got.shape == wg[f"{tag}_{u}"].shape and np.abs(got - wg[f"{tag}_{u}"]).max() <= 2e-5, (tag, u, np.abs(got - wg[f"{tag}_{u}"]).max())
    r = subprocess.run([exe, "--config=ivector_extractor.conf", f"--frame-weights-rspecifier=ark,t:{td}/w.txt", f"ark:{td}/spk2utt2", f"ark:{td}/feats.ark", f"ark:{td}/ivw0.ark"], capture_output=True, text=True, cwd=IV)
    assert r.returncode == 0 and "Estimated iVectors for 3 files, 1 with errors." in r.stderr, r.stderr      # utt2's weights are 2 frames short: an error at the default tolerance 0
    ivw0 = kio.read_ark(f"{td}/ivw0.ark"); assert sorted(ivw0) == ["utt0", "utt1", "utt3"] and np.abs(ivw0["utt3"] - wg["tol0_utt3"]).max() <= 2e-5
    open(f"{td}/w1.txt", "w").write("utt0  [ " + " ".join("1" for _ in wg["w_utt0"]) + " ]\n")
    r = subprocess.run([exe, "--config=ivector_extractor.conf", f"--frame-weights-rspecifier=ark,t:{td}/w1.txt", f"ark:{td}/spk2utt1", f"ark:{td}/feats.ark", f"ark:{td}/ivw1.ark"], capture_output=True, text=True, cwd=IV)
    assert r.returncode == 0 and "Did not find weights for utterance utt1" in r.stderr and "Estimated iVectors for 1 files, 3 with errors." in r.stderr, r.stderr
    bad = subprocess.run([exe, "--config=ivector_extractor.conf", "--diag-ubm=nonexistent.dubm", f"ark:{td}/spk2utt1", f"ark:{td}/feats.ark", f"ark:{td}/x.ark"], capture_output=True, text=True, cwd=IV)
    assert bad.returncode != 0 and "nonexistent.dubm" in bad.stderr
    # ---- the network side
    GOLD = os.path.join(ROOT, "tests", "golden"); n = np.load(os.path.join(GOLD, "nnet_ivector_io.npz")); mdl = os.path.join(GOLD, "nnet_ivector.raw")
    kio.write_ark(f"{td}/f.ark", {"u": n["feats"]}); nc = os.path.join(BIN, "nnet3-compute")
    for name, (s, chunk, period) in {"s3_c21_p7": (3, 21, 7), "s1_c20_p10_short": (1, 20, 10)}.items():
        kio.write_ark(f"{td}/oiv.ark", {"u": n["iv_" + name]})
        r = subprocess.run([nc, f"--frame-subsampling-factor={s}", f"--frames-per-chunk={chunk}", f"--online-ivectors=ark:{td}/oiv.ark", f"--online-ivector-period={period}", mdl, f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = kio.read_ark(f"{td}/o.ark")["u"]; assert got.shape == n["ref_" + name].shape and np.abs(got - n["ref_" + name]).max() <= 1e-4
    open(f"{td}/iv.txt", "w").write("spk  [ " + " ".join(repr(float(x)) for x in n["iv_s3_utt"]) + " ]\n"); open(f"{td}/utt2spk", "w").write("u spk\n")
    r = subprocess.run([nc, "--frame-subsampling-factor=3", f"--ivectors=ark,t:{td}/iv.txt", f"--utt2spk=ark:{td}/utt2spk", mdl, f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    got = kio.read_ark(f"{td}/o.ark")["u"]; assert np.abs(got - n["ref_s3_utt"]).max() <= 1e-4
    r = subprocess.run([nc, mdl, f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True)            # the model needs i-vectors: the reference's message
    assert r.returncode != 0 and "Neural net expects 'ivector' features with dimension 12 but you provided 0" in r.stderr
    kio.write_ark(f"{td}/oiv.ark", {"other": n["iv_s3_c21_p7"]})
    r = subprocess.run([nc, f"--online-ivectors=ark:{td}/oiv.ark", "--online-ivector-period=7", mdl, f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True)
    assert r.returncode != 0 and "No iVector available for utterance u" in r.stderr


def test_batched_wav_nnet3_cuda2_with_ivector_extraction(tmp_path):
    """wav -> fbank -> i-vectors (GPU, one per 10 frames) -> TDNN-F with the i-vector input, chunk by chunk -> lattices, in ONE program, against the chain of
    separate programs the recipes run: compute-fbank-feats | ivector-extract-online2 | nnet3-latgen-faster --online-ivectors (same best path, same lattice)"""
    from oracle import kaldi_io as kio
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 4000]; IV = os.path.join(ROOT, "tests", "golden", "ivector")
    _wavs(td, lens); open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    r = subprocess.run([os.path.join(BIN, "compute-fbank-feats-cuda"), f"--config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    feats = kio.read_ark(f"{td}/f.ark"); allf = np.concatenate(list(feats.values())).astype(np.float64); rng = np.random.default_rng(8)
    # the fixture's UBM and extractor live in the 20-dim LDA space; only the LDA matrix and the CMVN statistics see the 40-dim features
    tm = lambda path, m: open(path, "w").write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m) + " ]\n")
    st = np.zeros((2, 41)); st[0, :40] = allf.sum(0); st[1, :40] = (allf ** 2).sum(0); st[0, 40] = allf.shape[0]; tm(f"{td}/global_cmvn.stats", st)
    tm(f"{td}/final.mat", (rng.standard_normal((20, 7 * 40)) * 1.5 / np.sqrt(7 * 40)).astype(np.float32))
    open(f"{td}/splice.conf", "w").write("--left-context=3\n--right-context=3\n"); open(f"{td}/cmvn.conf", "w").write("\n")
    open(f"{td}/ivector.conf", "w").write(f"--lda-matrix={td}/final.mat\n--global-cmvn-stats={td}/global_cmvn.stats\n--cmvn-config={td}/cmvn.conf\n--splice-config={td}/splice.conf\n--diag-ubm={IV}/final.dubm\n"
                                          f"--ivector-extractor={IV}/final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n--max-count=100\n--ivector-period=10\n")
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=feats["utt0"], out_std=1.5, ivector_dim=16)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    common = ["--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", "--frames-per-chunk=51", "--determinize-lattice=false"]
    r = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", f"--ivector-extraction-config={td}/ivector.conf", "--max-batch-size=3", "--write-compact=false"] + common +
                       [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/lat.txt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Decoded 4 utterances, 0 with errors." in r.stderr
    # the chain of separate programs
    open(f"{td}/spk2utt", "w").write("".join(f"{k} {k}\n" for k in feats))
    r = subprocess.run([os.path.join(BIN, "ivector-extract-online2"), f"--config={td}/ivector.conf", f"ark:{td}/spk2utt", f"ark:{td}/f.ark", f"ark:{td}/iv.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    iv = kio.read_ark(f"{td}/iv.ark"); assert all(iv[k].shape == ((feats[k].shape[0] + 9) // 10, 16) for k in feats) and max(np.abs(v).max() for v in iv.values()) > 0.05
    r = subprocess.run([os.path.join(BIN, "nnet3-latgen-faster")] + common + [f"--online-ivectors=ark:{td}/iv.ark", "--online-ivector-period=10", f"{td}/final.mdl", f"{td}/HCLG.fst", f"ark:{td}/f.ark", f"ark,t:{td}/lat2.txt",
                        f"ark,t:{td}/words2.txt", f"ark,t:{td}/ali2.txt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    a, b = _parse_text_lattices(f"{td}/lat.txt"), _parse_text_lattices(f"{td}/lat2.txt"); assert list(a) == list(b) == list(feats)
    for k in a:      # the same raw lattices (state numbers depend on the order the GPU appended the arcs in: compare the arcs as (labels, costs) multisets)
        assert sorted(x[2:] for x in a[k][0]) == sorted(x[2:] for x in b[k][0]) and len(a[k][0]) > 10 and sorted(a[k][1].values()) == sorted(b[k][1].values()), k
    r = subprocess.run([os.path.join(BIN, "lattice-best-path"), "--acoustic-scale=1.0", f"ark,t:{td}/lat.txt", f"ark,t:{td}/words.txt", f"ark,t:{td}/ali.txt"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    assert open(f"{td}/words.txt").read() == open(f"{td}/words2.txt").read() and open(f"{td}/ali.txt").read() == open(f"{td}/ali2.txt").read()
    # without the extractor the model cannot run: the reference's message
    r = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf"] + common + [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark:{td}/x.ark"], capture_output=True, text=True)
    assert r.returncode != 0 and "Neural net expects 'ivector' features with dimension 16 but you provided 0" in r.stderr


def test_pipeline_class_with_callbacks_and_task_groups_equals_the_program(tmp_path):
    """kaldi_amd/host/k3_pipeline.h: BatchedThreadedNnet3CudaPipeline2's class surface (DecodeWithCallback, CreateTaskGroup / WaitForGroup /
    DestroyTaskGroup, WaitForAllTasks; cudadecoder/batched-threaded-nnet3-cuda-pipeline2.h:57-239) driven by a caller written like the
    reference's program: the determinized lattices its callbacks receive equal the ones batched-wav-nnet3-cuda2 writes, text record for record"""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 31000, 4100, 12000, 300]      # the last one is too short to decode
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000"]
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=4", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/prog.txt"], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    b = subprocess.run([os.path.join(BIN, "k3-pipeline-example")] + common + ["--max-batch-size=3", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/cls.txt"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    assert "Decoded 7 utterances, 1 with errors." in b.stderr, b.stderr
    assert open(f"{td}/prog.txt").read() == open(f"{td}/cls.txt").read() and open(f"{td}/cls.txt").read().count("utt") == 6


def test_compute_online_feats_batched_cuda_equals_the_offline_programs(tmp_path):
    """cudafeatbin/compute-online-feats-batched-cuda.cc's contract: audio fed in chunks of --chunk-length samples over a few channels; the feature table
    it writes must be bit-identical to compute-fbank-feats-cuda's (chunked == whole utterance), the i-vector table = the last i-vector
    ivector-extract-online2 estimates for each utterance (text table parsed here: the vectors are written by BaseFloatVectorWriter)"""
    from oracle import kaldi_io as kio
    td = str(tmp_path); lens = [16000, 9000, 23001, 4000, 31007, 1234]; IV = os.path.join(ROOT, "tests", "golden", "ivector")
    _wavs(td, lens); open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    r = subprocess.run([os.path.join(BIN, "compute-fbank-feats-cuda"), f"--config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    feats = kio.read_ark(f"{td}/f.ark"); allf = np.concatenate(list(feats.values())).astype(np.float64); rng = np.random.default_rng(8)
    tm = lambda path, m: open(path, "w").write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m) + " ]\n")
    st = np.zeros((2, 41)); st[0, :40] = allf.sum(0); st[1, :40] = (allf ** 2).sum(0); st[0, 40] = allf.shape[0]; tm(f"{td}/global_cmvn.stats", st)
    tm(f"{td}/final.mat", (rng.standard_normal((20, 7 * 40)) * 1.5 / np.sqrt(7 * 40)).astype(np.float32))
    open(f"{td}/splice.conf", "w").write("--left-context=3\n--right-context=3\n"); open(f"{td}/cmvn.conf", "w").write("\n")
    open(f"{td}/ivector.conf", "w").write(f"--lda-matrix={td}/final.mat\n--global-cmvn-stats={td}/global_cmvn.stats\n--cmvn-config={td}/cmvn.conf\n--splice-config={td}/splice.conf\n--diag-ubm={IV}/final.dubm\n"
                                          f"--ivector-extractor={IV}/final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n--max-count=100\n--ivector-period=10\n")
    exe = os.path.join(BIN, "compute-online-feats-batched-cuda")
    for chunk, lanes, nch in ((4000, 2, 3), (777, 4, 4)):
        r = subprocess.run([exe, "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", f"--ivector-extraction-config={td}/ivector.conf", f"--chunk-length={chunk}", f"--batch-size={lanes}", f"--num-channels={nch}",
                            f"scp:{td}/wav.scp", f"ark,t:{td}/iv.txt", f"ark:{td}/of.ark"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Computed online features for 6 files" in r.stderr
        of = kio.read_ark(f"{td}/of.ark")
        assert sorted(of) == sorted(feats)
        for k in feats: assert of[k].shape == feats[k].shape and np.array_equal(of[k], feats[k]), (chunk, k)
        open(f"{td}/spk2utt", "w").write("".join(f"{k} {k}\n" for k in feats))
        r2 = subprocess.run([os.path.join(BIN, "ivector-extract-online2"), f"--config={td}/ivector.conf", f"ark:{td}/spk2utt", f"ark:{td}/f.ark", f"ark:{td}/iv.ark"], capture_output=True, text=True); assert r2.returncode == 0, r2.stderr
        ivm = kio.read_ark(f"{td}/iv.ark"); got = {}
        for line in open(f"{td}/iv.txt"):
            key, rest = line.split(None, 1); got[key] = np.array([float(x) for x in rest.replace("[", "").replace("]", "").split()], np.float32)
        for k in feats: assert got[k].shape == (ivm[k].shape[1],) and np.abs(got[k] - ivm[k][-1]).max() <= 1e-5, (chunk, k)
    # no i-vector extractor: empty vectors, like the reference's IvectorDim() == 0
    r = subprocess.run([exe, "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark,t:{td}/iv0.txt", f"ark:{td}/of0.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    assert all(line.split(None, 1)[1].strip() == "[ ]" for line in open(f"{td}/iv0.txt"))
    assert subprocess.run([exe, "x"], capture_output=True).returncode == 1


def test_feature_program_channel_option_and_extensible_wav_header(tmp_path):
    """--channel like featbin/compute-fbank-feats.cc:140-160 (a stereo file: channel 1 = the right channel; -1 warns and takes the left; an absent channel skips the
    file) and WAVE_FORMAT_EXTENSIBLE headers with a PCM sub-format (feat/wave-reader.cc:176-205)"""
    import struct
    from oracle import kaldi_io as kio
    td = str(tmp_path); rng = np.random.default_rng(3); L, R = (rng.standard_normal(8000) * 3000).astype(np.int16), (rng.standard_normal(8000) * 3000).astype(np.int16)
    def wav(path, chans, extensible=False):
        data = np.stack(chans, 1).astype("<i2").tobytes(); nch = len(chans)
        fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else 1, nch, 16000, 16000 * 2 * nch, 2 * nch, 16)
        if extensible: fmt += struct.pack("<HHI", 22, 16, 3 if nch == 2 else 4) + struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        open(path, "wb").write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data)
    wav(f"{td}/st.wav", [L, R]); wav(f"{td}/l.wav", [L]); wav(f"{td}/r.wav", [R]); wav(f"{td}/ext.wav", [L, R], extensible=True)
    open(f"{td}/wav.scp", "w").write(f"st {td}/st.wav\nl {td}/l.wav\nr {td}/r.wav\next {td}/ext.wav\n")
    exe = os.path.join(BIN, "compute-fbank-feats-cuda"); run = lambda *a: subprocess.run([exe, "--dither=0", "--num-mel-bins=40", *a, f"scp:{td}/wav.scp", f"ark:{td}/o.ark"], capture_output=True, text=True)
    r = run("--channel=1"); assert r.returncode == 0, r.stderr
    f = kio.read_ark(f"{td}/o.ark"); assert sorted(f) == ["ext", "st"] and "has 1 channels but you specified channel 1" in r.stderr
    r0 = run("--channel=0"); f0 = kio.read_ark(f"{td}/o.ark"); assert sorted(f0) == ["ext", "l", "r", "st"]
    assert np.array_equal(f["st"], f0["r"]) and np.array_equal(f["ext"], f0["r"]) and np.array_equal(f0["st"], f0["l"]) and np.array_equal(f0["ext"], f0["l"])
    rm = run(); fm = kio.read_ark(f"{td}/o.ark"); assert "Channel not specified but you have data with 2 channels" in rm.stderr and np.array_equal(fm["st"], f0["l"])


def test_online_pipeline_class_with_correlation_ids_and_callbacks_equals_the_program(tmp_path):
    """kaldi_amd/host/k3_online_pipeline.h: BatchedThreadedNnet3CudaOnlinePipeline's class surface (TryInitCorrID, DecodeBatch per chunk, SetLatticeCallback,
    partial hypotheses / end-points, WaitForLatticeCallbacks; cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.h:119-330) driven by a caller written like the
    reference's streaming program: more streams than channels, chunk by chunk; the determinized lattices its callbacks receive equal the offline program's output"""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 31000, 4100, 12000, 20011]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000"]
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=4", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/prog.txt"], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    for extra in (["--max-batch-size=3", "--num-channels=3", "--frames-per-chunk=51"], ["--max-batch-size=2", "--num-channels=4", "--frames-per-chunk=150", "--print-partial-hypotheses=true"]):
        b = subprocess.run([os.path.join(BIN, "k3-online-pipeline-example")] + common + extra + ["--max-utterance-frames=400", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/cls.txt"], capture_output=True, text=True)
        assert b.returncode == 0, b.stderr
        assert "Decoded 7 utterances, 0 with errors." in b.stderr, b.stderr[-2000:]
        assert open(f"{td}/prog.txt").read() == open(f"{td}/cls.txt").read()
        if "--print-partial-hypotheses=true" in extra:
            assert " partial: " in b.stderr and " final: " in b.stderr
            n_partial = int(b.stderr.split("Non-empty partial hypotheses: ")[1].split(",")[0]); assert n_partial > 0


def test_segmentation_program_and_class_equal_decoding_the_pieces(tmp_path):
    """--segmentation (BatchedThreadedNnet3CudaPipeline2::SegmentedDecodeWithCallback, batched-threaded-nnet3-cuda-pipeline2.cc:265-337 + WriteLattices, cuda-pipeline-common.cc:38-62):
    files cut into --segment-length pieces that overlap by --segment-overlap, a last piece below --min-segment-length dropped, keys [utt]-[offset in whole seconds].  The program and the class
    (k3-pipeline-example --segmentation) must write, under those keys, exactly the lattices of the pieces decoded as files of their own."""
    from oracle import kaldi_io as kio
    td = str(tmp_path); N = 120; sr = 16000
    # 7.6 s -> pieces at 0, 2.5, 5.0 (3 s long; the last 2.6 s); 3.0 s -> one piece; 5.6 s -> 0, 2.5 and a 0.6 s tail that is dropped (min 1 s); 0.4 s: one (short) piece, like the reference (< one segment = 1 segment, min applies)
    lens = [int(7.6 * sr), int(3.0 * sr), int(5.6 * sr), int(1.2 * sr)]
    pcm = [synth.gaussian_pcm16(n, 70 + i) for i, n in enumerate(lens)]
    for i, p in enumerate(pcm): kio.write_wav(f"{td}/f{i}.wav", p)
    open(f"{td}/wav.scp", "w").write("".join(f"file{i} {td}/f{i}.wav\n" for i in range(len(lens))))
    seg_len, shift, seg_min = int(3.0 * sr), int(2.5 * sr), int(1.0 * sr); want = []
    for i, p in enumerate(pcm):
        off = 0
        while True:
            n = min(len(p) - off, seg_len)
            if n >= seg_min:
                key = f"file{i}-{int(np.floor(np.float32(off) / np.float32(sr)))}"; kio.write_wav(f"{td}/{key}.wav", p[off:off + n]); want.append(key)
            if off + n >= len(p): break
            off += shift
    assert want == ["file0-0", "file0-2", "file0-5", "file1-0", "file2-0", "file2-2", "file3-0"], want
    open(f"{td}/pieces.scp", "w").write("".join(f"{k} {td}/{k}.wav\n" for k in want))
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    synth.make_hclg(3000, 8000, N, seed=11, start_degree=50).write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000"]
    seg = ["--segmentation=true", "--segment-length=3.0", "--segment-overlap=0.5", "--min-segment-length=1.0"]
    ref = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=4", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/pieces.scp", f"ark,t:{td}/pieces.txt"], capture_output=True, text=True)
    assert ref.returncode == 0, ref.stderr
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + seg + ["--max-batch-size=3", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/prog.txt"], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    assert open(f"{td}/prog.txt").read() == open(f"{td}/pieces.txt").read() and open(f"{td}/prog.txt").read().count("file") == len(want)
    b = subprocess.run([os.path.join(BIN, "k3-pipeline-example")] + common + seg + ["--max-batch-size=3", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/cls.txt"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    assert f"Decoded 4 files in {len(want)} segments." in b.stderr, b.stderr
    def records(path):      # the class hands results over per file in completion order: compare as a key -> record map
        recs, cur = {}, []
        for line in open(path).read().split("\n"):
            if line.startswith("file"): key = line.split()[0]; cur = recs.setdefault(key, [line])
            elif line.strip(): cur.append(line)
        return recs
    ra, rb = records(f"{td}/prog.txt"), records(f"{td}/cls.txt")
    assert sorted(ra) == sorted(want) and ra == rb
    e = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--segmentation=true", "--segment-length=3.0", "--segment-overlap=3.5", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/x.txt"], capture_output=True, text=True)
    assert e.returncode != 0 and "overlap" in e.stderr
    # the class with a lattice post-processor and RESULT_TYPE_CTM (SetLatticePostprocessor, SetResultUsingLattice, MergeSegmentsToCTMOutput): one merged CTM per file -- of two overlapping
    # segments the earlier one keeps the words that begin before the later one starts; times carry the segments' offsets; the lattices are the post-processed ones
    open(f"{td}/pp.conf", "w").write("--acoustic-scale=0.8\n")
    c = subprocess.run([os.path.join(BIN, "k3-pipeline-example")] + common + seg + ["--max-batch-size=3", f"--lattice-postprocessor-rxfilename={td}/pp.conf", f"--ctm-out={td}/cls.ctm", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/cls2.txt"],
                       capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    ctm = [l.split() for l in open(f"{td}/cls.ctm")]
    assert ctm and all(len(l) == 6 for l in ctm) and {l[0] for l in ctm} == {"file0", "file1", "file2", "file3"}
    for f, dur in zip(("file0", "file1", "file2", "file3"), (7.6, 3.0, 5.6, 1.2)):
        mine = [l for l in ctm if l[0] == f]; starts = [float(l[2]) for l in mine]; segs = [int(l[1]) for l in mine]
        assert starts == sorted(starts) and segs == sorted(segs) and float(mine[-1][2]) + float(mine[-1][3]) <= dur + 0.05 and all(0.0 <= float(l[5]) <= 1.0 for l in mine), (f, mine)
    assert max(int(l[1]) for l in ctm if l[0] == "file0") == 2 and max(float(l[2]) for l in ctm if l[0] == "file0") >= 5.0      # the third segment of file0 contributes words beyond its 5 s offset
    d = subprocess.run([os.path.join(BIN, "k3-pipeline-example")] + common + seg + ["--max-batch-size=3", f"--ctm-out={td}/no.ctm", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/cls3.txt"], capture_output=True, text=True)
    assert d.returncode != 0 and "SetLatticePostprocessor" in d.stderr


def test_batched_wav_nnet3_cuda2_ctm_output(tmp_path):
    """<lattice-wspecifier|ctm-wxfilename> (cudadecoderbin/batched-wav-nnet3-cuda2.cc:67-71,135-224): an output argument that is not a table wspecifier names a CTM file; the lines are
    LatticePostprocessor::GetCTM (scales of --lattice-postprocessor-rxfilename, MBR) of every utterance's determinized lattice, in MergeSegmentsToCTMOutput's layout.  Checked against
    lattice-mbr-decode (pinned to the reference's lat/sausages.cc in tests/test_lattice_det.py) run on the lattices the same program writes, with the same scales."""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    synth.make_hclg(3000, 8000, N, seed=11, start_degree=50).write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n"); open(f"{td}/pp.conf", "w").write("--acoustic-scale=0.7\n--lm-scale=1.5\n")
    base = [os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=6.0",
            "--max-active=10000", "--max-batch-size=2", f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    r = subprocess.run(base + [f"{td}/out.ctm"], capture_output=True, text=True)
    assert r.returncode != 0 and "You must configure the lattice postprocessor" in r.stderr
    r = subprocess.run(base[:1] + [f"--lattice-postprocessor-rxfilename={td}/pp.conf"] + base[1:] + [f"{td}/out.ctm"], capture_output=True, text=True); assert r.returncode == 0, r.stderr[-2000:]
    ctm = [l.split() for l in open(f"{td}/out.ctm")]
    assert ctm and all(len(l) == 6 and l[1] == "0" for l in ctm) and [l[0] for l in ctm] == sorted(l[0] for l in ctm)
    # the lattices of the same run, then MBR with the post-processor's scales
    r = subprocess.run(base + [f"ark,t:{td}/det.txt"], capture_output=True, text=True); assert r.returncode == 0, r.stderr[-2000:]
    g = subprocess.run([os.path.join(BIN, "lattice-mbr-decode"), "--acoustic-scale=0.7", "--lm-scale=1.5", f"ark,t:{td}/det.txt", f"ark,t:{td}/w.txt", "", f"ark,t:{td}/s.txt"], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr[-2000:]
    words = {l.split()[0]: l.split()[1:] for l in open(f"{td}/w.txt")}
    conf = {}
    for l in open(f"{td}/s.txt"):      # sausage bins "[ word post ... ]": the posterior of the chosen (first) entry of every bin whose best entry is a word
        key = l.split()[0]; bins = [b.split() for b in l[len(key):].replace("]", "").split("[")[1:]]
        conf[key] = [float(b[1]) for b in bins if b[0] != "0"]
    for k, key in enumerate(("utt0", "utt1", "utt2")):
        mine = [l for l in ctm if l[0] == key]
        assert [l[4] for l in mine] == words[key] and len(mine) > 0, key
        # (the word TIMES of an MBR decode depend on where the arcs of the lattice happen to end -- the reference says as much: "times will only be very meaningful if you first use
        # lattice-word-align" -- and the table round trip re-packs the arcs; the confidences and the words do not)
        assert np.allclose([float(l[5]) for l in mine], conf[key], atol=0.011), (key, mine, conf[key])
        starts = [float(l[2]) for l in mine]; ends = [float(l[2]) + float(l[3]) for l in mine]
        assert starts == sorted(starts) and all(e >= b for b, e in zip(starts, ends)) and ends[-1] <= lens[k] / 16000.0 + 0.05, (key, mine)


def test_online_program_with_an_ivector_model_equals_the_python_pipeline(tmp_path):
    """batched-wav-nnet3-cuda-online --ivector-extraction-config: every chunk the network evaluates gets the extractor's latest i-vector of its stream (the rule of
    nnet3/decodable-online-looped.cc:182-197: the estimate at the last multiple of --ivector-period among the frames seen, minus the splice's right context while the stream goes on).
    The C++ drivers (kaldi_amd/host/k3_online.h: OnlineIvectors, StaticNnet3 with an i-vector per slot) and the Python ones (kaldi_amd/online.py, whose choice of i-vector rows
    tests/test_online_gpu.py holds to the whole-utterance extraction) are fed the same chunks and must write the same lattices."""
    import torch
    from oracle import kaldi_io as kio
    from kaldi_amd import feat, nnet3, decoder, online, fst as kfst
    from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001]; IV = os.path.join(ROOT, "tests", "golden", "ivector")
    _wavs(td, lens); open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    r = subprocess.run([os.path.join(BIN, "compute-fbank-feats-cuda"), f"--config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    feats = kio.read_ark(f"{td}/f.ark"); allf = np.concatenate(list(feats.values())).astype(np.float64); rng = np.random.default_rng(8)
    tm = lambda path, m: open(path, "w").write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m) + " ]\n")
    st = np.zeros((2, 41)); st[0, :40] = allf.sum(0); st[1, :40] = (allf ** 2).sum(0); st[0, 40] = allf.shape[0]; tm(f"{td}/global_cmvn.stats", st)
    tm(f"{td}/final.mat", (rng.standard_normal((20, 7 * 40)) * 1.5 / np.sqrt(7 * 40)).astype(np.float32))
    open(f"{td}/splice.conf", "w").write("--left-context=3\n--right-context=3\n"); open(f"{td}/cmvn.conf", "w").write("\n")
    open(f"{td}/ivector.conf", "w").write(f"--lda-matrix={td}/final.mat\n--global-cmvn-stats={td}/global_cmvn.stats\n--cmvn-config={td}/cmvn.conf\n--splice-config={td}/splice.conf\n--diag-ubm={IV}/final.dubm\n"
                                          f"--ivector-extractor={IV}/final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n--max-count=100\n--ivector-period=10\n")
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=feats["utt0"], out_std=1.5, ivector_dim=16)
    net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N); net.write(f"{td}/final.raw")
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
    C = 60
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", "--determinize-lattice=false", "--write-compact=false",
              "--num-channels=3", f"--frames-per-chunk={C}", "--max-utterance-frames=400", "--write-lattice=true"]
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    b = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + [f"--ivector-extraction-config={td}/ivector.conf"] + tail + [f"ark,t:{td}/on.txt"], capture_output=True, text=True)
    assert b.returncode == 0 and "Decoded 3 utterances, 0 with errors." in b.stderr, b.stderr[-2000:]
    on = _parse_text_lattices(f"{td}/on.txt")
    # the same chunks through the Python pipeline (the program hands every busy channel frames_per_chunk * 160 samples per round)
    dev = torch.device("cuda:0"); nn = nnet3.Nnet(f"{td}/final.raw"); cf = decoder.CudaFst(graph, synth.tid2pdf(N))
    ex = BatchedIvectorExtractor(OnlineIvectorExtractionInfo(f"{td}/ivector.conf"))
    cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, literal_order=1)
    pipe = online.BatchedOnlinePipeline(feat.fbank_options(dither=0.0, num_bins=40), nn, cf, cfg, 3, 400, frames_per_chunk=C, ivector_extractor=ex)
    waves = [torch.from_numpy(synth.gaussian_pcm16(n, 50 + i).astype(np.float32)).to(dev) for i, n in enumerate(lens)]; pos = [0, 0, 0]; lats = {}
    while any(p < n for p, n in zip(pos, lens)):
        chs = [u for u in range(3) if pos[u] < lens[u]]; chunks = [waves[u][pos[u]:pos[u] + C * 160] for u in chs]; first = [pos[u] == 0 for u in chs]
        for u in chs: pos[u] = min(lens[u], pos[u] + C * 160)
        for ch, lat in pipe.DecodeBatch(chs, chunks, first, [pos[u] >= lens[u] for u in chs]).items(): lats[f"utt{ch}"] = lat
    assert sorted(lats) == sorted(on) == ["utt0", "utt1", "utt2"]
    for k in on:
        want = sorted((int(i), int(o), float(np.float32(g)), float(np.float32(a))) for i, o, g, a in zip(lats[k].arc_ilabel, lats[k].arc_olabel, lats[k].arc_graph, lats[k].arc_ac))
        got = sorted((a[2], a[3], float(a[4]), float(a[5])) for a in on[k][0])
        assert len(got) > 20 and len(got) == len(want), k      # (the text form carries the costs with the stream's default precision: labels equal, costs to that precision)
        for x, y in zip(got, want): assert x[:2] == y[:2] and abs(x[2] - y[2]) <= 2e-5 * max(1.0, abs(y[2])) and abs(x[3] - y[3]) <= 2e-5 * max(1.0, abs(y[3])), (k, x, y)
    # without the extractor: the reference's message
    r = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + tail + [f"ark:{td}/x.ark"], capture_output=True, text=True)
    assert r.returncode != 0 and "Neural net expects 'ivector' features with dimension 16 but you provided 0" in r.stderr


# ---- round 4: the rest of the cudafeatbin family and the first-generation decoder driver
@pytest.mark.parametrize("kind,flags", [("fbank", ["--dither=0", "--num-mel-bins=40"]), ("mfcc", ["--dither=0", "--num-mel-bins=40", "--num-ceps=40", "--low-freq=20", "--high-freq=-400"])])
def test_online_batched_feature_programs_match_the_whole_utterance_programs_and_the_reference(tmp_path, kind, flags):
    """compute-{fbank,mfcc}-online-batched-cuda (cudafeatbin/compute-fbank-online-batched-cuda.cc:64): audio in chunks of samples over a few channels -> the rows of the
    whole-utterance program bit for bit, and the reference's CPU binary within its own rounding error"""
    from oracle import kaldi_io as kio
    td = str(tmp_path); lens = [16000, 4001, 23001, 399 + 160 * 3, 30011, 12345, 8000]; _wavs(td, lens)
    w = subprocess.run([os.path.join(BIN, f"compute-{kind}-feats-cuda")] + flags + [f"scp:{td}/wav.scp", f"ark:{td}/whole.ark"], capture_output=True, text=True); assert w.returncode == 0, w.stderr
    whole = kio.read_ark(f"{td}/whole.ark")
    for chunk, lanes, ch in ((10000, 3, 4), (777, 2, 2), (100000, 7, 7)):
        g = subprocess.run([os.path.join(BIN, f"compute-{kind}-online-batched-cuda")] + flags + [f"--chunk-length={chunk}", f"--batch-size={lanes}", f"--num-channels={ch}", f"scp:{td}/wav.scp", f"ark:{td}/on.ark"], capture_output=True, text=True)
        assert g.returncode == 0, g.stderr
        assert f"Computed Online Features for  {len(lens)} files" in g.stderr and "RTFX:" in g.stderr
        on = kio.read_ark(f"{td}/on.ark")
        assert list(on) == list(whole)
        for k in whole: assert on[k].shape == whole[k].shape and np.array_equal(on[k], whole[k]), (chunk, k)
    if os.path.exists(os.path.join(REF, f"compute-{kind}-feats")):
        r = subprocess.run([os.path.join(REF, f"compute-{kind}-feats")] + flags + [f"scp:{td}/wav.scp", f"ark:{td}/ref.ark"], env=ENV, capture_output=True, text=True); assert r.returncode == 0, r.stderr
        ref = kio.read_ark(f"{td}/ref.ark")
        for k in ref: assert np.abs(ref[k] - whole[k]).max() <= (1e-4 if kind == "fbank" else 3e-4), k
    exe = os.path.join(BIN, f"compute-{kind}-online-batched-cuda")
    assert subprocess.run([exe, f"scp:{td}/wav.scp"], capture_output=True).returncode == 1                                                        # usage
    b = subprocess.run([exe] + flags + ["--num-channels=2", "--batch-size=3", f"scp:{td}/wav.scp", f"ark:{td}/x.ark"], capture_output=True, text=True); assert b.returncode == 255 and "num-channels" in b.stderr
    b = subprocess.run([exe] + flags + ["--sample-frequency=8000", f"scp:{td}/wav.scp", f"ark:{td}/x.ark"], capture_output=True, text=True); assert b.returncode == 255 and "mismatched sampling rate" in b.stderr


def test_compute_online_feats_cuda_whole_utterance_driver(tmp_path):
    """compute-online-feats-cuda (cudafeatbin/compute-online-feats-cuda.cc:30): the whole-utterance form of the online feature pipeline; features = compute-fbank-feats-cuda's,
    an empty i-vector per utterance without an extractor, a broken file is a counted failure"""
    from oracle import kaldi_io as kio
    td = str(tmp_path); lens = [16000, 4001, 23001]; _wavs(td, lens); open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    open(f"{td}/wav.scp", "a").write(f"broken {td}/nonexistent.wav\n")
    w = subprocess.run([os.path.join(BIN, "compute-fbank-feats-cuda"), f"--config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/whole.ark"], capture_output=True, text=True); assert w.returncode == 0, w.stderr
    g = subprocess.run([os.path.join(BIN, "compute-online-feats-cuda"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/iv.ark", f"ark:{td}/f.ark"], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr
    assert "Processing Utterance utt0" in g.stderr and "Failed to compute features for utterance broken" in g.stderr and "Processed 4 utterances with 1 failures." in g.stderr
    whole, got, iv = kio.read_ark(f"{td}/whole.ark"), kio.read_ark(f"{td}/f.ark"), kio.read_ark(f"{td}/iv.ark")
    assert list(got) == list(whole) == ["utt0", "utt1", "utt2"] and list(iv) == list(got)
    for k in whole: assert np.array_equal(got[k], whole[k]) and iv[k].size == 0
    assert subprocess.run([os.path.join(BIN, "compute-online-feats-cuda"), f"scp:{td}/wav.scp"], capture_output=True).returncode == 1


@pytest.mark.parametrize("flags", [[], ["--cmn-window=100", "--speaker-frames=100", "--global-frames=10", "--norm-vars=true", "--skip-dims=0:5"]])
def test_apply_batched_cmvn_online_cuda_matches_the_reference_binary(tmp_path, cmvn_online_golden, flags):
    """apply-batched-cmvn-online-cuda (cudafeatbin/apply-batched-cmvn-online-cuda.cc:49): online CMVN in chunks of frames over a few channels = the rows of the
    reference's online2bin/apply-cmvn-online on the whole utterances, bit for bit, for any chunking (window statistics and raw-frame history carried per channel)"""
    from oracle import kaldi_io as kio
    g = cmvn_online_golden; td = str(tmp_path); rng = np.random.default_rng(11)
    feats = {"utt_a": g["feats_a"], "utt_b": g["feats_b"], "utt_c": (rng.normal(0, 3, (1303, g["feats_a"].shape[1])) + 1.0).astype(np.float32), "utt_d": g["feats_a"][:1]}
    kio.write_ark(f"{td}/in.ark", feats)
    with open(f"{td}/g.txt", "w") as f: f.write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in g["global"]) + " ]\n")
    a = subprocess.run([os.path.join(BIN, "apply-cmvn-online-cuda")] + flags + [f"{td}/g.txt", f"ark:{td}/in.ark", f"ark:{td}/whole.ark"], capture_output=True, text=True); assert a.returncode == 0, a.stderr
    whole = kio.read_ark(f"{td}/whole.ark")
    if os.path.exists(os.path.join(REF, "apply-cmvn-online")):
        q = subprocess.run([os.path.join(REF, "apply-cmvn-online")] + flags + [f"{td}/g.txt", f"ark:{td}/in.ark", f"ark:{td}/ref.ark"], env=ENV, capture_output=True, text=True); assert q.returncode == 0, q.stderr
        ref = kio.read_ark(f"{td}/ref.ark")
        for k in ref: assert np.array_equal(ref[k], whole[k]), k
    for chunk, lanes, ch in ((10000, 100, 200), (64, 2, 3), (150, 3, 3), (1, 5, 5)):
        if chunk == 1: feats_run = {k: v[:40] for k, v in feats.items()}; kio.write_ark(f"{td}/in1.ark", feats_run); src = f"ark:{td}/in1.ark"      # (one frame per call: short streams)
        else: feats_run = feats; src = f"ark:{td}/in.ark"
        b = subprocess.run([os.path.join(BIN, "apply-batched-cmvn-online-cuda")] + flags + [f"--chunk-length={chunk}", f"--batch-size={lanes}", f"--num-channels={ch}", "--stats-coarsening-factor=1", f"{td}/g.txt", src, f"ark:{td}/b.ark"],
                           capture_output=True, text=True)
        assert b.returncode == 0, b.stderr
        assert f"Applied online CMVN to {len(feats)} files, or {sum(v.shape[0] for v in feats_run.values())} frames." in b.stderr
        got = kio.read_ark(f"{td}/b.ark"); assert list(got) == list(feats)
        for k in got: assert got[k].shape == feats_run[k].shape and np.array_equal(got[k], whole[k][:feats_run[k].shape[0]]), (chunk, k)
    exe = os.path.join(BIN, "apply-batched-cmvn-online-cuda")
    assert subprocess.run([exe, f"{td}/g.txt"], capture_output=True).returncode == 1
    b = subprocess.run([exe, "--num-channels=2", "--batch-size=3", f"{td}/g.txt", f"ark:{td}/in.ark", f"ark:{td}/x.ark"], capture_output=True, text=True); assert b.returncode == 255 and "num-channels" in b.stderr


def test_batched_wav_nnet3_cuda_first_generation_driver(tmp_path):
    """batched-wav-nnet3-cuda (cudadecoderbin/batched-wav-nnet3-cuda.cc:114): the v1 driver's command line over the same pipeline -- same lattices as batched-wav-nnet3-cuda2,
    every iteration written under "<iteration>-<utt>", "~Group" lines, the likelihood-per-frame line, the v1-only options accepted"""
    td = str(tmp_path); N = 120; lens = [16000, 9000, 23001, 12000, 8000]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    synth.make_hclg(3000, 8000, N, seed=11, start_degree=50).write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-batch-size=3"]
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + tail + [f"ark,t:{td}/v2.txt"], capture_output=True, text=True); assert a.returncode == 0, a.stderr
    b = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda")] + common + ["--iterations=2", "--batch-drain-size=2", "--cuda-control-threads=1", "--max-outstanding-queue-length=100", "--cuda-worker-threads=4"] + tail + [f"ark,t:{td}/v1.txt"],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    assert "~Group 0 completed Aggregate Total Time:" in b.stderr and "~Group 1 completed" in b.stderr and "Decoded 5 utterances, 0 with errors." in b.stderr
    assert "Overall likelihood per frame was" in b.stderr and "Overall:  Aggregate Total Time:" in b.stderr
    rec = lambda path: {blk.split("\n", 1)[0].strip(): blk.split("\n", 1)[1] for blk in open(path).read().strip().split("\n\n")}
    r2, r1 = rec(f"{td}/v2.txt"), rec(f"{td}/v1.txt")
    assert sorted(r1) == sorted(list(r2) + ["1-" + k for k in r2]), sorted(r1)
    for k in r2: assert r1[k] == r2[k] and r1["1-" + k] == r2[k], k      # determinized CompactLattices, character for character
    assert float(b.stderr.split("per frame over ")[1].split(" frames")[0]) > 0
    # cuda2-only behaviour stays with cuda2: a CTM file name as output is refused without a postprocessor by both; the v1-only options are unknown to cuda2
    u = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--batch-drain-size=2"] + common + tail + [f"ark,t:{td}/x.txt"], capture_output=True, text=True); assert u.returncode != 0
    assert subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda"), "x"], capture_output=True).returncode == 1


def test_ntokens_pre_allocated_is_a_reservation_60s_utterance_equals_the_reference_decoder(tmp_path):
    """--ntokens-pre-allocated only reserves (cudadecoder/cuda-decoder.cc:232-238): a 60 s utterance that needs many times --ntokens-pre-allocated=200000 is decoded without
    error -- its lane moves to bigger pools inside the token-passing kernel -- and its raw lattice is the reference LatticeFasterDecoder's (oracle/_ref, the reference's own
    decoder source) on the same log-likelihoods: every arc, label and cost bit; the same file next to short ones, and with a roomy reservation, gives the same record."""
    from kaldi_amd import feat, nnet3
    from oracle import kaldi_io as kio, ref_decoder as rd, lattice_oracle as lo
    from tests import lattice_sig as lsig
    td = str(tmp_path); N = 120; lens = [960000, 16000, 9000]
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst"); t2p = synth.tid2pdf(N)
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
    base = [os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000",
            "--max-batch-size=3", "--determinize-lattice=false", "--write-compact=false"]
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    a = subprocess.run(base + ["--ntokens-pre-allocated=200000"] + tail + [f"ark,t:{td}/small.txt"], capture_output=True, text=True); assert a.returncode == 0, a.stderr
    assert "Decoded 3 utterances, 0 with errors." in a.stderr, a.stderr[-600:]
    b = subprocess.run(base + ["--ntokens-pre-allocated=12000000"] + tail + [f"ark,t:{td}/big.txt"], capture_output=True, text=True); assert b.returncode == 0, b.stderr
    ls_, lb_ = _parse_text_lattices(f"{td}/small.txt"), _parse_text_lattices(f"{td}/big.txt")
    canon = lambda lat: (sorted((x[2], x[3], float(x[4]), float(x[5])) for x in lat[0]), sorted(float(v) for v in lat[1].values()))      # (state numbers depend on the order the GPU happened to emit the arcs in)
    assert sorted(ls_) == sorted(lb_) == ["utt0", "utt1", "utt2"]
    for k in ls_: assert canon(ls_[k]) == canon(lb_[k]) and len(ls_[k][0]) > 0, k      # the same lattices whatever the reservation
    # the reference decoder on the log-likelihoods the program's network stage makes of the long file
    dev = torch.device("cuda:0"); sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    w = kio.read_wav(f"{td}/u0.wav")[0].astype(np.float32); wo, fo, total, fo_h = sf.offsets([len(w)], dev)
    nn = nnet3.Nnet(f"{td}/final.mdl"); nb = nnet3.NnetBatch(nn, [total], 3); ll = nb.forward(sf.ComputeFeatures(torch.from_numpy(w).to(dev), wo, fo, total)).cpu().numpy()
    from kaldi_amd import decoder
    dec = decoder.CudaDecoder(decoder.CudaFst(graph, t2p), decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, literal_order=1, lane_tokens_cap=200000, lane_links_cap=400000), 1, N)
    dec.DecodeBatch(torch.from_numpy(ll).to(dev), np.array([0, ll.shape[0]])); info = dec.LatticeInfo(); lat = dec.GetRawLattices(copy=True)[0]
    assert info[0, 2] == 0 and info[0, 4] > 2 * 200000 and dec.PoolGrowths()[0] >= 2, (info[0], dec.PoolGrowths())
    if rd.available():
        assert lsig.canonical_of_reference(rd.decode(graph, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, max_active=10000))) == lsig.canonical_of_raw(lat)
    # ... and it is the record the program wrote (Connect()-ed): same arcs
    t = subprocess.run(base + ["--ntokens-pre-allocated=200000", "--file-limit=1"] + tail + [f"ark,t:{td}/one.txt"], capture_output=True, text=True); assert t.returncode == 0, t.stderr
    arcs, fins = _parse_text_lattices(f"{td}/one.txt")["utt0"]; ref = lat.connect()
    assert len(arcs) == ref.num_arcs and sorted((x[2], x[3]) for x in arcs) == sorted(zip(ref.arc_ilabel.tolist(), ref.arc_olabel.tolist()))


def _online_setup(td, lens, N=120):
    _wavs(td, lens)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
    synth.make_hclg(3000, 8000, N, seed=11, start_degree=50).write_openfst(f"{td}/HCLG.fst")
    open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n"); open(f"{td}/pp.conf", "w").write("--acoustic-scale=0.7\n--lm-scale=1.5\n")
    return ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=6.0", "--max-active=10000"]

def _corr_lines(stderr, tag):
    """'corr_id #N<tag>...' log lines -> {N: [texts]} (cudadecoderbin/batched-wav-nnet3-cuda-online.cc:180-215)"""
    import re
    out = {}
    for l in stderr.splitlines():
        m = re.search(r"corr_id #(\d+)" + re.escape(tag) + r"(.*)$", l)
        if m: out.setdefault(int(m.group(1)), []).append(m.group(2).strip())
    return out

def test_online_program_print_hypotheses_equal_the_offline_lattices_best_path(tmp_path):
    """--print-hypotheses (cuda-bin-tools.h:83-85, batched-wav-nnet3-cuda-online.cc:207-210): the words of the final best path of every stream, 'corr_id #N : words'; with
    --word-symbol-table the symbols.  Equal to the cheapest path of the lattice batched-wav-nnet3-cuda2 writes for the same file.  No lattice is written without --write-lattice=true
    (the reference's default) and the program says so."""
    td = str(tmp_path); lens = [16000, 9000, 23001]; common = _online_setup(td, lens)
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=3"] + tail + [f"ark,t:{td}/det.txt"], capture_output=True, text=True); assert a.returncode == 0, a.stderr[-2000:]
    from tests.lattice_cases import parse_compact_text
    want = {k: _compact_best_words(c) for k, c in parse_compact_text(open(f"{td}/det.txt").read()).items()}
    b = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + ["--num-channels=3", "--frames-per-chunk=60", "--max-utterance-frames=400", "--print-hypotheses=true"] + tail + [f"ark:{td}/none.ark"],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    assert "please set --write-lattice=true" in b.stderr and not os.path.exists(f"{td}/none.ark")
    hyp = _corr_lines(b.stderr, " :")
    assert sorted(hyp) == [0, 1, 2] and all(len(v) == 1 for v in hyp.values())
    for u in range(3): assert [int(w) for w in hyp[u][0].split()] == want[f"utt{u}"] and len(want[f"utt{u}"]) > 0, u      # (stream N = the N-th file admitted)
    assert "Latency stats:" in b.stderr and "Latencies (s):" in b.stderr      # PrintLatencyStats (cuda-bin-tools.h:33-55)
    # the symbols of --word-symbol-table
    ids = sorted({w for v in want.values() for w in v}); open(f"{td}/words.txt", "w").write("<eps> 0\n" + "".join(f"W{w} {w}\n" for w in ids))
    c = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + ["--num-channels=3", "--frames-per-chunk=60", "--max-utterance-frames=400", "--print-hypotheses=true", f"--word-symbol-table={td}/words.txt"]
                       + tail + [f"ark:{td}/none.ark"], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    for u, v in _corr_lines(c.stderr, " :").items(): assert v[0].split() == [f"W{w}" for w in want[f"utt{u}"]], u

def test_online_program_partial_hypotheses_and_endpoints(tmp_path):
    """--print-partial-hypotheses / --print-endpoints (batched-wav-nnet3-cuda-online.cc:194-205): after every chunk of a stream that goes on, the words of its best path so far
    (no final-probs) and whether kaldi::EndpointDetected fires on it (online2/online-endpoint.cc:26-72; the rules' options as OnlineEndpointConfig registers them).  The
    partial hypotheses of a stream are the k3_decoder_get_best_path results the library's own test pins to the lattice's best path; here: one line per chunk but the last, and the
    end-point lines follow the rules -- rule 5 alone (utterance >= 0.9 s) fires from the chunk that brings a stream to 0.9 s on, with every rule out of reach never."""
    td = str(tmp_path); lens = [32000, 9000, 23001]; common = _online_setup(td, lens)
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark:{td}/none.ark"]
    C = 30; off = ["--endpoint.rule%d.min-trailing-silence=1000" % r for r in (1, 2, 3, 4)]
    run = lambda extra: subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + ["--num-channels=3", f"--frames-per-chunk={C}", "--max-utterance-frames=400", "--print-partial-hypotheses=true",
                                                                                                     "--print-endpoints=true", "--print-hypotheses=true"] + extra + tail, capture_output=True, text=True)
    b = run(off + ["--endpoint.rule5.min-utterance-length=0.9"]); assert b.returncode == 0, b.stderr[-2000:]
    part, ep, fin = _corr_lines(b.stderr, " [partial] :"), _corr_lines(b.stderr, " [endpoint detected]"), _corr_lines(b.stderr, " :")
    chunk = C * 160
    for u, n in enumerate(lens):
        nchunks = -(-n // chunk)
        # every chunk of the stream but its last (whose result is the final hypothesis) -- once the stream has decoded frames: the network's right context keeps the first chunk or two back
        assert max(0, nchunks - 3) <= len(part.get(u, [])) <= nchunks - 1, (u, part.get(u))
        # a chunk's frames are decoded once the network has its right context, so the decoded length lags the audio: rule 5 fires on the later chunks of the long streams only
        assert len(ep.get(u, [])) <= nchunks - 1
        assert len(fin[u]) == 1
        if part.get(u): assert len(part[u][-1].split()) <= len(fin[u][0].split()) + 3      # (a prefix-like partial result: never much longer than the final one)
    assert 1 <= len(ep.get(0, [])) <= len(part[0]) - 1 and len(ep.get(1, [])) == 0      # 2 s stream: 0.9 s of decoded frames before its end, not from its first chunks on; 0.56 s stream: never
    d = run(off + ["--endpoint.rule5.min-utterance-length=0.25"]); assert d.returncode == 0, d.stderr[-2000:]      # one chunk of decoded frames is enough: (almost) every partial result is an end-point
    ep2 = _corr_lines(d.stderr, " [endpoint detected]")
    assert len(part[0]) - 1 <= len(ep2.get(0, [])) <= len(part[0]) and len(ep2.get(0, [])) > len(ep.get(0, []))
    c = run(off + ["--endpoint.rule5.min-utterance-length=1000"]); assert c.returncode == 0, c.stderr[-2000:]
    assert not _corr_lines(c.stderr, " [endpoint detected]") and _corr_lines(c.stderr, " [partial] :") == part      # no rule in reach: no end-points; the hypotheses do not depend on them

def test_online_program_generate_lattice_postprocessor_and_ctm(tmp_path):
    """--lattice-postprocessor-rxfilename + a fourth argument that is not a table wspecifier -> CTM output (cuda-bin-tools.h:181-195, batched-wav-nnet3-cuda-online.cc:92-118,
    228-262); --write-lattice=true with the post-processor -> its scaled lattices.  Both equal to what batched-wav-nnet3-cuda2 produces from the same files; the real-time
    simulation (--simulate-realtime-writing=true: the reference's pacing) gives the same CTM and sane latencies."""
    td = str(tmp_path); lens = [16000, 9000, 23001]; common = _online_setup(td, lens)
    tail = [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp"]; pp = [f"--lattice-postprocessor-rxfilename={td}/pp.conf"]
    on = [os.path.join(BIN, "batched-wav-nnet3-cuda-online")] + common + ["--num-channels=3", "--frames-per-chunk=60", "--max-utterance-frames=400"]
    r = subprocess.run(on + tail + [f"{td}/on.ctm"], capture_output=True, text=True)
    assert r.returncode != 0 and "You must configure the lattice postprocessor" in r.stderr
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=3"] + pp + tail + [f"{td}/off.ctm"], capture_output=True, text=True); assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run(on + pp + tail + [f"{td}/on.ctm"], capture_output=True, text=True); assert b.returncode == 0, b.stderr[-2000:]
    ctm = lambda f: sorted(tuple(l.split()) for l in open(f))
    assert ctm(f"{td}/on.ctm") == ctm(f"{td}/off.ctm") and len(ctm(f"{td}/on.ctm")) >= 3 and {l[0] for l in ctm(f"{td}/on.ctm")} == {"utt0", "utt1", "utt2"}
    # the post-processed lattices
    a = subprocess.run([os.path.join(BIN, "batched-wav-nnet3-cuda2")] + common + ["--max-batch-size=3"] + pp + tail + [f"ark,t:{td}/off.txt"], capture_output=True, text=True); assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run(on + pp + ["--write-lattice=true"] + tail + [f"ark,t:{td}/on.txt"], capture_output=True, text=True); assert b.returncode == 0, b.stderr[-2000:]
    from tests.lattice_cases import parse_compact_text
    la, lb = parse_compact_text(open(f"{td}/off.txt").read()), parse_compact_text(open(f"{td}/on.txt").read())
    assert sorted(la) == sorted(lb) == ["utt0", "utt1", "utt2"]
    for k in la: assert _compact_best_words(la[k]) == _compact_best_words(lb[k]) and len(la[k]["arcs"]) == len(lb[k]["arcs"]), k
    # the reference's pacing: every stream played in real time (2 s of wall clock for these files)
    import re, time
    t0 = time.time()
    c = subprocess.run(on + pp + ["--simulate-realtime-writing=true"] + tail + [f"{td}/rt.ctm"], capture_output=True, text=True); assert c.returncode == 0, c.stderr[-2000:]
    assert time.time() - t0 >= max(lens) / 16000.0 and ctm(f"{td}/rt.ctm") == ctm(f"{td}/off.ctm")
    m = re.search(r"Latencies \(s\):.*\n.*?([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", c.stderr, re.M); assert m, c.stderr[-1500:]
    assert 0.0 <= float(m.group(1)) <= float(m.group(4)) + 1e-9 and float(m.group(4)) < 30.0      # (result time - end of speech; the first run of a process includes its warm-up)
