"""GPU parity: k3_nnet_forward (fused FP32-MFMA TDNN/TDNN-F forward through the C ABI) vs
(a) outputs of the REFERENCE's own nnet3-compute binary committed as fixtures -- a small TDNN-F written by the reference's
    nnet3-copy (tests/golden/nnet_small.*) and the full benchmark model (tests/golden/nnet_bench_io.npz) -- at the
    north_star bound |delta| <= 1e-4 on the pseudo log-likelihoods, and
(b) the float32 numpy oracle on seeded models.  numpy's BLAS on the test machine is NOT the reference's MKL build: two
    float32 evaluations of a 35-GEMM network differ by ~1e-5 relative whatever the BLAS (measured: reference vs float64
    7e-4, numpy here vs reference 6e-5, numpy on the GPU box vs this kernel 2e-4 at |x| ~ 20), so against
    numpy the bound is 1e-4 absolute + 1e-5 relative; the float64 evaluation is reported for information only."""
import os, numpy as np, pytest, torch
from kaldi_amd import synth
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4

def _check(no, onet, f, g, s, lp=None, acwt=1.0):
    ref32 = no.compute(onet, f, s, lp, acwt)
    assert g.shape == ref32.shape
    err = (np.abs(g - ref32) - 1e-5 * np.abs(ref32)).max()
    if err > TOL:
        ref64 = no.compute(onet, f, s, lp, acwt, dtype=np.float64)
        raise AssertionError(f"T={f.shape[0]}: |hip-f32 oracle|={err:.3g} |hip-f64|={np.abs(g - ref64).max():.3g} "
                             f"|f32 oracle-f64|={np.abs(ref32 - ref64).max():.3g} max|x|={np.abs(ref32).max():.3g}")

def _forward(model_path, feats_list, s, log_priors=None, acwt=1.0):
    from kaldi_amd import nnet3
    dev = torch.device("cuda:0")
    n = nnet3.Nnet(model_path)
    b = nnet3.NnetBatch(n, [f.shape[0] for f in feats_list], s, log_priors, acwt)
    x = torch.from_numpy(np.concatenate(feats_list)).to(dev)
    y = b.forward(x); torch.cuda.synchronize()
    y = y.cpu().numpy()
    return [y[b.out_offsets[i]:b.out_offsets[i + 1]] for i in range(len(feats_list))], b

@pytest.mark.parametrize("fmt", ["raw", "txt"])
@pytest.mark.parametrize("s", [1, 3])
def test_hip_vs_reference_nnet3_compute(fmt, s):
    g = np.load(os.path.join(GOLD, "nnet_small_io.npz"))
    got, _ = _forward(os.path.join(GOLD, "nnet_small." + fmt), [g["feats"]], s)
    ref = g[f"ref_out_{fmt}_s{s}"]
    assert got[0].shape == ref.shape
    assert np.abs(got[0] - ref).max() <= 1e-4, np.abs(got[0] - ref).max()

@pytest.mark.parametrize("s", [1, 3])
def test_hip_renorm_sigmoid_tanh_vs_reference_nnet3_compute(s):
    """tests/golden/nnet_renorm*: NormalizeComponent / SigmoidComponent / TanhComponent / LogSoftmax layers (model by the reference's nnet3-init, outputs by its nnet3-compute): the fused
    path folds ReLU / sigmoid / tanh into the producing GEMM's epilogue and runs the row normalisation behind it; the three utterances as one ragged batch and one by one"""
    g = np.load(os.path.join(GOLD, "nnet_renorm_io.npz")); us = ("u0", "u1", "u2")
    got, _ = _forward(os.path.join(GOLD, "nnet_renorm.raw"), [g["feats_" + u] for u in us], s)
    for y, u in zip(got, us):
        ref = g[f"ref_s{s}_{u}"]; assert y.shape == ref.shape and np.abs(y - ref).max() <= 1e-4, (u, np.abs(y - ref).max())
        one, _ = _forward(os.path.join(GOLD, "nnet_renorm.raw"), [g["feats_" + u]], s); assert np.array_equal(one[0], y), u

def test_hip_vs_reference_nnet3_compute_benchmark_model(tmp_path):
    """the 17L-768/96-6024 benchmark model (regenerated bit-identically, sha256 checked) vs the reference binary's output"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_golden_nnet_bench.py")); mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    g = np.load(os.path.join(GOLD, "nnet_bench_io.npz"))
    p = str(tmp_path / "m.raw")
    _, sha = mk.bench_model_and_feats(p)
    if sha != str(g["model_sha256"]): pytest.skip("synth model is not bit-reproducible on this machine; fixture does not apply")
    got, _ = _forward(p, [g["feats"]], 3)
    ref = g["ref_out_cols8"]
    assert got[0][:, ::8].shape == ref.shape
    err = np.abs(got[0][:, ::8] - ref).max()
    assert err <= TOL, (err, float(g["max_abs"]))

def _feats(rng, T, dim=40):
    return (rng.standard_normal((T, dim)) * 1.2 + 16.5).astype(np.float32)

def test_hip_vs_oracle_ragged_tdnnf(tmp_path):
    """17-layer layout at reduced width, ragged batch incl. T=1, T < context, T not a multiple of 3."""
    from oracle import nnet3_oracle as no
    net = synth.make_tdnnf(seed=2, dim=128, bottleneck=32, prefinal_small=64, num_pdfs=300, calib_frames=400, out_std=1.5)
    p = str(tmp_path / "m.raw"); net.write(p)
    rng = np.random.default_rng(5)
    lens = [1, 2, 3, 4, 17, 100, 257, 998, 131]
    feats = [_feats(rng, T) for T in lens]
    onet = no.read_nnet(p)
    for s in (3, 1):
        got, b = _forward(p, feats, s)
        assert b.flops > 0
        for f, g in zip(feats, got):
            _check(no, onet, f, g, s)

def test_hip_vs_oracle_tdnn_config1_priors_acwt(tmp_path):
    """BASELINE config 1 model (3x512 TDNN, 2000 outputs) incl. the -log(prior) * acwt epilogue."""
    from oracle import nnet3_oracle as no
    net = synth.make_tdnn(seed=1)
    p = str(tmp_path / "m.raw"); net.write(p)
    rng = np.random.default_rng(6)
    feats = [_feats(rng, T) for T in (300, 41)]
    lp = np.log(rng.dirichlet(np.ones(2000)).astype(np.float32) + 1e-8)
    got, _ = _forward(p, feats, 1, lp, 0.1)
    onet = no.read_nnet(p)
    for f, g in zip(feats, got):
        _check(no, onet, f, g, 1, lp, 0.1)

def test_two_halves_on_two_streams_equal_one_plan(tmp_path, monkeypatch):
    """K3_NNET_SPLIT=1 plans a batch as two halves of its utterances run on two streams (k3_nnet_batch_create; opt-in): the same tiles, the same
    arithmetic -- every value bit-identical to the single plan, ragged lengths (incl. a 1-frame utterance at the cut), priors and scale, on the caller's own stream,
    call after call (the fork / join events are reused)"""
    from kaldi_amd import nnet3
    net = synth.make_tdnnf(seed=3, dim=128, bottleneck=32, prefinal_small=64, num_pdfs=300, calib_frames=400, out_std=1.5)
    p = str(tmp_path / "m.raw"); net.write(p); n = nnet3.Nnet(p); dev = torch.device("cuda:0")
    rng = np.random.default_rng(11); lens = [257, 3, 998, 1, 1, 131, 640, 17, 100]; lp = np.log(rng.dirichlet(np.ones(300)).astype(np.float32) + 1e-8)
    xs = [torch.from_numpy(np.concatenate([_feats(rng, T) for T in lens])).to(dev) for _ in range(3)]
    for s_, kw in ((3, {}), (1, dict(log_priors=lp, acoustic_scale=0.3))):
        monkeypatch.setenv("K3_NNET_SPLIT", "0"); one = nnet3.NnetBatch(n, lens, s_, **kw)
        monkeypatch.setenv("K3_NNET_SPLIT", "1"); two = nnet3.NnetBatch(n, lens, s_, **kw)
        assert (one.out_offsets == two.out_offsets).all() and one.total_out_rows == two.total_out_rows and one.flops == two.flops
        st = torch.cuda.Stream(priority=-1)
        for k, x in enumerate(xs):
            ref = one.forward(x).clone()
            if k == 1:
                with torch.cuda.stream(st): got = two.forward(x)
                st.synchronize()
            else: got = two.forward(x)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (s_, k)
    monkeypatch.setenv("K3_NNET_SPLIT", "1"); single = nnet3.NnetBatch(n, [50], 3)      # one utterance: nothing to split
    assert single.forward(xs[0][:50]).shape[0] == single.total_out_rows

def test_hip_full_model_spot_check(tmp_path):
    """The benchmark model (17L-768/96-6024) on a 64-utterance batch: every utterance is the same signal, so all
    outputs must be identical across the batch (row-mapping / tile-boundary check at full width) and utterance 0
    must match the oracle."""
    from oracle import nnet3_oracle as no
    net = synth.make_tdnnf(seed=1)
    p = str(tmp_path / "m.raw"); net.write(p)
    rng = np.random.default_rng(8)
    f = _feats(rng, 333)
    got, _ = _forward(p, [f] * 64, 3)
    for g in got[1:]:
        assert np.array_equal(g, got[0])
    _check(no, no.read_nnet(p), f, got[0], 3)


def test_batched_static_nnet3_streaming_equals_whole_utterance(tmp_path):
    """Chunked streaming forward (BatchedStaticNnet3.RunBatch: per-channel input context stashed between calls, first/last chunk edge
    replication, right-context latency, end-of-stream flush) vs the whole-utterance forward: the same rows, bit for bit."""
    import torch
    from kaldi_amd import nnet3, synth
    rng = np.random.default_rng(11); dev = torch.device("cuda:0")
    calib = (rng.standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net_w = synth.make_tdnnf(seed=5, dim=96, bottleneck=24, strides=(1, 1, 0, 3, 3), prefinal_small=48, num_pdfs=120, calib_feats=calib, out_std=1.5)
    mp = str(tmp_path / "m.raw"); net_w.write(mp)
    net = nnet3.Nnet(mp); s = 3
    lens = [1, 2, 47, 150, 151, 333, 610, 29]
    utts = [torch.from_numpy((rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32)).to(dev) for T in lens]
    whole = nnet3.NnetBatch(net, lens, s); ref = whole.forward(torch.cat(utts, 0)); torch.cuda.synchronize()
    ref_u = [ref[whole.out_offsets[u]:whole.out_offsets[u + 1]] for u in range(len(lens))]
    for C, B in ((150, 4), (30, 3)):
        drv = nnet3.BatchedStaticNnet3(net, max_batch_size=B, nchannels=B + 2, frames_per_chunk=C, frame_subsampling_factor=s)
        assert drv.GetNOutputFramesPerChunk() == C // s
        got = [[] for _ in lens]; pos = [0] * len(lens); todo = list(range(len(lens))); chan_of = {}; free = list(range(B + 2))
        while todo or chan_of:
            # admit utterances to free channels, then run one batch of up to B slots with random chunk sizes
            while todo and free and len(chan_of) < B + 2: chan_of[todo.pop(0)] = free.pop(0)
            slots = list(chan_of.items())[:B]
            chans, chunks, first, last = [], [], [], []
            for u, ch in slots:
                n = int(rng.integers(0, C + 1)) if lens[u] - pos[u] > 3 else lens[u] - pos[u]
                n = min(n, lens[u] - pos[u])
                chans.append(ch); chunks.append(utts[u][pos[u]:pos[u] + n]); first.append(pos[u] == 0); pos[u] += n; last.append(pos[u] == lens[u])
            outs = drv.RunBatch(chans, chunks, first, last)
            for (u, ch), o, l in zip(slots, outs, last):
                got[u].append(o)
                if l: free.append(chan_of.pop(u))
        for u in range(len(lens)):
            g = torch.cat(got[u], 0)
            assert g.shape == ref_u[u].shape, (C, u, g.shape, ref_u[u].shape)
            assert torch.equal(g, ref_u[u]), (C, u, (g - ref_u[u]).abs().max().item())


def test_stateful_streaming_forward_equals_whole_utterance(tmp_path):
    """k3_nnet_stream_* (round 5): every node keeps its last rows per channel, a pass evaluates frames_per_chunk NEW frames per channel and nothing else -- no re-evaluation of a
    chunk's left / right context as in BatchedStaticNnet3::RunBatch.  Streams of every length (1 frame .. 12 chunks), random chunk sizes (frames are buffered to whole chunks),
    channels reused by later streams, channels that sit passes out, the end of a stream flushed with its last frame replicated: the rows must be those of the whole-utterance
    forward BIT FOR BIT, and every output row must come out exactly once."""
    import torch
    from kaldi_amd import nnet3, synth
    rng = np.random.default_rng(12); dev = torch.device("cuda:0")
    calib = (rng.standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net_w = synth.make_tdnnf(seed=5, dim=96, bottleneck=24, strides=(1, 1, 0, 3, 3), prefinal_small=48, num_pdfs=120, calib_feats=calib, out_std=1.5)
    mp = str(tmp_path / "m.raw"); net_w.write(mp)
    net = nnet3.Nnet(mp); s = 3
    lens = [1, 2, 47, 150, 151, 333, 610, 29, 51, 52, 102, 3]
    utts = [torch.from_numpy((rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32)).to(dev) for T in lens]
    whole = nnet3.NnetBatch(net, lens, s); ref = whole.forward(torch.cat(utts, 0)); torch.cuda.synchronize()
    ref_u = [ref[whole.out_offsets[u]:whole.out_offsets[u + 1]] for u in range(len(lens))]
    lp = rng.uniform(-6, -1, 120).astype(np.float32)
    whole2 = nnet3.NnetBatch(net, lens, s, lp, 0.7); ref2 = whole2.forward(torch.cat(utts, 0)); torch.cuda.synchronize()
    for C, nch, B, priors in ((51, 5, 3, False), (30, 4, 4, False), (150, 6, 2, True)):
        drv = nnet3.StreamNnet3(net, nch, frames_per_chunk=C, frame_subsampling_factor=s, log_priors=lp if priors else None, acoustic_scale=0.7 if priors else 1.0)
        assert drv.GetNOutputFramesPerChunk() == C // s and drv.info.right_context == net.info.right_context
        got = [[] for _ in lens]; pos = [0] * len(lens); todo = list(range(len(lens))); chan_of = {}; free = list(range(nch))
        while todo or chan_of:
            while todo and free: chan_of[todo.pop(0)] = free.pop(0)
            slots = [kv for kv in chan_of.items() if rng.uniform() < 0.8][:B] or list(chan_of.items())[:1]      # some channels sit this call out
            chans, chunks, first, last = [], [], [], []
            for u, ch in slots:
                n = int(rng.integers(0, C + 1)) if lens[u] - pos[u] > 3 else lens[u] - pos[u]
                n = min(n, lens[u] - pos[u])
                chans.append(ch); chunks.append(utts[u][pos[u]:pos[u] + n]); first.append(pos[u] == 0); pos[u] += n; last.append(pos[u] == lens[u])
            outs = drv.RunBatch(chans, chunks, first, last)
            for (u, ch), o, l in zip(slots, outs, last):
                got[u].append(o)
                if l: free.append(chan_of.pop(u))
        want = [ref2[whole2.out_offsets[u]:whole2.out_offsets[u + 1]] for u in range(len(lens))] if priors else ref_u
        for u in range(len(lens)):
            g = torch.cat(got[u], 0)
            assert g.shape == want[u].shape, (C, u, lens[u], g.shape, want[u].shape)
            assert torch.equal(g, want[u]), (C, u, lens[u], (g - want[u]).abs().max().item())
    # a configuration the stateful engine cannot run says so (the callers fall back to chunk + context)
    from kaldi_amd import lib
    with pytest.raises(lib.K3Error): nnet3.StreamNnet3(net, 2, frames_per_chunk=4, frame_subsampling_factor=3)      # a chunk that is not a whole number of output frames


def test_split_bf16_products_are_as_accurate_as_the_fp32_matrix_core(tmp_path):
    """EXPLORATORY path (k3_nnet_batch_set_precision(.., 1), never the default): every affine product as six bf16 matrix-core products over three-way split operands.  The split
    is exact and the dropped cross terms are below 2^-24, so the forward must sit as close to the float64 forward (oracle, the checker) as the FP32 matrix-core forward does --
    what it may NOT be asked is the reference's rounding (it sums in another order): only a loose bound on its distance to the FP32 path."""
    import torch
    from kaldi_amd import nnet3, synth
    from oracle import nnet3_oracle as no
    rng = np.random.default_rng(21); dev = torch.device("cuda:0")
    calib = (rng.standard_normal((300, 40)) * 1.2 + 16.5).astype(np.float32)
    mp = str(tmp_path / "m.raw"); synth.make_tdnnf(seed=9, dim=256, bottleneck=64, strides=(1, 1, 0, 3, 3, 3), prefinal_small=64, num_pdfs=500, calib_feats=calib, out_std=2.0).write(mp)
    net = nnet3.Nnet(mp); lens = [333, 200, 97]
    utts = [(rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32) for T in lens]
    nb = nnet3.NnetBatch(net, lens, 3); x = torch.from_numpy(np.concatenate(utts)).to(dev)
    y32 = nb.forward(x).clone(); nb.set_precision(1); y6 = nb.forward(x).clone()
    # mode 2: the same six products from operand planes written by the PRODUCING epilogue (the loader only loads): the split is exact either way, so bit for bit mode 1's output
    nb.set_precision(2); y6p = nb.forward(x).clone(); y6p2 = nb.forward(x).clone()
    nb.set_precision(0); y32b = nb.forward(x).clone(); torch.cuda.synchronize()
    assert torch.equal(y32, y32b)                      # switching back restores the parity path bit for bit
    assert torch.equal(y6p, y6) and torch.equal(y6p2, y6)
    onet = no.read_nnet(mp); e32 = e6 = 0.0
    for u, f in enumerate(utts):
        t = no.compute(onet, f, 3, dtype=np.float64); sl = slice(nb.out_offsets[u], nb.out_offsets[u + 1])
        e32 = max(e32, float(np.abs(y32[sl].cpu().numpy() - t).max())); e6 = max(e6, float(np.abs(y6[sl].cpu().numpy() - t).max()))
    assert not torch.equal(y6, y32), "the split-bf16 kernel did not run"
    assert e6 <= 1.5 * e32 + 2e-5, (e6, e32)
    assert float((y6 - y32).abs().max()) <= 4.0 * e32 + 1e-4


# ---------------------------------------------------------------------------------------------- models with an i-vector input
IV_CASES = {"s1_c50_p10": (1, 50, 10, False), "s3_c50_p10": (3, 50, 10, False), "s3_c21_p7": (3, 21, 7, False), "s1_c20_p10_short": (1, 20, 10, False), "s3_utt": (3, 50, 0, True), "s1_utt": (1, 50, 0, True)}

def _forward_iv(n, feats_list, s, iv_list, period, chunk, utt_level):
    from kaldi_amd import nnet3
    dev = torch.device("cuda:0")
    b = nnet3.NnetBatch(n, [f.shape[0] for f in feats_list], s, ivector_rows=None if utt_level else [iv.shape[0] for iv in iv_list], online_ivector_period=period, frames_per_chunk=chunk)
    x = torch.from_numpy(np.concatenate(feats_list)).to(dev); v = torch.from_numpy(np.concatenate([np.atleast_2d(iv) for iv in iv_list]).astype(np.float32)).to(dev)
    y = b.forward(x, ivectors=v); torch.cuda.synchronize(); y = y.cpu().numpy()
    return [y[b.out_offsets[i]:b.out_offsets[i + 1]] for i in range(len(feats_list))]

@pytest.mark.parametrize("name", sorted(IV_CASES))
def test_ivector_input_vs_reference_nnet3_compute(name):
    """tests/golden/nnet_ivector*: the REFERENCE's nnet3-compute --online-ivectors / --ivectors on a model with the recipe's i-vector input"""
    from kaldi_amd import nnet3
    g = np.load(os.path.join(GOLD, "nnet_ivector_io.npz")); n = nnet3.Nnet(os.path.join(GOLD, "nnet_ivector.raw")); assert n.info.ivector_dim == 12
    s, chunk, period, utt = IV_CASES[name]
    got = _forward_iv(n, [g["feats"]], s, [g["iv_" + name]], period, chunk, utt)[0]; ref = g["ref_" + name]
    assert got.shape == ref.shape and np.abs(got - ref).max() <= TOL, np.abs(got - ref).max()

def test_ivector_input_ragged_batch_vs_oracle(tmp_path):
    """several utterances of different lengths (one shorter than a chunk, one a single frame), each with its own online i-vectors, in one batch"""
    from kaldi_amd import nnet3
    from oracle import nnet3_oracle as no
    net = synth.make_tdnnf(seed=21, dim=64, bottleneck=16, strides=(1, 3, 0, 3), prefinal_small=32, num_pdfs=120, calib_frames=300, ivector_dim=20)
    p = str(tmp_path / "m.raw"); net.write(p); onet = no.read_nnet(p); n = nnet3.Nnet(p); rng = np.random.default_rng(3)
    for s, chunk, period in ((3, 50, 10), (1, 30, 10), (3, 150, 5)):
        lens = [260, 17, 1, 149, 75]
        feats = [_feats(rng, T) for T in lens]; ivs = [(rng.standard_normal(((T + period - 1) // period, 20)) * 0.7).astype(np.float32) for T in lens]
        got = _forward_iv(n, feats, s, ivs, period, chunk, False)
        for f, iv, g in zip(feats, ivs, got):
            want = no.compute(onet, f, s, online_ivectors=iv, online_ivector_period=period, frames_per_chunk=chunk)
            assert g.shape == want.shape and (np.abs(g - want) - 1e-5 * np.abs(want)).max() <= TOL, (s, chunk, f.shape[0], np.abs(g - want).max())
    # one i-vector per utterance
    feats = [_feats(rng, T) for T in (90, 33)]; ivs = [rng.standard_normal(20).astype(np.float32) for _ in range(2)]
    got = _forward_iv(n, feats, 3, ivs, 0, 50, True)
    for f, iv, g in zip(feats, ivs, got):
        want = no.compute(onet, f, 3, ivector=iv); assert g.shape == want.shape and (np.abs(g - want) - 1e-5 * np.abs(want)).max() <= TOL

def test_ivector_input_errors(tmp_path):
    from kaldi_amd import nnet3, lib
    g = np.load(os.path.join(GOLD, "nnet_ivector_io.npz")); n = nnet3.Nnet(os.path.join(GOLD, "nnet_ivector.raw")); dev = torch.device("cuda:0")
    b = nnet3.NnetBatch(n, [131], 1, ivector_rows=[14], online_ivector_period=10)
    with pytest.raises(lib.K3Error, match="i-vector"): b.forward(torch.from_numpy(g["feats"]).to(dev))            # the i-vectors are not optional for this model
    with pytest.raises(lib.K3Error, match="Could not get iVector"): nnet3.NnetBatch(n, [131], 1, ivector_rows=[3], online_ivector_period=10)   # far too few rows: the reference's error
    plain = nnet3.Nnet(os.path.join(GOLD, "nnet_small.raw"))
    with pytest.raises(lib.K3Error, match="no i-vector input"):
        h = __import__("ctypes").c_void_p(); nf = np.array([50], np.int32); rows = np.array([5], np.int32)
        lib.check(lib.load().k3_nnet_batch_create_ivector(plain._h, 1, nf.ctypes.data, 1, None, 1.0, 50, 10, rows.ctypes.data, __import__("ctypes").byref(h)))
