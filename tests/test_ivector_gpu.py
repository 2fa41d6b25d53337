"""GPU parity of the online i-vector extraction (k3_ivector_* through the C ABI) against
  - the REFERENCE binary's output (ivector-extract-online2, tests/golden/ivector/ivector_golden.npz), model files read by libk3host, and
  - oracle/ivector_oracle.py (itself pinned to that binary in tests/test_oracle_ivector.py) on seeded random models and options.
Tolerance: 2e-5 absolute on i-vector entries of magnitude ~0.2 .. 3 (the oracle itself is 8e-6 from the binary: float32 feature statistics summed
in a different order, amplified by 15 conjugate-gradient iterations); measured values are printed by the tests."""
import os, numpy as np, pytest, torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); DIR = os.path.join(ROOT, "tests", "golden", "ivector")
TOL = 2e-5


def _run(ex, feats_list):
    dev = torch.device("cuda:0"); fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats_list])])
    x = torch.from_numpy(np.concatenate(feats_list).astype(np.float32)).to(dev)
    out, ro = ex.GetIvectors(x, fo); torch.cuda.synchronize(); out = out.cpu().numpy()
    return [out[ro[u]:ro[u + 1]] for u in range(len(feats_list))]


@pytest.fixture
def golden(monkeypatch):
    monkeypatch.chdir(DIR)                      # the config names its files relative to the working directory, like the reference
    return np.load(os.path.join(DIR, "ivector_golden.npz"))


def test_reference_binary_ivectors_batch_of_four(golden):
    from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
    info = OnlineIvectorExtractionInfo("ivector_extractor.conf"); ex = BatchedIvectorExtractor(info)
    assert (ex.FeatDim(), ex.LdaDim(), ex.IvectorDim(), ex.NumGauss()) == (13, 20, 16, 32)
    utts = ["utt0", "utt1", "utt2", "utt3"]
    got = _run(ex, [golden["feat_" + u] for u in utts]); worst = 0.0
    for u, g in zip(utts, got):
        ref = golden["iv_default_" + u]; assert g.shape == ref.shape
        worst = max(worst, np.abs(g - ref).max())
    print("max |gpu - reference binary| =", worst)
    assert worst <= TOL, worst
    # one utterance alone and in a different batch position gives the same bits (no cross-utterance state)
    alone = _run(ex, [golden["feat_utt2"]])[0]; assert np.array_equal(alone, got[2])
    rev = _run(ex, [golden["feat_" + u] for u in utts[::-1]]); assert all(np.array_equal(a, b) for a, b in zip(rev[::-1], got))


def test_quadratic_term_in_hbm_gives_the_same_bits(golden, monkeypatch):
    from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
    info = OnlineIvectorExtractionInfo("ivector_extractor.conf")
    a = _run(BatchedIvectorExtractor(info), [golden["feat_utt0"], golden["feat_utt3"]])
    monkeypatch.setenv("K3_IVECTOR_QUAD_IN_HBM", "1")
    b = _run(BatchedIvectorExtractor(info), [golden["feat_utt0"], golden["feat_utt3"]])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def _random_model(rng, F, lc, rc, D, G, R, offset_col):
    lda = rng.standard_normal((D, F * (lc + rc + 1) + (1 if offset_col else 0))).astype(np.float32) * 0.3
    n = 1000.0; mean = rng.standard_normal(F); var = 0.5 + rng.random(F)
    st = np.zeros((2, F + 1)); st[0, :F] = n * mean; st[0, F] = n; st[1, :F] = n * (var + mean * mean)
    means = rng.standard_normal((G, D)) * 1.5; var_g = 0.5 + rng.random((G, D)); w = rng.random(G) + 0.2; w /= w.sum()
    inv = 1.0 / var_g; miv = means * inv
    gc = np.log(w) - 0.5 * (D * np.log(2 * np.pi) + np.log(var_g).sum(1) + (means * means * inv).sum(1))
    ubm = dict(gconsts=gc.astype(np.float32), means_invvars=miv.astype(np.float32), inv_vars=inv.astype(np.float32))
    M = rng.standard_normal((G, D, R)) * 0.3; S = np.zeros((G, D, D))
    for g in range(G): a = rng.standard_normal((D, D)) * 0.2; S[g] = a @ a.T + np.diag(0.5 + rng.random(D))
    ie = dict(M=M, sigma_inv=S, prior_offset=float(rng.choice([0.0, 10.0, 100.0])), ivector_dim=R)
    return lda, st, ubm, ie


CASES = [  # F, lc, rc, D, G, R, offset column, options, frames per utterance
    (13, 3, 3, 20, 32, 16, False, dict(), [57, 140]),
    (10, 2, 1, 12, 70, 24, True, dict(num_gselect=3, min_post=0.0, posterior_scale=1.0, max_count=0.0, ivector_period=7, num_cg_iters=5), [1, 8, 64]),
    (16, 1, 1, 24, 130, 40, False, dict(num_gselect=8, min_post=0.1, posterior_scale=0.3, max_count=20.0, ivector_period=4, num_cg_iters=15), [33, 90]),
    (8, 0, 0, 8, 5, 3, True, dict(num_gselect=5, min_post=0.3, posterior_scale=0.5, max_count=3.0, ivector_period=1, num_cg_iters=2), [12]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("exact", [0, 1])
def test_against_the_oracle_on_random_models(case, exact, monkeypatch):
    from kaldi_amd.ivector import BatchedIvectorExtractor
    from oracle import ivector_oracle as io
    F, lc, rc, D, G, R, off, o, lens = CASES[case]; rng = np.random.default_rng(100 + case)
    lda, st, ubm, ie = _random_model(rng, F, lc, rc, D, G, R, off)
    opts = dict(num_gselect=5, min_post=0.025, posterior_scale=0.1, max_count=0.0, ivector_period=10, num_cg_iters=15); opts.update(o)
    feats = [(rng.standard_normal((T, F)) * 2.0 + rng.standard_normal(F)).astype(np.float32) for T in lens]
    il = np.tril_indices(D); packed = np.stack([ie["sigma_inv"][g][il] for g in range(G)])
    ex = BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, ie["prior_offset"], left_context=lc, right_context=rc,
                                            exact_solve=exact, **opts)
    got = _run(ex, feats)
    if exact: monkeypatch.setattr(io, "_linear_cgd", lambda A, b, x, max_iters: np.linalg.solve(A, b))       # the fall-back branch of LinearCgd, taken always
    worst = 0.0
    for f, g in zip(feats, got):
        want = io.extract_online(f, ubm, ie, lda, st, left_context=lc, right_context=rc, **opts)
        assert g.shape == want.shape; scale = max(1.0, np.abs(want).max())
        worst = max(worst, np.abs(g - want).max() / scale)
    print("case", case, "exact", exact, "max scaled |gpu - oracle| =", worst)
    assert worst <= TOL, worst


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("variant", ["default", "short_cmn_window", "cmvn_for_the_statistics"])
def test_streaming_extraction_equals_the_whole_utterance_rows(case, variant):
    """k3_ivector_stream_*: the stream's frames in chunks of random length (1 frame .. several periods, an empty chunk, the end announced with or without frames) through the stages that
    carry state across calls -- CMVN window sums (with a window shorter than the stream: the subtraction branch), splice context, posterior-stage rows waiting for their period, statistics,
    the solver's warm start -- give, bit for bit, the rows of k3_ivector_extract_batch on the whole utterance, and Latest() follows the rule of the reference's online decodable"""
    from kaldi_amd.ivector import BatchedIvectorExtractor, IvectorStream
    F, lc, rc, D, G, R, off, o, lens = CASES[case]; rng = np.random.default_rng(300 + case)
    lda, st, ubm, ie = _random_model(rng, F, lc, rc, D, G, R, off)
    opts = dict(num_gselect=5, min_post=0.025, posterior_scale=0.1, max_count=0.0, ivector_period=10, num_cg_iters=15); opts.update(o)
    if variant == "short_cmn_window": opts.update(cmvn_cmn_window=17, cmvn_speaker_frames=17, cmvn_global_frames=9)
    if variant == "cmvn_for_the_statistics": opts.update(online_cmvn_iextractor=1, cmvn_normalize_variance=1)
    il = np.tril_indices(D); packed = np.stack([ie["sigma_inv"][g][il] for g in range(G)])
    ex = BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, ie["prior_offset"], left_context=lc, right_context=rc, **opts)
    dev = torch.device("cuda:0"); P = opts["ivector_period"]; stream = IvectorStream(ex)
    for T in lens + [3 * max(lens) + 5]:
        f = torch.from_numpy((rng.standard_normal((T, F)) * 2.0 + rng.standard_normal(F)).astype(np.float32)).to(dev)
        whole, _ = ex.GetIvectors(f, [0, T])
        for trial in range(3):
            stream.Reset(); pos = 0; got = []; end_with_frames = bool(rng.integers(0, 2))
            while pos < T:
                m = min(int(rng.integers(0, 3 * P + 2)) if trial else 1 + int(rng.integers(0, 4)), T - pos); last = pos + m >= T
                got.append(stream.AcceptFrames(f[pos:pos + m], last and end_with_frames).clone()); pos += m
                ready = pos - (0 if (last and end_with_frames) else rc)
                want = whole[(ready - 1) // P] if ready > 0 else torch.zeros_like(whole[0])
                assert torch.equal(stream.Latest(), want), (T, trial, pos)
            if not end_with_frames: got.append(stream.AcceptFrames(f[:0], True).clone())
            got = torch.cat(got)
            assert got.shape == whole.shape and torch.equal(got, whole), (T, trial, (got != whole).any(1).nonzero()[:3].tolist())
            assert stream.NumRows() == whole.shape[0] and torch.equal(stream.Latest(), whole[-1])
            with pytest.raises(Exception, match="has ended"): stream.AcceptFrames(f[:1], False)


@pytest.mark.parametrize("case", [0, 2, 3])
def test_batched_streaming_equals_the_per_stream_calls(case):
    """k3_ivector_stream_accept_batch: five streams of different lengths advancing together in chunks of random length (some empty, ends at different calls, one stream re-used for a
    second utterance after a reset) -- one launch per stage for the batch -- against the same chunks through k3_ivector_stream_accept stream by stream: the latest estimate after every
    call, the number of rows, and at the end the last row of the whole-utterance extraction"""
    from kaldi_amd.ivector import BatchedIvectorExtractor, IvectorStream, AcceptFramesBatch
    F, lc, rc, D, G, R, off, o, lens = CASES[case]; rng = np.random.default_rng(500 + case)
    lda, st, ubm, ie = _random_model(rng, F, lc, rc, D, G, R, off)
    opts = dict(num_gselect=5, min_post=0.025, posterior_scale=0.1, max_count=0.0, ivector_period=10, num_cg_iters=15); opts.update(o); opts.update(cmvn_cmn_window=23, cmvn_speaker_frames=23, cmvn_global_frames=11)
    il = np.tril_indices(D); packed = np.stack([ie["sigma_inv"][g][il] for g in range(G)])
    ex = BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, ie["prior_offset"], left_context=lc, right_context=rc, **opts)
    dev = torch.device("cuda:0"); P = opts["ivector_period"]; N = 5
    T = [int(x) for x in rng.integers(1, 120, N)]; T[1] = 1
    feats = [torch.from_numpy((rng.standard_normal((t, F)) * 2.0 + rng.standard_normal(F)).astype(np.float32)).to(dev) for t in T]
    batch = [IvectorStream(ex) for _ in range(N)]; single = [IvectorStream(ex) for _ in range(N)]; pos = [0] * N; done = [False] * N; second = False
    for call in range(400):
        live = [u for u in range(N) if not done[u]]
        if not live:
            if second: break
            second = True; u = 0; batch[u].Reset(); single[u].Reset(); pos[u] = 0; done[u] = False; live = [u]      # the stream object of utterance 0 takes a second utterance (the same frames)
        chunks, fins = [], []
        for u in live:
            m = min(int(rng.integers(0, 2 * P + 3)), T[u] - pos[u]); fin = pos[u] + m >= T[u] and bool(rng.integers(0, 2) or m == 0)
            chunks.append(feats[u][pos[u]:pos[u] + m]); fins.append(fin); pos[u] += m
        fo = np.concatenate([[0], np.cumsum([c.shape[0] for c in chunks])])
        got = AcceptFramesBatch([batch[u] for u in live], torch.cat(chunks) if fo[-1] else feats[0][:0], fo, fins)
        for i, u in enumerate(live):
            single[u].AcceptFrames(chunks[i], fins[i])
            assert torch.equal(got[i], single[u].Latest()) and batch[u].NumRows() == single[u].NumRows(), (call, u, pos[u], T[u])
            if fins[i]:
                done[u] = True; whole, _ = ex.GetIvectors(feats[u], [0, T[u]])
                assert batch[u].NumRows() == whole.shape[0] and torch.equal(got[i], whole[-1]), (call, u)
    assert second and all(done)


def test_argument_errors():
    from kaldi_amd.ivector import BatchedIvectorExtractor
    from kaldi_amd.lib import K3Error
    rng = np.random.default_rng(1); lda, st, ubm, ie = _random_model(rng, 8, 1, 1, 8, 4, 3, False)
    il = np.tril_indices(8); packed = np.stack([ie["sigma_inv"][g][il] for g in range(4)])
    mk = lambda **kw: BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, 0.0, **kw)
    with pytest.raises(K3Error): mk(left_context=2, right_context=2)                       # LDA columns do not match the splice width
    with pytest.raises(K3Error): mk(left_context=1, right_context=1, ivector_period=0)
    st0 = st.copy(); st0[0, 8] = 0.0
    with pytest.raises(K3Error, match="hold no frames"): BatchedIvectorExtractor.FromArrays(lda, st0, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, 0.0, left_context=1, right_context=1)      # OnlineCmvn: 'Global CMVN stats are required'
    ex = mk(left_context=1, right_context=1)
    x = torch.zeros((10, 8), device="cuda:0")
    with pytest.raises(K3Error): ex.GetIvectors(x, [0, 10, 10])                             # an utterance without frames


def test_adaptation_state_carried_to_the_speakers_next_utterance(golden):
    """spkA = utt0 utt1, spkB = utt2 utt3 (tests/golden/make_golden_ivector_adapt.py): the second utterance of a speaker starts from the CMVN and
    i-vector statistics of the first, and its i-vectors are far (0.6) from the fresh-state ones"""
    from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
    ref = np.load(os.path.join(DIR, "ivector_adapt_golden.npz"))
    ex = BatchedIvectorExtractor(OnlineIvectorExtractionInfo("ivector_extractor.conf")); dev = torch.device("cuda:0")
    first = [golden["feat_utt0"], golden["feat_utt2"]]; second = [golden["feat_utt1"], golden["feat_utt3"]]
    x = torch.from_numpy(np.concatenate(first)).to(dev); fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in first])])
    iv1, ro1, stats = ex.GetIvectors(x, fo, return_stats=True)
    cm = np.zeros((2, 2, 14))
    for k, f in enumerate(first): cm[k, 0, :13] = f.astype(np.float64).sum(0); cm[k, 0, 13] = f.shape[0]; cm[k, 1, :13] = (f.astype(np.float64) ** 2).sum(0)
    x2 = torch.from_numpy(np.concatenate(second)).to(dev); fo2 = np.concatenate([[0], np.cumsum([f.shape[0] for f in second])])
    iv2, ro2 = ex.GetIvectors(x2, fo2, cmvn_speaker_stats=cm, stats_in=stats); torch.cuda.synchronize()
    iv1, iv2 = iv1.cpu().numpy(), iv2.cpu().numpy(); worst = 0.0
    for k, u in enumerate(("utt0", "utt2")): worst = max(worst, np.abs(iv1[ro1[k]:ro1[k + 1]] - ref["iv_" + u]).max())
    for k, u in enumerate(("utt1", "utt3")): worst = max(worst, np.abs(iv2[ro2[k]:ro2[k + 1]] - ref["iv_" + u]).max())
    print("max |gpu - reference binary| with adaptation state =", worst)
    assert worst <= TOL, worst
    # --repeat=true of the reference program: the state handed on holds EVERY frame of the first utterance (k3_ivector_set_accumulate_tail)
    _, _, stats_all = ex.GetIvectors(x, fo, return_stats=True, accumulate_tail=True)
    iv2r, _ = ex.GetIvectors(x2, fo2, cmvn_speaker_stats=cm, stats_in=stats_all); torch.cuda.synchronize(); iv2r = iv2r.cpu().numpy(); worst_r = 0.0
    for k, u in enumerate(("utt1", "utt3")): worst_r = max(worst_r, np.abs(iv2r[ro2[k]:ro2[k + 1]] - ref["ivrep_" + u]).max())
    assert worst_r <= TOL and np.abs(ref["ivrep_utt1"] - ref["iv_utt1"]).max() > 0.01, worst_r


def test_frame_weights_on_the_statistics_equal_the_reference_binary(golden):
    """silence weighting: ivector-extract-online2 --frame-weights-rspecifier of the REFERENCE (tests/golden/make_golden_ivector_weighted.py: runs of 0/1, fractional weights incl.
    ones below min_post/0.99 and negative ones, a weight vector two frames short) -> GetIvectors(frame_weights=); all-ones weights are the unweighted call, bit for bit"""
    from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
    ref = np.load(os.path.join(DIR, "ivector_weighted_golden.npz")); plain = np.load(os.path.join(DIR, "ivector_adapt_golden.npz"))
    ex = BatchedIvectorExtractor(OnlineIvectorExtractionInfo("ivector_extractor.conf")); dev = torch.device("cuda:0")
    first = [golden["feat_utt0"], golden["feat_utt2"]]; second = [golden["feat_utt1"], golden["feat_utt3"]]
    pad = lambda w, f: np.concatenate([w, np.zeros(f.shape[0] - w.size, np.float32)])
    w1 = np.concatenate([pad(ref["w_utt0"], first[0]), pad(ref["w_utt2"], first[1])]); w2 = np.concatenate([pad(ref["w_utt1"], second[0]), pad(ref["w_utt3"], second[1])])
    x = torch.from_numpy(np.concatenate(first)).to(dev); fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in first])])
    x2 = torch.from_numpy(np.concatenate(second)).to(dev); fo2 = np.concatenate([[0], np.cumsum([f.shape[0] for f in second])])
    cm = np.zeros((2, 2, 14))
    for k, f in enumerate(first): cm[k, 0, :13] = f.astype(np.float64).sum(0); cm[k, 0, 13] = f.shape[0]; cm[k, 1, :13] = (f.astype(np.float64) ** 2).sum(0)
    for tag, tail in (("iv", False), ("ivrep", True)):
        iv1, ro1, stats = ex.GetIvectors(x, fo, return_stats=True, frame_weights=w1, accumulate_tail=tail)
        iv2, ro2 = ex.GetIvectors(x2, fo2, cmvn_speaker_stats=cm, stats_in=stats, frame_weights=torch.from_numpy(w2).to(dev)); torch.cuda.synchronize()
        a, b = iv1.cpu().numpy(), iv2.cpu().numpy(); worst = 0.0
        if tag == "iv":
            for k, u in enumerate(("utt0", "utt2")): worst = max(worst, np.abs(a[ro1[k]:ro1[k + 1]] - ref["iv_" + u]).max())
        for k, u in enumerate(("utt1", "utt3")): worst = max(worst, np.abs(b[ro2[k]:ro2[k + 1]] - ref[f"{tag}_" + u]).max())
        print(tag, "max |gpu - reference binary| with frame weights =", worst); assert worst <= TOL, (tag, worst)
    assert np.abs(ref["iv_utt0"] - plain["iv_utt0"]).max() > 0.1                                    # (the weights matter)
    ones, _ = ex.GetIvectors(x, fo, frame_weights=np.ones(int(fo[-1]), np.float32)); none, _ = ex.GetIvectors(x, fo); assert torch.equal(ones, none)
    with pytest.raises(ValueError): ex.GetIvectors(x, fo, frame_weights=np.ones(3, np.float32))

