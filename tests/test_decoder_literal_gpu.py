"""GPU parity of k3_decoder_config.literal_order: the HIP decoder must reproduce the reference's SERIAL LatticeFasterDecoder bit for bit --
the raw lattice of GetRawLattice (states, arcs, labels, graph / acoustic cost bits, sharing of states), the per-frame token counts and
cutoff bits, and the number of order-sensitive events.  Checked against
  * oracle/_ref/bin/ref-lattice-decoder = the reference's decoder/lattice-faster-decoder.cc compiled unmodified (live where oracle/_ref
    exists: it travels to the GPU box), and the digests recorded from it (tests/golden/decoder_ref_golden.json);
  * the restated oracle's literal mode (mode 0, itself pinned to that binary), which also gives the per-frame numbers.
The last test is the bench-configuration comparison VERDICT r1 asked for: the default (two-pass) mode and the literal mode against the
reference decoder on the TDNN-F's own log-likelihoods of 32 of the bench's utterances."""
import json, os, tempfile, numpy as np, pytest, torch
from concurrent.futures import ThreadPoolExecutor
from kaldi_amd import synth
from tests import decoder_cases as dcases, lattice_sig as lsig
pytestmark = pytest.mark.gpu
_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_ref_golden.json")))
_CAPS = dict(frame_tokens_cap=65536, frame_cands_cap=262144, lane_tokens_cap=2_500_000, lane_links_cap=3_500_000)

def _decode(cf, N, lls, literal=True, **cfg):
    from kaldi_amd import decoder
    kw = {k: v for k, v in cfg.items() if k in ("beam", "max_active", "min_active", "lattice_beam", "beam_delta", "hash_ratio", "fast_frame_tokens")}
    c = decoder.decoder_config(literal_order=int(literal), **dict(_CAPS, **kw))      # 1: component replay, 2: one-wavefront replay (the fall-back of 1)
    dec = decoder.CudaDecoder(cf, c, len(lls), N)
    ro = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls])])
    dec.DecodeBatch(torch.from_numpy(np.concatenate(lls)).cuda(), ro)
    info = dec.LatticeInfo()
    return dec.GetRawLattices(copy=True), info, dec

def _check_against_oracle(dec, u, lat, f, ll, t2p, kw):
    from oracle import lattice_oracle as lo
    ref, oi = lo.decode(f, ll, t2p, lo.Config(**kw), mode=0)
    st = dec.FrameStats(u)
    assert np.array_equal(st["ntoks"], oi["ntoks"]), (u, np.nonzero(st["ntoks"] != oi["ntoks"])[0][:5], st["ntoks"][:8], oi["ntoks"][:8])
    for k in ("cur_cutoff", "adaptive_beam", "next_cutoff", "cost_offset"):
        assert np.array_equal(st[k].view(np.int32), oi[k].view(np.int32)), (u, k)
    d = lat.diff(ref)
    assert d == "", (u, d)
    return oi

# replay 1: frames of <= 1536 tokens on the LDS-resident path (k3_decoder_fast.h), the others on the general path; 2 / 3: general path with the one-wavefront replay / with component stacks of
# length zero (every component that needs its stack falls back to the one-wavefront replay); (1, 0): general path only; (1, 96): an LDS path of 96 tokens -- most frames
# give up half-way and are redone, the two paths alternate all the time; (4, 0): general path with the large frames' hash-order passes on their HBM fall-back form
@pytest.mark.parametrize("replay,fast", [(1, -1), (2, -1), (3, -1), (1, 0), (1, 96), (4, 0)])
@pytest.mark.parametrize("name", sorted(dcases.CASES))
def test_literal_order_equals_the_reference_decoder(name, replay, fast):
    from kaldi_amd import decoder
    from oracle import ref_decoder as rd, lattice_oracle as lo
    f, t2p, ll, kw = dcases.make(name); N = ll.shape[1]
    cf = decoder.CudaFst(f, t2p)
    lats, info, dec = _decode(cf, N, [ll, ll[: max(1, ll.shape[0] // 2)]], literal=replay, fast_frame_tokens=fast, **kw)
    assert (info[:, 2] == 0).all(), info[:, 2]
    oi = _check_against_oracle(dec, 0, lats[0], f, ll, t2p, kw)
    _check_against_oracle(dec, 1, lats[1], f, ll[: max(1, ll.shape[0] // 2)], t2p, kw)
    assert int(dec.OrderSensitiveEvents()[0]) == oi["extra_links"]
    canon = lsig.canonical_of_raw(lats[0]); g = _GOLD[name]
    assert (lats[0].num_states, lats[0].num_arcs, bool(info[0, 3])) == (g["states"], g["arcs"], g["reached_final"])
    assert lsig.digest(canon) == g["digest"]                       # recorded from the reference binary
    if rd.available():                                             # and live
        assert lsig.canonical_of_reference(rd.decode(f, ll, t2p, lo.Config(**kw))) == canon

@pytest.mark.parametrize("replay,fast", [(1, -1), (2, -1), (3, -1), (1, 0), (1, 200)])
def test_literal_order_random_configurations_and_lane_reuse(replay, fast):
    """random graphs / lengths / spreads / every LatticeFasterDecoderConfig field incl. cost grids with exact ties; the same decoder object
    decodes every batch (scratch must return to its idle state), lanes hold different utterances"""
    from kaldi_amd import decoder
    rng = np.random.default_rng(4242)
    for it in range(6):
        N = int(rng.choice([20, 40, 80])); S = int(rng.choice([300, 1500, 6000, 30000])); A = int(S * rng.uniform(2.0, 3.5))
        f = synth.make_hclg(S, A, N, seed=int(rng.integers(0, 1 << 30)), start_degree=int(rng.choice([5, 30, 200]))); t2p = synth.tid2pdf(N)
        lls = [(rng.standard_normal((int(rng.integers(1, 70)), N)) * float(rng.choice([1.0, 2.5, 5.0]))).astype(np.float32) for _ in range(5)]
        if it % 3 == 0: lls = [np.round(x * 2) / 2 for x in lls]; f.weight[:] = np.round(f.weight * 4) / 4
        kw = dict(beam=float(rng.choice([4.0, 8.0, 15.0, 20.0])), lattice_beam=float(rng.choice([1.0, 4.0, 8.0, 12.0])), beam_delta=float(rng.choice([0.5, 0.1, 2.0])), hash_ratio=float(rng.choice([2.0, 1.0, 3.7])))
        if rng.random() < 0.5: kw["max_active"] = int(rng.choice([50, 200, 1000]))
        if rng.random() < 0.5: kw["min_active"] = int(rng.choice([0, 20, 500]))
        if "max_active" in kw and kw.get("min_active", 200) >= kw["max_active"]: kw["min_active"] = max(0, kw["max_active"] - 1)
        cf = decoder.CudaFst(f, t2p)
        lats, info, dec = _decode(cf, N, lls, literal=replay, fast_frame_tokens=fast, **kw)
        for u, ll in enumerate(lls):
            assert info[u, 2] in (0, 1), (it, u, info[u])
            if info[u, 2] == 0: _check_against_oracle(dec, u, lats[u], f, ll, t2p, kw)
        # second batch on the same decoder, utterances rotated over the lanes
        ro = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls[1:] + lls[:1]])])
        dec.DecodeBatch(torch.from_numpy(np.concatenate(lls[1:] + lls[:1])).cuda(), ro); info2 = dec.LatticeInfo(); lats2 = dec.GetRawLattices(copy=True)
        for u in range(len(lls)):
            if info2[u, 2] == 0: assert lats2[u].diff(lats[(u + 1) % len(lls)]) == "", (it, u)

@pytest.mark.parametrize("hash_ratio,replay", [(2.0, 1), (0.25, 1), (1.0, 4), (7.3, 1), (64.0, 1)])
def test_literal_order_large_frames_over_hash_ratios(hash_ratio, replay):
    """frames of 4 k - 20 k tokens (wide beam, no active-state limit, a graph whose start state fans out) with the reference's hash_ratio from a quarter (many tokens per bucket:
    the large-frame hash-order pass ranks long member lists) to 64 (bucket numbers far above 65535); replay 4 = the same frames through the HBM fall-back form of that pass"""
    from kaldi_amd import decoder
    N = 120; f = synth.make_hclg(60000, 200000, N, seed=77, start_degree=1500); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(f, t2p)
    rng = np.random.default_rng(11)
    lls = [(rng.standard_normal((T, N)) * 1.2).astype(np.float32) for T in (14, 9)]
    kw = dict(beam=17.0, lattice_beam=6.0, hash_ratio=hash_ratio)
    lats, info, dec = _decode(cf, N, lls, literal=replay, **kw)
    assert (info[:, 2] == 0).all(), info
    big = 0
    for u, ll in enumerate(lls):
        oi = _check_against_oracle(dec, u, lats[u], f, ll, t2p, kw); big = max(big, int(oi["ntoks"].max()))
    assert big > 4096, big      # (the case must reach the large-frame forms)

def test_literal_order_chunked_advance_equals_whole_utterance():
    from kaldi_amd import decoder
    N = 80; f = synth.make_hclg(3000, 8000, N, seed=9, start_degree=60); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(f, t2p)
    rng = np.random.default_rng(5)
    lls = [(rng.standard_normal((T, N)) * 2.5).astype(np.float32) for T in (120, 45, 77)]
    cfg = dict(beam=14.0, lattice_beam=7.0, max_active=3000)
    whole, _, _ = _decode(cf, N, lls, fast_frame_tokens=0, **cfg)      # (general path only; the chunked run below alternates between the paths)
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=1, fast_frame_tokens=400, **dict(_CAPS, **cfg)), 3, N)
    dec.InitDecoding(3, 130); done = [0, 0, 0]
    for chunk in ([50, 45, 0], [17, 0, 30], [53, 0, 47]):
        parts = [lls[u][done[u]:done[u] + c] for u, c in enumerate(chunk)]
        dec.AdvanceDecoding(torch.from_numpy(np.concatenate(parts)).cuda(), np.concatenate([[0], np.cumsum(chunk)]))
        done = [d + c for d, c in zip(done, chunk)]
    dec.FinalizeDecoding(); lats = dec.GetRawLattices()
    for u in range(3): assert lats[u].num_arcs > 0 and lats[u].diff(whole[u]) == "", u

def test_bench_configuration_against_the_reference_decoder(tmp_path):
    """BASELINE configs[2] as bench.py runs it (10 s utterances, 17L-768/96-6024 TDNN-F, 2.0 M-state / 5.0 M-arc HCLG, beam 15, lattice-beam 8,
    max-active 10000): 128 utterances (K3_PARITY_UTTS: up to the bench's 512), the reference's LatticeFasterDecoder (oracle/_ref/bin/ref-lattice-decoder) run on the GPU's own
    log-likelihoods.  literal_order: raw lattices identical to the reference's on every utterance (asserted).  Default (two-pass) mode: measured and
    reported, not asserted -- with this model's flat posteriors max-active binds on most frames, the tokens the serial code creates beyond
    the final bound get expanded on the next frame, and the two rules drift apart: round 2 measured the same best path on 25 of 32
    utterances only.  The numbers are written to gpurun_out/decoder_parity_bench_config.json (profiles/ keeps the copy of the round)."""
    from kaldi_amd import feat, nnet3, decoder
    from oracle import ref_decoder as rd, lattice_oracle as lo
    if not rd.available(): pytest.skip("oracle/_ref not built (needs /root/reference once; it travels to the GPU box)")
    dev = torch.device("cuda:0"); U, nsamp = int(os.environ.get("K3_PARITY_UTTS", "128")), 160000
    caps = dict(_CAPS, lane_tokens_cap=int(4500 * 10 * 33.4) + 65536, lane_links_cap=int(6000 * 10 * 33.4) + 131072, frame_cands_cap=131072)      # bench.py's sizing
    waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234 + i).astype(np.float32)) for i in range(U)]).to(dev)
    sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
    feats = sf.ComputeFeatures(waves, wo, fo, total)
    mp = str(tmp_path / "bench.raw"); synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
    net = nnet3.Nnet(mp); N = net.info.output_dim
    nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
    ll = nb.forward(feats); llh = ll.cpu().numpy()
    graph = synth.make_hclg(2_000_000, 5_000_000, N); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(graph, t2p)
    cfg = dict(beam=15.0, lattice_beam=8.0, max_active=10000)
    out = {}
    for literal in (1, 0):
        dec = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=literal, **dict(caps, **cfg)), U, N); dec.SetProfiling(True)
        dec.DecodeBatch(ll, nb.out_offsets); info = dec.LatticeInfo(); out[literal] = (dec.GetRawLattices(copy=True), info, dec.OrderSensitiveEvents(), dec.KernelTimes())
        assert (info[:, 2] == 0).all()
    def host_side(u):      # the reference's decoder and the restated oracle's literal mode on the GPU's log-likelihoods (subprocess / ctypes: both release the GIL)
        x = llh[nb.out_offsets[u]:nb.out_offsets[u + 1]]
        return rd.decode(graph, x, t2p, lo.Config(**cfg)), lo.decode(graph, x, t2p, lo.Config(**cfg), mode=0)
    with ThreadPoolExecutor(min(64, os.cpu_count() or 8)) as ex:
        host = list(ex.map(host_side, range(U)))
    refs = [h[0] for h in host]
    report = {"utterances": U, "literal_identical": 0, "default_best_path_identical": 0, "per_utt": []}
    for u in range(U):
        ref = refs[u]; rc = lsig.canonical_of_reference(ref)
        lit, dfl = out[1][0][u], out[0][0][u]
        same = lsig.canonical_of_raw(lit) == rc
        report["literal_identical"] += bool(same)
        # default mode: best path and arc sets vs the oracle's literal lattice (== the reference's, by the check above and the oracle pin)
        olit, oi = host[u][1]
        assert lsig.canonical_of_raw(olit) == rc, u
        bg, bl = dfl.connect().best_path(), olit.connect().best_path()
        best_same = bg[0] == bl[0] and bg[1] == bl[1] and np.float32(bg[2]) == np.float32(bl[2]) and np.float32(bg[3]) == np.float32(bl[3])
        report["default_best_path_identical"] += bool(best_same)
        a = set(map(tuple, dfl.canonical()[1].tolist())); b = set(map(tuple, olit.canonical()[1].tolist()))
        sym = len(a ^ b)
        report["per_utt"].append({"ref_arcs": int(ref["src"].size), "default_arcs": dfl.num_arcs, "literal_arcs": lit.num_arcs, "symmetric_difference": sym,
                                  "order_sensitive_events_literal": int(out[1][2][u]), "order_sensitive_upper_bound_default": int(out[0][2][u]), "oracle_order_sensitive": oi["order_sensitive_events"]})
        assert int(out[1][2][u]) == oi["extra_links"], u
    report["token_passing_ms"] = {"literal_order": out[1][3][0], "default": out[0][3][0]}; report["prune_ms"] = {"literal_order": out[1][3][1], "default": out[0][3][1]}
    report["max_symmetric_difference_frac"] = max(p["symmetric_difference"] / max(1, p["ref_arcs"]) for p in report["per_utt"])
    os.makedirs("gpurun_out", exist_ok=True); json.dump(report, open("gpurun_out/decoder_parity_bench_config.json", "w"), indent=1)
    assert report["literal_identical"] == U, report


@pytest.mark.parametrize("literal", [1, 0])
def test_lane_pools_are_a_reservation_and_grow_inside_the_kernel(literal):
    """k3_decoder_config::lane_tokens_cap / lane_links_cap (the reference's ntokens_pre_allocated) only RESERVE (cuda-decoder.cc:232-238): lanes that outgrow them move to
    bigger pools from the spare arena inside the token-passing kernel -- several times over a long utterance -- and the lattices are those of a decoder whose pools were big
    enough from the start, bit for bit (literal_order: also the reference decoder's); without a spare arena the reservation is the hard limit it used to be.  Whole-utterance
    and chunked (AdvanceDecoding) calls, lanes reused for a second batch."""
    from kaldi_amd import decoder
    N = 80; f = synth.make_hclg(3000, 8000, N, seed=9, start_degree=60); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(f, t2p)
    rng = np.random.default_rng(17)
    lls = [(rng.standard_normal((T, N)) * 2.5).astype(np.float32) for T in (700, 60, 333)]      # ~20 s, 2 s and 10 s at 3x subsampling
    cfg = dict(beam=14.0, lattice_beam=7.0, max_active=3000)
    big, info_big, dec_big = _decode(cf, N, lls, literal=literal, **cfg)
    assert (info_big[:, 2] == 0).all() and (dec_big.PoolGrowths() == 0).all()
    small = dict(_CAPS, frame_tokens_cap=8192, frame_cands_cap=32768, lane_tokens_cap=8192, lane_links_cap=40000, **cfg)      # the smallest legal reservation: one frame's worth
    need_t = int(info_big[:, 4].max()); assert need_t > 8 * small["lane_tokens_cap"], need_t      # the long utterance needs many times the reservation
    ro = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls])]); x = torch.from_numpy(np.concatenate(lls)).cuda()
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=literal, **small), len(lls), N)
    for rep in range(2):      # (the second batch starts on the pools the first one grew into)
        dec.DecodeBatch(x, ro); info = dec.LatticeInfo(); lats = dec.GetRawLattices(copy=True); g = dec.PoolGrowths()
        assert (info[:, 2] == 0).all(), info[:, 2]
        assert g[0] >= 3 and g[2] >= 2, g
        for u in range(len(lls)): assert lats[u].diff(big[u]) == "", (rep, u)
        if literal and rep == 0: _check_against_oracle(dec, 2, lats[2], f, lls[2], t2p, cfg)
    # chunked
    dec2 = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=literal, **small), 2, N)
    dec2.InitDecoding(2, 720); done = [0, 0]
    for chunk in ([300, 100], [1, 233], [399, 0]):
        parts = [lls[u][done[k]:done[k] + c] for k, (u, c) in enumerate(zip((0, 2), chunk))]
        dec2.AdvanceDecoding(torch.from_numpy(np.concatenate(parts)).cuda(), np.concatenate([[0], np.cumsum(chunk)])); done = [d + c for d, c in zip(done, chunk)]
    dec2.FinalizeDecoding(); l2 = dec2.GetRawLattices()
    assert l2[0].diff(big[0]) == "" and l2[1].diff(big[2]) == "" and (dec2.PoolGrowths() >= 2).all()
    # no spare arena: the reservation is a hard limit, reported per utterance
    dec3 = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=literal, spare_pool_bytes=0, **small), len(lls), N)
    dec3.DecodeBatch(x, ro); i3 = dec3.LatticeInfo(check=False)
    assert i3[0, 2] == -4 and i3[2, 2] == -4, i3[:, 2]      # K3_ERR_OVERFLOW
    # an arena that holds two growths of one lane: not enough for the long utterance
    dec4 = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=literal, spare_pool_bytes=(16 * 8192 + 20 * 40000) * 7, **small), 1, N)
    dec4.DecodeBatch(x[:700].contiguous(), np.array([0, 700])); i4 = dec4.LatticeInfo(check=False)
    assert i4[0, 2] == -4, i4[:, 2]

def test_init_decoding_and_first_frame_templates_and_longest_first_launch_change_nothing(monkeypatch):
    """The decoder works out once, when it is created, what InitDecoding leaves in a lane and -- when the tokens after InitDecoding do not exceed min_active, so that frame 0's
    adaptive beam is +inf -- the whole STRUCTURE of an utterance's first frame (tokens, links, closure sub-graph, components); two small kernels in front of the token-passing
    launch apply them (k3_decode_init_from_template_kernel, k3_decode_frame0_from_template_kernel) and the token-passing kernel resumes the lane at frame 1.  It also launches
    the lanes longest first.  None of this may change a bit: the same ragged batch (one-frame utterances included) decoded by decoders created with both templates, with the
    InitDecoding template only, and with neither (K3_LIT_NO_FRAME0_TEMPLATE / K3_LIT_NO_INIT_TEMPLATE: developer switches read by k3_decoder_create) gives identical lattices,
    per-frame token counts and cutoff bits, equal to the oracle's; so does a lane's second utterance, chunked AdvanceDecoding, and a configuration the first-frame
    template does not apply to (min_active below the start closure)."""
    from kaldi_amd import decoder
    rng = np.random.default_rng(99); N = 40
    f = synth.make_hclg(4000, 11000, N, seed=5, start_degree=60); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(f, t2p)
    lls = [(rng.standard_normal((T, N)) * 2.0).astype(np.float32) for T in (7, 61, 1, 33, 90, 18)]
    for kw in (dict(beam=13.0, lattice_beam=6.0, max_active=400, min_active=150), dict(beam=13.0, lattice_beam=6.0, max_active=400, min_active=3)):
        res = []
        for env in ((), ("K3_LIT_NO_FRAME0_TEMPLATE",), ("K3_LIT_NO_INIT_TEMPLATE",)):
            for k in ("K3_LIT_NO_FRAME0_TEMPLATE", "K3_LIT_NO_INIT_TEMPLATE"): monkeypatch.delenv(k, raising=False)
            for k in env: monkeypatch.setenv(k, "1")
            lats, info, dec = _decode(cf, N, lls, **kw)
            assert (info[:, 2] == 0).all(), info[:, 2]
            res.append((lats, [dec.FrameStats(u) for u in range(len(lls))]))
            if not env:
                for u, ll in enumerate(lls): _check_against_oracle(dec, u, lats[u], f, ll, t2p, kw)
                # the same decoder again, lanes rotated (a lane's second utterance starts from the templates as well)
                ro = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls[2:] + lls[:2]])])
                dec.DecodeBatch(torch.from_numpy(np.concatenate(lls[2:] + lls[:2])).cuda(), ro); l2 = dec.GetRawLattices(copy=True)
                for u in range(len(lls)): assert l2[u].diff(lats[(u + 2) % len(lls)]) == "", u
                # chunked: InitDecoding alone (no frames), then one frame, then the rest
                T = [l.shape[0] for l in lls]; dec.InitDecoding(len(lls), max(T))
                for lo, hi in ((0, 0), (0, 1), (1, None)):
                    part = [l[lo:hi] for l in lls]
                    dec.AdvanceDecoding(torch.from_numpy(np.concatenate(part + [np.zeros((1, N), np.float32)])).cuda(), np.concatenate([[0], np.cumsum([x.shape[0] for x in part])]))
                dec.FinalizeDecoding(); l3 = dec.GetRawLattices(copy=True)
                for u in range(len(lls)): assert l3[u].diff(lats[u]) == "", u
        for v in (1, 2):
            for u in range(len(lls)):
                assert res[0][0][u].diff(res[v][0][u]) == "", (v, u)
                for k in ("ntoks", "cur_cutoff", "adaptive_beam", "next_cutoff", "cost_offset"):
                    assert np.array_equal(np.asarray(res[0][1][u][k]).view(np.int32), np.asarray(res[v][1][u][k]).view(np.int32)), (v, u, k)
