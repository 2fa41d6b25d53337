"""Full-size GPU checks on the BASELINE.json workload shape (configs[2]: 10 s utterances, 17L-768/96-6024 TDNN-F, 2.0 M-state /
5.0 M-arc HCLG, beam 15, lattice-beam 8, max-active 10000) with a 64-utterance batch:
  * oracle parity at full size on a sample of lanes: the GPU log-likelihoods of those lanes go through the restated
    LatticeFasterDecoder (order-independent mode) on the host and the raw lattices must be identical, cost bits included;
  * size-independent properties on every lane: the same utterance in different lanes gives the identical lattice (lane
    independence), a second run reproduces every lattice (determinism up to state numbering), the pruned raw lattice is already
    trim (fst::Connect is the identity), emitting arcs advance exactly one frame and epsilon arcs stay inside a frame, final
    states sit on the last frame only, and the best path's labels are the same as the literal reference algorithm's."""
import os, tempfile, numpy as np, pytest, torch
from kaldi_amd import synth
pytestmark = pytest.mark.gpu

def test_full_size_pipeline_oracle_parity_and_properties():
    from kaldi_amd import feat, nnet3, decoder
    from oracle import lattice_oracle as lo
    dev = torch.device("cuda:0"); U, nsamp = 64, 160000
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    w = (torch.randn(32 * nsamp, generator=g) * 3000).round().clamp(-32768, 32767)
    waves = torch.cat([w, w]).to(dev)                                     # utterance u + 32 is a copy of utterance u
    sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
    feats = sf.ComputeFeatures(waves, wo, fo, total)
    mp = os.path.join(tempfile.gettempdir(), "k3_fullsize.raw")
    synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
    net = nnet3.Nnet(mp); N = net.info.output_dim
    nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
    ll = nb.forward(feats)
    graph = synth.make_hclg(2_000_000, 5_000_000, N); t2p = synth.tid2pdf(N)
    cfg = dict(beam=15.0, lattice_beam=8.0, max_active=10000)
    dec = decoder.CudaDecoder(decoder.CudaFst(graph, t2p), decoder.decoder_config(frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, **cfg), U, N)
    dec.DecodeBatch(ll, nb.out_offsets); info = dec.LatticeInfo(); lats = dec.GetRawLattices()
    dec.DecodeBatch(ll, nb.out_offsets); lats2 = dec.GetRawLattices()
    assert (info[:, 2] == 0).all() and (info[:, 9] == 333).all()
    llh = ll.cpu().numpy()
    for u in (0, 17, 63):                                                # oracle parity at full size
        ref, oi = lo.decode(graph, llh[nb.out_offsets[u]:nb.out_offsets[u + 1]], t2p, lo.Config(**cfg), mode=1)
        assert lats[u].num_arcs > 300 and lats[u].diff(ref) == "", u
        assert np.array_equal(dec.FrameStats(u, 333)["ntoks"], oi["ntoks"])
        lit, _ = lo.decode(graph, llh[nb.out_offsets[u]:nb.out_offsets[u + 1]], t2p, lo.Config(**cfg), mode=0)
        bg, bl = lats[u].best_path(), lit.connect().best_path()
        assert bg[0] == bl[0] and bg[1] == bl[1]
    for u in range(U):
        L = lats[u]
        assert L.diff(lats2[u]) == ""                                     # reproducible
        if u >= 32: assert L.diff(lats[u - 32]) == ""                     # lane independent
        c = L.connect(); assert c.num_states == L.num_states and c.num_arcs == L.num_arcs      # already trim
        df = L.st_frame[L.arc_dst] - L.st_frame[L.arc_src]
        assert ((df == 1) == (L.arc_ilabel != 0)).all() and ((df == 0) == (L.arc_ilabel == 0)).all()
        fin = np.isfinite(L.st_final); assert fin.any() and (L.st_frame[fin] == 333).all()
        assert (L.st_state[L.arc_dst] >= 0).all() and L.start_index() >= 0
