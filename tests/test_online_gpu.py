"""GPU test of the streaming (chunked) path -- kaldi_amd/online.py: features, nnet3 forward and decoding fed chunk by chunk per channel
(the reference's BatchedThreadedNnet3CudaOnlinePipeline::DecodeBatch / OnlineBatchedFeaturePipelineCuda / BatchedStaticNnet3 /
CudaDecoder::AdvanceDecoding roles).  Bar: the chunked stream produces the SAME feature rows, the same log-likelihood rows and the same
raw lattices, bit for bit, as the offline whole-utterance batch path over the same C ABI."""
import numpy as np, pytest, torch
from kaldi_amd import synth
pytestmark = pytest.mark.gpu

def test_online_features_equal_offline():
    from kaldi_amd import feat, online
    dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
    opts = feat.fbank_options(dither=0.0, num_bins=40)
    lens = [16000, 399, 400, 23001, 7777]
    waves = [torch.from_numpy(synth.gaussian_pcm16(n, 70 + i).astype(np.float32)).to(dev) for i, n in enumerate(lens)]
    sf = feat.SpectralFeatures(opts); wo, fo, total, fo_h = sf.offsets(lens, dev)
    ref = sf.ComputeFeatures(torch.cat(waves), wo, fo, total)
    pipe = online.OnlineBatchedFeaturePipeline(opts, num_channels=len(lens))
    got = [[] for _ in lens]; pos = [0] * len(lens)
    while any(p < n for p, n in zip(pos, lens)):
        chans = [u for u in range(len(lens)) if pos[u] < lens[u]]
        chunks, first = [], []
        for u in chans:
            n = min(int(rng.integers(1, 4000)), lens[u] - pos[u]); chunks.append(waves[u][pos[u]:pos[u] + n]); first.append(pos[u] == 0); pos[u] += n
        for u, f in zip(chans, pipe.ComputeFeaturesBatched(chans, chunks, first)): got[u].append(f)
    for u in range(len(lens)):
        g = torch.cat(got[u]) if got[u] else torch.zeros((0, 40), device=dev)
        assert torch.equal(g, ref[fo_h[u]:fo_h[u + 1]]), u

def test_online_pipeline_lattices_equal_offline(tmp_path):
    from kaldi_amd import feat, nnet3, decoder, online
    dev = torch.device("cuda:0"); rng = np.random.default_rng(5); N = 120
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net_w = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    mp = str(tmp_path / "m.raw"); net_w.write(mp); nn = nnet3.Nnet(mp)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(graph, t2p)
    cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000)
    opts = feat.fbank_options(dither=0.0, num_bins=40)
    lens = [16000, 9000, 23001, 52000]
    waves = [torch.from_numpy(synth.gaussian_pcm16(n, 50 + i).astype(np.float32)).to(dev) for i, n in enumerate(lens)]
    # offline
    sf = feat.SpectralFeatures(opts); wo, fo, total, fo_h = sf.offsets(lens, dev)
    feats = sf.ComputeFeatures(torch.cat(waves), wo, fo, total)
    nb = nnet3.NnetBatch(nn, [fo_h[i + 1] - fo_h[i] for i in range(len(lens))], 3); ll = nb.forward(feats)
    dec = decoder.CudaDecoder(cf, cfg, len(lens), N); dec.DecodeBatch(ll, nb.out_offsets); ref = dec.GetRawLattices(copy=True)
    # streaming: 3 channels serve the 4 utterances (a channel is reused as soon as its utterance ended), two chunk regimes
    for C, max_samples in ((150, 30000), (30, 5000)):
        nch = 3
        pipe = online.BatchedOnlinePipeline(opts, nn, cf, cfg, num_channels=nch, max_frames_per_channel=400, frames_per_chunk=C, frame_subsampling_factor=3)
        pos = [0] * len(lens); queue = list(range(len(lens))); on = {}; got = {}
        while queue or on:
            for ch in range(nch):
                if ch not in on and queue: on[ch] = queue.pop(0)
            chans = [ch for ch in on if rng.random() < 0.8]
            if not chans: continue
            chunks, first, last = [], [], []
            for ch in chans:
                u = on[ch]; n = min(int(rng.integers(1, max_samples)), lens[u] - pos[u])
                chunks.append(waves[u][pos[u]:pos[u] + n]); first.append(pos[u] == 0); pos[u] += n; last.append(pos[u] == lens[u])
            for ch, lat in pipe.DecodeBatch(chans, chunks, first, last).items():
                u = on.pop(ch); got[u] = lat
                assert pipe.frames_decoded[ch] == int(nb.out_offsets[u + 1] - nb.out_offsets[u])
        assert sorted(got) == list(range(len(lens)))
        for u in range(len(lens)):
            d = got[u].diff(ref[u]); assert d == "", (C, u, d)
            assert got[u].num_arcs > 0


def test_dynamic_batcher_threads_pushing_chunks_get_the_offline_lattices(tmp_path):
    """CudaOnlinePipelineDynamicBatcher: 6 streams pushed chunk by chunk from 3 client threads into a 3-channel pipeline (so streams wait for a free
    channel in the backlog); every stream's lattice equals the whole-utterance decode, and the batcher really batched (more than one chunk per call)"""
    import threading, time
    from kaldi_amd import feat, nnet3, decoder, online
    dev = torch.device("cuda:0"); N = 120
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net_w = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5)
    mp = str(tmp_path / "m.raw"); net_w.write(mp); nn = nnet3.Nnet(mp)
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(graph, t2p)
    cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000); opts = feat.fbank_options(dither=0.0, num_bins=40)
    lens = [16000, 9000, 23001, 30000, 12345, 8000]
    waves = [synth.gaussian_pcm16(n, 50 + i).astype(np.float32) for i, n in enumerate(lens)]
    sf = feat.SpectralFeatures(opts); wo, fo, total, fo_h = sf.offsets(lens, dev)
    feats = sf.ComputeFeatures(torch.from_numpy(np.concatenate(waves)).to(dev), wo, fo, total)
    nb = nnet3.NnetBatch(nn, [fo_h[i + 1] - fo_h[i] for i in range(len(lens))], 3); ll = nb.forward(feats)
    dec = decoder.CudaDecoder(cf, cfg, len(lens), N); dec.DecodeBatch(ll, nb.out_offsets); ref = dec.GetRawLattices(copy=True)
    pipe = online.BatchedOnlinePipeline(opts, nn, cf, cfg, num_channels=3, max_frames_per_channel=400, frames_per_chunk=30, frame_subsampling_factor=3)
    got = {}; batcher = online.CudaOnlinePipelineDynamicBatcher(pipe, max_batch_size=3, dynamic_batcher_timeout=5e-3, lattice_callback=lambda cid, lat: got.__setitem__(cid, lat))
    def client(streams, seed):
        rng = np.random.default_rng(seed); pos = {u: 0 for u in streams}
        while pos:
            u = list(pos)[int(rng.integers(0, len(pos)))]; n = min(int(rng.integers(800, 6000)), lens[u] - pos[u])
            batcher.Push(1000 + u, pos[u] == 0, pos[u] + n == lens[u], waves[u][pos[u]:pos[u] + n]); pos[u] += n
            if pos[u] == lens[u]: del pos[u]
            time.sleep(float(rng.random()) * 1e-3)
    threads = [threading.Thread(target=client, args=([2 * k, 2 * k + 1], 7 + k)) for k in range(3)]
    for t in threads: t.start()
    for t in threads: t.join()
    batcher.WaitForCompletion()
    assert all(batcher.GetNumPendingChunks(1000 + u) == 0 for u in range(6)); batcher.Close()
    assert sorted(got) == [1000 + u for u in range(6)]
    for u in range(6): d = got[1000 + u].diff(ref[u]); assert d == "", (u, d)
    assert max(batcher.batch_sizes) >= 2 and sum(batcher.batch_sizes) > len(batcher.batch_sizes), batcher.batch_sizes[:20]


def _iv_setup(tmp_path, N=90):
    """a TDNN-F with the recipes' i-vector input (dim 20) and a random i-vector extractor over the same 40-dim features"""
    import importlib.util, os
    from kaldi_amd import nnet3
    from kaldi_amd.ivector import BatchedIvectorExtractor
    spec = importlib.util.spec_from_file_location("tiv", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_ivector_gpu.py")); tiv = importlib.util.module_from_spec(spec); spec.loader.exec_module(tiv)
    rng = np.random.default_rng(77); F, lc, rc, D, G, R = 40, 3, 3, 24, 32, 20
    lda, st, ubm, ie = tiv._random_model(rng, F, lc, rc, D, G, R, False)
    il = np.tril_indices(D); packed = np.stack([ie["sigma_inv"][g][il] for g in range(G)])
    ex = BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, ie["prior_offset"], left_context=lc, right_context=rc, ivector_period=10)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net_w = synth.make_tdnnf(seed=21, dim=64, bottleneck=16, strides=(1, 3, 0, 3), prefinal_small=32, num_pdfs=N, calib_feats=calib, ivector_dim=R, out_std=1.5)
    mp = str(tmp_path / "miv.raw"); net_w.write(mp)
    return nnet3.Nnet(mp), ex

def test_streaming_network_with_one_ivector_per_chunk(tmp_path):
    """BatchedStaticNnet3 on a model with an i-vector input: with the SAME i-vector on every chunk the chunked log-likelihoods are, bit for bit, the whole-utterance forward with
    that utterance-level i-vector (nnet3-compute --ivectors); with a different i-vector per chunk every chunk's rows are those of the whole-utterance forward with THAT i-vector"""
    from kaldi_amd import nnet3
    dev = torch.device("cuda:0"); nn, ex = _iv_setup(tmp_path); rng = np.random.default_rng(9); C = 60
    T = [250, 77]; feats = [torch.from_numpy((rng.standard_normal((t, 40)) * 1.2 + 16.5).astype(np.float32)).to(dev) for t in T]
    ivs = torch.from_numpy(rng.standard_normal((6, nn.info.ivector_dim)).astype(np.float32)).to(dev)
    def offline(u, v):
        b = nnet3.NnetBatch(nn, [T[u]], 3); return b.forward(feats[u], ivectors=v[None, :]).clone()
    whole = [[offline(u, ivs[k]) for k in range(6)] for u in range(2)]
    st = nnet3.BatchedStaticNnet3(nn, 2, 2, frames_per_chunk=C, frame_subsampling_factor=3)
    for same in (True, False):
        pos = [0, 0]; got = [[], []]; k = 0; rows = [0, 0]
        while any(p < t for p, t in zip(pos, T)):
            chs = [u for u in range(2) if pos[u] < T[u]]; chunks = [feats[u][pos[u]:pos[u] + C] for u in chs]
            first = [pos[u] == 0 for u in chs]; last = [pos[u] + C >= T[u] for u in chs]
            iv = torch.stack([ivs[0 if same else (k + u) % 6] for u in chs])
            outs = st.RunBatch(chs, chunks, first, last, ivectors=iv)
            for u, o in zip(chs, outs):
                want = whole[u][0 if same else (k + u) % 6][rows[u]:rows[u] + o.shape[0]]
                assert torch.equal(o, want), (same, u, k); rows[u] += o.shape[0]; pos[u] += C
            k += 1
        assert rows == [whole[0][0].shape[0], whole[1][0].shape[0]]
    with pytest.raises(ValueError, match="i-vector input"): st.RunBatch([0], [feats[0][:C]], [True], [False])

def test_streaming_pipeline_uses_the_extractors_latest_ivector(tmp_path):
    """kaldi_amd/online.py with an i-vector extractor: the i-vector a chunk is evaluated with is the row of the WHOLE utterance's extraction that the reference's online decodable
    would use (decodable-online-looped.cc:182-197: the estimate at the last multiple of the period among the frames ready = frames so far minus the splice's right context, zero before
    the first) -- although the extractor only ever saw the stream's prefix, chunk by chunk (IvectorStream: every frame processed once); and a model with an i-vector input decodes to the
    same lattice twice"""
    from kaldi_amd import feat, decoder, online
    dev = torch.device("cuda:0"); nn, ex = _iv_setup(tmp_path); N = nn.info.output_dim
    graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); cf = decoder.CudaFst(graph, synth.tid2pdf(N))
    cfg = decoder.decoder_config(beam=13.0, lattice_beam=6.0, max_active=5000, literal_order=1); opts = feat.fbank_options(dither=0.0, num_bins=40)
    pipe = online.BatchedOnlinePipeline(opts, nn, cf, cfg, 2, 400, frames_per_chunk=60, ivector_extractor=ex)
    wave = torch.from_numpy(synth.gaussian_pcm16(30000, 5).astype(np.float32)).to(dev)
    sf = feat.SpectralFeatures(opts); full = sf.ComputeFeatures(wave, *sf.offsets([30000], dev)[:3]); rows, _ = ex.GetIvectors(full, [0, full.shape[0]])
    from kaldi_amd.ivector import IvectorStream
    st = IvectorStream(ex); rng = np.random.default_rng(4)
    for n, fin in [(1, False), (3, False), (4, False), (13, False), (14, False), (57, False), (186, False), (186, True), (full.shape[0], True)]:
        st.Reset(); pos = 0
        while True:      # the prefix in chunks of random length; the last call carries `fin`
            m = min(int(rng.integers(1, 40)), n - pos); last = pos + m >= n; st.AcceptFrames(full[pos:pos + m], fin and last); pos += m
            if last: break
        ready = n - (0 if fin else ex.right_context)
        want = rows[(ready - 1) // ex.ivector_period] if ready > 0 else torch.zeros_like(rows[0])
        assert torch.equal(st.Latest(), want) and st.NumRows() == (max(ready, 0) + ex.ivector_period - 1) // ex.ivector_period, (n, fin)
    def decode():
        pos = 0; lat = None
        while pos < wave.numel():
            n = min(4321, wave.numel() - pos); r = pipe.DecodeBatch([1], [wave[pos:pos + n]], [pos == 0], [pos + n >= wave.numel()]); pos += n
            if r: lat = r[1]
        return lat
    a = decode(); assert pipe.ivs[1].NumRows() == rows.shape[0] and torch.equal(pipe.ivs[1].Latest(), rows[-1])      # the channel's extractor saw the whole stream, once
    b = decode()
    assert a is not None and a.num_arcs > 50 and a.diff(b) == ""
    with pytest.raises(ValueError, match="needs an ivector_extractor"): online.BatchedOnlinePipeline(opts, nn, cf, cfg, 2, 400)
