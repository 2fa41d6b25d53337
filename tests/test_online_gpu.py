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
