"""Streaming (chunked) use of the three stages, the role of the reference's online pipeline
(cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.{h,cc}: DecodeBatch(channels, wave chunks, is_first_chunk, is_last_chunk);
cudafeat/online-batched-feature-pipeline-cuda.h:83-92 ComputeFeaturesBatched; cudadecoder/batched-static-nnet3.cc RunBatch;
CudaDecoder::InitDecoding / AdvanceDecoding).  Every stage recomputes nothing it has already emitted and keeps only input-level state
per channel (trailing samples, context frames, the decoder lane), so the chunked results are bit-identical to the offline batch path
(tests/test_online_gpu.py).  Host-side orchestration only: all arithmetic runs in the same HIP kernels through the C ABI."""
import numpy as np, torch
from . import feat as _feat, nnet3 as _nnet3, decoder as _decoder


class OnlineBatchedFeaturePipeline:
    """ComputeFeaturesBatched over channels: a channel's samples that do not yet complete a frame (and the overlap the next frames
    reach back into) are kept on the GPU between calls.  snip_edges=true framing (feat/feature-window.cc:30-87): frame f covers
    samples [f * shift, f * shift + window), so the frames of a chunked stream are exactly those of the whole waveform."""
    def __init__(self, opts, num_channels, device="cuda:0"):
        if not opts.snip_edges: raise ValueError("streaming features need snip_edges=true (edge reflection depends on the end of the stream)")
        self.sf = _feat.SpectralFeatures(opts); self.dev = torch.device(device)
        self.shift = int(opts.samp_freq * 0.001 * opts.frame_shift_ms); self.win = int(opts.samp_freq * 0.001 * opts.frame_length_ms)
        self.stash = [torch.zeros(0, dtype=torch.float32, device=self.dev) for _ in range(num_channels)]
        self.frames_done = np.zeros(num_channels, np.int64)

    def Dim(self): return self.sf.dim

    def ComputeFeaturesBatched(self, channels, wave_chunks, is_first_chunk, is_last_chunk=None):
        """wave_chunks[i]: float32 samples (int16 range) of channels[i]; returns a list of [n_new_frames_i, dim] feature tensors"""
        waves = []
        for ch, w, first in zip(channels, wave_chunks, is_first_chunk):
            if first: self.stash[ch] = torch.zeros(0, dtype=torch.float32, device=self.dev); self.frames_done[ch] = 0
            waves.append(torch.cat([self.stash[ch], w.to(self.dev, torch.float32)]))
        lens = [int(w.numel()) for w in waves]
        wo, fo, total, fo_h = self.sf.offsets(lens, self.dev)
        out = self.sf.ComputeFeatures(torch.cat(waves) if waves else torch.zeros(0, device=self.dev), wo, fo, total) if total > 0 else \
            torch.zeros((0, self.sf.dim), dtype=torch.float32, device=self.dev)
        res = []
        for i, ch in enumerate(channels):
            n = fo_h[i + 1] - fo_h[i]
            res.append(out[fo_h[i]:fo_h[i + 1]])
            self.stash[ch] = waves[i][n * self.shift:]            # the next frame starts n * shift samples further
            self.frames_done[ch] += n
        return res


class OnlineEndpointRule:
    """online2/online-endpoint.h:84-97"""
    def __init__(self, must_contain_nonsilence=True, min_trailing_silence=1.0, max_relative_cost=float("inf"), min_utterance_length=0.0):
        self.must_contain_nonsilence, self.min_trailing_silence, self.max_relative_cost, self.min_utterance_length = must_contain_nonsilence, min_trailing_silence, max_relative_cost, min_utterance_length

class OnlineEndpointConfig:
    """online2/online-endpoint.h:134-153: the five rules with the reference's defaults; silence_phones = set of phone ids"""
    def __init__(self, silence_phones=()):
        inf = float("inf"); self.silence_phones = set(int(x) for x in silence_phones)
        self.rules = [OnlineEndpointRule(False, 5.0, inf, 0.0), OnlineEndpointRule(True, 0.5, 2.0, 0.0), OnlineEndpointRule(True, 1.0, 8.0, 0.0),
                      OnlineEndpointRule(True, 2.0, inf, 0.0), OnlineEndpointRule(False, 0.0, inf, 20.0)]

def EndpointDetected(config, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds, final_relative_cost):
    """kaldi::EndpointDetected (online2/online-endpoint.cc:26-72), in float32 like the reference"""
    f = np.float32
    assert num_frames_decoded >= trailing_silence_frames
    utt_len = f(num_frames_decoded) * f(frame_shift_in_seconds); sil = f(trailing_silence_frames) * f(frame_shift_in_seconds)
    for r in config.rules:
        contains_nonsilence = utt_len > sil
        if (contains_nonsilence or not r.must_contain_nonsilence) and sil >= f(r.min_trailing_silence) and f(final_relative_cost) <= f(r.max_relative_cost) and utt_len >= f(r.min_utterance_length):
            return True
    return False


class BatchedOnlinePipeline:
    """`num_channels` concurrent audio streams decoded chunk by chunk, the channel model of BatchedThreadedNnet3CudaOnlinePipeline:
    DecodeBatch(channels, wave_chunks, is_first_chunk, is_last_chunk) pushes the new audio of the listed channels through
    features -> nnet3 (BatchedStaticNnet3) -> CudaDecoder.AdvanceDecoding; a channel whose stream ended is finalised on the spot
    (lattice-beam pruning on the GPU) and its raw lattice is returned, the channel is free for the next utterance.  One decoder lane
    per channel (k3_decoder_init_channels / k3_decoder_finalize_channels); `max_frames_per_channel` bounds an utterance's length."""
    def __init__(self, feat_opts, nnet, cuda_fst, decoder_config, num_channels, max_frames_per_channel, frames_per_chunk=150,
                 frame_subsampling_factor=3, log_priors=None, acoustic_scale=1.0, device="cuda:0", ivector_extractor=None):
        """ivector_extractor (kaldi_amd.ivector.BatchedIvectorExtractor; models with an i-vector input): every chunk the network evaluates gets the extractor's latest i-vector for
        its channel, by the rule of the reference's online decodable (nnet3/decodable-online-looped.cc:182-197 over OnlineIvectorFeature, online2/online-ivector-feature.cc:
        use_most_recent_ivector): the estimate made at the last multiple of --ivector-period among the frames the extractor has seen -- all feature frames of the stream so far
        minus the LDA splice's right context while the stream goes on -- and zero before the first.  The estimates are rows of the whole-utterance extraction (k3_ivector_extract_batch:
        row k = statistics of frames 0 .. k * period), made by one kaldi_amd.ivector.IvectorStream per channel: the CMVN window, the splice context, the statistics and the solver's warm
        start are carried across calls, every frame is processed once."""
        self.nch = num_channels; self.dev = torch.device(device)
        self.iv = ivector_extractor
        if (nnet.info.ivector_dim > 0) != (ivector_extractor is not None): raise ValueError("a model with an i-vector input needs an ivector_extractor (and only such a model)")
        if self.iv is not None and self.iv.ivector_dim != nnet.info.ivector_dim: raise ValueError("Ivector feature dimension mismatch: got %d but network expects %d" % (self.iv.ivector_dim, nnet.info.ivector_dim))
        self.ivs = None
        if self.iv is not None:
            from .ivector import IvectorStream
            self.ivs = [IvectorStream(self.iv) for _ in range(num_channels)]      # per channel: the extractor's state of the stream
        self.features = OnlineBatchedFeaturePipeline(feat_opts, num_channels, device)
        self.nnet = _nnet3.BatchedStaticNnet3(nnet, num_channels, num_channels, frames_per_chunk, frame_subsampling_factor, log_priors, acoustic_scale, device)
        self.C = frames_per_chunk
        self.decoder = _decoder.CudaDecoder(cuda_fst, decoder_config, num_channels, nnet.info.output_dim)
        self.decoder.InitDecoding(num_channels, int(max_frames_per_channel))
        self.num_pdfs = nnet.info.output_dim
        self._empty_feats = torch.zeros((0, self.features.Dim()), dtype=torch.float32, device=self.dev)
        self.pending = [self._empty_feats for _ in range(num_channels)]
        self.started = np.zeros(num_channels, bool); self.frames_decoded = np.zeros(num_channels, np.int64)
        self.frame_shift_seconds = 0.001 * feat_opts.frame_shift_ms * frame_subsampling_factor      # SetOutputFrameShiftInSeconds

    def GetPartialHypothesis(self, channels):
        """CudaDecoder::GetPartialHypothesis (cuda-decoder.h:286): the words of the best path to the tokens each channel holds now (no final-probs)"""
        return [bp["olabels"] for bp in self.decoder.GetBestPath(channels, use_final_probs=False)]

    def EndpointDetected(self, channels, config, tid2phone):
        """CudaDecoder::EndpointDetected (cuda-decoder.h:293, cuda-decoder.cc:1921-1950): kaldi::EndpointDetected on the best-path traceback of each
        channel -- trailing frames whose transition-id belongs to a silence phone, frames decoded, FinalRelativeCost.  tid2phone: phone of a transition-id."""
        out = []
        for ch, bp in zip(channels, self.decoder.GetBestPath(channels, use_final_probs=False)):
            sil = 0
            for t in reversed(bp["ilabels"]):
                if int(tid2phone[t]) not in config.silence_phones: break
                sil += 1
            out.append(EndpointDetected(config, int(self.decoder.NumFramesDecoded(int(ch))), sil, self.frame_shift_seconds, bp["relative_cost"]))
        return out

    def DecodeBatch(self, channels, wave_chunks, is_first_chunk, is_last_chunk):
        """returns {channel: RawLattice} for the channels whose stream ended with this call"""
        fresh = [ch for ch, first in zip(channels, is_first_chunk) if first]
        if fresh:
            self.decoder.InitChannels(fresh)
            for ch in fresh: self.pending[ch] = self._empty_feats; self.started[ch] = False; self.frames_decoded[ch] = 0
            if self.ivs is not None:
                for ch in fresh: self.ivs[ch].Reset()
        feats = self.features.ComputeFeaturesBatched(channels, wave_chunks, is_first_chunk, is_last_chunk)
        ivec = None
        if self.iv is not None:      # the extractor sees every feature frame as soon as it exists (OnlineIvectorFeature over the same base features)
            from .ivector import AcceptFramesBatch
            fo = np.concatenate([[0], np.cumsum([int(f.shape[0]) for f in feats])]).astype(np.int64)
            rows = AcceptFramesBatch([self.ivs[ch] for ch in channels], torch.cat(list(feats)) if fo[-1] else self._empty_feats, fo, [bool(x) for x in is_last_chunk])      # one launch per stage for the batch
            ivec = {ch: rows[i].clone() for i, ch in enumerate(channels)}
        lls = {ch: [] for ch in channels}
        # feature frames go to the network at most frames_per_chunk at a time; what does not fill a chunk waits (unless the stream ends)
        todo = {ch: torch.cat([self.pending[ch], f]) for ch, f in zip(channels, feats)}
        last = dict(zip(channels, is_last_chunk))
        while True:
            # a channel runs when it has a full chunk, or when its stream ends (then whatever is left, possibly nothing, closes it)
            chs = [ch for ch in channels if todo[ch] is not None and (todo[ch].shape[0] >= self.C or last[ch])]
            if not chs: break
            chunks, firsts, lasts = [], [], []
            for ch in chs:
                n = min(self.C, todo[ch].shape[0]); chunks.append(todo[ch][:n]); rest = todo[ch][n:]
                firsts.append(not self.started[ch]); self.started[ch] = True
                is_end = bool(last[ch]) and rest.shape[0] == 0
                lasts.append(is_end); todo[ch] = None if is_end else rest
            outs = self.nnet.RunBatch(chs, chunks, firsts, lasts, None if ivec is None else torch.stack([ivec[ch] for ch in chs]))
            for ch, o in zip(chs, outs): lls[ch].append(o)
        for ch in channels: self.pending[ch] = todo[ch] if todo[ch] is not None else self._empty_feats
        # one AdvanceDecoding call for all lanes: lanes without new frames get an empty row range
        per_lane = [torch.cat(lls[ch]) if ch in lls and lls[ch] else None for ch in range(self.nch)]
        ro = np.zeros(self.nch + 1, np.int64)
        for ch in range(self.nch): ro[ch + 1] = ro[ch] + (per_lane[ch].shape[0] if per_lane[ch] is not None else 0)
        if ro[-1] > 0 or fresh:
            mat = torch.cat([x for x in per_lane if x is not None]) if ro[-1] > 0 else torch.zeros((1, self.num_pdfs), dtype=torch.float32, device=self.dev)
            self.decoder.AdvanceDecoding(mat, ro)
        for ch in channels: self.frames_decoded[ch] += ro[ch + 1] - ro[ch]
        ended = [ch for ch in channels if last[ch]]
        if not ended: return {}
        self.decoder.FinalizeChannels(ended)
        lats = self.decoder.GetRawLattices(copy=True)
        return {ch: lats[k] for k, ch in enumerate(ended)}


class CudaOnlinePipelineDynamicBatcher:
    """cudadecoder/cuda-online-pipeline-dynamic-batcher.{h,cc}: client threads Push() audio chunks of many streams (correlation ids); a batcher thread
    forms batches -- as soon as max_batch_size chunks wait, or every `dynamic_batcher_timeout` seconds -- and runs them through the pipeline's
    DecodeBatch.  Same rules as the reference: a batch holds at most one chunk per stream; chunks that do not fit (batch full, stream already in the
    batch, no free channel for a new stream: "All decoding channels are in use") wait in a FIFO backlog, which is drained first into the next batch
    (:73-124); Push deep-copies the samples.  A stream's channel is claimed at its first chunk (TryInitCorrID) and released when its last chunk was
    decoded; `lattice_callback(corr_id, RawLattice)` delivers the lattice then (the pipeline's SetLatticeCallback role)."""
    LOOP_TICK = 100e-6      # kDynamicBatcherLoopTick

    def __init__(self, pipeline, max_batch_size=None, dynamic_batcher_timeout=2e-3, lattice_callback=None):
        import threading, collections
        self.pipe = pipeline; self.max_batch = int(max_batch_size or pipeline.nch); self.timeout = float(dynamic_batcher_timeout); self.cb = lattice_callback
        self._lock = threading.Lock(); self._backlog = collections.deque(); self._next = []; self._next_ids = set()
        self._chan = {}; self._free = list(range(pipeline.nch)); self._pending = collections.Counter(); self._not_done = 0
        self._run = True; self._error = None; self.batch_sizes = []
        self._thread = threading.Thread(target=self._loop, daemon=True); self._thread.start()

    def _try_add(self, chunk):                       # TryAddChunkToNextBatchDeepCopy; the caller holds the lock
        corr_id, first, last, samples = chunk
        if len(self._next) >= self.max_batch or corr_id in self._next_ids: return False
        if first:
            if not self._free: return False          # all decoding channels are in use
            self._chan[corr_id] = self._free.pop(0)
        elif corr_id not in self._chan: raise KeyError(f"chunk of stream {corr_id} before its first chunk")
        self._next.append(chunk); self._next_ids.add(corr_id); return True

    def Push(self, corr_id, is_first_chunk, is_last_chunk, wave_samples):
        if self._error: raise self._error
        chunk = (corr_id, bool(is_first_chunk), bool(is_last_chunk), wave_samples.clone() if torch.is_tensor(wave_samples) else torch.from_numpy(np.array(wave_samples, np.float32)))
        with self._lock:
            # FIFO per stream: a chunk may not overtake an earlier chunk of its stream that is still in the backlog
            if any(c[0] == corr_id for c in self._backlog) or not self._try_add(chunk): self._backlog.append(chunk)
            self._pending[corr_id] += 1; self._not_done += 1

    def GetNumPendingChunks(self, corr_id):
        with self._lock: return self._pending.get(corr_id, 0)

    def WaitForCompletion(self):
        import time
        while True:
            if self._error: raise self._error
            with self._lock:
                if self._not_done == 0: return
            time.sleep(self.LOOP_TICK)

    def Close(self):
        self._run = False; self._thread.join()

    def _loop(self):
        import time
        next_timeout = time.monotonic() + self.timeout
        try:
            while self._run:
                if self._not_done >= self.max_batch or time.monotonic() >= next_timeout:
                    with self._lock:
                        cur, self._next, self._next_ids = self._next, [], set()
                        kept = type(self._backlog)(); blocked = set()      # FillNextBatchWithBacklog, FIFO; a stream's later chunks stay behind its first waiting one
                        for c in self._backlog:
                            if c[0] in blocked or not self._try_add(c): kept.append(c); blocked.add(c[0])
                        self._backlog = kept
                    if cur:
                        chans = [self._chan[c[0]] for c in cur]
                        lats = self.pipe.DecodeBatch(chans, [c[3].to(self.pipe.dev) for c in cur], [c[1] for c in cur], [c[2] for c in cur])
                        self.batch_sizes.append(len(cur))
                        done = []
                        with self._lock:
                            for c in cur:
                                self._pending[c[0]] -= 1; self._not_done -= 1
                                if c[2]: ch = self._chan.pop(c[0]); self._free.append(ch); done.append((c[0], lats[ch]))
                        if self.cb:
                            for corr_id, lat in done: self.cb(corr_id, lat)
                    next_timeout = time.monotonic() + self.timeout
                else: time.sleep(self.LOOP_TICK)
        except Exception as e:      # surfaces in Push / WaitForCompletion
            self._error = e
