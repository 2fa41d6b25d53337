"""Synthetic workload generators: 16 kHz Gaussian PCM16 waveforms and random-init,
BatchNorm-calibrated TDNN / TDNN-F chain models written in Kaldi's own nnet3 BINARY model format
(Nnet::Write, nnet3/nnet-nnet.cc:630-662; component Write()s cited per writer below), so the same
file feeds the reference's nnet3-compute, the oracle and k3_nnet_load().

There is no network for real checkpoints; weights are seeded random with BatchNorm statistics set to
the empirical statistics of a calibration batch, which keeps activations O(1) like a trained model
(a fresh BatchNormComponent has count 0 and the reference would fabricate random stats,
nnet-normalize-component.cc:216-225)."""
import io, struct
import numpy as np

# ----------------------------------------------------------------------------- waves ----
def gaussian_pcm16(num_samples, seed, sigma=3000.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0.0, sigma, num_samples)), -32768, 32767).astype(np.int16)

# ----------------------------------------------------------------------------- binary writers ----
def _tok(f, s): f.write(s.encode() + b" ")
def _i32(f, v): f.write(b"\x04" + struct.pack("<i", int(v)))
def _f32(f, v): f.write(b"\x04" + struct.pack("<f", float(v)))
def _f64(f, v): f.write(b"\x08" + struct.pack("<d", float(v)))
def _bool(f, v): f.write(b"T" if v else b"F")
def _vec(f, v):
    v = np.ascontiguousarray(v, dtype="<f4"); f.write(b"FV \x04" + struct.pack("<i", v.size)); f.write(v.tobytes())
def _mat(f, m):
    m = np.ascontiguousarray(m, dtype="<f4"); f.write(b"FM \x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1])); f.write(m.tobytes())
def _ivec(f, v):
    v = np.asarray(v, dtype="<i4"); f.write(b"\x04" + struct.pack("<i", v.size) + v.tobytes())

def _w_affine(f, c):   # NaturalGradientAffineComponent::Write nnet-simple-component.cc:2948-2970
    _tok(f, "<NaturalGradientAffineComponent>"); _tok(f, "<MaxChange>"); _f32(f, 0.75); _tok(f, "<LearningRate>"); _f32(f, 0.001)
    _tok(f, "<LinearParams>"); _mat(f, c["W"]); _tok(f, "<BiasParams>"); _vec(f, c["b"])
    _tok(f, "<RankIn>"); _i32(f, 20); _tok(f, "<RankOut>"); _i32(f, 80); _tok(f, "<UpdatePeriod>"); _i32(f, 4)
    _tok(f, "<NumSamplesHistory>"); _f32(f, 2000.0); _tok(f, "<Alpha>"); _f32(f, 4.0); _tok(f, "</NaturalGradientAffineComponent>")

def _w_tdnn(f, c):     # TdnnComponent::Write nnet-tdnn-component.cc:382-408
    _tok(f, "<TdnnComponent>"); _tok(f, "<MaxChange>"); _f32(f, 0.75); _tok(f, "<LearningRate>"); _f32(f, 0.001)
    _tok(f, "<TimeOffsets>"); _ivec(f, c["offsets"]); _tok(f, "<LinearParams>"); _mat(f, c["W"]); _tok(f, "<BiasParams>"); _vec(f, c["b"])
    _tok(f, "<OrthonormalConstraint>"); _f32(f, c.get("orthonormal", 0.0)); _tok(f, "<UseNaturalGradient>"); _bool(f, True)
    _tok(f, "<NumSamplesHistory>"); _f32(f, 2000.0); _tok(f, "<AlphaInOut>"); _f32(f, 4.0); _f32(f, 4.0)
    _tok(f, "<RankInOut>"); _i32(f, 20); _i32(f, 80); _tok(f, "</TdnnComponent>")

def _w_linear(f, c):   # LinearComponent::Write nnet-simple-component.cc:3174-3201
    _tok(f, "<LinearComponent>"); _tok(f, "<MaxChange>"); _f32(f, 0.75); _tok(f, "<LearningRate>"); _f32(f, 0.001)
    _tok(f, "<Params>"); _mat(f, c["W"])
    if c.get("orthonormal", 0.0) != 0.0: _tok(f, "<OrthonormalConstraint>"); _f32(f, c["orthonormal"])      # (optional on read: nnet-simple-component.cc:3074-3078)
    _tok(f, "<UseNaturalGradient>"); _bool(f, True)
    _tok(f, "<RankInOut>"); _i32(f, 20); _i32(f, 80); _tok(f, "<Alpha>"); _f32(f, 4.0)
    _tok(f, "<NumSamplesHistory>"); _f32(f, 2000.0); _tok(f, "<UpdatePeriod>"); _i32(f, 4); _tok(f, "</LinearComponent>")

def _w_relu(f, c):     # NonlinearComponent::Write nnet-component-itf.cc:546-603
    d = c["dim"]; z = np.zeros(d, np.float32)
    _tok(f, "<RectifiedLinearComponent>"); _tok(f, "<Dim>"); _i32(f, d); _tok(f, "<ValueAvg>"); _vec(f, z); _tok(f, "<DerivAvg>"); _vec(f, z)
    _tok(f, "<Count>"); _f64(f, 0.0); _tok(f, "<OderivRms>"); _vec(f, z); _tok(f, "<OderivCount>"); _f64(f, 0.0)
    _tok(f, "<NumDimsSelfRepaired>"); _f64(f, 0.0); _tok(f, "<NumDimsProcessed>"); _f64(f, 0.0)
    _tok(f, "<SelfRepairScale>"); _f32(f, 1e-05); _tok(f, "</RectifiedLinearComponent>")

def _w_bn(f, c):       # BatchNormComponent::Write nnet-normalize-component.cc:616-645
    d = c["dim"]
    _tok(f, "<BatchNormComponent>"); _tok(f, "<Dim>"); _i32(f, d); _tok(f, "<BlockDim>"); _i32(f, d); _tok(f, "<Epsilon>"); _f32(f, 0.001)
    _tok(f, "<TargetRms>"); _f32(f, c.get("target_rms", 1.0)); _tok(f, "<TestMode>"); _bool(f, False); _tok(f, "<Count>"); _f64(f, c["count"])
    _tok(f, "<StatsMean>"); _vec(f, c["mean"]); _tok(f, "<StatsVar>"); _vec(f, c["var"]); _tok(f, "</BatchNormComponent>")

def _w_noop(f, c):     # NoOpComponent::Write nnet-simple-component.cc:480-487
    _tok(f, "<NoOpComponent>"); _tok(f, "<Dim>"); _i32(f, c["dim"]); _tok(f, "<BackpropScale>"); _f32(f, 1.0); _tok(f, "</NoOpComponent>")

_WRITERS = {"affine": _w_affine, "tdnn": _w_tdnn, "linear": _w_linear, "relu": _w_relu, "batchnorm": _w_bn, "noop": _w_noop}

class SynthNnet:
    """config_lines (exactly what xconfig_to_configs.py emits into final.config, minus 'component' lines)
    + components [(name, kind, params)]."""
    def __init__(self): self.config_lines, self.components = [], []

    def write(self, path, binary=True, as_mdl=False, num_pdfs=None, priors=None, left_context=0, right_context=0):
        """raw nnet3 (Nnet::Write) or, with as_mdl, final.mdl = TransitionModel + AmNnetSimple (am-nnet-simple.cc:34-45)"""
        assert binary
        with open(path, "wb") as f:
            f.write(b"\0B")
            if as_mdl: write_transition_model(f, num_pdfs)
            _tok(f, "<Nnet3>"); f.write(b"\n")
            for l in self.config_lines: f.write(l.encode() + b"\n")
            f.write(b"\n"); _tok(f, "<NumComponents>"); _i32(f, len(self.components))
            for name, kind, prm in self.components:
                _tok(f, "<ComponentName>"); _tok(f, name); _WRITERS[kind](f, prm)
            _tok(f, "</Nnet3>")
            if as_mdl:
                _tok(f, "<LeftContext>"); _i32(f, left_context); _tok(f, "<RightContext>"); _i32(f, right_context)
                _tok(f, "<Priors>"); _vec(f, np.zeros(0, np.float32) if priors is None else priors)

    def num_params(self):
        n = 0
        for _, kind, p in self.components:
            if kind in ("affine", "tdnn", "linear"): n += p["W"].size + (p["b"].size if "b" in p else 0)
        return n

# ----------------------------------------------------------------------------- model builders ----
def _calib_feats(rng, frames, dim):
    return (rng.standard_normal((frames, dim)) * 1.2 + 16.5).astype(np.float32)

def _splice(x, offsets):
    """x: [T x D] valid for t = 0..T-1 -> rows for t = -min..T-1-max, columns appended in offset order."""
    lo, hi = -min(offsets), x.shape[0] - 1 - max(offsets)
    return np.concatenate([x[lo + o: hi + o + 1] for o in offsets], axis=1)

# calibration activations are evaluated in float64 and only the statistics are rounded to float32, so that the generated model
# is bit-identical on every machine (float32 BLAS summation order differs between CPUs; the fixtures depend on the exact model)
def _bn_from(x):
    x = np.asarray(x, np.float64)
    return dict(dim=x.shape[1], count=float(x.shape[0]), mean=x.mean(0).astype(np.float32), var=x.var(0).astype(np.float32))

def _bn_apply(x, p, eps=1e-3):
    scale = (p["var"].astype(np.float64) + eps) ** -0.5 * p.get("target_rms", 1.0)
    return (np.asarray(x, np.float64) - p["mean"].astype(np.float64)) * scale

def make_tdnnf(seed=1, input_dim=40, dim=768, bottleneck=96, strides=(1, 1, 1, 0) + (3,) * 12, prefinal_small=192,
               num_pdfs=6024, bypass_scale=0.75, calib_frames=600, out_std=2.5, calib_feats=None, ivector_dim=0, orthonormal_constraint=0.0):
    """The 17-layer LibriSpeech TDNN-F layout (egs/librispeech/s5/local/chain/tuning/run_tdnn_1d.sh:220-249 minus
    ivector/LDA/xent/dropout) at mini_librispeech widths (run_tdnn_1k.sh:185-202): '17L-768/96-6024', ~6.28 M params.
    Node/component names follow steps/libs/nnet3/xconfig (composite_layers.py:68-227)."""
    rng = np.random.default_rng(seed)
    net = SynthNnet(); L = net.config_lines; C = net.components
    # calibration activations (shrinking as context is consumed); pass real features of the workload when available
    x = np.asarray(calib_feats if calib_feats is not None else _calib_feats(rng, calib_frames, input_dim), np.float64)
    def randn(r, c, std): return (rng.standard_normal((r, c)) * std).astype(np.float32)
    if ivector_dim: L.append(f"input-node name=ivector dim={ivector_dim}")      # ivector_dim > 0: the recipe's "input dim=100 name=ivector" and Append(-1,0,1,ReplaceIndex(ivector, t, 0)) (run_tdnn_1d.sh:222-228)
    L.append(f"input-node name=input dim={input_dim}")
    # tdnn1: relu-batchnorm-layer input=Append(-1,0,1)
    W = randn(dim, 3 * input_dim, 1.0 / np.sqrt(3 * input_dim)); b = (rng.standard_normal(dim) * 0.1).astype(np.float32)
    if ivector_dim:      # the i-vector columns come last, like the Append() order; drawn after W so that ivector_dim = 0 models keep their weights
        Wfull = np.concatenate([W, randn(dim, ivector_dim, 0.5 / np.sqrt(ivector_dim))], axis=1)
        C.append(("tdnn1.affine", "affine", dict(W=Wfull, b=b)))
        L.append("component-node name=tdnn1.affine component=tdnn1.affine input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))")
    else:
        C.append(("tdnn1.affine", "affine", dict(W=W, b=b)))
        L.append("component-node name=tdnn1.affine component=tdnn1.affine input=Append(Offset(input, -1), input, Offset(input, 1))")
    # centre the affine on the calibration input so the random first layer is not saturated by the fbank offset
    h = _splice(x, (-1, 0, 1)) @ W.T.astype(np.float64) + b
    b -= h.mean(0).astype(np.float32); h = _splice(x, (-1, 0, 1)) @ W.T.astype(np.float64) + b
    h = np.maximum(h, 0); C.append(("tdnn1.relu", "relu", dict(dim=dim))); L.append("component-node name=tdnn1.relu component=tdnn1.relu input=tdnn1.affine")
    bn = _bn_from(h); C.append(("tdnn1.batchnorm", "batchnorm", bn)); L.append("component-node name=tdnn1.batchnorm component=tdnn1.batchnorm input=tdnn1.relu")
    h = _bn_apply(h, bn); prev = "tdnn1.batchnorm"
    for li, s in enumerate(strides):
        n = f"tdnnf{li + 2}"
        o1 = (-s, 0) if s else (0,); o2 = (0, s) if s else (0,)
        W1 = randn(bottleneck, len(o1) * dim, 1.0 / np.sqrt(len(o1) * dim))
        W2 = randn(dim, len(o2) * bottleneck, 1.0 / np.sqrt(len(o2) * bottleneck)); b2 = (rng.standard_normal(dim) * 0.1).astype(np.float32)
        C.append((f"{n}.linear", "tdnn", dict(offsets=o1, W=W1, b=np.zeros(0, np.float32), orthonormal=orthonormal_constraint)))      # tdnnf-layer: orthonormal-constraint=-1 in the recipes (training only)
        L.append(f"component-node name={n}.linear component={n}.linear input={prev}")
        C.append((f"{n}.affine", "tdnn", dict(offsets=o2, W=W2, b=b2)))
        L.append(f"component-node name={n}.affine component={n}.affine input={n}.linear")
        z = _splice(h, o1) @ W1.T.astype(np.float64)                         # valid for t = s .. T-1
        y = np.maximum(_splice(z, o2) @ W2.T.astype(np.float64) + b2, 0)     # valid for t = s .. T-1-s  (relative to h's origin)
        C.append((f"{n}.relu", "relu", dict(dim=dim))); L.append(f"component-node name={n}.relu component={n}.relu input={n}.affine")
        bn = _bn_from(y); C.append((f"{n}.batchnorm", "batchnorm", bn)); L.append(f"component-node name={n}.batchnorm component={n}.batchnorm input={n}.relu")
        C.append((f"{n}.noop", "noop", dict(dim=dim)))
        L.append(f"component-node name={n}.noop component={n}.noop input=Sum(Scale({bypass_scale}, {prev}), {n}.batchnorm)")
        h = bypass_scale * h[s: h.shape[0] - s] + _bn_apply(y, bn)
        prev = f"{n}.noop"
    # prefinal-l (linear-component), prefinal-chain (prefinal-layer), output (output-layer include-log-softmax=false)
    Wl = randn(prefinal_small, dim, 1.0 / np.sqrt(dim)); C.append(("prefinal-l", "linear", dict(W=Wl, orthonormal=orthonormal_constraint)))
    L.append(f"component-node name=prefinal-l component=prefinal-l input={prev}")
    h = h @ Wl.T.astype(np.float64)
    Wa = randn(dim, prefinal_small, 1.0 / np.sqrt(prefinal_small)); ba = (rng.standard_normal(dim) * 0.1).astype(np.float32)
    C.append(("prefinal-chain.affine", "affine", dict(W=Wa, b=ba))); L.append("component-node name=prefinal-chain.affine component=prefinal-chain.affine input=prefinal-l")
    h = np.maximum(h @ Wa.T.astype(np.float64) + ba, 0)
    C.append(("prefinal-chain.relu", "relu", dict(dim=dim))); L.append("component-node name=prefinal-chain.relu component=prefinal-chain.relu input=prefinal-chain.affine")
    bn = _bn_from(h); C.append(("prefinal-chain.batchnorm1", "batchnorm", bn)); L.append("component-node name=prefinal-chain.batchnorm1 component=prefinal-chain.batchnorm1 input=prefinal-chain.relu")
    h = _bn_apply(h, bn)
    Wp = randn(prefinal_small, dim, 1.0 / np.sqrt(dim)); C.append(("prefinal-chain.linear", "linear", dict(W=Wp, orthonormal=orthonormal_constraint)))
    L.append("component-node name=prefinal-chain.linear component=prefinal-chain.linear input=prefinal-chain.batchnorm1")
    h = h @ Wp.T.astype(np.float64)
    bn = _bn_from(h); C.append(("prefinal-chain.batchnorm2", "batchnorm", bn)); L.append("component-node name=prefinal-chain.batchnorm2 component=prefinal-chain.batchnorm2 input=prefinal-chain.linear")
    h = _bn_apply(h, bn)
    Wo = randn(num_pdfs, prefinal_small, out_std / np.sqrt(prefinal_small)); bo = (rng.standard_normal(num_pdfs) * 0.5).astype(np.float32)
    C.append(("output.affine", "affine", dict(W=Wo, b=bo))); L.append("component-node name=output.affine component=output.affine input=prefinal-chain.batchnorm2")
    L.append("output-node name=output input=output.affine objective=linear")
    return net

def make_tdnn(seed=1, input_dim=40, dim=512, offsets=((-1, 0, 1), (-1, 0, 1), (-3, 0, 3)), num_pdfs=2000, calib_frames=400, out_std=2.5):
    """BASELINE config 1 '3x512' TDNN: 3 x [TdnnComponent -> ReLU -> BatchNorm] + AffineComponent."""
    rng = np.random.default_rng(seed)
    net = SynthNnet(); L = net.config_lines; C = net.components
    h = _calib_feats(rng, calib_frames, input_dim); L.append(f"input-node name=input dim={input_dim}")
    prev, d_in = "input", input_dim
    for i, offs in enumerate(offsets):
        n = f"tdnn{i + 1}"
        W = (rng.standard_normal((dim, len(offs) * d_in)) / np.sqrt(len(offs) * d_in)).astype(np.float32); b = (rng.standard_normal(dim) * 0.1).astype(np.float32)
        y = _splice(np.asarray(h, np.float64), offs) @ W.T.astype(np.float64) + b
        if i == 0: b -= y.mean(0).astype(np.float32); y = _splice(np.asarray(h, np.float64), offs) @ W.T.astype(np.float64) + b
        y = np.maximum(y, 0)
        C.append((f"{n}.affine", "tdnn", dict(offsets=offs, W=W, b=b))); L.append(f"component-node name={n}.affine component={n}.affine input={prev}")
        C.append((f"{n}.relu", "relu", dict(dim=dim))); L.append(f"component-node name={n}.relu component={n}.relu input={n}.affine")
        bn = _bn_from(y); C.append((f"{n}.batchnorm", "batchnorm", bn)); L.append(f"component-node name={n}.batchnorm component={n}.batchnorm input={n}.relu")
        h = _bn_apply(y, bn); prev, d_in = f"{n}.batchnorm", dim
    Wo = (rng.standard_normal((num_pdfs, dim)) * out_std / np.sqrt(dim)).astype(np.float32); bo = (rng.standard_normal(num_pdfs) * 0.5).astype(np.float32)
    C.append(("output.affine", "affine", dict(W=Wo, b=bo))); L.append(f"component-node name=output.affine component=output.affine input={prev}")
    L.append("output-node name=output input=output.affine objective=linear")
    return net

# ----------------------------------------------------------------------------- synthetic HCLG ----
def make_hclg(num_states=2_000_000, num_arcs=5_000_000, num_pdfs=6024, seed=4321, eps_frac=0.25, selfloop_frac=0.6,
              olabel_frac=0.08, final_frac=0.03, start_degree=4000, num_words=200_000, max_eps_depth=4):
    """A decodable graph with HCLG-like statistics: mean out-degree num_arcs/num_states with a heavy
    start/loop state, ~(1-eps_frac) emitting arcs with ilabel = transition-id in [1, 2*num_pdfs] (tid -> pdf is
    (tid-1) mod num_pdfs, see tid2pdf()), self-loops on selfloop_frac of the states, olabels on olabel_frac of the arcs,
    final_frac final states; every state is accessible and co-accessible.
    Input-epsilon arcs: only from a state of epsilon-level l to a state of level l+1 with a higher id, level(s) = s mod
    (max_eps_depth+1)  =>  no epsilon cycles (TopSortTokens asserts that: lattice-faster-decoder.cc:995) and epsilon chains of
    at most max_eps_depth arcs, like a real HCLG where they come from n-gram back-off and lexicon optional-silence arcs.
    Returns a kaldi_amd.fst.Fst."""
    from .fst import Fst
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order
    rng = np.random.default_rng(seed)
    S = int(num_states); assert S >= 4 * (max_eps_depth + 1)
    P = max_eps_depth + 1
    src, dst, eps = [], [], []
    # (b) self-loops (emitting)
    loops = np.nonzero(rng.random(S) < selfloop_frac)[0].astype(np.int64)
    src.append(loops); dst.append(loops); eps.append(np.zeros(loops.size, bool))
    # (c) the heavy start/loop state
    sd = int(min(start_degree, max(1, S // 2)))
    src.append(np.zeros(sd, np.int64)); dst.append(rng.integers(1, S, sd)); eps.append(np.zeros(sd, bool))
    # (a) accessibility: every state s >= 1 gets an in-arc from a random earlier state; for ~45 % of the states of level >= 1
    #     the parent is taken from the previous level and the arc is an epsilon arc
    s_all = np.arange(1, S, dtype=np.int64)
    par = (rng.random(S - 1) * s_all).astype(np.int64)
    sp_eps = (s_all % P != 0) & (rng.random(S - 1) < 0.45)
    cnt_b = (s_all - 1) // P + 1                                                     # earlier states of the previous level: s-1, s-1-P, ...
    par = np.where(sp_eps, s_all - 1 - P * (rng.random(S - 1) * cnt_b).astype(np.int64), par)
    sp_eps &= (par >= 0) & (par % P == (s_all % P) - 1)
    par = np.maximum(par, 0)
    src.append(par); dst.append(s_all); eps.append(sp_eps)
    # (e) every state without a child in (a) gets one emitting arc to a random state (keeps it co-accessible w.h.p.)
    has_child = np.zeros(S, bool); has_child[par] = True; has_child[0] = True
    lack = np.nonzero(~has_child)[0].astype(np.int64)
    src.append(lack); dst.append(rng.integers(0, S, lack.size)); eps.append(np.zeros(lack.size, bool))
    # (d) the rest: epsilon arcs level l -> level l+1 (higher id), then emitting arcs between random states
    n_rest = max(0, int(num_arcs) - (S - 1) - loops.size - sd - lack.size)
    n_eps = max(0, min(n_rest, int(round(eps_frac * num_arcs)) - int(sp_eps.sum())))
    es = rng.integers(0, S - 2 * P, n_eps); es -= (es % P == P - 1)                 # sources of level < max_eps_depth
    cnt = (S - 2 - es) // P + 1                                                      # states of the next level above es
    ed = es + 1 + P * (rng.random(n_eps) * cnt).astype(np.int64)
    ok = (ed > es) & (ed < S) & (ed % P == (es % P) + 1)
    src.append(es[ok]); dst.append(ed[ok]); eps.append(np.ones(int(ok.sum()), bool))
    n_emit = n_rest - int(ok.sum())
    src.append(rng.integers(0, S - 1, n_emit)); dst.append(rng.integers(0, S, n_emit)); eps.append(np.zeros(n_emit, bool))
    src, dst, eps = np.concatenate(src), np.concatenate(dst), np.concatenate(eps)
    final = np.full(S, np.inf, np.float32)
    fin = np.nonzero(rng.random(S) < final_frac)[0]
    if fin.size == 0: fin = np.array([S - 1])
    final[fin] = rng.uniform(0.0, 5.0, fin.size).astype(np.float32)
    # co-accessibility: reverse BFS from a super-final node; patch the states it misses with an arc to a final state
    rev = csr_matrix((np.ones(src.size + fin.size, np.int8), (np.concatenate([dst, np.full(fin.size, S)]), np.concatenate([src, fin]))), shape=(S + 1, S + 1))
    seen = np.zeros(S + 1, bool); seen[breadth_first_order(rev, S, directed=True, return_predecessors=False)] = True
    miss = np.nonzero(~seen[:S])[0]
    if miss.size:
        src = np.concatenate([src, miss]); dst = np.concatenate([dst, fin[rng.integers(0, fin.size, miss.size)]]); eps = np.concatenate([eps, np.zeros(miss.size, bool)])
    A = src.size
    ilabel = np.where(eps, 0, rng.integers(1, 2 * num_pdfs + 1, A)).astype(np.int32)
    has_word = rng.random(A) < olabel_frac
    olabel = np.where(has_word, rng.integers(1, num_words, A), 0).astype(np.int32)
    weight = np.where(has_word, rng.uniform(2.0, 12.0, A), rng.uniform(0.05, 2.5, A)).astype(np.float32)
    perm = rng.permutation(A)       # arcs of a state in random (but seeded) order, emitting and epsilon interleaved
    return Fst.from_arcs(S, 0, src[perm], ilabel[perm], olabel[perm], weight[perm], dst[perm], final)

def tid2pdf(num_pdfs):
    """the synthetic TransitionModel (write_transition_model): num_pdfs one-state phones, each with a self-loop and a forward
    transition sharing one pdf => transition-ids 2p-1, 2p -> pdf p-1, i.e. pdf = (tid - 1) // 2; index 0 unused."""
    t = np.arange(2 * num_pdfs + 1, dtype=np.int64)
    m = ((t - 1) // 2).astype(np.int32); m[0] = 0
    return m

def write_transition_model(f, num_pdfs):
    """TransitionModel::Write, binary (hmm/transition-model.cc:252-283; HmmTopology::Write hmm/hmm-topology.cc:160-215):
    one topology entry for phones 1..num_pdfs: state 0 <PdfClass 0> with transitions (0, 0.5) (1, 0.5), state 1 final;
    triples (phone p, hmm-state 0, pdf p-1); log-probs log(0.5)."""
    _tok(f, "<TransitionModel>"); _tok(f, "<Topology>")
    phones = np.arange(1, num_pdfs + 1, dtype="<i4"); phone2idx = np.zeros(num_pdfs + 1, "<i4"); phone2idx[0] = -1
    for v in (phones, phone2idx): f.write(b"\x04" + struct.pack("<i", v.size) + v.tobytes())
    _i32(f, 1)                       # one entry (HMM format: no -1 marker)
    _i32(f, 2)                       # two states
    _i32(f, 0); _i32(f, 2); _i32(f, 0); _f32(f, 0.5); _i32(f, 1); _f32(f, 0.5)     # state 0: pdf class 0, 2 transitions
    _i32(f, -1); _i32(f, 0)          # state 1: no pdf, no transitions
    _tok(f, "</Topology>"); _tok(f, "<Triples>"); _i32(f, num_pdfs)
    for p in range(1, num_pdfs + 1): _i32(f, p); _i32(f, 0); _i32(f, p - 1)
    _tok(f, "</Triples>"); _tok(f, "<LogProbs>"); _vec(f, np.concatenate([[0.0], np.full(2 * num_pdfs, np.log(0.5))]).astype(np.float32)); _tok(f, "</LogProbs>")
    _tok(f, "</TransitionModel>")


def make_den_fst(num_states=3000, num_pdfs=4000, seed=77, mean_degree=12.0, hub_frac=0.01, hub_degree=400):
    """A denominator graph with the shape chain-make-den-fst gives a phone language model compiled down to pdf-ids: an
    epsilon-free, recurrent acceptor whose labels are pdf-id + 1; out-degrees geometric around mean_degree with a few hub states (the states
    after silence / frequent phones) of ~hub_degree arcs, so in-degrees are skewed too; each state's outgoing probabilities sum to (1 - final
    probability) like a normalised LM.  Every state is reachable (state s > 0 has an in-arc from an earlier state)."""
    from .fst import Fst
    rng = np.random.default_rng(seed); S = int(num_states)
    deg = np.minimum(rng.geometric(1.0 / mean_degree, S), 8 * int(mean_degree)).astype(np.int64)
    hubs = rng.choice(S, max(1, int(hub_frac * S)), replace=False); deg[hubs] = np.minimum(hub_degree, S)
    src = np.repeat(np.arange(S), deg); A = src.size
    pop = rng.lognormal(0.0, 1.5, S); pop /= pop.sum()                          # popular destinations: skewed in-degree
    dst = rng.choice(S, A, p=pop)
    extra_src = (rng.random(S - 1) * np.arange(1, S)).astype(np.int64); extra_dst = np.arange(1, S)      # reachability
    src = np.concatenate([src, extra_src]); dst = np.concatenate([dst, extra_dst])
    order = np.argsort(src, kind="stable"); src, dst = src[order], dst[order]; A = src.size
    pdf = rng.integers(0, num_pdfs, A)
    raw = rng.gamma(0.7, 1.0, A) + 1e-3; tot = np.zeros(S); np.add.at(tot, src, raw)
    final_p = rng.uniform(0.01, 0.1, S)
    prob = raw / tot[src] * (1.0 - final_p[src])
    off = np.zeros(S + 1, np.int64); np.add.at(off, src + 1, 1); off = np.cumsum(off)
    return Fst(0, off, (pdf + 1).astype(np.int32), (pdf + 1).astype(np.int32), (-np.log(prob)).astype(np.float32), dst.astype(np.int32), (-np.log(final_p)).astype(np.float32))


def make_supervision_fst(frames, num_pdfs, seed, width=3, branch=2.0, pdf_pool=None):
    """One sequence's numerator FST in the shape chain-get-supervision gives it: an epsilon-free acceptor over pdf-id + 1, start
    state 0, states sorted by time, `width` alternative states per frame on average (the alignment's tolerance window), ~`branch` arcs per state
    to the next frame, small arc weights, final states (cost 0 or small) on the last level.  Every state is reachable and can reach a final state."""
    from .fst import Fst
    rng = np.random.default_rng(seed); T = int(frames)
    nper = [1] + [int(rng.integers(1, 2 * width)) for _ in range(T)]
    first = np.concatenate([[0], np.cumsum(nper)]); S = int(first[-1]); pool = np.arange(num_pdfs) if pdf_pool is None else np.asarray(pdf_pool)
    src, dst, lab, w = [], [], [], []
    for t in range(T):
        a, b = np.arange(first[t], first[t + 1]), np.arange(first[t + 1], first[t + 2])
        pdfs = rng.choice(pool, max(2, min(4, pool.size)), replace=False)      # a frame's alignment window allows a few pdfs
        for s_ in a:                                                            # every state goes somewhere
            for d in rng.choice(b, min(b.size, max(1, int(rng.poisson(branch)))), replace=False): src.append(s_); dst.append(d); lab.append(int(rng.choice(pdfs)) + 1); w.append(float(rng.choice([0.0, 0.0, rng.uniform(0, 2)])))
        reached = set(dst[-1 - k] for k in range(0)) ; reached = set(d for s_, d in zip(src, dst) if first[t] <= s_ < first[t + 1])
        for d in b:                                                             # every state is reached
            if d not in reached: src.append(int(rng.choice(a))); dst.append(d); lab.append(int(rng.choice(pdfs)) + 1); w.append(0.0)
    order = np.lexsort((np.arange(len(src)), np.array(src))); src, dst, lab, w = (np.asarray(x)[order] for x in (src, dst, lab, w))
    off = np.zeros(S + 1, np.int64); np.add.at(off, src + 1, 1); off = np.cumsum(off)
    fin = np.full(S, np.inf, np.float32); fin[first[T]:] = rng.choice([0.0, 0.5], S - first[T])
    return Fst(0, off, lab.astype(np.int32), lab.astype(np.int32), w.astype(np.float32), dst.astype(np.int32), fin)

def make_e2e_fst(frames, num_pdfs, seed, num_phones=None):
    """One sequence's END-TO-END numerator FST (chain-generic-numerator.h:30-60: what TrainingGraphToSupervision makes of a training graph): an epsilon-free acceptor over pdf-id + 1
    with self-loops and more than one final state -- a left-to-right chain of two-state phone HMMs (forward pdf, self-loop pdf per state; chain topology), an optional-silence
    branch at the start, a skip here and there.  Paths of exactly `frames` arcs exist (2 * num_phones <= frames)."""
    from .fst import Fst
    rng = np.random.default_rng(seed); K = int(num_phones if num_phones is not None else max(1, frames // 5)); assert 2 * K <= frames
    arcs = []      # (src, dst, pdf, weight)
    st = 0
    def hmm(s_in, s_out):      # two pdfs per phone: s_in -(fwd)-> mid [self-loop] -(.)-> s_out is how the chain topology expands; here mid has the self-loop and the exit
        nonlocal nxt
        f, l = int(rng.integers(0, num_pdfs)), int(rng.integers(0, num_pdfs)); mid = nxt; nxt += 1
        arcs.append((s_in, mid, f, float(rng.choice([0.0, 0.1, 0.69])))); arcs.append((mid, mid, l, 0.69)); arcs.append((mid, s_out, l, 0.69))
    nxt = 1; cur = 0
    for k in range(K):
        out = nxt; nxt += 1
        hmm(cur, out)
        if k == 0: hmm(cur, out)                                   # an alternative first phone (optional silence): two parallel branches
        elif rng.random() < 0.2: arcs.append((cur, out, int(rng.integers(0, num_pdfs)), 1.2))      # a one-frame alternative pronunciation
        cur = out
    S = nxt; fin = np.full(S, np.inf, np.float32); fin[cur] = 0.0
    arcs.append((cur, cur, int(rng.integers(0, num_pdfs)), 0.4))  # trailing silence self-loop on the final state
    if S > 3: fin[cur - 1] = 0.5 if np.isinf(fin[cur - 1]) else fin[cur - 1]      # a second final state
    arcs.sort(key=lambda a: a[0]); src = np.array([a[0] for a in arcs]); off = np.zeros(S + 1, np.int64); np.add.at(off, src + 1, 1); off = np.cumsum(off)
    lab = np.array([a[2] + 1 for a in arcs], np.int32)
    return Fst(0, off, lab, lab.copy(), np.array([a[3] for a in arcs], np.float32), np.array([a[1] for a in arcs], np.int32), fin)

def merge_supervision_fsts(fsts):
    """The merged FST chain::MergeSupervision builds for a minibatch (fst::Concat of the sequences + RmEpsilon, chain-supervision.cc:744-800), restated
    for the tests' reference run: the final states of sequence n take over copies of sequence n + 1's start arcs with their final cost added."""
    from .fst import Fst
    base = np.concatenate([[0], np.cumsum([f.num_states - 1 for f in fsts[1:]] and [fsts[0].num_states] + [f.num_states - 1 for f in fsts[1:]])])
    arcs = []; S = int(base[-1]); fin = np.full(S, np.inf, np.float32)
    gid = lambda n, s: int(s) if n == 0 else int(base[n] + s - 1)              # sequence n > 0 loses its start state
    for n, f in enumerate(fsts):
        for s in range(f.num_states):
            if n > 0 and s == 0: continue
            for a in range(int(f.arc_offsets[s]), int(f.arc_offsets[s + 1])): arcs.append((gid(n, s), int(f.ilabel[a]), float(f.weight[a]), gid(n, f.nextstate[a])))
            if np.isfinite(f.final[s]):
                if n + 1 < len(fsts):
                    g = fsts[n + 1]
                    for a in range(int(g.arc_offsets[0]), int(g.arc_offsets[1])): arcs.append((gid(n, s), int(g.ilabel[a]), float(np.float32(f.final[s]) + np.float32(g.weight[a])), gid(n + 1, g.nextstate[a])))
                else: fin[gid(n, s)] = f.final[s]
    arcs.sort(key=lambda x: x[0]); src = np.array([a[0] for a in arcs]); off = np.zeros(S + 1, np.int64); np.add.at(off, src + 1, 1); off = np.cumsum(off)
    lab = np.array([a[1] for a in arcs], np.int32)
    return Fst(0, off, lab, lab, np.array([a[2] for a in arcs], np.float32), np.array([a[3] for a in arcs], np.int32), fin)
