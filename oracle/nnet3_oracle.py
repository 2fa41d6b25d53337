"""oracle/nnet3_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy float32) of the nnet3 forward pass for "simple" feed-forward models made of the
components a TDNN / TDNN-F chain model uses, evaluated the way nnet3-compute does it:
DecodableNnetSimple (nnet3/nnet-am-decodable-simple.cc:93-276): outputs at t = 0, s, 2s, ... with
ceil(T/s) rows (:45-47, :244-247), left/right context filled by REPLICATING the first/last input frame
(:154-163); for a feed-forward net chunking does not change any output value, so the oracle evaluates
whole utterances.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Component semantics restated (paths relative to /root/reference/src/nnet3):
  AffineComponent / NaturalGradientAffineComponent  nnet-simple-component.cc:1242-1251  y = x W^T + b
  FixedAffineComponent                               nnet-simple-component.cc:3392-3398
  LinearComponent                                    nnet-simple-component.cc:3224-3229  y = x P^T
  TdnnComponent                                      nnet-tdnn-component.cc:181-211      y = b + sum_i x[t+o_i] W_i^T
  RectifiedLinearComponent                           nnet-simple-component.cc:964-972
  BatchNormComponent (test mode)                     nnet-normalize-component.cc:209-247 (ComputeDerived), :453-463
  NoOpComponent / (General)DropoutComponent in test mode: identity (nnet-simple-component.cc:440-445,
                                                     nnet-general-component.cc:1564-1576)
  LogSoftmaxComponent / SoftmaxComponent             nnet-simple-component.cc:3618-3625, :3494-3504
  SigmoidComponent / TanhComponent                   nnet-simple-component.cc (Propagate = CuMatrixBase::Sigmoid / Tanh; matrix/kaldi-vector.cc:888-960)
  NormalizeComponent                                 nnet-normalize-component.cc:132-151 (cu::NormalizePerRow, cudamatrix/cu-math.cc:280-318)
Descriptors (nnet-descriptor.h): node | Offset(d, t) | Append(d...) | Sum(d, d) | Scale(a, d).

Pinned by tests/test_oracle_nnet.py against outputs of the reference's own nnet3-compute binary
(oracle/_ref, built by oracle/build_ref.sh) on models created by the reference's nnet3-init:
fixtures tests/golden/nnet_small.* (generator tests/golden/make_golden_nnet.py).
"""
import re
import numpy as np

# ----------------------------------------------------------------------------- text model IO ----
class Component:
    def __init__(self, name, ctype, fields, order):
        self.name, self.type, self.fields, self.order = name, ctype, fields, order

class Nnet:
    def __init__(self):
        self.config_lines = []      # raw lines: input-node / component-node / output-node / dim-range-node
        self.components = {}        # name -> Component
        self.comp_order = []
        self.prefix = ""            # anything before <Nnet3> (e.g. a TransitionModel), kept verbatim

def _tokenize(text):
    """tokens; a newline INSIDE [ ] is kept as '\\n' (it ends a matrix row in Kaldi text format)."""
    out, depth = [], 0
    for line in text.split("\n"):
        toks = line.split()
        for t in toks:
            if t == "[": depth += 1
            elif t == "]": depth -= 1
            out.append(t)
        if depth > 0 and toks and toks[-1] != "[":
            out.append("\n")
    return out

def _parse_array(tokens, i):
    assert tokens[i] == "["; i += 1
    rows, cur = [], []
    while tokens[i] != "]":
        if tokens[i] == "\n":
            if cur: rows.append(cur); cur = []
        else:
            cur.append(float(tokens[i]))
        i += 1
    if cur: rows.append(cur)
    i += 1
    if len(rows) == 0: return np.zeros((0,), np.float32), i
    if len(rows) == 1: return np.asarray(rows[0], np.float32), i
    return np.asarray(rows, np.float32), i

def _read_binary(buf):
    """Kaldi binary model (after the \\0B marker).  Semi-generic: after a <Tag>, values are \\x04/\\x08-prefixed
    basic types (kept as strings, ints when 4-byte values are tagged integer below), T/F bools, FV/FM/DV/DM
    arrays; <TimeOffsets> is a WriteIntegerVector (base/io-funcs-inl.h:198-211)."""
    import struct
    INT_TAGS = {"<Dim>", "<BlockDim>", "<RankIn>", "<RankOut>", "<UpdatePeriod>", "<RankInOut>", "<NumComponents>", "<InputDim>", "<OutputDim>"}
    pos = [0]
    def tok():
        j = buf.index(b" ", pos[0]); t = buf[pos[0]:j].decode(); pos[0] = j + 1; return t
    net = Nnet()
    if buf[:1] == b"<" and buf.startswith(b"<TransitionModel>"):
        raise ValueError("oracle: binary .mdl with TransitionModel not supported; pass the raw nnet")
    assert tok() == "<Nnet3>"
    nl = buf.index(b"\n", pos[0]); pos[0] = nl + 1
    while True:
        nl = buf.index(b"\n", pos[0]); line = buf[pos[0]:nl].decode(); pos[0] = nl + 1
        if line.strip() == "": break
        net.config_lines.append(line.strip())
    def value(tag):
        vals = []
        while True:
            c = buf[pos[0]:pos[0] + 1]
            if c == b"<": break
            if tag == "<TimeOffsets>":
                assert c == b"\x04"; n = struct.unpack("<i", buf[pos[0] + 1:pos[0] + 5])[0]
                a = np.frombuffer(buf, "<i4", n, pos[0] + 5).astype(np.float32); pos[0] += 5 + 4 * n; return a
            if c in (b"\x04", b"\x08"):
                sz = c[0]; raw = buf[pos[0] + 1:pos[0] + 1 + sz]; pos[0] += 1 + sz
                if sz == 8: vals.append(repr(struct.unpack("<d", raw)[0]))
                elif tag in INT_TAGS: vals.append(str(struct.unpack("<i", raw)[0]))
                else: vals.append(repr(struct.unpack("<f", raw)[0]))
            elif c in (b"T", b"F") and buf[pos[0] + 1:pos[0] + 2] in (b"<", b" "):
                vals.append(c.decode()); pos[0] += 1
                if buf[pos[0]:pos[0] + 1] == b" ": pos[0] += 1
            elif buf[pos[0]:pos[0] + 3] in (b"FV ", b"DV "):
                dt = "<f4" if c == b"F" else "<f8"; pos[0] += 3
                n = struct.unpack("<i", buf[pos[0] + 1:pos[0] + 5])[0]; pos[0] += 5
                a = np.frombuffer(buf, dt, n, pos[0]).astype(np.float32); pos[0] += n * int(dt[-1]); return a
            elif buf[pos[0]:pos[0] + 3] in (b"FM ", b"DM "):
                dt = "<f4" if c == b"F" else "<f8"; pos[0] += 3
                r = struct.unpack("<i", buf[pos[0] + 1:pos[0] + 5])[0]; cc = struct.unpack("<i", buf[pos[0] + 6:pos[0] + 10])[0]; pos[0] += 10
                a = np.frombuffer(buf, dt, r * cc, pos[0]).astype(np.float32).reshape(r, cc); pos[0] += r * cc * int(dt[-1]); return a
            else:
                raise ValueError(f"oracle: cannot parse binary value after {tag} at {pos[0]}: {buf[pos[0]:pos[0]+8]!r}")
        return vals
    assert tok() == "<NumComponents>"; n = int(value("<NumComponents>")[0])
    for _ in range(n):
        assert tok() == "<ComponentName>"; name = tok(); ctype = tok().strip("<>")
        fields, order = {}, []
        while True:
            tag = tok()
            if tag == f"</{ctype}>": break
            fields[tag] = value(tag); order.append(tag)
        net.components[name] = Component(name, ctype, fields, order); net.comp_order.append(name)
    assert tok() == "</Nnet3>"
    return net

def read_nnet(path):
    """text or binary, auto-detected by the \\0B marker (base/kaldi-io)."""
    with open(path, "rb") as f: head = f.read(2)
    if head == b"\0B":
        return _read_binary(open(path, "rb").read()[2:])
    return read_nnet_text(path)

def read_nnet_text(path):
    text = open(path).read()
    k = text.index("<Nnet3>")
    net = Nnet(); net.prefix = text[:k]
    body = text[k + len("<Nnet3>"):]
    lines = body.split("\n")
    li = 0
    while li < len(lines) and not lines[li].startswith("<NumComponents>"):
        if lines[li].strip(): net.config_lines.append(lines[li].strip())
        li += 1
    tokens = _tokenize("\n".join(lines[li:]))
    i = 0
    assert tokens[i] == "<NumComponents>"; n = int(tokens[i + 1]); i += 2
    for _ in range(n):
        assert tokens[i] == "<ComponentName>", tokens[i]
        name = tokens[i + 1]; ctype = tokens[i + 2].strip("<>"); i += 3
        fields, order = {}, []
        while tokens[i] != f"</{ctype}>":
            tag = tokens[i]; assert tag.startswith("<"), (name, tag); i += 1
            if tokens[i] == "[":
                val, i = _parse_array(tokens, i)
            else:
                vals = []
                while not tokens[i].startswith("<"):
                    vals.append(tokens[i]); i += 1
                val = vals
            fields[tag] = val; order.append(tag)
        i += 1
        net.components[name] = Component(name, ctype, fields, order); net.comp_order.append(name)
    assert tokens[i] == "</Nnet3>"
    return net

def _fmt_array(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return " [ " + " ".join(repr(float(x)) for x in a) + " ]\n"
    s = " [\n"
    for r in range(a.shape[0]):
        s += "  " + " ".join(repr(float(x)) for x in a[r]) + (" ]\n" if r == a.shape[0] - 1 else "\n")
    return s

def write_nnet_text(net, path):
    with open(path, "w") as f:
        f.write(net.prefix + "<Nnet3> \n")
        for l in net.config_lines: f.write(l + "\n")
        f.write("\n<NumComponents> %d \n" % len(net.comp_order))
        for name in net.comp_order:
            c = net.components[name]
            f.write(f"<ComponentName> {name} <{c.type}> ")
            for tag in c.order:
                v = c.fields[tag]
                if isinstance(v, np.ndarray) and tag == "<TimeOffsets>":     # ReadIntegerVector
                    f.write(tag + " [ " + " ".join(str(int(x)) for x in np.atleast_1d(v)) + " ]\n")
                elif isinstance(v, np.ndarray): f.write(tag + " " + _fmt_array(v))
                else: f.write(tag + " " + " ".join(v) + " ")
            f.write(f"</{c.type}> \n")
        f.write("</Nnet3> ")

def randomize_for_test(net, seed=0, output_scale=None):
    """Deterministic, well-conditioned test weights: explicit BatchNorm stats (a fresh BatchNormComponent has
    count 0 and test mode would fabricate random stats, nnet-normalize-component.cc:216-225) and a non-zero
    output layer (output-layer initialises to zero)."""
    rng = np.random.default_rng(seed)
    for name in net.comp_order:
        c = net.components[name]
        if c.type == "BatchNormComponent":
            dim = int(c.fields["<BlockDim>"][0])
            c.fields["<Count>"] = ["1000"]
            c.fields["<StatsMean>"] = rng.uniform(0.1, 0.6, dim).astype(np.float32)
            c.fields["<StatsVar>"] = rng.uniform(0.3, 1.2, dim).astype(np.float32)
            c.fields["<TestMode>"] = ["F"]
        elif c.type in ("NaturalGradientAffineComponent", "AffineComponent") and "<LinearParams>" in c.fields:
            w = c.fields["<LinearParams>"]
            if not np.any(w):
                sc = output_scale if output_scale is not None else 1.0 / np.sqrt(w.shape[1])
                c.fields["<LinearParams>"] = (rng.standard_normal(w.shape) * sc).astype(np.float32)
                c.fields["<BiasParams>"] = (rng.standard_normal(w.shape[0]) * 0.5).astype(np.float32)

# ----------------------------------------------------------------------------- descriptors ----
def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(": depth += 1
        if ch == ")": depth -= 1
        if ch == "," and depth == 0: out.append(cur.strip()); cur = ""
        else: cur += ch
    if cur.strip(): out.append(cur.strip())
    return out

def parse_descriptor(s):
    s = s.strip()
    m = re.match(r"^(\w+)\((.*)\)$", s)
    if not m: return ("node", s)
    fn, args = m.group(1), _split_args(m.group(2))
    if fn == "Offset": return ("offset", parse_descriptor(args[0]), int(args[1]))
    if fn == "Append": return ("append", [parse_descriptor(a) for a in args])
    if fn == "Sum": return ("sum", [parse_descriptor(a) for a in args])
    if fn == "Scale": return ("scale", float(args[0]), parse_descriptor(args[1]))
    if fn == "ReplaceIndex":       # ReplaceIndex(ivector, t, 0): the value of `ivector` at t = 0 for every t (nnet-descriptor.h:246-268)
        assert args[1] == "t" and int(args[2]) == 0, s
        return ("const", parse_descriptor(args[0]))
    raise ValueError("unsupported descriptor " + s)

def _cfg(line):
    kind, rest = line.split(None, 1)
    d = {}
    for m in re.finditer(r"(\w[\w-]*)=((?:[^\s(]+\(.*\)(?=\s+\w[\w-]*=|\s*$))|\S+)", rest):
        d[m.group(1)] = m.group(2)
    return kind, d

# ----------------------------------------------------------------------------- evaluation ----
_DTYPE = np.float32
def _F(): return _DTYPE
class _Seq:
    """values of one node for integer times t0 .. t0+len-1 (dense, step 1)."""
    def __init__(self, t0, x): self.t0, self.x = t0, x
    @property
    def t1(self): return self.t0 + self.x.shape[0] - 1

def _bn_scale_offset(c):
    f = c.fields
    count = np.float64(float(f["<Count>"][0])); eps = np.float32(float(f["<Epsilon>"][0])); rms = np.float32(float(f["<TargetRms>"][0]))
    mean32, var32 = f["<StatsMean>"].astype(np.float32), f["<StatsVar>"].astype(np.float32)
    # Read(): stats_sumsq = var + mean*mean (double vectors); then both scaled by count  (:605-610)
    ssum = mean32.astype(np.float64); ssq = var32.astype(np.float64) + ssum * ssum
    ssum = ssum * count; ssq = ssq * count
    assert count > 0, "BatchNorm test mode without stats"
    offset = ssum.astype(np.float32) * np.float32(-1.0 / count)          # -mean  (float vector ops from here)
    scale = ssq.astype(np.float32) * np.float32(1.0 / count)
    scale = scale + np.float32(-1.0) * offset * offset
    scale = np.maximum(scale, np.float32(0.0)) + eps
    scale = np.power(scale, np.float32(-0.5)).astype(np.float32) * rms
    offset = offset * scale
    dim, bdim = int(f["<Dim>"][0]), int(f["<BlockDim>"][0])
    return np.tile(scale, dim // bdim), np.tile(offset, dim // bdim)

def _apply_component(c, seq):
    """seq -> seq for components without time context; TdnnComponent handled here too."""
    x, f, t = seq.x, c.fields, c.type
    if t in ("NaturalGradientAffineComponent", "AffineComponent", "FixedAffineComponent"):
        return _Seq(seq.t0, (x @ f["<LinearParams>"].T + f["<BiasParams>"]).astype(_F()))
    if t == "LinearComponent":
        return _Seq(seq.t0, (x @ f["<Params>"].T).astype(_F()))
    if t == "TdnnComponent":
        offs = [int(v) for v in np.atleast_1d(f["<TimeOffsets>"])]
        W = f["<LinearParams>"]; D = W.shape[1] // len(offs)
        lo, hi = seq.t0 - min(offs), seq.t1 - max(offs)
        n = hi - lo + 1
        y = np.zeros((n, W.shape[0]), _F())
        if f["<BiasParams>"].size: y += f["<BiasParams>"]
        for i, o in enumerate(offs):
            s = lo + o - seq.t0
            y += x[s:s + n] @ W[:, i * D:(i + 1) * D].T
        return _Seq(lo, y.astype(_F()))
    if t == "RectifiedLinearComponent": return _Seq(seq.t0, np.maximum(x, _F()(0)))
    if t in ("NoOpComponent", "DropoutComponent", "GeneralDropoutComponent"): return seq
    if t == "BatchNormComponent":
        sc, of = _bn_scale_offset(c)
        return _Seq(seq.t0, (x * sc + of).astype(_F()))
    if t == "LogSoftmaxComponent":
        m = x.max(axis=1, keepdims=True)
        return _Seq(seq.t0, (x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))).astype(_F()))
    if t == "SoftmaxComponent":
        m = x.max(axis=1, keepdims=True); e = np.exp(x - m)
        return _Seq(seq.t0, (e / e.sum(axis=1, keepdims=True)).astype(_F()))
    if t == "SigmoidComponent":      # matrix/kaldi-vector.cc:938-960 (the MKL build evaluates the same function as 0.5 (tanh(x / 2) + 1))
        x64 = x.astype(np.float64); return _Seq(seq.t0, (1.0 / (1.0 + np.exp(-x64))).astype(_F()))
    if t == "TanhComponent":         # matrix/kaldi-vector.cc:888-916
        return _Seq(seq.t0, np.tanh(x.astype(np.float64)).astype(_F()))
    if t == "NormalizeComponent":    # nnet-normalize-component.cc:132-151 -> cu::NormalizePerRow, cudamatrix/cu-math.cc:303-317 (CPU branch)
        rms = _F()(float(f["<TargetRms>"][0])) if "<TargetRms>" in f else _F()(1.0)
        D = x.shape[1]; block = int(f["<BlockDim>"][0]) if "<BlockDim>" in f else D
        add_log = "<AddLogStddev>" in f and str(f["<AddLogStddev>"][0]) in ("T", "1", "1.0", "True")
        xb = x.reshape(-1, block)
        norm = (xb * xb).sum(axis=1, dtype=_F()) * _F()(1.0 / (block * float(rms) * float(rms)))
        norm = np.maximum(norm, _F()(2.0 ** -66)) ** _F()(-0.5)
        yb = (xb * norm[:, None]).astype(_F())
        if add_log: yb = np.concatenate([yb, (-np.log(norm) + np.log(rms))[:, None].astype(_F())], axis=1)
        return _Seq(seq.t0, yb.reshape(x.shape[0], -1))
    raise ValueError("oracle: unsupported component type " + t)

def _eval_desc(d, vals):
    k = d[0]
    if k == "node": return vals[d[1]]
    if k == "offset":
        s = _eval_desc(d[1], vals); return _Seq(s.t0 - d[2], s.x)      # value at t is src at t+off
    if k == "scale":
        s = _eval_desc(d[2], vals); return _Seq(s.t0, (_F()(d[1]) * s.x).astype(_F()))
    if k == "const": return ("const", _eval_desc(d[1], vals).x[0])
    parts = [_eval_desc(p, vals) for p in d[1]]
    timed = [p for p in parts if not isinstance(p, tuple)]
    lo, hi = max(p.t0 for p in timed), min(p.t1 for p in timed)
    cut = [np.tile(p[1][None, :], (hi - lo + 1, 1)) if isinstance(p, tuple) else p.x[lo - p.t0: hi - p.t0 + 1] for p in parts]
    if k == "append": return _Seq(lo, np.concatenate(cut, axis=1))
    if k == "sum":
        y = cut[0].copy()
        for c in cut[1:]: y = y + c
        return _Seq(lo, y.astype(_F()))
    raise ValueError(k)

def context(net):
    """(left, right) context of the 'output' node w.r.t. 'input' (ComputeSimpleNnetContext, nnet-utils.cc)."""
    ctx = {"input": (0, 0)}
    def dctx(d):
        k = d[0]
        if k == "const": return (0, 0)
        if k == "node": return ctx[d[1]]
        if k == "offset": l, r = dctx(d[1]); return (l - d[2], r + d[2])
        if k == "scale": return dctx(d[2])
        cs = [dctx(p) for p in d[1]]; return (max(c[0] for c in cs), max(c[1] for c in cs))
    for line in net.config_lines:
        kind, a = _cfg(line)
        if kind == "component-node":
            l, r = dctx(parse_descriptor(a["input"]))
            c = net.components[a["component"]]
            if c.type == "TdnnComponent":
                offs = [int(v) for v in np.atleast_1d(c.fields["<TimeOffsets>"])]
                l, r = l - min(offs), r + max(offs)
            ctx[a["name"]] = (l, r)
        elif kind == "output-node" and a["name"] == "output":
            return dctx(parse_descriptor(a["input"]))
    raise ValueError("no output node")

def compute(net, feats, frame_subsampling_factor=1, log_priors=None, acoustic_scale=1.0, dtype=np.float32, ivector=None, online_ivectors=None,
            online_ivector_period=10, frames_per_chunk=50):
    """nnet3-compute semantics for one utterance: feats [T x input_dim] -> [ceil(T/s) x output_dim].
    dtype=np.float64 evaluates the same graph in double precision (the value both float32 implementations --
    the reference's MKL sgemm path and the MFMA kernel -- are roundings of); default float32 like BaseFloat.
    ivector [N] (--ivectors: one per utterance) or online_ivectors [rows x N] (--online-ivectors, one row per online_ivector_period frames):
    the network is then evaluated chunk by chunk like DecodableNnetSimple (nnet-am-decodable-simple.cc:93-168), every chunk of
    frames_per_chunk frames with the i-vector GetCurrentIvector picks for it (:178-213)."""
    global _DTYPE
    _DTYPE = dtype
    try:
        if ivector is None and online_ivectors is None: return _compute(net, feats, frame_subsampling_factor, log_priors, acoustic_scale)
        s = frame_subsampling_factor; T = np.asarray(feats).shape[0]; n_sub = (T + s - 1) // s; per = (frames_per_chunk + s - 1) // s       # CheckAndFixConfigs rounds the chunk up to a multiple of s (nnet-am-decodable-simple.h:120-134)
        rows = []
        for c0 in range(0, n_sub, per):
            n = min(n_sub - c0, per); first, last = c0 * s, (c0 + n - 1) * s
            if ivector is not None: iv = np.asarray(ivector, np.float32)
            else:
                oi = np.asarray(online_ivectors, np.float32); f = (first + (last - first) // 2) // online_ivector_period
                if f >= oi.shape[0]:
                    if (f - (oi.shape[0] - 1)) * online_ivector_period > 50: raise ValueError("Could not get iVector for frame (mismatched --online-ivector-period?)")
                    f = oi.shape[0] - 1
                iv = oi[f]
            rows.append(_compute(net, feats, s, log_priors, acoustic_scale, first_out=first, num_out=n, ivector=iv))
        return np.concatenate(rows, axis=0)
    finally:
        _DTYPE = np.float32

def _compute(net, feats, frame_subsampling_factor, log_priors, acoustic_scale, first_out=0, num_out=None, ivector=None):
    """outputs for t = first_out + k*s, k < num_out (default: the whole utterance)"""
    feats = np.asarray(feats, np.float32).astype(_F()); T = feats.shape[0]; s = frame_subsampling_factor
    L, R = context(net)
    n_out = (T + s - 1) // s if num_out is None else num_out
    t_first = first_out; t_last = first_out + (n_out - 1) * s
    idx = np.clip(np.arange(t_first - L, t_last + R + 1), 0, T - 1)          # edge replication (:154-163)
    vals = {"input": _Seq(t_first - L, feats[idx])}
    if ivector is not None: vals["ivector"] = _Seq(0, np.asarray(ivector, np.float32).astype(_F())[None, :])
    out = None
    for line in net.config_lines:
        kind, a = _cfg(line)
        if kind == "input-node":
            assert a["name"] in ("input", "ivector"), a["name"]
            if a["name"] == "ivector": assert ivector is not None and np.asarray(ivector).size == int(a["dim"]), "the model expects an i-vector input"
        elif kind == "component-node":
            vals[a["name"]] = _apply_component(net.components[a["component"]], _eval_desc(parse_descriptor(a["input"]), vals))
        elif kind == "output-node" and a["name"] == "output":
            o = _eval_desc(parse_descriptor(a["input"]), vals)
            assert o.t0 <= t_first and o.t1 >= t_last, (o.t0, o.t1, t_first, t_last)
            out = o.x[(t_first - o.t0):(t_last - o.t0 + 1):s].copy()
    if log_priors is not None: out = out - np.asarray(log_priors, np.float32).astype(_F())      # :268-269
    if acoustic_scale != 1.0: out = out * _F()(acoustic_scale)               # :271
    return out.astype(_F())
