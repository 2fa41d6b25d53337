"""Random state-level lattices and brute-force word-sequence enumeration for the determinizer tests (CPU only).
Text / binary layouts follow lat/kaldi-lattice.cc (FstPrinter text; OpenFst vector container with arc type lattice4 /
compactlattice44)."""
import struct
import numpy as np


def random_lattice(seed, frames=6, width=4, words=3, tids=9, p_word=0.35, eps_arcs=True, quant=None):
    """layered DAG like a decoder's raw lattice: states (frame, k); arcs frame f -> f+1 carrying a transition-id and mostly no word,
    plus a few non-emitting arcs inside a frame (tid 0) that may carry a word.  Returns dict(start, finals {s: (g, a)}, arcs [(s, d, tid, word, g, a)])."""
    rng = np.random.default_rng(seed)
    q = (lambda x: float(np.float32(np.round(x * quant) / quant))) if quant else (lambda x: float(np.float32(x)))
    ids = {}
    def sid(f, k): return ids.setdefault((f, k), len(ids))
    arcs = []; start = sid(0, 0)
    layer = [[0]] + [list(range(int(rng.integers(1, width + 1)))) for _ in range(frames)]
    for f in range(frames):
        for k in layer[f]:
            outs = rng.choice(layer[f + 1], size=min(len(layer[f + 1]), int(rng.integers(1, 4))), replace=False)
            for d in outs:
                for _ in range(int(rng.integers(1, 3))):          # parallel arcs: same word, different alignment
                    w = int(rng.integers(1, words + 1)) if rng.random() < p_word else 0
                    arcs.append((sid(f, k), sid(f + 1, int(d)), int(rng.integers(1, tids + 1)), w, q(rng.uniform(0, 3)), q(rng.uniform(0, 6))))
        if eps_arcs and len(layer[f + 1]) > 1 and rng.random() < 0.7:   # non-emitting arc k -> k' (k < k': acyclic)
            a, b = sorted(rng.choice(layer[f + 1], size=2, replace=False))
            w = int(rng.integers(1, words + 1)) if rng.random() < 0.5 else 0
            arcs.append((sid(f + 1, int(a)), sid(f + 1, int(b)), 0, w, q(rng.uniform(0, 2)), 0.0))
    finals = {sid(frames, k): (q(rng.uniform(0, 2)), 0.0) for k in layer[frames] if rng.random() < 0.8 or k == layer[frames][0]}
    # every state needs a way in from the start for the layout to be "connected enough"; unreachable ones are trimmed by the program
    return dict(start=start, n=len(ids), finals=finals, arcs=arcs)


def _w(g, a): return "%s,%s" % (repr(float(np.float32(g))), repr(float(np.float32(a))))


def lattice_text(key, lat):
    """FstPrinter layout: the start state's lines first."""
    by_src = {}
    for a in lat["arcs"]: by_src.setdefault(a[0], []).append(a)
    lines = [key + " "]
    order = [lat["start"]] + [s for s in range(lat["n"]) if s != lat["start"]]
    for s in order:
        for (_, d, tid, w, g, ac) in by_src.get(s, []):
            lines.append("%d\t%d\t%d\t%d" % (s, d, tid, w) + ("" if g == 0 and ac == 0 else "\t" + _w(g, ac)))
        if s in lat["finals"]:
            g, ac = lat["finals"][s]
            lines.append("%d" % s + ("" if g == 0 and ac == 0 else "\t" + _w(g, ac)))
    return "\n".join(lines) + "\n\n"


def lattice_binary(key, lat):
    by_src = {}
    for a in lat["arcs"]: by_src.setdefault(a[0], []).append(a)
    def s_(x): return struct.pack("<i", len(x)) + x
    o = key.encode() + b" " + struct.pack("<i", 2125659606) + s_(b"vector") + s_(b"lattice4") + struct.pack("<iiQqqq", 2, 0, 3, lat["start"], lat["n"], len(lat["arcs"]))
    inf = float("inf")
    for s in range(lat["n"]):
        g, ac = lat["finals"].get(s, (inf, inf))
        o += struct.pack("<ffq", g, ac, len(by_src.get(s, [])))
        for (_, d, tid, w, g, ac) in by_src.get(s, []): o += struct.pack("<iiffi", tid, w, g, ac, d)
    return o


def enumerate_raw(lat, acoustic_scale=1.0, limit=200000):
    """{word sequence: [(cost, graph, acoustic(scaled), tids)] of every path}; float64 sums of the float32 arc costs."""
    by_src = {}
    for a in lat["arcs"]: by_src.setdefault(a[0], []).append(a)
    out = {}; n = [0]
    def rec(s, words, tids, g, ac):
        if s in lat["finals"]:
            fg, fa = lat["finals"][s]
            out.setdefault(words, []).append((g + fg + (ac + fa * acoustic_scale), g + fg, ac + fa * acoustic_scale, tids))
            n[0] += 1
            assert n[0] < limit, "lattice too large to enumerate"
        for (_, d, tid, w, ag, aa) in by_src.get(s, []):
            rec(d, words + ((w,) if w else ()), tids + ((tid,) if tid else ()), g + ag, ac + float(np.float32(aa * np.float32(acoustic_scale))))
    rec(lat["start"], (), (), 0.0, 0.0)
    return out


def parse_compact_text(text):
    """{key: dict(start, finals {s: (g, a, tids)}, arcs [(s, d, word, g, a, tids)])} from an ark,t CompactLattice table."""
    res = {}; cur = None; first = True
    def weight(t):
        g, a, s = t.split(",")
        return float(g), float(a), tuple(int(x) for x in s.split("_")) if s else ()
    lines = text.split("\n"); i = 0
    while i < len(lines):
        ln = lines[i]; i += 1
        if cur is None:
            if not ln.strip(): continue
            key = ln.strip(); cur = dict(start=-1, finals={}, arcs=[]); res[key] = cur; first = True; continue
        if not ln.strip(): cur = None; continue
        c = ln.split("\t"); s = int(c[0])
        if first: cur["start"] = s; first = False
        if len(c) <= 2: cur["finals"][s] = weight(c[1]) if len(c) == 2 else (0.0, 0.0, ())
        else: cur["arcs"].append((s, int(c[1]), int(c[2])) + (weight(c[3]) if len(c) == 4 else (0.0, 0.0, ())))
    return res


def parse_compact_binary(buf):
    """same structure from a binary (ark:) CompactLattice table."""
    res = {}; p = 0
    def get(fmt):
        nonlocal p
        v = struct.unpack_from("<" + fmt, buf, p); p += struct.calcsize("<" + fmt); return v if len(v) > 1 else v[0]
    def s_():
        nonlocal p
        n = get("i"); v = buf[p:p + n]; p += n; return v
    def weight():
        g, a, n = get("ffi"); tids = tuple(get("i") for _ in range(n)); return g, a, tids
    while p < len(buf):
        e = buf.index(b" ", p); key = buf[p:e].decode(); p = e + 1
        assert get("i") == 2125659606
        assert s_() == b"vector" and s_() == b"compactlattice44"
        _, _, _, start, ns, na = get("iiQqqq")
        cur = dict(start=start, finals={}, arcs=[]); res[key] = cur; seen = 0
        for s in range(ns):
            g, a, tids = weight()
            if g != float("inf"): cur["finals"][s] = (g, a, tids)
            for _ in range(get("q")):
                il, ol = get("ii"); assert il == ol
                g, a, tids = weight(); d = get("i"); cur["arcs"].append((s, d, il, g, a, tids)); seen += 1
        assert seen == na
    return res


def enumerate_compact(c, acoustic_scale=1.0, limit=200000):
    """{word sequence: [(cost, graph, acoustic(scaled), tids)]} over all paths of a parsed compact lattice."""
    by_src = {}
    for a in c["arcs"]: by_src.setdefault(a[0], []).append(a)
    out = {}; n = [0]
    def rec(s, words, tids, g, ac):
        if s in c["finals"]:
            fg, fa, ft = c["finals"][s]
            out.setdefault(words, []).append((g + fg + (ac + fa * acoustic_scale), g + fg, ac + fa * acoustic_scale, tids + ft)); n[0] += 1
            assert n[0] < limit
        for (_, d, w, ag, aa, at) in by_src.get(s, []): rec(d, words + ((w,) if w else ()), tids + at, g + ag, ac + aa * acoustic_scale)
    if c["start"] >= 0: rec(c["start"], (), (), 0.0, 0.0)
    return out
