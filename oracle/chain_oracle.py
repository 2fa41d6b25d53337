"""oracle/chain_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
The LF-MMI denominator of chain training, two ways:
  * ref_den(...)    runs the REFERENCE's own code: oracle/_ref/bin/ref-chain-den = chain/chain-den-graph.cc + chain/chain-denominator.cc compiled
                    unmodified from /root/reference (oracle/build_ref.sh; driver oracle/ref_tools/ref_chain_den.cc);
  * den_oracle(...) a numpy restatement of the same computation -- DenominatorGraph's constructor (chain/chain-den-graph.cc:52-143) and
                    DenominatorComputation::Forward / Backward (chain/chain-denominator.cc:106-440, CPU path) -- pinned to the binary's output by
                    tests/test_oracle_chain.py and to tests/golden/chain_den_golden.npz where oracle/_ref does not exist.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import os, struct, subprocess, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
_BIN = os.path.join(HERE, "_ref", "bin", "ref-chain-den")

def available():
    return os.path.exists(_BIN)

def ref_den(fst, num_pdfs, nnet_output, num_sequences, leaky_hmm_coefficient=1.0e-05, deriv_weight=-1.0):
    """returns dict(objf, ok, initial_probs [S], deriv [T*B, P]) from the reference's DenominatorComputation"""
    out = np.ascontiguousarray(nnet_output, np.float32); TB, P = out.shape; assert P == num_pdfs and TB % num_sequences == 0
    S, A = int(fst.num_states), int(fst.arc_offsets[-1])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    with tempfile.TemporaryDirectory() as td:
        with open(f"{td}/in.bin", "wb") as f:
            f.write(struct.pack("<7i2f", 0x4b34, S, int(fst.start), A, P, num_sequences, TB // num_sequences, leaky_hmm_coefficient, deriv_weight))
            f.write(np.ascontiguousarray(fst.arc_offsets, np.int64).tobytes()); f.write(np.ascontiguousarray(fst.ilabel, np.int32).tobytes()); f.write(np.ascontiguousarray(fst.nextstate, np.int32).tobytes())
            f.write(np.ascontiguousarray(fst.weight, np.float32).tobytes()); f.write(np.ascontiguousarray(fst.final, np.float32).tobytes()); f.write(out.tobytes())
        subprocess.check_call([_BIN, f"{td}/in.bin", f"{td}/out.bin"], env=env, stderr=subprocess.DEVNULL)
        raw = open(f"{td}/out.bin", "rb").read()
    objf, ok = struct.unpack("<fi", raw[:8]); init = np.frombuffer(raw, np.float32, S, 8); deriv = np.frombuffer(raw, np.float32, TB * P, 8 + 4 * S).reshape(TB, P)
    return dict(objf=float(objf), ok=bool(ok), initial_probs=init.copy(), deriv=deriv.copy())

def initial_probs(fst):
    """DenominatorGraph::SetInitialProbs (chain-den-graph.cc:97-143): all mass on the start state, 100 steps through the HMM with every state's
    outgoing mass (final-prob included) normalised to one, renormalised after each step, the 100 distributions averaged"""
    S = int(fst.num_states); off = np.asarray(fst.arc_offsets, np.int64); src = np.repeat(np.arange(S), np.diff(off))
    pw = np.exp(-np.asarray(fst.weight, np.float32).astype(np.float64)); tot = np.exp(-np.asarray(fst.final, np.float32).astype(np.float64))
    np.add.at(tot, src, pw); norm = 1.0 / tot
    cur = np.zeros(S); cur[int(fst.start)] = 1.0; avg = np.zeros(S); nx = np.asarray(fst.nextstate, np.int64)
    for _ in range(100):
        avg += cur / 100.0
        nxt = np.zeros(S); np.add.at(nxt, nx, (cur * norm)[src] * pw)
        cur = nxt / nxt.sum()
    return avg.astype(np.float32)

def den_oracle(fst, num_pdfs, nnet_output, num_sequences, leaky_hmm_coefficient=1.0e-05, deriv_weight=-1.0):
    """float32 quantities with float64 accumulation per HMM state, like the CPU path of the reference.  Returns the same dict as ref_den()."""
    from scipy.sparse import csr_matrix
    f32, f64 = np.float32, np.float64
    out = np.asarray(nnet_output, f32); TB, P = out.shape; B = int(num_sequences); T = TB // B; S = int(fst.num_states)
    off = np.asarray(fst.arc_offsets, np.int64); A = int(off[-1]); src = np.repeat(np.arange(S), np.diff(off)); dst = np.asarray(fst.nextstate, np.int64)
    pdf = np.asarray(fst.ilabel, np.int64) - 1; tp = np.exp(-np.asarray(fst.weight, f32)).astype(f32)
    to_dst = csr_matrix((np.ones(A), (np.arange(A), dst)), shape=(A, S)); to_src = csr_matrix((np.ones(A), (np.arange(A), src)), shape=(A, S))
    to_pdf = csr_matrix((np.ones(A), (np.arange(A), pdf)), shape=(A, P))
    init = initial_probs(fst); leaky = f32(leaky_hmm_coefficient)
    probs = np.exp(np.clip(out, -30.0, 30.0)).astype(f32).reshape(T, B, P)               # ApplyExpLimited(-30, 30), :91
    alpha = np.zeros((T + 1, B, S + 1), f32)
    a = np.broadcast_to(init, (B, S)).astype(f32)                                         # AlphaFirstFrame
    for t in range(T + 1):
        if t > 0:                                                                         # AlphaGeneralFrame (:122-198)
            prev = alpha[t - 1]; scale = (1.0 / prev[:, S].astype(f64)).astype(f32)
            prod = (prev[:, src] * tp[None, :]) * probs[t - 1][:, pdf]
            a = ((prod.astype(f64) @ to_dst) * scale[:, None].astype(f64)).astype(f32)
        asum = a.astype(f64).sum(1).astype(f32)                                           # AlphaDash (:200-220)
        alpha[t, :, :S] = a + leaky * init[None, :] * asum[:, None]; alpha[t, :, S] = asum
    tot = alpha[T, :, :S].astype(f64).sum(1).astype(f32)                                  # ComputeTotLogLike (:262-300)
    objf = float(np.log(tot.astype(f64)).sum() + np.log(alpha[:T, :, S].astype(f64)).sum())
    deriv = np.zeros((T, B, P), f32)
    def beta_of(bd):                                                                      # Beta (:222-248)
        bsum = (leaky * (bd.astype(f64) * init[None, :]).sum(1)).astype(f32)
        return bd + bsum[:, None]
    bd = np.broadcast_to((1.0 / tot)[:, None], (B, S)).astype(f32)                        # BetaDashLastFrame (:320-336)
    beta = beta_of(bd); ab0 = ds0 = None
    for t in range(T - 1, -1, -1):                                                        # BetaDashGeneralFrame (:338-402)
        ad = alpha[t]; inv = ad[:, S]
        vf = (tp[None, :] * beta[:, dst]) * probs[t][:, pdf]
        occ = ad[:, :S] / inv[:, None]
        deriv[t] = ((vf * occ[:, src]).astype(f64) @ to_pdf).astype(f32)
        bd = ((vf.astype(f64) @ to_src) / inv[:, None].astype(f64)).astype(f32)
        if t == 0: ab0 = float((ad[:, :S].astype(f64) * bd).sum()); ds0 = float(deriv[0].astype(f64).sum())
        beta = beta_of(bd)
    ok = bool(np.isfinite(objf) and abs(ab0 - B) <= 2.0 and abs(ds0 - B) <= 2.0)
    return dict(objf=objf, ok=ok, initial_probs=init, deriv=(f32(deriv_weight) * deriv).reshape(TB, P))

# ---- numerator and the whole objective ------------------------------------------------------------------------------------------------------
_BIN_OBJF = os.path.join(HERE, "_ref", "bin", "ref-chain-objf")
def objf_available():
    return os.path.exists(_BIN_OBJF)

def _fst_bytes(f):
    return (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
            np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())

def ref_objf(den_fst, num_pdfs, merged_sup_fst, nnet_output, num_sequences, leaky_hmm_coefficient=1.0e-05, l2_regularize=0.0, weight=1.0):
    """the reference's ComputeChainObjfAndDeriv (out-of-range penalty off) on a MERGED supervision FST: dict(objf, l2_term, weight, deriv, xent_deriv)"""
    out = np.ascontiguousarray(nnet_output, np.float32); TB, P = out.shape
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    with tempfile.TemporaryDirectory() as td:
        with open(f"{td}/in.bin", "wb") as f:
            f.write(struct.pack("<9i3f", 0x4b35, int(den_fst.num_states), int(den_fst.start), int(den_fst.arc_offsets[-1]), P, num_sequences, TB // num_sequences, int(merged_sup_fst.num_states), int(merged_sup_fst.arc_offsets[-1]),
                                leaky_hmm_coefficient, l2_regularize, weight))
            f.write(_fst_bytes(den_fst)); f.write(_fst_bytes(merged_sup_fst)); f.write(out.tobytes())
        subprocess.check_call([_BIN_OBJF, f"{td}/in.bin", f"{td}/out.bin"], env=env, stderr=subprocess.DEVNULL)
        raw = open(f"{td}/out.bin", "rb").read()
    objf, l2, wt = struct.unpack("<3f", raw[:12]); d = np.frombuffer(raw, np.float32, TB * P, 12).reshape(TB, P); x = np.frombuffer(raw, np.float32, TB * P, 12 + 4 * TB * P).reshape(TB, P)
    return dict(objf=float(objf), l2_term=float(l2), weight=float(wt), deriv=d.copy(), xent_deriv=x.copy())

def ref_objf_e2e(den_fst, num_pdfs, e2e_fsts, nnet_output, leaky_hmm_coefficient=1.0e-05, l2_regularize=0.0, weight=1.0):
    """the reference's ComputeChainObjfAndDeriv on END-TO-END supervisions (Supervision::e2e_fsts -> chain-training.cc:86-215 with GenericNumeratorComputation, compiled unmodified;
    out-of-range penalty off): dict(objf, l2_term, weight, deriv, xent_deriv)"""
    out = np.ascontiguousarray(nnet_output, np.float32); TB, P = out.shape; B = len(e2e_fsts)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    so = np.concatenate([[0], np.cumsum([f.num_states for f in e2e_fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in e2e_fsts])])
    with tempfile.TemporaryDirectory() as td:
        with open(f"{td}/in.bin", "wb") as f:
            f.write(struct.pack("<9i3f", 0x4b37, int(den_fst.num_states), int(den_fst.start), int(den_fst.arc_offsets[-1]), P, B, TB // B, int(so[-1]), int(ab[-1]), leaky_hmm_coefficient, l2_regularize, weight))
            f.write(_fst_bytes(den_fst)); f.write(so.tobytes())
            f.write(np.concatenate([[0]] + [np.asarray(g.arc_offsets[1:], np.int64) + b for g, b in zip(e2e_fsts, ab[:-1])]).astype(np.int64).tobytes())
            for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): f.write(np.concatenate([getattr(g, k) for g in e2e_fsts]).astype(dt).tobytes())
            f.write(out.tobytes())
        subprocess.check_call([_BIN_OBJF, f"{td}/in.bin", f"{td}/out.bin"], env=env, stderr=subprocess.DEVNULL)
        raw = open(f"{td}/out.bin", "rb").read()
    objf, l2, wt = struct.unpack("<3f", raw[:12]); d = np.frombuffer(raw, np.float32, TB * P, 12).reshape(TB, P); x = np.frombuffer(raw, np.float32, TB * P, 12 + 4 * TB * P).reshape(TB, P)
    return dict(objf=float(objf), l2_term=float(l2), weight=float(wt), deriv=d.copy(), xent_deriv=x.copy())

def e2e_num_oracle(fsts, num_pdfs, nnet_output, weight=1.0):
    """GenericNumeratorComputation::ForwardBackward (chain/chain-generic-numerator.cc:238-275 with AlphaRemainingFrames :163-236 and BetaRemainingFrames :322-363), sequence by
    sequence in float64 log domain (the reference's per-frame normaliser and the offset of state 0's arcs cancel in exact arithmetic).  Returns (total log-prob -- NOT times the
    weight, like the reference's return value --, weight * occupation probabilities [T*B, P])."""
    out = np.asarray(nnet_output, np.float32).astype(np.float64); B = len(fsts); TB, P = out.shape; T = TB // B
    post = np.zeros((TB, P), np.float64); tot = 0.0
    for n, f in enumerate(fsts):
        S = int(f.num_states); off = np.asarray(f.arc_offsets, np.int64); src = np.repeat(np.arange(S), np.diff(off)); dst = np.asarray(f.nextstate); pdf = np.asarray(f.ilabel) - 1; w = -np.asarray(f.weight, np.float64)
        fin = np.where(np.isfinite(f.final), -np.asarray(f.final, np.float64), -np.inf)
        alpha = np.full((T + 1, S), -np.inf); alpha[0, 0] = 0.0
        for t in range(T):
            x = alpha[t, src] + w + out[t * B + n, pdf]
            for a in np.argsort(dst, kind="stable"): alpha[t + 1, dst[a]] = np.logaddexp(alpha[t + 1, dst[a]], x[a])
        lp = np.logaddexp.reduce(alpha[T] + fin); tot += lp
        beta = fin.copy()
        for t in range(T - 1, -1, -1):
            y = w + beta[dst] + out[t * B + n, pdf]; nb = np.full(S, -np.inf)
            for a in range(len(src)): nb[src[a]] = np.logaddexp(nb[src[a]], y[a]); post[t * B + n, pdf[a]] += np.exp(alpha[t, src[a]] + y[a] - lp)
            beta = nb
    return float(tot), (float(weight) * post).astype(np.float32)

def objf_oracle_e2e(den_fst, num_pdfs, fsts, nnet_output, leaky_hmm_coefficient=1.0e-05, l2_regularize=0.0, weight=1.0):
    """ComputeChainObjfAndDerivE2e (chain/chain-training.cc:86-215), out-of-range penalty off"""
    out = np.asarray(nnet_output, np.float32); B = len(fsts); TB, P = out.shape
    den = den_oracle(den_fst, num_pdfs, out, B, leaky_hmm_coefficient, -weight)
    num_lp, xent = e2e_num_oracle(fsts, num_pdfs, out, weight); deriv = den["deriv"].astype(np.float32) + xent
    objf = num_lp - weight * den["objf"]; wt = weight * TB; l2 = 0.0
    if l2_regularize != 0.0: scale = weight * l2_regularize; l2 = -0.5 * scale * float((out.astype(np.float64) ** 2).sum()); deriv = deriv - np.float32(scale) * out
    return dict(objf=float(objf), l2_term=float(l2), weight=float(wt), deriv=deriv.astype(np.float32), xent_deriv=xent)

def num_oracle(fsts, num_pdfs, nnet_output, weight=1.0):
    """NumeratorComputation (chain/chain-numerator.cc:115-213) sequence by sequence, log domain in double: returns (weight * total log-prob, weight * occupation
    probabilities [T*B, P]).  fsts: the UNMERGED supervision FSTs (the merged FST's total factorises over the sequences)."""
    out = np.asarray(nnet_output, np.float32).astype(np.float64); B = len(fsts); TB, P = out.shape; T = TB // B
    post = np.zeros((TB, P), np.float64); tot = 0.0
    for n, f in enumerate(fsts):
        S = int(f.num_states); off = np.asarray(f.arc_offsets, np.int64); time = np.full(S, -1); time[0] = 0
        for s in range(S):
            for a in range(off[s], off[s + 1]): time[f.nextstate[a]] = time[s] + 1
        la = np.full(S, -np.inf); la[0] = 0.0
        for s in range(S):                                                    # states are sorted by time: one serial sweep, like the reference
            for a in range(off[s], off[s + 1]):
                x = la[s] + out[time[s] * B + n, f.ilabel[a] - 1] - float(f.weight[a]); la[f.nextstate[a]] = np.logaddexp(la[f.nextstate[a]], x)
        fin = np.asarray(f.final, np.float64); lp = np.logaddexp.reduce([la[s] - fin[s] for s in range(S) if np.isfinite(fin[s])]); tot += lp
        lb = np.full(S, -np.inf)
        for s in range(S - 1, -1, -1):
            b = -fin[s] if np.isfinite(fin[s]) else -np.inf
            for a in range(off[s], off[s + 1]):
                y = out[time[s] * B + n, f.ilabel[a] - 1] - float(f.weight[a]) + lb[f.nextstate[a]]; b = np.logaddexp(b, y)
                post[time[s] * B + n, f.ilabel[a] - 1] += np.exp(la[s] + y - lp)
            lb[s] = b
    return float(weight) * tot, (float(weight) * post).astype(np.float32)

def objf_oracle(den_fst, num_pdfs, fsts, nnet_output, leaky_hmm_coefficient=1.0e-05, l2_regularize=0.0, out_of_range_regularize=0.0, apply_out_of_range_penalty=False, weight=1.0):
    """ComputeChainObjfAndDeriv (chain/chain-training.cc:242-337)"""
    out = np.asarray(nnet_output, np.float32); B = len(fsts); TB, P = out.shape
    den = den_oracle(den_fst, num_pdfs, out, B, leaky_hmm_coefficient, -weight)
    deriv = den["deriv"].astype(np.float32).copy()
    if apply_out_of_range_penalty and out_of_range_regularize != 0.0:
        sc = np.float32(2.0 * out_of_range_regularize); deriv -= sc * (np.where(out < -30, out + 30, 0) + np.where(out > 30, out - 30, 0)).astype(np.float32)
    num_lp, xent = num_oracle(fsts, num_pdfs, out, weight)
    deriv = deriv + xent
    objf = num_lp - weight * den["objf"]; wt = weight * TB
    if not np.isfinite(objf) or not den["ok"]: deriv[:] = 0; xent = np.zeros_like(xent); objf = -10.0 * wt
    l2 = 0.0
    if l2_regularize != 0.0:
        scale = weight * l2_regularize; l2 = -0.5 * scale * float((out.astype(np.float64) ** 2).sum()); deriv = deriv - np.float32(scale) * out
    return dict(objf=float(objf), l2_term=float(l2), weight=float(wt), deriv=deriv.astype(np.float32), xent_deriv=xent)
