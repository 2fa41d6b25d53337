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
