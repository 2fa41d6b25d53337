"""oracle/ivector_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
numpy restatement of the reference's online i-vector extraction for one utterance, the oracle of the NEXT scope row:
OnlineIvectorFeature as ivector-extract-online2 drives it (every frame weighted 1, no silence weighting, a fresh adaptation state).
Pinned to the reference binary's output in tests/test_oracle_ivector.py (tests/golden/ivector).  Paths relative to the reference's src/:

  features for the posteriors   OnlineCmvn (global stats) -> OnlineSpliceFrames -> OnlineTransform(LDA)   online2/online-ivector-feature.cc:144-163
  features for the statistics   OnlineSpliceFrames -> OnlineTransform(LDA), no CMVN (--online-cmvn-iextractor=false)   :239-243
  UBM log-likelihoods           DiagGmm::LogLikelihoods   gmm/diag-gmm.cc:557-586
  pruned posteriors             VectorToPosteriorEntry    hmm/posterior.cc:440-510 ; min-post GetMinPost :188-199 ; x posterior-scale :234-235
  statistics                    OnlineIvectorEstimationStats::AccStats   ivector/ivector-extractor.cc:611-670 (linear / quadratic terms, --max-count prior scaling)
  derived model terms           IvectorExtractor::ComputeDerivedVars(i)   :208-218 (U_i = M_i^T S_i^-1 M_i, S_i^-1 M_i)
  solution                      OnlineIvectorEstimationStats::GetIvector :732-756 = LinearCgd, warm-started, matrix/optimization.cc:453-560
  schedule                      UpdateStatsUntilFrame :248-277 (an i-vector after every frame t with t % period == 0), GetFrame :327-355
"""
import re
import numpy as np
from . import feat_oracle as fo


def read_models(dubm_txt, ie_txt):
    """the reference's own text dumps (gmm-global-copy / ivector-extractor-copy --binary=false) -> dict of arrays"""
    num = lambda s: np.array([float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", s)])
    g = open(dubm_txt).read(); sect = lambda t, a, b: num(t.split(a, 1)[1].split(b, 1)[0])
    w = sect(g, "<WEIGHTS>", "<MEANS_INVVARS>"); G = w.size
    mi = sect(g, "<MEANS_INVVARS>", "<INV_VARS>").reshape(G, -1); D = mi.shape[1]
    ubm = dict(gconsts=sect(g, "<GCONSTS>", "<WEIGHTS>").astype(np.float32), means_invvars=mi.astype(np.float32), inv_vars=sect(g, "<INV_VARS>", "</DiagGMM>").reshape(G, D).astype(np.float32))
    e = open(ie_txt).read()
    M = sect(e, "<M>", "<SigmaInv>")[1:].reshape(G, D, -1); R = M.shape[2]
    tri = sect(e, "<SigmaInv>", "<IvectorOffset>").reshape(G, D * (D + 1) // 2)
    S = np.zeros((G, D, D)); il = np.tril_indices(D)
    for i in range(G): S[i][il] = tri[i]; S[i] = S[i] + S[i].T - np.diag(np.diag(S[i]))
    offset = float(num(e.split("<IvectorOffset>", 1)[1].split("</IvectorExtractor>", 1)[0])[0])
    return ubm, dict(M=M, sigma_inv=S, prior_offset=offset, ivector_dim=R)


def _posterior(loglikes, num_gselect, min_post):
    """VectorToPosteriorEntry: [(gaussian, posterior)] sorted by decreasing posterior, renormalised after pruning"""
    ll = loglikes.astype(np.float32); mx = ll.max()
    cand = []
    if min_post != 0.0:
        cut = np.float32(mx + np.float32(np.log(np.float32(min_post))))
        cand = [(g, np.float32(np.exp(np.float32(l - mx)))) for g, l in enumerate(ll) if l > cut]
    if not cand: cand = [(g, np.float32(np.exp(np.float32(l - mx)))) for g, l in enumerate(ll)]
    cand.sort(key=lambda p: -p[1]); cand = cand[:min(num_gselect, len(cand))]
    tot = np.float32(sum(p for _, p in cand)); cutoff = np.float32(min_post) * tot
    while len(cand) > 1 and cand[-1][1] < cutoff: tot = np.float32(tot - cand[-1][1]); cand.pop()
    inv = np.float32(1.0) / tot
    return [(g, np.float32(p * inv)) for g, p in cand]


def _linear_cgd(A, b, x, max_iters):
    """LinearCgd<double> with the defaults of LinearCgdOptions (max_error 0, recompute_residual_factor 0.01)"""
    M = A.shape[0]; p = b - A @ x; r = -p; x_orig = x.copy()
    r_cur = r @ r; r_init = r_cur; r_rec = r_cur; rf = 0.01 * 0.01; k = 0
    while k < M + 5 and k != max_iters:
        Ap = A @ p; alpha = -(p @ r) / (p @ Ap)
        x = x + alpha * p; r = r + alpha * Ap; r_next = r @ r
        if r_next < rf * r_rec or r_next > r_rec / rf: r = A @ x - b; r_next = r @ r; r_rec = r_next
        if r_next <= np.finfo(np.float64).tiny: k += 1; break
        beta = r_next / r_cur; p = beta * p - r; r_cur = r_next; k += 1
    if r_cur > r_init and r_cur > r_init + 1.0e-10 * (b @ b): x = np.linalg.solve(A, b)        # "the squared residual has got worse": exact optimisation
    return x


def extract_online(feats, ubm, ie, lda, global_cmvn_stats, left_context=3, right_context=3, num_gselect=5, min_post=0.025, posterior_scale=0.1, max_count=0.0,
                   ivector_period=10, num_cg_iters=15, repeat=False):
    """what ivector-extract-online2 writes for one utterance: [ceil(T / period) x ivector_dim] float32 (repeat: one row per frame)"""
    f = np.ascontiguousarray(feats, np.float32); T = f.shape[0]; lda = np.asarray(lda, np.float32); R = ie["ivector_dim"]
    lin, off = (lda, np.zeros(lda.shape[0], np.float32)) if lda.shape[1] == f.shape[1] * (left_context + right_context + 1) else (lda[:, :-1], lda[:, -1])
    def spliced_lda(x):
        sp = np.concatenate([x[np.clip(np.arange(T) + o, 0, T - 1)] for o in range(-left_context, right_context + 1)], axis=1)
        return (off[None, :] + sp @ lin.T).astype(np.float32)
    x_post = spliced_lda(fo.cmvn_online(f, global_cmvn_stats)); x_stats = spliced_lda(f)
    loglikes = ubm["gconsts"][None, :] + x_post @ ubm["means_invvars"].T - np.float32(0.5) * (x_post * x_post) @ ubm["inv_vars"].T
    U = np.einsum("gdr,gde,ges->grs", ie["M"], ie["sigma_inv"], ie["M"]); SM = np.einsum("gde,ger->gdr", ie["sigma_inv"], ie["M"])      # derived variables
    prior = ie["prior_offset"]; quad = np.eye(R); linear = np.zeros(R); linear[0] = prior; nframes = 0.0
    cur = np.zeros(R); cur[0] = prior; history = []; pending = []
    for t in range(T):
        pending.append(t)
        if t % ivector_period == 0:
            post = {}                                           # Gaussian -> [(frame, weight)]
            for u in pending:
                for g, p in _posterior(loglikes[u], num_gselect, min(np.float32(min_post), np.float32(0.99))):
                    post.setdefault(g, []).append((u, np.float32(p * np.float32(posterior_scale * 1.0))))
            tot = 0.0
            for g, fw in post.items():
                wf = np.zeros(x_stats.shape[1]); gw = np.float32(0.0)
                for u, w_ in fw: wf += float(w_) * x_stats[u].astype(np.float64); gw = np.float32(gw + w_)
                linear += SM[g].T @ wf; quad += float(gw) * U[g]; tot += float(gw)
            if max_count > 0.0:
                change = max(nframes + tot, max_count) / max_count - max(nframes, max_count) / max_count
                if change != 0.0: linear[0] += prior * change; quad += change * np.eye(R)
            nframes += tot; pending = []
            if nframes > 0.0:
                if cur[0] == 0.0: cur[0] = prior
                cur = _linear_cgd(quad, linear, cur, num_cg_iters)
            else: cur = np.zeros(R); cur[0] = prior
            history.append(cur.astype(np.float32))
    rows = [history[t // ivector_period] for t in (range(T) if repeat else range(0, T, ivector_period))]
    out = np.array(rows, np.float32).reshape(-1, R); out[:, 0] -= np.float32(prior)
    return out
