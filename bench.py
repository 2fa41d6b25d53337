#!/usr/bin/env python3
"""bench.py -- RTFx of the batched acoustic pipeline on MI355X (driver contract in the task statement; metric definition SURVEY 8d).

A "step" = one pass of the hot path over one batch of synthetic 16 kHz audio, from the first waveform byte in (page-locked) HOST memory to the
last lattice handed to the writer:
  PCM16 H2D -> fbank (k3_feat_compute_batch_pcm16) -> 17-layer TDNN-F forward (k3_nnet_forward) -> HCLG lattice decode (k3_decoder_decode_batch:
  token passing + lattice-beam pruning on the GPU) -> raw lattices compacted and copied to host buffers (k3_decoder_get_raw_lattices)
  -> per utterance Connect + lattice determinization on a pool of host threads (k3h_postprocess_batch), one batch behind the GPU.
value = audio seconds processed by ALL ranks / wall seconds (max over ranks), everything above inside the timed region (the determinization
of the last batch included).

Decoder mode.  `value` is measured with k3_decoder_config.literal_order = 1: the raw lattices are then IDENTICAL to the reference's CPU
LatticeFasterDecoder (states, arcs, labels, cost bits; tests/test_decoder_literal_gpu.py checks 32 of this workload's utterances against the
reference's own decoder source).  `value_two_pass` is the same pipeline with the order-independent fast decoder (default mode), whose lattices
are a close but NOT identical relative of the reference's at this configuration (profiles/r02_decoder_parity_bench_config.json).

Weak scaling: every rank decodes its own --utts utterances; the decoding graph is built on rank 0 and broadcast ONCE over RCCL (before the
timed region); no data-path collective.
Extra objects in the JSON line: "roofline" (the dominant kernel vs its CDNA4 bound, timed with HIP events on the launch stream),
"roofline_gemm" (the TDNN-F affine GEMMs vs the FP32 MFMA peak), "stage_ms", "decode_stats", "value_kernels" and "cpu_baseline" (the reference's
own compute-fbank-feats + nnet3-compute + LatticeFasterDecoder binaries from oracle/_ref on the host cores; rank 0, N=1 only, bounded sample)."""
import argparse, ctypes, json, os, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# e2e_parity.reference_vs_itself: the second reference run takes MKL's "conditional numerical reproducibility" branch (tools/probe_mkl_paths.py: the setting
# that moves its sgemm rounding on Intel and AMD hosts alike)
ALT_BLAS = {"MKL_CBWR": "COMPATIBLE"}
BEAM, LATTICE_BEAM, MAX_ACTIVE = 15.0, 8.0, 10000      # BASELINE.json configs[2]: beam 15; recipes' lattice-beam 8; CudaDecoderConfig max-active

def _best_path(n, start, frame, final, src, dst, il, ol, cost):
    """tropical shortest path of a raw lattice given as arrays (states ordered by nothing in particular; arcs go forward in `frame`, epsilon arcs stay
    inside a frame): (ilabels without 0, olabels without 0, total cost), or None.  The same routine is applied to the reference's and to the GPU's
    lattice (GetBestPath + GetLinearSymbolSequence, decoder/decoder-wrappers.cc:322-331)."""
    if n == 0 or start < 0: return None
    best = np.full(n, np.inf)
    best[start] = 0.0
    back = np.full(n, -1, np.int64)
    order = np.argsort(frame[src], kind="stable").tolist()
    srcl, dstl, cl = src.tolist(), dst.tolist(), cost.tolist()
    bl = best.tolist()
    changed = True
    while changed:
        changed = False
        for a in order:
            v = bl[srcl[a]] + cl[a]
            if v < bl[dstl[a]]:
                bl[dstl[a]] = v
                back[dstl[a]] = a
                changed = True
    best = np.asarray(bl)
    fin = np.nonzero(np.isfinite(final))[0]
    if fin.size == 0: return None
    end = int(fin[np.argmin(best[fin] + final[fin])])
    if not np.isfinite(best[end]): return None
    ils, ols, s_ = [], [], end
    while s_ != start:
        a = int(back[s_])
        ils.append(int(il[a]))
        ols.append(int(ol[a]))
        s_ = srcl[a]
    return [i for i in ils[::-1] if i], [o for o in ols[::-1] if o], float(best[end] + final[end])

def _view_ref(r):      # oracle.ref_decoder.decode's dict -> the arrays _compare works on
    return dict(n=r["frame"].size, start=r["start"], frame=r["frame"], final=r["final_graph"].astype(np.float64) + r["final_ac"], src=r["src"], dst=r["dst"],
        il=r["ilabel"], ol=r["olabel"], cost=r["graph"].astype(np.float64) + r["ac"])
def _view_raw(l):      # kaldi_amd.lattice.RawLattice (the C ABI's k3_decoder_get_raw_lattices output)
    return dict(n=l.num_states, start=l.start_index(), frame=l.st_frame, final=l.st_final.astype(np.float64), src=l.arc_src, dst=l.arc_dst, il=l.arc_ilabel,
        ol=l.arc_olabel, cost=l.arc_graph.astype(np.float64) + l.arc_ac)

def _compare(a, a_ll, b, b_ll):
    """one utterance of the end-to-end gate (SURVEY 8d gate 4): two chains' raw lattices (views above) and the log-likelihoods their decoders consumed"""
    r = {"max_abs_loglike_diff": float(np.abs(a_ll - b_ll).max()) if a_ll.shape == b_ll.shape else float("inf")}
    pa, pb = (_best_path(x["n"], x["start"], x["frame"], x["final"], x["src"], x["dst"], x["il"], x["ol"], x["cost"]) for x in (a, b))
    r["best_path_identical"] = bool(pa is not None and pb is not None and pa[0] == pb[0] and pa[1] == pb[1])
    r["words_identical"] = bool(pa is not None and pb is not None and pa[1] == pb[1])
    r["best_cost_diff"] = abs(pa[2] - pb[2]) if (pa and pb) else float("inf")
    key = lambda x: np.sort((x["frame"][x["src"]].astype(np.int64) << 44) | ((x["frame"][x["dst"]] - x["frame"][x["src"]]).astype(np.int64) << 43) | (x["il"].
            astype(np.int64) << 21) | x["ol"].astype(np.int64))
    ka, kb = key(a), key(b)
    r["lattice_identical"] = bool(ka.size == kb.size and np.array_equal(ka, kb) and np.array_equal(np.bincount(a["frame"]), np.bincount(b["frame"])))
    r["ref_arcs"] = int(ka.size)
    r["gpu_arcs"] = int(kb.size)
    r["arcs_only_in_one"] = int(ka.size + kb.size - 2 * np.intersect1d(ka, kb).size) if not r["lattice_identical"] else 0
    return r

def _cpu_worker(job):
    """one host core: reference compute-fbank-feats -> nnet3-compute -> LatticeFasterDecoder::Decode on its utterances; returns (audio_s, wall_s, stage seconds,
    per-utterance comparisons with the GPU chain's results where the job carries them)"""
    from oracle import kaldi_io as kio, lattice_oracle as lo, ref_decoder as rd
    from kaldi_amd import synth
    # utts: [(name, int16 samples)]; gpu: None or (lattices, loglikes, out_offsets, U, features, frame offsets) of the GPU chain
    wid, utts, utt_seconds, model_path, graph, num_pdfs, gpu = job
    bindir = os.path.join(ROOT, "oracle", "_ref", "bin")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
    cmp_ = []
    cmp2 = []
    keep_ = {}
    with tempfile.TemporaryDirectory() as td:
        scp = []
        for name, pcm in utts:
            kio.write_wav(f"{td}/{name}.wav", pcm)
            scp.append(f"{name} {td}/{name}.wav")
        open(f"{td}/wav.scp", "w").write("\n".join(scp) + "\n")
        t0 = time.time()
        subprocess.check_call([f"{bindir}/compute-fbank-feats", "--dither=0", "--num-mel-bins=40", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], env=env,
            stderr=subprocess.DEVNULL)
        t1 = time.time()
        subprocess.check_call([f"{bindir}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", "--frames-per-chunk=150", model_path,
                 f"ark:{td}/f.ark", f"ark:{td}/o.ark"], env=env, stderr=subprocess.DEVNULL)
        t2 = time.time()
        lls = kio.read_ark(f"{td}/o.ark")
        cfg = lo.Config(beam=BEAM, lattice_beam=LATTICE_BEAM, max_active=MAX_ACTIVE)
        t2p = synth.tid2pdf(num_pdfs)
        dec_s = 0.0
        refs = {}
        for name, _ in utts:
            ref = rd.decode(graph, lls[name], t2p, cfg)      # time inside LatticeFasterDecoder::Decode, reported by the binary
            dec_s += ref["decode_seconds"]
            u = int(name[1:])
            if gpu is not None and u < gpu[3]:
                refs[name] = ref
                c = _compare(_view_ref(ref), lls[name], _view_raw(gpu[0][u]), gpu[1][gpu[2][u]:gpu[2][u + 1]])
                c["utt"] = u
                cmp_.append(c)
        # (untimed) the reference against ITSELF: the same features through nnet3-compute on another of MKL's code paths, the same decoder -- how far the
        # reference's own results move
        if refs:
            fr = kio.read_ark(f"{td}/f.ark")      # under a float32 rounding difference of the size that separates the two chains
            # (the checker) exact value of the reference's formulas on its float32 tables: float64 data path, oracle/feat_oracle_path.inc
            from oracle import feat_oracle as fo
            fopts = fo.fbank_opts(dither=0.0, num_bins=40)
            pcm_by_name = dict(utts)
            for c in cmp_:
                g_f = gpu[4][gpu[5][c["utt"]]:gpu[5][c["utt"] + 1]]
                r_f = fr["u%d" % c["utt"]]
                fd = np.abs(r_f - g_f)
                c["max_abs_feature_diff"] = float(fd.max())
                c["feat_n"] = int(fd.size)
                c["feat_above"] = int((fd > 1e-4).sum())
                c["feat_sum"] = float(fd.sum(dtype=np.float64))
                ex = fo.compute_features_f64path(pcm_by_name["u%d" % c["utt"]].astype(np.float32), fopts)
                c["feat_err_gpu_exact"] = float(np.abs(g_f - ex).max())
                c["feat_err_ref_exact"] = float(np.abs(r_f - ex).max())
                c["feat_ref_exact_above"] = int((np.abs(r_f - ex) > 1e-4).sum())
            for name in refs:      # the reference's features, log-likelihoods and lattice: inputs / expected outputs of the stage gates
                keep_[int(name[1:])] = (fr[name], lls[name], refs[name])
            try:
                subprocess.check_call([f"{bindir}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", "--frames-per-chunk=150", model_path,
                         f"ark:{td}/f.ark", f"ark:{td}/o2.ark"], env=dict(env, **ALT_BLAS), stderr=subprocess.DEVNULL)
                lls2 = kio.read_ark(f"{td}/o2.ark")
                for name in refs:
                    c = _compare(_view_ref(refs[name]), lls[name], _view_ref(rd.decode(graph, lls2[name], t2p, cfg)), lls2[name])
                    c["utt"] = int(name[1:])
                    cmp2.append(c)
            except Exception as e: cmp2.append({"error": repr(e)})
    return len(utts) * utt_seconds, (t2 - t0) + dec_s, (t1 - t0, t2 - t1, dec_s), cmp_, cmp2, keep_

def e2e_gate(flips_gpu, flips_self, utterances, alpha=0.01):
    """The end-to-end parity gate as a stated two-sample test (SURVEY 8d gate 4, "at WER parity"; what DecodeUtteranceLatticeFaster's best path is compared on,
    decoder/decoder-wrappers.cc:322-373).  Both chains are compared with the SAME reference run on the SAME utterances, so the samples are paired: flips_gpu = utterances
    whose best path differs between the GPU chain and the reference chain, flips_self = utterances whose best path differs between the reference chain and the reference
    chain run a second time (nnet3-compute on another MKL code path).  Exact one-sided McNemar test on the discordant pairs: b = flipped by the GPU chain only, c = flipped
    by the second reference run only; under H0 "the GPU chain flips no more often than the reference does against itself" b ~ Binomial(b + c, 1/2); p = P(X >= b).
    The gate passes when p >= alpha: no evidence at level alpha that the GPU chain is further from the reference than the reference is from itself.  Independent of the
    sample size (128 or 512 utterances give the same verdict for the same rates), unlike a threshold of "self rate - 1 / utterances"."""
    from math import comb
    g, s_ = set(flips_gpu), set(flips_self)
    b, c = len(g - s_), len(s_ - g)
    n = b + c
    p = 1.0 if n == 0 else sum(comb(n, k) for k in range(b, n + 1)) / float(2 ** n)
    # what the test could have seen: the smallest b that would have failed at this c
    b_fail = next((bb for bb in range(0, 10 * (c + 8)) if sum(comb(bb + c, k) for k in range(bb, bb + c + 1)) / float(2 ** (bb + c)) < alpha), None)
    return {"test": "exact one-sided McNemar test on paired per-utterance best-path flips (GPU chain vs reference) against (reference vs itself)", "alpha": alpha,
            "utterances": int(utterances), "flips_gpu": len(g), "flips_self": len(s_), "flipped_by_both": len(g & s_), "gpu_only_b": b, "self_only_c": c, "p_value": p,
            "pass": bool(p >= alpha), "gpu_only_flips_that_would_fail": b_fail,
            "note": "pass = p_value >= alpha; H0: the GPU chain's best path differs from the reference's no more often than the reference's own second run does"}

def _ctrl_worker(job):
    """control (ii) of the end-to-end gate: the GPU chain's FEATURES through the reference's nnet3-compute and LatticeFasterDecoder; returns the utterances whose best
    path differs from the all-reference chain's"""
    from oracle import kaldi_io as kio, lattice_oracle as lo, ref_decoder as rd
    from kaldi_amd import synth
    items, model_path, graph, num_pdfs = job      # items: [(utt, GPU features, (reference features, log-likelihoods, lattice))]
    bindir = os.path.join(ROOT, "oracle", "_ref", "bin")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
    flips = []
    with tempfile.TemporaryDirectory() as td:
        kio.write_ark(f"{td}/f.ark", {f"u{u}": f for u, f, _ in items})
        subprocess.check_call([f"{bindir}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", "--frames-per-chunk=150", model_path,
                 f"ark:{td}/f.ark", f"ark:{td}/o.ark"], env=env, stderr=subprocess.DEVNULL)
        lls = kio.read_ark(f"{td}/o.ark")
        cfg = lo.Config(beam=BEAM, lattice_beam=LATTICE_BEAM, max_active=MAX_ACTIVE)
        t2p = synth.tid2pdf(num_pdfs)
        for u, _, kept_u in items:
            c = _compare(_view_ref(kept_u[2]), kept_u[1], _view_ref(rd.decode(graph, lls[f"u{u}"], t2p, cfg)), lls[f"u{u}"])
            if not c["best_path_identical"]: flips.append(u)
    return flips

def cpu_baseline(model_path, graph, num_pdfs, utt_seconds, pcm_of, gpu=None, utts_per_core=12, max_procs=64):
    """The same workload on the host cores, bounded sample, the way decode.sh --nj splits it: P independent single-threaded workers, each running
    the REFERENCE's own binaries (oracle/_ref, built from /root/reference by oracle/build_ref.sh) on its utterances.  Utterance u of the sample is the
    GPU batch's utterance u (same PCM16, `pcm_of(u)`); with `gpu` = (raw lattices, log-likelihoods, row offsets, U) of the GPU chain on that batch the
    workers also compare the two chains' results utterance by utterance: the end-to-end parity gate of SURVEY 8d.  Returns (cpu_baseline, e2e_parity)."""
    from oracle import ref_decoder as rd
    bindir = os.path.join(ROOT, "oracle", "_ref", "bin")
    if not (os.path.exists(os.path.join(bindir, "nnet3-compute")) and rd.available()):
        return {"error": "oracle/_ref is not built (needs /root/reference once; it travels to the GPU box)"}, None, {}
    ncores = os.cpu_count() or 1
    P = max(1, min(ncores, max_procs))
    from kaldi_amd import synth
    # untimed warm-up: pages the binaries and MKL in
    _cpu_worker((99, [("u99999", synth.gaussian_pcm16(16000, 99))], 1.0, model_path, graph, num_pdfs, None))
    # round-robin over the workers, so that the compared utterances 0 .. P * utts_per_core - 1 are spread evenly
    jobs = [(w, [(f"u{u}", pcm_of(u)) for u in range(w, P * utts_per_core, P)], utt_seconds, model_path, graph, num_pdfs, gpu) for w in range(P)]
    t0 = time.time()
    with ThreadPoolExecutor(P) as ex:      # threads only launch and wait for the single-threaded reference processes (and compare lattices)
        res = list(ex.map(_cpu_worker, jobs))
    wall = time.time() - t0
    audio = sum(r[0] for r in res)
    per_core = [r[0] / r[1] for r in res]
    st = np.sum([r[2] for r in res], axis=0)
    base = {"value": audio / max(r[1] for r in res), "unit": "RTFx (audio-s/wall-s)", "cores": P, "kind": "reference", "host_cores_available": ncores,
         "per_core_rtfx_mean": float(np.mean(per_core)), "extrapolated_all_cores": float(np.mean(per_core)) * ncores,
            "extrapolated_all_cores_note": (f"per-core mean x {ncores} host cores (not measured: memory bandwidth and SMT are shared; an upper bound)" if P <
            ncores else f"all {ncores} host cores were used: `value` is the measured whole-host figure, this is its per-core mean x cores (no "
            f"extrapolation)"), "wall_s_including_process_startup": wall,
        "sample": f"{P} single-threaded workers x {utts_per_core} x {utt_seconds:g} s utts (like decode.sh --nj {P}; utterance u = the GPU "
        f"batch's utterance u, same PCM16); aggregate = audio / slowest worker (process "
         f"start-up and the comparison excluded: sum of the three binaries' own run times); per stage over all workers: "
        f"reference compute-fbank-feats {audio / st[0]:.0f}x RT, reference nnet3-compute {audio / st[1]:.0f}x RT, reference "
        f"LatticeFasterDecoder::Decode {audio / st[2]:.0f}x RT "
         "(decoder/lattice-faster-decoder.cc compiled unmodified; FST containers from third_party/minifst because OpenFst is not vendored)"}
    par = None
    def summary(cs):
        nb_ = sum(c["best_path_identical"] for c in cs)
        bad = [c for c in cs if not c["best_path_identical"]]
        return {"utterances": len(cs), "best_path_identical": nb_, "best_path_identical_frac": nb_ / len(cs), "flip_utts": [c["utt"] for c in bad],
                "words_identical": sum(c["words_identical"] for c in cs), "raw_lattice_identical": sum(c["lattice_identical"] for c in cs),
             "max_abs_loglike_diff": max(c["max_abs_loglike_diff"] for c in cs),
                    "mean_of_max_abs_loglike_diff": float(np.mean([c["max_abs_loglike_diff"] for c in cs])),
                "max_best_cost_diff": max(c["best_cost_diff"] for c in cs), "raw_lattice_arcs_only_in_one_total": sum(c["arcs_only_in_one"] for c in cs),
             "raw_lattice_arcs_total": sum(c["ref_arcs"] for c in cs), "best_path_mismatches": [{"utt": c["utt"], "best_cost_diff": c["best_cost_diff"],
                     "max_abs_loglike_diff": c["max_abs_loglike_diff"]} for c in bad[:16]]}
    cmp_ = sorted((c for r in res for c in r[3]), key=lambda c: c["utt"])
    cmp2 = [c for r in res for c in r[4]]
    kept = {}
    for r in res: kept.update(r[5])
    if cmp_:
        par = summary(cmp_)
        par["max_abs_feature_diff"] = max(c.get("max_abs_feature_diff", 0.0) for c in cmp_)
        nfe = max(1, sum(c.get("feat_n", 0) for c in cmp_))
        par["mean_abs_feature_diff"] = sum(c.get("feat_sum", 0.0) for c in cmp_) / nfe
        par["feature_values_above_1e-4_frac"] = sum(c.get("feat_above", 0) for c in cmp_) / nfe
        par["feature_values_above_1e-4"] = sum(c.get("feat_above", 0) for c in cmp_)
        par["feature_values"] = nfe
        par["feature_truth"] = {"gpu_vs_exact_max_abs": max(c.get("feat_err_gpu_exact", 0.0) for c in cmp_),
                    "reference_vs_exact_max_abs": max(c.get("feat_err_ref_exact", 0.0) for c in cmp_),
                    "reference_values_above_1e-4_from_exact": sum(c.get("feat_ref_exact_above", 0) for c in cmp_),
            "note": "exact = the reference's formulas on the reference's own float32 tables (window, mel weights, pre-emphasis coefficient) "
            "with every operation on the samples in float64 "
             "(oracle/feat_oracle_path.inc, REAL = double).  k3_feat_kernel's data path is float64: it is the exact value rounded "
            "once to float32 (<= 2e-6 on log-mel < 32); what separates "
             "it from compute-fbank-feats is that binary's own float32 rounding (max_abs_feature_diff <= reference_vs_exact_max_abs " "+ gpu_vs_exact_max_abs)"}
        err2 = [c for c in cmp2 if "error" in c]
        ok2 = [c for c in cmp2 if "error" not in c]
        par["reference_vs_itself"] = dict(summary(sorted(ok2, key=lambda c: c["utt"])),
            second_run=f"{ALT_BLAS} for nnet3-compute (same features, same decoder)") if ok2 else {"error": err2[:1]}
        if ok2:
            par["gate"] = e2e_gate(par["flip_utts"], par["reference_vs_itself"]["flip_utts"], par["utterances"])
            par["gate_pass"] = par["gate"]["pass"]
        else: par["gate_pass"] = False
        par["note"] = ("reference chain = compute-fbank-feats -> nnet3-compute -> LatticeFasterDecoder (oracle/_ref binaries built from "
            "/root/reference) on the SAME PCM16 as the GPU batch; GPU chain = the timed path "
             "(k3_feat -> k3_nnet_forward -> k3_decoder literal_order=1).  best_path_identical: (transition-ids, words) of the "
            "tropical best path of the two raw lattices equal "
             "(decoder-wrappers.cc:322-331).  raw_lattice_identical: same states per frame and the same multiset of arcs (source "
            "frame, emitting/epsilon, ilabel, olabel); cost BITS cannot be "
             "equal because the log-likelihoods the two decoders consume differ (max_abs_loglike_diff).  Stage by stage the GPU is "
            "inside north_star's bounds (features <= 1e-4: "
             "max_abs_feature_diff; log-likelihoods <= 1e-4 on the same features; lattices bit-identical on the same "
            "log-likelihoods: tests/), but this 17-layer model amplifies a 1e-5 feature "
             "difference to ~1e-3 in its output and max-active pruning on its flat posteriors is chaotic, so the chains can end on "
            "different paths.  reference_vs_itself measures the "
             "reference's own reproducibility under the same size of float32 difference (its nnet3-compute on another MKL code "
            "path): the GPU chain is at parity when its rates match these.")
    return base, par, kept

TRAFFIC_CMD = ("cd /tmp && TMPDIR=/tmp rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -d <dir> -o pmc -- python bench.py --steps "
    "1 --warmup 0 --no-cpu-baseline --no-pipeline "
     "--no-two-pass --no-extras  (one pass per counter, no trace domains; Counter_Value of k3_decode_forward_literal_kernel summed "
    "over its dispatches / dispatches x 1024; " "`bench.py --measure-traffic` runs exactly this and tools/profile_round.sh commits its raw csv)")

def measure_traffic(args):
    """HBM-side traffic of the token-passing kernel from the PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
    one rocprofv3 --pmc pass per counter, no trace domains, over a one-step run of this script.  Returns {fetch,write,traffic}_bytes_per_launch or {"error": ...}."""
    import csv, glob, shutil
    if shutil.which("rocprofv3") is None: return {"error": "rocprofv3 not on PATH"}
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"k3_pmc_{c}_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1",
             "--warmup", "0", "--no-cpu-baseline", "--no-pipeline", "--no-two-pass", "--no-extras", "--utts", str(args.utts), "--utt-seconds",
                str(args.utt_seconds), "--graph-states", str(args.graph_states), "--graph-arcs", str(args.graph_arcs)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
            f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
            if r.returncode != 0 or not f: return {"error": f"rocprofv3 --pmc {c} failed: rc {r.returncode} {r.stderr[-300:]}"}
            per = {}
            for row in csv.DictReader(open(f[0])):
                if row.get("Counter_Name") == c and "k3_decode_forward_literal_kernel" in row["Kernel_Name"]:
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per: return {"error": f"no dispatch of the kernel in the {c} pass"}
            # (a decoder object launches the kernel on one lane for zero / one frame when it is created -- the InitDecoding / first-frame templates --: those dispatches, below
            # 1 % of the largest, are not launches of the workload)
            work = [v for v in per.values() if v >= 0.01 * max(per.values())]
            out[c] = sum(work) / len(work) * 1024.0
            keep = os.path.join(ROOT, "gpurun_out", "pmc_in_run")
            os.makedirs(keep, exist_ok=True)
            shutil.copy(f[0], os.path.join(keep, f"{c}_counter_collection.csv"))
        except Exception as e: return {"error": repr(e)}
        finally: shutil.rmtree(d, ignore_errors=True)
    return {"fetch_bytes_per_launch": out["FETCH_SIZE"], "write_bytes_per_launch": out["WRITE_SIZE"],
            "traffic_bytes_per_launch": out["FETCH_SIZE"] + out["WRITE_SIZE"]}

def main():
    if os.environ.get("K3HIP_LIB") and "--allow-dev-lib" not in sys.argv:
        raise SystemExit("bench.py measures the shipped kaldi_amd/lib/libk3hip.so only: unset K3HIP_LIB (developer override for profiling builds), "
                         "or pass --allow-dev-lib (the line then carries \"dev_lib\": the path, and is not a result)")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=512)
    ap.add_argument("--utt-seconds", type=float, default=10.0)
    ap.add_argument("--graph-states", type=int, default=2_000_000)
    ap.add_argument("--graph-arcs", type=int, default=5_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-two-pass", action="store_true", help="skip the second (order-independent decoder) measurement")
    ap.add_argument("--wait-whole-decoder", action="store_true",
        help="A/B: the next front end waits for the whole decoder call of two batches ago (token passing + pruning) instead of its " "token-passing launch")
    ap.add_argument("--no-pipeline", action="store_true",
        help="one stream: H2D, fbank, TDNN-F and decoder of a batch strictly after the previous batch (stage_ms then adds up to ms_per_step)")
    ap.add_argument("--one-decoder", action="store_true",
        help="one decoder object instead of two alternating ones (the default keeps two sets of lane pools -- 2 x 41 GB of the 288 GB at "
        "the bench configuration -- so that a batch's token passing starts under the previous batch's pruning kernel and lattice " "fetch)")
    ap.add_argument("--cpu-procs", type=int, default=0,
        help="cpu_baseline / e2e_parity: single-threaded reference workers (0 = one per host core: the baseline is then MEASURED on the "
        "whole host, no extrapolation)")
    ap.add_argument("--cpu-utts-per-core", type=int, default=0,
        help="cpu_baseline / e2e_parity: utterances per worker (0 = the batch spread over the workers, at least 2 each; utterance u = "
        "the GPU batch's utterance u while u < --utts)")
    ap.add_argument("--no-extras", action="store_true", help="skip the chain_objf / chain_train legs (for the record only; not part of `value`)")
    ap.add_argument("--measure-traffic", action="store_true",
        help="roofline.traffic measured in this run: two extra rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains) of "
        "`bench.py --steps 1 --warmup 0` as child processes (adds ~2 min)")
    ap.add_argument("--lattice-digest", action="store_true",
        help="decode_stats.lattice_digest: a hash of the canonical form of every raw lattice of the last timed batch")
    ap.add_argument("--truth-utts", type=int, default=64,
        help="e2e_parity.stage_gates.nnet_truth: utterances whose log-likelihoods are also evaluated in float64 on the host (1.8 s each)")
    ap.add_argument("--host-load-replicas", type=int, default=1,
        help="rehearsal of an N-GPU node's HOST load on one GPU: every batch's host tail (Connect + determinization) runs this many "
        "times concurrently, each copy on --det-threads threads "
         "(default: cores / replicas), i.e. what the box's cores see when this many ranks hand over lattices at the same rate; "
        "`value` then says whether the host keeps up")
    ap.add_argument("--det-threads", type=int, default=0, help="host threads of the determinization pool (0 = all cores / ranks)")
    ap.add_argument("--allow-dev-lib", action="store_true", help="accept a K3HIP_LIB developer build (A/B experiments); the line is marked \"dev_lib\"")
    ap.add_argument("--no-strict-rccl", action="store_true",
                    help="N > 1 over RCCL: if the graph broadcast through the product's C ABI (k3_comm_create + k3_fst_bcast) fails on any rank, fall back to "
                         "torch.distributed and only mark the line (\"rccl_abi_failed\": true).  Default (strict): mark the line, print it, and exit with status 3")
    ap.add_argument("--decoder-stream-priority", type=int, default=0,
                    help="queue priority of the two decoder streams (0 = default, -1 = high): whose workgroups take a slot that frees up, a waiting lane's or the next front end's")
    ap.add_argument("--fe-ahead", type=int, default=1, help="pipelined steps: how many batches the front end (H2D + fbank + TDNN-F) runs ahead of the decoder (1 or 2; 2 = three log-likelihood buffers)")
    ap.add_argument("--no-split-bf16", action="store_true", help="skip the split-bf16 record (value_split_bf16: the same steps with the TDNN-F products on the bf16 matrix core)")
    ap.add_argument("--ragged", action="store_true",
        help="SURVEY 8d's second set instead of the equal-length one: --utts utterances of length U(2 s, 20 s), seed 1235 (same model, graph and decoder); prints "
        "its own line (metric ... ragged), no cpu_baseline / extras.  The default run spawns this as a child at its end and reports the child's value as `value_ragged`")
    ap.add_argument("--no-ragged", action="store_true", help="default run: do not measure the ragged set (value_ragged) at the end")
    ap.add_argument("--ragged-sort", action="store_true", help="--ragged: the batch's utterances sorted by length, longest first (lane u = the u-th longest)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin this rank's host threads to its share of the cores (cores / ranks, by LOCAL_RANK)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local %= max(1, torch.cuda.device_count())       # (a 1-GPU box can rehearse the N > 1 path with K3_DIST_BACKEND=gloo: all ranks share the device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("K3_DIST_BACKEND", "nccl")     # "nccl" = RCCL over xGMI
        if backend == "nccl": dist.init_process_group("nccl", device_id=dev)
        else: dist.init_process_group(backend)
    import __graft_entry__ as ge
    if rank == 0: ge.build()
    if world > 1: dist.barrier()
    from kaldi_amd import feat, nnet3, synth, decoder, parallel, hostlib, lib as k3lib
    # provenance: the library this run maps was built from the sources next to it (its build id ends with their digest); a developer build (--allow-dev-lib) is only reported
    build_id, src_digest = k3lib.build_id(), k3lib.source_digest()
    if not build_id.endswith("+" + src_digest) and not args.allow_dev_lib:
        raise SystemExit(f"bench.py: kaldi_amd/lib/libk3hip.so ({build_id}) was not built from the sources in this tree (digest {src_digest}): run make -C kaldi_amd/csrc")

    U, nsamp = args.utts, int(16000 * args.utt_seconds)
    if args.ragged:      # lengths U(2 s, 20 s), seed 1235; the same lengths on every step (the audio under them moves with the step, as in the equal-length set)
        args.no_cpu_baseline = True
        args.no_extras = True
        args.no_two_pass = True
        lens = (np.random.default_rng(1235).uniform(2.0, 20.0, U) * 16000).astype(np.int64)
        if args.ragged_sort: lens = np.sort(lens)[::-1].copy()
    else:
        lens = np.full(U, nsamp, np.int64)
    lens = [int(x) for x in lens]
    samp_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tot_samp = int(samp_off[-1])
    max_seconds = max(lens) / 16000.0
    # host budget of a rank on an N-GPU node: cores / N threads for its lattice tail, and -- unless --no-pin -- those threads on the rank's OWN contiguous share of
    # the cores (by LOCAL_RANK; the affinity mask is inherited by every thread created from here on: the determinization pool, the lattice fetch helpers), so that
    # eight ranks' pools do not migrate over each other's cores.  One rank: the whole host, no pinning.
    ncpu = os.cpu_count() or 1
    det_threads = args.det_threads or max(1, ncpu // (world * max(1, args.host_load_replicas)))
    pinned_cores = None
    if world > 1 and not args.no_pin and hasattr(os, "sched_setaffinity"):
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        share = max(1, ncpu // max(1, lw))
        first = (int(os.environ.get("LOCAL_RANK", "0")) % max(1, lw)) * share
        try:
            os.sched_setaffinity(0, range(first, min(ncpu, first + share)))
            pinned_cores = [first, min(ncpu, first + share) - 1]
        except OSError:
            pinned_cores = None
    # synthetic workload: Gaussian PCM16 sigma 3000, utterance u of rank r = synth.gaussian_pcm16(nsamp, 1234 + 100000 r + u), in page-locked host
    # memory
    # (one spare utterance behind the batch: timed step k reads the batch from sample offset shift(k), so no two steps decode the same audio);
    # 17L-768/96-6024 TDNN-F, seed 1
    pcm_host = torch.empty(tot_samp + nsamp, dtype=torch.int16).pin_memory()
    pcm_np = pcm_host.numpy()
    def pcm_of(u, r=rank): return synth.gaussian_pcm16(lens[u] if u < U else nsamp, 1234 + 100000 * r + u)
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        for u, w in enumerate(ex.map(pcm_of, range(U + 1))): pcm_np[samp_off[u] if u < U else tot_samp:][:w.size] = w
    shift_of = lambda k: (k * 40009) % nsamp            # batch k of a run starts here (k = 0: the utterances as generated)
    pcm_dev = torch.empty(tot_samp, dtype=torch.int16, device=dev)
    sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    wo, fo, total_frames, fo_h = sf.offsets(lens, dev)
    model_path = os.path.join(tempfile.gettempdir(), f"k3_bench_tdnnf_{rank}.raw")
    # BatchNorm calibration on real fbank features of rank 0's first utterance (same model on every rank)
    # (always --utt-seconds of it, also for the ragged set: the SAME model in both sets)
    w0 = torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234).astype(np.float32)).to(dev)
    calib = sf.ComputeFeatures(w0, *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
    net_spec = synth.make_tdnnf(seed=1, calib_feats=calib)
    net_spec.write(model_path)
    net = nnet3.Nnet(model_path)
    num_pdfs = net.info.output_dim
    nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
    feats = torch.empty((total_frames, sf.dim), dtype=torch.float32, device=dev)
    loglikes = torch.empty((nb.total_out_rows, num_pdfs), dtype=torch.float32, device=dev)

    # decoding graph: built and uploaded on rank 0, broadcast once over RCCL/xGMI to the other ranks
    graph = None
    decs = {}
    if not args.no_decode:
        graph = synth.make_hclg(args.graph_states, args.graph_arcs, num_pdfs) if rank == 0 else None
        if world > 1: dist.barrier()      # (the ranks enter the graph exchange together: its deadlines do not run while rank 0 builds the graph)
        t0 = time.perf_counter()
        # N > 1 over RCCL: the graph travels through the PRODUCT's own C ABI (k3_comm_create + k3_fst_bcast: parallel.broadcast_graph_abi), in a worker thread
        # with a deadline, and
        # the ranks agree (one all-reduce) on whether every one of them got it; otherwise -- and in the gloo rehearsal on one device, where RCCL cannot put two
        # ranks on a GPU --
        # the same image goes through torch.distributed (parallel.broadcast_graph).  Either way before the timed region, once.
        cfst = None
        rccl_ranks = 0
        rccl_abi_failed = False
        rccl_abi_error = None
        bcast_via = "none (one rank)" if world == 1 else "torch.distributed broadcast of the k3_fst image"
        if world > 1 and dist.get_backend() == "nccl" and os.environ.get("K3_BENCH_ABI_BCAST", "1") == "1":
            import threading
            # a nonce of this run, agreed over the process group that already exists: the id file of an earlier run (same port, same launcher run id) cannot be
            # mistaken for this one's
            nonce = [os.urandom(8).hex() if rank == 0 else None]
            dist.broadcast_object_list(nonce, src=0)
            os.environ["K3_COMM_NONCE"] = nonce[0]
            box = {}
            id_file = os.path.join(tempfile.gettempdir(), "k3_bench_rccl_%s_%s.id" % (os.environ.get("MASTER_PORT", "0"), nonce[0]))
            def _abi():
                try:
                    torch.cuda.set_device(local)
                    box["r"] = parallel.broadcast_graph_abi(graph, synth.tid2pdf(num_pdfs), rank, world, id_file, timeout_s=45)
                except Exception as e: box["e"] = repr(e)
            th = threading.Thread(target=_abi, daemon=True)
            th.start()
            th.join(75)
            ok = torch.tensor([1 if "r" in box else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                cfst, rccl_ranks = box["r"]
                bcast_via = "k3_comm_create + k3_fst_bcast (RCCL through the C ABI)"
            else:
                # NOT silent: every rank's reason is collected, the line carries "rccl_abi_failed": true + the reasons, and unless
                # --no-strict-rccl the run ends with exit status 3 (after printing the line, measured over the torch.distributed fall-back, when the ABI call
                # returned an error; at once when it is still blocked inside RCCL -- nothing on this device can be trusted to make progress then)
                mine = box.get("e") or ("still blocked in k3_comm_create / k3_fst_bcast after 75 s" if th.is_alive() else None)
                reasons = [None] * world
                dist.all_gather_object(reasons, mine)
                rccl_abi_failed = True
                rccl_abi_error = {"rank %d" % r: e for r, e in enumerate(reasons) if e}
                bcast_via += " (k3_fst_bcast did not complete on every rank)"
                if not args.no_strict_rccl and any(e and e.startswith("still blocked") for e in reasons):
                    if rank == 0:
                        print(json.dumps({"metric": "RTFx (audio-s/wall-s) batched fbank -> TDNN-F -> HCLG lattice decode", "value": None, "n_gpus": world,
                                          "rccl_abi_failed": True, "rccl_abi_error": rccl_abi_error,
                                          "error": "graph broadcast through the C ABI (RCCL) timed out; --no-strict-rccl falls back to torch.distributed"}), flush=True)
                    os._exit(3)
        if cfst is None: cfst = parallel.broadcast_graph(graph, synth.tid2pdf(num_pdfs), rank, world, dev)
        t_bcast = time.perf_counter() - t0
        caps = dict(frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=int(4500 * max_seconds * 33.4) + 65536,
                lane_links_cap=int(6000 * max_seconds * 33.4) + 131072)
        for mode in (["literal"] if args.no_two_pass else ["literal", "two_pass"]):
            d = decoder.CudaDecoder(cfst, decoder.decoder_config(beam=BEAM, lattice_beam=LATTICE_BEAM, max_active=MAX_ACTIVE,
                    literal_order=1 if mode == "literal" else 0, **caps), U, num_pdfs)
            d.SetProfiling(True)
            decs[mode] = d
        # a second decoder object (another set of lane pools: 41 GB at the bench configuration; the GPU has 288): batch k + 1's token passing starts under batch
        # k's pruning kernel and lattice fetch
        if not args.one_decoder and not args.no_pipeline:
            decs["literal_b"] = decoder.CudaDecoder(cfst, decoder.decoder_config(beam=BEAM, lattice_beam=LATTICE_BEAM, max_active=MAX_ACTIVE, literal_order=1,
                     **caps), U, num_pdfs)
            decs["literal_b"].SetProfiling(True)
    hl = hostlib.load()
    det_opts = hostlib.DetOpts()
    hl.k3h_det_opts_default(ctypes.byref(det_opts))
    extra_pool = ThreadPoolExecutor(max(1, args.host_load_replicas - 1)) if args.host_load_replicas > 1 else None
    pool = ThreadPoolExecutor(1)      # hands a batch of lattices to the native worker pool (k3h_postprocess_batch runs det_threads threads itself)
    def postprocess(lats):
        """Connect + DeterminizeLatticePruned (word level, beam = lattice-beam: what the CUDA pipeline does with --determinize-lattice=true and no
        phone pass) of every lattice of the batch; returns (determinized states, arcs)"""
        n = len(lats)
        so = np.ascontiguousarray(lats.state_offsets, np.int64)
        ao = np.ascontiguousarray(lats.arc_offsets, np.int64)
        si, sf_, ai, af = lats._si, lats._sf, lats._ai, lats._af
        def one(replica=0):
            # --host-load-replicas R (the host side of an R-GPU node rehearsed on one GPU): replica r's worker threads run on ITS share of the cores, cores / R wide, like
            # rank r's would (the affinity of this calling thread is inherited by the threads k3h_postprocess_batch starts)
            if args.host_load_replicas > 1 and not args.no_pin and hasattr(os, "sched_setaffinity"):
                share = max(1, ncpu // args.host_load_replicas)
                try: os.sched_setaffinity(0, range(replica * share, min(ncpu, (replica + 1) * share)))
                except OSError: pass
            cs = np.zeros(n, np.int32)
            ca = np.zeros(n, np.int64)
            ok = np.zeros(n, np.int32)
            hostlib.check(hl.k3h_postprocess_batch(None, n, so.ctypes.data, ao.ctypes.data, int(cfst.start), si[0].ctypes.data, si[1].ctypes.data,
                    sf_[1].ctypes.data, ai[0].ctypes.data, ai[1].ctypes.data, ai[2].ctypes.data, ai[3].ctypes.data, af[0].ctypes.data, af[1].ctypes.data,
                        float(LATTICE_BEAM), ctypes.byref(det_opts), det_threads, None, cs.ctypes.data, ca.ctypes.data, ok.ctypes.data))
            return int(cs.sum()), int(ca.sum())
        # (--host-load-replicas: the other ranks' host tails, same lattices, results dropped)
        others = [extra_pool.submit(one, k + 1) for k in range(args.host_load_replicas - 1)] if extra_pool else []
        r = one(0)
        for o in others: o.result()
        return r

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    pipelined = not args.no_pipeline      # the next batch's front end (H2D + fbank + TDNN-F) on a second stream, queued behind the present batch's decoder
    front = torch.cuda.Stream(device=dev)
    # log-likelihood buffers: batch j's front end writes buffer j % NB; NB = 2 (the front end runs ONE batch ahead of the decoder) or 3 (--fe-ahead 2: two batches ahead, so that
    # decoder lanes and GEMM workgroups are both queued at every moment of a step)
    NB = 1 + max(1, min(2, args.fe_ahead)) if pipelined else 1
    fev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(max(2, NB))]
    dec_done = [torch.cuda.Event() for _ in range(max(2, NB))]
    ll2 = [loglikes] + [torch.empty_like(loglikes) for _ in range(NB - 1)] if pipelined else [loglikes, loglikes]
    NBI = NB if pipelined else 2
    def run(mode, steps, warmup, pipelined=pipelined, vary=True, keep=None):
        """W untimed + K timed steps of the whole path in one decoder mode.
        Returns (wall seconds of the K steps, per-stage ms, last lattice sizes, determinized sizes)."""
        dec = decs.get(mode)
        lat_sizes = [0, 0, None]
        det_sizes = [0, 0]
        pending = []
        nstep = [0]
        nser = [0]
        last = [None]
        src = lambda k: pcm_host[shift_of(k) if vary else 0:][:tot_samp]      # batch k's audio
        for e in dec_done: e.record()
        reader = [None] * max(2, NB)      # the decoder object that read log-likelihood buffer b last
        def front_end(k):      # batch k's H2D + fbank + TDNN-F on the front stream, into log-likelihood buffer k & 1 (last read by the decoder two batches ago)
            with torch.cuda.stream(front):
                # the buffer's last reader is the token-passing launch of batch k - 2, not the pruning / output kernels behind it: those cannot run beside the
                # resident launch of batch k - 1
                # (LDS) and finish ~50 ms into it, which is when this front end used to start (profiles/r04c_pipeline_overlap.txt)
                if reader[k % NBI] is not None and not args.wait_whole_decoder: reader[k % NBI].StreamWaitTokenPassing(front)
                else: front.wait_event(dec_done[k % NBI])
                fev[k % NBI][0].record()
                # (the waveform copy stays on this stream, in front of the features: issued on a copy stream of its own into a second device buffer, so that it runs under
                # the previous batch's decoder, the step got SLOWER -- 85 - 87 ms against 79 --: the DMA traffic beside the token-passing kernel costs that kernel more than
                # the 2.9 ms the copy takes; measured in rounds 2 and 6)
                pcm_dev.copy_(src(k), non_blocking=True)
                fev[k % NBI][1].record()
                sf.ComputeFeatures(pcm_dev, wo, fo, total_frames, out=feats)
                fev[k % NBI][2].record()
                nb.forward(feats, out=ll2[k % NBI])
                fev[k % NBI][3].record()
        def step_pipelined(timed):
            k = nstep[0]
            nstep[0] += 1
            if k == 0: front_end(0)
            torch.cuda.current_stream().wait_event(fev[k % NBI][3])
            if timed: ev[3].record()
            dec.DecodeBatch(ll2[k % NBI], nb.out_offsets)      # (one decoder object: its latest token-passing launch is batch k's, not the buffer's last reader)
            dec_done[k % NBI].record()
            front_end(k + 1)       # queued behind the decoder: its workgroups take the CUs the decoder's lanes leave as they finish
            if timed: ev[4].record()
            while len(pending) >= 2:
                r = pending.pop(0).result()
                det_sizes[0], det_sizes[1] = r
            lats = dec.GetRawLattices()
            last[0] = lats
            lat_sizes[0] = int(lats.state_offsets[-1])
            lat_sizes[1] = int(lats.arc_offsets[-1])
            if timed: ev[5].record()
            pending.append(pool.submit(postprocess, lats))
        two = "literal_b" in decs and mode == "literal" and pipelined
        facc = [0.0, 0.0, 0.0, 0]
        first_timed = [0]
        if two:
            pair = (dec, decs["literal_b"])
            dstr = (torch.cuda.Stream(device=dev, priority=args.decoder_stream_priority), torch.cuda.Stream(device=dev, priority=args.decoder_stream_priority))
        def fetch(j):      # batch j's lattices (compaction kernel + D2H on its decoder's stream), handed to the determinization pool
            while len(pending) >= 2:
                r = pending.pop(0).result()
                det_sizes[0], det_sizes[1] = r
            t_f = time.perf_counter()
            lats = pair[j & 1].GetRawLattices()
            last[0] = lats
            lat_sizes[0] = int(lats.state_offsets[-1])
            lat_sizes[1] = int(lats.arc_offsets[-1])
            kt = pair[j & 1].KernelTimes()
            facc[0] += kt[0]
            facc[1] += kt[1]
            facc[2] += (time.perf_counter() - t_f) * 1e3
            facc[3] += 1
            pending.append(pool.submit(postprocess, lats))
        def step_two(timed):
            k = nstep[0]
            nstep[0] += 1
            if k == 0:
                for j in range(NB - 1): front_end(j)
            with torch.cuda.stream(dstr[k & 1]):
                dstr[k & 1].wait_event(fev[k % NB][3])
                if timed: ev[3].record()
                pair[k & 1].DecodeBatch(ll2[k % NB], nb.out_offsets)
                dec_done[k % NB].record()
                reader[k % NB] = pair[k & 1]
                if timed: ev[4].record()
            front_end(k + NB - 1)      # (buffer (k + NB - 1) % NB: last read by batch k - 1's token passing)
            if k > first_timed[0]: fetch(k - 1)
            if timed: ev[5].record()
        def step(timed):
            if two: return step_two(timed)
            if pipelined and dec is not None: return step_pipelined(timed)
            if timed: ev[0].record()
            pcm_dev.copy_(src(nser[0]), non_blocking=True)      # first waveform byte leaves host memory
            nser[0] += 1
            if timed: ev[1].record()
            sf.ComputeFeatures(pcm_dev, wo, fo, total_frames, out=feats)
            if timed: ev[2].record()
            nb.forward(feats, out=loglikes)
            if timed: ev[3].record()
            if dec is not None:
                dec.DecodeBatch(loglikes, nb.out_offsets)
                if timed: ev[4].record()
                while len(pending) >= 2:      # the host buffers of batch k-2 are about to be reused
                    r = pending.pop(0).result()
                    det_sizes[0], det_sizes[1] = r
                lats = dec.GetRawLattices(copy=keep is not None)          # synchronises: compaction kernel + D2H of the pruned lattices
                if keep is not None: keep.append(lats)
                last[0] = lats
                lat_sizes[0] = int(lats.state_offsets[-1])
                lat_sizes[1] = int(lats.arc_offsets[-1])
                if timed: ev[5].record()
                pending.append(pool.submit(postprocess, lats))
        for _ in range(warmup): step(False)
        if two and nstep[0] > 0:      # (the warm-up's last batch; batch nstep's front end is already queued, as in the one-decoder steps)
            fetch(nstep[0] - 1)
            first_timed[0] = nstep[0]
            torch.cuda.synchronize()
            facc[:] = [0.0, 0.0, 0.0, 0]
        while pending: pending.pop(0).result()
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
        acc = np.zeros(8)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
            if not two: torch.cuda.current_stream().synchronize()
            if two:
                fe = fev[(nstep[0] - 1) % NBI]
                fe[3].synchronize()
                acc[0] += fe[0].elapsed_time(fe[1])
                acc[1] += fe[1].elapsed_time(fe[2])
                acc[2] += fe[2].elapsed_time(fe[3])
            elif pipelined and dec is not None:
                fe = fev[(nstep[0] - 1) % NBI]
                acc[0] += fe[0].elapsed_time(fe[1])
                acc[1] += fe[1].elapsed_time(fe[2])
                acc[2] += fe[2].elapsed_time(fe[3])
            else:
                acc[0] += ev[0].elapsed_time(ev[1])
                acc[1] += ev[1].elapsed_time(ev[2])
                acc[2] += ev[2].elapsed_time(ev[3])
            if dec is not None and not two:
                acc[3] += ev[3].elapsed_time(ev[4])
                acc[4] += ev[4].elapsed_time(ev[5])
                k = dec.KernelTimes()
                acc[5] += k[0]
                acc[6] += k[1]
        if two and nstep[0] > 0:
            fetch(nstep[0] - 1)      # the last batch's lattices belong to the timed region
            if facc[3]:      # (per fetched batch: kernel times from the decoder objects' own events, the fetch as the host saw it)
                acc[5] = facc[0] / facc[3] * steps
                acc[6] = facc[1] / facc[3] * steps
                acc[4] = facc[2] / facc[3] * steps
                acc[3] = acc[5] + acc[6]
        while pending:      # last lattice handed to the writer
            r = pending.pop(0).result()
            det_sizes[0], det_sizes[1] = r
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
        dt = time.perf_counter() - t0
        # identity of the last batch's raw lattices (canonical form: states by (frame, HCLG state), arcs sorted, cost bits included)
        if args.lattice_digest and last[0] is not None:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
            for lat in last[0]:
                for a in lat.canonical(): h.update(np.ascontiguousarray(a).tobytes())
            lat_sizes[2] = h.hexdigest()
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, acc / steps, lat_sizes, det_sizes

    mode0 = "literal" if decs else None
    # (run BEFORE the measured steps: the decoders' statistics and the last lattices the line reports are then the FP32 path's)
    # VERDICT r5 item 2, for the record only (`value` and `dtype` stay the FP32 matrix core's): the same pipelined steps with the TDNN-F products as six bf16 matrix-core products over
    # exactly split operands, the activations' planes written by the producing epilogue (k3_nnet_batch_set_precision(.., 2)); forwards of both split forms back to back
    sb = None
    if decs and not args.no_split_bf16 and not args.no_extras:
        sb_steps = max(3, min(args.steps, 6))
        try:
            nb.set_precision(2)
            sb_dt = run(mode0, sb_steps, 2)[0]
            sb = {"ms_per_step": 1000.0 * sb_dt / sb_steps, "steps": sb_steps}
            ll_ref = None
            for mode_, key_ in ((0, "forward_back_to_back_ms_fp32_mfma"), (1, "forward_back_to_back_ms_split_in_loader"), (2, "forward_back_to_back_ms_producer_planes")):
                nb.set_precision(mode_)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(2): nb.forward(feats, out=loglikes)
                e0.record()
                for _ in range(3): nb.forward(feats, out=loglikes)
                e1.record(); torch.cuda.synchronize()
                sb[key_] = e0.elapsed_time(e1) / 3
                if mode_ == 0: ll_ref = loglikes.clone()
                elif mode_ == 2: sb["loglike_max_abs_diff_vs_fp32_mfma"] = float((loglikes - ll_ref).abs().max())
            del ll_ref
        finally:
            nb.set_precision(0)
    if decs: [d_.FramePathCounts() for n_, d_ in decs.items() if n_.startswith("literal")]      # (reset)
    dt, acc, lat_sizes, det_sizes = run(mode0, args.steps, args.warmup)
    paths = decs["literal"].FramePathCounts() if decs else None
    if decs and "literal_b" in decs:      # (two alternating decoder objects: their counts together)
        pb = decs["literal_b"].FramePathCounts()
        for k_ in ("lds_path", "given_up", "general_path", "cycles_lds_path", "cycles_general_path"): paths[k_] += pb[k_]
        for k_, v_ in pb["give_up_reasons"].items(): paths["give_up_reasons"][k_] = paths["give_up_reasons"].get(k_, 0) + v_
    audio_s = tot_samp / 16000.0 * world * args.steps
    two = run("two_pass", args.steps, args.warmup) if "two_pass" in decs else None
    # Stage and kernel durations (stage_ms, roofline, roofline_gemm) come from a short SERIAL pass of the same objects: in the pipelined steps a kernel shares
    # the GPU with the other
    # batch's kernels and its event-to-event time is not its own duration (token passing 86 -> 105 ms, fbank 1 -> 64 ms).  `value` / `ms_per_step` are the
    # pipelined steps'.
    acc_pipe = acc
    if pipelined and decs:
        acc = run(mode0, 3, 1, pipelined=False)[1]
        if two is not None: two = (two[0], run("two_pass", 3, 1, pipelined=False)[1], two[2], two[3], two[1])
    # the TDNN-F forward by itself, back to back (no decoder in between: no clock transient after 90 ms of a low-power kernel)
    fwd_ms = None
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): nb.forward(feats, out=loglikes)
        e0.record()
        for _ in range(4): nb.forward(feats, out=loglikes)
        e1.record()
        torch.cuda.synchronize()
        fwd_ms = e0.elapsed_time(e1) / 4
    if rank == 0:
        gemm_tf = nb.flops / (acc[2] * 1e-3) / 1e12
        kernels_ms = acc[1] + acc[2] + acc[3]
        line = {"metric": "RTFx (audio-s/wall-s) batched fbank -> TDNN-F -> HCLG lattice decode" if decs else
            "RTFx (audio-s/wall-s) batched fbank + TDNN-F forward (decode disabled by --no-decode)", "value": audio_s / dt, "unit": "RTFx (audio-s/wall-s)",
             "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
             "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (Gaussian PCM16 sigma=3000 seed 1234+rank in page-locked host memory; random-init BN-calibrated "
            "17L-768/96-6024 TDNN-F seed 1; synthetic HCLG seed 4321)",
                    "config": {"workload": (
                    f"configs[2]: PCM16 H2D -> fbank(40) -> 17-layer TDNN-F (frame-subsampling 3) -> HCLG lattice decode (beam {BEAM:g}, "
                    f"lattice-beam {LATTICE_BEAM:g}, max-active {MAX_ACTIVE}, "
                     f"{args.graph_states} states / {args.graph_arcs} arcs) -> raw lattices to the host -> Connect + determinization on "
                    f"{det_threads} host threads, {U} x {args.utt_seconds:g} s utts per GPU"
                ) if decs else f"configs[1]: PCM16 H2D -> fbank(40) + 17-layer TDNN-F forward, {U} x {args.utt_seconds:g} s utts per GPU",
                "decoder_mode": "literal_order=1: raw lattices identical to the reference's LatticeFasterDecoder" if decs else None,
                "host_load_replicas": args.host_load_replicas, "det_threads": det_threads, "utts_per_gpu": U, "frames_per_utt": fo_h[1],
                    "output_rows": int(nb.total_out_rows), "params": int(net.info.num_params), "parallelism": f"utterance-shard x{world}"},
            "value_kernels": tot_samp / 16000.0 * world / (kernels_ms * 1e-3),
            "value_kernels_note": "audio / (fbank + TDNN-F + decode kernel time of a step): what the GPU stages alone sustain, H2D / D2H / host tail excluded",
             "pipeline": ("batch k+1's PCM16 H2D + fbank + TDNN-F are issued on a second stream right behind batch k's decoder kernels "
                "(double-buffered log-likelihoods): the copy and the start of the network run while the decoder's last lanes finish; "
                "two decoder objects alternate (unless --one-decoder), so batch k+1's token passing starts under batch k's pruning "
                "kernel, compaction and lattice copy; one of each per step inside the timed region; stage_ms are the stages' own "
                "durations and no longer add up to ms_per_step" if pipelined else "none (--no-pipeline): one stream, stage after stage"),
                "stage_ms": {"pcm16_h2d": acc[0], "fbank": acc[1], "nnet3": acc[2], "decode": acc[3], "decode.token_passing_kernel": acc[5],
                 "decode.lattice_prune_kernel": acc[6], "lattice_compact_and_d2h": acc[4]},
                "stage_ms_note": ("stage and kernel durations of a serial pass (3 steps, one stream) run after the timed steps: each kernel has the GPU "
                "to itself, as in the committed rocprofv3 traces (tools/profile_round.sh uses --no-pipeline); stage_ms_in_pipeline are "
                "the event-to-event times of the same stages inside the timed, pipelined steps, where they share the GPU"
                if (pipelined and decs) else "stages of the timed steps (one stream)"), "stage_ms_in_pipeline": ({"pcm16_h2d": acc_pipe[0],
                    "fbank": acc_pipe[1], "nnet3": acc_pipe[2], "decode": acc_pipe[3], "decode.token_passing_kernel": acc_pipe[5],
                    "decode.lattice_prune_kernel": acc_pipe[6], "lattice_compact_and_d2h": acc_pipe[4]} if (pipelined and decs) else None),
                "roofline_gemm": {"bound": "mfma", "kernel": "k3_tdnn_gemm_kernel (all launches of one forward)", "achieved": gemm_tf, "peak": 157.3,
                "unit": "TFLOP/s", "frac": gemm_tf / 157.3, "forward_back_to_back_ms": fwd_ms, "frac_back_to_back": nb.flops / (fwd_ms * 1e-3) / 1e12 / 157.3,
                 "note": "exact sum(2MNK) of the launched GEMMs / HIP-event time of the forward on the launch stream; FP32 MFMA peak (the only "
                "MFMA class inside the 1e-4 bound); achieved / frac: the forward as a stage of the serial pass (after the decoder); "
                "frac_back_to_back: four forwards in a row"}}
        fb_bytes = float(total_frames) * (160 * 2 + sf.dim * 4)      # the 160 new PCM16 samples a frame brings in + its output row
        line["roofline_feat"] = {"bound": "hbm", "kernel": "k3_feat_kernel (k3_feat_compute_batch_pcm16: window, FFT, mel, log in one launch)",
            "achieved": fb_bytes / (acc[1] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": fb_bytes / (acc[1] * 1e-3) / 1e9 / 8000.0,
             "algorithmic_bytes_per_launch": fb_bytes,
            "note": "480 B per frame (160 new 16-bit samples in, 40 floats out) / HIP-event time of the stage in the serial pass; the "
            "kernel is ~15 kflop per frame of FFT + mel work through LDS, 1 ms of a step"}
        if decs:
            dec = decs["literal"]
            info = dec.LatticeInfo()
            ab = dec.algorithmic_bytes(info)
            gbs = ab / (acc[5] * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "k3_decode_forward_literal_kernel (one launch = all frames of all lanes)", "achieved": gbs,
                "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None, "traffic_measured_in_run": False, "algorithmic_bytes_per_launch": ab,
                 "note": "algorithmic bytes (SURVEY 8d: 32 B/emitting arc traversed + 28 B/eps arc traversed + 16 B/token) from device "
                "counters / HIP-event time of the kernel on its launch stream; "
                 "bound by the per-frame chain of ~70 barrier-separated phases (each a few thousand cycles of dependent LDS / L2 "
                "accesses), not by bandwidth (roofline_latency is the bound that applies): frames of <= 1536 tokens run entirely in "
                "LDS (decode_stats.frames_by_path), the first frames of each utterance (the start state's thousands of arcs) run on "
                "the HBM-scratch path and move `traffic`; `traffic` is FETCH_SIZE + WRITE_SIZE as reported, and the calibration of "
                "profiles/hbm_counter_calibration_r04.json says how to read it for this pattern: a random 4 - 16 B read counts 64 "
                "B, a random 4 - 16 B write or atomic 32 B -- the counters tally REQUESTS at line granularity (traffic_fetch / 64 and "
                "traffic_write / 32 are the request counts), so traffic_over_algorithmic is the granule around narrow scattered accesses, "
                "not re-reading; DESIGN.md section 4"}
            # The HBM roofline above is the wrong yardstick for this kernel: a lane is a DEPENDENT CHAIN -- frame after frame, and inside a frame
            # the ~70 barrier-separated
            # phases of the reference's serial algorithm unrolled (cutoff, bound pass, accept pass, eps rounds, closure sub-graph, two hash-order passes, queue
            # order, component replay,
            # creation labels).  The cheapest such a phase gets on this hardware -- LDS read -> DPP scan -> LDS atomic -> workgroup barrier over 8 wavefronts --
            # is what the LDS-resident
            # frames of <= 512 tokens cost per phase: 141 k cycles / 70 (profiles/r04_literal_frames_by_size.txt).  floor = frames x phases x that / clock; all
            # lanes run in parallel.
            # (this kernel runs at the chip's 2.4 GHz engine clock: the slowest lane's frame cycles / the kernel time = 2.38, tools/prof_frames.py; the GEMMs,
            # power-bound, at 2.06)
            n_frames = int(info[:, 9].max())
            ph, cpp, ghz = 70, 2014.0, 2.4
            floor_ms = n_frames * ph * cpp / (ghz * 1e6)
            # (a frame is counted once: on the LDS path, or -- given up there or not -- on the general path)
            # (lane launches = frames counted / mean frames per lane: the same for equal lengths, right for the ragged set)
            lane_launches = max(1.0, (paths["lds_path"] + paths["general_path"]) / max(1.0, float(info[:, 9].mean())))
            mean_lane_ms = (paths["cycles_lds_path"] + paths["cycles_general_path"]) / lane_launches / (ghz * 1e6)
            line["roofline_latency"] = {"bound": "latency: the per-lane chain of frames x barrier-separated phases",
                "kernel": "k3_decode_forward_literal_kernel", "frames_per_lane": n_frames, "phases_per_frame": ph, "cycles_per_phase_floor": cpp,
                 "shader_clock_ghz": ghz, "floor_ms": floor_ms, "achieved_ms": acc[5], "frac": floor_ms / acc[5], "mean_lane_ms": mean_lane_ms,
                "frac_mean_lane": floor_ms / mean_lane_ms,
                "note": "floor = every frame at the cost of an LDS-resident frame of <= 512 tokens (70 phases x 2.0 k cycles, measured); "
                "achieved_ms = the kernel (its slowest lane, 512 lanes two to a CU), "
                 "mean_lane_ms = shader cycles per lane from the kernel's own counters / 2.4 GHz.  What separates them from the "
                "floor: frames above 512 tokens (cost grows ~0.28 k cycles per token), the "
                 "frames beyond the LDS path's 1536 tokens on HBM scratch (general_path_cycles_share of the cycles), the first ~12 "
                "frames of every utterance (3 - 25 k tokens on every lane at once: 27 % of the cycles, bound by the chip's "
                 "rate of scattered read-modify-write accesses rather than by latency) -- profiles/r04_literal_frames_by_size.txt, "
                "r04_literal_phase_profile_by_size.txt", "general_path_cycles_share": paths["cycles_general_path"] / max(1,
                        paths["cycles_lds_path"] + paths["cycles_general_path"])}
            line["roofline"]["traffic_command"] = TRAFFIC_CMD
            tr = measure_traffic(args) if (args.measure_traffic and world == 1) else None
            if tr and "traffic_bytes_per_launch" in tr:
                line["roofline"].update(traffic=tr["traffic_bytes_per_launch"], traffic_measured_in_run=True, traffic_fetch=tr["fetch_bytes_per_launch"],
                     traffic_write=tr["write_bytes_per_launch"],
                    traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (KB x 1024, per launch of the kernel)")
            else:
                if tr: line["roofline"]["traffic_error"] = tr.get("error")
                try:      # the committed PMC passes of the round (the same command, run by tools/profile_round.sh)
                    tj = json.load(open(sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "hbm_traffic_r*.json")))[-1]))
                    if U == 512 and args.utt_seconds == 10.0 and not args.ragged:
                        line["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
                        line["roofline"]["traffic_source"] = tj["source"]
                        for k_ in ("fetch_bytes_per_launch", "write_bytes_per_launch"):
                            if k_ in tj: line["roofline"]["traffic_" + k_.split("_")[0]] = tj[k_]
                except Exception: pass
            if line["roofline"].get("traffic"):
                line["roofline"]["traffic_over_algorithmic"] = line["roofline"]["traffic"] / ab
            line["decode_stats"] = {"graph_broadcast_s": t_bcast if world > 1 else 0.0, "graph_broadcast_via": bcast_via, "rccl_ranks": rccl_ranks,
                "rccl_abi_failed": rccl_abi_failed, "rccl_abi_error": rccl_abi_error, "pinned_cores_rank0": pinned_cores,
                 "emitting_arcs_traversed": int(info[:, 7].sum()), "eps_arcs_traversed": int(info[:, 8].sum()), "tokens": int(info[:, 4].sum()),
                    "links": int(info[:, 5].sum()), "max_tokens_on_a_frame": int(info[:, 6].max()), "lattice_states": lat_sizes[0],
                "lattice_arcs": lat_sizes[1], "lattice_digest": lat_sizes[2], "determinized_states": det_sizes[0], "determinized_arcs": det_sizes[1],
                 "reached_final_frac": float(info[:, 3].mean()), "algorithmic_bytes": ab, "order_sensitive_events": int(dec.OrderSensitiveEvents().sum()),
                 "frames_by_path": dict(paths,
                    note="frames of the warm-up + timed steps of this rank: lds_path = processed entirely in LDS (k3_decoder_fast.h: frames "
                    "of <= 1536 tokens), given_up = started there, exceeded a capacity and redone, general_path = the HBM-scratch path "
                    "(redone frames included)")}
            if two is not None:
                dt2, acc2, ls2, ds2 = two[:4]
                d2 = decs["two_pass"]
                info2 = d2.LatticeInfo()
                ab2 = d2.algorithmic_bytes(info2)
                line["value_two_pass"] = audio_s / dt2
                line["two_pass"] = {"note": "same pipeline with the order-independent decoder (literal_order = 0): faster, lattices close to but NOT "
                    "identical with the reference's at this configuration", "ms_per_step": 1000.0 * dt2 / args.steps, "stage_ms": {"pcm16_h2d": acc2[0],
                        "fbank": acc2[1], "nnet3": acc2[2], "decode": acc2[3], "decode.token_passing_kernel": acc2[5], "decode.lattice_prune_kernel": acc2[6],
                         "lattice_compact_and_d2h": acc2[4]}, "roofline": {"bound": "hbm", "kernel": "k3_decode_forward_kernel",
                            "achieved": ab2 / (acc2[5] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": ab2 / (acc2[5] * 1e-3) / 1e9 / 8000.0},
                     "lattice_states": ls2[0], "lattice_arcs": ls2[1], "lattice_digest": ls2[2],
                        "order_sensitive_upper_bound": int(d2.OrderSensitiveEvents().sum())}
            if sb is not None:
                line["value_split_bf16"] = tot_samp / 16000.0 * world / (sb["ms_per_step"] * 1e-3)
                line["split_bf16"] = dict(sb, note="NOT the headline (dtype stays f32): the pipelined steps with every tile-aligned TDNN-F product as six v_mfma_f32_32x32x16_bf16 "
                    "over exactly three-way split operands, the activations' three bf16 planes written by the producing epilogue (6 B per element next to the fp32 copy) so that the "
                    "loader only loads; as accurate as the FP32 matrix core against the float64 forward (tests/test_nnet_gpu.py), rounds differently from the reference.  Slower than "
                    "the FP32 matrix core here, and slower than splitting in the loader: both layer kinds lose the same 2.1 x, i.e. on the operand side -- 6 B per input element in "
                    "64-byte row pieces per plane instead of 4 B in 128-byte pieces (DESIGN.md Appendix A.2)")
        else:
            line["roofline"] = dict(line["roofline_gemm"], traffic=None)
        # the LF-MMI objective + derivatives of a training-sized minibatch, for the record (not part of `value`)
        if world == 1 and not args.no_extras:
            try:
                from kaldi_amd import chain
                cB, cT, cP = 128, 50, 4000
                den = synth.make_den_fst(3000, cP)
                g_ = chain.DenominatorGraph(den, cP)
                sup = chain.Supervision([synth.make_supervision_fst(cT, cP, seed=1000 + i, width=4) for i in range(cB)], cT, cP, 1.0)
                o_ = torch.randn(cT * cB, cP, device=dev) * 2.0
                d_ = torch.zeros_like(o_)
                x_ = torch.zeros_like(o_)
                copts = chain.ChainTrainingOptions(1e-5, 5e-5)
                chain.ComputeChainObjfAndDeriv(copts, g_, sup, o_, d_, x_)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): objf_ = chain.ComputeChainObjfAndDeriv(copts, g_, sup, o_, d_, x_)
                torch.cuda.synchronize()
                line["chain_objf"] = {"ms_per_minibatch": (time.perf_counter() - t0) / 5 * 1e3, "objf_per_frame": objf_[0] / objf_[2],
                    "config": f"k3_chain_objf_and_deriv (ComputeChainObjfAndDeriv: denominator + numerator + l2 + xent derivative): {cB} "
                    f"sequences x {cT} frames, 3000-state / {int(den.arc_offsets[-1])}-transition denominator graph, {cP} pdfs"}
                del g_, sup, o_, d_, x_
            except Exception as e: line["chain_objf"] = {"error": repr(e)}
        exe_train = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-train")
        # SURVEY 8f row 4, for the record (not part of `value`): whole training iterations of THIS model -- the reference's unmodified nnet3 objects over the
        # CuMatrix adapter
        if world == 1 and not args.no_extras and os.path.exists(exe_train):
            try:
                import struct
                td = tempfile.mkdtemp(prefix="k3_train_")      # context of the 17-layer model: 1 + the sum of the TDNN-F strides
                tB, tT, tP, ts = 64, 50, num_pdfs, 3
                ctx = 40
                synth.make_tdnnf(seed=1, calib_feats=calib, orthonormal_constraint=-1.0).write(f"{td}/m.raw")
                rng = np.random.default_rng(7)
                m = np.ascontiguousarray(rng.standard_normal((((tT - 1) * ts + 1 + 2 * ctx) * tB, 40)) * 1.2 + 16.5, "<f4")
                open(f"{td}/in.mat", "wb").write(b"\0BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.tobytes())
                den = synth.make_den_fst(3000, tP)
                fsts = [synth.make_supervision_fst(tT, tP, seed=300 + i) for i in range(tB)]
                merged = synth.merge_supervision_fsts(fsts)
                fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel,
                    np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() + np.ascontiguousarray(f.weight,
                    np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
                so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32)
                ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
                with open(f"{td}/chain.spec", "wb") as fh:
                    fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), tP, tB, tT, merged.num_states,
                                int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
                    fh.write(fb(den))
                    fh.write(fb(merged))
                    fh.write(so.tobytes())
                    fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
                    for k_, dt_ in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)):
                        fh.write(np.concatenate([getattr(f, k_) for f in fsts]).astype(dt_).tobytes())
                r = subprocess.run([exe_train, f"{td}/m.raw", str(ts), f"{td}/in.mat", f"{td}/chain.spec", "24", "0.001", "0.0", f"{td}/out.raw",
                         f"{td}/out.vec"], capture_output=True, text=True, timeout=300, env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"))
                its = [float(l.rsplit("; ", 1)[1].split()[0]) for l in r.stderr.splitlines() if "iteration" in l and l.rstrip().endswith("ms")]
                gf = [float(l.split(": ")[-1].split()[0]) for l in r.stderr.splitlines() if "GFLOP in matrix products" in l]
                # the preconditioners (OnlineNaturalGradient) refresh their factors on every one of the first 10 minibatches and on every 4th after that
                # (host-side eigen-problems): the
                # steady state is the mean over whole 4-iteration cycles from iteration 12 on; the 2nd iteration (all of them refreshing) is what the CPU
                # reference's 2nd iteration is compared with
                line["chain_train"] = ({"ms_per_iteration": float(np.mean(its[12:24])),
                        "ms_per_iteration_preconditioners_refreshing": float(np.mean(its[2:10])),
                            "ms_per_iteration_between_refreshes": float(np.median(its[12:24])), "first_iteration_ms": its[0], "iterations": len(its),
                        "config": f"kaldi_amd/adapter/nnet3-chain-train.cc (NnetChainTrainer::TrainInternal's sequence: forward, "
                        f"k3_chain_objf_and_deriv, backward with natural-gradient updates, max-change, orthonormal constraint): the "
                        f"benchmark model, {tB} sequences x {tT} output frames, 3000-state denominator graph"
                        } if r.returncode == 0 and len(its) >= 24 else {"error": (r.stderr or "")[-300:]})
                if "error" not in line["chain_train"] and len(gf) >= 24:
                    tf = float(np.mean(gf[12:24])) * 1e9 / (line["chain_train"]["ms_per_iteration"] * 1e-3) / 1e12
                    line["chain_train"]["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                         "gflop_per_iteration": float(np.mean(gf[12:24])),
                        "note": "2MNK over every AddMatMat of an iteration (forward, both backward products, the natural-gradient "
                        "preconditioner's Gram products) / the iteration's wall time, "
                         "everything else (element-wise kernels, reductions, the LF-MMI objective, the host-side eigen-problems) "
                        "included in the time: the TDNN layers run one GEMM per time offset like the reference's cudamatrix path, " "not a fused kernel"}
                ref_train = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-train")
                # the reference's own CPU training iteration beside it (one core, MKL sequential)
                if "error" not in line["chain_train"] and os.path.exists(ref_train) and not args.no_cpu_baseline:
                    rr = subprocess.run([ref_train, f"{td}/m.raw", str(ts), f"{td}/in.mat", f"{td}/chain.spec", "2", "0.001", "0.0", f"{td}/ref.raw",
                             f"{td}/ref.vec"], capture_output=True, text=True, timeout=600, env=dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle",
                                 "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1"))
                    rits = [float(l.rsplit("; ", 1)[1].split()[0]) for l in rr.stderr.splitlines() if "iteration" in l and l.rstrip().endswith("ms")]
                    if rr.returncode == 0 and rits:
                        line["chain_train"]["cpu_reference"] = {"ms_per_iteration": rits[-1], "cores": 1, "kind": "reference", "gpu_ms_same_iteration": its[1],
                             "note": "the same driver over the reference's CPU matrices and chain code "
                            "(oracle/_ref/bin/ref-nnet3-chain-train), second iteration (every preconditioner refreshing; the GPU's "
                            "second iteration beside it)"}
                import shutil
                shutil.rmtree(td, ignore_errors=True)
            except Exception as e: line["chain_train"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                if graph is None: graph = synth.make_hclg(args.graph_states, args.graph_arcs, num_pdfs)
                gpu = None
                # the end-to-end gate: one more (serial) pass over the batch as generated (shift 0), its raw lattices and log-likelihoods kept for the
                # comparison
                if decs:
                    keep = []
                    run("literal", 1, 0, pipelined=False, vary=False, keep=keep)
                    gpu = (keep[0], loglikes.cpu().numpy(), np.asarray(nb.out_offsets), U, feats.cpu().numpy(), np.asarray(fo_h))
                P_ = args.cpu_procs or (os.cpu_count() or 1)
                upc_ = args.cpu_utts_per_core or max(2, -(-U // min(P_, os.cpu_count() or 1)))
                line["cpu_baseline"], par, kept = cpu_baseline(model_path, graph, num_pdfs, args.utt_seconds, pcm_of, gpu, utts_per_core=upc_, max_procs=P_)
                if par is not None:
                    line["e2e_parity"] = par
                    # The stage gates at the bench's own scale, on the REFERENCE's intermediate results (so that each stage is judged on identical inputs):
                    # N: k3_nnet_forward on the reference's features vs the reference's nnet3-compute;  D: k3_decoder on the reference's log-likelihoods vs the
                    # reference's lattices
                    us = sorted(kept)      # (utterances outside the sample keep the GPU's own features: the network is planned for the whole batch)
                    rf = gpu[4].copy()
                    oo = np.asarray(nb.out_offsets)
                    for u in us: rf[fo_h[u]:fo_h[u + 1]] = kept[u][0]
                    g_ll = nb.forward(torch.from_numpy(rf).to(dev))
                    torch.cuda.synchronize()
                    g_llh = g_ll.cpu().numpy()
                    nd = max(float(np.abs(g_llh[oo[u]:oo[u + 1]] - kept[u][1]).max()) for u in us)
                    # gate N against the exact value: the same network evaluated in float64 (oracle/nnet3_oracle.py, the checker) on the reference's features,
                    # for a bounded sample of the
                    # utterances (1.8 s each); where the largest difference to nnet3-compute sits
                    from oracle import nnet3_oracle as no
                    onet = no.read_nnet(model_path)
                    tu = us[:args.truth_utts]
                    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
                        tr64 = list(ex.map(lambda u: no.compute(onet, kept[u][0], 3, dtype=np.float64), tu))
                    # |k3_nnet_forward - nnet3-compute| over ALL log-likelihoods of the sample as a distribution, not only its maximum
                    ad = np.concatenate([np.abs(g_llh[oo[u]:oo[u + 1]] - kept[u][1]).ravel() for u in us])
                    nn_dist = {"values": int(ad.size), "max": float(ad.max()), "mean": float(ad.mean()),
                               "above_1e-4_frac": float((ad > 1e-4).mean()), "above_1e-4": int((ad > 1e-4).sum()),
                               "above_2e-4_frac": float((ad > 2e-4).mean()), "above_2e-4": int((ad > 2e-4).sum()),
                               "p99": float(np.quantile(ad, 0.99)), "p99.9": float(np.quantile(ad, 0.999)), "p99.999": float(np.quantile(ad, 0.99999))}
                    del ad
                    n_eg = max(float(np.abs(g_llh[oo[u]:oo[u + 1]] - t).max()) for u, t in zip(tu, tr64))
                    n_er = max(float(np.abs(kept[u][1] - t).max()) for u, t in zip(tu, tr64))
                    n_mg = float(np.mean([np.abs(g_llh[oo[u]:oo[u + 1]] - t).mean() for u, t in zip(tu, tr64)]))
                    n_mr = float(np.mean([np.abs(kept[u][1] - t).mean() for u, t in zip(tu, tr64)]))
                    uw = max(us, key=lambda u: float(np.abs(g_llh[oo[u]:oo[u + 1]] - kept[u][1]).max()))
                    dw = np.abs(g_llh[oo[uw]:oo[uw + 1]] - kept[uw][1])
                    iw = np.unravel_index(int(dw.argmax()), dw.shape)
                    worst = {"utt": int(uw), "output_row": int(iw[0]), "pdf": int(iw[1]), "gpu": float(g_llh[oo[uw] + iw[0], iw[1]]),
                                "reference": float(kept[uw][1][iw]), "exact": (float(no.compute(onet, kept[uw][0], 3, dtype=np.float64)[iw]))}
                    rl = np.concatenate([kept[u][1] for u in us])      # the sample's utterances, lane k = utterance us[k]
                    ro_k = np.concatenate([[0], np.cumsum([kept[u][1].shape[0] for u in us])])
                    dec = decs["literal"]
                    dec.DecodeBatch(torch.from_numpy(rl).to(dev), ro_k)
                    glats_k = dec.GetRawLattices(copy=True)
                    glats = {u: glats_k[k] for k, u in enumerate(us)}
                    # every state (frame, final-cost bits) and every arc (source frame, emitting / epsilon, labels, graph- and acoustic-cost BITS), as
                    # multisets: identity up to the names of the states
                    def same_lattice(u):
                        r, l = kept[u][2], glats[u]
                        bits = lambda x: (np.asarray(x, np.float32) + np.float32(0)).view(np.int32).astype(np.int64)
                        ka = np.stack([r["frame"][r["src"]], r["frame"][r["dst"]], r["ilabel"], r["olabel"], bits(r["graph"]), bits(r["ac"])], 1)
                        kb = np.stack([l.st_frame[l.arc_src], l.st_frame[l.arc_dst], l.arc_ilabel, l.arc_olabel, bits(l.arc_graph), bits(l.arc_ac)], 1)
                        sa = np.stack([r["frame"], bits(r["final_graph"])], 1)
                        sb = np.stack([l.st_frame, bits(l.st_final)], 1)
                        srt = lambda m: m[np.lexsort(m.T[::-1])]
                        return ka.shape == kb.shape and sa.shape == sb.shape and np.array_equal(srt(ka), srt(kb)) and np.array_equal(srt(sa), srt(sb))
                    ident = sum(bool(same_lattice(u)) for u in us)
                    # The controls that separate the causes of the end-to-end flips:
                    # (i) the reference's FEATURES through the GPU network and decoder -- what remains when the feature difference is taken away;
                    # (ii) the GPU's FEATURES through the reference's network and decoder -- what the feature difference alone does to the reference chain
                    ctrl = {}
                    try:
                        rl2 = np.concatenate([g_llh[oo[u]:oo[u + 1]] for u in us])
                        dec.DecodeBatch(torch.from_numpy(rl2).to(dev), ro_k)
                        gl2 = dec.GetRawLattices(copy=True)
                        f_i = [u for k, u in enumerate(us)
                               if not _compare(_view_ref(kept[u][2]), kept[u][1], _view_raw(gl2[k]), g_llh[oo[u]:oo[u + 1]])["best_path_identical"]]
                        Pc = min(len(us), args.cpu_procs or (os.cpu_count() or 1))
                        jobs_c = [([(u, gpu[4][fo_h[u]:fo_h[u + 1]], kept[u]) for u in us[w::Pc]], model_path, graph, num_pdfs) for w in range(Pc)]
                        with ThreadPoolExecutor(Pc) as ex: f_ii = sorted(u for r_ in ex.map(_ctrl_worker, jobs_c) for u in r_)
                        slf_f = par["reference_vs_itself"].get("flip_utts", [])
                        ctrl = {"utterances": len(us), "end_to_end_flips": par["flip_utts"], "reference_vs_itself_flips": slf_f,
                                "reference_features_gpu_net_gpu_decoder": {"flips": f_i, "gate": e2e_gate(f_i, slf_f, len(us))},
                                "gpu_features_reference_net_reference_decoder": {"flips": f_ii, "gate": e2e_gate(f_ii, slf_f, len(us))},
                                "note": "flips = utterances whose best path (transition-ids and words) differs from the all-reference chain's.  (i) isolates the network "
                                        "+ decoder (the decoder is bit-identical on identical log-likelihoods: decoder_on_reference_loglikes_lattices_identical), (ii) "
                                        "isolates the features: the float64 feature path is the exact value, compute-fbank-feats is up to 1.1e-4 from it, and the 17 "
                                        "layers amplify that ~20x -- the reference's own chain moves by that much when fed the exact features"}
                    except Exception as e: ctrl = {"error": repr(e)}
                    par["controls"] = ctrl
                    par["stage_gates"] = {"utterances": len(us), "features_max_abs_diff": par["max_abs_feature_diff"],
                            "features_mean_abs_diff": par["mean_abs_feature_diff"], "features_above_1e-4_frac": par["feature_values_above_1e-4_frac"],
                         "nnet_on_reference_features_max_abs_loglike_diff": nd, "nnet_on_reference_features_loglike_diff_distribution": nn_dist, "nnet_truth": {"utterances": len(tu), "gpu_vs_exact_max_abs": n_eg,
                             "reference_vs_exact_max_abs": n_er, "gpu_vs_exact_mean_abs": n_mg, "reference_vs_exact_mean_abs": n_mr,
                             "largest_difference_to_reference": worst,
                            "note": "exact = the same network in float64 (numpy) on the reference's features; k3_nnet_forward and nnet3-compute "
                            "are two float32 evaluations of it (17 layers, K up to 2304 per product): the gate is that the GPU is no "
                            "further from the exact value than the reference's own binary is"}, "decoder_on_reference_loglikes_lattices_identical": ident,
                         "features_gpu_vs_exact_max_abs": par["feature_truth"]["gpu_vs_exact_max_abs"],
                            "features_reference_vs_exact_max_abs": par["feature_truth"]["reference_vs_exact_max_abs"],
                        "note": "F: k3_feat (float64 data path) vs compute-fbank-feats on the same PCM16: the kernel is the exact value of "
                        "the reference's formulas rounded once (features_gpu_vs_exact_max_abs), the binary is up to ~1.1e-4 from it "
                        "over the batch's 2e7 values (features_reference_vs_exact_max_abs; a handful of low-mel-bin values where "
                        "pre-emphasis leaves 1e-3 of the frame's power), so features_max_abs_diff IS the reference's own rounding "
                        "error; N: (the reference against itself on MKL's other branch: "
                        "e2e_parity.reference_vs_itself.max_abs_loglike_diff); k3_nnet_forward vs nnet3-compute on the reference's "
                        "features; D: k3_decoder (literal_order) vs "
                         "LatticeFasterDecoder on the reference's log-likelihoods -- states and arcs with all cost bits, as "
                        "multisets (the strict signature test is tests/test_decoder_literal_gpu.py)"}
            except Exception as e: line["cpu_baseline"] = {"error": repr(e)}
        if args.ragged:
            fr = np.diff(np.asarray(nb.out_offsets))
            line["metric"] += " -- RAGGED set (SURVEY 8d: utterance lengths U(2 s, 20 s), seed 1235)"
            line["config"]["workload"] = line["config"]["workload"].replace(f"{U} x {args.utt_seconds:g} s utts per GPU",
                f"{U} utts of U(2 s, 20 s) (seed 1235{', sorted longest first' if args.ragged_sort else ', in generation order'}) per GPU = {tot_samp / 16000.0:.0f} s of audio")
            line["ragged"] = {"utt_seconds": {"min": min(lens) / 16000.0, "mean": tot_samp / 16000.0 / U, "max": max(lens) / 16000.0},
                              "decoder_frames_per_lane": {"min": int(fr.min()), "mean": float(fr.mean()), "max": int(fr.max())},
                              "token_passing_kernel_ms_serial": line["stage_ms"]["decode.token_passing_kernel"],
                              "mean_lane_ms": line.get("roofline_latency", {}).get("mean_lane_ms"),
                              "note": "one persistent workgroup per utterance: a launch lasts as long as its longest lane (token_passing_kernel_ms_serial vs mean_lane_ms); in the "
                                      "pipelined steps the CUs the short lanes free are taken by the next batch's front end and, through the second decoder object, by the next "
                                      "batch's lanes"}
        # SURVEY 8d's second set, measured by a child process of the same program once this process's decoders are gone (two sets of lane pools sized for 20 s utterances
        # do not fit beside the three this run holds): `value_ragged`
        if decs and world == 1 and not args.ragged and not args.no_ragged:
            try:
                import gc
                decs.clear()
                dec = d = d2 = d_ = None
                gc.collect()
                torch.cuda.empty_cache()
                cmd = [sys.executable, os.path.abspath(__file__), "--ragged", "--steps", str(min(args.steps, 10)), "--warmup", str(min(args.warmup, 2)), "--utts", str(U),
                       "--graph-states", str(args.graph_states), "--graph-arcs", str(args.graph_arcs)] + (["--allow-dev-lib"] if args.allow_dev_lib else [])
                rr = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                cl = json.loads([l for l in rr.stdout.splitlines() if l.startswith("{")][-1])
                line["value_ragged"] = cl["value"]
                line["ragged"] = dict(cl.get("ragged", {}), ms_per_step=cl["ms_per_step"], steps=cl["steps"], value_over_equal_length_value=cl["value"] / line["value"],
                                      command=" ".join(cmd[1:]))
            except Exception as e: line["ragged"] = {"error": repr(e)}
        line["provenance"] = {"libk3hip_build_id": build_id, "source_digest": src_digest, "match": build_id.endswith("+" + src_digest),
                              "note": "k3_build_id() of the mapped library vs the SHA-256 prefix of kaldi_amd/csrc/*.hip, *.h + include/k3hip.h as shipped (kaldi_amd/lib.py)"}
        if os.environ.get("K3HIP_LIB"): line["dev_lib"] = os.environ["K3HIP_LIB"]
        if decs and rccl_abi_failed: line["rccl_abi_failed"] = True      # (top level too: a reader of `value` cannot miss it)
        print(json.dumps(line), flush=True)
    pool.shutdown()
    if world > 1: dist.destroy_process_group()
    if decs and rccl_abi_failed and not args.no_strict_rccl:
        sys.exit(3)      # the product's RCCL path failed: the number above was measured over the torch.distributed fall-back and says so

if __name__ == "__main__":
    main()
