"""GPU: the end-to-end parity gate of SURVEY 8d (gate 4, "at WER parity") at BASELINE configs[2] shapes, the way bench.py reports it.
bench.py's own cpu_baseline leg runs the REFERENCE chain -- compute-fbank-feats -> nnet3-compute -> LatticeFasterDecoder (oracle/_ref, built
from /root/reference) -- on the same PCM16 as the GPU batch and compares, utterance by utterance, the two chains' raw lattices: best path
(transition-ids and words), lattice structure, and the log-likelihoods the two decoders consumed.  The bar: log-likelihoods within 1e-4
on the same features and lattices bit-identical on the same log-likelihoods are the stage gates (other tests); end to end the 17-layer model turns a 1e-5
feature difference into ~1e-3 and max-active pruning on flat posteriors is chaotic, so the bar here is the REFERENCE'S OWN reproducibility: its chain
run a second time with nnet3-compute on another MKL code path (e2e_parity.reference_vs_itself) -- the GPU chain must agree with the reference
as well as the reference agrees with itself."""
import json, os, subprocess, sys
import pytest
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_bench_line_carries_the_end_to_end_parity_gate():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "bin", "ref-lattice-decoder")): pytest.skip("oracle/_ref not built (needs /root/reference once; it travels to the GPU box)")
    procs = min(32, os.cpu_count() or 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-two-pass", "--no-extras", "--cpu-procs", str(procs), "--cpu-utts-per-core", "4"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" not in line["cpu_baseline"], line["cpu_baseline"]
    par = line["e2e_parity"]; os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); json.dump(par, open(os.path.join(ROOT, "gpurun_out", "e2e_parity_test.json"), "w"), indent=1)
    assert par["utterances"] == procs * 4
    # stage gate F on the bench's own audio: the kernel's data path is float64, so it must BE the exact value of the reference's formulas (rounded once to float32), and its
    # distance to compute-fbank-feats must be that binary's own float32 rounding error (<= ~1.1e-4 on a few low-mel-bin values out of 2e7, measured on 512 utterances) and nothing more
    ft = par["feature_truth"]
    assert ft["gpu_vs_exact_max_abs"] <= 2e-6, ft
    assert par["max_abs_feature_diff"] <= ft["reference_vs_exact_max_abs"] + 2e-6 and ft["reference_vs_exact_max_abs"] <= 1.5e-4, (par["max_abs_feature_diff"], ft)
    assert par["feature_values_above_1e-4"] <= ft["reference_values_above_1e-4_from_exact"] + 2 and par["feature_values_above_1e-4_frac"] <= 1e-6 and par["mean_abs_feature_diff"] <= 5e-6, par
    sg = par["stage_gates"]
    assert sg["decoder_on_reference_loglikes_lattices_identical"] == sg["utterances"] == par["utterances"], sg      # gate D at the bench configuration, on the reference's own log-likelihoods
    # gate N at the bench's scale: two float32 evaluations of a 17-layer network cannot be asked to agree to 1e-4 (the reference's own binary moves by 8.9e-4 between two MKL code paths and is
    # ~7e-4 from the float64 value); asserted instead: k3_nnet_forward is no further from the float64 evaluation of the network than nnet3-compute is, in the maximum and in the mean
    nt = sg["nnet_truth"]
    assert nt["gpu_vs_exact_max_abs"] <= nt["reference_vs_exact_max_abs"] + 5e-5 and nt["gpu_vs_exact_mean_abs"] <= 1.05 * nt["reference_vs_exact_mean_abs"], nt
    assert sg["nnet_on_reference_features_max_abs_loglike_diff"] <= max(1e-4, 0.5 * par["reference_vs_itself"]["max_abs_loglike_diff"]), sg      # ... and inside the reference's own BLAS-path spread
    slf = par["reference_vs_itself"]; assert "error" not in slf, slf
    # The two chains' log-likelihoods differ by what the reference's own feature rounding error becomes behind 17 layers (~1e-3), so the bar is the reference's own
    # reproducibility under a float32 difference of that size (its nnet3-compute on another MKL code path, same features, same decoder):
    assert slf["max_abs_loglike_diff"] > 0, "the second reference run did not take another code path: no yardstick"
    assert par["mean_of_max_abs_loglike_diff"] <= 3.0 * slf["mean_of_max_abs_loglike_diff"] + 1e-4, (par["mean_of_max_abs_loglike_diff"], slf["mean_of_max_abs_loglike_diff"])
    # the gate itself: bench.e2e_gate -- an exact one-sided McNemar test on the paired per-utterance best-path flips (GPU chain vs reference) against (reference vs itself),
    # alpha = 0.01; the same function gives the same verdict on this 128-utterance sample and on the driver's 512 (round 5's threshold `self - 1 / utterances` did not)
    sys.path.insert(0, ROOT)
    import bench
    gate = bench.e2e_gate(par["flip_utts"], slf["flip_utts"], par["utterances"])
    assert par["gate"] == gate and par["gate_pass"] is True and gate["pass"], (par.get("gate"), gate)
    assert len(par["flip_utts"]) == par["utterances"] - par["best_path_identical"]
    ctrl = par["controls"]; assert "error" not in ctrl, ctrl      # the two controls that separate the causes ran and carry their own gate
    assert ctrl["reference_features_gpu_net_gpu_decoder"]["gate"]["pass"], ctrl      # without the feature difference the GPU net + decoder is inside the reference's own spread
    dd = sg["nnet_on_reference_features_loglike_diff_distribution"]      # the whole distribution of |k3_nnet_forward - nnet3-compute| on the reference's features, not only its maximum
    assert dd["values"] >= 1e8 and dd["above_2e-4_frac"] <= 1e-6 and dd["above_1e-4_frac"] <= 1e-3 and dd["p99.9"] <= 1.5e-4 and dd["mean"] <= 5e-5, dd
    assert line["roofline_feat"]["frac"] > 0 and line["cpu_baseline"]["extrapolated_all_cores"] > line["cpu_baseline"]["value"] * 0.5
