"""GPU: the end-to-end parity gate of SURVEY 8d (gate 4, "at WER parity") at BASELINE configs[2] shapes, the way bench.py reports it.
bench.py's own cpu_baseline leg runs the REFERENCE chain -- compute-fbank-feats -> nnet3-compute -> LatticeFasterDecoder (oracle/_ref, built
from /root/reference) -- on the same PCM16 as the GPU batch and compares, utterance by utterance, the two chains' raw lattices: best path
(transition-ids and words), lattice structure, and the log-likelihoods the two decoders consumed.  The bar: log-likelihoods within 1e-4
(north_star), best paths identical on >= 99.9 % of the utterances or every difference a tie inside the log-likelihood tolerance."""
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
    assert par["max_abs_loglike_diff"] <= 1e-4, par
    frames = 334
    ties = all(m["best_cost_diff"] <= frames * par["max_abs_loglike_diff"] for m in par["best_path_mismatches"])
    assert par["best_path_identical_frac"] >= 0.999 or ties, par
    assert line["roofline_feat"]["frac"] > 0 and line["cpu_baseline"]["extrapolated_all_cores"] > line["cpu_baseline"]["value"] * 0.5
