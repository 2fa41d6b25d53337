"""CPU: the statistics of the end-to-end parity gate (bench.e2e_gate): exact one-sided McNemar test on paired per-utterance best-path flips."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench

def test_known_values():
    g = bench.e2e_gate(range(7), range(100, 104), 512)      # round 5's driver run: 7 flips against 4, none shared
    assert (g["gpu_only_b"], g["self_only_c"]) == (7, 4) and abs(g["p_value"] - 562 / 2048) < 1e-12 and g["pass"]
    assert bench.e2e_gate([], [], 128)["p_value"] == 1.0 and bench.e2e_gate([], [], 128)["pass"]
    assert abs(bench.e2e_gate([1, 2], [3], 128)["p_value"] - 0.5) < 1e-12
    assert not bench.e2e_gate(range(20), range(100, 104), 512)["pass"]      # 20 against 4: p = 7.7e-4

def test_shared_flips_do_not_count_and_the_verdict_does_not_depend_on_the_sample_size():
    a = bench.e2e_gate([1, 2, 3, 9], [3, 9, 11], 128)
    assert (a["gpu_only_b"], a["self_only_c"], a["flipped_by_both"]) == (2, 1, 2)
    assert bench.e2e_gate([1, 2, 3, 9], [3, 9, 11], 512)["p_value"] == a["p_value"]

def test_smallest_failing_count_is_reported():
    g = bench.e2e_gate(range(7), range(100, 104), 512)
    b = g["gpu_only_flips_that_would_fail"]
    assert not bench.e2e_gate(range(b), range(100, 104), 512)["pass"] and bench.e2e_gate(range(b - 1), range(100, 104), 512)["pass"]
