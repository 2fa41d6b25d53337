"""GPU: the drop-in boundary at the level of Kaldi's C++ classes.  kaldi_amd/adapter/_build/nnet3-compute is the REFERENCE's own
nnet3bin/nnet3-compute.cc + its unmodified nnet3 library (Nnet::Read, Compiler, Optimize, NnetComputer::ExecuteCommand, every
Component::Propagate) linked against kaldi_amd/adapter/cu-k3.cc -- CuMatrix / CuVector with HBM storage forwarding to the C ABI of
libk3hip.so -- in place of the reference's src/cudamatrix.  Its output must equal the reference's CPU run (`nnet3-compute --use-gpu=no`,
committed fixtures) within the north_star bound 1e-4, on the small model written by the reference's nnet3-copy (binary and text) and
on the full benchmark model."""
import os, subprocess, numpy as np, pytest
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden"); EXE = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-compute")

def _run(model, feats, s, tmp_path):
    from oracle import kaldi_io as kio              # Kaldi table IO of the test infrastructure (archives in / out of the binary)
    if not os.path.exists(EXE): pytest.fail("kaldi_amd/adapter/_build/nnet3-compute is missing: run kaldi_amd/adapter/build.sh where /root/reference exists (it travels to the GPU box)")
    fa, oa = str(tmp_path / "f.ark"), str(tmp_path / "o.ark")
    kio.write_ark(fa, {"u": feats})
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([EXE, "--use-gpu=no", f"--frame-subsampling-factor={s}", "--frames-per-chunk=150", model, f"ark:{fa}", f"ark:{oa}"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return kio.read_ark(oa)["u"], r.stderr

@pytest.mark.parametrize("fmt", ["raw", "txt"])
@pytest.mark.parametrize("s", [1, 3])
def test_reference_nnet_computer_over_the_k3_cumatrix_small_model(fmt, s, tmp_path):
    g = np.load(os.path.join(GOLD, "nnet_small_io.npz"))
    got, _ = _run(os.path.join(GOLD, "nnet_small." + fmt), g["feats"], s, tmp_path)
    ref = g[f"ref_out_{fmt}_s{s}"]          # made with --frames-per-chunk=50 (the default); chunking does not change a feed-forward model's output
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-4, np.abs(got - ref).max()

def test_reference_nnet_computer_over_the_k3_cumatrix_benchmark_model(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_golden_nnet_bench.py")); mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    g = np.load(os.path.join(GOLD, "nnet_bench_io.npz"))
    p = str(tmp_path / "m.raw"); _, sha = mk.bench_model_and_feats(p)
    assert sha == str(g["model_sha256"]), "the synthetic benchmark model is not bit-reproducible on this machine"
    got, log = _run(p, g["feats"], 3, tmp_path)
    ref = g["ref_out_cols8"]
    assert got[:, ::8].shape == ref.shape
    err = np.abs(got[:, ::8] - ref).max()
    assert err <= 1e-4, (err, float(g["max_abs"]))

def test_an_operation_outside_the_adapter_fails_loudly(tmp_path):
    """a model with a component the adapter does not cover (here: a log-softmax output) must stop with the member's name, not fall back"""
    cfg = str(tmp_path / "n.config"); raw = str(tmp_path / "n.raw")
    open(cfg, "w").write("input-node name=input dim=8\ncomponent name=a type=NaturalGradientAffineComponent input-dim=8 output-dim=6\ncomponent-node name=a component=a input=input\n"
                         "component name=ls type=LogSoftmaxComponent dim=6\ncomponent-node name=ls component=ls input=a\noutput-node name=output input=ls\n")
    init = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-init")
    if not os.path.exists(init): pytest.skip("oracle/_ref/bin/nnet3-init not built")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))
    assert subprocess.run([init, "--srand=1", cfg, raw], capture_output=True, env=env).returncode == 0
    from oracle import kaldi_io as kio
    fa = str(tmp_path / "f.ark"); kio.write_ark(fa, {"u": np.random.default_rng(0).standard_normal((20, 8)).astype(np.float32)})
    r = subprocess.run([EXE, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/o.ark"], capture_output=True, text=True)
    assert r.returncode != 0 and "not implemented on the MI355X path" in r.stderr and "LogSoftMaxPerRow" in r.stderr, r.stderr[-1500:]
