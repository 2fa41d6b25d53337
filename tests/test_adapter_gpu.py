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

def _init_model(tmp_path, tail):
    cfg = str(tmp_path / "n.config"); raw = str(tmp_path / "n.raw")
    open(cfg, "w").write("input-node name=input dim=8\ncomponent name=a type=NaturalGradientAffineComponent input-dim=8 output-dim=6\ncomponent-node name=a component=a input=input\n" + tail)
    init = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-init")
    if not os.path.exists(init): pytest.skip("oracle/_ref/bin/nnet3-init not built")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    assert subprocess.run([init, "--srand=1", cfg, raw], capture_output=True, env=env).returncode == 0
    from oracle import kaldi_io as kio
    fa = str(tmp_path / "f.ark"); kio.write_ark(fa, {"u": np.random.default_rng(0).standard_normal((20, 8)).astype(np.float32) * 3.0})
    return raw, fa, env

def test_an_operation_outside_the_adapter_fails_loudly(tmp_path):
    """a model with a component the adapter does not cover (here: a p-norm, CuMatrixBase::GroupPnorm) must stop with the member's name, not fall back"""
    raw, fa, _ = _init_model(tmp_path, "component name=pn type=PnormComponent input-dim=6 output-dim=3\ncomponent-node name=pn component=pn input=a\noutput-node name=output input=pn\n")
    r = subprocess.run([EXE, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/o.ark"], capture_output=True, text=True)
    assert r.returncode != 0 and "not implemented on the MI355X path" in r.stderr and "GroupPnorm" in r.stderr, r.stderr[-1500:]

RENORM_TAILS = {
    "sigmoid": "component name=x type=SigmoidComponent dim=6\ncomponent-node name=x component=x input=a\noutput-node name=output input=x\n",
    "tanh": "component name=x type=TanhComponent dim=6\ncomponent-node name=x component=x input=a\noutput-node name=output input=x\n",
    "renorm": "component name=x type=NormalizeComponent dim=6 target-rms=0.7\ncomponent-node name=x component=x input=a\noutput-node name=output input=x\n",
    "renorm_log_stddev": "component name=x type=NormalizeComponent dim=6 add-log-stddev=true\ncomponent-node name=x component=x input=a\noutput-node name=output input=x\n",
    "renorm_blocks": "component name=x type=NormalizeComponent dim=6 block-dim=3 target-rms=2.0\ncomponent-node name=x component=x input=a\noutput-node name=output input=x\n"}
@pytest.mark.parametrize("kind", sorted(RENORM_TAILS))
def test_sigmoid_tanh_renorm_layers_equal_the_reference(kind, tmp_path):
    """SigmoidComponent / TanhComponent / NormalizeComponent (plain, with the log-stddev column, in blocks): the reference's NnetComputer over the adapter (CuMatrixBase::Sigmoid / Tanh,
    cu-matrix.h:288,:386; cu::NormalizePerRow, cu-math.h:272) against the reference's own nnet3-compute on the CPU; the fused path of k3_nnet_load for the layouts it takes, a refusal by
    name for the two it does not"""
    from oracle import kaldi_io as kio
    raw, fa, env = _init_model(tmp_path, RENORM_TAILS[kind])
    ref = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-compute")
    assert subprocess.run([ref, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/r.ark"], capture_output=True, env=env).returncode == 0
    want = kio.read_ark(f"{tmp_path}/r.ark")["u"]
    r = subprocess.run([EXE, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/a.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr[-1500:]
    got = kio.read_ark(f"{tmp_path}/a.ark")["u"]
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-5, np.abs(got - want).max()
    prog = os.path.join(ROOT, "kaldi_amd", "bin", "nnet3-compute")
    r = subprocess.run([prog, raw, f"ark:{fa}", f"ark:{tmp_path}/g.ark"], capture_output=True, text=True)
    if kind in ("renorm_log_stddev", "renorm_blocks"): assert r.returncode != 0 and "NormalizeComponent" in r.stderr, r.stderr[-1500:]
    else:
        assert r.returncode == 0, r.stderr[-1500:]
        got = kio.read_ark(f"{tmp_path}/g.ark")["u"]; assert got.shape == want.shape and np.abs(got - want).max() <= 1e-5, np.abs(got - want).max()

@pytest.mark.parametrize("kind", ["LogSoftmaxComponent", "SoftmaxComponent"])
def test_softmax_output_layers_equal_the_reference(kind, tmp_path):
    """non-chain nnet3 models end in a (log-)softmax (nnet-simple-component.cc:3494-3504, :3618-3625): the reference's NnetComputer over the adapter (CuMatrixBase::SoftMaxPerRow /
    LogSoftMaxPerRow, cu-matrix.h:328,334) and the fused path of k3_nnet_load (the drop-in nnet3-compute program) against the reference's own nnet3-compute on the CPU"""
    from oracle import kaldi_io as kio
    raw, fa, env = _init_model(tmp_path, f"component name=ls type={kind} dim=6\ncomponent-node name=ls component=ls input=a\noutput-node name=output input=ls\n")
    ref = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-compute")
    assert subprocess.run([ref, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/r.ark"], capture_output=True, env=env).returncode == 0
    want = kio.read_ark(f"{tmp_path}/r.ark")["u"]
    r = subprocess.run([EXE, "--use-gpu=no", raw, f"ark:{fa}", f"ark:{tmp_path}/a.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr[-1500:]
    assert np.abs(kio.read_ark(f"{tmp_path}/a.ark")["u"] - want).max() <= 1e-5
    prog = os.path.join(ROOT, "kaldi_amd", "bin", "nnet3-compute")
    r = subprocess.run([prog, raw, f"ark:{fa}", f"ark:{tmp_path}/g.ark"], capture_output=True, text=True); assert r.returncode == 0, r.stderr[-1500:]
    got = kio.read_ark(f"{tmp_path}/g.ark")["u"]
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-5 and (kind != "SoftmaxComponent" or abs(float(got.sum(1).mean()) - 1.0) < 1e-5)


@pytest.mark.parametrize("kind,flags,warp", [("fbank", ["--num-mel-bins=40", "--dither=0"], None), ("mfcc", ["--num-mel-bins=40", "--num-ceps=40", "--low-freq=20", "--high-freq=-400", "--dither=0"], None),
                                             ("fbank", ["--num-mel-bins=23", "--dither=0", "--use-energy=true", "--window-type=hamming"], "1.1")])
def test_cuda_spectral_features_class_equals_the_reference_binaries(kind, flags, warp, tmp_path):
    """include/k3_cuda_features.h: kaldi::CudaSpectralFeatures with the reference's signatures (cudafeat/feature-spectral-cuda.h:70-107) in a caller that uses the reference's
    own option classes, WaveData and table IO (tests/adapter/cuda_features_example.cc): its features against the reference's CPU compute-{fbank,mfcc}-feats binaries"""
    from oracle import kaldi_io as kio
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "cuda-features-example"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", f"compute-{kind}-feats")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/cuda-features-example is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); scp = []
    for i, n in enumerate((16000, 5000, 23017)): kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(n, 70 + i)); scp.append(f"u{i} {td}/u{i}.wav")
    open(f"{td}/wav.scp", "w").write("\n".join(scp) + "\n")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([ref] + flags + ([f"--vtln-warp={warp}"] if warp else []) + [f"scp:{td}/wav.scp", f"ark:{td}/ref.ark"], capture_output=True, text=True, env=env); assert r.returncode == 0, r.stderr
    g = subprocess.run([exe] + flags + [kind, f"scp:{td}/wav.scp", f"ark:{td}/got.ark"] + ([warp] if warp else []), capture_output=True, text=True, env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")); assert g.returncode == 0, g.stderr[-2000:]
    a, b = kio.read_ark(f"{td}/ref.ark"), kio.read_ark(f"{td}/got.ark")
    assert sorted(a) == sorted(b) == ["u0", "u1", "u2"]
    for k in a: assert a[k].shape == b[k].shape and np.abs(a[k] - b[k]).max() <= (1e-4 if kind == "fbank" else 3e-4), (k, np.abs(a[k] - b[k]).max())


def _kaldi_matrix(path, m):
    import struct
    m = np.ascontiguousarray(m, "<f4"); open(path, "wb").write(b"\0BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.tobytes())
def _read_kaldi(path):
    import struct
    b = open(path, "rb").read(); assert b[:2] == b"\0B"
    if b[2:5] == b"FM ": r, c = struct.unpack("<i", b[6:10])[0], struct.unpack("<i", b[11:15])[0]; return np.frombuffer(b, "<f4", r * c, 15).reshape(r, c)
    assert b[2:5] == b"FV "; n = struct.unpack("<i", b[6:10])[0]; return np.frombuffer(b, "<f4", n, 10)

@pytest.mark.parametrize("B,T,s", [(4, 10, 3), (16, 25, 3), (3, 7, 1)])
def test_reference_training_computation_over_the_k3_cumatrix(B, T, s, tmp_path):
    """The reference's OWN training computation of one minibatch -- request with need_model_derivative, CachingOptimizingCompiler, NnetComputer with a gradient nnet: forward in
    TRAINING mode (BatchNorm on batch statistics), then Backprop of every component (tests/adapter/nnet3_train_grad.cc, the calls of nnet3/nnet-chain-training.cc:136-206) --
    linked against kaldi_amd/adapter/cu-k3.cc: every matrix operation runs on the MI355X.  Output and the gradient of all parameters against the same program on the reference's
    CPU matrices (oracle/_ref/bin/ref-nnet3-train-grad)."""
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-train-grad"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-train-grad")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-train-grad is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); N = 50
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=calib, out_std=1.5).write(f"{td}/m.raw")
    lc = rc = 8; Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(B * 100 + T)
    _kaldi_matrix(f"{td}/in.mat", rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5); _kaldi_matrix(f"{td}/od.mat", rng.standard_normal((T * B, N)) * 0.1)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([ref, f"{td}/m.raw", str(B), str(T), str(s), f"{td}/in.mat", f"{td}/od.mat", f"{td}/ro.mat", f"{td}/rg.vec"], capture_output=True, text=True, env=dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl")))
    if r.returncode != 0 and "context" in r.stderr: pytest.skip(r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-2000:]
    g = subprocess.run([exe, f"{td}/m.raw", str(B), str(T), str(s), f"{td}/in.mat", f"{td}/od.mat", f"{td}/go.mat", f"{td}/gg.vec"], capture_output=True, text=True, env=env)
    assert g.returncode == 0, g.stderr[-3000:]
    ro, go, rg, gg = _read_kaldi(f"{td}/ro.mat"), _read_kaldi(f"{td}/go.mat"), _read_kaldi(f"{td}/rg.vec"), _read_kaldi(f"{td}/gg.vec")
    assert ro.shape == go.shape and np.abs(ro - go).max() <= 2e-4 * max(1.0, np.abs(ro).max()), np.abs(ro - go).max()
    assert rg.shape == gg.shape and np.linalg.norm(rg) > 0
    assert np.linalg.norm(rg - gg) <= 1e-3 * np.linalg.norm(rg), (np.linalg.norm(rg - gg), np.linalg.norm(rg))
    assert np.abs(rg - gg).max() <= 2e-3 * np.abs(rg).max(), (np.abs(rg - gg).max(), np.abs(rg).max())


@pytest.mark.parametrize("variant", ["fixture", "log_stddev_and_blocks"])
def test_reference_training_computation_with_sigmoid_tanh_renorm_layers(variant, tmp_path):
    """forward in training mode + Backprop of SigmoidComponent / TanhComponent / NormalizeComponent (CuMatrixBase::DiffSigmoid / DiffTanh, cu-matrix.h:390-396; cu::DiffNormalizePerRow,
    cu-math.h:296, added to the input derivative or in place as the compiled computation has it) through the reference's NnetComputer over the adapter, against the same program on the
    reference's CPU matrices: tests/golden/nnet_renorm.raw, and a variant with the log-stddev column and a block dimension made by the reference's nnet3-init"""
    import importlib.util
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-train-grad"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-train-grad"); init = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-init")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-train-grad is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref) or not os.path.exists(init): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"); renv = dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))
    model = os.path.join(GOLD, "nnet_renorm.raw"); out_dim = 16
    if variant != "fixture":
        spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_golden_nnet_renorm.py")); mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
        cfg = mk.CONFIG.replace("type=NormalizeComponent dim=32 target-rms=0.5", "type=NormalizeComponent dim=32 block-dim=8 target-rms=0.5 add-log-stddev=true").replace("input-dim=64 output-dim=24", "input-dim=72 output-dim=24")
        cfg = cfg.replace("type=NormalizeComponent dim=24\n", "type=NormalizeComponent dim=24 add-log-stddev=true\n").replace("input-dim=24 output-dim=16", "input-dim=25 output-dim=16")
        open(f"{td}/n.config", "w").write(cfg); model = f"{td}/n.raw"
        r = subprocess.run([init, "--srand=5", f"{td}/n.config", model], capture_output=True, text=True, env=renv); assert r.returncode == 0, r.stderr[-1500:]
    B, T, s, lc, rc = 6, 12, 3, 4, 4; Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(77)
    x = rng.standard_normal((Tin * B, 20)) * 2.0; x[5] = 0.0; x[11] *= 40.0
    _kaldi_matrix(f"{td}/in.mat", x); _kaldi_matrix(f"{td}/od.mat", rng.standard_normal((T * B, out_dim)) * 0.1)
    r = subprocess.run([ref, model, str(B), str(T), str(s), f"{td}/in.mat", f"{td}/od.mat", f"{td}/ro.mat", f"{td}/rg.vec"], capture_output=True, text=True, env=renv); assert r.returncode == 0, r.stderr[-2000:]
    g = subprocess.run([exe, model, str(B), str(T), str(s), f"{td}/in.mat", f"{td}/od.mat", f"{td}/go.mat", f"{td}/gg.vec"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
    ro, go, rg, gg = _read_kaldi(f"{td}/ro.mat"), _read_kaldi(f"{td}/go.mat"), _read_kaldi(f"{td}/rg.vec"), _read_kaldi(f"{td}/gg.vec")
    assert ro.shape == go.shape and np.abs(ro - go).max() <= 1e-4, np.abs(ro - go).max()
    assert rg.shape == gg.shape and np.linalg.norm(rg) > 0 and np.linalg.norm(rg - gg) <= 1e-4 * np.linalg.norm(rg), (np.linalg.norm(rg - gg), np.linalg.norm(rg))


DROPOUTS = {"general_continuous": "type=GeneralDropoutComponent dim=32 dropout-proportion=0.3 continuous=true", "general_binary": "type=GeneralDropoutComponent dim=32 dropout-proportion=0.3",
            "per_element": "type=DropoutComponent dim=32 dropout-proportion=0.3", "per_frame": "type=DropoutComponent dim=32 dropout-proportion=0.3 dropout-per-frame=true"}
@pytest.mark.parametrize("kind", sorted(DROPOUTS))
def test_dropout_in_training_mode_through_the_adapter(kind, tmp_path):
    """The chain recipes train with a dropout schedule (GeneralDropoutComponent, dropout-per-dim-continuous): the masks come from CuRand<BaseFloat>::RandUniform (cu-rand.h:50), here the
    device generator of k3_mat_set_rand.  The reference's stream (rand() on the CPU, cuRAND on its GPU) is not reproducible, so the gate is structural: the mask recovered from
    output / (output of the same model without dropout) has the component's law (nnet-general-component.cc:1790-1806, nnet-simple-component.cc:139-176: continuous in [1 - 2p, 1 + 2p] per
    (sequence, dim) and constant over time; binary {0, 1 / (1 - p)} per (sequence, dim); {0, 1} per element / per frame with P(0) = p), and the parameter gradient of the backward pass is the
    one of exactly that mask (the memo of the forward pass)."""
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-train-grad"); init = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-init")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-train-grad is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(init): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"); renv = dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl")); p = 0.3
    cfg = "input-node name=input dim=20\ncomponent name=a type=NaturalGradientAffineComponent input-dim=20 output-dim=32\ncomponent-node name=a component=a input=input\ncomponent name=d %s\ncomponent-node name=d component=d input=a\noutput-node name=output input=d\n"
    for name, comp in (("drop", DROPOUTS[kind]), ("plain", "type=NoOpComponent dim=32")):
        open(f"{td}/{name}.config", "w").write(cfg % comp)
        r = subprocess.run([init, "--srand=4", f"{td}/{name}.config", f"{td}/{name}.raw"], capture_output=True, text=True, env=renv); assert r.returncode == 0, r.stderr[-1500:]
    B, T = 48, 40; rng = np.random.default_rng(3); x = (rng.standard_normal((T * B, 20)) * 2.0).astype(np.float32); od = (rng.standard_normal((T * B, 32)) * 0.1).astype(np.float32)
    _kaldi_matrix(f"{td}/in.mat", x); _kaldi_matrix(f"{td}/od.mat", od); outs = {}
    for name in ("drop", "plain"):
        g = subprocess.run([exe, f"{td}/{name}.raw", str(B), str(T), "1", f"{td}/in.mat", f"{td}/od.mat", f"{td}/{name}.o", f"{td}/{name}.g"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
        outs[name] = (_read_kaldi(f"{td}/{name}.o"), _read_kaldi(f"{td}/{name}.g"))
    a = outs["plain"][0]; y = outs["drop"][0]; assert np.abs(a).min() > 1e-6
    m = (y / a).reshape(T, B, 32)                      # rows are frame-major, sequence-minor
    if kind.startswith("general"):
        assert np.abs(m - m[0]).max() <= 1e-4          # one mask per (sequence, dim), shared by all frames
        m0 = m[0]
        if kind == "general_continuous": assert m0.min() >= 1 - 2 * p - 1e-4 and m0.max() <= 1 + 2 * p + 1e-4 and abs(m0.mean() - 1.0) < 0.05 and abs(m0.std() - 4 * p / np.sqrt(12)) < 0.03
        else: assert np.all((np.abs(m0) < 1e-4) | (np.abs(m0 - 1 / (1 - p)) < 1e-4)) and abs((np.abs(m0) < 1e-4).mean() - p) < 0.05
        assert len(np.unique(np.round(m0, 4), axis=0)) == B          # every sequence its own mask
    else:
        assert np.all((np.abs(m) < 1e-4) | (np.abs(m - 1) < 1e-4)) and abs((np.abs(m) < 1e-4).mean() - p) < (0.02 if kind == "per_element" else 0.04)
        if kind == "per_frame": assert np.abs(m - m[:, :, :1]).max() <= 1e-4 and 0 < (np.abs(m[:, :, 0]) < 1e-4).sum() < T * B          # whole rows on / off
        else: assert 0.2 < (np.abs(m[:, :, 0]) < 1e-4).mean() < 0.4 and np.abs(m - m[:, :, :1]).max() > 0.5
    # backward: d/dW = (od .* mask)^T x, d/db = column sums of od .* mask (VectorizeNnet: the affine's linear rows, then its bias)
    gm = (od * m.reshape(T * B, 32)).astype(np.float64); want = np.concatenate([(gm.T @ x.astype(np.float64)).ravel(), gm.sum(0)]); got = outs["drop"][1]
    assert got.shape == want.shape and np.linalg.norm(got - want) <= 1e-4 * np.linalg.norm(want), (np.linalg.norm(got - want), np.linalg.norm(want))


@pytest.mark.parametrize("B,T", [(4, 10), (16, 25)])
def test_lf_mmi_gradient_reference_nnet_computer_plus_native_objective(B, T, tmp_path):
    """One minibatch of chain training as nnet3/nnet-chain-training.cc:136-300 runs it: NnetComputer forward (training mode) -> ComputeChainObjfAndDeriv -> NnetComputer backward into a
    gradient nnet (tests/adapter/nnet3_chain_grad.cc).  MI355X build: the reference's unmodified NnetComputer over the CuMatrix adapter and the objective by k3_chain_objf_and_deriv on the
    adapter's device pointers; oracle build: the reference's nnet3 + chain code on its CPU matrices with the merged supervision FST.  Objective, l2 term and the gradient of every parameter."""
    import struct
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-grad"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-grad")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-chain-grad is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); P = 50; s = 3
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5).write(f"{td}/m.raw")
    lc = rc = 8; Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(B * 100 + T)
    _kaldi_matrix(f"{td}/in.mat", rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5)
    den = synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60); fsts = [synth.make_supervision_fst(T, P, seed=200 + i) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
    with open(f"{td}/chain.spec", "wb") as fh:
        fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, merged.num_states, int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
        fh.write(fb(den)); fh.write(fb(merged)); fh.write(so.tobytes())
        fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
        for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([ref, f"{td}/m.raw", str(s), f"{td}/in.mat", f"{td}/chain.spec", f"{td}/r.vec"], capture_output=True, text=True, env=dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))); assert r.returncode == 0, r.stderr[-2000:]
    g = subprocess.run([exe, f"{td}/m.raw", str(s), f"{td}/in.mat", f"{td}/chain.spec", f"{td}/g.vec"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
    rv, gv = _read_kaldi(f"{td}/r.vec"), _read_kaldi(f"{td}/g.vec")
    assert rv.shape == gv.shape and rv[2] == gv[2] == B * T
    assert abs(rv[0] - gv[0]) <= 2e-4 * abs(rv[0]) + 1e-3 and abs(rv[1] - gv[1]) <= 2e-4 * abs(rv[1]) + 1e-5, (rv[:3], gv[:3])
    rg, gg = rv[3:], gv[3:]
    assert np.linalg.norm(rg) > 0 and np.linalg.norm(rg - gg) <= 2e-3 * np.linalg.norm(rg), (np.linalg.norm(rg - gg), np.linalg.norm(rg))


@pytest.mark.parametrize("iters,momentum,nmb", [(1, 0.0, 1), (2, 0.5, 1), (3, 0.0, 1), (2, 0.0, 2)])      # nmb = 2: a list of two minibatches, iteration i trains on minibatch i mod 2 (the program's stand-in for an egs archive)
def test_chain_training_iterations_equal_the_reference(iters, momentum, nmb, tmp_path):
    """N iterations of LF-MMI TRAINING, the sequence NnetChainTrainer::TrainInternal runs (nnet3/nnet-chain-training.cc:100-144) -- forward in training mode with component
    statistics, objective, backward with every component's natural-gradient update (OnlineNaturalGradient), L2, UpdateNnetWithMaxChange, batch-norm statistics decay, the semi-orthogonal
    constraint of the TDNN-F bottlenecks, momentum (kaldi_amd/adapter/nnet3-chain-train.cc).  MI355X build: the reference's unmodified nnet3 objects over the CuMatrix adapter + k3_chain_objf_and_deriv;
    oracle build: the same source on the reference's CPU matrices and chain code.  Per-iteration objective and the trained parameters.
    Natural-gradient SGD amplifies float32 rounding discontinuously (the preconditioner's early eigen-decompositions): the
    REFERENCE ITSELF ends 10-35 % of the training's own parameter change apart between MKL's AVX2 and AVX-512 code paths after two / three iterations on this case (measured),
    so the MI355X result has to coincide with the reference under at least one of MKL's code paths (default, AVX2, AVX512, SSE4_2): to 2e-3 of the change after one iteration, 1 % after several."""
    import struct
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-train"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-train")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-chain-train is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); B, T, P, s = 8, 12, 50, 3
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5, orthonormal_constraint=-1.0); net.write(f"{td}/m.raw")
    lc = rc = 8; Tin = (T - 1) * s + 1 + lc + rc
    den = synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60)
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    ins, specs = [], []
    for m in range(nmb):
        rng = np.random.default_rng(B * 100 + T + 7 * m)
        _kaldi_matrix(f"{td}/in{m}.mat", rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5)
        fsts = [synth.make_supervision_fst(T, P, seed=200 + i + 50 * m) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
        so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
        with open(f"{td}/chain{m}.spec", "wb") as fh:
            fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, merged.num_states, int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
            fh.write(fb(den)); fh.write(fb(merged)); fh.write(so.tobytes())
            fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
            for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
        ins.append(f"{td}/in{m}.mat"); specs.append(f"{td}/chain{m}.spec")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"); args = [f"{td}/m.raw", str(s), ",".join(ins), ",".join(specs), str(iters), "0.002", str(momentum)]
    g = subprocess.run([exe] + args + [f"{td}/g.raw", f"{td}/g.vec"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
    gv = _read_kaldi(f"{td}/g.vec"); go = gv[:3 * iters].reshape(iters, 3); gp = gv[3 * iters:]
    if iters == 1:      # the same iteration as rank 0 of a one-rank data-parallel job: the parameter changes go through k3_comm_create + k3_comm_allreduce_f32 (RCCL) and come back unchanged
        g2 = subprocess.run([exe] + args + [f"{td}/g2.raw", f"{td}/g2.vec"], capture_output=True, text=True, env=dict(env, K3_TRAIN_ID_FILE=f"{td}/nccl.id", K3_TRAIN_RANK="0", K3_TRAIN_WORLD="1")); assert g2.returncode == 0, g2.stderr[-3000:]
        assert "data-parallel rank 0 of 1" in g2.stderr and np.array_equal(_read_kaldi(f"{td}/g2.vec"), gv)
    p0 = np.concatenate([np.concatenate([c[2]["W"].ravel()] + ([c[2]["b"].ravel()] if "b" in c[2] and c[2]["b"].size else [])) for c in net.components if c[1] in ("affine", "tdnn", "linear")])
    assert p0.shape == gp.shape
    tried = []
    for path in [None, "AVX512", "AVX2", "SSE4_2"]:      # (one iteration: the reference's own paths are 3.5e-3 apart, tests/test_oracle_chain.py; the MI355X result has to sit within 2e-3 of one of them)
        e = dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))
        if path: e["MKL_ENABLE_INSTRUCTIONS"] = path
        r = subprocess.run([ref] + args + [f"{td}/r.raw", f"{td}/r.vec"], capture_output=True, text=True, env=e); assert r.returncode == 0, r.stderr[-2000:]
        rv = _read_kaldi(f"{td}/r.vec"); assert rv.shape == gv.shape
        ro = rv[:3 * iters].reshape(iters, 3); rp = rv[3 * iters:]
        assert np.array_equal(ro[:, 2], go[:, 2]) and abs(ro[0, 0] - go[0, 0]) <= 2e-4 * abs(ro[0, 0]) + 1e-3      # the first objective does not depend on any update
        assert np.linalg.norm(rp - p0) > 0.1 and (iters == 1 or nmb > 1 or abs(ro[-1, 0] - ro[0, 0]) > 10.0)                  # the training moved the model
        rel = float(np.linalg.norm(rp - gp) / np.linalg.norm(rp - p0)); dobj = float(np.abs(ro[:, 0] - go[:, 0]).max() / np.abs(ro[:, 0]).max())
        tried.append((path or "default", rel, dobj))
        if rel <= (2e-3 if iters == 1 else 1e-2) and dobj <= (5e-4 if iters == 1 else 5e-3):
            assert g.stderr.count("ConstrainOrthonormalInternal") == r.stderr.count("ConstrainOrthonormalInternal")
            return
    pytest.fail(f"MI355X training result matches none of the reference's runs: (MKL path, |params - ref| / |ref - initial|, max relative objective difference) = {tried}")


def test_chain_training_iteration_with_end_to_end_supervisions(tmp_path):
    """the training program on END-TO-END (flat-start) supervisions -- a chain spec with magic 0x4b38: the per-sequence FSTs (self-loops, several final states) are Supervision::e2e_fsts
    in the oracle build (ComputeChainObjfAndDeriv's end-to-end branch with GenericNumeratorComputation, chain/chain-generic-numerator.cc compiled unmodified) and
    k3_chain_supervision_create_e2e in the MI355X build: one training iteration ends at the reference's objective and parameters."""
    import struct
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-train"); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-train")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-chain-train is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); B, T, P, s = 6, 20, 50, 3
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5, orthonormal_constraint=-1.0); net.write(f"{td}/m.raw")
    lc = rc = 8; Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(77)
    _kaldi_matrix(f"{td}/in.mat", rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5)
    den = synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60); fsts = [synth.make_e2e_fst(T, P, seed=900 + i, num_phones=int(rng.integers(2, 8))) for i in range(B)]
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
    with open(f"{td}/chain.spec", "wb") as fh:      # (the merged-FST slot of the format holds a copy of the first sequence's FST: unused for end-to-end supervisions)
        fh.write(struct.pack("<11i3f", 0x4b38, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, fsts[0].num_states, int(fsts[0].arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
        fh.write(fb(den)); fh.write(fb(fsts[0])); fh.write(so.tobytes())
        fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
        for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"); args = [f"{td}/m.raw", str(s), f"{td}/in.mat", f"{td}/chain.spec", "1", "0.002", "0.0"]
    g = subprocess.run([exe] + args + [f"{td}/g.raw", f"{td}/g.vec"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
    gv = _read_kaldi(f"{td}/g.vec"); p0 = np.concatenate([np.concatenate([c[2]["W"].ravel()] + ([c[2]["b"].ravel()] if "b" in c[2] and c[2]["b"].size else [])) for c in net.components if c[1] in ("affine", "tdnn", "linear")])
    tried = []
    for extra in ({}, {"MKL_CBWR": "COMPATIBLE"}):      # (the reference under MKL's two code paths on this host: natural-gradient SGD amplifies float32 rounding)
        r = subprocess.run([ref] + args + [f"{td}/r.raw", f"{td}/r.vec"], capture_output=True, text=True, env=dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), **extra)); assert r.returncode == 0, r.stderr[-2000:]
        rv = _read_kaldi(f"{td}/r.vec"); assert rv.shape == gv.shape and rv[2] == gv[2] == B * T
        assert abs(rv[0] - gv[0]) <= 2e-4 * abs(rv[0]) + 1e-3 and abs(rv[1] - gv[1]) <= 2e-4 * abs(rv[1]) + 1e-5, (rv[:3], gv[:3])      # objective and l2 term of the iteration (no update involved yet)
        rel = float(np.linalg.norm(rv[3:] - gv[3:]) / np.linalg.norm(rv[3:] - p0)); tried.append(rel)
        assert np.linalg.norm(rv[3:] - p0) > 0.1
        if rel <= 2e-3: return
    pytest.fail(f"|params - ref| / |ref - initial| = {tried}")


def test_nnet3_chain_train_with_the_reference_command_line_and_example_archives(tmp_path):
    """chainbin/nnet3-chain-train.cc ITSELF over the adapter (kaldi_amd/adapter/_build/nnet3-chain-train-egs: the reference's main, NnetChainTrainer, NnetChainExample readers and
    computation-request code unmodified; chain-k3.cc behind ComputeChainObjfAndDeriv / DenominatorGraph / Supervision::Read / ReadFstKaldi): the reference's command line
      nnet3-chain-train [options] <raw-nnet-in> <denominator-fst-in> <chain-training-examples-in> <raw-nnet-out>
    on merged example archives (text as tests/chain_egs.py writes them, and binary as the reference's nnet3-chain-copy-egs -- same objects -- rewrites them).  Checked: the objective
    of the first minibatch is the one the spec-driven program (nnet3-chain-train, itself held to the CPU reference above) gets on the same minibatch -- i.e. the merged supervision
    FST read from the archive, cut back into sequences, gives the per-sequence numerator exactly --; text and binary archives train to the same model bit for bit; several minibatches,
    xent regularisation, deriv weights and an end-to-end supervision run through; the parsed log-prob line is there."""
    import re, struct
    from kaldi_amd import synth
    from tests import chain_egs as ce
    B_ = os.path.join(ROOT, "kaldi_amd", "adapter", "_build"); exe = os.path.join(B_, "nnet3-chain-train-egs"); cp = os.path.join(B_, "nnet3-chain-copy-egs"); spec_exe = os.path.join(B_, "nnet3-chain-train")
    for e in (exe, cp, spec_exe):
        if not os.path.exists(e): pytest.fail(f"{e} is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    td = str(tmp_path); B, T, P, s = 8, 12, 50, 3; lc = rc = 8; Tin = (T - 1) * s + 1 + lc + rc
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5, orthonormal_constraint=-1.0).write(f"{td}/m.raw")
    den = synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60); den.write_openfst(f"{td}/den.fst")
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    egs = []; nmb = 3
    for m in range(nmb):
        rng = np.random.default_rng(B * 100 + T + 7 * m); x = (rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5).astype(np.float32)
        fsts = [synth.make_supervision_fst(T, P, seed=200 + i + 50 * m) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
        egs.append(ce.minibatch(f"mb{m}", x, B, T, s, lc, rc, P, merged_fst=merged))
        if m == 0:      # the same minibatch for the spec-driven program
            _kaldi_matrix(f"{td}/in0.mat", x)
            so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
            with open(f"{td}/chain0.spec", "wb") as fh:
                fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, merged.num_states, int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
                fh.write(fb(den)); fh.write(fb(merged)); fh.write(so.tobytes())
                fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
                for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
    ce.write_chain_egs_text(f"{td}/one.txt", egs[:1]); ce.write_chain_egs_text(f"{td}/all.txt", egs)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"); opts = ["--print-interval=1", "--leaky-hmm-coefficient=1.0e-05", "--l2-regularize=5.0e-05", "--out-of-range-regularize=0.0", "--srand=3"]
    def objf(stderr, which="Overall average objective function for 'output' is "):
        m_ = re.search(re.escape(which) + r"(-?[0-9.e+-]+)", stderr); assert m_, stderr[-1500:]; return float(m_.group(1))
    # (1) first minibatch: the objective against the spec-driven program's iteration 0
    g = subprocess.run([exe] + opts + [f"{td}/m.raw", f"{td}/den.fst", f"ark,t:{td}/one.txt", f"{td}/o1.raw"], capture_output=True, text=True, env=env); assert g.returncode == 0, g.stderr[-3000:]
    assert "[this line is to be parsed by a script:] log-prob-per-frame=" in g.stderr and "Wrote raw model to" in g.stderr
    q = subprocess.run([spec_exe, f"{td}/m.raw", str(s), f"{td}/in0.mat", f"{td}/chain0.spec", "1", "0.002", "0.0", f"{td}/q.raw", f"{td}/q.vec"], capture_output=True, text=True, env=env); assert q.returncode == 0, q.stderr[-3000:]
    qv = _read_kaldi(f"{td}/q.vec"); want = float(qv[0] / qv[2])
    assert abs(objf(g.stderr) - want) <= 2e-5 * max(1.0, abs(want)), (objf(g.stderr), want)
    # (2) text and binary archives: the same trained model, bit for bit; the binary one is what the reference's copy tool writes (Supervision::Write of the adapter)
    c = subprocess.run([cp, f"ark,t:{td}/all.txt", f"ark:{td}/all.egs"], capture_output=True, text=True, env=env); assert c.returncode == 0 and "wrote 3" in c.stderr, c.stderr[-1500:]
    c = subprocess.run([cp, f"ark:{td}/all.egs", f"ark,t:{td}/all2.txt"], capture_output=True, text=True, env=env); assert c.returncode == 0, c.stderr[-1500:]
    c = subprocess.run([cp, f"ark,t:{td}/all2.txt", f"ark:{td}/all2.egs"], capture_output=True, text=True, env=env); assert c.returncode == 0, c.stderr[-1500:]      # (all2.txt: the binary archive's values at the text form's six digits; all2.egs holds exactly those)
    from oracle import nnet3_oracle as no
    def params(path):
        net_ = no.read_nnet(path)
        return np.concatenate([np.asarray(net_.components[n_].fields[k_], np.float64).ravel() for n_ in net_.comp_order for k_ in ("<LinearParams>", "<BiasParams>", "<Params>")
                               if isinstance(net_.components[n_].fields.get(k_), np.ndarray)])
    p0 = params(f"{td}/m.raw"); outs = []
    for spec_ in (f"ark,t:{td}/all2.txt", f"ark:{td}/all2.egs", f"ark,t:{td}/all2.txt"):      # text, binary, text again (the yardstick: how far two runs on identical input end apart)
        t_ = subprocess.run([exe] + opts + [f"{td}/m.raw", f"{td}/den.fst", spec_, f"{td}/o.raw"], capture_output=True, text=True, env=env); assert t_.returncode == 0, t_.stderr[-3000:]
        assert "for minibatches 1-1 is" in t_.stderr      # (a phase is reported when the next one starts; the last one by PrintTotalStats)
        outs.append((params(f"{td}/o.raw"), objf(t_.stderr, "for minibatches 0-0 is "), objf(t_.stderr)))
    moved = np.linalg.norm(outs[0][0] - p0); assert moved > 0 and outs[0][0].shape == p0.shape
    d_bin, d_txt = np.linalg.norm(outs[1][0] - outs[0][0]) / moved, np.linalg.norm(outs[2][0] - outs[0][0]) / moved
    print("text vs binary archive: trained parameters %.3e of the training's change apart; two runs on the same text archive: %.3e; objectives" % (d_bin, d_txt), [o[1:] for o in outs])
    assert abs(outs[1][1] - outs[0][1]) <= 1e-5 * max(1.0, abs(outs[0][1]))      # the first minibatch's objective: the same supervision and features came out of both archives
    assert d_bin <= max(5.0 * d_txt, 1e-3), (d_bin, d_txt)      # (natural-gradient SGD amplifies the last bits of the LF-MMI kernels' atomic sums: two runs on the SAME archive are the yardstick)
    # (3) xent regularisation needs an output-xent node: refused by the reference's own check, loudly
    x_ = subprocess.run([exe] + opts + ["--xent-regularize=0.1", f"{td}/m.raw", f"{td}/den.fst", f"ark:{td}/all.egs", f"{td}/x.raw"], capture_output=True, text=True, env=env)
    assert x_.returncode != 0 and "xent" in x_.stderr
    # (4) deriv weights of zero on every frame: objective reported, model (but for the batch-norm statistics and l2) unchanged by the gradient; usage / bad den FST errors
    z = [(k, i, [(n_, idx, sup, np.zeros(B * T, np.float32)) for n_, idx, sup, dw in o]) for k, i, o in egs[:1]]; ce.write_chain_egs_text(f"{td}/z.txt", z)
    zr = subprocess.run([exe] + opts + [f"{td}/m.raw", f"{td}/den.fst", f"ark,t:{td}/z.txt", f"{td}/z.raw"], capture_output=True, text=True, env=env); assert zr.returncode == 0, zr.stderr[-2000:]
    assert abs(objf(zr.stderr) - want) <= 2e-5 * max(1.0, abs(want))
    assert subprocess.run([exe, f"{td}/m.raw"], capture_output=True).returncode == 1
    bad = subprocess.run([exe] + opts + [f"{td}/m.raw", f"{td}/m.raw", f"ark:{td}/all.egs", f"{td}/b.raw"], capture_output=True, text=True, env=env); assert bad.returncode != 0


@pytest.mark.parametrize("threads,iters", [(2, 6), (4, 3)])
def test_adapter_on_several_host_threads_reproduces_the_single_threaded_results(threads, iters, tmp_path):
    """SURVEY 8b "threading" (cudamatrix/cu-device.cc:112-124: every host thread gets cudaStreamPerThread): the reference's training computation -- NnetComputer forward in training
    mode + Backprop, tests/adapter/nnet3_two_threads.cc -- run by several host threads AT ONCE over the adapter (one stream per thread, a memory pool that synchronises only when a
    block changes threads, index arrays cached on the device) must reproduce every thread's single-threaded output and gradient bit for bit: any difference is a race."""
    from kaldi_amd import synth
    exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-two-threads")
    if not os.path.exists(exe): pytest.fail("kaldi_amd/adapter/_build/nnet3-two-threads is missing: run kaldi_amd/adapter/build.sh where /root/reference exists")
    td = str(tmp_path)
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    synth.make_tdnnf(seed=5, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=60, calib_feats=calib, out_std=1.5).write(f"{td}/m.raw")
    g = subprocess.run([exe, f"{td}/m.raw", str(threads), str(iters), "8", "12", "3"], capture_output=True, text=True, env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"), timeout=600)
    assert g.returncode == 0 and "two-threads ok" in g.stdout, (g.stdout[-500:], g.stderr[-3000:])
