"""CPU: the product's model reader + layer fuser (host logic of k3_nnet_load) and the oracle reading the
same Kaldi-format files; the oracle is pinned to the reference's nnet3-compute by the committed fixture."""
import os, numpy as np, pytest
from kaldi_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

def test_reader_and_fuser_on_synth_tdnnf(tmp_path):
    import __graft_entry__ as ge; ge.build()
    from kaldi_amd import nnet3
    net = synth.make_tdnnf(seed=3, dim=64, bottleneck=16, strides=(1, 0, 3, 3), prefinal_small=32, num_pdfs=120, calib_frames=300)
    p = tmp_path / "m.raw"; net.write(str(p))
    n = nnet3.Nnet(p)
    i = n.info
    assert (i.input_dim, i.output_dim) == (40, 120)
    assert (i.left_context, i.right_context) == (1 + 1 + 0 + 3 + 3, 1 + 1 + 0 + 3 + 3)
    assert i.num_params == net.num_params()
    # tdnn1 + 4 x (linear, affine) + prefinal-l + prefinal affine + prefinal linear + output = 13 fused GEMM nodes
    assert i.num_fused_nodes == 13 and i.num_components == len(net.components)

def test_reader_accepts_reference_written_files():
    """text and binary files written by the REFERENCE's nnet3-copy (fixture) parse to the same model."""
    import __graft_entry__ as ge; ge.build()
    from kaldi_amd import nnet3
    a = nnet3.Nnet(os.path.join(GOLD, "nnet_small.txt")).info
    b = nnet3.Nnet(os.path.join(GOLD, "nnet_small.raw")).info
    for f, _ in a._fields_:
        assert getattr(a, f) == getattr(b, f), f
    assert a.num_fused_nodes == 11 and a.output_dim == 96   # tdnn1 + 3 x (linear, affine) + prefinal-l + 2 prefinal + output

def test_oracle_vs_reference_nnet3_compute():
    from oracle import nnet3_oracle as no
    g = np.load(os.path.join(GOLD, "nnet_small_io.npz"))
    for fmt in ("txt", "raw"):
        net = no.read_nnet(os.path.join(GOLD, "nnet_small." + fmt))
        for s in (1, 3):
            got = no.compute(net, g["feats"], s)
            ref = g[f"ref_out_{fmt}_s{s}"]
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-4, np.abs(got - ref).max()

def test_unsupported_model_fails_loudly(tmp_path):
    import __graft_entry__ as ge; ge.build()
    from kaldi_amd import nnet3, lib
    txt = open(os.path.join(GOLD, "nnet_small.txt")).read().replace("<RectifiedLinearComponent>", "<PnormComponent>").replace("</RectifiedLinearComponent>", "</PnormComponent>")
    p = tmp_path / "bad.txt"; p.write_text(txt)
    with pytest.raises(lib.K3Error, match="PnormComponent"):
        nnet3.Nnet(p)

def test_oracle_renorm_sigmoid_tanh_vs_reference_nnet3_compute():
    """NormalizeComponent (target-rms 0.5 and default), SigmoidComponent, TanhComponent, LogSoftmax output: the oracle against the REFERENCE's nnet3-compute on a model made by the
    reference's nnet3-init (tests/golden/make_golden_nnet_renorm.py); inputs include saturating and all-zero utterances"""
    from oracle import nnet3_oracle as no
    g = np.load(os.path.join(GOLD, "nnet_renorm_io.npz")); net = no.read_nnet(os.path.join(GOLD, "nnet_renorm.raw"))
    for s in (1, 3):
        for u in ("u0", "u1", "u2"):
            got = no.compute(net, g["feats_" + u], s); ref = g[f"ref_s{s}_{u}"]
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5, (s, u, np.abs(got - ref).max())

def test_reader_and_fuser_on_the_renorm_model(tmp_path):
    """the fused model of the relu-renorm fixture: 4 GEMM nodes (ReLU, sigmoid, tanh folded into their producers; the two NormalizeComponents and the LogSoftmax as row operations
    behind them); a NormalizeComponent with add-log-stddev or a block dimension is refused by name"""
    import __graft_entry__ as ge; ge.build()
    from kaldi_amd import nnet3, lib
    i = nnet3.Nnet(os.path.join(GOLD, "nnet_renorm.raw")).info
    assert (i.input_dim, i.output_dim, i.left_context, i.right_context, i.num_fused_nodes) == (20, 16, 4, 4, 4)
    from oracle import nnet3_oracle as no
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet3-copy")
    if not os.path.exists(exe): pytest.skip("oracle/_ref not built")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))
    assert subprocess.run([exe, "--binary=false", os.path.join(GOLD, "nnet_renorm.raw"), str(tmp_path / "m.txt")], capture_output=True, env=env).returncode == 0
    txt = open(tmp_path / "m.txt").read()
    assert nnet3.Nnet(tmp_path / "m.txt").info.num_fused_nodes == 4
    (tmp_path / "als.txt").write_text(txt.replace("<AddLogStddev> F", "<AddLogStddev> T", 1))
    with pytest.raises(lib.K3Error, match="add-log-stddev"): nnet3.Nnet(tmp_path / "als.txt")
    (tmp_path / "blk.txt").write_text(txt.replace("<InputDim> 32 <TargetRms>", "<InputDim> 32 <BlockDim> 16 <TargetRms>", 1))
    with pytest.raises(lib.K3Error, match="block-dim"): nnet3.Nnet(tmp_path / "blk.txt")


def test_random_architectures_against_the_reference_nnet3_compute(tmp_path):
    """fuzz (live only): random TDNN-F stacks (strides 0 / 1 / 3 in any order, widths, bottlenecks) and plain TDNNs with asymmetric splice
    offsets, random lengths, --frame-subsampling-factor 1 / 3 and --frames-per-chunk: the numpy oracle vs the reference's nnet3-compute
    on the same model file, inside the 1e-4 bound of the path (a 16-model run of this loop: worst 2.6e-5)"""
    import subprocess
    from oracle import kaldi_io as kio, nnet3_oracle as no
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); exe = os.path.join(root, "oracle", "_ref", "bin", "nnet3-compute")
    if not os.path.exists(exe): pytest.skip("oracle/_ref not built (needs /root/reference)")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    rng = np.random.default_rng(9); td = str(tmp_path)
    for it in range(6):
        if it % 2 == 0:
            nl = int(rng.integers(2, 6)); strides = tuple(int(x) for x in rng.choice([0, 1, 3], nl))
            net = synth.make_tdnnf(seed=int(rng.integers(0, 1000)), dim=int(rng.choice([32, 48, 96])), bottleneck=int(rng.choice([8, 12, 24])), strides=strides, prefinal_small=int(rng.choice([16, 24])),
                                   num_pdfs=int(rng.choice([30, 96])), calib_frames=200)
        else:
            offs = tuple(tuple(int(x) for x in sorted(set(rng.choice([-3, -2, -1, 0, 1, 2, 3], int(rng.integers(1, 4)))))) for _ in range(int(rng.integers(1, 4))))
            net = synth.make_tdnn(seed=int(rng.integers(0, 1000)), dim=int(rng.choice([32, 64])), offsets=offs, num_pdfs=int(rng.choice([30, 80])), calib_frames=200)
        net.write(f"{td}/m.raw")
        feats = (rng.standard_normal((int(rng.integers(1, 120)), 40)) * 1.2 + 16.5).astype(np.float32); kio.write_ark(f"{td}/f.ark", {"u": feats})
        s = int(rng.choice([1, 3])); chunk = int(rng.choice([50, 20, 150]))
        r = subprocess.run([exe, "--use-gpu=no", f"--frame-subsampling-factor={s}", f"--frames-per-chunk={chunk}", f"{td}/m.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        ref = kio.read_ark(f"{td}/o.ark")["u"]; mine = no.compute(no.read_nnet(f"{td}/m.raw"), feats, s)
        assert mine.shape == ref.shape and np.abs(mine - ref).max() <= 1e-4, (it, mine.shape, ref.shape)


IV_CASES = {"s1_c50_p10": (1, 50, 10, False), "s3_c50_p10": (3, 50, 10, False), "s3_c21_p7": (3, 21, 7, False), "s1_c20_p10_short": (1, 20, 10, False), "s3_utt": (3, 50, 0, True), "s1_utt": (1, 50, 0, True)}

def test_oracle_with_ivector_input_vs_reference_nnet3_compute():
    """the recipe's i-vector input (Append(-1,0,1,ReplaceIndex(ivector, t, 0))): the oracle's chunk-by-chunk evaluation against the REFERENCE's
    nnet3-compute --online-ivectors / --ivectors (tests/golden/make_golden_nnet_ivector.py), 1e-4 like the rest of the nnet path"""
    from oracle import nnet3_oracle as no
    g = np.load(os.path.join(GOLD, "nnet_ivector_io.npz")); net = no.read_nnet(os.path.join(GOLD, "nnet_ivector.raw"))
    for name, (s, chunk, period, utt) in IV_CASES.items():
        kw = dict(ivector=g["iv_" + name]) if utt else dict(online_ivectors=g["iv_" + name], online_ivector_period=period)
        got = no.compute(net, g["feats"], s, frames_per_chunk=chunk, **kw); ref = g["ref_" + name]
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4, (name, np.abs(got - ref).max())
    # the chunking matters: the whole-utterance evaluation with one of the rows is NOT what the reference computes
    other = no.compute(net, g["feats"], 1, ivector=g["iv_s1_c50_p10"][0])
    assert np.abs(other - g["ref_s1_c50_p10"]).max() > 1e-2


def test_random_relu_sigmoid_tanh_renorm_stacks_against_the_reference_nnet3_compute(tmp_path):
    """fuzz (live only): random stacks of spliced affines with ReLU / sigmoid / tanh, NormalizeComponents (random target-rms, with and without the log-stddev column and a block
    dimension) and a (log-)softmax or plain output, created by the reference's nnet3-init: the numpy oracle against the reference's nnet3-compute on the same file"""
    import subprocess
    from oracle import kaldi_io as kio, nnet3_oracle as no
    binr = os.path.join(ROOT, "oracle", "_ref", "bin")
    if not os.path.exists(os.path.join(binr, "nnet3-init")): pytest.skip("oracle/_ref not built (needs /root/reference)")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    rng = np.random.default_rng(31); td = str(tmp_path); worst = 0.0
    for it in range(8):
        dim = 12; prev = "input"; lines = [f"input-node name=input dim={dim}"]
        for l in range(int(rng.integers(1, 4))):
            offs = sorted(set(int(x) for x in rng.choice([-2, -1, 0, 1, 3], int(rng.integers(1, 4))))); out = int(rng.choice([8, 16, 24]))
            app = ", ".join(prev if o == 0 else f"Offset({prev}, {o})" for o in offs)
            lines += [f"component name=a{l} type=NaturalGradientAffineComponent input-dim={dim * len(offs)} output-dim={out}", f"component-node name=a{l} component=a{l} input=Append({app})"]
            kind = str(rng.choice(["RectifiedLinearComponent", "SigmoidComponent", "TanhComponent"]))
            lines += [f"component name=n{l} type={kind} dim={out}", f"component-node name=n{l} component=n{l} input=a{l}"]; prev, dim = f"n{l}", out
            if rng.integers(0, 2):
                als = bool(rng.integers(0, 2)); blk = int(rng.choice([out, out // 2])); rms = float(rng.choice([1.0, 0.5, 2.0]))
                lines += [f"component name=r{l} type=NormalizeComponent dim={out} block-dim={blk} target-rms={rms} add-log-stddev={'true' if als else 'false'}", f"component-node name=r{l} component=r{l} input=n{l}"]
                prev = f"r{l}"; dim = out + (out // blk if als else 0)
        tail = str(rng.choice(["LogSoftmaxComponent", "SoftmaxComponent", ""]))
        if tail: lines += [f"component name=o type={tail} dim={dim}", f"component-node name=o component=o input={prev}"]; prev = "o"
        lines.append(f"output-node name=output input={prev}")
        open(f"{td}/n.config", "w").write("\n".join(lines) + "\n")
        r = subprocess.run([os.path.join(binr, "nnet3-init"), f"--srand={it}", f"{td}/n.config", f"{td}/n.raw"], capture_output=True, text=True, env=env); assert r.returncode == 0, r.stderr[-1500:]
        feats = (rng.standard_normal((int(rng.integers(1, 40)), 12)) * float(rng.choice([0.5, 3.0, 20.0]))).astype(np.float32); kio.write_ark(f"{td}/f.ark", {"u": feats}); s_ = int(rng.choice([1, 3]))
        r = subprocess.run([os.path.join(binr, "nnet3-compute"), "--use-gpu=no", f"--frame-subsampling-factor={s_}", f"{td}/n.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark"], capture_output=True, text=True, env=env); assert r.returncode == 0, r.stderr[-1500:]
        ref = kio.read_ark(f"{td}/o.ark")["u"]; mine = no.compute(no.read_nnet(f"{td}/n.raw"), feats, s_)
        assert mine.shape == ref.shape, (it, mine.shape, ref.shape)
        worst = max(worst, float(np.abs(mine - ref).max())); assert np.abs(mine - ref).max() <= 2e-5, (it, np.abs(mine - ref).max(), lines)
    print("worst |oracle - nnet3-compute| over the fuzz:", worst)
