"""CPU tests of the host-side C++ glue (kaldi_amd/host): Kaldi file formats and CLI contract, no GPU involved.
The TransitionModel parser is pinned against the REFERENCE's own class (oracle/_ref/bin/dump-tid2pdf links
hmm/transition-model.cc) on binary and text .mdl files (the text one written by the reference's nnet3-am-copy)."""
import os, subprocess, numpy as np, pytest
from kaldi_amd import synth
from kaldi_amd.fst import Fst
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi_amd", "bin"); REF = os.path.join(ROOT, "oracle", "_ref", "bin")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))

@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()

def _run(*a, env=None): return subprocess.run(list(a), capture_output=True, text=True, env=env)

def test_transition_model_parser_matches_the_reference_class(tmp_path):
    if not os.path.exists(os.path.join(REF, "dump-tid2pdf")): pytest.skip("oracle/_ref not built (needs /root/reference)")
    N = 30; mdl = str(tmp_path / "m.mdl")
    synth.make_tdnn(seed=1, dim=64, num_pdfs=N).write(mdl, as_mdl=True, num_pdfs=N, left_context=5, right_context=5)
    ref = _run(os.path.join(REF, "dump-tid2pdf"), mdl, env=ENV).stdout
    assert ref.split()[:2] == [str(N), str(2 * N)]
    assert _run(os.path.join(BIN, "k3-host-tool"), "tid2pdf", mdl).stdout == ref
    txt = str(tmp_path / "m.txt.mdl")
    assert _run(os.path.join(REF, "nnet3-am-copy"), "--binary=false", mdl, txt, env=ENV).returncode == 0
    assert _run(os.path.join(BIN, "k3-host-tool"), "tid2pdf", txt).stdout == ref
    assert np.array_equal(np.array(ref.split()[2:], int), synth.tid2pdf(N)[1:])

def test_transition_id_phone_info_matches_the_reference_class(tmp_path):
    """TransitionIdToPhone / IsSelfLoop / TransitionIdIsStartOfPhone for every transition-id (what the phone-level determinization
    pass asks of the model) against the reference's TransitionModel, binary and text .mdl"""
    if not os.path.exists(os.path.join(REF, "dump-tidinfo")): pytest.skip("oracle/_ref not built (needs /root/reference)")
    N = 30; mdl = str(tmp_path / "m.mdl")
    synth.make_tdnn(seed=1, dim=64, num_pdfs=N).write(mdl, as_mdl=True, num_pdfs=N, left_context=5, right_context=5)
    ref = _run(os.path.join(REF, "dump-tidinfo"), mdl, env=ENV).stdout
    assert len(ref.splitlines()) > N
    assert _run(os.path.join(BIN, "k3-host-tool"), "tidinfo", mdl).stdout == ref
    txt = str(tmp_path / "m.txt")
    assert _run(os.path.join(REF, "nnet3-am-copy"), "--binary=false", mdl, txt, env=ENV).returncode == 0
    assert _run(os.path.join(BIN, "k3-host-tool"), "tidinfo", txt).stdout == ref

def test_openfst_binary_reader_and_writer_round_trip(tmp_path):
    f = synth.make_hclg(300, 700, 20, seed=3, start_degree=8)
    a, b = str(tmp_path / "g.fst"), str(tmp_path / "g2.fst")
    f.write_openfst(a)
    r = _run(os.path.join(BIN, "k3-host-tool"), "fstinfo", a)
    assert r.stdout.split()[:3] == [str(f.num_states), str(f.num_arcs), str(f.start)], r.stderr
    assert _run(os.path.join(BIN, "k3-host-tool"), "copy-fst", a, b).returncode == 0
    g = Fst.read_openfst(b)
    assert all(np.array_equal(getattr(f, k), getattr(g, k)) for k in ("arc_offsets", "ilabel", "olabel", "weight", "nextstate", "final")) and g.start == f.start

def test_cli_exit_codes_follow_the_reference():
    """usage -> 1 (batched-wav-nnet3-cuda2.cc:110-113), exception -> -1 with the message on stderr (:256-259)"""
    exe = os.path.join(BIN, "batched-wav-nnet3-cuda2")
    r = _run(exe); assert r.returncode == 1 and "Usage: batched-wav-nnet3-cuda2" in r.stderr
    r = _run(exe, "--no-such-option=1", "a", "b", "c", "d"); assert r.returncode == 255 and "Invalid option" in r.stderr
    r = _run(exe, "--add-pitch=true", "a", "b", "c", "d"); assert r.returncode == 255 and "outside the accelerated path" in r.stderr
    r = _run(os.path.join(BIN, "compute-fbank-feats-cuda"), "only-one-arg"); assert r.returncode == 1
    r = _run(exe, "--help"); assert r.returncode == 0 and "--lattice-beam" in r.stderr and "--max-batch-size" in r.stderr

def test_streaming_and_cmvn_programs_usage_and_option_errors():
    """the newer programs follow the same contract: usage -> 1, bad option / unsupported feature -> message on stderr and -1, --help -> 0"""
    on = os.path.join(BIN, "batched-wav-nnet3-cuda-online")
    r = _run(on); assert r.returncode == 1 and "Usage: batched-wav-nnet3-cuda-online" in r.stderr
    r = _run(on, "--feature-type=plp", "--print-endpoints=true", "a", "b", "c", "d"); assert r.returncode == 255 and "Invalid feature type" in r.stderr      # (--print-endpoints itself is an option of the program since round 6)
    r = _run(on, "--help"); assert "--print-hypotheses" in r.stderr and "--endpoint.rule2.min-trailing-silence" in r.stderr and "--lattice-postprocessor-rxfilename" in r.stderr and "not supported" not in r.stderr
    r = _run(on, "--help"); assert r.returncode == 0 and "--frames-per-chunk" in r.stderr and "--num-channels" in r.stderr
    cm = os.path.join(BIN, "apply-cmvn-online-cuda")
    r = _run(cm, "a"); assert r.returncode == 1 and "Usage: apply-cmvn-online-cuda" in r.stderr
    r = _run(cm, "--skip-dims=1:x", "a", "b", "c"); assert r.returncode == 255 and "skip-dims" in r.stderr
    r = _run(cm, "--help"); assert r.returncode == 0 and "--cmn-window" in r.stderr and "--spk2utt" in r.stderr


def test_command_line_options_win_over_config_files_whatever_their_order(tmp_path):
    """util/parse-options.cc:329-371: every --config file is read in a first pass, the command line is applied afterwards"""
    tool = os.path.join(BIN, "k3-host-tool"); c1 = str(tmp_path / "a.conf"); c2 = str(tmp_path / "b.conf")
    open(c1, "w").write("# recipe defaults\n--beam=3.0\n--max-active=100 # trailing comment\n--flag=true\n")
    open(c2, "w").write("--name=from_b\n--beam=4.5\n")
    out = lambda *a: _run(tool, "parse-options", "--print-args=false", *a).stdout.strip()
    assert out("--beam=100", "--config=" + c1) == "beam=100 max-active=100 flag=true name=dflt nargs=0"
    assert out("--config=" + c1, "--beam=100") == "beam=100 max-active=100 flag=true name=dflt nargs=0"
    assert out("--config=" + c1, "--config=" + c2, "--flag=false", "x", "y") == "beam=4.5 max-active=100 flag=false name=from_b nargs=2"
    assert out("--name=cli", "--config=" + c2, "--", "--beam=1") == "beam=4.5 max-active=7 flag=false name=cli nargs=1"
    bad = str(tmp_path / "bad.conf"); open(bad, "w").write("--no-such-option=1\n")
    assert _run(tool, "parse-options", "--config=" + bad).returncode != 0
