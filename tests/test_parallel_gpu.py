"""GPU: the multi-GPU entry points of the C ABI on the one GPU the test box has.  k3_comm_create + k3_fst_bcast with a one-rank RCCL communicator
(rank 0 writes the ncclUniqueId file, creates the communicator, broadcasts shape and image in place): the RCCL binding (dlopen librccl.so.1, call
sequence, data types) is exercised end to end; a graph attached from an image (the receiving side's code path: k3_fst_create_empty + image bytes)
decodes identically.  N > 1 ranks need N GPUs: the driver's scaling run covers that; tests/test_parallel_cpu.py covers the sharding logic."""
import ctypes, os, numpy as np, pytest, torch
from kaldi_amd import synth
pytestmark = pytest.mark.gpu

def test_graph_broadcast_through_the_c_abi_single_rank(tmp_path):
    from kaldi_amd import decoder, lib
    L = lib.load(); N = 50
    f = synth.make_hclg(2000, 5000, N, seed=1, start_degree=40); t2p = synth.tid2pdf(N); cf = decoder.CudaFst(f, t2p)
    comm = ctypes.c_void_p()
    lib.check(L.k3_comm_create(str(tmp_path / "nccl.id").encode(), 0, 1, 10, ctypes.byref(comm)))
    assert os.path.getsize(tmp_path / "nccl.id") == 144      # the 128-byte ncclUniqueId + {magic, run identity}; a one-rank communicator leaves it in place
    h = ctypes.c_void_p(cf._h.value)
    lib.check(L.k3_fst_bcast(ctypes.byref(h), comm, 0, 0, None))
    assert h.value == cf._h.value                       # the root keeps its graph
    # the gradient exchange of data-parallel training on the same communicator: in-place sum over the (one) rank
    L.k3_comm_allreduce_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    g = torch.arange(100000, dtype=torch.float32, device="cuda") * 0.25; g0 = g.clone()
    lib.check(L.k3_comm_allreduce_f32(comm, g.data_ptr(), g.numel(), None)); torch.cuda.synchronize()
    assert torch.equal(g, g0)
    L.k3_comm_destroy(comm)
    rng = np.random.default_rng(0); ll = (rng.standard_normal((40, N)) * 2.5).astype(np.float32)
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, literal_order=1), 1, N)
    dec.DecodeBatch(torch.from_numpy(ll).cuda(), np.array([0, 40])); a = dec.GetRawLattices(copy=True)[0]
    # receiving side: an empty graph of the broadcast shape + the image bytes
    ptr, nbytes = cf.image(); buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); cf.export_image(buf)
    cf2 = decoder.CudaFst.empty(f.num_states, f.num_arcs, f.start); cf2.import_image(buf)
    dec2 = decoder.CudaDecoder(cf2, decoder.decoder_config(beam=15.0, lattice_beam=8.0, literal_order=1), 1, N)
    dec2.DecodeBatch(torch.from_numpy(ll).cuda(), np.array([0, 40])); b = dec2.GetRawLattices(copy=True)[0]
    assert a.num_arcs > 0 and a.diff(b) == ""

def test_bench_two_ranks_on_one_device_with_gloo(tmp_path):
    """bench.py's N > 1 path rehearsed on one GPU (K3_DIST_BACKEND=gloo: both ranks share the device): graph built on rank 0, broadcast_graph to
    rank 1, per-rank decode, max-over-ranks timing, one JSON line with n_gpus = 2"""
    import json, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, K3_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr=127.0.0.1", "--master-port=29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--utts", "8", "--utt-seconds", "2", "--graph-states", "20000", "--graph-arcs", "50000", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["decode_stats"]["graph_broadcast_s"] > 0 and line["decode_stats"]["lattice_arcs"] > 0


def test_bench_pipelined_steps_give_the_single_stream_lattices():
    """bench.py's default steps pipeline batches over two streams (batch k+1's upload, features and network queued behind batch k's decoder, double-buffered log-likelihoods;
    k3_decoder_decode_batch does not wait for its own kernels).  The lattices and the determinized lattices of the last batch must be those of --no-pipeline, and the JSON line must
    carry the serial pass's stage times next to the pipelined steps'."""
    import json, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--utts", "24", "--utt-seconds", "3", "--graph-states", "30000", "--graph-arcs", "80000", "--no-cpu-baseline", "--lattice-digest"]
    lines = []
    for extra in ([], ["--no-pipeline"]):
        r = subprocess.run(common + extra, capture_output=True, text=True, timeout=600); assert r.returncode == 0, r.stderr[-3000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    a, b = lines
    for k in ("lattice_states", "lattice_arcs", "determinized_states", "determinized_arcs", "tokens", "emitting_arcs_traversed", "order_sensitive_events"):      # (eps_arcs_traversed counts the fixpoint's re-expansions, which depend on timing)
        assert a["decode_stats"][k] == b["decode_stats"][k] and a["decode_stats"][k] > 0, k
    assert a["decode_stats"]["lattice_digest"] == b["decode_stats"]["lattice_digest"] and len(a["decode_stats"]["lattice_digest"]) == 32      # the lattices themselves (canonical form, cost bits included)
    assert a["two_pass"]["lattice_arcs"] == b["two_pass"]["lattice_arcs"] and a["two_pass"]["lattice_digest"] == b["two_pass"]["lattice_digest"]
    assert a["pipeline"].startswith("batch k+1") and b["pipeline"].startswith("none") and a["stage_ms_in_pipeline"] is not None and b["stage_ms_in_pipeline"] is None
    assert set(a["stage_ms"]) == set(b["stage_ms"]) and all(v > 0 for v in a["stage_ms"].values())
    assert a["roofline"]["frac"] > 0 and a["roofline_gemm"]["frac_back_to_back"] > 0 and a["cpu_baseline"] is None if "cpu_baseline" in a else True


def test_c_abi_rccl_collectives_on_two_devices(tmp_path):
    """N > 1 through the product's C ABI, the moment two devices are visible (skipped on the one-GPU test box; the driver's multi-GPU node runs it): k3_comm_create with a real
    2-rank RCCL communicator, k3_fst_bcast from rank 0 (the only rank that built the graph) -- identical image bytes on both ranks, identical lattices decoded from them --
    and k3_comm_allreduce_f32 summing rank-dependent buffers."""
    import json, subprocess, sys
    ndev = torch.cuda.device_count()
    if ndev < 2: pytest.skip("torch.cuda.device_count() == %d: needs two GPUs (RCCL refuses two ranks on one device); with >= 2 visible devices this test runs, there is no other skip" % ndev)
    assert ndev >= 2      # (the only way past the skip above; nothing below may skip)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, K3_COMM_NONCE="test-%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py"), str(r), "2", str(tmp_path / "nccl.id"), str(tmp_path / f"r{r}.json")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    try: outs = [p.communicate(timeout=240)[0] for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs: p.kill()      # (the two workers this test started, by handle)
        pytest.fail("the two RCCL ranks did not finish within 240 s: " + " | ".join((p.communicate()[0] or b"").decode(errors="replace")[-800:] for p in procs))
    assert all(p.returncode == 0 for p in procs), outs
    a, b = (json.load(open(tmp_path / f"r{r}.json")) for r in range(2))
    assert a["ranks"] == b["ranks"] == 2 and (a["states"], a["arcs"], a["start"]) == (b["states"], b["arcs"], b["start"]) == (2000, a["arcs"], a["start"])
    assert a["image_sha"] == b["image_sha"] and a["lattice_sha"] == b["lattice_sha"] and a["lattice_arcs"] > 0
    assert a["allreduce_ok"] and b["allreduce_ok"]
