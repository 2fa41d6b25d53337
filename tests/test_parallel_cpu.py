"""world_size-2 gloo tests (CPU) of the N>1 plumbing: utterance sharding, the one-off graph broadcast, the RTFx reduce."""
import os, sys, numpy as np, pytest, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kaldi_amd import parallel, synth
    fst = synth.make_hclg(500, 1300, 20, seed=9, start_degree=10) if rank == 0 else None
    got = parallel.broadcast_host_fst(fst, rank, world)
    ref = synth.make_hclg(500, 1300, 20, seed=9, start_degree=10)
    same = all(np.array_equal(getattr(got, n), getattr(ref, n)) for n in ("arc_offsets", "ilabel", "olabel", "weight", "nextstate", "final")) and got.start == ref.start
    shard = parallel.shard_utterances(11, rank, world)
    rtfx = parallel.reduce_rtfx(100.0 * (rank + 1), 2.0 + rank)
    q.put((rank, same, shard, rtfx))
    dist.destroy_process_group()

def test_world2_gloo_graph_broadcast_sharding_and_rtfx_reduce():
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29500 + os.getpid() % 500
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(30)
    assert all(r[1] for r in res)
    assert res[0][2] == [0, 2, 4, 6, 8, 10] and res[1][2] == [1, 3, 5, 7, 9]
    assert sorted(res[0][2] + res[1][2]) == list(range(11))
    for r in res: assert abs(r[3] - 300.0 / 3.0) < 1e-9


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kaldi_amd import parallel
    g = torch.Generator().manual_seed(5); W = [torch.randn(7, 5, generator=g, dtype=torch.float64), torch.randn(3, 7, generator=g, dtype=torch.float64)]; X = torch.randn(12, 5, generator=g, dtype=torch.float64); Y = torch.randn(12, 3, generator=g, dtype=torch.float64)
    def grads(x, y):      # objective summed over the sequences of the (share of the) minibatch
        w = [t.clone().requires_grad_(True) for t in W]
        (((torch.tanh(x @ w[0].T) @ w[1].T) - y) ** 2).sum().backward()
        return [t.grad for t in w]
    mine = parallel.allreduce_gradients(grads(X[rank::world], Y[rank::world]), bucket_bytes=64)      # tiny buckets: several all-reduces
    whole = grads(X, Y)
    q.put((rank, max(float((a - b).abs().max()) for a, b in zip(mine, whole))))
    dist.destroy_process_group()

def test_world2_gloo_summed_gradients_of_a_split_minibatch_equal_the_whole_minibatch():
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29100 + os.getpid() % 300
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(30)
    assert all(r[1] < 1e-12 for r in res), res

def test_comm_rendezvous_refuses_files_of_other_runs(tmp_path, monkeypatch):
    """k3_comm_create's file protocol (k3_comm_exchange_id, no RCCL involved): a file a previous run left behind -- old, or carrying another run's
    identity -- is never taken for this run's; the same path serves run after run"""
    import ctypes, struct, threading, time
    from kaldi_amd import lib
    L = lib.load(); path = str(tmp_path / "nccl.id").encode()
    L.k3_comm_exchange_id.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    def run(stale_payload, nonce, age, launcher_id=None):
        monkeypatch.delenv("K3_COMM_NONCE", raising=False); monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
        if nonce is not None: monkeypatch.setenv("K3_COMM_NONCE", nonce)
        if launcher_id is not None: monkeypatch.setenv("TORCHELASTIC_RUN_ID", launcher_id)
        if stale_payload is not None:
            open(path, "wb").write(stale_payload); t = time.time() - age; os.utime(path, (t, t))
        out = ctypes.create_string_buffer(128); rc = [None]
        th = threading.Thread(target=lambda: rc.__setitem__(0, L.k3_comm_exchange_id(path, 1, 10, 5, None, out)))      # rank 1 starts first and finds the stale file
        th.start(); time.sleep(0.6)
        assert th.is_alive(), "rank 1 accepted a stale file"
        fresh = bytes(range(128)); got0 = ctypes.create_string_buffer(128)
        assert L.k3_comm_exchange_id(path, 0, 10, 5, fresh, got0) == 0
        th.join(10); assert rc[0] == 0 and out.raw == fresh
    magic = 0x4b33636f6d6d3031
    run(b"\x07" * 128 + struct.pack("<QQ", magic, 0), None, 3600)          # no run identity: an hour-old file of a crashed run
    run(b"\x09" * 128 + struct.pack("<QQ", magic, 12345), "run-2", 0)      # a fresh file of ANOTHER run (different nonce), same path reused
    run(b"\x01" * 128, "run-3", 0)                                         # a file in the old format (no magic)
    h = 1469598103934665603
    for ch in b"none": h = ((h ^ ch) * 1099511628211) % (1 << 64)
    run(b"\x05" * 128 + struct.pack("<QQ", magic, h), None, 3600, launcher_id="none")      # the SAME launcher identity (torchrun's static rendezvous calls every run "none") on an hour-old file: age decides
    run((b"\x05" * 128 + struct.pack("<QQ", magic, h))[:100], None, 0, launcher_id="none")   # a TRUNCATED file of this very run identity, fresh: not a record
    run(b"\x05" * 128 + struct.pack("<QQ", magic, h) + b"x", None, 0, launcher_id="none")    # ... and an over-long one

def test_comm_rendezvous_late_rank_with_a_per_run_nonce_is_not_rejected_for_the_files_age(tmp_path, monkeypatch):
    """a rank that enters the rendezvous long after rank 0 published (slow model load) must still be admitted when the run identity is a per-run
    K3_COMM_NONCE -- the mtime window only guards launcher ids that repeat ("none")"""
    import ctypes, time
    from kaldi_amd import lib
    L = lib.load(); path = str(tmp_path / "late.id").encode()
    L.k3_comm_exchange_id.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False); monkeypatch.setenv("K3_COMM_NONCE", "run-late-1")
    fresh = bytes(range(128)); got0 = ctypes.create_string_buffer(128); out = ctypes.create_string_buffer(128)
    assert L.k3_comm_exchange_id(path, 0, 10, 5, fresh, got0) == 0
    t = time.time() - 1000; os.utime(path, (t, t))      # rank 0 published 1000 s ago
    assert L.k3_comm_exchange_id(path, 1, 2, 5, None, out) == 0 and out.raw == fresh

def _rendezvous_rank(path, rank, world, timeout, nonce, delay, q):
    import ctypes, time
    os.environ.pop("TORCHELASTIC_RUN_ID", None)
    if nonce is None: os.environ.pop("K3_COMM_NONCE", None)      # (a launcher that supplies no run identity: only the files' ages and the handshake tell two runs apart)
    else: os.environ["K3_COMM_NONCE"] = nonce
    from kaldi_amd import lib
    L = lib.load()
    L.k3_comm_rendezvous.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.k3_last_error.restype = ctypes.c_char_p
    time.sleep(delay)
    mine = bytes((7 * i + 3) % 256 for i in range(128)); out = ctypes.create_string_buffer(128)
    t0 = time.time(); rc = L.k3_comm_rendezvous(path.encode(), rank, world, timeout, 120, mine if rank == 0 else None, out)
    q.put((rank, rc, out.raw == mine, time.time() - t0, (L.k3_last_error() or b"").decode(), t0, time.time()))

def _run_rendezvous(path, ranks, world, timeout, nonce="fault-run", delays=None):
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=_rendezvous_rank, args=(path, r, world, timeout, nonce, (delays or {}).get(r, 0.0), q)) for r in ranks]
    for p in ps: p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps: p.join(30)
    return {r[0]: r for r in res}

def test_comm_rendezvous_all_ranks_arrive(tmp_path):
    """the arrival handshake k3_comm_create runs in front of ncclCommInitRank (k3_comm_rendezvous, real processes, no RCCL): four ranks entering at different
    times all leave with rank 0's id, and rank 0 leaves only after it has seen the other three"""
    path = str(tmp_path / "rv.id")
    res = _run_rendezvous(path, range(4), 4, 20, delays={0: 0.8, 2: 1.5})
    assert all(r[1] == 0 and r[2] for r in res.values()), res
    assert min(r[6] for r in res.values()) >= max(r[5] for r in res.values()) - 0.06, res      # nobody left before everybody had entered (one 50 ms poll of slack for the clocks' read-out)

def test_comm_rendezvous_a_rank_that_never_arrives_is_an_error_not_a_hang(tmp_path):
    """fault injection: world 3, rank 2 never starts.  Rank 0 returns an error that names rank 2 inside its timeout and withdraws the id file;
    rank 1, which did arrive, returns an error too ("rank 0 never confirmed") instead of walking into ncclCommInitRank, which has no deadline"""
    path = str(tmp_path / "rv.id")
    res = _run_rendezvous(path, [0, 1], 3, 3)
    assert res[0][1] != 0 and "missing: 2" in res[0][4] and res[0][3] < 10, res
    assert res[1][1] != 0 and "never confirmed" in res[1][4] and res[1][3] < 10, res
    assert not os.path.exists(path) and not os.path.exists(path + ".go"), os.listdir(tmp_path)

def test_comm_rendezvous_without_rank_zero_times_out(tmp_path):
    path = str(tmp_path / "rv.id")
    res = _run_rendezvous(path, [1], 2, 2)
    assert res[1][1] != 0 and "timed out" in res[1][4] and res[1][3] < 8, res

def test_comm_rendezvous_ignores_a_confirmation_left_by_another_run(tmp_path):
    """a crashed run's id + .go files (another nonce) at the same path: rank 1, first to start, must not leave with them; once this run's rank 0 arrives both agree"""
    import struct
    path = str(tmp_path / "rv.id"); magic = 0x4b33636f6d6d3031
    open(path, "wb").write(b"\x11" * 128 + struct.pack("<QQ", magic, 999)); open(path + ".go", "wb").write(struct.pack("<QQQ", magic, 999, 42))
    res = _run_rendezvous(path, [0, 1], 2, 20, delays={0: 1.0})
    assert all(r[1] == 0 and r[2] for r in res.values()), res


def test_comm_rendezvous_ignores_a_fresh_confirmation_of_a_crashed_run_with_the_same_identity(tmp_path):
    """no per-run nonce (torchrun's static rendezvous), and a run that crashed seconds ago left BOTH its id file and the matching .go -- same identity, inside
    the age window, hashes that agree.  Rank 1 of the new run starts before rank 0 has cleaned up: it must not leave with the dead run's id (it would block in
    ncclCommInitRank for ever); the .go is older than rank 1's own announcement, so it waits for this run's rank 0 and both end with the new id."""
    import struct
    def fnv(b, h=1469598103934665603):
        for c in b: h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    path = str(tmp_path / "rv.id"); magic = 0x4b33636f6d6d3031; dead = b"\x11" * 128
    open(path, "wb").write(dead + struct.pack("<QQ", magic, 0)); open(path + ".go", "wb").write(struct.pack("<QQQ", magic, 0, fnv(dead)))
    res = _run_rendezvous(path, [0, 1], 2, 20, nonce=None, delays={0: 1.5})
    assert all(r[1] == 0 and r[2] for r in res.values()), res      # r[2]: the id a rank left with is this run's rank-0 id
    assert res[1][3] >= 1.0, res                                       # rank 1 really waited for rank 0
