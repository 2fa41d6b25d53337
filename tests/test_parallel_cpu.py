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
    def run(stale_payload, nonce, age):
        if nonce is None: monkeypatch.delenv("K3_COMM_NONCE", raising=False); monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
        else: monkeypatch.setenv("K3_COMM_NONCE", nonce)
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
    run(b"\x05" * 128 + struct.pack("<QQ", magic, h), "none", 3600)        # the SAME run identity (torchrun's static rendezvous calls every run "none") on an hour-old file: age decides
