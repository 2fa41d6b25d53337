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
