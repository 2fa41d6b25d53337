"""Multi-GPU plumbing: one process per GPU, utterances sharded round-robin, the read-only decoding graph
loaded on rank 0 and broadcast ONCE (RCCL over xGMI when the process group is "nccl"; gloo on CPU in the tests), zero
steady-state communication, two scalars reduced at the end for the RTFx report.  The reference has no multi-GPU path at
all (one process per decode job, egs/wsj/s5/steps/nnet3/decode.sh:123 `lat.JOB.gz`); per-rank lattice archives keep that
contract."""
import numpy as np, torch
import torch.distributed as dist

def shard_utterances(num_utts, rank, world):
    """static round-robin: utterance i is decoded by rank i mod world (what `utils/split_scp.pl` does for --nj jobs)"""
    return list(range(rank, num_utts, world))

def broadcast_host_fst(fst, rank, world, src=0):
    """broadcast a kaldi_amd.fst.Fst held on `src` to every rank through the process group (CPU tensors => gloo)."""
    from .fst import Fst
    if world == 1: return fst
    meta = torch.zeros(3, dtype=torch.int64)
    if rank == src: meta[:] = torch.tensor([fst.num_states, fst.num_arcs, fst.start])
    dist.broadcast(meta, src)
    S, A, start = (int(v) for v in meta)
    arrs = []
    for name, n, dt in (("arc_offsets", S + 1, torch.int32), ("ilabel", A, torch.int32), ("olabel", A, torch.int32), ("weight", A, torch.float32),
                        ("nextstate", A, torch.int32), ("final", S, torch.float32)):
        t = torch.from_numpy(getattr(fst, name).copy()) if rank == src else torch.empty(n, dtype=dt)
        dist.broadcast(t, src); arrs.append(t.numpy())
    return fst if rank == src else Fst(start, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], arrs[5])

def broadcast_graph(fst, tid2pdf, rank, world, device, src=0):
    """rank `src` converts the host Fst to the packed device CSR image (k3_fst_create); the image is broadcast once as raw
    bytes over the device process group (RCCL), the other ranks attach it (k3_fst_create_empty + k3_fst_import_image)."""
    from . import decoder
    if world == 1: return decoder.CudaFst(fst, tid2pdf)
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    cf = None
    if rank == src:
        cf = decoder.CudaFst(fst, tid2pdf)
        meta[:] = torch.tensor([fst.num_states, fst.num_arcs, fst.start, cf.image()[1]], device=device)
    dist.broadcast(meta, src)
    S, A, start, nbytes = (int(v) for v in meta.cpu())
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src: cf.export_image(buf)
    dist.broadcast(buf, src)
    if rank != src:
        cf = decoder.CudaFst.empty(S, A, start); cf.import_image(buf)
    torch.cuda.synchronize(device)
    return cf

def broadcast_graph_abi(fst, tid2pdf, rank, world, id_file, src=0, timeout_s=120):
    """The same broadcast through the PRODUCT's C ABI (what batched-wav-nnet3-cuda2 --nccl-id-file does): k3_comm_create (an RCCL communicator of the library's own, the 128-byte
    ncclUniqueId through `id_file`) + k3_fst_bcast (shape, then the packed device image, one ncclBroadcast each) + k3_comm_destroy.  Returns (CudaFst, ranks of the communicator).
    The caller sets the device first (one process per GPU)."""
    import ctypes
    from . import decoder, lib
    L = lib.load()
    if world == 1: return decoder.CudaFst(fst, tid2pdf), 1
    assert src == 0, "k3_comm_create: rank 0 publishes the id"
    cf = decoder.CudaFst(fst, tid2pdf) if rank == src else None
    comm = ctypes.c_void_p()
    lib.check(L.k3_comm_create(str(id_file).encode(), rank, world, int(timeout_s), ctypes.byref(comm)))
    try:
        h = ctypes.c_void_p(cf._h.value if cf is not None else None)
        lib.check(L.k3_fst_bcast(ctypes.byref(h), comm, src, rank, None))
        torch.cuda.synchronize()
        if cf is None: cf = decoder.CudaFst.adopt(h.value)
    finally:
        L.k3_comm_destroy(comm)
    return cf, world

def reduce_rtfx(audio_s, wall_s, device=None):
    """whole-job RTFx = sum(audio) / max(wall) over ranks"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1: return audio_s / wall_s
    a = torch.tensor([audio_s], dtype=torch.float64, device=device); w = torch.tensor([wall_s], dtype=torch.float64, device=device)
    dist.all_reduce(a, op=dist.ReduceOp.SUM); dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return a.item() / w.item()

def allreduce_gradients(grads, world=None, bucket_bytes=64 << 20):
    """Synchronous data-parallel chain training: the ranks' gradients (a list of tensors, the same shapes on every rank, each
    computed on that rank's share of the minibatch with the objective SUMMED over sequences) are summed in place over the process group --
    RCCL over xGMI for CUDA tensors, gloo on CPU in the tests.  The tensors travel in flat buckets of ~bucket_bytes: xGMI is point-to-point
    (ring collectives are per-link bound), so a few large all-reduces beat one per parameter matrix.  (C ABI: k3_comm_allreduce_f32.)"""
    if world is None: world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1 or not grads: return grads
    bucket, size = [], 0
    def flush():
        nonlocal bucket, size
        if not bucket: return
        flat = torch.cat([g.reshape(-1) for g in bucket]); dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        o = 0
        for g in bucket: g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()
        bucket, size = [], 0
    for g in grads:
        bucket.append(g); size += g.numel() * g.element_size()
        if size >= bucket_bytes: flush()
    flush()
    return grads
