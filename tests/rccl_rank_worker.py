"""One rank of tests/test_parallel_gpu.py::test_c_abi_rccl_collectives_on_two_devices: the multi-GPU entry points of the C ABI with a real N > 1 communicator.
usage: rccl_rank_worker.py <rank> <world> <id-file> <out-json>"""
import ctypes, hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import synth, decoder, parallel, lib

def main():
    rank, world, id_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.cuda.set_device(rank % torch.cuda.device_count()); L = lib.load(); N = 50
    graph = synth.make_hclg(2000, 5000, N, seed=1, start_degree=40) if rank == 0 else None      # only rank 0 has the graph
    cf, n = parallel.broadcast_graph_abi(graph, synth.tid2pdf(N), rank, world, id_file, timeout_s=60)
    ptr, nbytes = cf.image(); buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); cf.export_image(buf)
    res = {"rank": rank, "ranks": n, "states": cf.num_states, "arcs": cf.num_arcs, "start": cf.start, "image_sha": hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()}
    rng = np.random.default_rng(0); ll = (rng.standard_normal((40, N)) * 2.5).astype(np.float32)      # every rank decodes the same utterance on its copy of the graph
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, literal_order=1), 1, N)
    dec.DecodeBatch(torch.from_numpy(ll).cuda(), np.array([0, 40])); lat = dec.GetRawLattices(copy=True)[0]
    h = hashlib.sha256()
    for a in lat.canonical(): h.update(np.ascontiguousarray(a).tobytes())
    res["lattice_sha"] = h.hexdigest(); res["lattice_arcs"] = int(lat.num_arcs)
    # the gradient exchange of data-parallel training (k3_comm_allreduce_f32) on a second communicator of the library's own
    comm = ctypes.c_void_p(); lib.check(L.k3_comm_create((id_file + ".2").encode(), rank, world, 60, ctypes.byref(comm)))
    L.k3_comm_allreduce_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    g = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * (rank + 1)
    lib.check(L.k3_comm_allreduce_f32(comm, g.data_ptr(), g.numel(), None)); torch.cuda.synchronize()
    want = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * (world * (world + 1) // 2)
    res["allreduce_ok"] = bool(torch.equal(g, want)); L.k3_comm_destroy(comm)
    json.dump(res, open(out, "w"))

if __name__ == "__main__":
    main()
