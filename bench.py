#!/usr/bin/env python3
"""bench.py -- RTFx of the batched acoustic pipeline on MI355X (driver contract in the task statement).

A "step" = one pass of the hot path over one batch of synthetic 16 kHz audio already resident in HBM:
  fbank (k3_feat_compute_batch) -> 17-layer TDNN-F forward (k3_nnet_forward) [-> HCLG lattice decode once the
  decoder lands].  value = audio seconds processed by ALL ranks / wall seconds (max over ranks).
Weak scaling: every rank processes its own --utts utterances; no data-path collective.
Extra objects in the JSON line: "roofline" (dominant kernel vs its CDNA4 peak, timed with HIP events on the
launch stream) and "cpu_baseline" (the reference's own compute-fbank-feats + nnet3-compute binaries from
oracle/_ref when present, else the oracle port; rank 0, N=1 only, bounded sample)."""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def cpu_baseline(model_path, utt_seconds, budget_s=20.0):
    """Reference binaries (kind=reference) on ONE host core, bounded sample of the same workload."""
    from oracle import kaldi_io as kio
    from kaldi_amd import synth
    bindir = os.path.join(ROOT, "oracle", "_ref", "bin")
    n_utts = 4
    with tempfile.TemporaryDirectory() as td:
        scp = []
        for i in range(n_utts):
            kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(int(16000 * utt_seconds), 1234 + i)); scp.append(f"u{i} {td}/u{i}.wav")
        open(f"{td}/wav.scp", "w").write("\n".join(scp) + "\n")
        if os.path.exists(os.path.join(bindir, "nnet3-compute")):
            env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
            t0 = time.time()
            subprocess.check_call([f"{bindir}/compute-fbank-feats", "--dither=0", "--num-mel-bins=40", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], env=env, stderr=subprocess.DEVNULL)
            t1 = time.time()
            subprocess.check_call([f"{bindir}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", "--frames-per-chunk=150", model_path, f"ark:{td}/f.ark", f"ark:{td}/o.ark"], env=env, stderr=subprocess.DEVNULL)
            t2 = time.time()
            audio = n_utts * utt_seconds
            return {"value": audio / (t2 - t0), "unit": "RTFx (audio-s/wall-s)", "cores": 1, "kind": "reference",
                    "sample": f"{n_utts} x {utt_seconds:g} s utts through the reference's compute-fbank-feats ({audio/(t1-t0):.0f}x RT) + nnet3-compute ({audio/(t2-t1):.0f}x RT), 1 core, MKL sequential"}
        from oracle import feat_oracle as fo, nnet3_oracle as no
        net = no.read_nnet(model_path); t0 = time.time()
        for i in range(n_utts):
            w, _ = kio.read_wav(f"{td}/u{i}.wav")
            no.compute(net, fo.compute_features(w.astype(np.float32), fo.fbank_opts(dither=0.0, num_bins=40)), 3)
        dt = time.time() - t0
        return {"value": n_utts * utt_seconds / dt, "unit": "RTFx (audio-s/wall-s)", "cores": 1, "kind": "port",
                "sample": f"{n_utts} x {utt_seconds:g} s utts through oracle/feat_oracle.c + oracle/nnet3_oracle.py (numpy BLAS)"}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=512); ap.add_argument("--utt-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local); dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    import __graft_entry__ as ge
    if rank == 0: ge.build()
    if world > 1: dist.barrier()
    from kaldi_amd import feat, nnet3, synth

    U, nsamp = args.utts, int(16000 * args.utt_seconds)
    # synthetic workload: Gaussian PCM16 sigma 3000, per-rank seed; 17L-768/96-6024 TDNN-F, seed 1
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    waves = (torch.randn(U * nsamp, generator=g) * 3000).round().clamp(-32768, 32767).to(dev)
    sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
    wo, fo, total_frames, fo_h = sf.offsets([nsamp] * U, dev)
    model_path = os.path.join(tempfile.gettempdir(), f"k3_bench_tdnnf_{rank}.raw")
    # BatchNorm calibration on real fbank features of this workload (first utterance)
    calib = sf.ComputeFeatures(waves[:nsamp].contiguous(), *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
    net_spec = synth.make_tdnnf(seed=1, calib_feats=calib); net_spec.write(model_path)
    net = nnet3.Nnet(model_path)
    nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
    feats = torch.empty((total_frames, sf.dim), dtype=torch.float32, device=dev)
    loglikes = torch.empty((nb.total_out_rows, net.info.output_dim), dtype=torch.float32, device=dev)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    def step(timed=None):
        if timed is not None: ev[0].record()
        sf.ComputeFeatures(waves, wo, fo, total_frames, out=feats)
        if timed is not None: ev[1].record()
        nb.forward(feats, out=loglikes)
        if timed is not None: ev[2].record()
    for _ in range(args.warmup): step()
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t_feat = t_nnet = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
        # events are read after the loop's final sync; accumulate lazily
        torch.cuda.current_stream().synchronize(); t_feat += ev[0].elapsed_time(ev[1]); t_nnet += ev[1].elapsed_time(ev[2])
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = t.item()
    audio_s = U * args.utt_seconds * world * args.steps
    if rank == 0:
        nnet_ms = t_nnet / args.steps
        line = {"metric": "RTFx (audio-s/wall-s) batched fbank + TDNN-F forward (HCLG decode: not yet in the timed path)",
                "value": audio_s / dt, "unit": "RTFx (audio-s/wall-s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic (Gaussian PCM16 sigma=3000 seed 1234+rank; random-init BN-calibrated 17L-768/96-6024 TDNN-F seed 1)",
                "config": {"workload": f"configs[1]: fbank(40) + 17-layer TDNN-F forward, {U} x {args.utt_seconds:g} s utts per GPU, frame-subsampling 3",
                           "utts_per_gpu": U, "frames_per_utt": fo_h[1], "output_rows": int(nb.total_out_rows), "params": int(net.info.num_params), "parallelism": f"utterance-shard x{world}"},
                "stage_ms": {"fbank": t_feat / args.steps, "nnet3": nnet_ms},
                "roofline": {"bound": "mfma", "kernel": "k3_tdnn_gemm_kernel (all 35 launches of one forward)", "achieved": nb.flops / (nnet_ms * 1e-3) / 1e12,
                             "peak": 157.3, "unit": "TFLOP/s", "frac": nb.flops / (nnet_ms * 1e-3) / 1e12 / 157.3, "traffic": None,
                             "note": "achieved = exact sum(2MNK) of the launched GEMMs / HIP-event time of the forward on the launch stream; FP32 MFMA peak"}}
        if world == 1 and not args.no_cpu_baseline:
            try: line["cpu_baseline"] = cpu_baseline(model_path, args.utt_seconds)
            except Exception as e: line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line))
    if world > 1: dist.destroy_process_group()

if __name__ == "__main__":
    main()
