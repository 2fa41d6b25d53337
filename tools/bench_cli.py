#!/usr/bin/env python3
"""End-to-end throughput of the drop-in program, files in -> lattice archive out (the reference's own way of quoting RTFx:
batched-wav-nnet3-cuda2's closing "RealTimeX" line).  Unlike bench.py this includes wav reading, PCIe, lattice post-processing,
determinization on the host pool and archive writing.
  python tools/bench_cli.py [utts=512] [seconds=10] [iterations=2]
Writes U synthetic wav files, the bench model (.mdl) and the bench HCLG to a temp dir, then runs the program three ways:
--write-lattice=false, --determinize-lattice=false (raw lattices), default (determinized CompactLattices)."""
import os, sys, time, tempfile, subprocess, wave, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from kaldi_amd import feat, synth
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
td = tempfile.mkdtemp(prefix="k3cli_"); nsamp = int(16000 * secs); dev = torch.device("cuda:0")
t0 = time.time()
g = torch.Generator(device="cpu"); g.manual_seed(1234)
pcm = (torch.randn(U * nsamp, generator=g) * 3000).round().clamp(-32768, 32767).to(torch.int16).numpy().reshape(U, nsamp)
with open(f"{td}/wav.scp", "w") as scp:
    for u in range(U):
        with wave.open(f"{td}/u{u}.wav", "wb") as w: w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm[u].tobytes())
        scp.write(f"utt{u:04d} {td}/u{u}.wav\n")
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
w0 = torch.from_numpy(pcm[0].astype(np.float32)).to(dev)
calib = sf.ComputeFeatures(w0, *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
net = synth.make_tdnnf(seed=1, calib_feats=calib); P = 6024
net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=P)
synth.make_hclg(2_000_000, 5_000_000, P).write_openfst(f"{td}/HCLG.fst")
open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
print("setup %.1f s (%d wav files of %.0f s, model, graph) in %s" % (time.time() - t0, U, secs, td), flush=True)
exe = os.path.join(ROOT, "kaldi_amd", "bin", "batched-wav-nnet3-cuda2")
common = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000",
          f"--max-batch-size={U}", f"--iterations={iters}", "--verbose=1", "--cuda-decoder-copy-threads=%s" % os.environ.get("COPY_THREADS", "8"), "--main-q-capacity=65536", "--aux-q-capacity=131072", f"--ntokens-pre-allocated={int(4500 * secs * 33.4) + 65536}"]
for name, extra, out in (("no lattice output", ["--write-lattice=false"], "ark:/dev/null"), ("raw lattices", ["--determinize-lattice=false"], f"ark:{td}/raw.ark"), ("determinized (default)", [], f"ark:{td}/det.ark")):
    t0 = time.time()
    r = subprocess.run([exe] + common + extra + [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", out], capture_output=True, text=True)
    last = [l for l in r.stderr.splitlines() if "RealTimeX" in l or "Decoded" in l]
    print("%-24s rc=%d wall %.1f s | %s" % (name, r.returncode, time.time() - t0, " | ".join(l.split(") ", 1)[-1] for l in last)), flush=True)
    if r.returncode != 0: print(r.stderr[-2000:])
    for l in r.stderr.splitlines():
        if l.startswith("VLOG"): print("   ", l.split(") ", 1)[-1])
if os.environ.get("K3CLI_ONLINE"):      # the streaming program on the same files: every file played as an audio stream in chunks of --frames-per-chunk frames over --num-channels channels
    exe_o = os.path.join(ROOT, "kaldi_amd", "bin", "batched-wav-nnet3-cuda-online")
    for fpc in (51, 150):
        args = ["--feature-type=fbank", f"--fbank-config={td}/fbank.conf", "--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", f"--max-batch-size={U}", f"--num-channels={U}",
                f"--frames-per-chunk={fpc}", f"--iterations={iters}", "--main-q-capacity=65536", "--aux-q-capacity=131072", "--write-lattice=true"]
        t0 = time.time()
        r = subprocess.run([exe_o] + args + [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark:{td}/online.ark"], capture_output=True, text=True)
        last = [l for l in r.stderr.splitlines() if "RealTimeX" in l or "Decoded" in l]
        print("online, %3d frames per chunk rc=%d wall %.1f s | %s" % (fpc, r.returncode, time.time() - t0, " | ".join(l.split(") ", 1)[-1] for l in last)), flush=True)
        if r.returncode != 0: print(r.stderr[-1500:])
    if os.environ.get("K3CLI_ONLINE") == "profile":      # kernel trace + stats of the 51-frame run -> gpurun_out/prof_online
        out = os.path.join(ROOT, "gpurun_out", "prof_online"); os.makedirs(out, exist_ok=True)
        args[9] = "--frames-per-chunk=51"; args[10] = "--iterations=%s" % os.environ.get("K3CLI_PROFILE_ITERS", "1")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--"] + [exe_o] + args + [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark:{td}/online2.ark"],
                           capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        print("profiled online run rc=%d | %s" % (r.returncode, " | ".join(l.split(") ", 1)[-1] for l in r.stderr.splitlines() if "RealTimeX" in l)), flush=True)
if os.environ.get("K3CLI_PROFILE"):      # kernel trace of one more run of the default configuration -> gpurun_out/prof_cli (rocprofv3 --kernel-trace --stats)
    out = os.path.join(ROOT, "gpurun_out", "prof_cli"); os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--"] + [exe] + common + [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark:{td}/det2.ark"], capture_output=True, text=True, cwd="/tmp", env=env)
    print("profiled run rc=%d | %s" % (r.returncode, " | ".join(l.split(") ", 1)[-1] for l in r.stderr.splitlines() if "RealTimeX" in l)), flush=True)
for f in ("raw.ark", "det.ark"):
    if os.path.exists(f"{td}/{f}"): print(f, "%.1f MB" % (os.path.getsize(f"{td}/{f}") / 1e6))
