#!/usr/bin/env python3
"""GPU box: bench-configuration decode with literal_order; kernel times and (with a -DK3_LIT_PROF library, K3HIP_LIB=...) per-phase cycles."""
import os, sys, time, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder, lib as _l
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0"); nsamp = 160000
g = torch.Generator(device="cpu"); g.manual_seed(1234)
waves = (torch.randn(U * nsamp, generator=g) * 3000).round().clamp(-32768, 32767).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
wo, fo, total_frames, fo_h = sf.offsets([nsamp] * U, dev)
calib = sf.ComputeFeatures(waves[:nsamp].contiguous(), *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
mp = os.path.join(tempfile.gettempdir(), "proflit.raw"); synth.make_tdnnf(seed=1, calib_feats=calib).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
ll = nb.forward(sf.ComputeFeatures(waves, wo, fo, total_frames)); torch.cuda.synchronize()
f = synth.make_hclg(); cf = decoder.CudaFst(f, synth.tid2pdf(net.info.output_dim))
CAP = int(os.environ.get("K3_PROF_CAP", 65536))
for literal in (1, 0):
    cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, fast_frame_tokens=int(os.environ.get("K3_FAST", -1)), frame_tokens_cap=CAP, frame_cands_cap=max(131072, CAP + 1) if CAP > 30000 else 65536, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, literal_order=literal)
    dec = decoder.CudaDecoder(cf, cfg, U, net.info.output_dim); dec.SetProfiling(True)
    for it in range(2):
        dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize(); kt = dec.KernelTimes()
    info = dec.LatticeInfo(check=False)
    print("literal" if literal else "default", "U", U, "token passing ms %.2f prune ms %.2f" % kt, "status", np.unique(info[:, 2], return_counts=True), "tokens/frame", info[:, 4].mean() / 334, "max frame", info[:, 6].max(),
          "eps arcs/frame", info[:, 8].mean() / 334, "emit arcs/frame", info[:, 7].mean() / 334)
    if os.environ.get("K3_PRUNE_PROF"):      # library built with -DK3_PRUNE_PROF: cycles of the pruning kernel's stages, per lane
        cyc = np.zeros(16, np.int64); _l.load().k3_decoder_phase_cycles(dec._h, cyc.ctypes.data)
        print("prune kernel cycles/lane:", dict(zip(["last frame", "staging", "emitting links", "eps fixpoint", "offsets", "HBM-path frames"], (cyc[:6] // U).tolist())))
    elif literal and (not os.environ.get("K3HIP_LIB") or os.environ.get("K3_PROF_PATHS")):      # the shipped library: frames by path (LDS-resident / given up and redone / general) and why the fast path gave up
        cyc = np.zeros(16, np.int64); _l.load().k3_decoder_phase_cycles(dec._h, cyc.ctypes.data)
        print("cycles/lane (two decodes): fast frames", cyc[15] // U, "general frames", cyc[11] // U, "| frames: fast", cyc[12], "gave up", cyc[13], "general (incl. redone)", cyc[14], "| give-up reasons", dict(zip(["tokens", "table", "hash", "labels", "worklist", "eps links", "degree", "closure", "queue", "stack", "mismatch"], cyc[:11].tolist())))
    elif literal:
        cyc = np.zeros(16, np.int64); _l.load().k3_decoder_phase_cycles(dec._h, cyc.ctypes.data)
        names = ["cutoff", "hash resize+prepass", "passA+chunk scan", "(non-LDS replay cycles)", "passB+c0", "(non-LDS replay pops)", "order1", "closure", "csr build", "replay", "order2", "publish"]
        nl_cyc, nl_pops = cyc[3], cyc[5]
        if not os.environ.get("K3_LIT_PROF_FINE"): cyc[3] = 0; cyc[5] = 0
        tot = cyc[:12].sum()
        if os.environ.get("K3_LIT_PROF_Q"):      # library built with -DK3_LIT_PROF=3: the frame's phases split further
            qn = ["cutoff (+LDS-path attempt)", "pre-pass", "pass A", "chunk scan", "pass B", "c0", "closure fixpoint", "final costs + passing-arc counts + table clear", "order1", "ids scan + arc slots", "records (step 2)", "initial queue",
                  "replay", "order2", "publish"]
            fr = max(1, cyc[15]); tq = cyc[:15].sum()
            print("frames", int(fr), "cycles/lane/frame", int(tq / fr)); print("phase: cycles per frame (share %):", {n: "%d (%.1f)" % (c / fr, 100.0 * c / tq) for n, c in zip(qn, cyc[:15])})
        elif os.environ.get("K3_LIT_PROF_FINE"):      # library built with -DK3_LIT_PROF=2: sub-phases of the hash-order passes and of the component replay
            sub = ["ho:bitmap", "ho:word scan", "ho:dense+buckets", "ho:leader scan", "ho:group fill", "ho:order", "ho:reset", "cr:init", "cr:union", "cr:count", "cr:scan4", "cr:group roots", "cr:workers", "cr:labels"]
            fr = max(1, cyc[15]); print("sub-phase cycles/lane/frame:", {n: int(c / fr) for n, c in zip(sub, cyc[:14])})
        elif tot:
            print("phase share %:", {n: round(100.0 * c / tot, 1) for n, c in zip(names, cyc[:12])})
            fr = max(1, cyc[15]); print("frames counted (two decodes)", int(cyc[15]), "Gcycles", tot / 1e9, "cycles/lane/frame", tot / fr, "replay pops/frame", cyc[12] / fr, "tokens/frame", cyc[13] / fr, "LDS-replay frames frac", cyc[14] / fr)
            print("replay: LDS mode cycles/pop", (cyc[9] - nl_cyc) / max(1, cyc[12] - nl_pops), "pops", (cyc[12] - nl_pops) / fr, "| other modes cycles/pop", nl_cyc / max(1, nl_pops), "pops/frame", nl_pops / fr, "share of replay", nl_cyc / max(1, cyc[9]))
    del dec
