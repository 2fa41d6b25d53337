"""debug: batched-wav-nnet3-cuda2 --ivector-extraction-config vs the chain of separate programs"""
import os, subprocess, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import synth
from oracle import kaldi_io as kio
BIN = os.path.join(ROOT, "kaldi_amd", "bin"); IV = os.path.join(ROOT, "tests", "golden", "ivector")
td = tempfile.mkdtemp(); N = 120; lens = [16000, 9000, 23001, 4000]
lines = []
for i, n in enumerate(lens): kio.write_wav(f"{td}/u{i}.wav", synth.gaussian_pcm16(n, 50 + i)); lines.append(f"utt{i} {td}/u{i}.wav")
open(f"{td}/wav.scp", "w").write("\n".join(lines) + "\n"); open(f"{td}/fbank.conf", "w").write("--num-mel-bins=40\n--dither=0\n")
run = lambda c: subprocess.run(c, capture_output=True, text=True)
r = run([os.path.join(BIN, "compute-fbank-feats-cuda"), f"--config={td}/fbank.conf", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"]); assert r.returncode == 0, r.stderr
feats = kio.read_ark(f"{td}/f.ark"); allf = np.concatenate(list(feats.values())).astype(np.float64); rng = np.random.default_rng(8)
tm = lambda path, m: open(path, "w").write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m) + " ]\n")
st = np.zeros((2, 41)); st[0, :40] = allf.sum(0); st[1, :40] = (allf ** 2).sum(0); st[0, 40] = allf.shape[0]; tm(f"{td}/global_cmvn.stats", st)
tm(f"{td}/final.mat", (rng.standard_normal((20, 7 * 40)) * 1.5 / np.sqrt(7 * 40)).astype(np.float32))
open(f"{td}/splice.conf", "w").write("--left-context=3\n--right-context=3\n"); open(f"{td}/cmvn.conf", "w").write("\n")
open(f"{td}/ivector.conf", "w").write(f"--lda-matrix={td}/final.mat\n--global-cmvn-stats={td}/global_cmvn.stats\n--cmvn-config={td}/cmvn.conf\n--splice-config={td}/splice.conf\n--diag-ubm={IV}/final.dubm\n"
                                      f"--ivector-extractor={IV}/final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n--max-count=100\n--ivector-period=10\n")
net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=N, calib_feats=feats["utt0"], out_std=1.5, ivector_dim=16)
net.write(f"{td}/final.mdl", as_mdl=True, num_pdfs=N)
graph = synth.make_hclg(3000, 8000, N, seed=11, start_degree=50); graph.write_openfst(f"{td}/HCLG.fst")
common = ["--frame-subsampling-factor=3", "--acoustic-scale=1.0", "--beam=15.0", "--lattice-beam=8.0", "--max-active=10000", "--frames-per-chunk=51", "--determinize-lattice=false"]
for mb in (3, 4):
    for lit in ("true", "false"):
        r = run([os.path.join(BIN, "batched-wav-nnet3-cuda2"), "--feature-type=fbank", f"--fbank-config={td}/fbank.conf", f"--ivector-extraction-config={td}/ivector.conf", f"--max-batch-size={mb}", "--write-compact=false", f"--literal-order={lit}"] + common +
                [f"{td}/final.mdl", f"{td}/HCLG.fst", f"scp:{td}/wav.scp", f"ark,t:{td}/lat_{mb}_{lit}.txt"]); assert r.returncode == 0, r.stderr
open(f"{td}/spk2utt", "w").write("".join(f"{k} {k}\n" for k in feats))
r = run([os.path.join(BIN, "ivector-extract-online2"), f"--config={td}/ivector.conf", f"ark:{td}/spk2utt", f"ark:{td}/f.ark", f"ark:{td}/iv.ark"]); assert r.returncode == 0, r.stderr
for lit in ("true", "false"):
    r = run([os.path.join(BIN, "nnet3-latgen-faster")] + common + [f"--literal-order={lit}", f"--online-ivectors=ark:{td}/iv.ark", "--online-ivector-period=10", f"{td}/final.mdl", f"{td}/HCLG.fst", f"ark:{td}/f.ark", f"ark,t:{td}/lat2_{lit}.txt"]); assert r.returncode == 0, r.stderr
def summary(p):
    out = {}; cur = None
    for l in open(p):
        t = l.split()
        if len(t) == 1: cur = t[0]; out[cur] = [0, 0.0]
        elif len(t) >= 4: out[cur][0] += 1; c = t[4].split(",") if len(t) > 4 else ["0", "0"]; out[cur][1] += float(c[1]) if len(c) > 1 and c[1] else 0.0
    return out
for f in sorted(os.listdir(td)):
    if f.startswith("lat") and f.endswith(".txt"): print(f, summary(f"{td}/{f}"))
# loglikes: CLI nnet3-compute with the CLI's i-vectors vs the python path (features -> extractor -> network) the batched program mirrors
r = run([os.path.join(BIN, "nnet3-compute"), "--frame-subsampling-factor=3", "--frames-per-chunk=51", f"--online-ivectors=ark:{td}/iv.ark", "--online-ivector-period=10", f"{td}/final.mdl", f"ark:{td}/f.ark", f"ark:{td}/ll.ark"]); assert r.returncode == 0, r.stderr
ll = kio.read_ark(f"{td}/ll.ark")
from kaldi_amd import nnet3
from kaldi_amd.ivector import OnlineIvectorExtractionInfo, BatchedIvectorExtractor
dev = torch.device("cuda:0"); keys = list(feats)
ex = BatchedIvectorExtractor(OnlineIvectorExtractionInfo(f"{td}/ivector.conf"))
x = torch.from_numpy(np.concatenate([feats[k] for k in keys])).to(dev); fo = np.concatenate([[0], np.cumsum([feats[k].shape[0] for k in keys])])
ivs, ro = ex.GetIvectors(x, fo); civ = kio.read_ark(f"{td}/iv.ark")
print("ivector cli vs python", max(np.abs(ivs.cpu().numpy()[ro[i]:ro[i + 1]] - civ[k]).max() for i, k in enumerate(keys)))
nn = nnet3.Nnet(f"{td}/final.mdl"); nb = nnet3.NnetBatch(nn, [feats[k].shape[0] for k in keys], 3, ivector_rows=[int(ro[i + 1] - ro[i]) for i in range(4)], online_ivector_period=10, frames_per_chunk=51)
y = nb.forward(x, ivectors=ivs).cpu().numpy()
print("loglikes cli vs python", max(np.abs(y[nb.out_offsets[i]:nb.out_offsets[i + 1]] - ll[k]).max() for i, k in enumerate(keys)))
from oracle import lattice_oracle as lo
t2p = synth.tid2pdf(N)
for k in keys:
    for mode in (0, 1):
        ref, info = lo.decode(graph, ll[k], t2p, lo.Config(beam=15.0, lattice_beam=8.0, max_active=10000), mode=mode)
        c = ref.connect(); print(k, "oracle mode", mode, "arcs", c.num_arcs, "ac sum", float(c.arc_ac.astype(np.float64).sum()), {x: info[x] for x in info if "order" in x or "final" in x})
