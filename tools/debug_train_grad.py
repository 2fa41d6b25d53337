import numpy as np, struct, subprocess, os, tempfile, sys
sys.path.insert(0, '/root/repo')
from kaldi_amd import synth
ROOT='/root/repo'; td = tempfile.mkdtemp()
calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=50, calib_feats=calib, out_std=1.5); net.write(f"{td}/m.raw")
def wm(path, m): m = np.ascontiguousarray(m, '<f4'); open(path,'wb').write(b'\0BFM ' + b'\x04' + struct.pack('<i', m.shape[0]) + b'\x04' + struct.pack('<i', m.shape[1]) + m.tobytes())
B, T, s, lc, rc = 4, 10, 3, 8, 8; Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(0)
wm(f"{td}/in.mat", (rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5)); wm(f"{td}/od.mat", rng.standard_normal((T * B, 50)) * 0.1)
r = subprocess.run([f"{ROOT}/kaldi_amd/adapter/_build/nnet3-train-grad", f"{td}/m.raw", str(B), str(T), str(s), f"{td}/in.mat", f"{td}/od.mat", f"{td}/out.mat", f"{td}/grad.vec"], capture_output=True, text=True, env=dict(os.environ, K3_ADAPTER_LIST_MISSING="1", MKL_THREADING_LAYER="SEQUENTIAL"))
print(r.returncode); print("\n".join(sorted(set(l for l in r.stderr.splitlines() if "MISSING" in l)))); print(r.stderr[-800:])
