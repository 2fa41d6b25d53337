import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kaldi_amd import chain, synth
P, B, T = 4000, 64, 150; rng = np.random.default_rng(5)
fsts = [synth.make_e2e_fst(T, P, seed=700 + i, num_phones=int(rng.integers(20, 60))) for i in range(B)]
out = torch.from_numpy((rng.standard_normal((T * B, P)) * 2.0).astype(np.float32)).cuda()
sup = chain.Supervision(fsts, T, P, weight=1.0, e2e=True); num = chain.NumeratorComputation(sup, out); d = torch.zeros_like(out)
print("states per FST", np.mean([f.num_states for f in fsts]), "arcs", np.mean([int(f.arc_offsets[-1]) for f in fsts]))
for name, fn in (("forward", lambda: num.Forward()), ("forward + backward", lambda: num.Backward(d))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, "%.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
