"""kaldi_amd -- MI355X (gfx950) native batched acoustic pipeline behind Kaldi's interfaces.

Product code lives in kaldi_amd/csrc (hand-written HIP + the C ABI of include/k3hip.h, built into
kaldi_amd/lib/libk3hip.so) and kaldi_amd/host (C++ adapters with the reference's class names).
The Python modules here are thin ctypes plumbing used by tests/ and bench.py: torch supplies device
memory and streams only.  There is NO CPU fallback: importing kaldi_amd.lib without the built
library raises."""
__version__ = "0.1.0"
