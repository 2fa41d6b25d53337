"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference (kaldi-asr/kaldi) algorithms on the hot path, used solely as the
parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
kaldi_amd/ imports this package; the product path fails loudly when its HIP library is missing.
"""
