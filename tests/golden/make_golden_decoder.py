#!/usr/bin/env python3
"""Records tests/golden/decoder_ref_golden.json: for every case of tests/decoder_cases.py the size and the canonical digest
(tests/lattice_sig.py) of the raw lattice produced by the REFERENCE's own LatticeFasterDecoder (oracle/_ref/bin/ref-lattice-decoder =
/root/reference/src/decoder/lattice-faster-decoder.cc compiled unmodified against third_party/minifst).  Needs /root/reference
(run oracle/build_ref.sh first); the JSON travels to machines that have neither."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import lattice_oracle as lo, ref_decoder as rd
from tests import decoder_cases as dc, lattice_sig as ls
out = {}
for name in dc.CASES:
    f, t2p, ll, kw = dc.make(name)
    r = rd.decode(f, ll, t2p, lo.Config(**kw))
    out[name] = dict(states=int(r["frame"].size), arcs=int(r["src"].size), reached_final=r["reached_final"], digest=ls.digest(ls.canonical_of_reference(r)))
    print(name, out[name])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "decoder_ref_golden.json"), "w"), indent=1, sort_keys=True)
