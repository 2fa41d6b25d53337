#!/usr/bin/env python3
"""Generates tests/golden/word_align_golden.json: the output of the REFERENCE's WordAlignLattice + MinimumBayesRisk (oracle/_ref/bin/ref-word-align: lat/word-align-lattice.cc and
lat/sausages.cc compiled unmodified over third_party/minifst by oracle/build_ref.sh) on the lattices of tests/test_word_align.py.  Run where /root/reference exists."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tests import test_word_align as t
td = tempfile.mkdtemp(); mdl, wb = t.write_model(td); out = {}
for name in sorted(t.CASES):
    out[name] = t.run_reference(name, td, mdl, wb); print(name, len(out[name]), "bytes")
json.dump(out, open(t.GOLD, "w"), indent=0, sort_keys=True)
