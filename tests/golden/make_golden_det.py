#!/usr/bin/env python3
"""Records tests/golden/det_ref_golden.json: the text output of the REFERENCE's lattice determinization (oracle/_ref/bin/
ref-lattice-determinize = /root/reference/src/lat/determinize-lattice-pruned.cc compiled unmodified against third_party/minifst,
called the way lattice-determinize-pruned / lattice-determinize-phone-pruned call it) on the lattices of tests/test_lattice_det.py's
REF_CASES.  Needs /root/reference (run oracle/build_ref.sh first); the JSON travels to machines that have neither."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tests import test_lattice_det as t
out = {}
with tempfile.TemporaryDirectory() as td:
    mdl = t.write_model(td)
    for name in t.REF_CASES:
        out[name] = t.run_reference(name, td, mdl)
        print(name, len(out[name]), "bytes")
    out["convert_lattice"] = t.run_reference_convert(td)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "det_ref_golden.json"), "w"), indent=0, sort_keys=True)
