"""CPU-side checks of the boundary: libk3hip.so loads and exports every symbol include/k3hip.h declares
(no compute calls without a GPU)."""
import ctypes, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _declared():
    src = open(os.path.join(ROOT, "include", "k3hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(k3_[a-z0-9_]+)\s*\(", src)))

def test_library_exports_all_declared_symbols():
    import __graft_entry__ as ge
    ge.build()
    from kaldi_amd import lib
    L = ctypes.CDLL(lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 8
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing

def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from kaldi_amd import lib
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(lib, "_lib", None)
    import pytest
    with pytest.raises(lib.K3Error):
        lib.load()


def _prototypes(header, prefix):
    """{name: [argument declarations]} of every function declared in an include/*.h"""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S); src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(%s[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;" % prefix, src, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        out[m.group(1)] = [] if args == ["void"] or args == [""] else args
    return out

def test_ctypes_declarations_agree_with_the_headers():
    """every argtypes list in kaldi_amd/lib.py and kaldi_amd/hostlib.py has as many entries as the C prototype has parameters, pointer
    parameters are declared as pointers / c_void_p / c_char_p and 64-bit integers as 64-bit: a mismatch here would only show up as a
    crash on the GPU box"""
    from kaldi_amd import lib, hostlib
    import ctypes as C
    for header, prefix, L in (("k3hip.h", "k3_", lib.load()), ("k3host.h", "k3h_", hostlib.load())):
        protos = _prototypes(header, prefix); assert len(protos) >= 10
        checked = 0
        for name, params in protos.items():
            f = getattr(L, name)
            if f.argtypes is None: continue                   # not used from Python
            assert len(f.argtypes) == len(params), (name, len(f.argtypes), params)
            for t, p in zip(f.argtypes, params):
                is_ptr = "*" in p
                py_ptr = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) == "P"
                assert is_ptr == py_ptr, (name, p, t)
                if not is_ptr:
                    if re.search(r"\bint64_t\b", p): assert C.sizeof(t) == 8, (name, p, t)
                    elif re.search(r"\b(int32_t|int)\b", p): assert C.sizeof(t) == 4, (name, p, t)
                    elif re.search(r"\bdouble\b", p): assert t is C.c_double, (name, p, t)
                    elif re.search(r"\bfloat\b", p): assert t is C.c_float, (name, p, t)
            checked += 1
        assert checked >= 8, (header, checked)


def test_struct_layouts_agree_with_the_header():
    """the ctypes mirrors of the POD structs have the fields of include/k3hip.h in the same order with the same C types"""
    from kaldi_amd import lib
    import ctypes as C
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "k3hip.h")).read(), flags=re.S)
    ctype = {"float": C.c_float, "int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double}
    for cname, mirror in (("k3_feat_opts", lib.FeatOpts), ("k3_online_cmvn_opts", lib.OnlineCmvnOpts), ("k3_nnet_info", lib.NnetInfo), ("k3_decoder_config", lib.DecoderConfig)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl: continue
            t, names = decl.split(None, 1)
            fields += [(n.strip(), ctype[t]) for n in names.split(",")]
        assert [(n, t) for n, t in mirror._fields_] == fields, cname
