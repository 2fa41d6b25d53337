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
