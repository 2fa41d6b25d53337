import os, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")

@pytest.fixture(scope="session")
def feat_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "feat_golden.npz"))

@pytest.fixture(scope="session")
def cmvn_online_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "cmvn_online_golden.npz"))
