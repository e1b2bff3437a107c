import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build the native pieces if this checkout has not been built yet (hipcc cross-compiles here)
    import subprocess
    lib = os.path.join(ROOT, "linearmpc.jl_amd", "lib", "liblmpc_hip.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-j6", "-C", os.path.join(ROOT, "linearmpc.jl_amd", "csrc")], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def oracle_ldp_from(pk):
    """Wrap a pack dict (golden fixture or BatchedQP.ldp()) as the oracle's LDP struct."""
    from oracle import ldp as oldp
    M = np.asarray(pk["M"], float)
    m, n = M.shape
    nth = int(pk["nth"]) if "nth" in pk else (np.asarray(pk["Dth"]).size // m if m else np.asarray(pk["Xth"]).shape[-1])
    Dth = np.asarray(pk["Dth"], float).reshape(m, nth)
    Rout = np.asarray(pk["Rout"], float).reshape(-1, n)
    sense = np.asarray(pk["sense"] if "sense" in pk else pk["senses"], np.int32)
    ms = int(pk["ms"]) if "ms" in pk else 0
    return oldp.LDP(n, m, ms, nth, Rout.shape[0], M, np.asarray(pk["du"], float),
                    np.asarray(pk["dl"], float), Dth, Rout, np.asarray(pk["x0"], float),
                    np.asarray(pk["Xth"], float).reshape(Rout.shape[0], -1), sense, np.ones(m)).contiguous()


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
