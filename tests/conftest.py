import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def load_golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def ulp_distance_half(a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
    """Distance in units of the last place between two fp16/bf16 bit-pattern arrays
    (monotone integer mapping of sign-magnitude bit patterns; +0 and -0 are 0 apart)."""
    def key(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))


def ulp_distance_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    def key(x):
        u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.int64)
        return np.where(u & 0x80000000, -(u & 0x7FFFFFFF), u & 0x7FFFFFFF)
    return np.abs(key(a) - key(b))
