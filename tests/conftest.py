import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


# ---- limb helpers shared by the tests (canonical little-endian uint64 limbs, the C-ABI layout)
def L(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def I(limbs):
    v = 0
    for i, x in enumerate(np.asarray(limbs).reshape(-1)):
        v |= int(x) << (64 * i)
    return v


def g1_limbs(pt):
    import bls12_381 as o
    return np.array(o.g1_to_limbs(pt), dtype=np.uint64)


def g2_limbs(pt):
    import bls12_381 as o
    return np.array(o.g2_to_limbs(pt), dtype=np.uint64)


def dec1(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def dec2(p):
    return None if p is None else ((int(p[0][0], 16), int(p[0][1], 16)), (int(p[1][0], 16), int(p[1][1], 16)))


def fr_array(vals):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_ints(arr):
    return [I(row) for row in np.asarray(arr).reshape(-1, 4)]


def fr_ints_fast(arr):
    """[n,4] uint64 -> list of python ints (vectorised through object arrays; for 10^6-size checks)."""
    a = np.asarray(arr).reshape(-1, 4)
    v = a[:, 3].astype(object)
    for k in (2, 1, 0):
        v = (v << 64) | a[:, k].astype(object)
    return v.tolist()


def rand_fr_array(n, seed):
    """n uniform-ish Fr values as [n,4] uint64 (numpy RNG; top bit cleared then reduced by the oracle when needed)."""
    import bls12_381 as o
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)      # < 2^254 < r: canonical without a reduction
    return a


@pytest.fixture(scope="session")
def cref():
    import cref as c
    c.build()
    c.lib()
    return c


@pytest.fixture(scope="session")
def ctx():
    import vote_saver_protocol_amd as v
    c = v.Context(0)
    yield c
    c.close()
