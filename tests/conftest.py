"""pytest configuration: the `gpu` marker, import paths, and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors / compiled reference, host logic, C-ABI exports.
`-m gpu`       : parity of the HIP path (through the C-ABI) against the oracle.
"""
import os
import subprocess
import sys

import pytest

os.environ.setdefault("OMP_NUM_THREADS", "1")  # the reference's render() is only deterministic single-threaded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "raytracer.c_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 1666943821  # reference main.c:182


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip.so"),
            os.path.join(ROOT, "raytracer.c_amd", "host", "libraytracer_amd.so"),
            os.path.join(ROOT, "oracle", "libpt_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-C", ROOT, "all"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


@pytest.fixture(scope="session")
def pt():
    import oracle_py
    return oracle_py.PtOracle()


@pytest.fixture(scope="session")
def ref():
    """factory: depth -> RefOracle, or skip when the compiled reference is not present"""
    import oracle_py
    cache = {}

    def get(depth):
        if not oracle_py.ref_available(depth):
            pytest.skip(f"oracle/_ref/libref_oracle_d{depth}.so not built (needs /root/reference)")
        if depth not in cache:
            cache[depth] = oracle_py.RefOracle(depth)
        return cache[depth]
    return get


@pytest.fixture(scope="session")
def ref_mesh():
    """factory: depth -> RefMeshOracle (the reference's compiled code with its commented-out mesh
    scan revived, oracle/ref_harness.c ORACLE_MESH_HOOK), or skip when not built"""
    import oracle_py
    cache = {}

    def get(depth):
        if not oracle_py.ref_mesh_available(depth):
            pytest.skip(f"oracle/_ref/libref_mesh_d{depth}.so not built (needs /root/reference)")
        if depth not in cache:
            cache[depth] = oracle_py.RefMeshOracle(depth)
        return cache[depth]
    return get
