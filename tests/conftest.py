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
            os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_dev.so"),    # children of the GPU tests: development switches
            os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_diag.so"),   # ... and the PT_DIAG re-checks
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


# ---- which kernels of the family a GPU test run launched, and under an oracle comparison or not --------------------------
# rt_hip_kernel_launches() counts render launches per family member in this process.  Around every GPU test the counters
# are read; a test during which util.assert_parity passed at least once (a comparison with the CPU oracle or with a golden
# fixture the compiled reference produced) credits the kernels it launched as "oracle-compared".  The last-collected test
# (tests/test_zz_kernel_coverage.py) prints the table and asserts that no shipped kernel went unreached.
KERNEL_COVERAGE = {}   # name -> dict(launches, compared_launches, tests=[...])


def _launch_counts():
    import ctypes as C
    from rt_amd import abi
    shim = abi.load_shim()
    out = {}
    for k in range(shim.rt_hip_kernel_count()):
        n = C.c_uint64(0)
        name = shim.rt_hip_kernel_launches(k, C.byref(n))
        out[name.decode()] = n.value
    return out


@pytest.fixture(autouse=True)
def _kernel_coverage(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import util
    before, parity0 = _launch_counts(), util.PARITY_PASSED[0]
    yield
    after = _launch_counts()
    compared = util.PARITY_PASSED[0] > parity0
    for name, n in after.items():
        d = n - before.get(name, 0)
        rec = KERNEL_COVERAGE.setdefault(name, dict(launches=0, compared_launches=0, tests=[]))
        if d:
            rec["launches"] += d
            if compared:
                rec["compared_launches"] += d
                if len(rec["tests"]) < 3:
                    rec["tests"].append(request.node.name)


def kernel_coverage_lines():
    from rt_amd import abi
    shim = abi.load_shim()
    names = [shim.rt_hip_kernel_launches(k, None).decode() for k in range(shim.rt_hip_kernel_count())]
    lines = ["kernel coverage of this run (%d shipped kernels):" % len(names),
             "  %-42s %9s %9s  %s" % ("kernel", "launches", "compared", "first tests that compared it")]
    unreached = []
    for n in names:
        rec = KERNEL_COVERAGE.get(n, dict(launches=0, compared_launches=0, tests=[]))
        lines.append("  %-42s %9d %9d  %s" % (n, rec["launches"], rec["compared_launches"], ", ".join(rec["tests"])))
        if rec["compared_launches"] == 0:
            unreached.append(n)
    lines.append("  unreached: %d%s" % (len(unreached), (" -- " + ", ".join(unreached)) if unreached else ""))
    return lines, unreached


def pytest_terminal_summary(terminalreporter):
    """the coverage table in the test log itself, whatever the capture mode (and in gpurun_out/, when that exists)"""
    if not KERNEL_COVERAGE:
        return
    lines, _ = kernel_coverage_lines()
    terminalreporter.write_line("")
    for ln in lines:
        terminalreporter.write_line(ln)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "kernel_coverage.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
