"""Drop-in proof at source level: the reference's OWN callers (main.c, test.c) compile against
include/raytracer.h + include/vector.h and link against libraytracer_amd.so, unmodified.
They are compiled from a scratch copy outside the repo (a quote-include resolves to the
including file's directory first, which would be the reference's own headers).  Needs
/root/reference, so these run in the build container only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "raytracer.c_amd", "host")
CSRC = os.path.join(ROOT, "raytracer.c_amd", "csrc")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.c")), reason="reference not mounted")


def _compile(tmp_path, name, extra=()):
    src = tmp_path / name
    shutil.copy(os.path.join(REF, name), src)          # scratch copy in the test's tmp dir, never in the repo
    exe = tmp_path / name.replace(".c", "")
    cmd = ["gcc", "--std=c99", "-D_DEFAULT_SOURCE", "-O3", "-fopenmp", "-Wno-unused-variable",
           f"-I{ROOT}/include", f"-I{REF}", str(src), "-o", str(exe), f"-L{HOST}", "-lraytracer_amd",
           f"-Wl,-rpath,{HOST}", f"-Wl,-rpath-link,{CSRC}", "-lm", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(exe)


def test_reference_test_c_builds_and_behaves_like_the_reference(tmp_path):
    """test.c: cross product check passes; the surface-normal check fails exactly as it does
    against the reference's own raytracer.o, and the binary exits 0"""
    exe = _compile(tmp_path, "test.c")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0
    assert "OK" in r.stdout and "0063" in r.stdout
    assert "FAIL" in r.stderr and "0078" in r.stderr


def test_reference_main_c_builds_against_the_boundary(tmp_path):
    """main.c (the only caller of render(), main.c:429) compiles and links unchanged; without a
    GPU its render() call fails loudly instead of falling back to a CPU path"""
    exe = _compile(tmp_path, "main.c")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage:" in r.stderr          # main.c:189-193
    from rt_amd import abi
    if abi.load_shim().rt_hip_device_count() == 0:
        r = subprocess.run([exe, "-w", "32", "-h", "18", "-s", "1", "-o", str(tmp_path / "o.png")],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "GPU path failed" in r.stderr
