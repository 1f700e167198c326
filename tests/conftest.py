"""pytest configuration: the `gpu` marker and a session-wide check that the in-tree native
libraries exist (they are built by `make all` / __graft_entry__.build(); the built .so files travel
to the GPU box with the tree, so nothing is compiled inside a test on that box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NATIVE = [
    os.path.join(ROOT, "raytracer_amd", "_lib", "librt_amd.so"),
    os.path.join(ROOT, "raytracer_amd", "_lib", "librt_scenes.so"),
    os.path.join(ROOT, "oracle", "_build", "liboracle.so"),
    os.path.join(ROOT, "oracle", "_build", "oracle_kat"),
    os.path.join(ROOT, "tests", "cpp", "_build", "libhostsim.so"),
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_libraries():
    missing = [p for p in NATIVE if not os.path.exists(p)]
    if missing:
        # CPU container only (hipcc cross-compiles gfx950 without a GPU); never triggers on the GPU box
        # because the built libraries are part of the tree that is shipped there.
        subprocess.run(["make", "-C", ROOT, "all"], check=True, stdout=subprocess.DEVNULL)
    still = [p for p in NATIVE if not os.path.exists(p)]
    assert not still, f"native libraries missing after build: {still}"


def has_gpu() -> bool:
    try:
        import ctypes as C

        from raytracer_amd import lib

        n = C.c_int(0)
        return lib().rt_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False
