"""Tests that need the TEST library (wavenet_vocoder_amd/libwnv_test.so: the sources built with -DWNV_KNOBS -DWNV_TEST_HOOKS).

The product library reads no environment variable and exports no test hook (csrc/wnv_knobs.h, include/wnv_test.h), and a process
binds ONE library at import (WNV_LIB).  A test decorated with ``needs_test_lib`` therefore re-runs ITSELF -- the same pytest node --
in a child process that loads the test library, and passes when the child does; inside the child the decorator is a no-op."""
import functools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(ROOT, "wavenet_vocoder_amd", "libwnv_test.so")


def in_test_lib_process() -> bool:
    return os.path.abspath(os.environ.get("WNV_LIB", "")) == TEST_LIB


def needs_test_lib(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if in_test_lib_process():
            return fn(*args, **kwargs)
        assert os.path.exists(TEST_LIB), f"{TEST_LIB} is missing: run __graft_entry__.build()"
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]           # "tests/x.py::test_y[param] (call)"
        env = dict(os.environ, WNV_LIB=TEST_LIB)
        env.pop("PYTEST_CURRENT_TEST", None)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", node], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"the test-library child failed (rc {r.returncode})\n{r.stdout[-4000:]}\n{r.stderr[-2000:]}"
    return wrapper
