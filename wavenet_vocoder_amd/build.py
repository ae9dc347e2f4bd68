"""Build recipe of the HIP engine: ``hipcc`` per translation unit (in parallel), gfx950 only, outputs in-tree so that they travel with
the source tree to the GPU box:

  * ``wavenet_vocoder_amd/libwnv_hip.so``  -- the PRODUCT library: the C ABI of include/wnv.h and nothing else; reads no environment
    variable (csrc/wnv_knobs.h);
  * ``wavenet_vocoder_amd/libwnv_test.so`` -- the same sources with ``-DWNV_KNOBS -DWNV_TEST_HOOKS`` plus csrc/wnv_ubench.hip: the
    measurement knobs of the experiment scripts / variant tests and the two hooks of include/wnv_test.h (time-out injection, the LDS
    read-peak microbenchmark bench.py quotes).  Selected per process with ``WNV_LIB=<path>``.

    python -m wavenet_vocoder_amd.build [--force] [--out <lib> --flags "<extra flags>"]     (--out: one more variant, e.g. a trace build)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libwnv_hip.so")
TEST_OUT = os.path.join(HERE, "libwnv_test.so")
SOURCES = ["wnv_host.cpp", "wnv_layers.cpp", "wnv_generic.hip", "wnv_upsample.hip", "wnv_ring.hip", "wnv_wide.hip", "wnv_post.hip",
           "wnv_forward.hip", "wnv_mel.hip"]
TEST_ONLY_SOURCES = ["wnv_ubench.hip"]
# translation units whose code depends on the knob / hook macros (the others are shared between the two libraries)
KNOB_SOURCES = {"wnv_host.cpp", "wnv_ring.hip", "wnv_wide.hip"}
TEST_FLAGS = ("-DWNV_KNOBS", "-DWNV_TEST_HOOKS")
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc and PATH)")


def _deps_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(HERE, "..", "include", h) for h in ("wnv.h", "wnv_test.h")]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _stale(path: str) -> bool:
    return not os.path.exists(path) or os.path.getmtime(path) < _deps_mtime()


def _compile(src: str, tag: str, flags, verbose: bool) -> str:
    """One translation unit -> build/<tag>/<src>.o (recompiled when any source or header is newer: the headers are shared)."""
    os.makedirs(os.path.join(OBJ, tag), exist_ok=True)
    obj = os.path.join(OBJ, tag, src + ".o")
    if not _stale(obj):
        return obj
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden", *flags,
           "-c", "-x", "hip", os.path.join(CSRC, src), "-o", obj + ".tmp"]
    if verbose:
        print("[wnv build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp", obj)
    return obj


def _link(objs, out: str, verbose: bool) -> str:
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'wnv_exports.map')}",
           "-o", out + ".tmp", *objs]
    if verbose:
        print("[wnv build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


def build(force: bool = False, verbose: bool = True, out: str = OUT, extra_flags=()) -> str:
    """Compile every HIP translation unit for gfx950 and link the C-ABI shared libraries.  Default (``out`` = the product library):
    builds the product library AND the test library.  Another ``out`` with ``extra_flags``: one more variant of the product sources
    (debug / trace builds, selected at run time with WNV_LIB)."""
    if force:
        shutil.rmtree(OBJ, ignore_errors=True)
    jobs = int(os.environ.get("WNV_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    if out != OUT:                                   # a variant: all units with the extra flags, its own object directory
        tag = "variant_" + os.path.basename(out).replace(".", "_")
        if extra_flags:
            shutil.rmtree(os.path.join(OBJ, tag), ignore_errors=True)
        srcs = SOURCES + (TEST_ONLY_SOURCES if "-DWNV_TEST_HOOKS" in extra_flags else [])
        with ThreadPoolExecutor(jobs) as ex:
            objs = list(ex.map(lambda s: _compile(s, tag, tuple(extra_flags), verbose), srcs))
        return _link(objs, out, verbose)
    if not force and not _stale(OUT) and not _stale(TEST_OUT):
        return OUT
    work = [(s, "product", ()) for s in SOURCES]
    work += [(s, "test", TEST_FLAGS) for s in SOURCES if s in KNOB_SOURCES] + [(s, "test", TEST_FLAGS) for s in TEST_ONLY_SOURCES]
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda w: _compile(w[0], w[1], w[2], verbose), work))
    by = {(w[0], w[1]): o for w, o in zip(work, objs)}
    _link([by[(s, "product")] for s in SOURCES], OUT, verbose)
    _link([by[(s, "test" if s in KNOB_SOURCES else "product")] for s in SOURCES] + [by[(s, "test")] for s in TEST_ONLY_SOURCES],
          TEST_OUT, verbose)
    return OUT


if __name__ == "__main__":
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else OUT
    flags = argv[argv.index("--flags") + 1].split() if "--flags" in argv else ()
    print(build(force="--force" in argv, out=os.path.abspath(out), extra_flags=flags))
