"""Build recipe of the HIP engine: one ``hipcc`` invocation, gfx950 only, output in-tree
(``wavenet_vocoder_amd/libwnv_hip.so``) so that it travels with the source tree to the GPU box.

    python -m wavenet_vocoder_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libwnv_hip.so")
SOURCES = ["wnv_host.cpp", "wnv_layers.cpp", "wnv_generic.hip", "wnv_upsample.hip", "wnv_ring.hip", "wnv_wide.hip", "wnv_post.hip", "wnv_forward.hip", "wnv_mel.hip", "wnv_ubench.hip"]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc and PATH)")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "wnv.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, out: str = OUT, extra_flags=()) -> str:
    """Compile every HIP translation unit for gfx950 and link the C-ABI shared library (``out`` / ``extra_flags``: debug or
    trace builds next to the product library, selected at run time with WNV_LIB)."""
    if out == OUT and not force and not _stale():
        return OUT
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", *extra_flags, "-o", out + ".tmp", "-x", "hip"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[wnv build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else OUT
    flags = argv[argv.index("--flags") + 1].split() if "--flags" in argv else ()
    print(build(force="--force" in argv, out=os.path.abspath(out), extra_flags=flags))
