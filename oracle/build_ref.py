#!/usr/bin/env python3
"""Build recipe of ``oracle/_ref``: the REAL reference package (r9y9/wavenet_vocoder), compiled where it lies.

TEST INFRASTRUCTURE -- never part of the product.  ``oracle/_ref/`` is the checker the parity tests and ``bench.py``'s
``cpu_baseline`` leg compare with / time; nothing under ``wavenet_vocoder_amd/`` may import it
(``tests/test_host_cpu.py::test_product_never_imports_the_oracle``).

The reference's hot path is six pure-Python modules (``wavenet_vocoder/{wavenet,modules,conv,mixture,upsample,util}.py`` + the package's
``__init__`` / ``version``).  This recipe byte-compiles them FROM ``/root/reference`` -- unmodified, by CPython's own compiler -- into
sourceless ``oracle/_ref/wavenet_vocoder/*.pyc``.  Outputs only, binaries only: no reference source text enters the repository, and
``oracle/_ref/`` is git-ignored (it is NOT gpurun-ignored, so it travels to the GPU box with the tree exactly as the built ``.so`` does).
A manifest records the sha256 of every source file that was compiled and the interpreter's bytecode magic.

    python oracle/build_ref.py            (called from __graft_entry__.build(); a no-op when /root/reference is absent and a
                                           previously built oracle/_ref is still importable, e.g. on the GPU box)

Loader: ``oracle/reference.py::load_reference()``.
"""
import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
PKG = "wavenet_vocoder"
MODULES = ("__init__", "version", "wavenet", "modules", "conv", "mixture", "upsample", "util")
REFERENCE_ROOT = os.environ.get("WNV_REFERENCE_ROOT", "/root/reference")


def build(verbose=True):
    src_dir = os.path.join(REFERENCE_ROOT, PKG)
    manifest_path = os.path.join(OUT, "MANIFEST.json")
    if not os.path.isdir(src_dir):
        if os.path.exists(manifest_path):
            if verbose:
                print(f"[oracle/_ref] {REFERENCE_ROOT} absent; keeping the prebuilt {OUT}")
            return OUT
        raise FileNotFoundError(f"{src_dir} not found and no prebuilt oracle/_ref: build it where the reference tree is present")
    dst_dir = os.path.join(OUT, PKG)
    tmp = dst_dir + ".tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    manifest = {"reference_root": REFERENCE_ROOT, "python": sys.version.split()[0],
                "bytecode_magic": importlib.util.MAGIC_NUMBER.hex(), "modules": {}}
    for mod in MODULES:
        src = os.path.join(src_dir, mod + ".py")
        # dfile: what tracebacks show -- the path inside the reference tree (the source itself does not travel)
        py_compile.compile(src, cfile=os.path.join(tmp, mod + ".pyc"), dfile=f"<reference>/{PKG}/{mod}.py", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest["modules"][mod] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    shutil.rmtree(dst_dir, ignore_errors=True)
    os.replace(tmp, dst_dir)
    with open(manifest_path, "w") as f:
        json.dump(manifest, f, indent=1)
    if verbose:
        print(f"[oracle/_ref] compiled {len(MODULES)} modules of {src_dir} -> {dst_dir} (bytecode only)")
    return OUT


if __name__ == "__main__":
    build()
