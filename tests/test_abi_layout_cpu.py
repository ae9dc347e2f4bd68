"""The C side of the boundary, checked with a C compiler (no GPU):

* `include/wnv.h` is plain C: it compiles as C99 with -pedantic -Werror and as C++17, with nothing but <stdint.h> in scope;
* every struct the header declares has, in the ctypes mirror of `wavenet_vocoder_amd/_lib.py`, the same size and the same
  offset for every field -- the Python host fills these structs byte for byte, so a field added on one side only would
  shift everything behind it silently;
* a C client linked against `libwnv_hip.so` (the binding a maintainer of a C / cgo / JNI host would write)
  reaches the pure-host entry points: ABI version, receptive field (wavenet.py:42-60 known answers), noise width,
  upsampled length, and the error path of wnv_create (bad ABI version -> WNV_ERR_INVALID_ARG with a message).
"""
import ctypes
import os
import shutil
import subprocess

import pytest

from wavenet_vocoder_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")

STRUCTS = {
    "wnv_config": _lib.Config, "wnv_tensor": _lib.Tensor, "wnv_generate_args": _lib.GenerateArgs,
    "wnv_glu_config": _lib.GluConfig, "wnv_forward_args": _lib.ForwardArgs, "wnv_post_args": _lib.PostArgs,
    "wnv_mel_config": _lib.MelConfig, "wnv_logmel_args": _lib.LogmelArgs,
}

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}"
    return r.stdout


def test_header_declares_exactly_the_mirrored_structs():
    import re
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, "wnv.h")).read(), flags=re.S)
    declared = set(re.findall(r"typedef\s+struct\s+(wnv_[a-z_]+)\s*\{", src))
    assert declared == set(STRUCTS), declared ^ set(STRUCTS)


def test_header_is_plain_c99_and_cxx17(tmp_path):
    c = tmp_path / "only_header.c"
    c.write_text('#include "wnv.h"\nint main(void) { return sizeof(wnv_config) > 0 ? 0 : 1; }\n')
    run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c", str(c), "-o", str(tmp_path / "c.o")])
    if shutil.which("g++"):
        run(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-x", "c++", "-I", INC, "-c", str(c),
             "-o", str(tmp_path / "cxx.o")])


def test_ctypes_mirror_has_the_layout_of_the_c_structs(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wnv.h"', "int main(void) {"]
    for cname, mirror in STRUCTS.items():
        lines.append(f'  printf("{cname} . %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            # sizeof of a member through a null pointer expression: an unevaluated operand, valid C99
            lines.append(f'  printf("{cname} {fname} %zu %zu\\n", offsetof({cname}, {fname}), sizeof((({cname}*)0)->{fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, str(src), "-o", str(exe)])
    seen = {}
    for ln in run([str(exe)]).splitlines():
        parts = ln.split()
        seen[(parts[0], parts[1])] = tuple(int(v) for v in parts[2:])
    n_fields = 0
    for cname, mirror in STRUCTS.items():
        assert seen[(cname, ".")] == (ctypes.sizeof(mirror),), (cname, seen[(cname, ".")], ctypes.sizeof(mirror))
        for fname, ftype in mirror._fields_:
            desc = getattr(mirror, fname)
            assert seen[(cname, fname)] == (desc.offset, ctypes.sizeof(ftype)), (cname, fname, seen[(cname, fname)], desc.offset)
            n_fields += 1
    assert n_fields == sum(len(m._fields_) for m in STRUCTS.values()) >= 80
    # and the header has no field the mirror lacks: equal sizes with equal offsets of the LAST field leave no room for one
    # behind it; one in the middle would have shifted an offset above


C_CLIENT = r"""
#include <stdio.h>
#include <string.h>
#include "wnv.h"
#define CHECK(c) do { if (!(c)) { printf("FAILED: %s\n", #c); return 1; } } while (0)
int main(void) {
    wnv_config cfg;
    wnv_handle h = 0;
    CHECK(wnv_abi_version() == WNV_ABI_VERSION);
    CHECK(wnv_receptive_field(30, 3, 3) == 6139);      /* wavenet.py:42-60 known answers (tests/test_host_cpu.py) */
    CHECK(wnv_receptive_field(24, 4, 3) == 505);
    CHECK(wnv_receptive_field(5, 2, 3) == -1);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = WNV_ABI_VERSION;
    cfg.out_channels = 30; cfg.layers = 24; cfg.stacks = 4; cfg.residual_channels = 128; cfg.gate_channels = 256;
    cfg.skip_out_channels = 128; cfg.kernel_size = 3; cfg.cin_channels = 80; cfg.gin_channels = -1;
    cfg.scalar_input = 1; cfg.output_distribution = 1; cfg.upsample_kind = WNV_UPSAMPLE_CONVIN;
    cfg.n_upsample_scales = 4; cfg.upsample_scales[0] = cfg.upsample_scales[1] = cfg.upsample_scales[2] = cfg.upsample_scales[3] = 4;
    cfg.freq_axis_kernel_size = 3; cfg.cin_pad = 2;
    CHECK(wnv_noise_width(&cfg) == 11);                /* 10 Gumbel uniforms + 1 logistic uniform per step (mixture.py:138-152) */
    CHECK(wnv_upsampled_length(&cfg, 94 + 4) == 94 * 256);
    CHECK(wnv_upsampled_length(&cfg, 3) == -1);
    /* a host-only handle: checkpoint packing without a device (there is no CPU compute path behind it) */
    CHECK(wnv_create(&cfg, -1, &h) == WNV_OK);
    CHECK(h != 0);
    CHECK(wnv_macs_per_sample(h) == -1);               /* no weights yet */
    {
        float dummy[4] = {0};
        CHECK(wnv_upsample(h, dummy, 1, 8, dummy, -1, 0) == WNV_ERR_INVALID_ARG);   /* refused: no CPU path */
        CHECK(strstr(wnv_last_error(), "host-only") != 0);
    }
    CHECK(wnv_destroy(h) == WNV_OK);
    /* error behaviour: wrong ABI version */
    cfg.abi_version = WNV_ABI_VERSION + 1;
    h = 0;
    CHECK(wnv_create(&cfg, -1, &h) == WNV_ERR_INVALID_ARG);
    CHECK(h == 0);
    CHECK(strlen(wnv_last_error()) > 0);
    printf("c client ok: abi %d\n", (int)wnv_abi_version());
    return 0;
}
"""


def test_a_c_client_links_against_the_library_and_calls_the_host_entry_points(tmp_path):
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    src = tmp_path / "client.c"
    src.write_text(C_CLIENT)
    exe = tmp_path / "client"
    libdir = os.path.dirname(_lib.LIB_PATH)
    # the product library is named libwnv_hip.so: -l:name links that exact file
    run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, str(src), "-o", str(exe), "-L", libdir,
         "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = run([str(exe)])
    assert "c client ok" in out, out
