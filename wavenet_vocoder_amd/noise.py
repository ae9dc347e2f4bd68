"""Noise tapes: the random numbers the sampling stage consumes, generated up front.

The reference draws from torch's global generator inside the sample loop
(wavenet_vocoder/mixture.py:138,151,247,266; wavenet.py:334-335).  The HIP engine instead reads a
``(T, B, NZ)`` float32 tape (or runs its own counter-based generator when no tape is given).  This
module produces a tape by replaying, on the host, exactly the draws the reference would make on CPU, in
the same order and from the same generator -- so ``torch.manual_seed(s); model.incremental_forward(...)``
sees the same random numbers the reference's CPU path sees for seed ``s`` (the replay is
re-verified against the real reference by tests/golden/make_golden.py).

Tape layout per (t, b), NZ floats:
  Logistic (MoL)        : u1[0..nr_mix) ~ U(1e-5, 1-1e-5) (Gumbel-max), then u2 ~ U(1e-5, 1-1e-5)
  Normal, C in {2, 3}   : n ~ N(0, 1)
  Normal, C = 3k > 3    : u1[0..nr_mix), then n ~ N(0, 1)
  one-hot (categorical) : e[0..C) ~ Exp(1)            (torch.multinomial draws argmax(p / e))

Threads: the fast replay path advances the generator through ``get_state()`` / a native draw / ``set_state()`` under a module lock, not
under the generator's own mutex (torch's ``uniform_`` holds that for the whole draw).  Calls of this module from several threads are
serialised; a generator that another thread draws from THROUGH TORCH at the same time must not be shared with it (its draws could be
duplicated or lost) -- give the replay its own ``torch.Generator``.
"""
from __future__ import annotations

from typing import Optional

import torch

__all__ = ["noise_width", "make_noise_tape", "exponential_draws"]

_EPS = 1e-5
_BULK_OK: dict = {}


def _bulk_matches_per_step(kind: str) -> bool:
    """One-time probe per draw kind: does ONE bulk call of this torch build walk the generator's stream exactly like a sequence of
    per-step calls?  It does wherever the CPU kernels consume the Mersenne-Twister stream element by element (this ROCm build: yes,
    pinned by tests/test_host_cpu.py) -- a build that routes a distribution through a vectorised library path of its own (MKL VSL)
    need not.  The probe draws two steps both ways from scratch generators; on a mismatch the tape is drawn step by step."""
    ok = _BULK_OK.get(kind)
    if ok is None:
        g1, g2 = torch.Generator().manual_seed(1234), torch.Generator().manual_seed(1234)
        if kind == "exponential":
            a = torch.empty(2, 3, 40).exponential_(1.0, generator=g1)
            b = torch.stack([torch.empty(3, 40).exponential_(1.0, generator=g2) for _ in range(2)])
        elif kind == "uniform":
            a = torch.empty(2, 3 * 11).uniform_(_EPS, 1.0 - _EPS, generator=g1)
            b = torch.stack([torch.cat([torch.empty(3, 1, 10).uniform_(_EPS, 1.0 - _EPS, generator=g2).reshape(-1),
                                        torch.empty(3, 1).uniform_(_EPS, 1.0 - _EPS, generator=g2).reshape(-1)]) for _ in range(2)])
        elif kind == "normal16":
            a = torch.empty(2, 16, 1).normal_(0.0, 1.0, generator=g1)
            b = torch.stack([torch.empty(16, 1).normal_(0.0, 1.0, generator=g2) for _ in range(2)])
        else:                                                      # "normal_strided": batches below 16 through a strided view
            buf = torch.empty(2 * 5, 2)
            buf[:, 0].normal_(0.0, 1.0, generator=g1)
            a = buf[:, 0].reshape(2, 5, 1)
            b = torch.stack([torch.empty(5, 1).normal_(0.0, 1.0, generator=g2) for _ in range(2)])
        ok = _BULK_OK[kind] = bool(torch.equal(a.reshape(-1), b.reshape(-1)))
    return ok


import threading

_NATIVE_DRAW_LOCK = threading.Lock()   # serialises get_state -> native draw -> set_state of the replay path (the generator itself is the caller's)
_U_SCRATCH = threading.local()     # per thread: uniform_ and the ctypes call release the GIL, two callers must not share the staging buffer
_U_CHUNK = 1 << 21                 # values per staging pass: 16 MB of float64, whatever the size of the request


def exponential_draws(out: torch.Tensor, generator: Optional[torch.Generator] = None) -> bool:
    """Fill the contiguous CPU float32 tensor ``out`` with what ``out.exponential_(1.0, generator=generator)`` would put there -- same
    numbers, same generator state afterwards -- several times faster: ATen's CPU kernel draws one 53-bit uniform per element and maps
    it with ``-log1p(-u)`` in double, serially (13-25 ns per value); here the uniforms come from one ``uniform_`` call on a float64
    tensor (the same draws: DistributionsHelper.h) and the transform runs on all cores in the engine's library
    (wnv_exponential_from_uniform, the same libm ``log1p``).  Returns False -- nothing drawn -- when the library is missing or a
    one-time probe finds that this torch build does not produce the same numbers that way (then the caller uses ``exponential_``)."""
    ok = _BULK_OK.get("exp_fast")
    if ok is None:
        try:
            import os
            from . import _lib
            lib = _lib.lib()
            g1, g2 = torch.Generator().manual_seed(4321), torch.Generator().manual_seed(4321)
            a = torch.empty(5000).exponential_(1.0, generator=g1)
            u = torch.empty(5000, dtype=torch.float64).uniform_(0.0, 1.0, generator=g2)
            b = torch.empty(5000)
            lib.wnv_exponential_from_uniform(u.data_ptr(), b.data_ptr(), 5000, 2)
            ok = bool(torch.equal(a, b)) and bool(torch.equal(torch.empty(3).uniform_(generator=g1), torch.empty(3).uniform_(generator=g2)))
        except Exception:
            ok = False
        _BULK_OK["exp_fast"] = ok
    if not ok:
        return False
    import os
    from . import _lib
    assert out.dtype == torch.float32 and out.is_contiguous() and out.device.type == "cpu"
    n = out.numel()
    kw = {} if generator is None else {"generator": generator}
    # (the float64 staging buffer is kept: a fresh 4-MB tensor per chunk costs more in page faults than the draw itself -- measured
    #  on the GPU box, inside a process with a HIP context: 4.8 ms per chunk against 1.2 ms for uniform_ alone.  It is bounded --
    #  a large request is staged in passes of _U_CHUNK values: uniform_ walks the generator element by element, so consecutive
    #  passes see the stream one bulk call would -- and belongs to the calling thread.)
    u = getattr(_U_SCRATCH, "u", None)
    need = min(n, _U_CHUNK)
    if u is None or u.numel() < need:
        u = _U_SCRATCH.u = torch.empty(max(need, 1 << 19), dtype=torch.float64)
    flat = out.view(-1)
    nthreads = min(os.cpu_count() or 1, 16)
    native = _native_uniform_ok()
    gen = generator if generator is not None else torch.default_generator
    for a in range(0, n, _U_CHUNK):
        m = min(_U_CHUNK, n - a)
        if native:
            # the same draws from the library's own Mersenne Twister (wnv_mt19937_uniform53: ~1.5 ns per value against the 5-10 ns torch's
            # element-by-element walk costs inside a busy process), the generator advanced through its state blob
            # (torch's own uniform_ holds the generator's mutex for the whole draw; get_state / draw / set_state do not -- this module's
            #  lock keeps two threads of THIS path from reading the same state twice, and a generator that another thread draws from
            #  through torch itself at the same time must not be handed to this function: module docstring)
            with _NATIVE_DRAW_LOCK:
                st = gen.get_state()
                drawn = _lib.lib().wnv_mt19937_uniform53(st.data_ptr(), st.numel(), u.data_ptr(), m) == 0
                if drawn:
                    gen.set_state(st)
            if drawn:
                uu = u[:m]
            else:                                                 # (a state blob the library does not recognise: nothing was drawn, torch draws)
                uu = u[:m].uniform_(0.0, 1.0, **kw)
        else:
            uu = u[:m].uniform_(0.0, 1.0, **kw)
        _lib.check(_lib.lib().wnv_exponential_from_uniform(uu.data_ptr(), flat[a:a + m].data_ptr(), m, nthreads))
    return True


def _native_uniform_ok() -> bool:
    """One-time probe: does wnv_mt19937_uniform53 produce torch's float64 uniform_ draws AND leave the generator where torch would
    (values, state blob, the draws that follow)?  It reads CPUGeneratorImpl's state layout -- another torch build may lay it out
    differently: then the probe says no and the draws come from uniform_ itself."""
    ok = _BULK_OK.get("mt_native")
    if ok is None:
        try:
            from . import _lib
            lib = _lib.lib()
            ok = True
            for seed, sizes in ((4321, (1, 311, 5000, 7)), (77, (624, 313))):
                g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
                for m in sizes:
                    a = torch.empty(m, dtype=torch.float64).uniform_(0.0, 1.0, generator=g1)
                    st = g2.get_state()
                    b = torch.empty(m, dtype=torch.float64)
                    _lib.check(lib.wnv_mt19937_uniform53(st.data_ptr(), st.numel(), b.data_ptr(), m))
                    g2.set_state(st)
                    ok = ok and bool(torch.equal(a, b)) and bool(torch.equal(g1.get_state(), g2.get_state()))
                ok = ok and bool(torch.equal(torch.empty(9).normal_(generator=g1), torch.empty(9).normal_(generator=g2)))
        except Exception:
            ok = False
        _BULK_OK["mt_native"] = ok
    return ok


def noise_width(scalar_input: bool, output_distribution: str, out_channels: int) -> int:
    if scalar_input:
        if output_distribution == "Logistic":
            assert out_channels % 3 == 0
            return out_channels // 3 + 1
        if output_distribution == "Normal":
            if out_channels in (2, 3):
                return 1
            assert out_channels % 3 == 0
            return out_channels // 3 + 1
        raise ValueError(f"unknown output_distribution {output_distribution!r}")
    return out_channels


def make_noise_tape(T: int, B: int, *, scalar_input: bool, output_distribution: str,
                    out_channels: int, generator: Optional[torch.Generator] = None,
                    per_step: bool = False) -> torch.Tensor:
    """Replay the reference's per-step CPU draws; returns a CPU float32 tensor (T, B, NZ).

    torch's CPU ``uniform_`` and ``exponential_`` consume the Mersenne-Twister stream element by element, so ONE bulk
    draw of the whole tape yields exactly the numbers the reference's per-step calls see (T = 65 536, B = 8, MoL:
    0.1 s instead of 1.3 s; tests/test_host_cpu.py pins bulk == per-step for every distribution and batch size).
    ``normal_`` does not: it switches to a 16-wide vectorised Box-Muller for tensors of >= 16 elements, so only calls
    of the same size reproduce the same stream -- Gaussian tapes are drawn in bulk when B is a multiple of 16 (every
    per-step call is vectorised then, and so is the bulk one), through a strided view when B < 16 (every per-step call takes the
    element-by-element path then, and a non-contiguous bulk call does too), and step by step otherwise.  ``per_step=True`` forces the
    literal replay (what the tests compare the bulk path with)."""
    nz = noise_width(scalar_input, output_distribution, out_channels)
    kw = {} if generator is None else {"generator": generator}
    if not scalar_input:
        per_step = per_step or not _bulk_matches_per_step("exponential")
        if per_step:
            return torch.stack([torch.empty(B, out_channels).exponential_(1.0, **kw) for _ in range(T)]) if T else torch.empty(0, B, nz)
        tape = torch.empty(T, B, out_channels)
        if T == 0 or not exponential_draws(tape, generator):
            tape.exponential_(1.0, **kw)
        return tape
    mix = nz - 1
    normal = output_distribution == "Normal"
    tape = torch.empty(T, B, nz, dtype=torch.float32)
    if not per_step and not normal and _bulk_matches_per_step("uniform"):
        raw = torch.empty(T, B * mix + B).uniform_(_EPS, 1.0 - _EPS, **kw)       # per step: u1 (B, 1, mix) then u2 (B, 1)
        tape[:, :, :mix] = raw[:, :B * mix].view(T, B, mix)
        tape[:, :, mix] = raw[:, B * mix:]
        return tape
    if not per_step and mix == 0 and B % 16 == 0 and _bulk_matches_per_step("normal16"):
        return torch.empty(T, B, 1).normal_(0.0, 1.0, **kw)
    if not per_step and mix == 0 and B < 16 and T > 0 and _bulk_matches_per_step("normal_strided"):
        # per-step calls of fewer than 16 elements take normal_'s element-by-element path (one Box-Muller pair per two values, the
        # second one cached in the generator), and so does a call on a NON-CONTIGUOUS tensor of any size: a strided view makes one
        # bulk draw walk the very same stream (T = 24 064, B = 8: 10 ms instead of 0.1 s of per-step calls)
        buf = torch.empty(T * B, 2)
        buf[:, 0].normal_(0.0, 1.0, **kw)
        return buf[:, 0].reshape(T, B, 1).contiguous()
    for t in range(T):
        if mix > 0:
            tape[t, :, :mix] = torch.empty(B, 1, mix).uniform_(_EPS, 1.0 - _EPS, **kw)[:, 0, :]
        if normal:
            tape[t, :, mix] = torch.empty(B, 1).normal_(0.0, 1.0, **kw)[:, 0]
        else:
            tape[t, :, mix] = torch.empty(B, 1).uniform_(_EPS, 1.0 - _EPS, **kw)[:, 0]
    return tape
