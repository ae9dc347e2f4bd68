"""Loader and harness of ``oracle/_ref`` -- the REAL reference package, byte-compiled by ``oracle/build_ref.py``.

TEST INFRASTRUCTURE: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

What the harness re-creates (none of it imports the reference's scripts, which need docopt / nnmnkwii / librosa):
  * ``build_model``      = ``train.build_model`` (train.py:887-918): ``WaveNet(**kwargs)``, then the caller's weights
  * ``Capture``          = the head output handed to the sampler at every step (the public API hides it for scalar-input models):
                           wraps ``wavenet.sample_from_discretized_mix_logistic`` / ``sample_from_mix_gaussian`` (wavenet.py:322-329)
  * ``incremental``      = ``synthesis.batch_wavegen``'s call (synthesis.py:55-64) under ``torch.no_grad()``; the noise comes from torch's
                           default CPU generator after ``torch.manual_seed(seed)`` -- the tape ``wavenet_vocoder_amd.noise.make_noise_tape``
                           draws from the same seed is bit-identical to those draws (tests/golden/make_golden.py::verify_tape_replay,
                           tests/test_gpu_vs_reference.py checks it again on the GPU box).
"""
import importlib
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


class ReferenceMissing(RuntimeError):
    pass


def available():
    return os.path.exists(os.path.join(REF_DIR, "wavenet_vocoder", "wavenet.pyc"))


def load_reference():
    """Import ``wavenet_vocoder`` (the reference) from ``oracle/_ref``; raises ``ReferenceMissing`` with the build instruction."""
    mod = sys.modules.get("wavenet_vocoder")
    if mod is not None and hasattr(mod, "WaveNet"):
        return mod                                                  # (already imported -- from _ref or from /root/reference itself)
    if not available():
        raise ReferenceMissing("oracle/_ref is not built: run `python oracle/build_ref.py` (or __graft_entry__.build()) where "
                               "/root/reference is present; the built directory travels with the tree")
    man = json.load(open(os.path.join(REF_DIR, "MANIFEST.json")))
    import importlib.util
    if man.get("bytecode_magic") != importlib.util.MAGIC_NUMBER.hex():
        raise ReferenceMissing(f"oracle/_ref was compiled by Python {man.get('python')} (magic {man.get('bytecode_magic')}); this "
                               f"interpreter is {sys.version.split()[0]} -- rebuild with oracle/build_ref.py")
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = importlib.import_module("wavenet_vocoder")
    return mod


def build_model(kwargs, state_dict=None, fast=True):
    """The reference's ``WaveNet`` with the given constructor arguments (train.py:887-918), eval mode, optionally
    ``make_generation_fast_()`` (wavenet.py:355-361; what synthesis.py:195 / evaluate.py:140 do) and the caller's (fused) weights."""
    import torch
    ref = load_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g = torch.random.get_rng_state()
        m = ref.WaveNet(**kwargs).eval()
        torch.random.set_rng_state(g)                               # constructing the model must not move the caller's noise stream
        if fast:
            m.make_generation_fast_()
    if state_dict is not None:
        missing = m.load_state_dict(state_dict, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys, missing
    return m


class _FunctionalShim:
    """Stands in for the name ``F`` inside the reference's ``wavenet`` module only: records what ``F.softmax`` is given
    (wavenet.py:332 -- the logits of a one-hot model), forwards everything else to ``torch.nn.functional``."""

    def __init__(self, real, rows):
        self._real, self._rows = real, rows

    def softmax(self, x, *a, **kw):
        self._rows.append(x.detach().clone().view(x.size(0), -1))
        return self._real.softmax(x, *a, **kw)

    def __getattr__(self, name):
        return getattr(self._real, name)


class Capture:
    """Record what the reference hands to its samplers, (B, O) per step: the arguments of ``sample_from_discretized_mix_logistic`` /
    ``sample_from_mix_gaussian`` (wavenet.py:322-329) and of ``F.softmax`` (wavenet.py:332).  ``params()`` -> (B, O, T)."""

    def __init__(self):
        self.rows = []
        self._orig = {}

    def __enter__(self):
        ref_wavenet = importlib.import_module("wavenet_vocoder.wavenet")
        self._mod = ref_wavenet
        for name in ("sample_from_discretized_mix_logistic", "sample_from_mix_gaussian"):
            orig = getattr(ref_wavenet, name)
            self._orig[name] = orig

            def wrapped(y, _orig=orig, **kw):
                self.rows.append(y.detach().clone().view(y.size(0), -1))
                return _orig(y, **kw)
            setattr(ref_wavenet, name, wrapped)
        self._orig["F"] = ref_wavenet.F
        ref_wavenet.F = _FunctionalShim(ref_wavenet.F, self.rows)
        return self

    def __exit__(self, *a):
        for name, orig in self._orig.items():
            setattr(self._mod, name, orig)

    def params(self):
        import torch
        return torch.stack(self.rows).permute(1, 2, 0).contiguous()


def incremental(model, *, seed, T, c=None, g=None, initial_input=None, test_inputs=None, softmax=True, quantize=True,
                log_scale_min=-16.0, capture=True):
    """``model.incremental_forward`` of the unmodified reference under ``torch.manual_seed(seed)``.
    Returns (output (B, C, T), captured head outputs (B, O, T) -- None without ``capture``)."""
    import torch
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(seed)
        if capture:
            with Capture() as cap:
                y = model.incremental_forward(initial_input=initial_input, c=c, g=g, T=T, test_inputs=test_inputs, softmax=softmax,
                                              quantize=quantize, log_scale_min=log_scale_min)
            return y, cap.params()
        y = model.incremental_forward(initial_input=initial_input, c=c, g=g, T=T, test_inputs=test_inputs, softmax=softmax,
                                      quantize=quantize, log_scale_min=log_scale_min)
    return y, None
