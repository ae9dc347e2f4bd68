"""Input-type predicates of the reference (``wavenet_vocoder/util.py:9-25``): same function names, same answers, same
AssertionError for an unknown ``input_type``.  The three types and what they mean for the engine:

    "raw"             scalar waveform in [-1, 1]                  -> scalar_input, MoL / Gaussian head
    "mulaw"           mu-law companded scalar in [-1, 1]          -> scalar_input, MoL / Gaussian head
    "mulaw-quantize"  mu-law class index, fed back as a one-hot   -> categorical head
"""

_SCALAR = {"raw": True, "mulaw": True, "mulaw-quantize": False}


def _scalar(input_type):
    assert input_type in _SCALAR, f"unknown input_type {input_type!r} (expected one of {sorted(_SCALAR)})"
    return _SCALAR[input_type]


def is_scalar_input(input_type):
    """True for the two scalar-sample types ("raw", "mulaw")."""
    return _scalar(input_type)


def is_mulaw_quantize(input_type):
    return not _scalar(input_type)


def is_mulaw(input_type):
    return _scalar(input_type) and input_type == "mulaw"


def is_raw(input_type):
    return _scalar(input_type) and input_type == "raw"
