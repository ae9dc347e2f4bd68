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


# (the parameter is called ``s`` as in the reference, util.py:9-25: a keyword call ``is_mulaw(s=...)`` keeps working)
def is_scalar_input(s):
    """True for the two scalar-sample types ("raw", "mulaw")."""
    return _scalar(s)


def is_mulaw_quantize(s):
    return not _scalar(s)


def is_mulaw(s):
    return _scalar(s) and s == "mulaw"


def is_raw(s):
    return _scalar(s) and s == "raw"
