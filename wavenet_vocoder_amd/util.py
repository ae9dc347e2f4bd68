"""Input-type predicates (same names and meaning as the reference's wavenet_vocoder/util.py:9-25)."""

_VALID = ("mulaw-quantize", "mulaw", "raw")


def _check(s):
    assert s in _VALID, f"input_type must be one of {_VALID}, got {s!r}"


def is_mulaw_quantize(s):
    _check(s)
    return s == "mulaw-quantize"


def is_mulaw(s):
    _check(s)
    return s == "mulaw"


def is_raw(s):
    _check(s)
    return s == "raw"


def is_scalar_input(s):
    return is_raw(s) or is_mulaw(s)
