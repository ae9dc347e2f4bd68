"""Trained-magnitude weights for parity tests.

Every other model in tests/ is the reference's Kaiming initialisation (modules.py:13-18): pre-activations O(1), gates never
saturated, log-scales at -3.  A TRAINED vocoder is nothing like that -- gates saturate (|z| of 10-30), residual gains are large,
the sampler's log-scales sit between -5 and -14, mel frames have outliers.  No checkpoint is reachable offline, so the
numerical regime is produced by rule: `stress_state` maps a list of (key, shape) of the reference's weight-normed ``state_dict``
 to tensors whose magnitudes are chosen per role.  The SAME function feeds the real reference in
tests/golden/make_golden.py (which stores only the reference's outputs; the weights are re-made here, bit for bit, from the spec
in the fixture's meta) and the models under test.

spec = {"gain": g, "seed": s, "mode": "all" | "drive"}:
  mode "all" (default): weight_g of the dilated / 1x1 convolutions ~ U(0.6 g, 1.4 g) per output channel (a row of the folded weight
      then has that 2-norm: |z| ~ g |h|; the reference's initialisation has row norms of ~2, so g = 2 / 4 / 8 are x1 / x2 / x4),
      conv1x1_out / conv1x1_skip ~ U(0.3 g, 0.8 g), biases ~ U(-2, 2).  Scaling the RECURRENT path makes the 24-layer map itself
      ill-conditioned (a perturbation grows ~g-fold per layer: at g = 8 ANY two f32 evaluation orders differ by O(1), ATen's own
      online and offline paths included) -- tests/test_gpu_stress.py therefore measures against the float64 answer;
  mode "drive": the recurrent path keeps the initialisation's norms (dilated conv ~2, conv1x1_out / skip ~1) and the SATURATION comes
      from what drives the gates of a trained vocoder -- the conditioning 1x1s ~ U(0.6 g, 1.4 g) and biases ~ U(-g, g): gate
      pre-activations of +-10 ... +-60 on a well-conditioned map, where the strict tolerance applies.
The head's last 1x1: mixture logits / means modest, log-scale biases ~ U(-14, -5) (scalar-input models) or logit biases ~ N(0, 2)
(one-hot models)."""
import hashlib

import torch


def _gen(seed, key):
    h = int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:6], "little")
    return torch.Generator().manual_seed(h)


def _u(shape, lo, hi, g):
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def stress_state(keys_shapes, spec, *, scalar_input, out_channels, output_distribution="Logistic"):
    """{key: tensor} in the reference's WEIGHT-NORMED layout for the given [(key, shape), ...]."""
    gain, seed = float(spec["gain"]), int(spec["seed"])
    drive = spec.get("mode", "all") == "drive"
    sd = {}
    for key, shape in keys_shapes:
        shape = tuple(int(x) for x in shape)
        g = _gen(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        mod = key.rsplit(".", 1)[0]
        head_out = mod == "last_conv_layers.3"
        if leaf == "weight_v":
            t = torch.randn(shape, generator=g)
        elif leaf == "weight_g":
            if mod == "first_conv":
                t = _u(shape, 0.5, 2.0, g)
            elif mod.endswith(".conv") and drive:
                t = _u(shape, 1.2, 2.8, g)
            elif mod.endswith(".conv") or mod.endswith("conv1x1c") or mod.endswith("conv1x1g"):
                t = _u(shape, 0.6 * gain, 1.4 * gain, g)
            elif (mod.endswith("conv1x1_out") or mod.endswith("conv1x1_skip")) and drive:
                t = _u(shape, 0.6, 1.4, g)
            elif mod.endswith("conv1x1_out") or mod.endswith("conv1x1_skip"):
                t = _u(shape, 0.3 * gain, 0.8 * gain, g)
            elif mod == "last_conv_layers.1":
                t = _u(shape, 0.5, 2.0, g)
            elif head_out:
                t = _u(shape, 0.05, 0.3, g) if scalar_input else _u(shape, 0.5, 3.0, g)
            else:                                               # upsampling convs etc.: unit gain
                t = _u(shape, 0.8, 1.2, g)
        elif leaf == "bias":
            if head_out and scalar_input:
                C = out_channels
                t = _u(shape, -0.8, 0.8, g)
                single = output_distribution == "Normal" and C in (2, 3)
                if single:
                    t[C - 1] = float(_u((1,), -14.0, -5.0, g))
                else:
                    k = C // 3
                    t[:k] = torch.randn(k, generator=g)                     # mixture logits
                    t[2 * k:] = _u((k,), -14.0, -5.0, g)                     # log-scales of a trained model
            elif head_out:
                t = torch.randn(shape, generator=g) * 2.0
            elif mod == "first_conv":
                t = _u(shape, -1.0, 1.0, g)
            elif drive and mod.endswith(".conv"):
                t = _u(shape, -gain, gain, g)
            else:
                t = _u(shape, -2.0, 2.0, g)
        elif leaf == "weight":                                   # embed_speakers.weight, plain (un-normed) convolutions
            t = torch.randn(shape, generator=g)
        else:
            raise KeyError(f"no stress rule for {key}")
        sd[key] = t.float().contiguous()
    return sd


def keys_shapes_of(state_dict):
    return [(k, list(v.shape)) for k, v in state_dict.items()]


def stress_mel(shape, seed, amp=6.0):
    """Conditioning frames with outliers: N(0, 2.5) clipped to +-amp."""
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * 2.5).clamp_(-amp, amp)


def stress_teacher(B, T, seed):
    """Scalar teacher input that touches the rails: tanh-shaped noise blown up and clipped, so a good share of the samples is
    exactly +-1 (what a clipped recording or a saturated free run feeds back)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.tanh(torch.randn(B, 1, T, generator=g) * 0.8) * 1.4).clamp_(-1.0, 1.0)


def close_enough(got, want, atol=1e-4, rtol=1e-5):
    """head outputs <= 1e-4 absolute or 1e-5 relative.  Returns (ok, worst excess, worst abs err)."""
    d = (got.double() - want.double()).abs()
    lim = atol + rtol * want.double().abs()
    return bool((d <= lim).all()), float((d - lim).max()), float(d.max())


CORE_PREFIXES = ("first_conv.", "conv_layers.", "last_conv_layers.")


def apply_stress_(model, spec):
    """Give THIS package's model (fused ``weight`` / ``bias`` layout) the trained-magnitude weights of `spec`: the weight-normed keys
    the reference would have are made by ``stress_state`` for the network proper (first_conv, the gated layers, the
    head) and folded exactly as ``make_generation_fast_`` folds them; the upsampling network and the speaker table keep their
    initial values.  Returns the model."""
    from wavenet_vocoder_amd.conv import fold_weight_norm_
    sd = model.state_dict()
    ks = []
    for k, v in sd.items():
        if not k.startswith(CORE_PREFIXES):
            continue
        if k.endswith(".weight"):
            pre = k[:-len("weight")]
            ks.append((pre + "weight_g", [v.shape[0]] + [1] * (v.dim() - 1)))
            ks.append((pre + "weight_v", list(v.shape)))
        else:
            ks.append((k, list(v.shape)))
    cfgk = model._wnv_config_kwargs()
    wn = stress_state(ks, spec, scalar_input=cfgk["scalar_input"], out_channels=cfgk["out_channels"],
                      output_distribution=cfgk["output_distribution"])
    for k in [k for k in list(wn) if k.endswith("weight_g")]:
        fold_weight_norm_(wn, k[:-len("weight_g")])
    sd.update(wn)
    model.load_state_dict(sd)
    return model
