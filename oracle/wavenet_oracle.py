"""CPU oracle for the WaveNet-vocoder autoregressive synthesis path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under ``wavenet_vocoder_amd/``
does (``tests/test_host_cpu.py::test_product_never_imports_the_oracle`` enforces it).  The product path is the HIP engine behind
``include/wnv.h``; it fails loudly when its shared library is missing and never falls back to this file.

What it is: a plain ``torch``-on-CPU, float32 restatement of the reference algorithm
(r9y9/wavenet_vocoder @ v0.2.0).  Every function cites the reference lines it follows.  The arithmetic
library is the same ATen the reference itself dispatches to, the op ORDER is the
reference's (queue shift, strided gather, ``F.linear`` ...), so that timing this file is a fair stand-in
for timing the reference's CPU path on a machine where ``/root/reference`` does not exist.

Parity pinning: ``tests/golden/make_golden.py`` (run in the authoring container, where the reference
package is importable) generates fixtures from the REAL reference; ``tests/test_oracle_golden.py`` checks
this restatement against them (teacher-forced, free-running with a shared noise tape, all conditioning
modes, all three output distributions, the upsampler, receptive_field_size known answers from the
reference's tests/test_misc.py:6-10).

Randomness never comes from a generator in here: every sampler takes an explicit NOISE TAPE
(shape ``(T, B, NZ)``) whose layout is documented in ``noise_width``.  The same tape drives the HIP engine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "OracleConfig", "Oracle", "fold_weight_norm", "receptive_field_size", "noise_width",
    "sample_mol", "sample_gaussian", "sample_categorical",
]


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Constructor arguments of the reference ``WaveNet`` (wavenet_vocoder/wavenet.py:98-111)."""
    out_channels: int = 256
    layers: int = 20
    stacks: int = 2
    residual_channels: int = 512
    gate_channels: int = 512
    skip_out_channels: int = 512
    kernel_size: int = 3
    cin_channels: int = -1
    gin_channels: int = -1
    n_speakers: Optional[int] = None
    upsample_conditional_features: bool = False
    upsample_net: str = "ConvInUpsampleNetwork"
    upsample_scales: List[int] = field(default_factory=lambda: [4, 4, 4, 4])
    freq_axis_kernel_size: int = 1
    upsample_activation: str = "none"                      # upsample.py:30,47-49: getattr(nn, name)(**params) after every stage
    upsample_activation_params: Dict[str, float] = field(default_factory=dict)
    upsample_mode: str = "nearest"                         # upsample.py:20: the mode of Stretch2d's F.interpolate
    cin_pad: int = 0
    scalar_input: bool = False
    use_speaker_embedding: bool = False
    output_distribution: str = "Logistic"

    @property
    def dilations(self) -> List[int]:
        # wavenet.py:125-126 : dilation = 2 ** (layer % layers_per_stack)
        per = self.layers // self.stacks
        return [2 ** (i % per) for i in range(self.layers)]


def receptive_field_size(total_layers: int, num_cycles: int, kernel_size: int,
                         dilation=lambda x: 2 ** x) -> int:
    """wavenet.py:42-60 : (kernel_size - 1) * sum(dilations) + 1."""
    assert total_layers % num_cycles == 0
    per = total_layers // num_cycles
    return (kernel_size - 1) * sum(dilation(i % per) for i in range(total_layers)) + 1


def fold_weight_norm(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Fold ``weight_g``/``weight_v`` pairs into ``weight`` (what make_generation_fast_ does,
    wavenet.py:355-361 -> torch.nn.utils.remove_weight_norm): w = g * v / ||v||, the 2-norm taken over
    every dim except 0 (weight_norm default dim=0; applied at modules.py:18, upsample.py:44)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in state.items():
        v = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
        if k.endswith("weight_g"):
            base = k[: -len("weight_g")]
            vv = torch.as_tensor(np.asarray(state[base + "weight_v"])) if not torch.is_tensor(
                state[base + "weight_v"]) else state[base + "weight_v"]
            vv = vv.float()
            dims = tuple(range(1, vv.dim()))
            norm = vv.pow(2).sum(dim=dims, keepdim=True).sqrt()
            out[base + "weight"] = vv * (v.float() / norm)
        elif k.endswith("weight_v"):
            continue
        else:
            out[k] = v.float() if v.is_floating_point() else v
    return out


def noise_width(cfg: OracleConfig) -> int:
    """Floats of noise consumed per utterance per time step, in tape order:

    * Logistic (MoL, mixture.py:138,151):  nr_mix uniforms u1 (Gumbel) then 1 uniform u2      -> nr_mix + 1
    * Normal, C == 2 or 3 (mixture.py:258-267):  1 standard normal                             -> 1
    * Normal, C == 3k > 3 (mixture.py:245-267):  nr_mix uniforms then 1 standard normal        -> nr_mix + 1
    * one-hot / softmax (wavenet.py:332-335 -> torch.multinomial): out_channels Exp(1) draws   -> out_channels
    """
    if cfg.scalar_input:
        C = cfg.out_channels
        if cfg.output_distribution == "Logistic":
            return C // 3 + 1
        if cfg.output_distribution == "Normal":
            return 1 if C in (2, 3) else C // 3 + 1
        raise AssertionError(cfg.output_distribution)
    return cfg.out_channels


# ----------------------------------------------------------------------------------------------
# samplers with explicit noise
# ----------------------------------------------------------------------------------------------
def _select(y: torch.Tensor, u1: torch.Tensor, nr_mix: int):
    """Gumbel-max component pick + one-hot select (mixture.py:138-146 / :245-256)."""
    logit = y[:, :nr_mix]
    temp = logit - torch.log(-torch.log(u1))
    arg = temp.max(dim=-1)[1]
    one_hot = torch.zeros_like(logit).scatter_(1, arg.unsqueeze(-1), 1.0)
    means = torch.sum(y[:, nr_mix:2 * nr_mix] * one_hot, dim=-1)
    log_scales = torch.sum(y[:, 2 * nr_mix:3 * nr_mix] * one_hot, dim=-1)
    return means, log_scales


def sample_mol(y: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """mixture.py:118-156 with the two ``uniform_`` draws replaced by ``noise`` (B, nr_mix+1).
    ``y`` is (B, 3*nr_mix) = [logit_probs | means | log_scales].  No log-scale clamp: the caller
    (wavenet.py:324-325) never passes clamp_log_scale=True."""
    assert y.size(1) % 3 == 0
    nr_mix = y.size(1) // 3
    means, log_scales = _select(y, noise[:, :nr_mix], nr_mix)
    u = noise[:, nr_mix]
    x = means + torch.exp(log_scales) * (torch.log(u) - torch.log(1.0 - u))
    return torch.clamp(torch.clamp(x, min=-1.0), max=1.0)


def sample_gaussian(y: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """mixture.py:221-270; ``Normal(loc, scale).sample()`` is ``loc + scale * n`` with n ~ N(0,1)
    (torch.normal(mean, std) computes ``n * std + mean``)."""
    C = y.size(1)
    if C == 2:
        means, log_scales, n = y[:, 0], y[:, 1], noise[:, 0]
    elif C == 3:
        means, log_scales, n = y[:, 1], y[:, 2], noise[:, 0]
    else:
        assert C % 3 == 0
        nr_mix = C // 3
        means, log_scales = _select(y, noise[:, :nr_mix], nr_mix)
        n = noise[:, nr_mix]
    x = n * torch.exp(log_scales) + means
    return torch.clamp(x, min=-1.0, max=1.0)


def sample_categorical(p: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """wavenet.py:334-335 : OneHotCategorical(p).sample().  Categorical renormalises ``p`` by its
    sum and torch.multinomial(.,1) draws ``argmax(p_hat / e)``, e ~ Exp(1) (SURVEY.md A.3 [probe];
    re-verified by tests/golden/make_golden.py).  Returns the index (B,) int64."""
    p_hat = p / p.sum(-1, keepdim=True)
    return (p_hat / noise).argmax(-1)


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
class _QueueConv:
    """conv.py:7-65 : nn.Conv1d evaluated one step at a time from a shifted history buffer."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], dilation: int = 1, dtype=torch.float32):
        self.weight = weight.to(dtype).contiguous()         # (Cout, Cin, kw)
        self.bias = None if bias is None else bias.to(dtype).contiguous()
        self.kw = weight.size(2)
        self.dilation = int(dilation)
        # conv.py:51-62 : [o, k*Cin + i] = W[o, i, k]
        self.lin = self.weight.transpose(1, 2).contiguous().view(weight.size(0), -1)
        self.buf: Optional[torch.Tensor] = None

    def clear(self):
        self.buf = None                                     # conv.py:48-49

    def step(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, 1, Cin) -> (B, 1, Cout); conv.py:17-46."""
        bsz = x.size(0)
        if self.kw > 1:
            if self.buf is None:
                rows = self.kw + (self.kw - 1) * (self.dilation - 1)
                self.buf = x.new_zeros(bsz, rows, x.size(2))
            else:
                self.buf[:, :-1, :] = self.buf[:, 1:, :].clone()      # conv.py:39 (the 70 % copy_)
            self.buf[:, -1, :] = x[:, -1, :]
            x = self.buf
            if self.dilation > 1:
                x = x[:, 0::self.dilation, :].contiguous()
        return F.linear(x.reshape(bsz, -1), self.lin, self.bias).view(bsz, 1, -1)

    def full(self, x: torch.Tensor, padding: int) -> torch.Tensor:
        """batch form (B, Cin, T) -> (B, Cout, T + ...), used by Oracle.forward."""
        return F.conv1d(x, self.weight, self.bias, padding=padding, dilation=self.dilation)


class _Layer:
    """modules.py:52-169 ResidualConv1dGLU (eval mode, dropout = identity)."""

    def __init__(self, st: Dict[str, torch.Tensor], prefix: str, dilation: int, kw: int, dtype=torch.float32):
        g = lambda n: st.get(prefix + n)
        self.dilation = dilation
        self.kw = kw
        self.conv = _QueueConv(g("conv.weight"), g("conv.bias"), dilation, dtype)
        self.c1 = _QueueConv(g("conv1x1c.weight"), None, 1, dtype) if g("conv1x1c.weight") is not None else None
        self.g1 = _QueueConv(g("conv1x1g.weight"), None, 1, dtype) if g("conv1x1g.weight") is not None else None
        self.out = _QueueConv(g("conv1x1_out.weight"), g("conv1x1_out.bias"), 1, dtype)
        self.skip = _QueueConv(g("conv1x1_skip.weight"), g("conv1x1_skip.bias"), 1, dtype)

    def clear(self):
        for c in (self.conv, self.c1, self.g1, self.out, self.skip):
            if c is not None:
                c.clear()

    def _tail(self, x, residual, c, g, split, inc):
        a, b = x.split(x.size(split) // 2, dim=split)                 # modules.py:138
        if c is not None:                                             # modules.py:141-145
            cc = self.c1.step(c) if inc else self.c1.full(c, 0)
            ca, cb = cc.split(cc.size(split) // 2, dim=split)
            a, b = a + ca, b + cb
        if g is not None:                                             # modules.py:148-152
            gg = self.g1.step(g) if inc else self.g1.full(g, 0)
            ga, gb = gg.split(gg.size(split) // 2, dim=split)
            a, b = a + ga, b + gb
        z = torch.tanh(a) * torch.sigmoid(b)                          # modules.py:154
        s = self.skip.step(z) if inc else self.skip.full(z, 0)        # modules.py:157
        o = self.out.step(z) if inc else self.out.full(z, 0)          # modules.py:160
        return (o + residual) * math.sqrt(0.5), s                     # modules.py:162

    def step(self, x, c=None, g=None):
        """(B,1,R),(B,1,cin),(B,1,gin) -> (B,1,R),(B,1,K); modules.py:112-113,127-163."""
        return self._tail(self.conv.step(x), x, c, g, -1, True)

    def full(self, x, c=None, g=None):
        """(B,R,T) teacher-forced; causal padding then drop the future frames, modules.py:132-136."""
        T = x.size(-1)
        y = self.conv.full(x, (self.kw - 1) * self.dilation)[:, :, :T]
        return self._tail(y, x, c, g, 1, False)


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
class Oracle:
    """Reference ``WaveNet`` restated; weights from a reference ``state_dict`` (either layout).

    ``dtype``: float32 is the reference's arithmetic and what every parity test compares with.  float64 evaluates the SAME float32
    weights (folded in float32, then widened) in double: the exact answer an f32 evaluation order -- ATen's or a HIP kernel's -- is an
    approximation of; tests/test_gpu_stress.py uses it to tell an inaccurate kernel from an ill-conditioned model."""

    def __init__(self, cfg: OracleConfig, state: Dict[str, torch.Tensor], dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        st = fold_weight_norm({k: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)))
                               for k, v in state.items()})
        self.st = st
        self.first = _QueueConv(st["first_conv.weight"], st["first_conv.bias"], 1, dtype)
        self.layers = [_Layer(st, f"conv_layers.{i}.", d, cfg.kernel_size, dtype)
                       for i, d in enumerate(cfg.dilations)]
        self.last1 = _QueueConv(st["last_conv_layers.1.weight"], st["last_conv_layers.1.bias"], 1, dtype)
        self.last3 = _QueueConv(st["last_conv_layers.3.weight"], st["last_conv_layers.3.bias"], 1, dtype)
        self.embed = st.get("embed_speakers.weight")
        if self.embed is not None:
            self.embed = self.embed.to(dtype)
        self.receptive_field = receptive_field_size(cfg.layers, cfg.stacks, cfg.kernel_size)

    # -- upsample.py ---------------------------------------------------------------------------
    def upsample(self, c: torch.Tensor) -> torch.Tensor:
        """upsample.py:69-85 (ConvInUpsampleNetwork) / :29-66 (UpsampleNetwork).
        (B, cin, Tc + 2*cin_pad) -> (B, cin, Tc * prod(scales))."""
        cfg = self.cfg
        conv_in = cfg.upsample_net == "ConvInUpsampleNetwork"
        if conv_in:
            c = F.conv1d(c, self.st["upsample_net.conv_in.weight"])          # upsample.py:78,84
            pre = "upsample_net.upsample.up_layers."
        else:
            pre = "upsample_net.up_layers."
        c = c.unsqueeze(1)                                                   # upsample.py:58
        fk = cfg.freq_axis_kernel_size
        act = None if cfg.upsample_activation == "none" else getattr(torch.nn, cfg.upsample_activation)(**cfg.upsample_activation_params)
        stride = 2 if act is None else 3                                     # up_layers = [stretch, conv(, activation)] per scale
        for i, s in enumerate(cfg.upsample_scales):
            c = F.interpolate(c, scale_factor=(1, s), mode=cfg.upsample_mode)   # upsample.py:19-21
            w = self.st[f"{pre}{stride * i + 1}.weight"]
            c = F.conv2d(c, w, padding=((fk - 1) // 2, s))                   # upsample.py:39-42
            if act is not None:
                c = act(c)                                                   # upsample.py:47-49
        c = c.squeeze(1)
        if not conv_in:
            indent = cfg.cin_pad * int(np.prod(cfg.upsample_scales))         # upsample.py:36,64-65
            if indent > 0:
                c = c[:, :, indent:-indent]
        return c

    def _embed_g(self, g, B):
        if g is None:
            return None
        if self.embed is not None:                                           # wavenet.py:264-268
            g = F.embedding(g.view(B, -1).long(), self.embed).transpose(1, 2)
        g = g.to(self.dtype)
        return g.unsqueeze(-1) if g.dim() == 2 else g                        # (B, gin, 1)

    def clear_buffer(self):
        self.first.clear()
        for f in self.layers:
            f.clear()
        self.last1.clear()
        self.last3.clear()

    # -- wavenet.py:164-213 --------------------------------------------------------------------
    def forward(self, x, c=None, g=None, softmax=False):
        B, _, T = x.shape
        g = self._embed_g(g, B)
        g_bct = None if g is None else g.expand(B, -1, T).contiguous()
        if c is not None and self.cfg.upsample_conditional_features:
            c = self.upsample(c)
            assert c.size(-1) == T
        x = x.to(self.dtype)
        c = None if c is None else c.to(self.dtype)
        h = self.first.full(x, 0)
        skips = 0
        for f in self.layers:
            h, s = f.full(h, c, g_bct)
            skips = skips + s
        skips = skips * math.sqrt(1.0 / len(self.layers))
        y = self.last3.full(F.relu(self.last1.full(F.relu(skips), 0)), 0)
        return F.softmax(y, dim=1) if softmax else y

    # -- wavenet.py:215-343 --------------------------------------------------------------------
    def incremental_forward(self, initial_input=None, c=None, g=None, T=100, test_inputs=None,
                            softmax=True, quantize=True, noise=None, return_params=False,
                            max_seconds=None):
        """Same contract as the reference; ``noise`` is the (T, B, noise_width) tape that replaces
        the generator.  With ``return_params`` also returns the pre-sampling head output (B, O, T).
        ``max_seconds`` (bench.py's bounded CPU-baseline sample) stops the loop after that much wall time;
        ``self.last_steps`` / ``self.last_seconds`` then tell how far it got."""
        import time as _time
        _t0 = _time.perf_counter()
        cfg = self.cfg
        self.clear_buffer()
        B = 1
        if test_inputs is not None:                                          # wavenet.py:245-258
            if cfg.scalar_input:
                if test_inputs.size(1) == 1:
                    test_inputs = test_inputs.transpose(1, 2).contiguous()
            elif test_inputs.size(1) == cfg.out_channels:
                test_inputs = test_inputs.transpose(1, 2).contiguous()
            B = test_inputs.size(0)
            T = test_inputs.size(1) if T is None else max(T, test_inputs.size(1))
        T = int(T)
        g = self._embed_g(g, B if c is None else c.shape[0])
        if c is not None:                                                    # wavenet.py:272-278
            B = c.shape[0]
            if cfg.upsample_conditional_features:
                c = self.upsample(c)
                assert c.size(-1) == T, (c.size(-1), T)
            if c.size(-1) == T:
                c = c.transpose(1, 2).contiguous()
        g_btc = None if g is None else g.expand(g.size(0), -1, T).transpose(1, 2).contiguous()
        if initial_input is None:                                            # wavenet.py:281-289
            if cfg.scalar_input:
                initial_input = torch.zeros(B, 1, 1)
            else:
                initial_input = torch.zeros(B, 1, cfg.out_channels)
                initial_input[:, :, 127] = 1
        elif initial_input.size(1) == cfg.out_channels:                      # wavenet.py:291-292
            initial_input = initial_input.transpose(1, 2).contiguous()
        cur = initial_input.to(self.dtype)
        if c is not None:
            c = c.to(self.dtype)
        outs, params = [], []
        scale = math.sqrt(1.0 / len(self.layers))
        self.last_steps, self.last_seconds = 0, 0.0
        for t in range(T):
            if max_seconds is not None and t > 0 and (t & 15) == 0:
                self.last_seconds = _time.perf_counter() - _t0
                if self.last_seconds > max_seconds:
                    break
            self.last_steps = t + 1
            if test_inputs is not None and t < test_inputs.size(1):          # wavenet.py:297-301
                cur = test_inputs[:, t, :].unsqueeze(1).to(self.dtype)
            elif t > 0:
                cur = outs[-1]
            ct = None if c is None else c[:, t, :].unsqueeze(1)
            gt = None if g_btc is None else g_btc[:, t, :].unsqueeze(1)
            x = self.first.step(cur)
            skips = 0
            for f in self.layers:
                x, s = f.step(x, ct, gt)
                skips = skips + s
            skips = skips * scale
            x = self.last3.step(F.relu(self.last1.step(F.relu(skips))))      # wavenet.py:315-319
            if return_params:
                params.append(x.view(B, -1).clone())
            nz = None if noise is None else torch.as_tensor(noise[t])
            if cfg.scalar_input:                                             # wavenet.py:322-330
                if cfg.output_distribution == "Logistic":
                    x = sample_mol(x.view(B, -1), nz).view(B, 1)
                elif cfg.output_distribution == "Normal":
                    x = sample_gaussian(x.view(B, -1), nz).view(B, 1)
                else:
                    raise AssertionError(cfg.output_distribution)
            else:                                                            # wavenet.py:332-335
                x = F.softmax(x.view(B, -1), dim=1) if softmax else x.view(B, -1)
                if quantize:
                    idx = sample_categorical(x, nz)
                    x = torch.zeros_like(x).scatter_(1, idx.unsqueeze(-1), 1.0)
            outs.append(x.reshape(B, 1, -1))
        self.last_seconds = _time.perf_counter() - _t0
        y = torch.stack([o.view(B, -1) for o in outs])                       # T x B x C
        y = y.transpose(0, 1).transpose(1, 2).contiguous()                   # B x C x T
        self.clear_buffer()
        if return_params:
            p = torch.stack(params).transpose(0, 1).transpose(1, 2).contiguous()
            return y, p
        return y
