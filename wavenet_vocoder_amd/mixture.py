"""Samplers with the reference's names (wavenet_vocoder/mixture.py:109-156, :221-270), in torch ops, for
callers that sample outside the engine.  Inside ``WaveNet.incremental_forward`` sampling is fused into the
HIP kernel (sample_scalar / sample_categorical in csrc/wnv_generic.hip).  The training losses of the
reference's mixture.py are out of scope."""
from __future__ import annotations

import torch

__all__ = ["to_one_hot", "sample_from_discretized_mix_logistic", "sample_from_mix_gaussian"]


def to_one_hot(tensor, n, fill_with=1.0):
    one_hot = torch.zeros(tensor.size() + (n,), dtype=torch.float32, device=tensor.device)
    return one_hot.scatter_(tensor.dim(), tensor.unsqueeze(-1), fill_with)


def _pick(y, nr_mix):
    logit = y[:, :, :nr_mix]
    u = torch.empty_like(logit).uniform_(1e-5, 1.0 - 1e-5)
    arg = (logit - torch.log(-torch.log(u))).max(dim=-1)[1]
    sel = to_one_hot(arg, nr_mix)
    return (torch.sum(y[:, :, nr_mix:2 * nr_mix] * sel, dim=-1),
            torch.sum(y[:, :, 2 * nr_mix:3 * nr_mix] * sel, dim=-1))


def sample_from_discretized_mix_logistic(y, log_scale_min=-7.0, clamp_log_scale=False):
    """y (B, 3*nr_mix, T) -> (B, T) in [-1, 1]."""
    assert y.size(1) % 3 == 0
    nr_mix = y.size(1) // 3
    means, log_scales = _pick(y.transpose(1, 2), nr_mix)
    if clamp_log_scale:
        log_scales = torch.clamp(log_scales, min=log_scale_min)
    u = torch.empty_like(means).uniform_(1e-5, 1.0 - 1e-5)
    x = means + torch.exp(log_scales) * (torch.log(u) - torch.log(1.0 - u))
    return torch.clamp(x, min=-1.0, max=1.0)


def sample_from_mix_gaussian(y, log_scale_min=-7.0):
    """y (B, C, T), C == 2 | 3 | 3*nr_mix -> (B, T) in [-1, 1]."""
    C = y.size(1)
    y = y.transpose(1, 2)
    if C == 2:
        means, log_scales = y[:, :, 0], y[:, :, 1]
    elif C == 3:
        means, log_scales = y[:, :, 1], y[:, :, 2]
    else:
        assert C % 3 == 0
        means, log_scales = _pick(y, C // 3)
    x = torch.normal(means, torch.exp(log_scales))
    return torch.clamp(x, min=-1.0, max=1.0)
