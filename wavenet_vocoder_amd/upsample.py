"""Host-side definitions of the local-conditioning upsamplers.

They exist for two reasons: (1) a reference checkpoint must load, so the module tree carries the reference's parameter names --
``upsample_net.conv_in.weight`` and ``upsample_net.upsample.up_layers.{1,3,5,7}.weight`` (weight-normed or fused; the even
``up_layers`` slots are the parameter-free stretch modules) -- see ``wavenet_vocoder/upsample.py:29-85``; (2) the teacher-forced
batch ``WaveNet.forward`` evaluates them with torch ops.  During synthesis the upsampling is ``wnv_upsample`` in the HIP engine
(``csrc/wnv_upsample.hip``), which also writes the time-major layout the sample loop reads.

What a network computes: optionally a valid ``Conv1d(cin, cin, 2 cin_pad + 1, bias=False)`` over the frames, then
per scale ``s``: every frame repeated ``s`` times along time, followed by ONE ``2 s + 1``-tap FIR shared by all mel bins with zero
padding ``s`` (initialised to the moving average ``1 / (2 s + 1)``).  The plain network trims ``cin_pad * prod(scales)`` samples at
both ends instead of consuming the context frames in a ``conv_in``.
"""
from __future__ import annotations

from math import prod

from torch import nn

import torch.nn.functional as F

from .conv import WeightNormCompat

__all__ = ["Stretch2d", "UpsampleNetwork", "ConvInUpsampleNetwork"]


class Stretch2d(nn.Module):
    """Stretch of a ``(B, 1, C, T)`` map, ``x_scale`` along time and ``y_scale`` along the channel axis (upsample.py:14-21:
    ``F.interpolate(x, scale_factor=(y_scale, x_scale), mode=mode)``).  ``mode``: any mode F.interpolate takes for a 4-D map -- "nearest" (every reference preset), "bilinear", "bicubic", "area",
    "nearest-exact"."""

    def __init__(self, x_scale, y_scale, mode="nearest"):
        super().__init__()
        if mode not in ("nearest", "bilinear", "bicubic", "area", "nearest-exact"):
            raise NotImplementedError(f"mode={mode!r}: F.interpolate's modes for a 4-D map are nearest, bilinear, bicubic, area, nearest-exact")
        self.x_scale, self.y_scale, self.mode = int(x_scale), int(y_scale), mode

    def forward(self, x):
        if self.mode != "nearest":
            return F.interpolate(x, scale_factor=(self.y_scale, self.x_scale), mode=self.mode)
        if self.y_scale != 1:
            x = x.repeat_interleave(self.y_scale, dim=2)
        return x.repeat_interleave(self.x_scale, dim=3)


class _SharedFir(WeightNormCompat, nn.Conv2d):
    """``Conv2d(1, 1, (freq, 2 s + 1))``: one filter for every mel bin (loads ``weight`` or ``weight_g`` / ``weight_v``)."""

    def __init__(self, scale, freq_axis_kernel_size):
        taps = (freq_axis_kernel_size, 2 * scale + 1)
        super().__init__(1, 1, kernel_size=taps, padding=((freq_axis_kernel_size - 1) // 2, scale), bias=False)
        nn.init.constant_(self.weight, 1.0 / (taps[0] * taps[1]))


def _stages(scales, freq_axis_kernel_size, mode, activation, activation_params):
    """[stretch, FIR(, activation)] per scale -- the module indices are the reference's state_dict keys (upsample.py:38-49)."""
    layers = []
    for s in scales:
        layers += [Stretch2d(s, 1, mode), _SharedFir(s, freq_axis_kernel_size)]
        if activation != "none":
            layers.append(getattr(nn, activation)(**activation_params))
    return nn.ModuleList(layers)


class UpsampleNetwork(nn.Module):
    def __init__(self, upsample_scales, upsample_activation="none", upsample_activation_params={},
                 mode="nearest", freq_axis_kernel_size=1, cin_pad=0, cin_channels=80):
        super().__init__()
        self.upsample_scales = [int(s) for s in upsample_scales]
        self.freq_axis_kernel_size = freq_axis_kernel_size
        self.upsample_activation, self.upsample_activation_params = upsample_activation, dict(upsample_activation_params)
        self.indent = cin_pad * prod(self.upsample_scales)          # samples dropped at either end
        self.up_layers = _stages(self.upsample_scales, freq_axis_kernel_size, mode, upsample_activation, self.upsample_activation_params)

    def forward(self, c):
        y = c[:, None]                                               # (B, 1, C, T)
        for layer in self.up_layers:
            y = layer(y)
        y = y[:, 0]
        return y[:, :, self.indent:y.size(-1) - self.indent] if self.indent else y


class ConvInUpsampleNetwork(nn.Module):
    def __init__(self, upsample_scales, upsample_activation="none", upsample_activation_params={},
                 mode="nearest", freq_axis_kernel_size=1, cin_pad=0, cin_channels=80):
        super().__init__()
        # the context frames are consumed here, so the inner network trims nothing
        self.conv_in = nn.Conv1d(cin_channels, cin_channels, kernel_size=2 * cin_pad + 1, bias=False)
        self.upsample = UpsampleNetwork(upsample_scales, upsample_activation, upsample_activation_params, mode,
                                        freq_axis_kernel_size, cin_pad=0, cin_channels=cin_channels)

    def forward(self, c):
        return self.upsample(self.conv_in(c))
