"""Local-conditioning upsampling networks with the reference's module/parameter names
(wavenet_vocoder/upsample.py:12-85) so its checkpoints load.  ``forward`` here is the batch path in
torch ops; inside ``WaveNet.incremental_forward`` the upsampling runs in the HIP engine
(``wnv_upsample``), which writes the time-major layout the sample loop consumes."""
from __future__ import annotations

import numpy as np
from torch import nn
from torch.nn import functional as F

from .conv import WeightNormCompat

__all__ = ["Stretch2d", "UpsampleNetwork", "ConvInUpsampleNetwork"]


class Stretch2d(nn.Module):
    def __init__(self, x_scale, y_scale, mode="nearest"):
        super().__init__()
        self.x_scale, self.y_scale, self.mode = x_scale, y_scale, mode

    def forward(self, x):
        return F.interpolate(x, scale_factor=(self.y_scale, self.x_scale), mode=self.mode)


class _Conv2d(WeightNormCompat, nn.Conv2d):
    pass


class _Conv1dPlain(nn.Conv1d):
    pass


class UpsampleNetwork(nn.Module):
    def __init__(self, upsample_scales, upsample_activation="none", upsample_activation_params={},
                 mode="nearest", freq_axis_kernel_size=1, cin_pad=0, cin_channels=80):
        super().__init__()
        self.upsample_scales = list(upsample_scales)
        self.freq_axis_kernel_size = freq_axis_kernel_size
        self.up_layers = nn.ModuleList()
        total = int(np.prod(upsample_scales)) if len(upsample_scales) else 1
        self.indent = cin_pad * total
        for s in upsample_scales:
            k_size = (freq_axis_kernel_size, 2 * s + 1)
            conv = _Conv2d(1, 1, kernel_size=k_size, padding=((freq_axis_kernel_size - 1) // 2, s), bias=False)
            conv.weight.data.fill_(1.0 / np.prod(k_size))
            self.up_layers.append(Stretch2d(s, 1, mode))
            self.up_layers.append(conv)
            if upsample_activation != "none":
                # the HIP prologue implements the "none" activation of every reference preset
                raise NotImplementedError("upsample_activation != 'none' is not supported by the HIP engine")

    def forward(self, c):
        c = c.unsqueeze(1)
        for f in self.up_layers:
            c = f(c)
        c = c.squeeze(1)
        if self.indent > 0:
            c = c[:, :, self.indent:-self.indent]
        return c


class ConvInUpsampleNetwork(nn.Module):
    def __init__(self, upsample_scales, upsample_activation="none", upsample_activation_params={},
                 mode="nearest", freq_axis_kernel_size=1, cin_pad=0, cin_channels=80):
        super().__init__()
        self.conv_in = _Conv1dPlain(cin_channels, cin_channels, kernel_size=2 * cin_pad + 1, bias=False)
        self.upsample = UpsampleNetwork(upsample_scales, upsample_activation, upsample_activation_params,
                                        mode, freq_axis_kernel_size, cin_pad=0, cin_channels=cin_channels)

    def forward(self, c):
        return self.upsample(self.conv_in(c))
