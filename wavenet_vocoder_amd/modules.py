"""Layer factories and the gated residual block, API-compatible with the reference's
``wavenet_vocoder.modules`` (modules.py:13-169).  ``ResidualConv1dGLU.incremental_forward`` runs as one
HIP kernel per step (``wnv_glu_step``); the batch ``forward`` is ordinary torch ops (parity oracle and
likelihood scoring, not the hot path)."""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.nn import functional as F

from . import conv
from .engine import GluLayer, require_gpu_tensor

__all__ = ["Conv1d", "Embedding", "Conv1d1x1", "ResidualConv1dGLU"]


def Conv1d(in_channels, out_channels, kernel_size, dropout=0, **kwargs):
    """He-initialised conv (the reference adds weight norm on top, modules.py:13-18; with g initialised
    to ||v|| the effective weight is the same tensor, and this package keeps weights fused)."""
    m = conv.Conv1d(in_channels, out_channels, kernel_size, **kwargs)
    nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)
    return m


def Embedding(num_embeddings, embedding_dim, padding_idx, std=0.01):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    m.weight.data.normal_(0, std)
    return m


def Conv1d1x1(in_channels, out_channels, bias=True):
    return Conv1d(in_channels, out_channels, kernel_size=1, padding=0, dilation=1, bias=bias)


class ResidualConv1dGLU(nn.Module):
    """Dilated causal conv -> (+ local / global conditioning 1x1) -> tanh * sigmoid -> skip 1x1 and
    residual 1x1.  Constructor signature of reference modules.py:71-75."""

    def __init__(self, residual_channels, gate_channels, kernel_size, skip_out_channels=None,
                 cin_channels=-1, gin_channels=-1, dropout=1 - 0.95, padding=None, dilation=1,
                 causal=True, bias=True, *args, **kwargs):
        super().__init__()
        self.dropout = dropout
        if skip_out_channels is None:
            skip_out_channels = residual_channels
        if padding is None:
            padding = (kernel_size - 1) * dilation if causal else (kernel_size - 1) // 2 * dilation
        self.causal = causal
        self.conv = Conv1d(residual_channels, gate_channels, kernel_size, padding=padding,
                           dilation=dilation, bias=bias, *args, **kwargs)
        self.conv1x1c = Conv1d1x1(cin_channels, gate_channels, bias=False) if cin_channels > 0 else None
        self.conv1x1g = Conv1d1x1(gin_channels, gate_channels, bias=False) if gin_channels > 0 else None
        half = gate_channels // 2
        self.conv1x1_out = Conv1d1x1(half, residual_channels, bias=bias)
        self.conv1x1_skip = Conv1d1x1(half, skip_out_channels, bias=bias)
        self._geom = dict(residual_channels=residual_channels, gate_channels=gate_channels,
                          kernel_size=kernel_size, skip_out_channels=skip_out_channels,
                          cin_channels=cin_channels, gin_channels=gin_channels, dilation=dilation, bias=bias)
        self._glu = None
        self._glu_key = None

    # -- batch path (torch ops) ------------------------------------------------------------------
    def forward(self, x, c=None, g=None):
        """x (B,R,T), c (B,cin,T), g (B,gin,T) -> (x', s); reference modules.py:109-110,127-163."""
        residual = x
        x = F.dropout(x, p=self.dropout, training=self.training)
        x = self.conv(x)
        if self.causal:
            x = x[:, :, :residual.size(-1)]
        a, b = x.split(x.size(1) // 2, dim=1)
        if c is not None:
            assert self.conv1x1c is not None
            ca, cb = self.conv1x1c(c).split(x.size(1) // 2, dim=1)
            a, b = a + ca, b + cb
        if g is not None:
            assert self.conv1x1g is not None
            ga, gb = self.conv1x1g(g).split(x.size(1) // 2, dim=1)
            a, b = a + ga, b + gb
        z = torch.tanh(a) * torch.sigmoid(b)
        s = self.conv1x1_skip(z)
        o = self.conv1x1_out(z)
        return (o + residual) * math.sqrt(0.5), s

    # -- incremental path (HIP) ------------------------------------------------------------------
    def _engine(self) -> GluLayer:
        params = [p for p in self.parameters()]
        require_gpu_tensor(params[0], "ResidualConv1dGLU parameters")
        key = tuple((p.device, p.data_ptr(), p._version) for p in params)
        if self._glu is None or self._glu_key != key:
            self._glu = GluLayer(device=params[0].device, **self._geom)
            self._glu.load_weights({k: v for k, v in self.state_dict().items()})
            self._glu_key = key
        return self._glu

    def incremental_forward(self, x, c=None, g=None):
        """x (B,1,R), c (B,1,cin), g (B,1,gin) -> (x' (B,1,R), s (B,1,K)); reference modules.py:112-113."""
        if self.training:
            raise RuntimeError('incremental_forward only supports eval mode')     # conv.py:19-20
        if c is not None:
            assert self.conv1x1c is not None                                      # modules.py:142
        if g is not None:
            assert self.conv1x1g is not None                                      # modules.py:149
        B = x.size(0)
        xo, so = self._engine().step(x[:, -1, :], None if c is None else c[:, -1, :],
                                     None if g is None else g[:, -1, :])
        return xo.view(B, 1, -1), so.view(B, 1, -1)

    def invalidate_engine(self):
        """Drop the packed weights of the layer-level engine (see WaveNet.invalidate_engine)."""
        self._glu, self._glu_key = None, None

    def clear_buffer(self):
        if self._glu is not None:
            self._glu.reset()
        for m in (self.conv, self.conv1x1_out, self.conv1x1_skip, self.conv1x1c, self.conv1x1g):
            if m is not None:
                m.clear_buffer()
