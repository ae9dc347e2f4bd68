"""``Conv1d`` with a queue-cached single-step evaluation, API-compatible with the reference's
``wavenet_vocoder.conv.Conv1d`` (conv.py:7-65) -- ``incremental_forward(input (B,1,C)) -> (B,1,Cout)``
and ``clear_buffer()`` -- but the step runs in the HIP engine (``wnv_qconv_*`` in include/wnv.h): the
history is a device ring buffer written once per step instead of a buffer that is shifted by a full clone
(conv.py:39), and the weight is stored K-major once instead of being re-linearised (conv.py:51-62).

Checkpoint compatibility: parameters are kept fused (``weight``/``bias``, the layout the reference has
after ``make_generation_fast_``); ``load_state_dict`` also accepts the weight-normed ``weight_g`` /
``weight_v`` pairs of a training checkpoint and folds them (w = g * v / ||v||).
"""
from __future__ import annotations

from torch import nn

from .engine import QueueConv, require_gpu_tensor

__all__ = ["Conv1d", "WeightNormCompat", "fold_weight_norm_"]


def fold_weight_norm_(state_dict, prefix: str) -> None:
    """In ``state_dict`` replace ``prefix+weight_g/_v`` by the fused ``prefix+weight`` (in place)."""
    kg, kv = prefix + "weight_g", prefix + "weight_v"
    if kg in state_dict and kv in state_dict:
        g, v = state_dict.pop(kg), state_dict.pop(kv)
        v32 = v.float()
        norm = v32.reshape(v32.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v32.dim() - 1)))
        state_dict[prefix + "weight"] = (v32 * (g.float() / norm)).to(v.dtype)


class WeightNormCompat:
    """Mixin: accept weight-normed checkpoints in ``load_state_dict``."""

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        fold_weight_norm_(state_dict, prefix)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class Conv1d(WeightNormCompat, nn.Conv1d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._qconv = None
        self._qconv_key = None

    def _engine(self) -> QueueConv:
        w = self.weight
        require_gpu_tensor(w, "Conv1d.weight")
        key = (w.device, w.data_ptr(), w._version, None if self.bias is None else self.bias._version)
        if self._qconv is None or self._qconv_key != key:
            self._qconv = QueueConv(self.in_channels, self.out_channels, self.kernel_size[0], self.dilation[0],
                                    w.device)
            self._qconv.set_weights(w, self.bias)
            self._qconv_key = key
        return self._qconv

    def incremental_forward(self, input):
        """input: (B, 1, C) (only the last time step is used, as in the reference) -> (B, 1, Cout)."""
        if self.training:
            raise RuntimeError('incremental_forward only supports eval mode')     # conv.py:19-20
        y = self._engine().step(input[:, -1, :])
        return y.view(input.size(0), 1, -1)

    def invalidate_engine(self):
        self._qconv, self._qconv_key = None, None

    def clear_buffer(self):
        if getattr(self, "_qconv", None) is not None:
            self._qconv.reset()
