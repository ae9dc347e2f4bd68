"""MI355X-native WaveNet-vocoder synthesis engine: a drop-in for the ``incremental_forward`` path of
r9y9/wavenet_vocoder (same ``WaveNet`` / ``ResidualConv1dGLU`` API and checkpoint layout), with the
autoregressive loop, the samplers and the conditioning upsampler running as hand-written HIP kernels for
gfx950 behind the C ABI of ``include/wnv.h``.

Importing the package is cheap and GPU-free; the shared library is loaded on first use and its absence is
a hard error -- there is no CPU fallback for the synthesis path."""
from .wavenet import WaveNet, receptive_field_size  # noqa: F401

__version__ = "0.1.0"
