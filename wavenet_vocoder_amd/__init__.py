"""MI355X-native WaveNet-vocoder synthesis engine (drop-in for r9y9/wavenet_vocoder's
``WaveNet.incremental_forward`` path).  Importing the package is cheap and GPU-free; the HIP
shared library is loaded on first use and its absence is a hard error (no CPU fallback)."""
__version__ = "0.1.0"
