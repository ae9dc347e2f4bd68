"""Throughput of the device mel front end (wnv_logmel): 8 utterances x 10 s at 22.05 kHz (the audio the
bench batch of the sample loop corresponds to) and a 64 x 60 s batch.  HBM roofline: algorithmic bytes = 4 n (signal, once)
+ 4 frames num_mels (features, once) per utterance."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wavenet_vocoder_amd.audio import MelFrontEnd, default_hparams

fe = MelFrontEnd(default_hparams())
for B, secs in ((8, 10), (64, 60)):
    n = int(22050 * secs)
    y = torch.randn(B, n, device="cuda") * 0.1
    for _ in range(3):
        out = fe.feats(y)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 20
    ev[0].record()
    for _ in range(reps):
        out = fe.feats(y)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    alg = 4.0 * B * n + 4.0 * out.numel()
    print(json.dumps({"workload": f"logmel {B} x {secs} s @22050 Hz, fft 1024 / hop 256 / 80 mels", "ms": round(ms, 4),
                      "frames_per_s": round(out.shape[0] * out.shape[1] / ms * 1e3), "audio_seconds_per_s": round(B * secs / ms * 1e3),
                      "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak_8TBps": round(alg / ms / 1e6 / 8000.0, 4)}))
