"""Loader for the committed golden fixtures (tests/golden/*.npz, made by make_golden.py from the
real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASE_NAMES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f not in ("layers.npz", "mel_preset.npz"))


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.name = name
        self.meta = json.loads(str(z["__meta__"]))
        self.kwargs = self.meta["kwargs"]
        self.wn = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("wn/")}
        self.fused = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fused/")}
        self.stress = self.meta.get("extras", {}).get("stress")
        if self.stress:                # trained-magnitude cases: the weights are re-made from the spec make_golden.py used
            from tests._stress import stress_state
            self.wn = stress_state(self.meta["keys_shapes"], self.stress, scalar_input=self.kwargs.get("scalar_input", False),
                                   out_channels=self.kwargs["out_channels"],
                                   output_distribution=self.kwargs.get("output_distribution", "Logistic"))
        self.fused_is_derived = not self.fused
        if self.fused_is_derived:      # large cases store only the reference's weight-normed state_dict; make_golden.py proved
            from wavenet_vocoder_amd.conv import fold_weight_norm_     # this fold equal to make_generation_fast_ when it wrote them
            self.fused = {k: v.clone() for k, v in self.wn.items()}
            for k in [k for k in list(self.fused) if k.endswith("weight_g")]:
                fold_weight_norm_(self.fused, k[:-len("weight_g")])
        self.io = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("io/")}

    def get(self, k):
        return self.io.get(k)


def oracle_config(kwargs):
    from oracle.wavenet_oracle import OracleConfig
    up = kwargs.get("upsample_params", {})
    return OracleConfig(
        out_channels=kwargs["out_channels"], layers=kwargs["layers"], stacks=kwargs["stacks"],
        residual_channels=kwargs["residual_channels"], gate_channels=kwargs["gate_channels"],
        skip_out_channels=kwargs["skip_out_channels"], kernel_size=kwargs["kernel_size"],
        cin_channels=kwargs.get("cin_channels", -1), gin_channels=kwargs.get("gin_channels", -1),
        n_speakers=kwargs.get("n_speakers"),
        upsample_conditional_features=kwargs.get("upsample_conditional_features", False),
        upsample_net=kwargs.get("upsample_net", "ConvInUpsampleNetwork"),
        upsample_scales=list(up.get("upsample_scales", [4, 4, 4, 4])),
        freq_axis_kernel_size=up.get("freq_axis_kernel_size", 1),
        upsample_activation=up.get("upsample_activation", "none"),
        upsample_activation_params=dict(up.get("upsample_activation_params", {})),
        upsample_mode=up.get("mode", "nearest"),
        cin_pad=kwargs.get("cin_pad", 0), scalar_input=kwargs.get("scalar_input", False),
        use_speaker_embedding=kwargs.get("use_speaker_embedding", False),
        output_distribution=kwargs.get("output_distribution", "Logistic"))


def load_layers():
    z = np.load(os.path.join(GOLDEN, "layers.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}
