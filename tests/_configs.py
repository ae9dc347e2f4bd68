"""The five BASELINE.json configurations (SURVEY.md 8d shape resolutions) as constructor kwargs, shared by
the GPU parity tests, smoke() and bench.py.  Weights are random-init under a seed (no checkpoints exist
offline) with the last 1x1 tamed so free-running samples do not saturate."""
import torch

MEL = dict(cin_channels=80, cin_pad=2, upsample_conditional_features=True,
           upsample_params=dict(upsample_scales=[4, 4, 4, 4], cin_channels=80, cin_pad=2))

CONFIGS = {
    # cfg0: mulaw256 softmax, 8 layers, 64 residual channels, no conditioning (CPU-runnable plumbing case)
    "cfg0_mulaw256_small": dict(out_channels=256, layers=8, stacks=2, residual_channels=64, gate_channels=128,
                                skip_out_channels=64, kernel_size=3, dropout=0.0),
    # cfg1: egs/mulaw256 as BASELINE.json words it: 24 layers, 128 res / 256 skip, mel-conditioned
    "cfg1_mulaw256": dict(out_channels=256, layers=24, stacks=4, residual_channels=128, gate_channels=256,
                          skip_out_channels=256, kernel_size=3, dropout=0.0, **MEL),
    # cfg1b: the in-tree egs/mulaw256 preset: 30 layers / 3 stacks (dilation up to 512), 128/256/128
    "cfg1b_mulaw256_intree": dict(out_channels=256, layers=30, stacks=3, residual_channels=128, gate_channels=256,
                                  skip_out_channels=128, kernel_size=3, dropout=0.0, **MEL),
    # cfg2: egs/mol -- THE metric configuration
    "cfg2_mol": dict(out_channels=30, layers=24, stacks=4, residual_channels=128, gate_channels=256,
                     skip_out_channels=128, kernel_size=3, dropout=0.0, scalar_input=True,
                     output_distribution="Logistic", **MEL),
    # cfg3: egs/gaussian (in-tree: 24 layers / 4 stacks, out_channels 2)
    "cfg3_gaussian": dict(out_channels=2, layers=24, stacks=4, residual_channels=128, gate_channels=256,
                          skip_out_channels=128, kernel_size=3, dropout=0.0, scalar_input=True,
                          output_distribution="Normal", **MEL),
    # cfg3b: egs/gaussian as BASELINE.json words it: 30 layers / 3 stacks (dilation up to 512)
    "cfg3b_gaussian30": dict(out_channels=2, layers=30, stacks=3, residual_channels=128, gate_channels=256,
                             skip_out_channels=128, kernel_size=3, dropout=0.0, scalar_input=True,
                             output_distribution="Normal", **MEL),
    # cfg4: MoL + global speaker embedding, 512 skip channels
    "cfg4_mol_multispeaker": dict(out_channels=30, layers=24, stacks=4, residual_channels=128, gate_channels=256,
                                  skip_out_channels=512, kernel_size=3, dropout=0.0, scalar_input=True,
                                  output_distribution="Logistic", gin_channels=16, n_speakers=7,
                                  use_speaker_embedding=True, **MEL),
    # the geometry of the reference's published CMU-ARCTIC models (docs/content/index.md:390-402: 24 layers, 512 / 512 / 256) with the
    # scalar MoL output of egs/mol -- what the group-ring kernel for wide models (csrc/wnv_wide.hip) is for
    "wide_mol_512": dict(out_channels=30, layers=24, stacks=4, residual_channels=512, gate_channels=512,
                         skip_out_channels=256, kernel_size=3, dropout=0.0, scalar_input=True,
                         output_distribution="Logistic", **MEL),
}


def tame_head_(model):
    """Shrink the last 1x1 and push log-scale biases down so random-init samples stay inside (-1, 1)
    (unscaled random init saturates at +-1)."""
    last = model.last_conv_layers[3]
    with torch.no_grad():
        last.weight.mul_(0.25)
        C = model.out_channels
        if model.scalar_input:
            if C == 2:
                last.bias[1] = -3.0
            elif C % 3 == 0:
                last.bias[2 * (C // 3):] = -3.0
    return model


def build(name, seed=0):
    import wavenet_vocoder_amd as wnv
    torch.manual_seed(seed)
    m = wnv.WaveNet(**CONFIGS[name]).eval()
    return tame_head_(m)


def inputs(name, B, T, seed=1):
    """Seeded synthetic inputs: mel c ~ N(0,1) of shape (B, 80, T/256 + 4), speaker ids."""
    kw = CONFIGS[name]
    g = torch.Generator().manual_seed(seed)
    c = gids = None
    if kw.get("cin_channels", -1) > 0:
        assert T % 256 == 0
        c = torch.randn(B, 80, T // 256 + 2 * kw["cin_pad"], generator=g)
    if kw.get("gin_channels", -1) > 0:
        gids = torch.randint(0, kw["n_speakers"], (B, 1), generator=g)
    return c, gids
