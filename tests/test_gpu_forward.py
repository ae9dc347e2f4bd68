"""-m gpu: the MFMA teacher-forced batch evaluation (wnv_forward) against
  * the CPU oracle's forward (torch CPU f32, pinned to the reference's own forward by tests/test_oracle_golden.py),
  * the same module evaluated with torch ops (the reference graph) on the GPU,
  * the on-line path: incremental_forward's teacher-forced head outputs (online == offline, reference
    tests/test_model.py:147-366, atol 1e-4)."""
import pytest
import torch

from oracle.wavenet_oracle import Oracle
from tests._configs import CONFIGS, build, inputs
from tests._golden import oracle_config
from tests.test_gpu_configs import teacher

pytestmark = pytest.mark.gpu
TOL = 1e-4
NAMES = ["cfg1_mulaw256", "cfg1b_mulaw256_intree", "cfg2_mol", "cfg3_gaussian", "cfg4_mol_multispeaker"]


@pytest.mark.parametrize("name", NAMES)
def test_forward_vs_oracle_and_torch_graph(name):
    kw = CONFIGS[name]
    B, T = 2, 512 + 37          # not a multiple of the 128-step tile, nor of 4 (c is given at sample rate below)
    m = build(name)
    o = Oracle(oracle_config(kw), m.state_dict())
    g = torch.Generator().manual_seed(8)
    x = teacher(kw, B, T)
    c_up = torch.randn(B, T, 80, generator=g)
    gids = torch.randint(0, kw["n_speakers"], (B, 1), generator=g) if kw.get("gin_channels", -1) > 0 else None
    torch.set_num_threads(8)
    saved = o.cfg.upsample_conditional_features
    o.cfg.upsample_conditional_features = False
    want = o.forward(x, c=c_up.transpose(1, 2).contiguous(), g=gids)
    o.cfg.upsample_conditional_features = saved
    m = m.to("cuda")
    eng = m._get_engine()
    got = eng.forward(x.cuda(), c_up=c_up.cuda(), g_ids=None if gids is None else gids[:, 0].cuda())
    err = (got.cpu() - want).abs().max().item()
    assert err < TOL, f"{name}: forward differs from the oracle by {err}"
    sm = eng.forward(x.cuda(), c_up=c_up.cuda(), g_ids=None if gids is None else gids[:, 0].cuda(), softmax=True)
    assert (sm.cpu() - torch.softmax(want, dim=1)).abs().max().item() < TOL


@pytest.mark.parametrize("name", ["cfg2_mol", "cfg1b_mulaw256_intree"])
def test_module_forward_dispatch_and_online_equals_offline(name):
    kw = CONFIGS[name]
    B, T = 2, 256
    m = build(name).to("cuda")
    c, gids = inputs(name, B, T)
    x = teacher(kw, B, T).cuda()
    with torch.no_grad():
        off = m(x, c=c.cuda())                                        # MFMA path (eval, no grad, GPU)
    # the same graph with torch ops: grad mode keeps the module on the reference evaluation
    with torch.enable_grad():
        ref = m(x, c=c.cuda())
    assert (off - ref.detach()).abs().max().item() < TOL
    eng = m._get_engine()
    c_up = eng.upsample(c.cuda(), T_expected=T)
    _, params, _ = eng.generate(B=B, T=T, c_up=c_up, teacher=x.transpose(1, 2).contiguous(), softmax=True, quantize=False,
                                want_params=True, kernel=0)
    assert (params - off).abs().max().item() < TOL, "online == offline"


def test_unsupported_shapes_raise_and_module_falls_back():
    m = build("cfg0_mulaw256_small").to("cuda")
    x = teacher(CONFIGS["cfg0_mulaw256_small"], 1, 64).cuda()
    with pytest.raises(NotImplementedError, match="MFMA forward"):
        m._get_engine().forward(x)
    with torch.no_grad():
        y = m(x)                                                      # torch ops
    assert y.shape == (1, 256, 64)


ODD = {
    # conditioning that is no multiple of the 16-row chunk (zero-padded rows, the clamped partial weight chunk), kernel size 2 (an odd
    # number of GEMM1 chunks), three blocks of output channels
    "cin20_k2_skip256": (dict(out_channels=30, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=256, kernel_size=2,
                              dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=20, upsample_conditional_features=False), 3, 1000),
    # no conditioning at all, a single partial tile
    "plain_T100": (dict(out_channels=2, layers=6, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
                        dropout=0.0, scalar_input=True, output_distribution="Normal"), 2, 100),
    # fewer time steps than a wave owns, and a single one
    "tiny_T31": (dict(out_channels=30, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
                      dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80, upsample_conditional_features=False), 2, 31),
    "tiny_T1": (dict(out_channels=30, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
                     dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80, upsample_conditional_features=False), 3, 1),
    # first_conv on the matrix pipe (one-hot models): a class count that is odd and no multiple of the 8-row trip, a DENSE input
    # (every row of the 251 x T operand matters, unlike a one-hot one), T one more than a wave's 32 steps
    "soft251_T33": (dict(out_channels=251, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
                         dropout=0.0, cin_channels=80, upsample_conditional_features=False), 2, 33),
    "soft6_T300": (dict(out_channels=6, layers=4, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=2,
                        dropout=0.0), 3, 300),
    # speaker embedding + conditioning, kernel size 3, T a multiple of the tile
    "speakers_T384": (dict(out_channels=256, layers=6, stacks=3, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
                           dropout=0.0, cin_channels=36, gin_channels=8, n_speakers=5, use_speaker_embedding=True,
                           upsample_conditional_features=False), 2, 384),
}


@pytest.mark.parametrize("name", list(ODD))
def test_forward_odd_shapes_vs_the_torch_graph(name):
    """Shapes that exercise the edges of the tile kernel, against the same module evaluated with torch ops on the GPU (the reference
    graph: grad mode keeps the module off the MFMA path)."""
    import wavenet_vocoder_amd as wnv
    kw, B, T = ODD[name]
    torch.manual_seed(5)
    m = wnv.WaveNet(**kw).eval().to("cuda")
    g = torch.Generator().manual_seed(3)
    scalar = kw.get("scalar_input", False)
    x = torch.tanh(torch.randn(B, 1, T, generator=g)) if scalar else torch.nn.functional.one_hot(
        torch.randint(0, kw["out_channels"], (B, T), generator=g), kw["out_channels"]).transpose(1, 2).float()
    if name.startswith("soft"):
        x = torch.softmax(2.0 * torch.randn(B, kw["out_channels"], T, generator=g), dim=1)
    c = torch.randn(B, kw["cin_channels"], T, generator=g).cuda() if kw.get("cin_channels", -1) > 0 else None
    gid = torch.randint(0, kw["n_speakers"], (B, 1), generator=g).cuda() if kw.get("gin_channels", -1) > 0 else None
    x = x.cuda()
    with torch.no_grad():
        got = m(x, c=c, g=gid)
        assert m._mfma_forward_covers(x, c, gid), "this shape must be served by the MFMA kernels"
    with torch.enable_grad():
        ref = m(x, c=c, g=gid).detach()
    err = (got - ref).abs().max().item()
    assert err < TOL, f"{name}: {err}"
    with torch.no_grad():
        sm = m(x, c=c, g=gid, softmax=True)
    assert (sm - torch.softmax(ref, dim=1)).abs().max().item() < TOL


def test_raw_forward_arguments_are_validated_before_the_launch():
    """Pointers cross the C boundary raw: the engine wrapper checks shape, dtype, layout and device first (no out-of-bounds device reads)."""
    name = "cfg2_mol"
    m = build(name).to("cuda")
    eng = m._get_engine()
    B, T = 2, 256
    x = teacher(CONFIGS[name], B, T).cuda()
    c_up = torch.randn(B, T, 80, device="cuda")
    eng.forward(x, c_up=c_up)
    with pytest.raises(ValueError, match="c_up"):
        eng.forward(x, c_up=c_up[:, : T - 1].contiguous())                # too short
    with pytest.raises(ValueError, match="c_up"):
        eng.forward(x, c_up=c_up.double())
    with pytest.raises(ValueError, match="c_up"):
        eng.forward(x, c_up=c_up.transpose(1, 2))                         # (B, cin, T) view: wrong layout
    with pytest.raises(RuntimeError, match="HIP device"):
        eng.forward(x, c_up=c_up.cpu())
    with pytest.raises(ValueError, match="channels"):
        eng.forward(torch.cat([x, x], dim=1), c_up=c_up)
