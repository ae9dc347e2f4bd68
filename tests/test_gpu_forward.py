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
    B, T = 2, 512 + 37          # not a multiple of the 64-step tile (c is given at sample rate below)
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
