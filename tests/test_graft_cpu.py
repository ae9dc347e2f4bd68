"""CPU: the reference-side graft as a real file (wavenet_vocoder_amd/graft.py), exercised against the REAL reference class
where it is importable (the authoring container; /root/reference does not travel to the GPU box, so those cases skip there).

  * ``make_wavenet_amd(wavenet_vocoder.WaveNet)`` subclasses the reference's own class: constructor, parameters, state_dict
    keys, forward stay the reference's;
  * the engine configuration is READ OFF the module tree (the reference keeps few constructor arguments) and equals the
    explicit one for every configuration of the test-suite;
  * a reference state_dict -- weight-normed as trained, and fused after make_generation_fast_() -- goes through
    wnv_create(device = -1) + wnv_load_weights (validation, fold, packing) and reports SURVEY.md 8d's work figures;
  * incremental_forward is the engine's: eval-mode error as the reference's, loud failure on a CPU module (no fallback)."""
import os
import sys
import warnings

import pytest
import torch

import wavenet_vocoder_amd as wnv
from wavenet_vocoder_amd.graft import EngineHost, infer_config_kwargs, make_wavenet_amd
from tests._configs import CONFIGS
from tests._golden import CASE_NAMES, Case

REF = "/root/reference"
have_ref = os.path.isdir(os.path.join(REF, "wavenet_vocoder"))
needs_ref = pytest.mark.skipif(not have_ref, reason="the reference tree is only present in the authoring container")


def ref_wavenet():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import wavenet_vocoder
    return wavenet_vocoder.WaveNet


ALL_KW = [(n, kw) for n, kw in CONFIGS.items()] + [(n, Case(n).kwargs) for n in CASE_NAMES]


@pytest.mark.parametrize("name,kw", ALL_KW, ids=[n for n, _ in ALL_KW])
def test_config_read_off_the_module_equals_the_explicit_one(name, kw):
    m = wnv.WaveNet(**kw)
    assert infer_config_kwargs(m) == m._cfg_kwargs


@needs_ref
@pytest.mark.parametrize("name,kw", ALL_KW, ids=[n for n, _ in ALL_KW])
def test_config_read_off_the_reference_module(name, kw):
    WaveNetAMD = make_wavenet_amd(ref_wavenet())
    torch.manual_seed(0)
    assert infer_config_kwargs(WaveNetAMD(**kw)) == wnv.WaveNet(**kw)._cfg_kwargs


@needs_ref
def test_graft_is_a_subclass_of_the_reference_and_loads_its_checkpoints():
    Ref = ref_wavenet()
    WaveNetAMD = make_wavenet_amd(Ref)
    assert issubclass(WaveNetAMD, Ref) and issubclass(WaveNetAMD, EngineHost)
    assert WaveNetAMD.incremental_forward is EngineHost.incremental_forward          # the loop is the engine's ...
    assert WaveNetAMD.forward is Ref.forward and WaveNetAMD.make_generation_fast_ is Ref.make_generation_fast_   # ... the rest is not
    kw = CONFIGS["cfg2_mol"]
    torch.manual_seed(0)
    ref_model = Ref(**kw)                                        # "a trained reference model"
    sd = ref_model.state_dict()
    assert any(k.endswith("weight_g") for k in sd)
    m = WaveNetAMD(**kw)
    m.load_state_dict(sd)                                        # the reference's own loader, the reference's own keys
    assert list(m.state_dict()) == list(sd)
    # native checkpoint path on the host: weight-normed as trained ...
    work = m.check_engine_checkpoint(batch=8)
    assert work == {"macs_per_sample": 3657600, "bytes_per_step": 15076504, "receptive_field": 505}
    # ... and fused, after the reference's make_generation_fast_ (wavenet.py:355-361)
    m.eval()
    m.make_generation_fast_()
    assert not any(k.endswith("weight_g") for k in m.state_dict())
    assert m.check_engine_checkpoint(batch=8) == work
    # the fold the engine applies == the reference's remove_weight_norm: same packed bytes either way is covered by
    # tests/test_host_cpu.py (golden wn vs fused layouts); here: a corrupted checkpoint is refused with the reference's words
    bad = dict(m.state_dict())
    bad["conv_layers.3.conv.weight"] = bad["conv_layers.3.conv.weight"][:, :64]
    from wavenet_vocoder_amd.engine import check_checkpoint, make_config
    with pytest.raises(ValueError, match="size mismatch for conv_layers.3.conv.weight"):
        check_checkpoint(make_config(**infer_config_kwargs(m)), bad)


@needs_ref
def test_graft_keeps_the_reference_contract_around_the_engine():
    WaveNetAMD = make_wavenet_amd(ref_wavenet())
    m = WaveNetAMD(**CONFIGS["cfg3_gaussian"])
    with pytest.raises(RuntimeError, match="only supports eval mode"):          # conv.py:19-20
        m.incremental_forward(T=4)
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):                  # the product path never runs on the host
        m.incremental_forward(c=torch.zeros(1, 80, 5), T=256)
    # the reference's batch forward is untouched (its own torch graph on CPU; T = 256 from one frame + 2 x 2 context frames)
    with torch.no_grad():
        y = m(torch.zeros(1, 1, 256), c=torch.zeros(1, 80, 5))
    assert y.shape == (1, 2, 256)


def test_our_wavenet_uses_the_same_mixin():
    assert issubclass(wnv.WaveNet, EngineHost)
    m = wnv.WaveNet(**CONFIGS["cfg2_mol"]).eval()
    assert m.check_engine_checkpoint(8)["bytes_per_step"] == 15076504
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.incremental_forward(c=torch.zeros(1, 80, 5), T=256)
