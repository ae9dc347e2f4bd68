"""-m gpu: THE GRAFT ON THE REAL REFERENCE CLASS, on a GPU (VERDICT r05 missing #2).

north_star: "a drop-in for synthesis.py".  The reference's caller is ``synthesis.batch_wavegen`` (synthesis.py:61-64)

    y_hat = model.incremental_forward(c=c, g=g, T=T, tqdm=tqdm, softmax=True, quantize=True, log_scale_min=hparams.log_scale_min)

on an instance of ``wavenet_vocoder/wavenet.py:63`` (``class WaveNet``).  Here that class is the reference's OWN (``oracle/_ref``: the
package byte-compiled from /root/reference, loaded through ``oracle.reference.load_reference``), built by its own constructor in the
WEIGHT-NORMED layout it trains and checkpoints in (``weight_g`` / ``weight_v`` parameters, the reference's ``conv.Conv1d`` with its
``_linearized_weight`` cache and backward hooks); ``make_wavenet_amd(ref.WaveNet)`` grafts the engine on, the model goes ``.eval().cuda()``
and is called exactly as ``batch_wavegen`` calls it -- against the unmodified reference's CPU run of the same checkpoint under the same
seed (the reference draws its noise from torch's default CPU generator; the graft's ``rng = "replay"`` replays that stream).  Once
before and once after ``make_generation_fast_()`` (wavenet.py:355-361, what synthesis.py:195 / evaluate.py:140 do).

Criteria as everywhere (tests/_margins.py): head outputs <= 1e-4 (the reference's own tolerance, tests/test_model.py:361-366); a free
run parts from the reference's only through a flipped discrete choice at a near tie; one-hot classes equal otherwise."""
import warnings

import pytest
import torch

from oracle import reference as R
from tests._configs import CONFIGS, inputs
from tests._margins import assert_free_run_agrees_until_near_tie
from tests._refrun import tape_for, teacher
from wavenet_vocoder_amd.graft import EngineHost, make_wavenet_amd

pytestmark = pytest.mark.gpu
TOL = 1e-4


def tame_weight_normed_head_(m):
    """tests/_configs.py::tame_head_ for the weight-normed layout: ``weight`` is recomputed from (weight_g, weight_v) by the pre-hook, so
    the gain is what gets scaled."""
    last = m.last_conv_layers[3]
    with torch.no_grad():
        last.weight_g.mul_(0.25)
        C = m.out_channels
        if m.scalar_input:
            if C == 2:
                last.bias[1] = -3.0
            elif C % 3 == 0:
                last.bias[2 * (C // 3):] = -3.0
    return m


def reference_pair(name):
    """(the reference's model on the CPU, the grafted reference class on the GPU) -- one weight-normed checkpoint, the reference's own
    constructor and loader on both sides."""
    ref = R.load_reference()                                   # ReferenceMissing = FAIL on a GPU box
    kw = CONFIGS[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(0)
        cpu = tame_weight_normed_head_(ref.WaveNet(**kw)).eval()
        sd = cpu.state_dict()
        assert any(k.endswith("weight_g") for k in sd) and not any(k.endswith("conv.weight") for k in sd)
        WaveNetAMD = make_wavenet_amd(ref.WaveNet)
        assert issubclass(WaveNetAMD, ref.WaveNet) and WaveNetAMD.incremental_forward is EngineHost.incremental_forward
        assert WaveNetAMD.forward is ref.WaveNet.forward
        gpu = WaveNetAMD(**kw)
        gpu.load_state_dict(sd)                                # the reference's loader, the reference's keys
        gpu = gpu.eval().cuda()
    assert type(gpu.first_conv).__module__.startswith("wavenet_vocoder.") and not hasattr(gpu, "_cfg_kwargs")
    return ref, kw, cpu, gpu


def batch_wavegen_call(model, c, g, T, seed, capture):
    """synthesis.py:61-64, argument for argument (tqdm as the reference imports it: a callable wrapping the range)."""
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(seed)
        if capture:
            with R.Capture() as cap:
                y = model.incremental_forward(c=c, g=g, T=T, tqdm=lambda x: x, softmax=True, quantize=True, log_scale_min=-16.0)
            return y, cap.params()
        return model.incremental_forward(c=c, g=g, T=T, tqdm=lambda x: x, softmax=True, quantize=True, log_scale_min=-16.0), None


def check_free_run(kw, y, params, want, wparams, tape, what):
    scalar = kw.get("scalar_input", False)
    assert y.shape == want.shape and y.dtype == want.dtype, (y.shape, want.shape)
    err0 = float((params[:, :, 0] - wparams[:, :, 0]).abs().max())
    assert err0 < TOL, f"{what}: head outputs of step 0 differ by {err0:.3e}"
    if scalar:
        gs, ws = y, want
    else:
        assert torch.equal(y.sum(1), torch.ones_like(y.sum(1)))
        gs, ws = y.argmax(1), want.argmax(1)
    hz = assert_free_run_agrees_until_near_tie(gs, ws, params, wparams, tape, kw, t0=0, what=what)
    # while the trajectories agree the head outputs do too, to the reference's tolerance
    for b, s in enumerate(hz):
        e = float((params[b, :, :s] - wparams[b, :, :s]).abs().max())
        assert e < TOL, f"{what}: utterance {b}: head outputs differ by {e:.3e} within the agreeing horizon {s}"
    return hz


# (cfg4 at B = 1: with ``g`` given and no ``test_inputs`` the reference embeds the speaker ids while its B is still 1 -- wavenet.py:241,262-266
#  ``g.view(B, -1)`` runs ahead of ``B = c.shape[0]`` :272 -- so ``batch_wavegen``'s call only works for one speaker-conditioned utterance
#  at a time; larger speaker-conditioned batches are compared in tests/test_gpu_vs_reference.py, where a one-step ``test_inputs`` tells the
#  reference the batch size)
@pytest.mark.parametrize("name,B,T", [("cfg2_mol", 2, 256), ("cfg1_mulaw256", 2, 256), ("cfg4_mol_multispeaker", 1, 256)])
def test_grafted_reference_class_on_the_gpu_vs_the_reference_cpu_run(name, B, T):
    ref, kw, cpu, gpu = reference_pair(name)
    c, g = inputs(name, B, T)
    seed = 23
    tape = tape_for(kw, T, B, seed)
    torch.set_num_threads(8)
    horizons = {}
    for phase in ("weight-normed", "after make_generation_fast_()"):
        if phase != "weight-normed":
            cpu.make_generation_fast_()
            gpu.make_generation_fast_()                        # the REFERENCE's method (wavenet.py:355-361) on the grafted class
            assert not any(k.endswith("weight_g") for k in gpu.state_dict())
        want, wparams = batch_wavegen_call(cpu, c, g, T, seed, capture=True)
        gpu.capture_params = True
        y, _ = batch_wavegen_call(gpu, c.cuda(), None if g is None else g.cuda(), T, seed, capture=False)
        assert y.is_cuda and y.is_contiguous() and not y.requires_grad              # wavenet.py:336-340
        assert gpu._engine.last_kernel() in (2, 3), "the graft ran on the fallback kernel"
        hz = check_free_run(kw, y.cpu(), gpu.last_params.cpu(), want, wparams, tape, f"{name}, {phase}")
        horizons[phase] = hz
        assert min(hz) >= 32, hz
        # the generator ends where the reference's does: the NEXT draw of the caller is the same on both sides
        torch.manual_seed(seed)
        cpu.incremental_forward(c=c[:, :, :5], g=g, T=256, softmax=True, quantize=True)
        a = torch.rand(3)
        torch.manual_seed(seed)
        gpu.incremental_forward(c=c[:, :, :5].cuda(), g=None if g is None else g.cuda(), T=256, softmax=True, quantize=True)
        assert torch.equal(a, torch.rand(3)), f"{name}, {phase}: the graft leaves torch's generator elsewhere than the reference does"
    print(f"{name}: make_wavenet_amd(reference WaveNet) on cuda vs the reference's CPU run: agreeing horizons {horizons} of {T}")
    # the reference contract around the engine, on the grafted class
    gpu.train()
    with pytest.raises(RuntimeError, match="only supports eval mode"):              # conv.py:19-20
        gpu.incremental_forward(c=c.cuda(), g=None if g is None else g.cuda(), T=T)
    gpu.eval()
    gpu.cpu()


def test_grafted_reference_class_teacher_forced_and_batch_forward():
    """tests/test_model.py:147-366's pattern on the grafted class: teacher-forced ``incremental_forward(test_inputs=x, c=c, T=None)`` of the
    ENGINE against the batch ``forward`` the class inherits from the reference (its own torch graph, here on the GPU through ATen) and
    against the reference's CPU ``incremental_forward`` -- every head output, <= 1e-4."""
    name, B, T = "cfg2_mol", 2, 512
    ref, kw, cpu, gpu = reference_pair(name)
    c, _ = inputs(name, B, T)
    x = teacher(kw, B, T)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(5)
        with R.Capture() as cap:
            cpu.incremental_forward(test_inputs=x, c=c, T=None, softmax=False, quantize=False)
        wparams = cap.params()
        gpu.capture_params = True
        torch.manual_seed(5)
        y = gpu.incremental_forward(test_inputs=x.cuda(), c=c.cuda(), T=None, softmax=False, quantize=False)
        assert y.shape == (B, 1, T)
        params = gpu.last_params.cpu()
        offline = gpu(x.cuda(), c=c.cuda(), softmax=False).cpu()               # the reference's forward, inherited
    e_ref = float((params - wparams).abs().max())
    e_fwd = float((params - offline).abs().max())
    print(f"grafted {name}: engine (teacher-forced) vs the reference's CPU incremental_forward {e_ref:.2e}, vs the inherited batch forward on the GPU {e_fwd:.2e}")
    assert e_ref < TOL and e_fwd < TOL
    gpu.cpu()


def test_grafted_mulaw_model_at_its_baseline_batch_through_the_streamed_tape():
    """BASELINE configs[1]: the mu-law model at batch 1, long enough (T >= 1024) that the graft draws the reference's noise stream
    WHILE the ring kernel runs (graft.py::_generate_streamed) -- on the reference's class, after its own make_generation_fast_()."""
    name, B, T = "cfg1_mulaw256", 1, 1280
    ref, kw, cpu, gpu = reference_pair(name)
    cpu.make_generation_fast_()
    gpu.make_generation_fast_()
    c, _ = inputs(name, B, T)
    seed = 29
    tape = tape_for(kw, T, B, seed)
    torch.set_num_threads(8)
    want, wparams = batch_wavegen_call(cpu, c, None, T, seed, capture=True)
    gpu.capture_params = True
    y, _ = batch_wavegen_call(gpu, c.cuda(), None, T, seed, capture=False)
    assert gpu._engine.last_kernel() == 2
    hz = check_free_run(kw, y.cpu(), gpu.last_params.cpu(), want, wparams, tape, f"{name} streamed")
    print(f"grafted {name} B = 1, T = {T} (streamed replay tape): agreeing horizon {hz}")
    assert min(hz) >= 64
    gpu.cpu()
