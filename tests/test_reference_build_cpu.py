"""oracle/_ref -- the REAL reference, byte-compiled by oracle/build_ref.py -- builds, loads, is binary-only, and the harness around it
(tests/_refrun.py) means what it says: the oracle restatement, fed the replayed tape, reproduces the reference's run.  CPU only."""
import hashlib
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/wavenet_vocoder"


def _ensure_built():
    from oracle import build_ref, reference as R
    if os.path.isdir(REF_SRC):
        build_ref.build(verbose=False)
    if not R.available():
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref on this machine")
    return R


def test_ref_build_is_bytecode_only_and_matches_the_reference_tree():
    R = _ensure_built()
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    man = json.load(open(os.path.join(ref_dir, "MANIFEST.json")))
    files = sorted(os.listdir(os.path.join(ref_dir, "wavenet_vocoder")))
    assert files and all(f.endswith(".pyc") for f in files), files            # outputs only: no reference source text in the tree
    assert {f[:-4] for f in files} == set(man["modules"])
    if os.path.isdir(REF_SRC):
        for mod, sha in man["modules"].items():
            assert hashlib.sha256(open(os.path.join(REF_SRC, mod + ".py"), "rb").read()).hexdigest() == sha, mod
    ref = R.load_reference()
    assert ref.receptive_field_size(30, 3, 3) == 6139 and ref.receptive_field_size(24, 4, 3) == 505       # reference tests/test_misc.py:6-7
    # git must not see it
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in gi


def test_tape_replay_is_exact_against_the_reference_samplers():
    _ensure_built()
    from tests._refrun import tape_replay_is_exact
    assert tape_replay_is_exact()


@pytest.mark.parametrize("name,B", [("cfg0_mulaw256_small", 2), ("cfg2_mol", 2), ("cfg4_mol_multispeaker", 3)])
def test_harness_oracle_reproduces_the_reference_run(name, B):
    """Forced, then free-running under one seed: the oracle fed the replayed tape returns the reference's waveform and head outputs."""
    _ensure_built()
    from oracle.wavenet_oracle import Oracle
    from tests._golden import oracle_config
    from tests._refrun import reference_case
    T = 256
    d = reference_case(name, B, 24, T, seed=21, threads=4)
    o = Oracle(oracle_config(d["kw"]), d["model"].state_dict())
    got, gparams = o.incremental_forward(test_inputs=d["x"], c=d["c"], g=d["gids"], T=T, noise=d["tape"], return_params=True)
    assert d["wparams"].shape == gparams.shape
    assert float((gparams - d["wparams"]).abs().max()) <= 1e-6
    assert torch.equal(got, d["want"])


def test_bench_cpu_leg_times_the_reference_itself():
    _ensure_built()
    import bench
    from tests._configs import CONFIGS, build, inputs
    name = "cfg2_mol"
    c, g = inputs(name, 8, 24064)
    r = bench.cpu_baseline(build(name), CONFIGS[name], c, 256, budget_s=1.0, B=8, gids=g)
    assert r["kind"] == "reference" and r["value"] > 0 and set(r["all_threads_kSamples_s"]) >= {"1", "4"}


def test_public_signatures_of_the_host_mirrors_equal_the_references():
    """The drop-in claim, mechanically: every public callable of the reference's hot-path modules that this package mirrors has the
    reference's own signature -- parameter names, order and defaults (a keyword call written against the reference keeps working).
    What the package does not mirror is exactly the training side (losses) and one factory no model of the reference uses."""
    import importlib
    import inspect
    R = _ensure_built()
    ref = R.load_reference()
    import wavenet_vocoder_amd as amd

    def sig(f):
        # (defaults that are functions -- the tqdm hook's identity lambda -- compare by kind, not by address)
        return [(p.name, p.kind, "<callable>" if callable(p.default) and p.default is not inspect.Parameter.empty else p.default)
                for p in inspect.signature(f).parameters.values()]

    for name in ("__init__", "forward", "incremental_forward", "clear_buffer", "make_generation_fast_", "has_speaker_embedding",
                 "local_conditioning_enabled"):
        assert sig(getattr(ref.WaveNet, name)) == sig(getattr(amd.WaveNet, name)), name
    assert sig(ref.receptive_field_size) == sig(amd.receptive_field_size)
    not_mirrored = {"modules": {"ConvTranspose2d"},                                # (imported by wavenet.py:13, used by no model)
                    "mixture": {"discretized_mix_logistic_loss", "mix_gaussian_loss", "log_sum_exp"}}          # training losses
    checked = 0
    for mod in ("modules", "mixture", "upsample", "util", "conv"):
        rm = importlib.import_module(ref.__name__ + "." + mod)
        am = importlib.import_module("wavenet_vocoder_amd." + mod)
        for n, obj in inspect.getmembers(rm, lambda o: inspect.isfunction(o) or inspect.isclass(o)):
            if n.startswith("_") or getattr(obj, "__module__", None) != rm.__name__:
                continue
            if n in not_mirrored.get(mod, ()):
                assert not hasattr(am, n)
                continue
            assert hasattr(am, n), f"{mod}.{n} is missing"
            mine = getattr(am, n)
            assert sig(obj) == sig(mine), f"{mod}.{n}: {inspect.signature(obj)} != {inspect.signature(mine)}"
            checked += 1
            if inspect.isclass(obj):
                for meth in ("forward", "incremental_forward", "clear_buffer"):
                    if hasattr(obj, meth) and meth in vars(obj):
                        assert hasattr(mine, meth), f"{mod}.{n}.{meth}"
                        assert sig(getattr(obj, meth)) == sig(getattr(mine, meth)), f"{mod}.{n}.{meth}"
                        checked += 1
    assert checked >= 20, checked
