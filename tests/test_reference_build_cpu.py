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
