"""CPU: the launch path of bench.py.  `python bench.py --gpus N` must start its own N ranks when no launcher set WORLD_SIZE
(that is how the driver's N = 1 command looks with a larger N), must keep working under torch.distributed.run, and must print
exactly ONE JSON line (from rank 0) carrying n_gpus = N.  `--dry-run` swaps the engine for a no-op and RCCL for gloo; the
rendezvous, the barriers, the MAX-reduce of the elapsed time and the printing are the code the GPU run uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(cmd, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.parametrize("n", [1, 2, 3])
def test_self_launch_prints_one_line_with_n_gpus(n):
    r, lines = run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["dry_run"] is True
    # ranks sleep 2 ms x (rank + 1) per step: the reported time is the slowest rank's (MAX over ranks), not rank 0's
    assert j["ms_per_step"] >= 2.0 * n * 0.9
    # what every rank ran travels in the line too (a rank that fell back to the generic kernel must not hide inside the MAX):
    # per-rank kernel time and the kernel that served each rank, gathered by the same collective path as the real run
    rk = j["ranks"]
    assert rk["last_kernel_by_rank"] == ["ring"] * n
    assert rk["kernel_ms_by_rank"] == [2.0 * (r + 1) for r in range(n)]
    assert rk["kernel_ms_min"] == 2.0 and rk["kernel_ms_max"] == 2.0 * n


def test_under_torch_distributed_run():
    r, lines = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29613", BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_world_size_mismatch_is_an_error():
    r, lines = run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and not lines


def test_a_dying_rank_takes_the_job_down():
    """Rank 1 refuses to start (bad LOCAL_RANK handling is simulated by an impossible --steps): nobody is left waiting."""
    r, lines = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "not-a-number"])
    assert r.returncode != 0 and not lines


def test_cpu_baseline_takes_an_unconditioned_workload():
    """BASELINE configs[0] (mu-law, 8 layers, no conditioning) through bench.py's bounded CPU leg: no mel to slice or upsample.  With
    oracle/_ref built the leg times the REAL reference at the bench's batch (one forced first step tells it B, wavenet.py:253); without
    it the oracle port runs one utterance and says so (`batch`), so that speedup_vs_cpu_baseline compares like with like."""
    import importlib
    import torch
    from oracle import reference as R
    from tests._configs import CONFIGS, build, inputs
    bench = importlib.import_module("bench")
    name = "cfg0_mulaw256_small"
    c, g = inputs(name, 8, 2048)
    assert c is None and g is None
    threads = torch.get_num_threads()
    try:
        r = bench.cpu_baseline(build(name, seed=0), CONFIGS[name], c, 64, budget_s=1.0, B=8)
        assert r["value"] > 0 and r["unit"] == "kSamples/s"
        if R.available():
            assert r["kind"] == "reference" and "B=8" in r["sample"]
        # the port leg (what a tree without oracle/_ref reports)
        saved, R.available = R.available, (lambda: False)
        try:
            r = bench.cpu_baseline(build(name, seed=0), CONFIGS[name], c, 64, budget_s=1.0, B=8)
        finally:
            R.available = saved
        assert r["kind"] == "port" and r["batch"] == 1 and "B=1" in r["sample"] and r["value"] > 0
    finally:
        torch.set_num_threads(threads)


@pytest.mark.parametrize("workload,n_utt,packed", [("cfg3b_gaussian30", 64, True), ("cfg4_mol_multispeaker", 128, True), ("cfg4_mol_multispeaker", 128, False)])
def test_job_dry_run_of_the_multi_gpu_baseline_jobs(workload, n_utt, packed):
    """BASELINE.json configs[3] (64 utterances over 8 GPUs) and configs[4] (128) as `--job` lines on 1 / 2 / 4 / 8 gloo ranks: the
    scheduler's decisions (lpt_assign over the ranks, packed slots or padded groups per rank), gathered by the collectives the GPU run
    uses -- strong scaling: the same job on every world size, load imbalance and padding loss reported."""
    true = None
    for n in (1, 2, 4, 8):
        cmd = [sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--job", str(n_utt), "--workload", workload, "--steps", "1"]
        r, lines = run(cmd + (["--packed"] if packed else []))
        assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
        j = json.loads(lines[0])
        job = j["job"]
        assert j["n_gpus"] == n and j["scaling"] == "strong" and job["utterances"] == n_utt and len(job["per_rank_true_padded_launches"]) == n
        true = true or job["true_samples"]
        assert job["true_samples"] == true == sum(r_[0] for r_ in job["per_rank_true_padded_launches"])      # the same job whatever the world size
        assert all(r_[0] > 0 and r_[1] >= r_[0] for r_ in job["per_rank_true_padded_launches"])
        tr = [r_[0] for r_ in job["per_rank_true_padded_launches"]]
        assert max(tr) / (sum(tr) / n) < 1.02 and job["load_imbalance"] < 1.15, job                        # longest-first keeps the ranks level
        if packed:
            slots = 32 if "cfg4" in workload else 48
            assert all(r_[2] == 1 for r_ in job["per_rank_true_padded_launches"]) and f"x{slots}" in job["scheduler"], job
            # (a rank with fewer utterances than slots gives every utterance a slot of its own: its launch runs as long as its longest
            #  utterance -- idle slots cost no time, the step is the chain's latency --, so "padding" only measures loss once slots are shared)
            if n_utt // n >= 2 * slots:
                assert job["padding_loss"] < 0.15, job
