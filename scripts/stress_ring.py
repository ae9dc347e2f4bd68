"""Stress the ring kernel: long utterances, every batch size 1..17, many back-to-back launches (launch-unique tags)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests._configs import build
m = build("cfg2_mol").to("cuda")
eng = m._get_engine()
def cond(B, T, seed=0):
    return torch.randn(B, T, 80, generator=torch.Generator().manual_seed(seed)).cuda()
t0 = time.time()
for B in list(range(1, 18)) + [24, 32, 40, 47, 48, 56, 64, 72]:      # (round 6: the throughput instantiation's batch sizes too; 72 = two launches)
    T = 1500 + 13 * B
    out, _, _ = eng.generate(B=B, T=T, c_up=cond(B, T, B), seed=B, kernel=2)
    ref, _, _ = eng.generate(B=B, T=T, c_up=cond(B, T, B), seed=B, kernel=2)
    assert torch.equal(out, ref) and torch.isfinite(out).all(), B
print(f"batch sizes 1..17, 24, 32, 40, 47, 48, 56, 64, 72: deterministic, finite ({time.time()-t0:.1f} s)")
t0 = time.time()
c = cond(8, 256)
first = None
for i in range(300):
    out, _, _ = eng.generate(B=8, T=256, c_up=c, seed=7, kernel=2)
    if first is None: first = out.clone()
    assert torch.equal(out, first), i
print(f"300 back-to-back launches identical ({time.time()-t0:.1f} s)")
t0 = time.time()
T = 240640
out, _, _ = eng.generate(B=8, T=T, c_up=cond(8, T, 3), seed=3, kernel=2)
torch.cuda.synchronize()
dt = time.time() - t0
assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
print(f"T = {T} (10 s of audio x 8): {dt:.2f} s wall incl. input generation, std {float(out.std()):.3f}")
pre, _, _ = eng.generate(B=8, T=4096, c_up=cond(8, T, 3)[:, :4096].contiguous(), seed=3, kernel=2)
assert torch.equal(pre, out[:, :, :4096]), "prefix property at length"
print("prefix property holds")
# the wide-skip instantiations (K = 256 one-hot with two head parts, K = 512 with four): repeated launches, determinism
for name, Bs in (("cfg1_mulaw256", (1, 8)), ("cfg4_mol_multispeaker", (8, 16))):
    from tests._configs import inputs
    mm = build(name).to("cuda")
    ee = mm._get_engine()
    t0 = time.time()
    for B in Bs:
        T = 2048
        c, gids = inputs(name, B, T)
        c_up = ee.upsample(c.cuda(), T_expected=T)
        gi = None if gids is None else gids[:, 0].cuda()
        first = None
        for i in range(20):
            out, _, _ = ee.generate(B=B, T=T, c_up=c_up, g_ids=gi, seed=5, kernel=2)
            if first is None: first = out.clone()
            assert torch.equal(out, first) and torch.isfinite(out).all(), (name, B, i)
    print(f"{name}: 20 launches each at B = {Bs}: identical ({time.time()-t0:.1f} s)")
# more utterances per ring than round 1 exercised, and the group-ring kernel for wide models
t0 = time.time()
for B in (40, 48, 56, 64):
    T = 1024
    out, _, _ = eng.generate(B=B, T=T, c_up=cond(B, T, B), seed=B, kernel=2)
    ref, _, _ = eng.generate(B=B, T=T, c_up=cond(B, T, B), seed=B, kernel=2)
    assert torch.equal(out, ref) and torch.isfinite(out).all(), B
print(f"batch sizes 40..64: deterministic, finite ({time.time()-t0:.1f} s)")
from tests._configs import inputs
wm = build("wide_mol_512").to("cuda")
we = wm._get_engine()
t0 = time.time()
WIDE_BS = (1, 2, 3, 5, 8, 9, 13, 16, 17, 33, 37)      # 16 utterances per launch: deferred history copies, host slices above 16
for B in WIDE_BS:
    T = 1024 if B <= 16 else 512
    c, _ = inputs("wide_mol_512", B, T)
    c_up = we.upsample(c.cuda(), T_expected=T)
    first = None
    for i in range(10 if B <= 8 else 4):
        out, _, _ = we.generate(B=B, T=T, c_up=c_up, seed=5, kernel=3)
        if first is None: first = out.clone()
        assert torch.equal(out, first) and torch.isfinite(out).all(), ("wide", B, i)
print(f"wide_mol_512 (group ring): 10 / 4 launches each at B = {WIDE_BS}: identical ({time.time()-t0:.1f} s)")
t0 = time.time()
T = 48000
c, _ = inputs("wide_mol_512", 1, 48128)
out, _, _ = we.generate(B=1, T=48128, c_up=we.upsample(c.cuda(), T_expected=48128), seed=3, kernel=3)
torch.cuda.synchronize()
print(f"wide_mol_512: 48128 samples (3 s at 16 kHz) in {time.time()-t0:.2f} s, std {float(out.std()):.3f}")
