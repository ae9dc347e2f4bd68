import sys, time, torch
sys.path.insert(0, "/root/repo")
from tests._configs import build, inputs
name, B, T = "cfg2_mol", 8, 256
m = build(name).to("cuda"); eng = m._get_engine()
c, _ = inputs(name, B, T); c_up = eng.upsample(c.cuda(), T_expected=T)
hog = build(name).to("cuda")._get_engine()
for HB in (240, 256):
    HT = 4096
    hc, _ = inputs(name, 1, HT)
    hc_up = hog.upsample(hc.cuda(), T_expected=HT).expand(HB, -1, -1).contiguous()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    t0 = time.time()
    with torch.cuda.stream(side):
        hog.generate(B=HB, T=HT, c_up=hc_up, seed=3, kernel=1)
    t1 = time.time()
    time.sleep(0.05)
    out, _, _ = eng.generate(B=B, T=T, c_up=c_up, seed=11, kernel=0)
    t2 = time.time()
    k = eng.last_kernel()
    torch.cuda.synchronize()
    t3 = time.time()
    print(f"HB={HB}: hog launch call {t1-t0:.3f}s, auto call {t2-t1:.3f}s (kernel {k}), hog+all done after {t3-t0:.3f}s", flush=True)
    eng.reset()
