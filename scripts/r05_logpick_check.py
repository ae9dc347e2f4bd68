"""How often does the log-domain pick of the one-hot THROUGHPUT instantiation (ring kernel, MODE 1) differ from the quotient form of the generic
kernel on the same logits and the same noise tape, and at what top-2 margins?  Teacher-forced (no divergence): every step is an independent draw."""
import sys, torch
sys.path.insert(0, ".")
from tests._configs import CONFIGS, build, inputs
from tests._margins import choice_margin
from wavenet_vocoder_amd.noise import make_noise_tape
name, B, T = "cfg1_mulaw256", 40, 4096
kw = CONFIGS[name]
m = build(name).to("cuda"); eng = m._get_engine()
c, _ = inputs(name, B, T)
g = torch.Generator().manual_seed(5)
idx = torch.randint(0, 256, (B, T), generator=g)
x = torch.zeros(B, T, 256).scatter_(2, idx.unsqueeze(2), 1.0).cuda()
tape = make_noise_tape(T, B, scalar_input=False, output_distribution="Logistic", out_channels=256, generator=torch.Generator().manual_seed(6))
c_up = eng.upsample(c.cuda(), T_expected=T)
res = {}
for k in (1, 2):
    out, params, index = eng.generate(B=B, T=T, c_up=c_up, teacher=x, noise=tape.cuda(), want_params=True, want_index=True, kernel=k)
    res[k] = (params.cpu(), index.cpu())
pa, ia = res[1]; pb, ib = res[2]
print("head outputs ring vs generic max diff", float((pa - pb).abs().max()))
diff = ia != ib
print("samples", B * T, "picks that differ", int(diff.sum()))
margin, _ = choice_margin(pa, tape, kw)
if diff.any():
    print("top-2 margins at the differing picks:", sorted(float(v) for v in margin[diff])[:20])
print("fraction of all samples with margin < 1e-5:", float((margin < 1e-5).double().mean()))
