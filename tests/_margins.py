"""When may a sample differ from the oracle's?  Only where the sampler's DISCRETE choice sat on a near tie.

Every sampler of the path makes at most one discrete choice per step -- the Gumbel-max component pick of the mixture
samplers (mixture.py:138-143, :245-253) or torch.multinomial's argmax(p_hat / e) (wavenet.py:334-335) -- and is a smooth
function of the head outputs otherwise.  Head outputs agree to ~1e-6, so a sample may legitimately differ by more than the
tolerance only at a step whose top-2 margin (in the oracle's own numbers, under the shared noise tape) is smaller than the
difference in the head outputs could bridge.  These helpers compute that margin and turn "98 % of the samples agree" into
"every disagreement is a near tie"."""
import torch


def n_mix(kw):
    if not kw.get("scalar_input", False):
        return 0
    C = kw["out_channels"]
    if kw.get("output_distribution", "Logistic") == "Logistic":
        return C // 3
    return 0 if C in (2, 3) else C // 3


def choice_scores(params, tape, kw, softmax=True):
    """Scores whose argmax over dim 1 is the sampler's discrete choice: params (B, O, T) head outputs, tape (T, B, NZ).
    Returns (B, n_choices, T), or None when the distribution has no discrete choice (single Gaussian)."""
    tape_bt = tape.permute(1, 2, 0)                                   # (B, NZ, T)
    if kw.get("scalar_input", False):
        k = n_mix(kw)
        if k == 0:
            return None
        return params[:, :k].double() - torch.log(-torch.log(tape_bt[:, :k].double()))           # logit + Gumbel
    logp = torch.log_softmax(params.double(), dim=1) if softmax else torch.log(params.double() / params.double().sum(1, keepdim=True))
    return logp - torch.log(tape_bt.double())                                                   # log(p_hat / e)


def choice_margin(params, tape, kw, softmax=True):
    """(B, T) top-2 margin of the discrete choice (+inf when there is none) and the chosen index (B, T)."""
    s = choice_scores(params, tape, kw, softmax)
    if s is None or s.shape[1] < 2:                                   # no discrete choice (a single Gaussian, a one-component mixture)
        B, _, T = params.shape
        return torch.full((B, T), float("inf"), dtype=torch.float64), torch.zeros(B, T, dtype=torch.long)
    top = s.topk(2, dim=1)
    return top.values[:, 0] - top.values[:, 1], top.indices[:, 0]


def assert_match_or_near_tie(got, want, params_want, tape, kw, tol=1e-4, tie=1e-5, what="samples"):
    """got / want: (B, 1, T) scalar samples or (B, T) class indices produced from (nearly) the same head outputs under the
    same tape.  Every position that differs (by >= tol for scalars) must be a near tie of the oracle's discrete choice."""
    margin, _ = choice_margin(params_want, tape, kw)
    if got.dim() == 3:
        bad = (got[:, 0].double() - want[:, 0].double()).abs() >= tol
    else:
        bad = got.long() != want.long()
    n_bad = int(bad.sum())
    if n_bad:
        worst = float(margin[bad].max())
        assert worst < tie, (f"{what}: {n_bad} positions differ and the largest top-2 margin among them is {worst:.3e} "
                             f"(a legitimate flip needs a near tie, < {tie:g})")
    return n_bad


def assert_free_run_agrees_until_near_tie(got, want, params_got, params_want, tape, kw, t0=0, tol=1e-3, what="free run"):
    """Free-running trajectories (from step t0 on) may part only through a flipped discrete choice at a near tie.
    got / want: (B, 1, T) samples or (B, T) class indices; params_*: (B, O, T) head outputs of the two runs.  Per utterance:
    up to the first step whose discrete choice differs, samples agree to `tol` (scalars) / exactly (classes); at that step the
    oracle's top-2 margin is no larger than what the head-output difference of that very step can bridge.  Returns the number of
    agreeing steps per utterance."""
    B, T = got.shape[0], got.shape[-1]
    m_want, c_want = choice_margin(params_want, tape, kw)
    _, c_got = choice_margin(params_got, tape, kw)
    scalar = got.dim() == 3
    horizon = []
    for b in range(B):
        flips = (c_got[b, t0:] != c_want[b, t0:]).nonzero()
        s = T if flips.numel() == 0 else t0 + int(flips[0])
        if scalar:
            d = (got[b, 0, t0:s].double() - want[b, 0, t0:s].double()).abs()
            assert d.numel() == 0 or float(d.max()) < tol, f"{what}: utterance {b} drifts to {float(d.max()):.2e} before any discrete choice differs (step {t0 + int(d.argmax())})"
        else:
            assert torch.equal(got[b, t0:s].long(), want[b, t0:s].long()), f"{what}: utterance {b}: classes differ before any near tie"
        if s < T:
            gap = 2.0 * float((params_got[b, :, s].double() - params_want[b, :, s].double()).abs().max()) + 1e-6
            assert float(m_want[b, s]) <= gap, (f"{what}: utterance {b} flips its choice at step {s} where the oracle's top-2 margin is "
                                               f"{float(m_want[b, s]):.3e} but the head outputs differ by only {gap / 2:.3e}")
        horizon.append(s)
    return horizon
