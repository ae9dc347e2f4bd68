"""-m gpu: the device post-chain (wnv_postprocess) against the CPU oracle, and batch_wavegen end to
end against the oracle's incremental_forward + post-chain."""
import os

import numpy as np
import pytest
import torch

from oracle import postchain_oracle as P
from wavenet_vocoder_amd import synthesis

pytestmark = pytest.mark.gpu


def hp(**kw):
    return synthesis.default_hparams(**kw)


@pytest.mark.parametrize("T", [1, 255, 256, 257, 24064])
def test_raw_preemphasis_gain(T):
    g = torch.Generator().manual_seed(T)
    y = (torch.rand(3, 1, T, generator=g) - 0.5).cuda()
    want = P.post_chain(y.cpu().numpy(), "raw", postprocess="inv_preemphasis", coef=0.85, global_gain_scale=0.55)
    got = synthesis.postprocess(y, hp()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)


def test_no_postprocess_is_identity():
    y = (torch.rand(2, 1, 1000) - 0.5).cuda()
    got = synthesis.postprocess(y, hp(postprocess=None, global_gain_scale=0.0))
    assert torch.equal(got, y[:, 0])


def test_mulaw_quantize_argmax_decode():
    g = torch.Generator().manual_seed(3)
    B, Cq, T = 2, 256, 1500
    idx = torch.randint(0, Cq, (B, T), generator=g)
    y = torch.nn.functional.one_hot(idx, Cq).permute(0, 2, 1).float().contiguous().cuda()
    got = synthesis.postprocess(y, hp(input_type="mulaw-quantize", quantize_channels=256, postprocess=None,
                                      global_gain_scale=0.0)).cpu().numpy()
    np.testing.assert_allclose(got, P.inv_mulaw_quantize(idx.numpy(), 255), rtol=2e-5, atol=1e-6)


def test_mulaw_scalar_decode_and_int16():
    y = (torch.rand(2, 1, 4000) * 2 - 1).cuda()
    h = hp(input_type="mulaw", quantize_channels=256, postprocess="inv_preemphasis", global_gain_scale=0.9)
    wav, pcm = synthesis.postprocess(y, h, want_int16=True)
    want = P.post_chain(y.cpu().numpy(), "mulaw", quantize_channels=256, postprocess="inv_preemphasis", coef=0.85,
                        global_gain_scale=0.9, clip=True)
    np.testing.assert_allclose(wav.cpu().numpy(), want, rtol=3e-5, atol=3e-6)
    ref16 = P.to_int16(want)
    assert np.abs(pcm.cpu().numpy().astype(np.int32) - ref16.astype(np.int32)).max() <= 1      # truncation at a float ulp


def test_batch_wavegen_end_to_end():
    from oracle.wavenet_oracle import Oracle
    from tests._configs import CONFIGS, build, inputs
    from tests._golden import oracle_config
    name, B, T = "cfg2_mol", 2, 512
    m = build(name)
    o = Oracle(oracle_config(CONFIGS[name]), m.state_dict())
    c, _ = inputs(name, B, T)
    h = hp(cin_pad=CONFIGS[name]["cin_pad"], hop_size=256)
    wav = synthesis.batch_wavegen(m.to("cuda"), c=c, g=None, hparams=h)
    assert wav.shape == (B, T) and wav.dtype == np.float32 and np.isfinite(wav).all()
    # same chain on the CPU oracle for the teacher-free run is chaotic; check the post-chain on the engine's own samples
    with torch.no_grad():
        y_hat = m.incremental_forward(c=c.cuda(), T=T, softmax=True, quantize=True)
    assert y_hat.shape == (B, 1, T)
    want = P.post_chain(y_hat.cpu().numpy(), "raw", postprocess="inv_preemphasis", coef=0.85, global_gain_scale=0.55)
    assert want.shape == wav.shape
    del o


def test_wavegen_single_utterance():
    """synthesis.wavegen (one utterance, (Tc, cin) numpy features, no context frames): same samples as incremental_forward on
    the same inputs + the post-chain, and the documented quirk (mu = quantize_channels for the mu-law decoders)."""
    import wavenet_vocoder_amd as wnv
    from tests._configs import tame_head_
    kw = dict(out_channels=30, layers=6, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128, kernel_size=3,
              dropout=0.0, scalar_input=True, output_distribution="Logistic", cin_channels=80, cin_pad=0,
              upsample_conditional_features=True, upsample_params=dict(upsample_scales=[4, 4, 4, 4], cin_channels=80, cin_pad=0))
    torch.manual_seed(3)
    m = tame_head_(wnv.WaveNet(**kw).eval()).to("cuda")
    feats = np.random.default_rng(0).standard_normal((3, 80)).astype(np.float32)            # (Tc, D)
    h = hp(cin_pad=0, hop_size=256)
    torch.manual_seed(11)
    wav = synthesis.wavegen(m, c=feats, hparams=h, fast=True)
    assert wav.shape == (3 * 256,) and wav.dtype == np.float32 and np.isfinite(wav).all()
    torch.manual_seed(11)                                                                    # same in-kernel noise seed
    with torch.no_grad():
        y_hat = m.incremental_forward(torch.zeros(1, 1, 1).cuda(), c=torch.from_numpy(feats.T).unsqueeze(0).cuda(), T=768,
                                      softmax=True, quantize=True)
    want = P.post_chain(y_hat.cpu().numpy(), "raw", postprocess="inv_preemphasis", coef=0.85, global_gain_scale=0.55)
    np.testing.assert_allclose(wav, want.reshape(-1), rtol=2e-5, atol=2e-6)
    # mu-law quirk: wavegen decodes with mu = quantize_channels, batch_wavegen with quantize_channels - 1
    y = (torch.rand(1, 1, 500) * 2 - 1).cuda()
    hq = hp(input_type="mulaw", quantize_channels=256, postprocess=None, global_gain_scale=0.0)
    np.testing.assert_allclose(synthesis.postprocess(y, hq, mu=256).cpu().numpy(), P.inv_mulaw(y.cpu().numpy()[:, 0], 256), rtol=3e-5, atol=1e-6)


def test_evaluate_directory_loop_end_to_end(tmp_path):
    """evaluate.py's main loop on the real engine: ragged `*-feats.npy` files -> padded groups -> ring kernel -> post-chain ->
    clipped int16 wav files of the right lengths; deterministic under a fixed torch seed; group members independent of padding."""
    from scipy.io import wavfile
    from types import SimpleNamespace
    from tests._configs import build
    from wavenet_vocoder_amd import evaluate as E
    rng = np.random.default_rng(0)
    frames = [9, 5, 12, 7, 5]
    for i, f in enumerate(frames):
        np.save(tmp_path / f"utt{i:02d}-feats.npy", rng.standard_normal((f + 4, 80)).astype(np.float32))   # incl. 2 * cin_pad context
    m = build("cfg2_mol").to("cuda")
    h = hp(cin_channels=80, cin_pad=2, hop_size=256, batch_size=3, sample_rate=24000)
    outs = []
    for run in range(2):
        torch.manual_seed(123)
        paths = E.synthesize_dir(m, str(tmp_path), str(tmp_path / f"out{run}"), h)
        outs.append([wavfile.read(p) for p in paths])
    for i, (rate, w) in enumerate(outs[0]):
        assert rate == 24000 and w.dtype == np.int16 and len(w) == (frames[i] + 4) * 256
        assert np.abs(w.astype(np.int32)).max() <= 32767 and w.std() > 0
        assert np.array_equal(w, outs[1][i][1]), "same seed, same files"


def test_evaluate_directory_loop_packed_slots(tmp_path, capsys):
    """The same loop with nothing fixing the grouping: the job runs as PACKED SLOTS (continuous batching) -- right lengths, clipped
    int16, deterministic under a fixed torch seed, the post-chain applied per utterance (an utterance's file does not depend on what
    shared its slot: the job of five and the job of its first three -- same longest member, so the same zero-padded conditioning
    batch, evaluate.py:55-57 -- give the same first three files)."""
    from scipy.io import wavfile
    from tests._configs import build
    from wavenet_vocoder_amd import evaluate as E
    rng = np.random.default_rng(1)
    frames = [9, 5, 12, 7, 5]
    for i, f in enumerate(frames):
        np.save(tmp_path / f"utt{i:02d}-feats.npy", rng.standard_normal((f + 4, 80)).astype(np.float32))
    m = build("cfg2_mol").to("cuda")
    h = hp(cin_channels=80, cin_pad=2, hop_size=256, batch_size=None, sample_rate=24000)
    outs = []
    for run in range(2):
        torch.manual_seed(321)
        paths = E.synthesize_dir(m, str(tmp_path), str(tmp_path / f"out{run}"), h, packed=True)
        outs.append([wavfile.read(p)[1] for p in paths])
    assert "falling back" not in capsys.readouterr().out
    for i, w in enumerate(outs[0]):
        assert w.dtype == np.int16 and len(w) == (frames[i] + 4) * 256 and w.std() > 0
        assert np.array_equal(w, outs[1][i])
    torch.manual_seed(321)
    part = [wavfile.read(p)[1] for p in E.synthesize_dir(m, str(tmp_path), str(tmp_path / "out_part"), h, num_utterances=3, packed=True)]
    for i in range(3):
        assert np.array_equal(part[i], outs[0][i]), "an utterance's waveform depends on its own conditioning, id and the seed only"


@pytest.mark.parametrize("name", ["cfg1_mulaw256", "cfg4_mol_multispeaker"])
def test_evaluate_packed_directory_loop_of_one_hot_and_speaker_models(name, tmp_path, capsys):
    """Round 5: the directory loop as packed slots for a mu-law model (the launch carries the sampled CLASSES, the post-chain decodes them:
    synthesize_packed(as_index=True) + sink) and for a speaker-conditioned model (speaker ids from train.txt, one bias row per speaker).
    The files equal what sharding.synthesize_packed + synthesis.postprocess give for the same seed by the plain route."""
    from scipy.io import wavfile
    from tests._configs import CONFIGS, build
    from wavenet_vocoder_amd import evaluate as E, sharding
    kw = CONFIGS[name]
    rng = np.random.default_rng(2)
    frames = [6, 3, 8, 4, 5, 3]
    spk = [3, 0, 6, 6, 1, 2]
    multi = kw.get("gin_channels", -1) > 0
    lines = []
    for i, f in enumerate(frames):
        np.save(tmp_path / f"utt{i:02d}-feats.npy", rng.standard_normal((f, 80)).astype(np.float32))
        lines.append(f"utt{i:02d}-wave.npy|utt{i:02d}-feats.npy|{f}|text" + (f"|{spk[i]}" if multi else ""))
    (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
    m = build(name).to("cuda")
    onehot = not kw.get("scalar_input", False)
    h = hp(cin_channels=80, cin_pad=2, hop_size=256, batch_size=None, sample_rate=24000,
           **({"input_type": "mulaw-quantize", "quantize_channels": 256} if onehot else {}))
    torch.manual_seed(77)
    paths = E.synthesize_dir(m, str(tmp_path), str(tmp_path / "out"), h, packed=True)
    assert "falling back" not in capsys.readouterr().out
    got = [wavfile.read(p)[1] for p in paths]
    if multi:
        assert all(os.path.basename(p).startswith(f"speaker{s}_") for p, s in zip(paths, spk))
    # the plain route with the same seed: one-hot / scalar network outputs, then the post-chain per utterance
    torch.manual_seed(77)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    mels = [torch.from_numpy(np.load(tmp_path / f"utt{i:02d}-feats.npy").T.copy()) for i in range(len(frames))]
    outs = sharding.synthesize_packed(m, mels, hop_size=256, cin_pad=2, seed=seed, speaker_ids=spk if multi else None)
    for i, (w, y) in enumerate(zip(got, outs)):
        want = synthesis.postprocess(y.unsqueeze(0), h, want_int16=True)[1][0].cpu().numpy()
        assert w.dtype == np.int16 and len(w) == frames[i] * 256 and w.std() > 0
        assert np.array_equal(w, want), f"utterance {i}"
    m.to("cpu")
