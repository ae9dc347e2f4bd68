"""CPU: the post-chain oracle (oracle/postchain_oracle.py) against the closed-form properties that pin it -- nnmnkwii,
where the reference gets these functions from, is not installed and the reference's tests hold no vectors for them
(parity unpinned)."""
import numpy as np
import pytest

from oracle import postchain_oracle as P


@pytest.mark.parametrize("mu", [255, 65535])
def test_mulaw_is_inverted_by_inv_mulaw(mu):
    x = np.linspace(-1, 1, 4001)
    np.testing.assert_allclose(P.inv_mulaw(P.mulaw(x, mu), mu), x, atol=1e-12)
    assert P.inv_mulaw(0.0, mu) == 0.0 and abs(P.inv_mulaw(1.0, mu) - 1.0) < 1e-12 and abs(P.inv_mulaw(-1.0, mu) + 1.0) < 1e-12


def test_mulaw_quantize_codes_round_trip():
    mu = 255
    codes = np.arange(mu + 1)
    x = P.inv_mulaw_quantize(codes, mu)
    assert x[0] == -1.0 and abs(x[-1] - 1.0) < 1e-12 and np.all(np.diff(x) > 0)
    # re-quantising a value decoded from inside a code's cell gives the code back (the quantiser truncates)
    back = P.mulaw_quantize(P.inv_mulaw_quantize(codes[:-1] + 0.25, mu), mu)
    assert np.array_equal(back, codes[:-1])
    # mid code decodes to (almost) silence
    assert abs(P.inv_mulaw_quantize(np.array([127.5]), mu)[0]) < 1e-12


def test_inv_preemphasis_inverts_preemphasis():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(5000)
    np.testing.assert_allclose(P.inv_preemphasis(P.preemphasis(x, 0.85), 0.85), x, atol=1e-9)
    # impulse response of the one-pole filter
    imp = np.zeros(8); imp[0] = 1
    np.testing.assert_allclose(P.inv_preemphasis(imp, 0.85), 0.85 ** np.arange(8), atol=1e-15)


def test_to_int16_truncates_like_the_reference():
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 3.05e-5], dtype=np.float32)
    assert P.to_int16(x).tolist() == [0, 32767, -32767, 16383, -16383, 0]
    with pytest.raises(AssertionError):
        P.to_int16(np.array([1.5], dtype=np.float32))


def test_post_chain_modes():
    rng = np.random.default_rng(1)
    B, T = 2, 300
    raw = rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32)
    y = P.post_chain(raw, "raw", postprocess=None, global_gain_scale=0.0)
    np.testing.assert_array_equal(y, raw.reshape(B, T))
    y = P.post_chain(raw, "raw", postprocess="inv_preemphasis", coef=0.85, global_gain_scale=0.55)
    np.testing.assert_allclose(y, np.stack([P.inv_preemphasis(raw[i, 0]) for i in range(B)]) / 0.55, rtol=1e-6)
    idx = rng.integers(0, 256, (B, T))
    onehot = np.zeros((B, 256, T), dtype=np.float32)
    for b in range(B):
        onehot[b, idx[b], np.arange(T)] = 1
    y = P.post_chain(onehot, "mulaw-quantize", quantize_channels=256, postprocess=None, global_gain_scale=0.0)
    np.testing.assert_allclose(y, P.inv_mulaw_quantize(idx, 255), rtol=1e-6)
    y = P.post_chain(raw * 2, "mulaw", quantize_channels=256, postprocess=None, global_gain_scale=0.0)
    np.testing.assert_allclose(y, P.inv_mulaw(raw.reshape(B, T) * 2, 255), rtol=1e-6)


def test_wavegen_argument_errors_need_no_gpu():
    """synthesis.wavegen's checks (sanity_check, 2-dim features, missing length) fire before anything touches a device."""
    import pytest
    import torch
    import wavenet_vocoder_amd as wnv
    from wavenet_vocoder_amd import synthesis
    local = wnv.WaveNet(out_channels=30, layers=2, stacks=1, residual_channels=8, gate_channels=16, skip_out_channels=8,
                        cin_channels=4, scalar_input=True).eval()
    plain = wnv.WaveNet(out_channels=30, layers=2, stacks=1, residual_channels=8, gate_channels=16, skip_out_channels=8,
                        scalar_input=True).eval()
    with pytest.raises(RuntimeError, match="conditional features, but not given"):
        synthesis.wavegen(local, length=10)
    with pytest.raises(RuntimeError, match="no conditional features, but given"):
        synthesis.wavegen(plain, c=np.zeros((3, 4), np.float32))
    with pytest.raises(RuntimeError, match="no speaker embedding"):
        synthesis.wavegen(plain, length=4, g=1)
    with pytest.raises(RuntimeError, match="Expected 2-dim shape"):
        synthesis.wavegen(local, c=np.zeros((1, 3, 4), np.float32)[:, :, :, None])
    with pytest.raises(AssertionError):
        synthesis.wavegen(plain)                                     # neither length nor features
    assert synthesis._to_numpy(torch.zeros(1, 3, 4)).shape == (3, 4)
