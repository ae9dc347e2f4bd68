"""The engine-backed ``incremental_forward`` as a MIXIN, and the reference-side graft built from it.

``EngineHost`` holds everything that turns a WaveNet module tree into calls on the C ABI (include/wnv.h): building the
``wnv_config`` from the module, handing the ``state_dict`` to ``wnv_load_weights`` (weight-normed or fused), caching the packed
weights, and the argument handling of ``incremental_forward`` (reference wavenet.py:215-343).  It assumes only the attribute and
sub-module names of the reference class (wavenet.py:98-156), so it serves two hosts:

  * ``wavenet_vocoder_amd.WaveNet`` -- this package's stand-alone module (``class WaveNet(EngineHost, nn.Module)``);
  * ``make_wavenet_amd(wavenet_vocoder.WaveNet)`` -- the graft a maintainer of the reference adds:
    a subclass of the REFERENCE's own class whose constructor, parameters, ``state_dict``, ``forward`` and training code stay
    the reference's, while ``incremental_forward`` runs in libwnv_hip.so.  ``synthesis.py`` / ``evaluate.py`` / ``train.eval_model``
    call it unchanged.

Nothing here imports the reference: the graft takes its class as an argument.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import Engine, PinnedBuffer, check_checkpoint, make_config, require_gpu_tensor
from .noise import exponential_draws, make_noise_tape

__all__ = ["EngineHost", "infer_config_kwargs", "make_wavenet_amd"]


def infer_config_kwargs(m) -> dict:
    """The constructor arguments the engine needs (``engine.make_config``), read off a WaveNet module tree with the reference's
    attribute names (wavenet.py:112-156) -- the reference keeps only some of its constructor arguments as attributes, the rest is
    recovered from the sub-modules."""
    layers = len(m.conv_layers)
    first = m.conv_layers[0]
    dil = [int(f.conv.dilation[0]) for f in m.conv_layers]
    per = next((i for i in range(1, layers) if dil[i] == 1), layers)                # dilation = 2 ** (i % per)  (wavenet.py:126)
    if layers % per or any(d != 2 ** (i % per) for i, d in enumerate(dil)):
        raise NotImplementedError(f"dilation pattern {dil} is not 2 ** (i % layers_per_stack)")
    gin = int(first.conv1x1g.in_channels) if getattr(first, "conv1x1g", None) is not None else -1
    emb = getattr(m, "embed_speakers", None)
    kw = dict(out_channels=int(m.out_channels), layers=layers, stacks=layers // per,
              residual_channels=int(m.first_conv.out_channels), gate_channels=int(first.conv.out_channels),
              skip_out_channels=int(first.conv1x1_skip.out_channels), kernel_size=int(first.conv.kernel_size[0]),
              cin_channels=int(m.cin_channels), gin_channels=gin,
              n_speakers=None if emb is None else int(emb.num_embeddings), use_speaker_embedding=emb is not None,
              scalar_input=bool(m.scalar_input), output_distribution=m.output_distribution,
              upsample_net=None, upsample_scales=[], freq_axis_kernel_size=1, cin_pad=0, upsample_activation="none",
              upsample_activation_params={}, upsample_mode="nearest")
    up = getattr(m, "upsample_net", None)
    if up is not None:
        kind = type(up).__name__
        inner = up.upsample if kind == "ConvInUpsampleNetwork" else up
        from ._lib import UPSAMPLE_ACT
        scales, freq, act, act_params, mode = [], 1, "none", {}, "nearest"
        for layer in inner.up_layers:
            name = type(layer).__name__
            if name == "Stretch2d":
                from ._lib import UPSAMPLE_MODE
                if getattr(layer, "mode", "nearest") not in UPSAMPLE_MODE or int(getattr(layer, "y_scale", 1)) != 1 \
                        or float(layer.x_scale) != int(layer.x_scale):
                    raise NotImplementedError("stretching along time by an integer factor is implemented (upsample.py:20)")
                mode = getattr(layer, "mode", "nearest")
                scales.append(int(layer.x_scale))
            elif hasattr(layer, "kernel_size"):
                freq = int(layer.kernel_size[0])
            elif name in UPSAMPLE_ACT:                                  # the per-stage activation module (upsample.py:47-49)
                pname = UPSAMPLE_ACT[name][1]
                act, act_params = name, ({pname: float(getattr(layer, pname))} if pname else {})
            else:
                raise NotImplementedError(f"upsample activation {name} is not implemented")
        total = 1
        for s in scales:
            total *= s
        cin_pad = (int(up.conv_in.kernel_size[0]) - 1) // 2 if kind == "ConvInUpsampleNetwork" else int(up.indent) // total
        kw.update(upsample_net=kind, upsample_scales=scales, freq_axis_kernel_size=freq, cin_pad=cin_pad, upsample_activation=act,
                  upsample_activation_params=act_params, upsample_mode=mode)
    return kw


class EngineHost:
    """Mixin for an ``nn.Module`` with the reference WaveNet's attribute names; see the module docstring."""

    # engine state: class-level defaults so that a grafted subclass needs no __init__ of its own
    rng = "replay"            # "replay": the stream torch's CPU generator gives the reference for the current seed | "philox"
    kernel = 0                # 0 auto, 1 generic single-workgroup kernel, 2 pipelined ring kernel
    capture_params = False    # keep the head outputs (B, O, T) of the last call in ``last_params``
    last_params = None
    _engine: Optional[Engine] = None
    _engine_key = None

    def _wnv_config_kwargs(self) -> dict:
        kw = getattr(self, "_cfg_kwargs", None)
        return dict(kw) if kw is not None else infer_config_kwargs(self)

    def check_engine_checkpoint(self, batch: int = 8) -> dict:
        """This module's ``state_dict`` through the engine's native checkpoint path on the HOST (no GPU needed): validation,
        weight-norm fold, packing; returns {macs_per_sample, bytes_per_step, receptive_field} as the engine computes them."""
        return check_checkpoint(make_config(**self._wnv_config_kwargs()), self.state_dict(), batch)

    def _check_speaker_ids(self, g_ids):
        """nn.Embedding raises IndexError for an id outside [0, n_speakers) (modules.py:21-24 via wavenet.py:264-268); the
        device kernel would read past the table instead, so the check happens here (one tiny reduction, once per call)."""
        n = self.embed_speakers.num_embeddings
        if g_ids.numel() and (int(g_ids.min()) < 0 or int(g_ids.max()) >= n):
            raise IndexError(f"speaker id out of range [0, {n})")

    def invalidate_engine(self):
        """Drop the packed weights (call after changing parameters through ``.data`` or any other route that does not bump
        the tensors' version counters; ordinary in-place updates are detected by ``_get_engine``)."""
        if self._engine is not None:
            self._engine.close()
        self._engine, self._engine_key = None, None
        for f in self.conv_layers:
            if hasattr(f, "invalidate_engine"):
                f.invalidate_engine()

    # ---- engine plumbing ---------------------------------------------------------------------------
    def _get_engine(self) -> Engine:
        params = list(self.parameters())
        require_gpu_tensor(params[0], "WaveNet parameters")
        dev = params[0].device
        key = tuple((p.device, p.data_ptr(), p._version) for p in params)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            eng = Engine(make_config(**self._wnv_config_kwargs()), dev)
            eng.load_weights(self.state_dict())
            self._engine, self._engine_key = eng, key
        return self._engine

    def incremental_forward(self, initial_input=None, c=None, g=None, T=100, test_inputs=None,
                            tqdm=lambda x: x, softmax=True, quantize=True, log_scale_min=-50.0):
        """Autoregressive generation; same signature and return layout as reference wavenet.py:215-343.
        ``tqdm`` is accepted and ignored (there is no per-sample host iteration to wrap); ``log_scale_min``
        is accepted and unused exactly as in the reference (mixture.py:147-148: clamp_log_scale=False)."""
        if self.training:
            raise RuntimeError('incremental_forward only supports eval mode')     # conv.py:19-20
        eng = self._get_engine()
        dev = eng.device
        C_in = 1 if self.scalar_input else self.out_channels

        def prep(t):
            return None if t is None else t.detach().to(device=dev)

        initial_input, c, g, test_inputs = prep(initial_input), prep(c), prep(g), prep(test_inputs)
        B = 1
        if test_inputs is not None:                                               # wavenet.py:245-258
            if self.scalar_input:
                if test_inputs.size(1) == 1:
                    test_inputs = test_inputs.transpose(1, 2)
            elif test_inputs.size(1) == self.out_channels:
                test_inputs = test_inputs.transpose(1, 2)
            test_inputs = test_inputs.float().contiguous()                        # (B, Tt, C)
            B = test_inputs.size(0)
            T = test_inputs.size(1) if T is None else max(int(T), test_inputs.size(1))
        T = int(T)
        if c is not None:
            B = c.shape[0]
        elif test_inputs is None:
            if initial_input is not None:
                B = initial_input.size(0)
            elif g is not None:
                B = g.size(0)
        # global conditioning (wavenet.py:262-269): ids -> embedding, or external float features
        g_ids = g_feat = None
        if g is not None:
            if self.embed_speakers is not None:
                g_ids = g.reshape(B, -1)[:, 0].to(torch.int64).contiguous()
                self._check_speaker_ids(g_ids)
            else:
                g_feat = g.float().reshape(B, -1).contiguous()
                assert g_feat.size(1) == self._wnv_config_kwargs()["gin_channels"]
        # local conditioning (wavenet.py:272-278)
        c_up = None
        if c is not None:
            c = c.float()
            if self.upsample_net is not None:
                c_up = eng.upsample(c.contiguous(), T_expected=T)                 # asserts length == T
            elif c.size(-1) == T:
                c_up = c.transpose(1, 2).contiguous()
            else:
                c_up = c.contiguous()                                             # already (B, T, cin)
            assert c_up.shape == (B, T, self.cin_channels), (tuple(c_up.shape), (B, T, self.cin_channels))
        # first input (wavenet.py:281-292)
        init = None
        if initial_input is not None:
            if initial_input.size(1) == self.out_channels and not self.scalar_input:
                initial_input = initial_input.transpose(1, 2)
            init = initial_input.float().reshape(B, -1).contiguous()
            assert init.size(1) == C_in, (tuple(init.shape), C_in)
        # noise
        noise, seed = None, 0
        if self.rng == "replay":
            if not self.scalar_input and quantize and self.kernel in (0, 2) and self.stream_replay_tape and T >= 1024:
                # a mu-law model draws B x out_channels exponentials per step (wavenet.py:334-335): the tape takes longer to draw than
                # the kernel runs -- so it is drawn WHILE the kernel runs (ring kernel only; anything else takes the path below)
                skip = self.__dict__.get("_stream_cooldown", 0)
                if skip > 0 and self.kernel == 0:
                    # auto mode, and the last streamed launch timed out: the device does not keep the ring co-resident right now.  The
                    # streamed path names the ring kernel explicitly (it has to: WNV_GEN_ASYNC), which would bypass the handle's own
                    # pause after a time-out -- so it pauses the same way (2, 4 ... 32 calls) and the ordinary path, which honours the
                    # handle's state, serves these calls.
                    self.__dict__["_stream_cooldown"] = skip - 1
                else:
                    out = self._generate_streamed(eng, B, T, c_up, g_feat, g_ids, init, test_inputs, softmax)
                    if out is not None:
                        return out
            tape = make_noise_tape(T, B, scalar_input=self.scalar_input,
                                   output_distribution=self.output_distribution, out_channels=self.out_channels)
            noise = tape.to(dev, non_blocking=False).contiguous()
        elif self.rng == "philox":
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        else:
            raise ValueError(f"unknown rng mode {self.rng!r}")
        out, params, _ = eng.generate(B=B, T=T, c_up=c_up, g=g_feat, g_ids=g_ids, initial=init,
                                      teacher=test_inputs, noise=noise, seed=seed, softmax=softmax,
                                      quantize=quantize, want_params=self.capture_params, kernel=self.kernel)
        self.last_params = params
        return out


    stream_replay_tape = True     # one-hot models: draw the replay tape while the ring kernel runs (False: draw it first, as for the other kernels)
    stream_replay_tape_max_bytes = 1 << 30   # ... unless the tape would need more pinned host memory than this
    stream_replay_max_batch = 64             # ... or the batch is more than one ring launch pipelines (WNV_GEN_ASYNC: wnv.h)

    def _generate_streamed(self, eng, B, T, c_up, g_feat, g_ids, init, test_inputs, softmax):
        """``rng = "replay"`` for one-hot models at kernel speed: the tape of B x out_channels exponentials per step -- the numbers
        torch's CPU generator hands the reference's ``OneHotCategorical.sample`` (wavenet.py:334-335) -- lives in coherent host memory
        the device reads directly; the ring kernel is launched asynchronously at once and waits, step by step, for a counter the host
        advances as it draws the tape chunk by chunk (same draws, same order, same generator: ``exponential_draws``).  Returns None
        -- with the generator untouched -- when the ring kernel does not take the call (other kernels get their tape up front)."""
        nz = self.out_channels
        if B > self.stream_replay_max_batch:                    # an asynchronous launch is ONE launch: larger batches run in slices (tape up front)
            return None
        probe = torch.empty(1)
        state = torch.get_rng_state()
        if not exponential_draws(probe):                        # the fast draw is unavailable on this build: let the caller draw as before
            return None
        torch.set_rng_state(state)
        # (pinning memory costs ~0.4 ms per MB -- 67 MB of tape for one second of mu-law audio at B = 8 --, so the buffers stay with the
        #  module, grow-only, and are unpinned when it goes away: PinnedBuffer.__del__)
        cache = self.__dict__.setdefault("_pinned_tape", {})
        need = T * B * nz * 4
        if cache.get("device") != eng.device:                   # (the module moved to another GPU: the mapping was made for the old one)
            for key in ("tape", "ready"):
                if cache.get(key) is not None:
                    cache[key].free()
                cache[key] = None
            cache["device"] = eng.device
        if need > self.stream_replay_tape_max_bytes:            # (ten seconds of mu-law audio at B = 8 are 2 GB of tape: not pinned)
            return None
        try:
            if cache.get("tape") is None or not cache["tape"].host or cache["tape"].nbytes < need:
                if cache.get("tape") is not None:
                    cache["tape"].free()
                    cache["tape"] = None
                cache["tape"] = PinnedBuffer(need)
            if cache.get("ready") is None or not cache["ready"].host:
                cache["ready"] = PinnedBuffer(64)
        except RuntimeError:                                    # the host refuses to pin that much: draw the tape up front instead
            return None
        tape_buf, ready_buf = cache["tape"], cache["ready"]
        try:
            tape = tape_buf.view(torch.float32, (T, B, nz))
            ready = ready_buf.view(torch.int32, (1,))
            ready[0] = 0
            timed_out = False
            try:
                out, params, _ = eng.generate(B=B, T=T, c_up=c_up, g=g_feat, g_ids=g_ids, initial=init, teacher=test_inputs, noise=tape_buf.dev,
                                              noise_ready=ready_buf.dev, softmax=softmax, quantize=True, want_params=self.capture_params,
                                              kernel=2, asynchronous=True)
            except (NotImplementedError, TimeoutError, ValueError) as e:
                # not a ring configuration / no room for the ring / an argument the asynchronous launch refuses: nothing was drawn, the
                # generator is untouched -- the caller takes the ordinary path (which reports a real argument error itself)
                if isinstance(e, TimeoutError):
                    self._note_stream_timeout()
                return None
            step = max(64, (1 << 19) // (B * nz))               # ~0.5 M values per chunk: a few milliseconds of drawing
            for t0 in range(0, T, step):
                t1 = min(T, t0 + step)
                exponential_draws(tape[t0:t1])
                ready[0] = t1                                    # (x86: the tape's stores are visible before the counter's)
            try:
                eng.wait()
            except TimeoutError:
                # the launch lost its CUs on the way: the tape is complete by now -- serve the call through the ordinary path with it
                self._note_stream_timeout()
                timed_out = True
                noise = tape.clone().to(eng.device)
                out, params, _ = eng.generate(B=B, T=T, c_up=c_up, g=g_feat, g_ids=g_ids, initial=init, teacher=test_inputs, noise=noise,
                                              softmax=softmax, quantize=True, want_params=self.capture_params, kernel=1)
            if not timed_out:
                self.__dict__["_stream_backoff"] = 0
            self.last_params = params
            return out
        finally:
            torch.cuda.synchronize(eng.device)                   # nothing on the device reads the buffers any more

    def _note_stream_timeout(self):
        """A streamed launch gave up its bounded waits: pause the streamed path for 2, 4 ... 32 calls (the handle's own policy for
        auto mode, wnv_host.cpp ``persist_cooldown``), so that a device that cannot keep the ring resident does not cost every
        one-hot call a time-out before the fallback."""
        back = self.__dict__.get("_stream_backoff", 0)
        back = min(2 * back, 32) if back else 2
        self.__dict__["_stream_backoff"] = back
        self.__dict__["_stream_cooldown"] = back


def make_wavenet_amd(reference_wavenet_cls):
    """``class WaveNetAMD(EngineHost, wavenet_vocoder.WaveNet)``: the reference's class with its sample loop in libwnv_hip.so.

        from wavenet_vocoder import WaveNet
        from wavenet_vocoder_amd.graft import make_wavenet_amd
        WaveNetAMD = make_wavenet_amd(WaveNet)
        model = WaveNetAMD(**kwargs); model.load_state_dict(checkpoint["state_dict"]); model.eval().to("cuda")
        y = model.incremental_forward(c=c, g=g, T=T, softmax=True, quantize=True)      # synthesis.py:61-64, unchanged

    The constructor, ``forward``, ``state_dict`` keys (weight-normed until ``make_generation_fast_``; the engine folds either
    form) and everything training uses are inherited from the reference; only ``incremental_forward`` (and with it
    ``clear_buffer``'s meaning for the engine) comes from ``EngineHost``."""
    if not isinstance(reference_wavenet_cls, type):
        raise TypeError("make_wavenet_amd expects the reference's WaveNet class")
    return type("WaveNetAMD", (EngineHost, reference_wavenet_cls),
                {"__doc__": "wavenet_vocoder.WaveNet with incremental_forward running in the MI355X engine (libwnv_hip.so)",
                 "__module__": __name__})
