"""Host-side mirror of the reference's model objects for the decode path.

`DiffAudioRep` here exposes the attributes `synthesis()` touches in the reference
(srcs/sample.py:56-131): `.get_cond(wav)`, `.encoder(wav)`, `.decoder(z)`, `.quantizer(...)`,
`.diff_model(x, t, cond)`, `.diff_model.upsampling_layers` (applied in order),
`.diffusion.halfway_sampling(img, t, condition)`, `.diffusion.p_sample(x, t, cond)`.
Every one of them is a thin call into libladiffcodec.so on torch CUDA(ROCm) tensors; nothing is
computed in Python.  One `Engine` (= one ldc_ctx) holds both models of a decode session, as the
hipGraph, workspaces and step tables are shared.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import numpy as np

from . import lib as L
from .spec import CodecConfig, UnetConfig


@dataclass
class QuantizedResult:            # reference srcs/quantization/vq.py:19-25
    quantized: "object"
    codes: "object"
    bandwidth: "object"
    penalty: Optional["object"] = None


class Engine:
    """Owns the ldc_ctx.  dtype: 'bf16' (throughput), 'f32' (exact-fp32 MFMA path, parity) or 'fp8' (bf16 activations,
    UNet conv weights as OCP fp8 e4m3 with per-output-channel scales)."""

    def __init__(self, main_codec: CodecConfig, unet: UnetConfig, cond_codec: Optional[CodecConfig] = None,
                 dtype: str = "bf16", device: int = 0, noise_seed: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("ladiffcodec_amd needs an MI355X (gfx950) GPU; no CPU fallback exists")
        self.lib = L.load()
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.main_codec, self.unet, self.cond_codec = main_codec, unet, cond_codec
        cfg = L.LdcConfig()
        if dtype not in L.DTYPES:
            raise ValueError(f"dtype {dtype!r}: one of {sorted(L.DTYPES)}")
        cfg.compute_dtype = L.DTYPES[dtype]
        self.dtype = dtype
        cfg.rep_dims, cfg.n_filters = main_codec.rep_dims, main_codec.n_filters
        cfg.n_residual_layers, cfg.lstm = main_codec.n_residual_layers, main_codec.lstm
        cfg.n_enc_ratios = len(main_codec.enc_ratios)
        for i, r in enumerate(main_codec.enc_ratios):
            cfg.enc_ratios[i] = r
        cfg.diff_dims = unet.dim
        ups = list(unet.upsampling_ratios or [])
        cfg.n_upsampling_ratios = len(ups)
        for i, r in enumerate(ups):
            cfg.upsampling_ratios[i] = r
        cfg.unet_scale_cond, cfg.unet_scale_x = int(unet.unet_scale_cond), int(unet.unet_scale_x)
        cfg.has_cond_model = int(cond_codec is not None)
        cfg.cond_bandwidth = float(cond_codec.bandwidth) if cond_codec is not None else 3.0
        cfg.noise_seed = noise_seed
        if main_codec.final_activation not in L.FINAL_ACTIVATIONS:
            raise ValueError(f"final_activation {main_codec.final_activation!r}: one of {sorted(k for k in L.FINAL_ACTIVATIONS if k)}")
        if cond_codec is not None and cond_codec.final_activation != main_codec.final_activation:
            raise ValueError("both models are built with the same --final_activation (sample.py:54,63)")
        cfg.final_activation = L.FINAL_ACTIVATIONS[main_codec.final_activation]
        self._ctx = C.c_void_p()
        L.check(self.lib.ldc_create(C.byref(cfg), device, C.byref(self._ctx)))
        self.stream = torch.cuda.Stream(device=self.device)
        self._finalized = False

    def set_option(self, name: str, value: int) -> None:
        """ldc_set_option: 'split' (chains per batch), 'lstm_stream' (no cooperative LSTM), 'side_streams', 'fp8_act',
        'train_fp32_mfma' (training GEMMs on the exact-fp32 MFMA instead of the split-bf16 path; process-wide), 'train_bf16' (plain bf16
        products in the training GEMMs, opt-in; process-wide)."""
        L.check(self.lib.ldc_set_option(self._ctx, name.encode(), int(value)))

    def debug_raise_failure(self, code: int) -> None:
        """test hook: raise the device-side failure flag (1 = cooperative LSTM gave up, 2 = fused GroupNorm wait gave up)"""
        L.check(self.lib.ldc_debug_raise_failure(self._ctx, int(code)))

    def host_stats(self, reset: bool = True):
        """-> (ms inside hipGraphLaunch, ms waiting for the look-ahead window, graph replays) since the last reset"""
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        L.check(self.lib.ldc_host_stats(self._ctx, int(reset), C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def stream_info(self):
        """What the part-stream calibration measured: {'overlapping': streams of the context that run side by side with the caller's stream
        and each other (-1: no calibration yet), 'candidates', 'one_spin_ms', 'all_spin_ms', 'parts'}."""
        g, n, p = C.c_int(), C.c_int(), C.c_int()
        a, b = C.c_double(), C.c_double()
        L.check(self.lib.ldc_stream_info(self._ctx, C.byref(g), C.byref(n), C.byref(a), C.byref(b), C.byref(p)))
        return {"overlapping": g.value, "candidates": n.value, "one_spin_ms": a.value, "all_spin_ms": b.value, "parts": p.value}

    def clock_sample(self):
        """(100 MHz wall-clock ticks, shader cycles) behind everything queued on the current stream; synchronises it."""
        buf = (C.c_uint64 * 2)()
        s = self._enter()
        L.check(self.lib.ldc_clock_sample(self._ctx, buf, s))
        self._exit()
        return int(buf[0]), int(buf[1])

    def reseed(self, seed: int) -> None:
        """torch.manual_seed counterpart for the device-drawn noise: sets the Philox seed and rewinds the call counter."""
        L.check(self.lib.ldc_reseed(self._ctx, int(seed)))

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.torch.cuda.synchronize(self.device)
            self.lib.ldc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- load_model(model, path, strict) : utils.py:98-108 --------------------------------------
    def load_state_dict(self, which: int, state_dict: Dict[str, np.ndarray]) -> None:
        for key, arr in state_dict.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            L.check(self.lib.ldc_set_weight(self._ctx, which, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def finalize(self, strict: bool = True) -> None:
        L.check(self.lib.ldc_finalize_weights(self._ctx, int(strict)))
        self._finalized = True

    # ---- plumbing --------------------------------------------------------------------------------
    def _f32(self, t):
        torch = self.torch
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(self.device, torch.float32).contiguous()
        return t

    def _enter(self):
        cur = self.torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:      # (a caller already running on the engine's stream needs no hand-over:
            self.stream.wait_stream(cur)                    # the training step makes ~1 600 calls, see DiffusionTrainer._on_engine_stream)
        return C.c_void_p(self.stream.cuda_stream)

    def _exit(self):
        cur = self.torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def _empty(self, *shape, dtype=None):
        return self.torch.empty(*shape, device=self.device, dtype=dtype or self.torch.float32)

    # ---- stages ----------------------------------------------------------------------------------
    def encode(self, which: int, wav):
        wav = self._f32(wav)
        B, _, T = wav.shape
        hop = (self.cond_codec if which == L.MODEL_COND else self.main_codec).hop_length
        z = self._empty(B, self.main_codec.rep_dims, -(-T // hop))
        s = self._enter()
        L.check(self.lib.ldc_seanet_encode(self._ctx, which, wav.data_ptr(), B, T, z.data_ptr(), s))
        self._exit()
        return z

    def decode_latents(self, which: int, z):
        z = self._f32(z)
        B, _, Lz = z.shape
        hop = (self.cond_codec if which == L.MODEL_COND else self.main_codec).hop_length
        wav = self._empty(B, 1, Lz * hop)
        s = self._enter()
        L.check(self.lib.ldc_seanet_decode(self._ctx, which, z.data_ptr(), B, Lz, wav.data_ptr(), s))
        self._exit()
        return wav

    def rvq(self, z, n_q: int):
        z = self._f32(z)
        B, D, F = z.shape
        codes = self._empty(n_q, B, F, dtype=self.torch.int64)
        q = self._empty(B, D, F)
        s = self._enter()
        L.check(self.lib.ldc_rvq_encode(self._ctx, z.data_ptr(), B, F, n_q, codes.data_ptr(), q.data_ptr(), s))
        self._exit()
        return q, codes

    def rvq_decode(self, codes):
        codes = codes.to(self.device, self.torch.int64).contiguous()
        n_q, B, F = codes.shape
        q = self._empty(B, self.main_codec.rep_dims, F)
        s = self._enter()
        L.check(self.lib.ldc_rvq_decode(self._ctx, codes.data_ptr(), B, F, n_q, q.data_ptr(), s))
        self._exit()
        return q

    def resample(self, wav, orig_freq: int, new_freq: int = 16000):
        """torchaudio.functional.resample(wav, orig_freq, new_freq) (sample.py:84); wav [C, T] -> [C, ceil(T * new / orig)]."""
        wav = self._f32(wav)
        Cc, T = wav.shape
        out = self._empty(Cc, int(self.lib.ldc_resample_out_len(T, int(orig_freq), int(new_freq))))
        s = self._enter()
        L.check(self.lib.ldc_resample(self._ctx, wav.data_ptr(), Cc, T, int(orig_freq), int(new_freq), out.data_ptr(), s))
        self._exit()
        return out

    def get_cond(self, wav, bandwidth: float = 0.0, return_codes: bool = False):
        wav = self._f32(wav)
        B, _, T = wav.shape
        F = -(-T // self.cond_codec.hop_length)
        n_q = self.cond_codec.n_q_for_bandwidth(bandwidth if bandwidth > 0 else None)
        cond = self._empty(B, self.cond_codec.rep_dims, F)
        codes = self._empty(n_q, B, F, dtype=self.torch.int64) if return_codes else None
        s = self._enter()
        L.check(self.lib.ldc_get_cond(self._ctx, wav.data_ptr(), B, T, float(bandwidth), cond.data_ptr(),
                                      codes.data_ptr() if codes is not None else None, s))
        self._exit()
        return (cond, codes) if return_codes else cond

    def cond_upsample(self, cond, normalise: int = 0):
        cond = self._f32(cond)
        B, Cc, F = cond.shape
        f = int(np.prod(self.unet.upsampling_ratios or ()))
        img = self._empty(B, Cc, F * f)
        s = self._enter()
        L.check(self.lib.ldc_cond_upsample(self._ctx, cond.data_ptr(), B, F, normalise, img.data_ptr(), s))
        self._exit()
        return img

    def unet_forward(self, x, t: int, cond):
        x, cond = self._f32(x), self._f32(cond)
        B, Cx, Lx = x.shape
        eps = self._empty(B, Cx, Lx)
        s = self._enter()
        L.check(self.lib.ldc_unet_forward(self._ctx, x.data_ptr(), int(t), cond.data_ptr(), B, Lx, cond.shape[2],
                                          eps.data_ptr(), s))
        self._exit()
        return eps

    def debug_tap(self, name: str, shape):
        out = self._empty(*shape)
        s = self._enter()
        L.check(self.lib.ldc_unet_debug_tap(self._ctx, name.encode(), out.data_ptr(), out.numel(), s))
        self._exit()
        return out

    def p_sample(self, x, t: int, cond, noise=None):
        x = self._f32(x).clone()
        cond = self._f32(cond)
        noise = self._f32(noise) if noise is not None else None
        B, _, Lx = x.shape
        s = self._enter()
        L.check(self.lib.ldc_p_sample(self._ctx, x.data_ptr(), int(t), cond.data_ptr(),
                                      noise.data_ptr() if noise is not None else None, B, Lx, cond.shape[2], s))
        self._exit()
        return x

    def denoise(self, img, cond, n_steps: int, noise=None, inplace: bool = False):
        img = self._f32(img)
        if not inplace:
            img = img.clone()
        cond = self._f32(cond)
        noise = self._f32(noise) if noise is not None else None
        B, _, Lx = img.shape
        s = self._enter()
        L.check(self.lib.ldc_denoise(self._ctx, img.data_ptr(), cond.data_ptr(),
                                     noise.data_ptr() if noise is not None else None, int(n_steps), B, Lx, cond.shape[2], s))
        self._exit()
        return img

    def p_sample_loop(self, cond, img=None, noise=None, length: Optional[int] = None):
        """diffusion.p_sample_loop: all `timesteps` ancestral steps from `img` (or from a device-drawn N(0,1) image)."""
        cond = self._f32(cond)
        B, _, F = cond.shape
        if img is None:
            Lx = int(length) if length is not None else F * int(np.prod(self.unet.upsampling_ratios or ()))
            img = self.torch.empty(B, self.unet.inp_channels, Lx, device=cond.device, dtype=self.torch.float32)
            fill = 1
        else:
            img = self._f32(img).clone()
            fill = 0
        noise = self._f32(noise) if noise is not None else None
        s = self._enter()
        L.check(self.lib.ldc_p_sample_loop(self._ctx, img.data_ptr(), cond.data_ptr(),
                                           noise.data_ptr() if noise is not None else None, fill, B, img.shape[2], F, s))
        self._exit()
        return img

    def infilling(self, infill_img, cond, midway_t: int, lam: float = 0.8, img=None, noise=None):
        """diffusion.infilling; returns (img, infill_img) after the loop (the reference returns img)."""
        cond = self._f32(cond)
        infill = self._f32(infill_img).clone()
        B, _, Lx = infill.shape
        if img is None:
            img = self.torch.empty_like(infill)
            fill = 1
        else:
            img = self._f32(img).clone()
            fill = 0
        noise = self._f32(noise) if noise is not None else None
        s = self._enter()
        L.check(self.lib.ldc_infilling(self._ctx, img.data_ptr(), infill.data_ptr(), cond.data_ptr(), int(midway_t),
                                       noise.data_ptr() if noise is not None else None, float(lam), fill, B, Lx, cond.shape[2], s))
        self._exit()
        return img, infill

    def output_normalise(self, wav, per_item: bool = False):
        wav = self._f32(wav).clone()
        B = wav.shape[0]
        s = self._enter()
        L.check(self.lib.ldc_output_normalise(self._ctx, wav.data_ptr(), B, wav.numel() // B, int(per_item), s))
        self._exit()
        return wav

    def decode(self, wav, n_steps: int, noise=None, per_item: bool = False, want_stages: bool = False):
        """The per-batch body of synthesis() (sample.py:94-134) in one library call."""
        wav = self._f32(wav)
        B, _, T = wav.shape
        F, Lz = T // self.cond_codec.hop_length, T // self.main_codec.hop_length
        out = self._empty(B, 1, T)
        lat = self._empty(B, self.main_codec.rep_dims, Lz) if want_stages else None
        cond = self._empty(B, self.main_codec.rep_dims, F) if want_stages else None
        n_q = self.cond_codec.n_q_for_bandwidth(None)
        codes = self._empty(n_q, B, F, dtype=self.torch.int64) if want_stages else None
        noise = self._f32(noise) if noise is not None else None
        s = self._enter()
        p = lambda t: t.data_ptr() if t is not None else None
        L.check(self.lib.ldc_decode(self._ctx, wav.data_ptr(), B, T, int(n_steps), p(noise), int(per_item), out.data_ptr(),
                                    p(lat), p(cond), p(codes), s))
        self._exit()
        if want_stages:
            return {"wav": out, "latents": lat, "cond": cond, "codes": codes}
        return out

    # ---- accounting ------------------------------------------------------------------------------
    def unet_step_cost(self, B: int, Lz: int):
        fl, by = C.c_double(), C.c_double()
        L.check(self.lib.ldc_unet_step_cost(self._ctx, B, Lz, C.byref(fl), C.byref(by)))
        return fl.value, by.value

    def timeline(self, n_steps: int, parts: int = 2):
        """Device-side timeline of the last sampler call (after timeline_enable(True)): per batch part an array
        [n_steps, 2] of begin / end times in microseconds relative to the earliest stamp."""
        out = []
        for k in range(parts):
            buf = (C.c_uint64 * (2 * n_steps))()
            L.check(self.lib.ldc_timeline_read(self._ctx, k, n_steps, buf))
            out.append(np.array(buf, dtype=np.float64).reshape(n_steps, 2))
        t0 = min(float(a[a > 0].min()) for a in out if (a > 0).any())
        return [(a - t0) / 100.0 for a in out]      # 100 MHz ticks -> us

    def kstamps_enable(self, on: bool):
        L.check(self.lib.ldc_kstamps_enable(self._ctx, int(on)))

    def kstamps_reset(self):
        L.check(self.lib.ldc_kstamps_reset(self._ctx))

    def kstamps(self, part: int, n_steps: int):
        """Timed-mode stamps of batch part `part` after a decode with kstamps_enable(True): (ticks [n_steps, n_ops, 2] in
        microseconds on the device's 100 MHz clock (0 where the op is not a pipelined conv), op descriptions, class codes)."""
        n = C.c_int()
        L.check(self.lib.ldc_kstamps_read(self._ctx, part, n_steps, C.byref(n), None, None, 0, None))
        nops, cap = n.value, 96
        ticks = (C.c_uint64 * (n_steps * nops * 2))()
        infos = C.create_string_buffer(nops * cap)
        classes = (C.c_int * nops)()
        L.check(self.lib.ldc_kstamps_read(self._ctx, part, n_steps, C.byref(n), ticks, infos, cap, classes))
        t = np.array(ticks, dtype=np.float64).reshape(n_steps, nops, 2) / 100.0
        names = [infos.raw[o * cap:(o + 1) * cap].split(b"\0", 1)[0].decode() for o in range(nops)]
        return t, names, list(classes)

    def timeline_enable(self, on: bool):
        L.check(self.lib.ldc_timeline_enable(self._ctx, int(on)))

    def profile(self, on: bool):
        L.check(self.lib.ldc_profile_enable(self._ctx, int(on)))

    def profile_read(self):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        L.check(self.lib.ldc_profile_read(self._ctx, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    CLASS_NAMES = ("other", "conv_gemm", "gn_apply", "layernorm", "linear_attention", "attention_full", "elementwise")

    def profile_read_classes(self):
        """Per kernel class: (name, ms, launches, algorithmic flops, algorithmic bytes) of the profiling pass."""
        n = len(self.CLASS_NAMES)
        ms, la, fl, by = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)()
        L.check(self.lib.ldc_profile_read_classes(self._ctx, n, ms, la, fl, by))
        return [(self.CLASS_NAMES[k], ms[k], la[k], fl[k], by[k]) for k in range(n)]

    # ---- L1 primitives (parity tests) ------------------------------------------------------------
    def sconv1d(self, x, w, b, stride=1, dilation=1, causal=True, pre_elu=False):
        x = self._f32(x)
        w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        B, Cin, Lx = x.shape
        Cout, _, k = w.shape
        Lout = -(-Lx // stride)
        y = self._empty(B, Cout, Lout)
        s = self._enter()
        L.check(self.lib.ldc_sconv1d(self._ctx, x.data_ptr(), B, Cin, Lx, w.ctypes.data_as(C.c_void_p),
                                     b.ctypes.data_as(C.c_void_p), Cout, k, stride, dilation, int(causal), int(pre_elu),
                                     y.data_ptr(), s))
        self._exit()
        return y

    def sconvtr1d(self, x, w, b, stride, causal):
        x = self._f32(x)
        w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        B, Cin, Lx = x.shape
        _, Cout, k = w.shape
        y = self._empty(B, Cout, Lx * stride)
        s = self._enter()
        L.check(self.lib.ldc_sconvtr1d(self._ctx, x.data_ptr(), B, Cin, Lx, w.ctypes.data_as(C.c_void_p),
                                       b.ctypes.data_as(C.c_void_p), Cout, k, stride, int(causal), y.data_ptr(), s))
        self._exit()
        return y

    def slstm(self, x, weights: Sequence[np.ndarray], layers: int):
        x = self._f32(x)
        B, H, T = x.shape
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        arr = (C.c_void_p * len(ws))(*[w.ctypes.data_as(C.c_void_p) for w in ws])
        y = self._empty(B, H, T)
        s = self._enter()
        L.check(self.lib.ldc_slstm(self._ctx, x.data_ptr(), B, H, T, arr, layers, y.data_ptr(), s))
        self._exit()
        return y


# --------------------------------------------------------------------------------------------------
# reference-shaped facade
# --------------------------------------------------------------------------------------------------
class _Upsampler:
    """Stands for `diff_model.upsampling_layers`: iterating yields one callable that applies the WHOLE
    stack (the reference applies them in sequence, sample.py:127-128), so `for layer in
    model.diff_model.upsampling_layers: img = layer(img)` gives the same tensor."""

    def __init__(self, eng: Engine):
        self._eng = eng

    def __iter__(self):
        yield lambda img: self._eng.cond_upsample(img, 0)

    def __len__(self):
        return 1


class _DiffModel:
    def __init__(self, eng: Engine):
        self._eng = eng
        self.upsampling_layers = _Upsampler(eng)
        self.channels = eng.unet.inp_channels
        self.self_condition = False

    def __call__(self, x, time, x_cond=None):
        t = int(time.reshape(-1)[0].item()) if hasattr(time, "reshape") else int(time)
        if hasattr(time, "reshape") and bool((time != t).any()):
            raise ValueError("per-item timesteps are not supported (the sampler never uses them)")
        return self._eng.unet_forward(x, t, x_cond)


class _Diffusion:
    def __init__(self, eng: Engine):
        self._eng = eng
        self.seq_length = None
        self.num_timesteps = eng.unet.timesteps

    def p_sample(self, x, t: int, condition=None, noise=None):
        return self._eng.p_sample(x, t, condition, noise), None

    def halfway_sampling(self, img=None, t=None, condition=None, noise=None):
        if tuple(img.shape) == tuple(condition.shape):       # ddpm_loss.py:376-378
            if self._eng.unet.upsampling_ratios is None:
                # the reference iterates model.upsampling_layers here, an attribute that does not exist without
                # upsampling_ratios (unet.py:372): same error, same place
                raise AttributeError("'Unet1D' object has no attribute 'upsampling_layers'")
            img = self._eng.cond_upsample(img, 0)
        return self._eng.denoise(img, condition, int(t), noise)

    def p_sample_loop(self, shape, condition=None, img=None, noise=None):
        """ddpm_loss.py:253-266; `img`/`noise` inject the start image and the per-step draws (parity runs)."""
        return self._eng.p_sample_loop(condition, img=img, noise=noise, length=shape[2])

    def sample(self, batch_size=16, condition=None):
        """ddpm_loss.py:305-309 (DDIM sampling is not on the hot path: is_ddim_sampling is False at 1000 steps)."""
        assert self.seq_length is not None, "set diffusion.seq_length as the reference's constructor does"
        return self.p_sample_loop((batch_size, self._eng.unet.inp_channels, self.seq_length), condition)

    def infilling(self, infill_img, condition, midway_t=None, noise=None, offset=0, lam=0.8, img=None, noises=None):
        """ddpm_loss.py:331-367 (`noise` and `offset` are accepted and unused, as in the reference); `img`/`noises`
        inject the start image and the 2*midway_t draws (parity runs)."""
        out, _ = self._eng.infilling(infill_img, condition, int(midway_t), lam=lam, img=img, noise=noises)
        return out


class DiffAudioRep:
    """Facade with the reference's attribute names over one side (main or cond) of an Engine."""

    def __init__(self, eng: Engine, which: int):
        self._eng, self._which = eng, which
        cfg = eng.cond_codec if which == L.MODEL_COND else eng.main_codec
        self.frame_rate = cfg.frame_rate
        self.bandwidth = cfg.bandwidth
        self.quantization = cfg.quantization
        if which == L.MODEL_MAIN:
            self.diff_model = _DiffModel(eng)
            self.diffusion = _Diffusion(eng)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def encoder(self, wav):
        return self._eng.encode(self._which, wav)

    def decoder(self, z):
        return self._eng.decode_latents(self._which, z)

    def quantizer(self, x, sample_rate=None, bandwidth=None, n_q=None):
        cfg = self._eng.cond_codec
        n = n_q if n_q is not None else cfg.n_q_for_bandwidth(bandwidth)
        q, codes = self._eng.rvq(x, n)
        torch = self._eng.torch
        bw = torch.tensor(n * 0.5).to(q)
        return QuantizedResult(q, codes, bw, penalty=torch.zeros((), device=q.device))

    def get_cond(self, x):
        if self._which != L.MODEL_COND:
            return self.encoder(x)
        return self._eng.get_cond(x)
