"""Host-side mirror of the training-step slice: `diffusion.q_sample`, the objective of
`diffusion.p_losses` and `Block` forward / backward (srcs/losses/ddpm_loss.py:386-438, srcs/modules/unet.py:137-154)
as calls into libladiffcodec.so on torch device tensors.  Weights are the TRAINED tensors (raw `proj.weight`: the weight
standardisation is differentiated through), layouts are the reference's [B, C, L] fp32.
"""
from __future__ import annotations

import ctypes as C

from . import lib as L


def grad_buffer(eng, param, shape=None):
    """Where a layer's backward writes the gradient of `param`: the view of the trainer's flat gradient buffer registered for
    this parameter (DiffusionTrainer binds every parameter's storage address, so the kernels write straight into the buffer
    the gradient reduction and Adam read: no per-step concatenation of 135 M elements), else a fresh tensor."""
    t = eng.torch
    views = getattr(eng, "_grad_views", None)
    if views is not None:
        v = views.get(param.data_ptr())
        if v is not None and v.numel() == param.numel():
            eng._grad_written.add(param.data_ptr())
            return v.view(*(shape if shape is not None else param.shape))
    return t.empty(*(shape if shape is not None else param.shape), dtype=t.float32, device=eng.device)


class Block:
    """unet.py:137-154.  forward(x, scale_shift) keeps what backward needs in a device workspace."""

    def __init__(self, eng, weight, bias, norm_weight, norm_bias, groups: int = 8):
        self.eng, self.lib, self.torch = eng, eng.lib, eng.torch
        f = lambda v: v.to(eng.device, self.torch.float32).contiguous()
        self.weight, self.bias, self.gamma, self.beta = f(weight), f(bias), f(norm_weight), f(norm_bias)
        self.groups = groups
        self._saved = None

    def forward(self, x, scale_shift=None):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, Cin, Lx = x.shape
        Cout = self.weight.shape[0]
        ss = None
        if scale_shift is not None:
            scale, shift = scale_shift
            ss = t.cat([scale.reshape(B, Cout), shift.reshape(B, Cout)], dim=1).to(self.eng.device, t.float32).contiguous()
        ws = t.empty(int(self.lib.ldc_train_block_ws_floats(B, Cin, Cout, Lx, self.groups)), dtype=t.float32, device=self.eng.device)
        y = t.empty(B, Cout, Lx, dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_block_forward(self.eng._ctx, x.data_ptr(), self.weight.data_ptr(), self.bias.data_ptr(),
                                                 self.gamma.data_ptr(), self.beta.data_ptr(), ss.data_ptr() if ss is not None else None,
                                                 B, Cin, Cout, Lx, self.groups, y.data_ptr(), ws.data_ptr(), s))
        self.eng._exit()
        self._saved = (x, ss, ws)
        return y

    def backward(self, dy):
        """-> dict(dx, dw, db, dgamma, dbeta[, dscale, dshift])"""
        t = self.torch
        x, ss, ws = self._saved
        B, Cin, Lx = x.shape
        Cout = self.weight.shape[0]
        dy = dy.to(self.eng.device, t.float32).contiguous()
        e = lambda *shape: t.empty(*shape, dtype=t.float32, device=self.eng.device)
        dx = e(B, Cin, Lx)
        dw, db = grad_buffer(self.eng, self.weight), grad_buffer(self.eng, self.bias)
        dg, dbt = grad_buffer(self.eng, self.gamma), grad_buffer(self.eng, self.beta)
        dss = e(B, 2 * Cout) if ss is not None else None
        s = self.eng._enter()
        L.check(self.lib.ldc_train_block_backward(self.eng._ctx, dy.data_ptr(), x.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(),
                                                  ss.data_ptr() if ss is not None else None, B, Cin, Cout, Lx, self.groups, ws.data_ptr(),
                                                  dx.data_ptr(), dw.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr(),
                                                  dss.data_ptr() if dss is not None else None, s))
        self.eng._exit()
        out = {"dx": dx, "dw": dw, "db": db, "dgamma": dg, "dbeta": dbt}
        if dss is not None:
            out["dscale"], out["dshift"] = dss[:, :Cout].reshape(B, Cout, 1), dss[:, Cout:].reshape(B, Cout, 1)
        return out


def _keep_for_side(eng, *tensors):
    """A backward call that put a parameter gradient on the library's side stream (option train_dw_side, set by DiffusionTrainer around its
    backward pass) still reads these tensors there: tell torch's caching allocator, so that their memory is not handed out again before
    the side stream is through (the layers' saved inputs live until their next forward and need no mark)."""
    ext = getattr(eng, "_dw_side_ext", None)
    if ext is not None:
        for v in tensors:
            if v is not None:
                v.record_stream(ext)


class _SideRegion:
    """`with _SideRegion(eng):` -- the layer calls and the ATen glue inside run on the library's training side stream (when the trainer has
    switched it on; otherwise in line): the backward of the time-embedding branch feeds nothing but parameter gradients, so it need not sit
    on the dX chain.  The side stream first waits for what the main stream has produced so far; DiffusionTrainer joins the two after the
    backward pass (ldc_train_join).  Launches on the side stream take their own workspaces in the library (csrc/train_mm3.hip: lane_of)."""

    def __init__(self, eng):
        self.eng, self.ext = eng, getattr(eng, "_dw_side_ext", None)

    def __enter__(self):
        if self.ext is not None:
            self.main = self.eng.stream
            self.ext.wait_stream(self.main)
            self.eng.stream = self.ext
            self.ctx = self.eng.torch.cuda.stream(self.ext)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ext is not None:
            self.ctx.__exit__(*exc)
            self.eng.stream = self.main
        return False


def q_sample(eng, x_start, t, noise):
    """diffusion.q_sample (ddpm_loss.py:386-392)."""
    tt = eng.torch
    x_start, noise = eng._f32(x_start), eng._f32(noise)
    t = t.to(eng.device, tt.int64).contiguous()
    B, Cc, Lx = x_start.shape
    out = tt.empty_like(x_start)
    s = eng._enter()
    L.check(eng.lib.ldc_train_q_sample(eng._ctx, x_start.data_ptr(), t.data_ptr(), noise.data_ptr(), B, Cc, Lx, out.data_ptr(), s))
    eng._exit()
    return out


def p_losses_objective(eng, model_out, target, t, want_grad: bool = True):
    """The loss of diffusion.p_losses (ddpm_loss.py:434-438, l1) and its gradient w.r.t. the model output."""
    tt = eng.torch
    model_out, target = eng._f32(model_out), eng._f32(target)
    t = t.to(eng.device, tt.int64).contiguous()
    B, Cc, Lx = model_out.shape
    loss = tt.empty(1, dtype=tt.float32, device=eng.device)
    grad = tt.empty_like(model_out) if want_grad else None
    s = eng._enter()
    L.check(eng.lib.ldc_train_l1_loss(eng._ctx, model_out.data_ptr(), target.data_ptr(), t.data_ptr(), B, Cc, Lx, loss.data_ptr(),
                                      grad.data_ptr() if grad is not None else None, s))
    eng._exit()
    return (loss, grad) if want_grad else loss


class LayerNorm:
    """Channel LayerNorm of the attention blocks (srcs/modules/unet.py:82-101) with its backward pass."""

    def __init__(self, eng, g):
        self.eng, self.lib, self.torch = eng, eng.lib, eng.torch
        self.g = g.to(eng.device, eng.torch.float32).reshape(-1).contiguous()

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, Cc, Lx = x.shape
        y = t.empty_like(x)
        stats = t.empty(B, Lx, 2, dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_layernorm_forward(self.eng._ctx, x.data_ptr(), self.g.data_ptr(), B, Cc, Lx, y.data_ptr(), stats.data_ptr(), s))
        self.eng._exit()
        self._saved = (x, stats)
        return y

    def backward(self, dy):
        """-> (dx, dg)"""
        t = self.torch
        x, stats = self._saved
        B, Cc, Lx = x.shape
        dy = dy.to(self.eng.device, t.float32).contiguous()
        dx, dg = t.empty_like(x), grad_buffer(self.eng, self.g)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_layernorm_backward(self.eng._ctx, dy.data_ptr(), x.data_ptr(), self.g.data_ptr(), stats.data_ptr(), B, Cc, Lx,
                                                      dx.data_ptr(), dg.data_ptr(), s))
        self.eng._exit()
        _keep_for_side(self.eng, dy)
        return dx, dg


class Adam:
    """optim.Adam(params, lr) of srcs/train.py:365-371 over ONE flat fp32 parameter buffer (the layout parallel.allreduce_gradients
    reduces): state and update live on the device, `step(grad)` updates `param` in place."""

    def __init__(self, eng, param, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        t = eng.torch
        assert param.dtype == t.float32 and param.is_contiguous() and param.device.type == "cuda"
        self.eng, self.param, self.lr, self.betas, self.eps = eng, param, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg, self.exp_avg_sq = t.zeros_like(param), t.zeros_like(param)
        self.steps = 0
        self.step_dev = None      # device-side step count (int32[1]) once the optimisation step is replayed from a hipGraph

    def use_device_count(self):
        """From here on the step count lives on the device (a captured step: a host-side count would be frozen into the graph)."""
        t = self.eng.torch
        if self.step_dev is None:
            self.step_dev = t.tensor([self.steps], dtype=t.int32, device=self.param.device)

    def step(self, grad):
        t = self.eng.torch
        grad = grad.to(self.param.device, t.float32).contiguous()
        assert grad.numel() == self.param.numel()
        self.steps += 1
        if self.step_dev is not None:
            s = self.eng._enter()
            L.check(self.eng.lib.ldc_train_adam_step_dev(self.eng._ctx, self.param.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                                                         self.exp_avg_sq.data_ptr(), self.param.numel(), self.step_dev.data_ptr(), self.lr,
                                                         self.betas[0], self.betas[1], self.eps, s))
            self.eng._exit()
            return self.param
        s = self.eng._enter()
        L.check(self.eng.lib.ldc_train_adam_step(self.eng._ctx, self.param.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                                                 self.exp_avg_sq.data_ptr(), self.param.numel(), self.steps, self.lr, self.betas[0],
                                                 self.betas[1], self.eps, s))
        self.eng._exit()
        return self.param


class Pointwise:
    """y = W a(x) + b on [B, C, L] (a = identity | SiLU): the 1x1 res_conv and the time-embedding MLP of a ResnetBlock."""

    def __init__(self, eng, weight, bias, pre_silu: bool = False):
        t = eng.torch
        self.eng, self.lib, self.torch = eng, eng.lib, t
        self.weight = weight.to(eng.device, t.float32).reshape(weight.shape[0], -1).contiguous()      # [Cout, Cin]
        self.bias = bias.to(eng.device, t.float32).contiguous() if bias is not None else None
        self.pre_silu = bool(pre_silu)

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        x3 = x if x.dim() == 3 else x.reshape(x.shape[0], x.shape[1], 1)
        B, Cin, Lx = x3.shape
        Cout = self.weight.shape[0]
        y = t.empty(B, Cout, Lx, dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_pointwise_forward(self.eng._ctx, x3.data_ptr(), self.weight.data_ptr(),
                                                     self.bias.data_ptr() if self.bias is not None else None, B, Cin, Cout, Lx,
                                                     int(self.pre_silu), y.data_ptr(), s))
        self.eng._exit()
        self._saved = (x3, x.dim())
        return y if x.dim() == 3 else y.reshape(B, Cout)

    def backward(self, dy, want_dx: bool = True):
        """-> dict(dx, dw, db)"""
        t = self.torch
        x3, nd = self._saved
        B, Cin, Lx = x3.shape
        Cout = self.weight.shape[0]
        dy = dy.to(self.eng.device, t.float32).contiguous().reshape(B, Cout, Lx)
        dx = t.empty_like(x3) if want_dx else None
        dw = grad_buffer(self.eng, self.weight)
        db = grad_buffer(self.eng, self.bias) if self.bias is not None else None
        s = self.eng._enter()
        L.check(self.lib.ldc_train_pointwise_backward(self.eng._ctx, dy.data_ptr(), x3.data_ptr(), self.weight.data_ptr(), B, Cin, Cout, Lx,
                                                      int(self.pre_silu), dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                                      db.data_ptr() if db is not None else None, s))
        self.eng._exit()
        _keep_for_side(self.eng, dy)
        out = {"dw": dw}
        if db is not None:
            out["db"] = db
        if dx is not None:
            out["dx"] = dx if nd == 3 else dx.reshape(B, Cin)
        return out


class ResnetBlock:
    """ResnetBlock.forward (srcs/modules/unet.py:157-192) and its backward pass, composed of the slice's pieces:
    time_emb -> SiLU -> Linear -> (scale, shift); Block1(x; scale, shift) -> Block2 -> + res_conv(x)."""

    def __init__(self, eng, p: dict, groups: int = 8):
        """p: the block's state-dict entries: mlp.1.weight/bias, block1|block2.proj.weight/bias, block1|block2.norm.weight/bias,
        optionally res_conv.weight/bias."""
        self.eng, self.torch = eng, eng.torch
        self.mlp = Pointwise(eng, p["mlp.1.weight"], p["mlp.1.bias"], pre_silu=True)
        self.block1 = Block(eng, p["block1.proj.weight"], p["block1.proj.bias"], p["block1.norm.weight"], p["block1.norm.bias"], groups)
        self.block2 = Block(eng, p["block2.proj.weight"], p["block2.proj.bias"], p["block2.norm.weight"], p["block2.norm.bias"], groups)
        self.res = Pointwise(eng, p["res_conv.weight"], p["res_conv.bias"]) if "res_conv.weight" in p else None

    def precompute_scale_shift(self, time_emb):
        """The block's (scale, shift) = mlp(time_emb) ahead of its forward (Unet1D.forward runs the whole time-embedding branch on the side
        stream at the start of a training step); an event marks it ready for the stream the block itself runs on."""
        ss = self.mlp.forward(time_emb)
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.eng.device))
        self._ss_pre = (ss, ev)

    def forward(self, x, time_emb):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        pre, self._ss_pre = getattr(self, "_ss_pre", None), None
        if pre is not None:
            ss, ev = pre
            cur = t.cuda.current_stream(self.eng.device)
            cur.wait_event(ev)
            ss.record_stream(cur)
        else:
            ss = self.mlp.forward(time_emb)                                 # [B, 2 * Cout]
        cout = ss.shape[1] // 2
        scale, shift = ss[:, :cout].reshape(-1, cout, 1), ss[:, cout:].reshape(-1, cout, 1)
        h = self.block1.forward(x, (scale, shift))
        h = self.block2.forward(h)
        return h + (self.res.forward(x) if self.res is not None else x)

    def backward(self, dy):
        """-> dict of gradients keyed like the parameters, plus dx and dtime_emb"""
        t = self.torch
        dy = dy.to(self.eng.device, t.float32).contiguous()
        g2 = self.block2.backward(dy)
        g1 = self.block1.backward(g2["dx"])
        _keep_for_side(self.eng, g1["dscale"], g1["dshift"])
        with _SideRegion(self.eng):      # (the time-embedding branch: parameter gradients only)
            dss = t.cat([g1["dscale"].reshape(dy.shape[0], -1), g1["dshift"].reshape(dy.shape[0], -1)], dim=1)
            gm = self.mlp.backward(dss)
        out = {"mlp.1.weight": gm["dw"], "mlp.1.bias": gm["db"], "dtime_emb": gm["dx"],
               "block1.proj.weight": g1["dw"], "block1.proj.bias": g1["db"], "block1.norm.weight": g1["dgamma"], "block1.norm.bias": g1["dbeta"],
               "block2.proj.weight": g2["dw"], "block2.proj.bias": g2["db"], "block2.norm.weight": g2["dgamma"], "block2.norm.bias": g2["dbeta"]}
        dx = g1["dx"]
        if self.res is not None:
            gr = self.res.backward(dy)
            out["res_conv.weight"], out["res_conv.bias"] = gr["dw"].reshape(gr["dw"].shape[0], -1, 1), gr["db"]
            dx = dx + gr["dx"]
        else:
            dx = dx + dy
        out["dx"] = dx
        return out


class LinearAttention:
    """Residual(PreNorm(dim, LinearAttention(dim))) of the UNet (srcs/modules/unet.py:194-222 inside :103-116's wrappers),
    forward and backward: LayerNorm -> to_qkv (1x1, no bias) -> attention core -> to_out (1x1 + LayerNorm) -> + x."""

    def __init__(self, eng, p: dict, heads: int = 4, dim_head: int = 32):
        """p: 'norm.g' (PreNorm), 'to_qkv.weight', 'to_out.0.weight', 'to_out.0.bias', 'to_out.1.g'."""
        self.eng, self.lib, self.torch = eng, eng.lib, eng.torch
        self.heads, self.dim_head = heads, dim_head
        self.norm = LayerNorm(eng, p["norm.g"])
        self.to_qkv = Pointwise(eng, p["to_qkv.weight"], None)
        self.to_out = Pointwise(eng, p["to_out.0.weight"], p["to_out.0.bias"])
        self.out_norm = LayerNorm(eng, p["to_out.1.g"])

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, _, N = x.shape
        qkv = self.to_qkv.forward(self.norm.forward(x))
        hd = self.heads * self.dim_head
        o = t.empty(B, hd, N, dtype=t.float32, device=self.eng.device)
        ws = t.empty(int(self.lib.ldc_train_linattn_ws_floats(B, self.heads, self.dim_head, N)), dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_linattn_forward(self.eng._ctx, qkv.data_ptr(), B, self.heads, self.dim_head, N, o.data_ptr(), ws.data_ptr(), s))
        self.eng._exit()
        self._saved = (qkv, ws)
        return self.out_norm.forward(self.to_out.forward(o)) + x

    def backward(self, dy):
        """-> dict of parameter gradients plus dx"""
        t = self.torch
        dy = dy.to(self.eng.device, t.float32).contiguous()
        qkv, ws = self._saved
        B, _, N = qkv.shape
        d_lin, dg_out = self.out_norm.backward(dy)
        g_out = self.to_out.backward(d_lin)
        dqkv = t.empty_like(qkv)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_linattn_backward(self.eng._ctx, g_out["dx"].data_ptr(), qkv.data_ptr(), B, self.heads, self.dim_head, N,
                                                    ws.data_ptr(), dqkv.data_ptr(), s))
        self.eng._exit()
        g_qkv = self.to_qkv.backward(dqkv)
        dxn, dg_pre = self.norm.backward(g_qkv["dx"])
        return {"norm.g": dg_pre, "to_qkv.weight": g_qkv["dw"], "to_out.0.weight": g_out["dw"], "to_out.0.bias": g_out["db"],
                "to_out.1.g": dg_out, "dx": dxn + dy}


class Conv1d:
    """nn.Conv1d(Cin, Cout, K, stride, padding) forward / backward (init_conv, Downsample, the k3 convs, final_conv)."""

    def __init__(self, eng, weight, bias, stride: int = 1, padding: int = 0):
        t = eng.torch
        self.eng, self.lib, self.torch = eng, eng.lib, t
        self.weight = weight.to(eng.device, t.float32).contiguous()
        self.bias = bias.to(eng.device, t.float32).contiguous() if bias is not None else None
        self.stride, self.padding = int(stride), int(padding)

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, Cin, Lin = x.shape
        Cout, _, K = self.weight.shape
        Lout = (Lin + 2 * self.padding - K) // self.stride + 1
        y = t.empty(B, Cout, Lout, dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_conv_forward(self.eng._ctx, x.data_ptr(), self.weight.data_ptr(), self.bias.data_ptr() if self.bias is not None else None,
                                                B, Cin, Cout, Lin, K, self.stride, self.padding, y.data_ptr(), s))
        self.eng._exit()
        self._saved = x
        return y

    def backward(self, dy, want_dx: bool = True):
        t = self.torch
        x = self._saved
        B, Cin, Lin = x.shape
        Cout, _, K = self.weight.shape
        dy = dy.to(self.eng.device, t.float32).contiguous()
        dx = t.empty_like(x) if want_dx else None
        dw = grad_buffer(self.eng, self.weight)
        db = grad_buffer(self.eng, self.bias) if self.bias is not None else None
        s = self.eng._enter()
        L.check(self.lib.ldc_train_conv_backward(self.eng._ctx, dy.data_ptr(), x.data_ptr(), self.weight.data_ptr(), B, Cin, Cout, Lin, K, self.stride,
                                                 self.padding, dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                                 db.data_ptr() if db is not None else None, s))
        self.eng._exit()
        _keep_for_side(self.eng, dy)
        return {"dx": dx, "dw": dw, "db": db}


def upsample2(eng, x, backward: bool = False):
    """nn.Upsample(scale_factor=2, mode='nearest') on [B, C, L] and its adjoint (dy [B, C, 2L] -> dx [B, C, L])."""
    t = eng.torch
    x = x.to(eng.device, t.float32).contiguous()
    B, Cc, Lx = x.shape
    Lin = Lx // 2 if backward else Lx
    out = t.empty(B, Cc, Lin if backward else 2 * Lx, dtype=t.float32, device=eng.device)
    s = eng._enter()
    L.check(eng.lib.ldc_train_upsample2(eng._ctx, x.data_ptr(), B * Cc, Lin, int(backward), out.data_ptr(), s))
    eng._exit()
    return out


ACT_TANH, ACT_GELU, ACT_SILU = 0, 1, 2


def activation(eng, x, kind: int, dy=None):
    """y = f(x), or with dy: dx = dy * f'(x)."""
    t = eng.torch
    x = x.to(eng.device, t.float32).contiguous()
    out = t.empty_like(x)
    dyc = dy.to(eng.device, t.float32).contiguous() if dy is not None else None
    s = eng._enter()
    L.check(eng.lib.ldc_train_activation(eng._ctx, x.data_ptr(), dyc.data_ptr() if dyc is not None else None, x.numel(), int(kind), out.data_ptr(), s))
    eng._exit()
    return out


class Attention:
    """Residual(PreNorm(dim, Attention(dim))) of the bottleneck (unet.py:224-246 inside :103-116's wrappers)."""

    def __init__(self, eng, p: dict, heads: int = 4, dim_head: int = 32):
        """p: 'norm.g', 'to_qkv.weight', 'to_out.weight', 'to_out.bias'."""
        self.eng, self.lib, self.torch = eng, eng.lib, eng.torch
        self.heads, self.dim_head = heads, dim_head
        self.norm = LayerNorm(eng, p["norm.g"])
        self.to_qkv = Pointwise(eng, p["to_qkv.weight"], None)
        self.to_out = Pointwise(eng, p["to_out.weight"], p["to_out.bias"])

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, _, N = x.shape
        qkv = self.to_qkv.forward(self.norm.forward(x))
        o = t.empty(B, self.heads * self.dim_head, N, dtype=t.float32, device=self.eng.device)
        ws = t.empty(int(self.lib.ldc_train_attn_ws_floats(B, self.heads, N)), dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_attn_forward(self.eng._ctx, qkv.data_ptr(), B, self.heads, self.dim_head, N, o.data_ptr(), ws.data_ptr(), s))
        self.eng._exit()
        self._saved = (qkv, ws)
        return self.to_out.forward(o) + x

    def backward(self, dy):
        t = self.torch
        dy = dy.to(self.eng.device, t.float32).contiguous()
        qkv, ws = self._saved
        B, _, N = qkv.shape
        g_out = self.to_out.backward(dy)
        dqkv = t.empty_like(qkv)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_attn_backward(self.eng._ctx, g_out["dx"].data_ptr(), qkv.data_ptr(), B, self.heads, self.dim_head, N, ws.data_ptr(),
                                                 dqkv.data_ptr(), s))
        self.eng._exit()
        g_qkv = self.to_qkv.backward(dqkv)
        dxn, dg_pre = self.norm.backward(g_qkv["dx"])
        return {"norm.g": dg_pre, "to_qkv.weight": g_qkv["dw"], "to_out.weight": g_out["dw"], "to_out.bias": g_out["db"], "dx": dxn + dy}


class ConvTranspose1d:
    """SConvTranspose1d(C, C, kernel 2r, stride r, non-causal) of the condition upsampler (unet.py:372-377, conv.py:235-274)."""

    def __init__(self, eng, weight, bias, ratio: int):
        t = eng.torch
        self.eng, self.lib, self.torch, self.ratio = eng, eng.lib, t, int(ratio)
        self.weight = weight.to(eng.device, t.float32).contiguous()          # [Cin, Cout, 2r]
        self.bias = bias.to(eng.device, t.float32).contiguous() if bias is not None else None
        assert self.weight.shape[2] == 2 * self.ratio

    def forward(self, x):
        t = self.torch
        x = x.to(self.eng.device, t.float32).contiguous()
        B, Cin, Lx = x.shape
        Cout = self.weight.shape[1]
        y = t.empty(B, Cout, Lx * self.ratio, dtype=t.float32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_train_convtr_forward(self.eng._ctx, x.data_ptr(), self.weight.data_ptr(), self.bias.data_ptr() if self.bias is not None else None,
                                                  B, Cin, Cout, Lx, self.ratio, y.data_ptr(), s))
        self.eng._exit()
        self._saved = x
        return y

    def backward(self, dy):
        t = self.torch
        x = self._saved
        B, Cin, Lx = x.shape
        Cout = self.weight.shape[1]
        dy = dy.to(self.eng.device, t.float32).contiguous()
        dx, dw = t.empty_like(x), grad_buffer(self.eng, self.weight)
        db = grad_buffer(self.eng, self.bias) if self.bias is not None else None
        s = self.eng._enter()
        L.check(self.lib.ldc_train_convtr_backward(self.eng._ctx, dy.data_ptr(), x.data_ptr(), self.weight.data_ptr(), B, Cin, Cout, Lx, self.ratio,
                                                   dx.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, s))
        self.eng._exit()
        return {"dx": dx, "dw": dw, "db": db}


def maxscale(eng, x, dy=None):
    """Unet1D.scaling (unet.py:401-403): x / (max|x| per item + 1e-20); with dy: the gradient w.r.t. x."""
    t = eng.torch
    x = x.to(eng.device, t.float32).contiguous()
    out = t.empty_like(x)
    dyc = dy.to(eng.device, t.float32).contiguous() if dy is not None else None
    s = eng._enter()
    L.check(eng.lib.ldc_train_maxscale(eng._ctx, x.data_ptr(), dyc.data_ptr() if dyc is not None else None, x.shape[0], x.numel() // x.shape[0],
                                       out.data_ptr(), s))
    eng._exit()
    return out


class _Sub:
    """the entries of a state dict under a prefix, read on demand (so that the set of keys a network actually uses can be recorded)"""

    def __init__(self, sd, prefix, alias=None):
        self.sd, self.prefix, self.alias = sd, prefix, alias or {}

    def _key(self, k):
        return self.alias.get(k, self.prefix + k)

    def __getitem__(self, k):
        return self.sd[self._key(k)]

    def __contains__(self, k):
        return self._key(k) in self.sd


class Unet1D:
    """Unet1D.forward (srcs/modules/unet.py:422-469) and its backward pass over the reference's own state dict
    (`other_cond` layout: x_cond is concatenated in front of x; process_cond's upsampler / scaling are applied by the caller).
    fp32 correctness path of the training step: every layer is one of this module's forward/backward pairs."""

    def __init__(self, eng, sd: dict, dim: int, dim_mults=(1, 2, 4, 8), heads: int = 4, dim_head: int = 32, groups: int = 8,
                 upsampling_ratios=None, unet_scale_cond: bool = False):
        self.eng, self.torch, self.dim = eng, eng.torch, dim
        # process_cond (unet.py:407-420): the upsampling layers are parameters of diff_model and are trained with it
        self.upsampling = [ConvTranspose1d(eng, sd[f"upsampling_layers.{i}.convtr.convtr.weight"], sd[f"upsampling_layers.{i}.convtr.convtr.bias"], r)
                           for i, r in enumerate(upsampling_ratios or ())]
        self.scale_cond = bool(unet_scale_cond)
        sub = lambda prefix: _Sub(sd, prefix)
        att = lambda prefix: _Sub(sd, prefix + "fn.fn.", {"norm.g": prefix + "fn.norm.g"})
        self.init_conv = Conv1d(eng, sd["init_conv.weight"], sd["init_conv.bias"], 1, 3)
        self.t1 = Pointwise(eng, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
        self.t2 = Pointwise(eng, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
        n = len(dim_mults)
        self.downs, self.ups = [], []
        for i in range(n):
            last = i == n - 1
            res = Conv1d(eng, sd[f"downs.{i}.3.weight"], sd[f"downs.{i}.3.bias"], 1 if last else 2, 1)
            self.downs.append((ResnetBlock(eng, sub(f"downs.{i}.0."), groups), ResnetBlock(eng, sub(f"downs.{i}.1."), groups),
                               LinearAttention(eng, att(f"downs.{i}.2."), heads, dim_head), res))
        self.mid1 = ResnetBlock(eng, sub("mid_block1."), groups)
        self.mid_attn = Attention(eng, att("mid_attn."), heads, dim_head)
        self.mid2 = ResnetBlock(eng, sub("mid_block2."), groups)
        for i in range(n):
            last = i == n - 1
            key = f"ups.{i}.3." if last else f"ups.{i}.3.1."
            self.ups.append((ResnetBlock(eng, sub(f"ups.{i}.0."), groups), ResnetBlock(eng, sub(f"ups.{i}.1."), groups),
                             LinearAttention(eng, att(f"ups.{i}.2."), heads, dim_head), Conv1d(eng, sd[key + "weight"], sd[key + "bias"], 1, 1), not last))
        self.final_res = ResnetBlock(eng, sub("final_res_block."), groups)
        self.final_conv = Conv1d(eng, sd["final_conv.weight"], sd["final_conv.bias"], 1, 0)

    def _sinusoidal(self, time):
        """SinusoidalPosEmb (unet.py:104-116): parameter-free."""
        import math
        t = self.torch
        half = self.dim // 2
        emb = t.exp(t.arange(half, device=self.eng.device, dtype=t.float32) * -(math.log(10000) / (half - 1)))
        emb = time.to(self.eng.device, t.float32)[:, None] * emb[None, :]
        return t.cat((emb.sin(), emb.cos()), dim=-1).contiguous()

    def forward(self, x, time, x_cond):
        t = self.torch
        self._ccond = x_cond.shape[1]
        for layer in self.upsampling:
            x_cond = layer.forward(x_cond)
        if self.scale_cond:
            self._cond_pre_scale = x_cond.to(self.eng.device, t.float32).contiguous()
            x_cond = maxscale(self.eng, self._cond_pre_scale)
        x = t.cat((x_cond.to(self.eng.device, t.float32), x.to(self.eng.device, t.float32)), dim=1).contiguous()
        if getattr(self.eng, "_dw_side_ext", None) is not None:
            # training step: the time-embedding branch (time_mlp and the 23 per-block SiLU -> Linear maps) does not depend on x: all of it on
            # the side stream now, under init_conv and the first blocks; every ResnetBlock waits for its own (scale, shift)
            with _SideRegion(self.eng):
                self._t_pre = self.t1.forward(self._sinusoidal(time))
                temb = self.t2.forward(activation(self.eng, self._t_pre, ACT_GELU))
                blocks = [b for rb1, rb2, _, _ in self.downs for b in (rb1, rb2)] + [self.mid1, self.mid2]
                blocks += [b for rb1, rb2, _, _, _ in self.ups for b in (rb1, rb2)] + [self.final_res]
                for rb in blocks:
                    rb.precompute_scale_shift(temb)
            x = self.init_conv.forward(x)
            r = x
        else:
            x = self.init_conv.forward(x)
            r = x
            self._t_pre = self.t1.forward(self._sinusoidal(time))
            temb = self.t2.forward(activation(self.eng, self._t_pre, ACT_GELU))
        h = []
        for rb1, rb2, attn, down in self.downs:
            x = rb1.forward(x, temb); h.append(x)
            x = rb2.forward(x, temb)
            x = attn.forward(x); h.append(x)
            x = down.forward(x)
        x = self.mid1.forward(x, temb)
        x = self.mid_attn.forward(x)
        x = self.mid2.forward(x, temb)
        self._cat_splits = []
        for rb1, rb2, attn, conv, up in self.ups:
            skip = h.pop(); self._cat_splits.append(x.shape[1]); x = rb1.forward(t.cat((x, skip), dim=1), temb)
            skip = h.pop(); self._cat_splits.append(x.shape[1]); x = rb2.forward(t.cat((x, skip), dim=1), temb)
            x = attn.forward(x)
            x = conv.forward(upsample2(self.eng, x) if up else x)
        self._final_split = x.shape[1]
        x = self.final_res.forward(t.cat((x, r), dim=1), temb)
        self._pre_tanh = x
        return self.final_conv.forward(activation(self.eng, x, ACT_TANH))

    def backward(self, dy):
        """-> (grads keyed like the state dict, dx of the noisy input, dx_cond)"""
        t = self.torch
        grads = {}
        dtemb = None

        def rb_back(rb, prefix, d):
            nonlocal dtemb
            g = rb.backward(d)
            with _SideRegion(self.eng):
                dtemb = g["dtime_emb"] if dtemb is None else dtemb + g["dtime_emb"]
            for k, v in g.items():
                if k not in ("dx", "dtime_emb"):
                    grads[prefix + k] = v
            return g["dx"]

        def att_back(a, prefix, d, linear=True):
            g = a.backward(d)
            grads[prefix + "fn.norm.g"] = g["norm.g"]
            for k, v in g.items():
                if k not in ("dx", "norm.g"):
                    grads[prefix + "fn.fn." + k] = v
            return g["dx"]

        g = self.final_conv.backward(dy)
        grads["final_conv.weight"], grads["final_conv.bias"] = g["dw"], g["db"]
        d = activation(self.eng, self._pre_tanh, ACT_TANH, dy=g["dx"])
        d = rb_back(self.final_res, "final_res_block.", d)
        d, dr = d[:, :self._final_split].contiguous(), d[:, self._final_split:].contiguous()
        dh = []                                    # gradients of the skip tensors, in the order they were popped
        n = len(self.ups)
        for i in reversed(range(n)):
            rb1, rb2, attn, conv, up = self.ups[i]
            key = f"ups.{i}.3.1." if up else f"ups.{i}.3."
            g = conv.backward(d)
            grads[key + "weight"], grads[key + "bias"] = g["dw"], g["db"]
            d = upsample2(self.eng, g["dx"], backward=True) if up else g["dx"]
            d = att_back(attn, f"ups.{i}.2.", d)
            d = rb_back(rb2, f"ups.{i}.1.", d)
            c2 = self._cat_splits[2 * i + 1]
            d, ds2 = d[:, :c2].contiguous(), d[:, c2:].contiguous()
            d = rb_back(rb1, f"ups.{i}.0.", d)
            c1 = self._cat_splits[2 * i]
            d, ds1 = d[:, :c1].contiguous(), d[:, c1:].contiguous()
            dh = [ds1, ds2] + dh                   # this level popped ds1's tensor first, then ds2's
        d = rb_back(self.mid2, "mid_block2.", d)
        g = self.mid_attn.backward(d)
        grads["mid_attn.fn.norm.g"] = g["norm.g"]
        for k in ("to_qkv.weight", "to_out.weight", "to_out.bias"):
            grads["mid_attn.fn.fn." + k] = g[k]
        d = rb_back(self.mid1, "mid_block1.", g["dx"])
        # dh holds, for ups level 0..n-1, (grad of the tensor popped first, grad of the one popped second); the pops ran from
        # the END of h: h = [d0.a, d0.b, d1.a, d1.b, ...] (a: after block1, b: after attention) -> ups level 0 popped d_{n-1}.b, d_{n-1}.a
        for i in reversed(range(len(self.downs))):
            rb1, rb2, attn, down = self.downs[i]
            lvl = len(self.downs) - 1 - i                       # the ups level that consumed this level's skips
            d_b, d_a = dh[2 * lvl], dh[2 * lvl + 1]
            g = down.backward(d)
            grads[f"downs.{i}.3.weight"], grads[f"downs.{i}.3.bias"] = g["dw"], g["db"]
            d = g["dx"] + d_b
            d = att_back(attn, f"downs.{i}.2.", d)
            d = rb_back(rb2, f"downs.{i}.1.", d)
            d = d + d_a
            d = rb_back(rb1, f"downs.{i}.0.", d)
        d = d + dr
        g = self.init_conv.backward(d)
        grads["init_conv.weight"], grads["init_conv.bias"] = g["dw"], g["db"]
        with _SideRegion(self.eng):
            g2 = self.t2.backward(dtemb)
            grads["time_mlp.3.weight"], grads["time_mlp.3.bias"] = g2["dw"], g2["db"]
            g1 = self.t1.backward(activation(self.eng, self._t_pre, ACT_GELU, dy=g2["dx"]), want_dx=False)
            grads["time_mlp.1.weight"], grads["time_mlp.1.bias"] = g1["dw"], g1["db"]
        cc = self._ccond
        dx, dcond = g["dx"][:, cc:].contiguous(), g["dx"][:, :cc].contiguous()
        if self.scale_cond:
            dcond = maxscale(self.eng, self._cond_pre_scale, dy=dcond)
        for i in reversed(range(len(self.upsampling))):
            gu = self.upsampling[i].backward(dcond)
            grads[f"upsampling_layers.{i}.convtr.convtr.weight"], grads[f"upsampling_layers.{i}.convtr.convtr.bias"] = gu["dw"], gu["db"]
            dcond = gu["dx"]
        return grads, dx, dcond


def predict_x_start(eng, x_t, eps, t):
    """predicted_x_start of p_losses (ddpm_loss.py:416-420 -> predict_start_from_noise, :175-179; not clamped on this path)."""
    tt = eng.torch
    x_t, eps = eng._f32(x_t), eng._f32(eps)
    t = t.to(eng.device, tt.int64).contiguous()
    B, Cc, Lx = x_t.shape
    out = tt.empty_like(x_t)
    s = eng._enter()
    L.check(eng.lib.ldc_train_predict_x_start(eng._ctx, x_t.data_ptr(), eps.data_ptr(), t.data_ptr(), B, Cc, Lx, out.data_ptr(), s))
    eng._exit()
    return out


def neg_sdsdr(eng, est, tgt, clip_min: float = -30.0):
    """sdr_loss of the reference (ClippedSDR, losses_fn.py:56-66, over asteroid's MultiSrcNegSDR('sdsdr')) per item; the
    reference calls it as sdr_loss(x, x_hat) (model.py:194).  est, tgt: [B, 1, T]."""
    tt = eng.torch
    est, tgt = eng._f32(est), eng._f32(tgt)
    B = est.shape[0]
    out = tt.empty(B, dtype=tt.float32, device=eng.device)
    s = eng._enter()
    L.check(eng.lib.ldc_train_neg_sdsdr(eng._ctx, est.data_ptr(), tgt.data_ptr(), B, est.numel() // B, float(clip_min), out.data_ptr(), s))
    eng._exit()
    return out


class DiffusionTrainer:
    """One optimisation step of the diffusion UNet as srcs/train.py:110-177 runs it for --run_diff (the codec is frozen, only
    model.diff_model's parameters are optimised, train.py:365): q_sample -> Unet1D forward -> p_losses objective -> Unet1D backward
    -> gradient averaging over ranks (one flat reduce-scatter + all-gather) -> Adam.  `x_start` is the scaled latent
    (model.py:165) produced by the inference kernels, `cond` the condition of model_for_cond.get_cond: with `upsampling_ratios` /
    `unet_scale_cond` given, Unet1D.process_cond and the upsampler's own parameters are part of the step.

    The parameters live in ONE flat fp32 buffer, the network object is built once over views of it (Adam updates them in
    place), and every layer's backward writes its parameter gradients straight into views of ONE flat gradient buffer -- the
    buffers the gradient reduction and Adam work on (until round 3 the network was rebuilt and 135 M gradient elements were
    concatenated every step)."""

    def __init__(self, eng, sd: dict, dim: int, dim_mults=(1, 2, 4, 8), lr: float = 1e-4, frontend=None, **kw):
        """kw: heads, dim_head, groups, upsampling_ratios, unet_scale_cond of Unet1D (with upsampling_ratios the raw condition is passed to step).
        frontend: a second Engine holding the same (frozen) codec weights.  With it, `step_from_wav(wav, next_wav=...)` runs the two
        frozen encoders of the NEXT batch on a side stream while this batch's UNet forward / backward occupy the main one (they
        do not depend on the parameters being optimised; the LSTM recurrences are latency-bound and use a fraction of the CUs)."""
        t = eng.torch
        self.eng, self.torch = eng, t
        self.frontend = frontend
        self._side = t.cuda.Stream(device=eng.device) if frontend is not None else None
        self._prefetched = None        # (wav, cond, x_rep (unscaled), event)
        self.dim, self.dim_mults, self.kw = dim, tuple(dim_mults), kw
        # the trainable parameters are what Unet1D consumes for this configuration (not every key of the state dict: buffers,
        # or the upsampling layers when upsampling_ratios is None, are not parameters of the step)
        probe = _KeyRecorder(sd)
        Unet1D(eng, probe, dim, self.dim_mults, **kw)
        self.names = sorted(probe.used)
        self.shapes = {k: tuple(sd[k].shape) for k in self.names}
        self.flat = t.cat([sd[k].to(eng.device, t.float32).reshape(-1) for k in self.names]).contiguous()
        self.flat_g = t.zeros_like(self.flat)
        self.net = Unet1D(eng, self.state_dict(), dim, self.dim_mults, **kw)          # layers alias views of self.flat
        self._grad_views = {p.data_ptr(): g for p, g in zip(self.state_dict().values(), self._views(self.flat_g).values())}
        self.num_timesteps = int(eng.lib.ldc_train_num_timesteps(eng._ctx))
        self.opt = Adam(eng, self.flat, lr=lr)
        self.dw_side = True        # parameter gradients (weight-gradient GEMMs, their reductions, norm gains) on the library's side stream, under the dX chain (round 6: 48.5 -> 41.9 ms per full-width step, same box; bit-identical gradients)
        self._side_ext = None
        self.use_graph = False     # the step as one replayed hipGraph (see _step_graphed): opt-in attribute, measured equal (49.96 vs 49.75 ms)
        self._graph, self._graph_key, self._graph_seen, self._graph_in, self._graph_out = None, None, 0, None, None

    def _views(self, flat):
        out, off = {}, 0
        for k in self.names:
            n = 1
            for d in self.shapes[k]:
                n *= d
            out[k] = flat[off:off + n].view(self.shapes[k])
            off += n
        return out

    def state_dict(self):
        return self._views(self.flat)

    def gradients(self):
        """views of the flat gradient buffer, keyed like the state dict (valid after `step`)"""
        return self._views(self.flat_g)

    def _on_engine_stream(self, fn):
        """Run `fn` with the engine's stream as torch's current stream: every layer call then launches on the stream it is already on
        (no event record / wait pair per call to hop streams: ~3 200 of them per step otherwise) and the ATen glue runs there too.  The
        caller's stream waits for the result; returned tensors are marked as used on it."""
        tt = self.torch
        dev = self.eng.device
        outer = tt.cuda.current_stream(dev)
        es = self.eng.stream
        if outer.cuda_stream == es.cuda_stream:
            return fn()
        es.wait_stream(outer)
        with tt.cuda.stream(es):
            out = fn()
        outer.wait_stream(es)
        for v in (out.values() if isinstance(out, dict) else (out,)):
            if isinstance(v, tt.Tensor) and v.is_cuda:
                v.record_stream(outer)
        return out

    def step(self, x_start, cond, t, noise, monitor: bool = False, wav=None, latent_scale: float = 18.0, update: bool = True):
        if self.use_graph and update and not monitor:
            return self._on_engine_stream(lambda: self._step_graphed(x_start, cond, t, noise))
        return self._on_engine_stream(lambda: self._step(x_start, cond, t, noise, monitor, wav, latent_scale, update))

    def _step_graphed(self, x_start, cond, t, noise):
        """The optimisation step replayed from ONE hipGraph (round 5; set `use_graph`).  The step's shapes are static and
        its ~1 600 launches are issued layer by layer from Python; captured once on the engine's stream (a single-stream graph, which
        ROCm 7.2 replays from recorded AQL packets) they cost the host one call.  The first step of a shape runs eagerly (lazy
        workspaces, function attributes), the second is captured over static input buffers and replayed, every later one copies its
        inputs into those buffers and replays.  The loss tensor returned is the graph's static output (valid until the next step).
        A training step with gradient reduction over ranks stays eager (the collectives are not captured)."""
        tt = self.torch
        eng = self.eng
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return self._step(x_start, cond, t, noise)
        x_start = x_start.to(eng.device, tt.float32).contiguous()
        cond = cond.to(eng.device, tt.float32).contiguous()
        t = t.to(eng.device, tt.int64).contiguous()
        noise = noise.to(eng.device, tt.float32).contiguous()
        key = (tuple(x_start.shape), tuple(cond.shape))
        if self._graph_key != key:
            self._graph, self._graph_key, self._graph_seen = None, key, 0
        if self._graph is None:
            self._graph_seen += 1
            if self._graph_seen < 2:
                return self._step(x_start, cond, t, noise)
            self._graph_in = tuple(tt.empty_like(v) for v in (x_start, cond, t, noise))
            for dst, src in zip(self._graph_in, (x_start, cond, t, noise)):
                dst.copy_(src)
            self.opt.use_device_count()
            g = tt.cuda.CUDAGraph()
            with tt.cuda.graph(g, stream=eng.stream):
                self._graph_out = self._step(*self._graph_in)
            self.opt.steps -= 1           # (the capture itself executed nothing)
            self._graph = g
        else:
            for dst, src in zip(self._graph_in, (x_start, cond, t, noise)):
                dst.copy_(src)
        self._graph.replay()
        self.opt.steps += 1
        return self._graph_out

    def _step(self, x_start, cond, t, noise, monitor: bool = False, wav=None, latent_scale: float = 18.0, update: bool = True):
        """-> loss (float tensor [1]) of this step, evaluated before the update.  monitor=True returns what DiffAudioRep.forward
        reports besides (model.py:181-209): {'diff_loss', 'neg_loss', 'predicted_x_start', 'x_hat', 'x_t'} -- predicted_x_start
        from the step's own forward pass (the reference runs the UNet a second time under no_grad for the same numbers,
        ddpm_loss.py:416-420), x_hat = decoder(predicted_x_start * scale), neg_loss = mean clamp(-SD-SDR(wav, x_hat), -30):
        the value srcs/train.py:401-408 selects the best checkpoint by.  update=False is the validation pass of the reference
        (model.eval() under torch.no_grad(), train.py:396-399): forward and losses only, no backward, no optimiser step."""
        from . import lib as LL, parallel
        eng = self.eng
        # ONE upload of the timesteps (and of a host-drawn noise): a pageable host-to-device copy blocks the host until the stream has
        # drained, and the three layers that take `t` would each do their own -- the last of them behind the whole forward pass
        t = t.to(eng.device, self.torch.int64).contiguous()
        noise = noise.to(eng.device, self.torch.float32).contiguous()
        if not update:
            x_t = q_sample(eng, x_start, t, noise)
            out = self.net.forward(x_t, t, cond)
            loss = p_losses_objective(eng, out, noise, t, want_grad=False)
            return self._report(loss, x_t, out, t, wav, latent_scale) if monitor else loss
        eng._grad_views, eng._grad_written = self._grad_views, set()
        # What feeds nothing but parameter gradients -- the weight-gradient GEMMs with their reductions, norm gains, the whole time-embedding
        # branch -- runs on the library's side stream, under the dX chain (csrc/train.hip: dw_side_fork; _SideRegion); so does the
        # time-embedding branch of the forward pass.  Joined before anything reads a parameter gradient.  The layers keep what those
        # launches read until their next forward; temporaries are marked for the allocator (_keep_for_side).
        side = bool(self.dw_side) and not self.torch.cuda.is_current_stream_capturing()
        if side:
            if self._side_ext is None:
                import ctypes
                h = ctypes.c_void_p()
                LL.check(eng.lib.ldc_train_side_stream(eng._ctx, ctypes.byref(h)))
                self._side_ext = self.torch.cuda.ExternalStream(h.value, device=eng.device)
            eng.set_option("train_dw_side", 1)
            eng._dw_side_ext = self._side_ext
        try:
            x_t = q_sample(eng, x_start, t, noise)
            out = self.net.forward(x_t, t, cond)
            loss, grad = p_losses_objective(eng, out, noise, t)
            grads, _, _ = self.net.backward(grad)
            missing = set(self._grad_views) - eng._grad_written
            if missing or set(grads) != set(self.names):
                raise RuntimeError(f"backward left {len(missing)} parameter gradient(s) unwritten; key mismatch: "
                                   f"{sorted(set(self.names) ^ set(grads))[:5]}")
        finally:
            eng._grad_views = None
            if side:
                eng.set_option("train_dw_side", 0)
                eng._dw_side_ext = None
                s = eng._enter()
                LL.check(eng.lib.ldc_train_join(eng._ctx, s))
                eng._exit()
        parallel.allreduce_gradients(self.flat_g)      # no-op without a process group
        self.opt.step(self.flat_g)
        return self._report(loss, x_t, out, t, wav, latent_scale) if monitor else loss

    def _report(self, loss, x_t, out, t, wav, latent_scale):
        from . import lib as LL
        eng = self.eng
        x0 = predict_x_start(eng, x_t, out, t)
        rep = {"diff_loss": loss, "predicted_x_start": x0, "x_t": x_t}
        if wav is not None:
            x_hat = eng.decode_latents(LL.MODEL_MAIN, x0 * float(latent_scale))
            rep["x_hat"] = x_hat
            rep["neg_per_item"] = neg_sdsdr(eng, wav, x_hat)
            rep["neg_loss"] = rep["neg_per_item"].mean()
        return rep

    def step_from_wav(self, wav, t=None, noise=None, latent_scale: float = 18.0, generator=None, monitor: bool = False, next_wav=None,
                      update: bool = True):
        return self._on_engine_stream(lambda: self._step_from_wav(wav, t, noise, latent_scale, generator, monitor, next_wav, update))

    def _step_from_wav(self, wav, t=None, noise=None, latent_scale: float = 18.0, generator=None, monitor: bool = False, next_wav=None,
                       update: bool = True):
        """The step as srcs/train.py:110-160 + DiffAudioRep.forward (model.py:146-182) drive it from audio: cond =
        model_for_cond.get_cond(x); x_rep = encoder(x) (frozen) / 18 (--scaling_global); t ~ U{0..T-1}, noise ~ N(0, I)
        (ddpm_loss.py:443-449) unless given; then `step`.  The engine's inference kernels run the two frozen encoders."""
        from . import lib as LL
        tt = self.torch
        wav = wav.to(self.eng.device, tt.float32).contiguous()
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf[0].data_ptr() == wav.data_ptr() and pf[0].shape == wav.shape:
            tt.cuda.current_stream(self.eng.device).wait_event(pf[3])
            cond, x_rep = pf[1], pf[2] / float(latent_scale)
        else:
            cond = self.eng.get_cond(wav)
            x_rep = self.eng.encode(LL.MODEL_MAIN, wav) / float(latent_scale)
        B = x_rep.shape[0]
        if t is None:      # on the device like the reference (ddpm_loss.py:447) unless a (CPU) generator asks for a reproducible host draw
            t = (tt.randint(0, self.num_timesteps, (B,), generator=generator) if generator is not None
                 else tt.randint(0, self.num_timesteps, (B,), device=self.eng.device))
        if noise is None:      # on the device unless a (CPU) generator asks for a reproducible host draw
            noise = tt.randn(x_rep.shape, generator=generator) if generator is not None else tt.randn(x_rep.shape, device=self.eng.device)
        # enqueued BEFORE this step's ~1 600 launches: the host needs most of a step's GPU time to issue them, so anything queued
        # behind them would start when the GPU is nearly through
        if next_wav is not None and self.frontend is not None:
            self.prefetch(next_wav)
        return self.step(x_rep, cond, t, noise, monitor=monitor, wav=wav, latent_scale=latent_scale, update=update)

    def prefetch(self, wav):
        """frozen encoders of a coming batch on the side stream / second engine, consumed by the step_from_wav call that gets the SAME tensor
        (matched by storage address and shape: the caller must not rewrite that buffer in place in between -- BatchWalker hands out a
        fresh tensor per batch; a prefetched batch that is never stepped on is simply dropped by the next call)"""
        from . import lib as LL
        tt = self.torch
        dev = self.eng.device
        wav = wav.to(dev, tt.float32).contiguous()
        main = tt.cuda.current_stream(dev)
        self._side.wait_stream(main)                   # the audio is ready (it may have been produced on the main stream)
        with tt.cuda.stream(self._side):
            cond = self.frontend.get_cond(wav)
            x_rep = self.frontend.encode(LL.MODEL_MAIN, wav)
            ev = tt.cuda.Event()
            ev.record(self._side)
        for x in (cond, x_rep):
            x.record_stream(main)
        self._prefetched = (wav, cond, x_rep, ev)


class _KeyRecorder(dict):
    """state dict that remembers which keys a Unet1D constructor read (all its reads go through __getitem__, see _Sub)"""

    def __init__(self, sd):
        super().__init__(sd)
        self.used = set()

    def __getitem__(self, k):
        self.used.add(k)
        return super().__getitem__(k)
