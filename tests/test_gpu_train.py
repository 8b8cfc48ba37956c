"""GPU training-step slice through the C ABI: q_sample, the p_losses objective and Block forward / backward against the
reference's own autograd results (tests/golden/train_block.npz) and, at a wider shape, against autograd of the oracle.
Tolerance 1e-4 relative.  The GEMM shapes run on the split-bf16 MFMA path by default (csrc/train_mm3.hip: three bf16 MFMAs per
product, 2^-16-class products, 4e-6 of a tensor's maximum measured) and, with the option train_fp32_mfma, on the exact-fp32 MFMA;
every test here runs on both (`gemm` fixture).  Two assertions are conditioned on it, both for the same reason -- they look at a
DISCONTINUOUS function of the network output: the L1 objective's gradient is sign(pred - noise) (one flipped element of 3e5 moves
final_conv.bias by 1.8e-3 of its maximum), and Adam's first step is lr * sign(g)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ladiffcodec_amd import train as TR  # noqa: E402
from oracle import train_oracle as TO  # noqa: E402
from helpers import T, load_golden  # noqa: E402
from gpu_common import engine, rel  # noqa: E402

TOL = 1e-4


@pytest.fixture(params=["split_bf16", "fp32_mfma"], autouse=True)
def gemm(request):
    """both GEMM paths under every test of this file (the option is process-wide: restored to the default afterwards)"""
    e = engine("r84", "f32")
    e.set_option("train_fp32_mfma", int(request.param == "fp32_mfma"))
    yield request.param
    e.set_option("train_fp32_mfma", 0)


def test_block_forward_backward_reference_vectors():
    g = load_golden("train_block")
    e = engine("r84", "f32")
    for tag, with_ss in (("a", True), ("b", False)):
        blk = TR.Block(e, T(g[f"{tag}.w"]), T(g[f"{tag}.b"]), T(g[f"{tag}.gamma"]), T(g[f"{tag}.beta"]))
        ss = (T(g[f"{tag}.scale"]), T(g[f"{tag}.shift"])) if with_ss else None
        y = blk.forward(T(g[f"{tag}.x"]), ss)
        assert rel(y.cpu().numpy(), g[f"{tag}.y"]) < TOL
        grads = blk.backward(T(g[f"{tag}.dy"]))
        for name, val in grads.items():
            assert rel(val.cpu().numpy(), g[f"{tag}.{name}"]) < TOL, (tag, name)


def test_block_gradients_wide_against_oracle_autograd():
    """diff_dims = 256 widths of the first level (256 -> 256 channels, L = 300), B = 3."""
    e = engine("r84", "f32")
    gen = torch.Generator().manual_seed(12)
    B, Cin, Cout, Lx = 3, 256, 256, 300
    leaf = lambda *s, k=1.0: (torch.randn(*s, generator=gen) * k).requires_grad_()
    x, w, b = leaf(B, Cin, Lx), leaf(Cout, Cin, 3, k=0.05), leaf(Cout, k=0.1)
    gamma, beta = (torch.rand(Cout, generator=gen) + 0.5).requires_grad_(), leaf(Cout, k=0.1)
    scale, shift = leaf(B, Cout, 1, k=0.3), leaf(B, Cout, 1, k=0.3)
    y = TO.block_forward(x, w, b, gamma, beta, scale, shift)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    blk = TR.Block(e, w.detach(), b.detach(), gamma.detach(), beta.detach())
    got_y = blk.forward(x.detach(), (scale.detach(), shift.detach()))
    assert rel(got_y.cpu().numpy(), y.detach().numpy()) < TOL
    grads = blk.backward(dy)
    want = {"dx": x.grad, "dw": w.grad, "db": b.grad, "dgamma": gamma.grad, "dbeta": beta.grad, "dscale": scale.grad, "dshift": shift.grad}
    for name, val in grads.items():
        assert rel(val.cpu().numpy(), want[name].numpy()) < TOL, name


def test_q_sample_and_objective_reference_vectors():
    g = load_golden("train_block")
    e = engine("r84", "f32")
    t = torch.from_numpy(g["q.t"])
    x_t = TR.q_sample(e, T(g["q.x0"]), t, T(g["q.noise"]))
    assert rel(x_t.cpu().numpy(), g["q.x_t"]) < 1e-6
    loss, grad = TR.p_losses_objective(e, T(g["q.model_out"]), T(g["q.noise"]), t)
    assert abs(float(loss.cpu()[0]) - float(g["q.loss"][0])) < 1e-5
    assert rel(grad.cpu().numpy(), g["q.grad"]) < 1e-6


def test_layernorm_forward_backward_reference_vectors_and_wide():
    g = load_golden("train_block")
    e = engine("r84", "f32")
    ln = TR.LayerNorm(e, T(g["ln.g"]))
    y = ln.forward(T(g["ln.x"]))
    assert rel(y.cpu().numpy(), g["ln.y"]) < 1e-5
    dx, dg = ln.backward(T(g["ln.dy"]))
    assert rel(dx.cpu().numpy(), g["ln.dx"]) < TOL and rel(dg.cpu().numpy(), g["ln.dg"].reshape(-1)) < TOL
    # the widths of the deepest attention block: C = 1024, L = 75, against autograd of the oracle
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 1024, 75, generator=gen) * 2.0 - 0.3).requires_grad_()
    gg = (torch.rand(1024, generator=gen) + 0.5).requires_grad_()
    yo = TO.layer_norm(x, gg)
    dy = torch.randn(yo.shape, generator=gen)
    yo.backward(dy)
    ln = TR.LayerNorm(e, gg.detach())
    assert rel(ln.forward(x.detach()).cpu().numpy(), yo.detach().numpy()) < 1e-5
    dx, dg = ln.backward(dy)
    assert rel(dx.cpu().numpy(), x.grad.numpy()) < TOL and rel(dg.cpu().numpy(), gg.grad.numpy()) < TOL


def test_adam_steps_match_torch_optim_adam():
    """Three steps against the reference's optimiser (optim.Adam(params, lr), train.py:365-371): golden from torch.optim.Adam."""
    g = load_golden("train_block")
    e = engine("r84", "f32")
    p = T(g["adam.p0"]).cuda().contiguous()
    opt = TR.Adam(e, p, lr=3e-4)
    for k in range(3):
        opt.step(T(g[f"adam.g{k}"]))
        assert float((p.cpu() - T(g[f"adam.p{k + 1}"])).abs().max()) < 3e-7, k      # |p| ~ 0.2, steps of 3e-4: a few ulp
    assert not bool(torch.isnan(p).any())


def test_resnet_block_forward_backward_reference_vectors():
    """A whole ResnetBlock with time embedding (unet.py:157-192), with and without res_conv, against the reference's autograd."""
    g = load_golden("train_block")
    e = engine("r84", "f32")
    for tag in ("rb1", "rb2"):
        p = {k[len(tag) + 3:]: T(g[k]) for k in list(g.keys()) if k.startswith(tag + ".p.")}
        rb = TR.ResnetBlock(e, p)
        y = rb.forward(T(g[f"{tag}.x"]), T(g[f"{tag}.temb"]))
        assert rel(y.cpu().numpy(), g[f"{tag}.y"]) < TOL
        grads = rb.backward(T(g[f"{tag}.dy"]))
        assert rel(grads["dx"].cpu().numpy(), g[f"{tag}.dx"]) < TOL and rel(grads["dtime_emb"].cpu().numpy(), g[f"{tag}.dtemb"]) < TOL
        for name in p:
            assert rel(grads[name].cpu().numpy().reshape(g[f"{tag}.g.{name}"].shape), g[f"{tag}.g.{name}"]) < TOL, (tag, name)


def test_linear_attention_block_forward_backward_reference_vectors():
    """Residual(PreNorm(LinearAttention)) (unet.py:103-116,194-222) against the reference's autograd."""
    g = load_golden("train_block")
    e = engine("r84", "f32")
    p = {k[5:]: T(g[k]) for k in list(g.keys()) if k.startswith("la.p.")}
    att = TR.LinearAttention(e, p)
    y = att.forward(T(g["la.x"]))
    assert rel(y.cpu().numpy(), g["la.y"]) < TOL
    grads = att.backward(T(g["la.dy"]))
    assert rel(grads["dx"].cpu().numpy(), g["la.dx"]) < TOL
    for name in p:
        assert rel(grads[name].cpu().numpy().reshape(g[f"la.g.{name}"].shape), g[f"la.g.{name}"]) < TOL, name


def test_conv_upsample_activation_attention_against_autograd():
    """The remaining Unet1D layers, each forward / backward against torch autograd of the same functional op on the CPU."""
    import torch.nn.functional as F
    e = engine("r84", "f32")
    gen = torch.Generator().manual_seed(9)
    for cin, cout, k, st, pd, Lx in ((16, 32, 7, 1, 3, 50), (32, 64, 4, 2, 1, 48), (24, 24, 3, 1, 1, 37), (32, 8, 1, 1, 0, 20)):
        x = torch.randn(2, cin, Lx, generator=gen, requires_grad=True)
        w = (torch.randn(cout, cin, k, generator=gen) * 0.2).requires_grad_()
        b = (torch.randn(cout, generator=gen) * 0.1).requires_grad_()
        y = F.conv1d(x, w, b, stride=st, padding=pd)
        dy = torch.randn(y.shape, generator=gen)
        y.backward(dy)
        cv = TR.Conv1d(e, w.detach(), b.detach(), st, pd)
        assert rel(cv.forward(x.detach()).cpu().numpy(), y.detach().numpy()) < 1e-5
        gr = cv.backward(dy)
        assert rel(gr["dx"].cpu().numpy(), x.grad.numpy()) < TOL and rel(gr["dw"].cpu().numpy(), w.grad.numpy()) < TOL
        assert rel(gr["db"].cpu().numpy(), b.grad.numpy()) < TOL
    x = torch.randn(2, 5, 11, generator=gen)
    up = TR.upsample2(e, x)
    assert torch.equal(up.cpu(), F.interpolate(x, scale_factor=2, mode="nearest"))
    d = torch.randn(2, 5, 22, generator=gen)
    assert rel(TR.upsample2(e, d, backward=True).cpu().numpy(), (d[..., 0::2] + d[..., 1::2]).numpy()) < 1e-6
    for kind, fn in ((TR.ACT_TANH, torch.tanh), (TR.ACT_GELU, F.gelu), (TR.ACT_SILU, F.silu)):
        x = (torch.randn(3, 40, generator=gen) * 2).requires_grad_()
        y = fn(x)
        dy = torch.randn(y.shape, generator=gen)
        y.backward(dy)
        assert rel(TR.activation(e, x.detach(), kind).cpu().numpy(), y.detach().numpy()) < 1e-5
        assert rel(TR.activation(e, x.detach(), kind, dy=dy).cpu().numpy(), x.grad.numpy()) < 1e-5


def test_assembled_unet_forward_backward_reference_vectors():
    """Unet1D.forward / backward over the reference's own state dict (two levels, dim 16): output, input gradients and the
    gradient of every one of the 160 parameters against the reference's autograd (tests/golden/train_unet.npz); then one Adam
    step of the flattened parameters against torch.optim.Adam."""
    g = load_golden("train_unet")
    e = engine("r84", "f32")
    sd = {k[2:]: T(g[k]) for k in list(g.keys()) if k.startswith("p.")}
    net = TR.Unet1D(e, sd, dim=16, dim_mults=(1, 2))
    y = net.forward(T(g["x"]), torch.from_numpy(g["time"]), T(g["xc"]))
    assert rel(y.cpu().numpy(), g["y"]) < TOL
    grads, dx, dxc = net.backward(T(g["dy"]))
    assert rel(dx.cpu().numpy(), g["dx"]) < TOL and rel(dxc.cpu().numpy(), g["dxc"]) < TOL
    assert set(grads) == set(sd), (set(sd) - set(grads), set(grads) - set(sd))
    worst = max((rel(grads[k].cpu().numpy().reshape(g["g." + k].shape), g["g." + k]), k) for k in sd)
    assert worst[0] < 2e-4, worst
    # one optimiser step over the flat buffers (the layout parallel.allreduce_gradients reduces).  Adam's first step is
    # lr * g / (|g| + 1e-8): it maps a gradient element to +-lr whatever its size, so it amplifies any rounding of the elements that are
    # ~0 relative to their tensor.  The kernel is therefore pinned on the REFERENCE's gradients (tight), and the step from this path's
    # own gradients (split-bf16 GEMMs: 2^-16-class products, csrc/train_mm3.hip) is required to agree wherever the gradient is not
    # negligible against its tensor's scale.
    names = sorted(sd)
    flat_p0 = torch.cat([sd[k].reshape(-1) for k in names]).cuda().contiguous()
    ref_g = torch.cat([T(g["g." + k]).reshape(-1) for k in names])
    ref_p = torch.nn.Parameter(flat_p0.cpu().clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    ref_p.grad = ref_g.clone()
    opt.step()
    flat_p = flat_p0.clone()
    TR.Adam(e, flat_p, lr=1e-3).step(ref_g.cuda().contiguous())
    dev = (flat_p.cpu() - ref_p.detach()).abs()
    big = ref_g.abs() > 1e-5
    assert float(dev[big].max()) < 2e-6 and float(big.float().mean()) > 0.9
    assert float(dev.max()) <= 1.05e-3
    flat_g = torch.cat([grads[k].reshape(-1) for k in names]).contiguous()
    flat_p = flat_p0.clone()
    TR.Adam(e, flat_p, lr=1e-3).step(flat_g)
    dev = (flat_p.cpu() - ref_p.detach()).abs()
    scale = torch.cat([T(g["g." + k]).abs().max().expand(g["g." + k].size) for k in names])
    clear = ref_g.abs() > 1e-3 * scale
    assert float(dev[clear].max()) < 2e-6 and float(clear.float().mean()) > 0.8, (float(dev[clear].max()), float(clear.float().mean()))
    assert float(dev.max()) <= 2.1e-3


def test_training_steps_follow_the_reference_loop():
    """Four optimisation steps of the diffusion UNet on the GPU (q_sample -> UNet forward -> objective -> UNet backward -> Adam)
    against the same loop run by torch autograd + torch.optim.Adam on the oracle (CPU): the loss of every step and the final
    parameters must agree, and the loss on the fixed batch must go down."""
    from ladiffcodec_amd import synth
    from ladiffcodec_amd.spec import UnetConfig
    g = load_golden("train_unet")
    e = engine("r84", "f32")
    sd = {k[2:]: T(g[k]) for k in list(g.keys()) if k.startswith("p.")}
    u = UnetConfig(dim=16, dim_mults=(1, 2), inp_channels=8, cond_channels=8, upsampling_ratios=None, unet_scale_cond=False)
    sched = {"diffusion." + k: torch.from_numpy(v) for k, v in synth.cosine_schedule_buffers(1000).items()}
    gen = torch.Generator().manual_seed(31)
    x0 = torch.randn(2, 8, 32, generator=gen).clamp(-1, 1)
    cond = torch.randn(2, 8, 32, generator=gen)
    ts = [torch.tensor([40, 700]), torch.tensor([5, 333]), torch.tensor([40, 700]), torch.tensor([40, 700])]
    noises = [torch.randn(2, 8, 32, generator=gen) for _ in ts]
    noises[2] = noises[0]; noises[3] = noises[0]                       # steps 0, 2, 3 see the same sample: their loss must fall
    want_losses, want_sd = TO.training_steps({"diff_model." + k: v for k, v in sd.items()}, u, x0, cond, ts, noises, 2e-3, sched)
    tr = TR.DiffusionTrainer(e, sd, dim=16, dim_mults=(1, 2), lr=2e-3)
    got = [float(tr.step(x0, cond, t, n).cpu()[0]) for t, n in zip(ts, noises)]
    for a, b in zip(got, want_losses):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (got, want_losses)
    assert got[3] < got[2] < got[0]
    final = tr.state_dict()
    dev = torch.cat([(final[k].cpu() - want_sd["diff_model." + k]).abs().reshape(-1) for k in sd])
    # parameters move by ~lr = 2e-3 per step; elements whose gradient sits at Adam's eps amplify summation-order noise (see above),
    # so the bar is on the bulk: mean deviation and the share of elements off by more than a tenth of a step
    assert float(dev.mean()) < 2e-5 and float((dev > 2e-4).float().mean()) < 2e-3, (float(dev.mean()), float(dev.max()))


def test_assembled_unet_with_process_cond_reference_vectors():
    """Unet1D with upsampling_ratios [5, 2] and unet_scale_cond: the condition upsamplers' parameters and the raw condition get
    their gradients through the max-abs scaling and both transposed convs (every parameter of diff_model is covered)."""
    g = load_golden("train_unet")
    e = engine("r84", "f32")
    sd = {k[4:]: T(g[k]) for k in list(g.keys()) if k.startswith("u.p.")}
    net = TR.Unet1D(e, sd, dim=16, dim_mults=(1, 2), upsampling_ratios=(5, 2), unet_scale_cond=True)
    y = net.forward(T(g["u.x"]), torch.from_numpy(g["u.time"]), T(g["u.cond"]))
    assert rel(y.cpu().numpy(), g["u.y"]) < TOL
    grads, dx, dcond = net.backward(T(g["u.dy"]))
    assert rel(dx.cpu().numpy(), g["u.dx"]) < TOL and rel(dcond.cpu().numpy(), g["u.dcond"]) < 2e-4
    assert set(grads) == set(sd), (set(sd) - set(grads), set(grads) - set(sd))
    worst = max((rel(grads[k].cpu().numpy().reshape(g["u.g." + k].shape), g["u.g." + k]), k) for k in sd)
    assert worst[0] < 2e-4, worst


def test_training_step_from_audio_matches_oracle(gemm):
    """train.py's step driven from audio on the small fixture model (five UNet levels, two upsamplers, scaled condition): frozen
    SEANet encoder + condition codec on the inference kernels, then q_sample -> UNet -> objective -> backward -> Adam; the loss
    and the updated parameters against the oracle's encoders + autograd + torch.optim.Adam."""
    from ladiffcodec_amd import synth
    from oracle import ldc_oracle as O
    from helpers import CASES, COND_CFG, cond_sd_np, main_sd_np
    mc, u, _ = CASES["r84"]
    e = engine("r84", "f32")
    full = main_sd_np("r84")
    sd_np = {k[len("diff_model."):]: v for k, v in full.items() if k.startswith("diff_model.")}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    tr = TR.DiffusionTrainer(e, sd, dim=u.dim, dim_mults=u.dim_mults, lr=1e-3, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=u.unet_scale_cond)
    T_ = 2560
    wav = torch.from_numpy(synth.synthetic_wav(2, T_, seed=77))
    gen = torch.Generator().manual_seed(8)
    t = torch.tensor([123, 871])
    noise = torch.randn(2, 128, T_ // mc.hop_length, generator=gen)
    loss = float(tr.step_from_wav(wav, t=t, noise=noise).cpu()[0])
    # oracle
    sdm, sdc = synth.to_torch(full), synth.to_torch(cond_sd_np())
    with torch.no_grad():
        cond = O.get_cond(sdc, COND_CFG, wav)[0]
        x0 = O.seanet_encode(sdm, mc, wav) / 18.0
    params = {k: v.clone().requires_grad_() for k, v in sdm.items() if k.startswith("diff_model.")}
    sched = {k: v for k, v in sdm.items() if k.startswith("diffusion.")}
    x_t = TO.q_sample(sched, x0, t, noise)
    want = TO.p_losses_objective(sched, O.unet_forward(params, u, x_t, t, cond), noise, t)
    assert abs(loss - float(want.detach())) < 2e-4 * max(1.0, abs(float(want.detach())))
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    want.backward()
    opt.step()
    final = tr.state_dict()
    dev = torch.cat([(final[k[len("diff_model."):]].cpu() - v.detach()).abs().reshape(-1) for k, v in params.items()])
    flips = 2e-3 if gemm == "fp32_mfma" else 2e-2    # share of parameters whose first Adam step differs: sign(g) of gradients ~0
    assert float(dev.mean()) < 1e-5 and float((dev > 1e-4).float().mean()) < flips, (float(dev.mean()), float(dev.max()), float((dev > 1e-4).float().mean()))   # first step: every parameter moves by ~lr = 1e-3


def test_split_bf16_step_is_bit_reproducible(gemm):
    """No atomics on the split-bf16 path (split reductions go through ordered partial sums): the same forward / backward twice gives
    the same bits, at a size where dW, db, forward and dX all split (B = 8, 256 -> 256 channels, L = 1200: 12 dW tiles -> 8 parts)."""
    if gemm != "split_bf16":
        pytest.skip("the exact-fp32 path adds its dW parts with fp32 atomics")
    e = engine("r84", "f32")
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(8, 256, 1200, generator=gen).cuda()
    w = (torch.randn(256, 256, 3, generator=gen) * 0.05).cuda()
    b = (torch.randn(256, generator=gen) * 0.1).cuda()
    dy = torch.randn(8, 256, 1200, generator=gen).cuda()
    outs = []
    for _ in range(2):
        cv = TR.Conv1d(e, w, b, 1, 1)
        y = cv.forward(x)
        g = cv.backward(dy)
        outs.append((y.clone(), g["dx"].clone(), g["dw"].clone(), g["db"].clone()))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    # and a deep-level shape whose forward / dX split their reduction (19 x 8 tiles -> 4 parts)
    x = torch.randn(32, 1024, 75, generator=gen).cuda()
    w = (torch.randn(1024, 1024, 3, generator=gen) * 0.02).cuda()
    dy = torch.randn(32, 1024, 75, generator=gen).cuda()
    outs = []
    for _ in range(2):
        cv = TR.Conv1d(e, w, None, 1, 1)
        y = cv.forward(x)
        g = cv.backward(dy)
        outs.append((y.clone(), g["dx"].clone(), g["dw"].clone()))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    want = torch.nn.functional.conv1d(x.cpu().double(), w.cpu().double(), None, padding=1)
    assert rel(outs[0][0].cpu().numpy(), want.numpy()) < 2e-5


def test_plain_bf16_option_is_bf16_class_and_off_by_default(gemm):
    """option train_bf16: the GEMM shapes with one bf16 MFMA per product (no lo terms) -- an opt-in for runs that accept
    autocast-class numerics; errors must be bf16-class (2^-9 products: 1e-4 .. 2e-2 of the maximum), and fp32-class again once it is off."""
    if gemm != "split_bf16":
        pytest.skip("an option of the split-bf16 kernels")
    import torch.nn.functional as F
    e = engine("r84", "f32")
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(4, 256, 300, generator=gen, requires_grad=True)
    w = (torch.randn(128, 256, 3, generator=gen) * 0.05).requires_grad_()
    y = F.conv1d(x.double(), w.double(), None, padding=1)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    errs = {}
    for on in (1, 0):
        e.set_option("train_bf16", on)
        try:
            cv = TR.Conv1d(e, w.detach(), None, 1, 1)
            got = cv.forward(x.detach())
            g = cv.backward(dy)
            errs[on] = (rel(got.cpu().numpy(), y.detach().numpy()), rel(g["dx"].cpu().numpy(), x.grad.numpy()), rel(g["dw"].cpu().numpy(), w.grad.numpy()))
        finally:
            e.set_option("train_bf16", 0)
    assert all(1e-4 < v < 2e-2 for v in errs[1]), errs
    assert all(v < 2e-5 for v in errs[0]), errs


def test_frozen_encoders_prefetched_on_a_second_engine_give_the_same_steps():
    """DiffusionTrainer(frontend=...): the encoders of the next batch run on a side stream / second engine under the current step; three
    steps over two alternating batches must give the encodings (bitwise) and the losses of the in-line trainer."""
    from ladiffcodec_amd import lib as L, synth
    from ladiffcodec_amd.model import Engine
    from helpers import CASES, COND_CFG, cond_sd_np, main_sd_np
    mc, u, _ = CASES["r84"]
    e = engine("r84", "f32")
    front = Engine(mc, u, COND_CFG, dtype="f32")
    front.load_state_dict(L.MODEL_MAIN, main_sd_np("r84"))
    front.load_state_dict(L.MODEL_COND, cond_sd_np())
    front.finalize(strict=True)
    sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in main_sd_np("r84").items() if k.startswith("diff_model.")}
    kw = dict(dim=u.dim, dim_mults=u.dim_mults, lr=1e-3, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=u.unet_scale_cond)
    wavs = [torch.from_numpy(synth.synthetic_wav(2, 2560, seed=s)).cuda() for s in (5, 6)]
    gen = torch.Generator().manual_seed(3)
    ts = [torch.tensor([10, 900]), torch.tensor([500, 77]), torch.tensor([3, 650])]
    noises = [torch.randn(2, 128, 2560 // mc.hop_length, generator=gen) for _ in ts]
    order = [0, 1, 0]
    plain = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, **kw)
    want = [float(plain.step_from_wav(wavs[i], t=t, noise=n).cpu()[0]) for i, t, n in zip(order, ts, noises)]
    pre = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, frontend=front, **kw)
    got = []
    for k, (i, t, n) in enumerate(zip(order, ts, noises)):
        nxt = wavs[order[k + 1]] if k + 1 < len(order) else None
        got.append(float(pre.step_from_wav(wavs[i], t=t, noise=n, next_wav=nxt).cpu()[0]))
        assert (pre._prefetched is not None) == (nxt is not None)
        if nxt is not None:       # what the side stream produced is what the in-line engine produces, bit for bit
            torch.cuda.current_stream().wait_event(pre._prefetched[3])
            assert torch.equal(pre._prefetched[1], e.get_cond(nxt)) and torch.equal(pre._prefetched[2], e.encode(L.MODEL_MAIN, nxt))
    # the first loss is bit-identical; later ones see parameters after Adam's first steps (lr * sign(g)).  The split-bf16 GEMMs are
    # atomic-free and bit-reproducible (test_split_bf16_step_is_bit_reproducible); what still differs between two trainers here is
    # the exact-fp32 path's and the norms' fp32 atomics, which decide the sign of ~0 gradients: hence the looser bar
    assert got[0] == want[0] and max(abs(a - b) for a, b in zip(got, want)) < 2e-4, (got, want)


def test_training_loop_from_the_dataset_walker(tmp_path):
    """The pieces of the reference's --run_diff loop together (srcs/train.py:110-177, 322-336): Dataset_Libri-shaped walker -> float32
    [B, 1, T] batches uploaded one ahead on a side stream -> DiffusionTrainer.step_from_wav with the next batch's frozen encoders
    prefetched on a second engine; the losses must be those of the plain loop over the same crops."""
    from ladiffcodec_amd import lib as L
    from ladiffcodec_amd.dataset import BatchWalker, DatasetLibri
    from ladiffcodec_amd.model import Engine
    from helpers import CASES, COND_CFG, cond_sd_np, libri_tree, main_sd_np
    mc, u, _ = CASES["r84"]
    libri_tree(str(tmp_path))
    e = engine("r84", "f32")
    front = Engine(mc, u, COND_CFG, dtype="f32")
    front.load_state_dict(L.MODEL_MAIN, main_sd_np("r84"))
    front.load_state_dict(L.MODEL_COND, cond_sd_np())
    front.finalize(strict=True)
    sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in main_sd_np("r84").items() if k.startswith("diff_model.")}
    kw = dict(dim=u.dim, dim_mults=u.dim_mults, lr=1e-3, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=u.unet_scale_cond)
    ds = DatasetLibri(task="train", seq_len_p_sec=0.16, data_folder_path=str(tmp_path))      # 2560-sample crops
    ds.files = sorted(ds.files)
    gen = torch.Generator().manual_seed(4)
    ts = [torch.randint(0, 1000, (2,), generator=gen) for _ in range(3)]
    noises = [torch.randn(2, 128, 2560 // mc.hop_length, generator=gen) for _ in range(3)]
    torch.manual_seed(7)
    crops = [torch.from_numpy(np.stack([np.asarray(ds[i]) for i in idx])).unsqueeze(1).float() for idx in ([0, 1], [2, 3], [4])]
    plain = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, **kw)
    want = [float(plain.step_from_wav(c, t=t[:c.shape[0]], noise=n[:c.shape[0]]).cpu()[0]) for c, t, n in zip(crops, ts, noises)]
    torch.manual_seed(7)
    tr = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, frontend=front, **kw)
    walker = BatchWalker(ds, batch_size=2, device="cuda")
    got = []
    for k, wav in enumerate(walker):
        assert wav.is_cuda and wav.dtype == torch.float32 and torch.equal(wav.cpu(), crops[k])
        got.append(float(tr.step_from_wav(wav, t=ts[k][:wav.shape[0]], noise=noises[k][:wav.shape[0]], next_wav=walker.peek()).cpu()[0]))
    assert len(got) == 3 and got[0] == want[0] and max(abs(a - b) for a, b in zip(got, want)) < 2e-4, (got, want)


def test_run_diff_training_loop_end_to_end(tmp_path):
    """`python -m srcs.train --run_diff --freeze_ed ...` (ladiffcodec_amd/train_loop.py, srcs/train.py:227-417): three outer steps over a
    synthetic LibriSpeech-shaped tree with the dim-32 checkpoints -- every step trains an epoch, validates without updating, keeps
    model_best.amlt by the monitored neg_loss; the saved file must carry the trainer's parameters under both prefixes and load back
    into an engine whose UNet output differs from the initial one."""
    from ladiffcodec_amd import checkpoint, synth, train_loop
    from helpers import CASES, cond_sd_np, libri_tree, main_sd_np
    mc, u, _ = CASES["r84"]
    libri_tree(str(tmp_path / "libri"))
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    (tmp_path / "cond").mkdir()
    synth.save_amlt(cond_sd_np(), str(tmp_path / "cond" / "model_best.amlt"))
    lines = []
    a = train_loop.build_parser().parse_args([
        "--run_diff", "--freeze_ed", "--scaling_global", "--unet_scale_cond", "--cond_quantization", "--model_for_cond", str(tmp_path / "cond"),
        "--finetune_model", str(tmp_path / "ladiff"), "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--cond_bandwidth", "3", "--data_folder_path", str(tmp_path / "libri"), "--seq_len_p_sec", "0.16", "--batch_size", "2", "--lr", "1e-3",
        "--output_dir", str(tmp_path / "out"), "--exp_name", "t", "--num_steps", "6"])
    res = train_loop.run(a, log=lines.append)
    assert [h[0] for h in res["history"]] == [0, 5] and len(lines) == 2            # write_on_every = 5 (train.py:379)
    assert all(np.isfinite(v) for h in res["history"] for d in h[1:] for v in d.values())
    assert res["saved"] and res["saved"][0].endswith("/out/t/model_best.amlt")
    back = checkpoint.read_amlt(res["saved"][-1])
    base = main_sd_np("r84")
    assert list(back) == list(base)
    trained = res["trainer"].state_dict()
    moved = 0
    for k, v in base.items():
        for prefix in ("diff_model.", "diffusion.model."):
            if k.startswith(prefix) and k[len(prefix):] in trained:
                moved += int(not np.array_equal(back[k], v))
                break
        else:
            assert np.array_equal(back[k], np.asarray(v, np.float32)), k               # frozen codec / schedule carried over
    assert moved > 300


def test_full_width_training_step_reference_vectors(gemm):
    """ONE optimisation step at the size BASELINE configs[3] names (diff_dims 256, seq_length 1200, enc_ratios 8 4; the grids
    `bench.py --config c4` times) driven from audio, against the reference under torch autograd (tests/golden/train256.npz,
    tools/gen_golden_train256.py: DiffAudioRep.forward of srcs/model.py:146-209): diff_loss, x_t, predicted_x_start, the decoder's
    x_hat and the SD-SDR monitoring loss, sampled gradients of 28 parameters (k = 7 init conv, k = 4 stride-2 downsample, k = 3
    convs incl. concatenated inputs, upsample conv, 1x1 / Linear layers, norms, both transposed-conv condition upsamplers: the
    split-over-items dW path and every convmm grid class) and the sum of |g| of EVERY one of the 340 parameters."""
    from ladiffcodec_amd import lib as L, synth
    from ladiffcodec_amd.model import Engine
    from ladiffcodec_amd.spec import CodecConfig, UnetConfig
    from helpers import sub
    g = load_golden("train256")
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    full = synth.ladiff_state_dict(mc, u, seed=1)
    e = Engine(mc, u, cc, dtype="f32")
    e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in full.items() if not k.startswith("diffusion.model.")})
    e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=11))
    e.finalize(strict=True)
    sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in full.items() if k.startswith("diff_model.")}
    tr = TR.DiffusionTrainer(e, sd, dim=256, dim_mults=u.dim_mults, lr=1e-4, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=True)
    assert tr.num_timesteps == 1000 and set(tr.names) == set(str(n) for n in g["names"])
    wav = torch.from_numpy(synth.synthetic_wav(2, 38400, seed=5))
    noise = torch.randn(2, 128, 1200, generator=torch.Generator().manual_seed(6))
    rep = tr.step_from_wav(wav, t=torch.from_numpy(g["t"]), noise=noise, monitor=True)
    assert abs(float(rep["diff_loss"].cpu()[0]) - float(g["loss"][0])) < 2e-5 * max(1.0, abs(float(g["loss"][0])))
    for key, tol in (("x_t", 1e-5), ("predicted_x_start", 1e-4), ("x_hat", 2e-4)):
        got = rep[key].cpu().numpy()
        assert rel(sub(got, g[key + ".stride"]), g[key]) < tol, key
    assert np.abs(rep["neg_per_item"].cpu().numpy() - g["neg_per_item"]).max() < 2e-3          # dB
    assert abs(float(rep["neg_loss"].cpu()) - float(g["neg_loss"][0])) < 2e-3
    grads = tr.gradients()
    errs = []
    for k in [k for k in g if k.startswith("g.") and not k.endswith(".stride")]:
        name = k[2:]
        got = grads[name].cpu().numpy().reshape(-1)[::int(g[k + ".stride"])]
        errs.append((rel(got, g[k]), name))
    errs.sort(reverse=True)
    # split-bf16: a handful of L1 sign flips in 307 200 outputs (module docstring); measured 1.8e-3 (final_conv), 1.0e-3 elsewhere
    assert errs[0][0] < (5e-4 if gemm == "fp32_mfma" else 4e-3), errs[:6]
    sums = {str(n): float(v) for n, v in zip(g["names"], g["abs_sums"])}
    off = max((abs(float(grads[n].double().abs().sum().cpu()) - sums[n]) / (sums[n] + 1e-12), n) for n in tr.names)
    assert off[0] < 2e-3, off
    e.close()


def test_graphed_optimisation_step_equals_the_eager_one():
    """Round 5: the optimisation step replayed from one hipGraph (DiffusionTrainer.use_graph: the first step of a shape
    eager, the second captured over static input buffers, the rest replayed; Adam's step count on the device) against the same
    steps issued layer by layer: different inputs every step (the static buffers are refreshed), same losses and parameters
    (srcs/train.py:110-177 is the loop both stand for)."""
    g = load_golden("train_unet")
    sd = {k[2:]: T(g[k]) for k in list(g.keys()) if k.startswith("p.")}
    gen = torch.Generator().manual_seed(77)
    steps = [(torch.randn(2, 8, 32, generator=gen).clamp(-1, 1), torch.randn(2, 8, 32, generator=gen), torch.randint(0, 1000, (2,), generator=gen),
              torch.randn(2, 8, 32, generator=gen)) for _ in range(6)]
    out = []
    for use_graph in (False, True):
        e = engine("r84", "f32")
        tr = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, dim=16, dim_mults=(1, 2), lr=2e-3)
        tr.use_graph = use_graph
        losses = [float(tr.step(*s).cpu()[0]) for s in steps]
        out.append((losses, torch.cat([v.reshape(-1).cpu() for v in tr.state_dict().values()])))
        assert tr.opt.steps == len(steps)
        if use_graph:
            assert tr._graph is not None
    (l0, p0), (l1, p1) = out
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 2e-5, (l0, l1)
    dev = (p0 - p1).abs()
    assert float(dev.mean()) < 2e-6 and float((dev > 2e-4).float().mean()) < 2e-3, (float(dev.mean()), float(dev.max()))


def test_weight_gradients_on_the_side_stream_equal_the_in_line_ones():
    """Round 6: DiffusionTrainer.dw_side (default on) puts the Blocks' weight-gradient GEMMs, their ordered reductions, bias
    gradients and weight-standardisation backward on the library's side stream under the dX chain (csrc/train.hip:
    dw_side_fork; ldc_train_join before the gradients are used; `loss.backward()` of srcs/train.py:150-158 is what both orders
    stand for).  Same kernels on the same operands: losses, the flat gradient of every step and the parameters after six Adam
    steps are bit-identical with the in-line order -- at a width where the GEMM path is taken (L >= 16) -- and switching the
    setting between steps continues the same trajectory (nothing is left on the side stream after a step)."""
    g = load_golden("train_unet")
    sd = {k[2:]: T(g[k]) for k in list(g.keys()) if k.startswith("p.")}
    gen = torch.Generator().manual_seed(78)
    steps = [(torch.randn(2, 8, 32, generator=gen).clamp(-1, 1), torch.randn(2, 8, 32, generator=gen), torch.randint(0, 1000, (2,), generator=gen),
              torch.randn(2, 8, 32, generator=gen)) for _ in range(6)]
    out = []
    for pattern in ((False,) * 6, (True,) * 6, (True, False, True, True, False, True)):
        e = engine("r84", "f32")
        tr = TR.DiffusionTrainer(e, {k: v.clone() for k, v in sd.items()}, dim=16, dim_mults=(1, 2), lr=2e-3)
        losses, grads = [], []
        for side, st in zip(pattern, steps):
            tr.dw_side = side
            losses.append(float(tr.step(*st).cpu()[0]))
            grads.append(tr.flat_g.detach().cpu().clone())
        out.append((losses, grads, torch.cat([v.reshape(-1).cpu() for v in tr.state_dict().values()])))
    for losses, grads, params in out[1:]:
        assert losses == out[0][0], (losses, out[0][0])
        for a, b in zip(grads, out[0][1]):
            assert float(a.abs().max()) > 0 and torch.equal(a, b)
        assert torch.equal(params, out[0][2])
