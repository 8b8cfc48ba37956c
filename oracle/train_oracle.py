"""CPU restatement (differentiable, torch autograd) of the training-step slice -- TEST INFRASTRUCTURE ONLY.

  block_forward      srcs/modules/unet.py:137-154 (Block) with the weight path of WeightStandardizedConv2d, :67-80
  q_sample           srcs/losses/ddpm_loss.py:386-392
  p_losses_objective ddpm_loss.py:434-438 (loss_type 'l1', objective 'pred_noise')
Pinned by tests/golden/train_block.npz: outputs AND gradients of the reference's own Block / p_losses under autograd
(tools/gen_golden.py, GOLDEN_ONLY=train).
"""
import torch
import torch.nn.functional as F

from oracle.ldc_oracle import ws_fold


def block_forward(x, w, b, gamma, beta, scale=None, shift=None, groups: int = 8):
    h = F.conv1d(x, ws_fold(w), b, padding=1)
    h = F.group_norm(h, groups, gamma, beta, eps=1e-5)
    if scale is not None:
        h = h * (scale + 1) + shift
    return F.silu(h)


def q_sample(sd, x_start, t, noise):
    a = sd["diffusion.sqrt_alphas_cumprod"][t].view(-1, 1, 1)
    c = sd["diffusion.sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1)
    return a * x_start + c * noise


def p_losses_objective(sd, model_out, target, t):
    loss = F.l1_loss(model_out, target, reduction="none").flatten(1).mean(dim=1)
    return (loss * sd["diffusion.p2_loss_weight"][t]).mean()


def layer_norm(x, g):
    """channel LayerNorm, unet.py:82-101 (float32 branch: eps 1e-5, biased variance)."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g.view(1, -1, 1)


def adam_step(p, g, m, v, step: int, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam's update (the reference's optimiser, train.py:365-371): returns (p, m, v) after one step."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / (1 - b2 ** step) ** 0.5 + eps
    return p - (lr / (1 - b1 ** step)) * (m / denom), m, v


def resnet_block_forward(p, x, time_emb, groups: int = 8):
    """ResnetBlock.forward, unet.py:157-192 (use_film False): p holds the block's parameters keyed as in its state dict."""
    ss = F.linear(F.silu(time_emb), p["mlp.1.weight"], p["mlp.1.bias"]).unsqueeze(-1)
    scale, shift = ss.chunk(2, dim=1)
    h = block_forward(x, p["block1.proj.weight"], p["block1.proj.bias"], p["block1.norm.weight"], p["block1.norm.bias"], scale, shift, groups)
    h = block_forward(h, p["block2.proj.weight"], p["block2.proj.bias"], p["block2.norm.weight"], p["block2.norm.bias"], None, None, groups)
    res = F.conv1d(x, p["res_conv.weight"], p["res_conv.bias"]) if "res_conv.weight" in p else x
    return h + res


def linear_attention_block(p, x, heads: int = 4, dim_head: int = 32):
    """Residual(PreNorm(dim, LinearAttention(dim))), unet.py:103-116 and 194-222; p keyed as train.LinearAttention expects."""
    b, c, n = x.shape
    xn = layer_norm(x, p["norm.g"].reshape(-1))
    q, k, v = F.conv1d(xn, p["to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in (q, k, v))
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * dim_head, n)
    out = F.conv1d(out, p["to_out.0.weight"], p["to_out.0.bias"])
    return layer_norm(out, p["to_out.1.g"].reshape(-1)) + x


def training_steps(sd, u, x_start, cond, ts, noises, lr: float, sched):
    """Reference training loop of the UNet (train.py:110-177 with --run_diff: optim.Adam over the UNet's parameters, loss =
    p_losses' l1 objective) on the CPU: `sd` parameters keyed 'diff_model.*', one (t, noise) per step.  Returns the loss of
    every step (before its update) and the final parameters."""
    from oracle import ldc_oracle as O
    params = {k: v.clone().requires_grad_() for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    losses = []
    for t, noise in zip(ts, noises):
        opt.zero_grad()
        x_t = q_sample(sched, x_start, t, noise)
        out = O.unet_forward(params, u, x_t, t, cond)
        loss = p_losses_objective(sched, out, noise, t)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, {k: v.detach() for k, v in params.items()}
