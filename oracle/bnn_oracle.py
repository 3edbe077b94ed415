"""CPU oracle for the Bayesian attack / expected-loss-gradient hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under robustbnns_amd/ may import this module; it is
used by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg, and
there only as the checker / the reported CPU baseline, never as the product path.

Parity status: PINNED for rows a1, a3-a10 of SURVEY.md section 8 — every function here is
checked in tests/test_oracle_golden.py against tests/golden/*.npz, which were produced
by running the reference's own functions (tests/golden/make_golden.py).  Row a2 (the SVI
weight draw) is "parity unpinned": its arithmetic lives in pyro-ppl==1.3.0 / torch==1.4.0
(requirements.txt:45,56), neither present here; `svi_materialize` restates
model_bnn.py:121-130 (w = loc + softplus(scale) * eps) given explicit eps.

Two restatements of the same algorithm live here:

* closed form, batched over all points and samples (fp32 or fp64) — what the HIP
  kernels are compared with;
* `loop_*`: the reference's own loop nest (adversarialAttacks.py:118 -> :95 ->
  model_bnn.py:251; batch 1, autograd, one Python iteration per point / PGD iteration /
  posterior sample) — the honest "reference --device=cpu" stand-in timed by bench.py.

A posterior is a dict of stacked tensors (one leading axis S = posterior samples), keyed
like the reference's state_dict (model_nn.py:77-91):
  fc :  "model.1.weight" [S,H,D]  "model.1.bias" [S,H]  "model.3.weight" [S,C,H]  "model.3.bias" [S,C]
  fc2:  + "model.3.*" [S,H,H]/[S,H] and "model.5.*" [S,C,H]/[S,C]
  conv: "model.0.*" [S,32,Cin,5,5], "model.3.*" [S,Hc,32,5,5], "model.7.*" [S,C,F]
"""
import math

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01          # torch.nn.LeakyReLU() default, model_nn.py:68-69


# --------------------------------------------------------------------------- helpers
def _act(a, act):
    """model_nn.py:66-75"""
    if act == "relu":
        return torch.relu(a)
    if act == "leaky":
        return torch.where(a > 0, a, a * LEAKY_SLOPE)
    if act == "sigm":
        return torch.sigmoid(a)
    if act == "tanh":
        return torch.tanh(a)
    raise AssertionError("\nWrong activation name.")


def _act_grad(a, act):
    if act == "relu":
        return (a > 0).to(a.dtype)
    if act == "leaky":
        return torch.where(a > 0, torch.ones_like(a), torch.full_like(a, LEAKY_SLOPE))
    if act == "sigm":
        s = torch.sigmoid(a)
        return s * (1 - s)
    if act == "tanh":
        t = torch.tanh(a)
        return 1 - t * t
    raise AssertionError("\nWrong activation name.")


def mlp_layers(post, arch):
    """Stacked (W[S,out,in], b[S,out]) per Linear of the fc / fc2 nets (model_nn.py:77-91)."""
    if arch == "fc":
        ks = ["model.1", "model.3"]
    elif arch == "fc2":
        ks = ["model.1", "model.3", "model.5"]
    else:
        raise NotImplementedError(arch)
    return [(post[k + ".weight"], post[k + ".bias"]) for k in ks]


def select(post, idx):
    """model_bnn.py:246-252: `seeds` are indices into the stored samples."""
    idx = torch.as_tensor(list(idx), dtype=torch.long)
    return {k: v[idx] for k, v in post.items()}


def cast(post, dtype):
    return {k: v.to(dtype) for k, v in post.items()}


# ------------------------------------------------------------ a1: per-sample network
def nn_logits(x, post, arch, act):
    """NN.forward for every stacked sample at once -> logits [S,N,C] (model_nn.py:126-141)."""
    if arch in ("fc", "fc2"):
        h = x.reshape(x.shape[0], -1).unsqueeze(0)                       # nn.Flatten
        layers = mlp_layers(post, arch)
        for li, (W, b) in enumerate(layers):
            h = torch.matmul(h, W.transpose(1, 2)) + b.unsqueeze(1)      # nn.Linear
            if li + 1 < len(layers):
                h = _act(h, act)
        return h
    if arch == "conv":                                                   # model_nn.py:98-106
        outs = []
        for s in range(post["model.0.weight"].shape[0]):
            h = F.conv2d(x, post["model.0.weight"][s], post["model.0.bias"][s])
            h = F.max_pool2d(_act(h, act), 2)
            h = F.conv2d(h, post["model.3.weight"][s], post["model.3.bias"][s])
            h = F.max_pool2d(_act(h, act), 2, stride=1)
            h = h.flatten(1)
            outs.append(F.linear(h, post["model.7.weight"][s], post["model.7.bias"][s]))
        return torch.stack(outs)
    raise NotImplementedError(arch)


def _mlp_forward_cache(xf, layers, act):
    """xf [N,D] -> (logits [S,N,C], pre-activations list)."""
    h = xf.unsqueeze(0)
    pre = []
    for li, (W, b) in enumerate(layers):
        a = torch.matmul(h, W.transpose(1, 2)) + b.unsqueeze(1)
        if li + 1 < len(layers):
            pre.append(a)
            h = _act(a, act)
        else:
            h = a
    return h, pre


def _mlp_input_grad(dz, layers, pre, act):
    """dz [S,N,C] = dL/dlogits -> per-sample dL/dx [S,N,D] (hand-rolled backward)."""
    d = dz
    for li in range(len(layers) - 1, -1, -1):
        W, _ = layers[li]
        d = torch.matmul(d, W)                                           # [S,N,in]
        if li > 0:
            d = d * _act_grad(pre[li - 1], act)
    return d


# ------------------------------------------------------------ a3/a4: BNN.forward
def bnn_forward(x, post, arch, act, n_samples=10, seeds=None):
    """BNN.forward HMC branch, model_bnn.py:198-202,243-258: mean over samples of softmax."""
    if seeds:
        if len(seeds) != n_samples:
            raise ValueError("Number of seeds should match number of samples.")
    if seeds is None:
        seeds = range(n_samples)
    z = nn_logits(x, select(post, seeds), arch, act)
    return torch.softmax(z, dim=-1).mean(0)


def ensemble_forward(x, post, arch, act, n_samples):
    """Ensemble_NN.forward, model_ensemble.py:57-67: mean of LOGITS of the first n members."""
    S = next(iter(post.values())).shape[0]
    if n_samples is not None and n_samples > S:
        raise ValueError("Maximum number of samples allowed is ", S)
    return nn_logits(x, select(post, range(S)[:n_samples]), arch, act).mean(0)


# ------------------------------------------------- dL/dlogits for the loss definitions
def _onehot(label, C, dtype):
    return F.one_hot(label, C).to(dtype)


def dz_mean_prob(p, label):
    """fgsm/pgd loss (adversarialAttacks.py:74-78): L = CE(mean_s p_s, y) = -log softmax(pbar)[y]
    — CrossEntropyLoss applied to PROBABILITIES (the double softmax)."""
    S, _, C = p.shape
    G = (torch.softmax(p.mean(0), -1) - _onehot(label, C, p.dtype)) / S          # dL/dp_s
    return p * (G.unsqueeze(0) - (G.unsqueeze(0) * p).sum(-1, keepdim=True))


def dz_per_sample(p, label):
    """loss_gradient (lossGradients.py:29-40): CE per sample on that sample's probabilities,
    gradients averaged afterwards (the 1/S is folded in here)."""
    S, _, C = p.shape
    G = (torch.softmax(p, -1) - _onehot(label, C, p.dtype).unsqueeze(0)) / S
    return p * (G - (G * p).sum(-1, keepdim=True))


def dz_mean_logit(z, label):
    """Ensemble_NN / deterministic NN under fgsm/pgd: CE on (mean) logits."""
    S, _, C = z.shape
    G = (torch.softmax(z.mean(0), -1) - _onehot(label, C, z.dtype)) / S
    return G.unsqueeze(0).expand(S, -1, -1)


def kink_margin(x, post, arch, act, n_samples):
    """Per point: the smallest |pre-activation| over the used samples and hidden units (fc/fc2, relu/leaky).
    act' jumps at 0, so the input gradient of a point whose margin is within fp32 rounding of the
    pre-activation (~1e-6 at these sizes) legitimately depends on summation order; parity tests exclude
    such points explicitly (same status as the sign(g) rule for adversarial images)."""
    if act not in ("relu", "leaky"):
        return torch.full((x.shape[0],), float("inf"), dtype=torch.float64)
    if arch == "conv":
        # conv: the gradient is discontinuous where (i) the maximum of a pooling window changes (gap between its two
        # largest pre-activations) and (ii) the pre-activation AT a window's maximum changes sign
        post = select(post, range(n_samples))
        margin = torch.full((x.shape[0],), float("inf"), dtype=torch.float64)
        for s in range(n_samples):
            a1 = F.conv2d(x, post["model.0.weight"][s], post["model.0.bias"][s])
            for a, k, st in ((a1, 2, 2), (None, 2, 1)):
                if a is None:
                    a = F.conv2d(F.max_pool2d(_act(a1, act), 2), post["model.3.weight"][s], post["model.3.bias"][s])
                win = F.unfold(a.reshape(-1, 1, a.shape[2], a.shape[3]), k, stride=st)          # [N*C, 4, L]
                top2 = win.topk(2, dim=1)[0]
                m = torch.minimum(top2[:, 0] - top2[:, 1], top2[:, 0].abs()).reshape(x.shape[0], -1).amin(1)
                margin = torch.minimum(margin, m.double())
        return margin
    if arch not in ("fc", "fc2"):
        return torch.full((x.shape[0],), float("inf"), dtype=torch.float64)
    layers = mlp_layers(select(post, range(n_samples)), arch)
    _, pre = _mlp_forward_cache(x.reshape(x.shape[0], -1), layers, act)
    return torch.stack([a.abs().amin(dim=(0, 2)) for a in pre]).amin(0).double()


# ------------------------------------------------------------ a5/a6: loss_gradient(s)
def _input_grad(x, label, post, arch, act, mode):
    """Summed-over-samples input gradient [N,*x.shape[1:]] under `mode`."""
    if arch in ("fc", "fc2"):
        layers = mlp_layers(post, arch)
        z, pre = _mlp_forward_cache(x.reshape(x.shape[0], -1), layers, act)
        if mode == "mean_logit":
            dz = dz_mean_logit(z, label)
        else:
            p = torch.softmax(z, -1)
            dz = dz_mean_prob(p, label) if mode == "mean_prob" else dz_per_sample(p, label)
        return _mlp_input_grad(dz, layers, pre, act).sum(0).reshape(x.shape)
    # conv: the oracle lets autograd do the backward (checker only)
    xr = x.detach().clone().requires_grad_(True)
    z = nn_logits(xr, post, arch, act)
    S, N, C = z.shape
    if mode == "mean_logit":
        loss = F.cross_entropy(z.mean(0), label, reduction="sum")
    elif mode == "mean_prob":
        loss = F.cross_entropy(torch.softmax(z, -1).mean(0), label, reduction="sum")
    else:
        loss = F.cross_entropy(torch.softmax(z, -1).reshape(S * N, C), label.repeat(S), reduction="sum") / S
    loss.backward()
    return xr.grad.detach()


def loss_gradients(x, y_onehot, post, arch, act, n_samples):
    """lossGradients.loss_gradient for every point of x at once (lossGradients.py:20-40,52-62):
    mean_i d/dx CE(p_i(x), y), i = seeds 0..n_samples-1.  Returns x's shape."""
    label = y_onehot.argmax(-1)
    return _input_grad(x, label, select(post, range(n_samples)), arch, act, "per_sample")


def meanprob_gradients(x, label, post, arch, act, n_samples, kind="bnn"):
    """The gradient whose sign fgsm/pgd take (adversarialAttacks.py:74-79)."""
    mode = "mean_prob" if kind == "bnn" else "mean_logit"
    return _input_grad(x, label, select(post, range(n_samples)), arch, act, mode)


# ------------------------------------------------------------ a7/a8: fgsm / pgd
def fgsm_attack(x, label, post, arch, act, n_samples, hyperparams=None, kind="bnn"):
    """adversarialAttacks.py:69-83, all points at once."""
    epsilon = hyperparams["epsilon"] if hyperparams is not None else 0.3
    g = meanprob_gradients(x, label, post, arch, act, n_samples, kind)
    return torch.clamp(x + epsilon * g.sign(), 0, 1)


def pgd_params(x, hyperparams):
    """adversarialAttacks.py:88-91: alpha = 2/image.max() per image (from the clean image)."""
    if hyperparams is not None:
        flat = x.reshape(x.shape[0], -1)
        alpha = (2 / flat.max(dim=1)[0]).reshape((-1,) + (1,) * (x.dim() - 1))
        return hyperparams["epsilon"], alpha, 40
    return 0.5, torch.full((x.shape[0],) + (1,) * (x.dim() - 1), 2 / 225, dtype=x.dtype), 40


def pgd_step(xi, x0, label, post, arch, act, n_samples, epsilon, alpha, kind="bnn"):
    """ONE iteration of adversarialAttacks.py:95-105 from the iterate xi towards the eps-ball around x0 (alpha: per-image
    [N,1,..] tensor or a float), in the reference's fp32 operation order."""
    g = meanprob_gradients(xi, label, post, arch, act, n_samples, kind)
    a = alpha.to(xi.dtype) if torch.is_tensor(alpha) else alpha
    pert = xi + a * g.sign()
    eta = torch.clamp(pert - x0, min=-epsilon, max=epsilon)
    return torch.clamp(x0 + eta, min=0, max=1)


def pgd_attack(x, label, post, arch, act, n_samples, hyperparams=None, kind="bnn", iters=None):
    """adversarialAttacks.py:86-108, all points at once."""
    epsilon, alpha, it = pgd_params(x, hyperparams)
    it = it if iters is None else iters
    x0 = x.clone()
    xi = x.clone()
    for _ in range(it):
        xi = pgd_step(xi, x0, label, post, arch, act, n_samples, epsilon, alpha, kind)
    return xi


# ------------------------------------------------------------ a10: evaluation
def softmax_difference(orig, adv):
    """adversarialAttacks.py:30-51 (softmax applied AGAIN to whatever forward returned)."""
    d = (torch.softmax(orig, -1) - torch.softmax(adv, -1)).abs().max(dim=-1)[0]
    if d.min() < 0. or d.max() > 1.:
        raise ValueError("Softmax difference should be in [0,1]")
    return d


def attack_evaluation(x, x_attack, y_onehot, post, arch, act, n_samples, kind="bnn"):
    """adversarialAttacks.py:151-198 -> (orig_acc %, adv_acc %, softmax_rob [N])."""
    def fwd(v):
        if kind == "bnn":
            return bnn_forward(v, post, arch, act, n_samples)
        return ensemble_forward(v, post, arch, act, n_samples)
    o, a = fwd(x), fwd(x_attack)
    lab = y_onehot.argmax(-1)
    oc = float((o.argmax(-1) == lab).sum().item())
    ac = float((a.argmax(-1) == lab).sum().item())
    return 100 * oc / len(x), 100 * ac / len(x), 1 - softmax_difference(o, a)


# ------------------------------------------------------------ a2: SVI draw (unpinned)
def svi_materialize(loc, scale_raw, eps):
    """model_bnn.py:124-130: Normal(loc, softplus(scale)).rsample() = loc + softplus(scale)*eps.
    loc/scale_raw: dict key -> tensor; eps: dict key -> [S,...].  PARITY UNPINNED (pyro absent)."""
    return {k: loc[k].unsqueeze(0) + F.softplus(scale_raw[k]).unsqueeze(0) * eps[k] for k in loc}


# The in-place draw (robustbnns_amd/csrc/rbnn_svi.hip, rbnn_svi_draw) generates eps itself: Philox4x32-10 + Box-Muller.  Restated
# here in numpy so that the kernel is checked element by element (generator, counter layout, Box-Muller pairing, the padding rule).
# PARITY UNPINNED against pyro-ppl 1.3.0 / torch 1.4.0's own stream, like everything about the draw.
SVI_TENSOR_IDS = {"W1": 0, "b1": 1, "Wm": 2, "bm": 3, "W2": 4, "b2": 5}


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy uint32 arrays (broadcastable) -> four uint32 arrays; Random123's Philox4x32 with 10 rounds."""
    import numpy as np
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c0, c1, c2, c3, k0, k1 = (np.asarray(v, dtype=np.uint32) for v in np.broadcast_arrays(c0, c1, c2, c3, k0, k1))
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0, p1 = M0 * c0.astype(np.uint64), M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = k0 + W0, k1 + W1
    return c0, c1, c2, c3


def philox_normals(key, draw_id, tensor_id, sample, rows, cols, sample_is_key=False):
    """eps [rows, cols] of one tensor of one sample, as rbnn_svi_draw generates it: element (r, c) is component c % 4 of the Philox
    block with counter (r * ceil(cols/4) + c // 4, tensor_id, sample — or 0 when the key IS the sample's seed —, draw_id) under the
    64-bit key; components (0,1) and (2,3) are Box-Muller pairs of u = (x + 0.5) 2^-32: sqrt(-2 ln u1) * (cos, sin)(2 pi u2)."""
    import numpy as np
    Q = (cols + 3) // 4
    q = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(Q) + np.arange(Q, dtype=np.uint64)[None, :]).astype(np.uint32)
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    x = philox4x32_10(q, np.uint32(tensor_id), np.uint32(0 if sample_is_key else sample), np.uint32(draw_id & 0xFFFFFFFF),
                      np.uint32(key & 0xFFFFFFFF), np.uint32(key >> 32))
    u = [np.minimum((xi.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32), np.float32(1.0)).astype(np.float64) for xi in x]
    u[0], u[2] = np.minimum(u[0], 0.99999994), np.minimum(u[2], 0.99999994)
    r01, r23 = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    n = np.stack([r01 * np.cos(2 * np.pi * u[1]), r01 * np.sin(2 * np.pi * u[1]),
                  r23 * np.cos(2 * np.pi * u[3]), r23 * np.sin(2 * np.pi * u[3])], axis=-1)            # [rows, Q, 4]
    return n.reshape(rows, 4 * Q)[:, :cols]


def svi_draw_philox(loc, scale_raw, key, draw_id, n_samples, sample_keys=None):
    """The stacked weights rbnn_svi_draw writes: loc / scale_raw: dict name in SVI_TENSOR_IDS -> tensor (matrices [rows, cols], vectors
    [n]); returns dict name -> float64 tensor [S, ...] and the eps used (same layout)."""
    import numpy as np
    W, E = {}, {}
    for name, l in loc.items():
        l64, sp = l.double(), F.softplus(scale_raw[name].double())
        rows, cols = (l.shape if l.dim() == 2 else (1, l.numel()))
        eps = np.stack([philox_normals(sample_keys[s] if sample_keys is not None else key, draw_id, SVI_TENSOR_IDS[name], s, rows, cols,
                                       sample_is_key=sample_keys is not None) for s in range(n_samples)])
        E[name] = torch.from_numpy(eps).reshape((n_samples,) + tuple(l.shape))
        W[name] = l64.unsqueeze(0) + sp.unsqueeze(0) * E[name]
    return W, E


# =====================================================================================
# Loop-structured port: the reference's nest, batch 1, autograd.  CPU baseline for bench.py.
# =====================================================================================
def _net_forward_one(image, post, s, arch, act):
    """One stored sample's NN.forward on a batch-1 image (model_nn.py:126-141)."""
    if arch in ("fc", "fc2"):
        h = image.flatten(1)
        layers = mlp_layers(post, arch)
        for li, (W, b) in enumerate(layers):
            h = F.linear(h, W[s], b[s])
            if li + 1 < len(layers):
                h = _act(h, act)
        return h
    return nn_logits(image, select(post, [s]), arch, act)[0]


def loop_bnn_forward(image, post, arch, act, n_samples, seeds=None):
    """model_bnn.py:243-258"""
    preds = []
    for seed in (range(n_samples) if seeds is None else seeds):
        preds.append(torch.softmax(_net_forward_one(image, post, seed, arch, act), dim=-1))
    return torch.stack(preds).mean(0)


def loop_loss_gradient(image, label_onehot, post, arch, act, n_samples):
    """lossGradients.py:20-40: one backward per sample."""
    image = image.unsqueeze(0)
    label = label_onehot.argmax(-1).unsqueeze(0)
    grads = []
    for i in range(n_samples):
        x_copy = image.clone().requires_grad_(True)
        out = loop_bnn_forward(x_copy, post, arch, act, 1, seeds=[i])
        F.cross_entropy(out, label).backward()
        grads.append(x_copy.grad.detach()[0].clone())
    return torch.stack(grads, 0).mean(0)


def loop_fgsm_attack(image, label, post, arch, act, n_samples, hyperparams=None):
    """adversarialAttacks.py:69-83"""
    epsilon = hyperparams["epsilon"] if hyperparams is not None else 0.3
    image = image.clone().requires_grad_(True)
    out = loop_bnn_forward(image, post, arch, act, n_samples)
    F.cross_entropy(out, label).backward()
    return torch.clamp(image + epsilon * image.grad.sign(), 0, 1).detach()


def loop_pgd_attack(image, label, post, arch, act, n_samples, hyperparams=None, iters=40):
    """adversarialAttacks.py:86-108"""
    if hyperparams is not None:
        epsilon, alpha = hyperparams["epsilon"], 2 / image.max()
    else:
        epsilon, alpha = 0.5, 2 / 225
    original = image.clone()
    for _ in range(iters):
        image = image.detach().requires_grad_(True)
        out = loop_bnn_forward(image, post, arch, act, n_samples)
        F.cross_entropy(out, label).backward()
        pert = image + alpha * image.grad.sign()
        eta = torch.clamp(pert - original, min=-epsilon, max=epsilon)
        image = torch.clamp(original + eta, min=0, max=1).detach()
    return image


def loop_attack(x, y_onehot, post, arch, act, method, n_samples, hyperparams=None):
    """adversarialAttacks.py:118-133 (no file side effects)."""
    res = []
    for idx in range(len(x)):
        image = x[idx].unsqueeze(0)
        label = y_onehot[idx].argmax(-1).unsqueeze(0)
        fn = loop_fgsm_attack if method == "fgsm" else loop_pgd_attack
        res.append(fn(image, label, post, arch, act, n_samples, hyperparams))
    return torch.cat(res)


# ------------------------------------------------------------ synthetic posteriors
def param_shapes(arch, D, H, C, in_ch=1, head=None):
    if arch == "fc":
        return [("model.1.weight", (H, D)), ("model.1.bias", (H,)),
                ("model.3.weight", (C, H)), ("model.3.bias", (C,))]
    if arch == "fc2":
        return [("model.1.weight", (H, D)), ("model.1.bias", (H,)),
                ("model.3.weight", (H, H)), ("model.3.bias", (H,)),
                ("model.5.weight", (C, H)), ("model.5.bias", (C,))]
    if arch == "conv":
        return [("model.0.weight", (32, in_ch, 5, 5)), ("model.0.bias", (32,)),
                ("model.3.weight", (H, 32, 5, 5)), ("model.3.bias", (H,)),
                # model_nn.py:106 sizes the head int(H/16) * D (= 49*H, right for 1x28x28 only); `head` = the build-defined
                # flattened conv output for other input sizes (81*H at 3x32x32, BASELINE config 5: parity unpinned)
                ("model.7.weight", (C, int(H / 16) * D if head is None else head)), ("model.7.bias", (C,))]
    raise NotImplementedError(arch)


def synthetic_posterior(arch, D, H, C, S, std, in_ch=1, head=None):
    """The HMC-style synthetic posterior of tests/golden/make_golden.py::fill_net:
    sample i = manual_seed(100+i), every parameter tensor ~ N(0, std^2) in
    nn.Module.parameters() order.  Bit-identical to the fixtures' weights (checked by sha256)."""
    shapes = param_shapes(arch, D, H, C, in_ch, head)
    out = {k: torch.empty((S,) + shp, dtype=torch.float32) for k, shp in shapes}
    for i in range(S):
        torch.manual_seed(100 + i)
        for k, shp in shapes:
            out[k][i] = torch.empty(shp, dtype=torch.float32).normal_(0.0, std)
    return out


def synthetic_inputs(n, shape, n_classes, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n,) + tuple(shape), generator=g, dtype=torch.float32)
    y = torch.randint(0, n_classes, (n,), generator=g)
    onehot = torch.zeros(n, n_classes, dtype=torch.float32)
    onehot[torch.arange(n), y] = 1.0
    return x, onehot
