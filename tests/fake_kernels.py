"""CPU test double for robustbnns_amd._hip.HipKernels — TESTS ONLY.

Implements the tensor-level kernel interface with the oracle's closed forms so that the engine's
orchestration (workspace handling, sample sharding, all-reduces under gloo, attack loops, file side
effects of the drivers) can run in CPU-only tests.  The product never constructs this class.
"""
import torch

from oracle import bnn_oracle as O
from robustbnns_amd import _hip


class FakeKernels:
    name = "fake-cpu"

    def __init__(self):
        self._real = _hip.HipKernels()          # host-only entry points (workspace sizes) come from the real library

    def workspace_sizes(self, net, N, S, chunk=0):
        return self._real.workspace_sizes(net, N, S, chunk)

    @staticmethod
    def _layers(net, sidx, S):
        idx = torch.arange(S) if sidx is None else sidx.long()
        L = [(net.W1[idx], net.b1[idx])]
        if net.arch == "fc2":
            L.append((net.Wm[idx], net.bm[idx]))
        L.append((net.W2[idx], net.b2[idx]))
        return L

    def fc_forward(self, net, X, sidx, S, out_kind, ws):
        layers = self._layers(net, sidx, S)
        z, pre = O._mlp_forward_cache(X, layers, net.activation)
        out = z if out_kind == _hip.OUT_LOGITS else torch.softmax(z, -1)
        P = ws["P"].view(S, X.shape[0], _hip.CPAD)
        P.zero_()
        P[:, :, :net.C] = out
        ws["_fake_stash"] = (layers, pre, net.activation)      # like the HIP kernels: the backward state lives in the workspace

    def reduce_samples(self, P, S, N, C, scale, out):
        out[:, :C] = P.view(S, N, _hip.CPAD)[:, :, :C].sum(0) * scale

    def loss_dlogits(self, mode, P, Psum, G_up, labels, S, inv_S, N, C, dZ):
        p = P.view(S, N, _hip.CPAD)[:, :, :C]
        lab = None if labels is None else labels.long()
        if mode in (_hip.LOSS_UPSTREAM, _hip.LOSS_UPSTREAM_LOGIT):
            G = (G_up[:, :C] * inv_S).unsqueeze(0)
        elif mode == _hip.LOSS_PER_SAMPLE:
            G = (torch.softmax(p, -1) - torch.nn.functional.one_hot(lab, C).float().unsqueeze(0)) * inv_S
        else:
            G = ((torch.softmax(Psum[:, :C] * inv_S, -1) - torch.nn.functional.one_hot(lab, C).float()) * inv_S).unsqueeze(0)
        out = G.expand(S, N, C) if mode in (_hip.LOSS_MEAN_LOGIT, _hip.LOSS_UPSTREAM_LOGIT) else p * (G - (G * p).sum(-1, keepdim=True))
        d = dZ.view(S, N, _hip.CPAD)
        d.zero_()
        d[:, :, :C] = out

    def fc_input_grad(self, net, sidx, S, N, chunk, ws):
        layers, pre, act = ws["_fake_stash"]
        dz = ws["dZ"].view(S, N, _hip.CPAD)[:, :, :net.C]
        g = O._mlp_input_grad(dz, layers, pre, act)                   # [S, N, Dp]
        n_slabs = (S + chunk - 1) // chunk
        slabs = ws["slabs"].view(-1)[:n_slabs * N * net.Dp].view(n_slabs, N, net.Dp)
        for k in range(n_slabs):
            slabs[k] = g[k * chunk:(k + 1) * chunk].sum(0)
        return n_slabs

    def sum_slabs(self, slabs, K, N, Dp, scale, out):
        out[:, :Dp] = slabs.view(-1)[:K * N * Dp].view(K, N, Dp).sum(0) * scale

    def sum_slabs_norms(self, slabs, K, N, Dp, D, scale, out, linf, l2):
        self.sum_slabs(slabs, K, N, Dp, scale, out)
        linf.copy_(out[:, :D].abs().max(1)[0])
        l2.copy_(out[:, :D].norm(dim=1))

    def pgd_alpha(self, X0, D, alpha):
        alpha.copy_(2 / X0[:, :D].max(dim=1)[0])

    def attack_step(self, X, X0, G, K, slab_stride, ldg, alpha, alpha_scalar, eps, project, D):
        N = X.shape[0]
        flat = G.view(-1)
        g = sum(flat[k * slab_stride:k * slab_stride + N * ldg].view(N, ldg)[:, :D] for k in range(K))
        step = alpha.view(N, 1) if alpha is not None else alpha_scalar
        pert = X[:, :D] + step * g.sign()
        if project:
            pert = X0[:, :D] + torch.clamp(pert - X0[:, :D], -eps, eps)
        X[:, :D] = torch.clamp(pert, 0, 1)

    def eval_metrics(self, A, B, labels, C, counts, rob):
        lab = labels.long()
        counts[0] = int((A[:, :C].argmax(-1) == lab).sum())
        counts[1] = int((B[:, :C].argmax(-1) == lab).sum())
        rob.copy_(1 - (torch.softmax(A[:, :C], -1) - torch.softmax(B[:, :C], -1)).abs().max(-1)[0])

    def svi_materialize(self, loc, scale_raw, eps, out):
        out.copy_(loc.unsqueeze(0) + torch.nn.functional.softplus(scale_raw).unsqueeze(0) * eps)
