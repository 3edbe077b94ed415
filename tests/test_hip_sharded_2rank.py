"""Sample-sharded step with the REAL kernels in TWO processes (-m gpu): world_size 2 over gloo on device tensors, both ranks on cuda:0.

The 8-GPU job runs this engine code over RCCL with one rank per GPU; a 1-GPU box cannot host two RCCL ranks, but gloo accepts device
tensors, so the whole sharded step — every rank's forward / backward kernels through the C-ABI, the asynchronous all-reduces of
sum_s p_s and of the summed gradients, pipelined over point blocks — runs here across two processes and must reproduce the
single-process result on the full posterior: gradients to 1e-5 (the partial sums are added in a different order), FGSM images equal
except noise-level gradient components, identical replicas on both ranks.  Runs in the package's default precision (auto = triple at
H = 512) and on the fp32 MFMA, and for the conv architecture (ConvEngine: the plain, un-pipelined sharded sequence)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("built_library")]

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _problem(arch, dev, S_fc=6):
    """(full posterior, this-rank builder, x, y, D, S, N, PGD points) for the fc-512 case and a small conv net."""
    from oracle import bnn_oracle as O
    if arch == "fc":
        from robustbnns_amd.posterior import StackedPosterior
        D, H, C, S, N = 784, 512, 10, S_fc, 1100
        post = O.synthetic_posterior("fc", D, H, C, S, 0.05)
        full = StackedPosterior("fc", "leaky", (1, 28, 28), C, H, post, dev)
        x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed=5)
        return full, (lambda r, w: full.shard(r, w)), x, y, D, S, N, 600
    from robustbnns_amd.conv import ConvStackedPosterior
    if arch == "conv3":                                          # 3x32x32 (BASELINE config 5's geometry): conv1 over three input channels, the pair-tiled conv2 forward
        shape, Hc, C, S, N = (3, 32, 32), 32, 10, 4, 24
        D = shape[0] * shape[1] * shape[2]
        q2 = ((shape[1] - 4) // 2) - 5
        post = O.synthetic_posterior("conv", D, Hc, C, S, 0.05, in_ch=shape[0], head=q2 * q2 * Hc)
        full = ConvStackedPosterior("leaky", shape, C, Hc, post, dev)
        x, y = O.synthetic_inputs(N, shape, C, seed=7)
        part = lambda r, w: ConvStackedPosterior("leaky", shape, C, Hc, {k: v[r * S // w:(r + 1) * S // w] for k, v in post.items()}, dev)
        return full, part, x, y, D, S, N, 8
    D, Hc, C, S, N = 784, 32, 10, 4, 48
    post = O.synthetic_posterior("conv", D, Hc, C, S, 0.05)
    full = ConvStackedPosterior("leaky", (1, 28, 28), C, Hc, post, dev)
    x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed=6)
    part = lambda r, w: ConvStackedPosterior("leaky", (1, 28, 28), C, Hc, {k: v[r * S // w:(r + 1) * S // w] for k, v in post.items()}, dev)
    return full, part, x, y, D, S, N, 16


def _worker(rank, world, port, precision, q, arch="fc", S_fc=6):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RBNN_COMM_BLOCKS"], os.environ["RBNN_COMM_MIN_POINTS"] = "2", "256"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from robustbnns_amd import _hip
        from robustbnns_amd.factory import make_engine
        dev = "cuda:0"
        full, part, x, y, D, S, N, NP = _problem(arch, dev, S_fc)
        eng = make_engine(part(rank, world), group=dist.group.WORLD, total_samples=S, precision=precision)
        assert eng.world == 2 and eng.post.S == S * (rank + 1) // world - S * rank // world      # shards may be unequal (S = 7: 3 + 4)
        if arch == "fc":
            assert eng._comm_blocks(N) == 2                     # pipelined over two point blocks; ConvEngine keeps the plain sequence
        lab = y.argmax(-1).int().to(dev)
        out = {"probs": eng.forward(x, eng.post.S).cpu(), "lg": eng.loss_gradients(x, y, eng.post.S).cpu(),
               "gm": eng.gradient(eng.pad_inputs(x), lab, None, eng.post.S, _hip.LOSS_MEAN_PROB)[:, :D].cpu().clone(),
               "fgsm": eng.fgsm(x, y, eng.post.S, 0.3).cpu(), "pgd": eng.pgd(x[:NP], y[:NP], eng.post.S, 0.3, iters=4).cpu()}
        # BASELINE config 4's step on ONE forward, sample-sharded (its first gradient exchange overlapped with the second backward): the two
        # calls' results, bit for bit, on every rank (the fc engine pipelines fgsm() over two point blocks here, the shared call does not)
        both = eng.loss_gradients_and_fgsm(x, y, eng.post.S, 0.3)
        d_lg, d_adv = (both[0].cpu() - out["lg"]).abs(), (both[1].cpu() - out["fgsm"]).abs()
        assert torch.equal(both[0].cpu(), out["lg"]) and torch.equal(both[1].cpu(), out["fgsm"]), (
            f"rank {rank}: shared call vs two calls — gradients differ in {int((d_lg > 0).sum())} of {d_lg.numel()} (max {float(d_lg.max()):.3e}, "
            f"points {sorted(set((d_lg.reshape(N, -1) > 0).any(1).nonzero().flatten().tolist()))[:12]}), adversarial inputs in {int((d_adv > 0).sum())} (max {float(d_adv.max()):.3e})")
        torch.cuda.synchronize()
        t = out["fgsm"].clone()                                  # every rank holds the same (replicated) adversarial images
        dist.broadcast(t, src=0)
        assert torch.equal(t, out["fgsm"])
        if rank == 0:
            single = make_engine(full, precision=precision)
            ref = {"probs": single.forward(x, S).cpu(), "lg": single.loss_gradients(x, y, S).cpu(),
                   "gm": single.gradient(single.pad_inputs(x), lab, None, S, _hip.LOSS_MEAN_PROB)[:, :D].cpu().clone(),
                   "fgsm": single.fgsm(x, y, S, 0.3).cpu(), "pgd": single.pgd(x[:NP], y[:NP], S, 0.3, iters=4).cpu()}
            errs = {"mode": eng.precision}
            for k in ("probs", "lg", "gm"):
                a, b = out[k].reshape(N, -1).double(), ref[k].reshape(N, -1).double()
                errs[k] = float(((a - b).abs().max(1)[0] / b.abs().max(1)[0]).max())
            safe = ref["gm"].abs() > 1e-3 * ref["gm"].abs().max(1, keepdim=True)[0]
            errs["fgsm_bad"] = int((((out["fgsm"] - ref["fgsm"]).abs().reshape(N, -1) > 1e-6) & safe).sum())
            errs["pgd_frac"] = float(((out["pgd"] - ref["pgd"]).abs() > 1e-6).double().mean())
            q.put(errs)
    except Exception as exc:                                     # report instead of hanging the parent on q.get
        if rank == 0:
            q.put({"error": repr(exc)})
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("arch,precision,S_fc", [("fc", "auto", 6), ("fc", "exact", 6), ("conv", "auto", 6), ("fc", "auto", 7)])
def test_two_ranks_real_kernels_match_single_process(arch, precision, S_fc):
    """S_fc = 7: UNEQUAL sample shards (3 + 4) — BASELINE config 5's n_samples = 500 over 8 GPUs is 62 / 63 per rank."""
    assert torch.cuda.is_available(), "this test needs the MI355X"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, precision, q, arch, S_fc)) for r in range(2)]
    for p in procs:
        p.start()
    errs = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert "error" not in errs, errs
    assert all(p.exitcode == 0 for p in procs)
    print(f"[2 ranks, real kernels, {errs['mode']}] {errs}")
    assert errs["mode"] == ("triple" if precision == "auto" else "exact")
    assert errs["probs"] < 1e-6 and errs["lg"] < 1e-5 and errs["gm"] < 1e-5, errs
    assert errs["fgsm_bad"] == 0 and errs["pgd_frac"] < 0.02, errs


# ------------------------------------------------------------------ point-sharded: the zero-communication spelling (bench.py --shard points)
def _worker_points(rank, world, port, q):
    """Every rank holds ALL samples and its own block of points (second bullet; strong scaling of C2): no collective in the data
    path at all — only the final gather of the adversarial points, here over gloo."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from robustbnns_amd import _hip
        from robustbnns_amd.factory import make_engine
        dev = "cuda:0"
        full, _, x, y, D, S, N, NP = _problem("fc", dev)
        cut = [0, 600, N]                                        # unequal point blocks
        lo, hi = cut[rank], cut[rank + 1]
        eng = make_engine(full)                                  # no group: nothing to exchange
        assert eng.world == 1 and eng.precision == "triple"
        xs, ys = x[lo:hi], y[lo:hi]
        mine = {"probs": eng.forward(xs, S), "lg": eng.loss_gradients(xs, ys, S), "fgsm": eng.fgsm(xs, ys, S, 0.3),
                "pgd": eng.pgd(xs[:100], ys[:100], S, 0.3, iters=4)}
        parts = [None] * world                                   # the job's ONLY communication: gather the per-rank blocks
        dist.all_gather_object(parts, {k: v.cpu() for k, v in mine.items()})
        got = {k: torch.cat([parts[r][k] for r in range(world)]) for k in mine}
        if rank == 0:
            single = make_engine(full)
            lab = y.argmax(-1).int().to(dev)
            ref = {"probs": single.forward(x, S).cpu(), "lg": single.loss_gradients(x, y, S).cpu(), "fgsm": single.fgsm(x, y, S, 0.3).cpu(),
                   "pgd": torch.cat([single.pgd(x[:100], y[:100], S, 0.3, iters=4), single.pgd(x[600:700], y[600:700], S, 0.3, iters=4)]).cpu()}
            gm = single.gradient(single.pad_inputs(x), lab, None, S, _hip.LOSS_MEAN_PROB)[:, :D].cpu()
            errs = {}
            for k in ("probs", "lg"):
                a, b = got[k].reshape(N, -1).double(), ref[k].reshape(N, -1).double()
                errs[k] = float(((a - b).abs().max(1)[0] / b.abs().max(1)[0]).max())
            safe = gm.abs() > 1e-3 * gm.abs().max(1, keepdim=True)[0]
            errs["fgsm_bad"] = int((((got["fgsm"] - ref["fgsm"]).abs().reshape(N, -1) > 1e-6) & safe).sum())
            errs["pgd_frac"] = float(((got["pgd"] - ref["pgd"]).abs() > 1e-6).double().mean())
            q.put(errs)
    except Exception as exc:
        if rank == 0:
            q.put({"error": repr(exc)})
        raise
    finally:
        dist.destroy_process_group()


def test_point_sharded_two_ranks_real_kernels_match_single_process():
    assert torch.cuda.is_available(), "this test needs the MI355X"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker_points, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    errs = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert "error" not in errs, errs
    assert all(p.exitcode == 0 for p in procs)
    print(f"[2 ranks, point-sharded, real kernels] {errs}")
    # a point's result does not depend on which other points share the launch except through the slab plan (samples per partial sum)
    assert errs["probs"] < 1e-6 and errs["lg"] < 1e-5 and errs["fgsm_bad"] == 0 and errs["pgd_frac"] < 0.02, errs


def _concurrent_worker(rank, world, port, q, arch):
    """This process's own posterior shard, no group, no collectives: forward calls, then whole steps, while the OTHER process runs its own."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from robustbnns_amd.factory import make_engine
        dev = "cuda:0"
        full, part, x, y, D, S, N, NP = _problem(arch, dev)
        eng = make_engine(part(rank, world), precision="auto")
        xd, Sl = x.to(dev), eng.post.S
        keys = ("P1", "st1", "Q2", "st2", "P") if arch != "fc" else ()      # conv: the forward's whole workspace, stage by stage
        ws = eng.workspace(N, Sl) if keys else {}
        dist.barrier()
        p0 = eng.forward(xd, Sl).clone(); torch.cuda.synchronize()
        ref = {k: ws[k].clone() for k in keys}
        bad = []
        for i in range(40):
            pi = eng.forward(xd, Sl); torch.cuda.synchronize()
            d = {k: int((ws[k] != ref[k]).sum()) for k in keys}
            d["probs"] = int((pi != p0).sum())
            if any(d.values()):
                bad.append((i, d))
        # ... and the whole step (forward + the backward kernels): the expected loss gradients of 20 calls against the first call's
        g0 = eng.loss_gradients(xd, y, Sl).clone()
        for i in range(20):
            n_diff = int((eng.loss_gradients(xd, y, Sl) != g0).sum())
            if n_diff:
                bad.append((100 + i, {"loss_gradients": n_diff}))
        dist.barrier()
        q.put((rank, bad[:3], len(bad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("arch", ["conv", "conv3", "fc"])
def test_two_processes_sharing_the_gpu_do_not_disturb_each_other(arch):
    """Found with this setup in round 5 (profiles/r05w): conv1_pool_kernel's packed FMA had its broadcast operand as src1 (op_sel on src1) — a form that
    gfx950 does not execute reliably when waves of another kernel share the SIMD: with two processes at once, one in seven forward calls left a P1 that
    differed from the process's own reference (the low result lane took the other half of a patch pair).  Two processes, different posteriors (conv on 1x28x28 and on 3x32x32; the fc-512 net), no
    collectives: every call's whole forward workspace — and then the gradients of the whole step — must reproduce the first call's, bit for bit, in both."""
    assert torch.cuda.is_available(), "this test needs the MI355X"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_concurrent_worker, args=(r, 2, port, q, arch)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    print(f"[2 processes at once, {arch}: forward x 40 + loss_gradients x 20] differing calls: {[(r, n) for r, _, n in res]}")
    assert all(n == 0 for _, _, n in res), res
