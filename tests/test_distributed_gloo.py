"""Sample-sharded multi-process path on CPU: world_size 2 over gloo (the GPU job uses the same engine code
over RCCL).  Each rank holds half of the posterior samples; the engine all-reduces sum_s p_s and the summed
input gradients.  Kernels are the CPU test double (tests/fake_kernels.py); the result must equal the
single-process result on the full posterior."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.usefixtures("built_library")       # the workers' test double takes its workspace sizes from the library

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, arch, q, comm_blocks=2, dims=(784, 32, 10, 6, 12, (1, 28, 28))):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # the attack step is pipelined over point blocks with asynchronous all-reduces: force 2 (or 3, ragged) blocks on 12 points
    os.environ["RBNN_COMM_BLOCKS"], os.environ["RBNN_COMM_MIN_POINTS"] = str(comm_blocks), "2"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import FakeKernels
        from oracle import bnn_oracle as O
        from robustbnns_amd import _hip
        from robustbnns_amd.engine import AttackEngine
        from robustbnns_amd.posterior import StackedPosterior
        D, H, C, S, N, shape = dims
        post = O.synthetic_posterior(arch, D, H, C, S, 0.06 if D > 100 else 0.4)
        x, y = O.synthetic_inputs(N, shape, C, seed=5)
        full = StackedPosterior(arch, "leaky", shape, C, H, post, "cpu")
        eng = AttackEngine(full.shard(rank, world), kernels=FakeKernels(), group=dist.group.WORLD)
        # shards may be UNEQUAL (S = 7 over 2 ranks: 3 + 4; BASELINE config 5's n_samples = 500 over 8 GPUs: 62 / 63): rank r holds
        # floor(S (r+1) / G) - floor(S r / G) samples and the job's total comes from one all-reduce (or the caller's total_samples)
        assert eng.post.S == S * (rank + 1) // world - S * rank // world and eng.total_samples(eng.post.S) == S
        out = {"probs": eng.forward(x, eng.post.S), "lg": eng.loss_gradients(x, y, eng.post.S),
               "fgsm": eng.fgsm(x, y, eng.post.S, 0.3), "pgd": eng.pgd(x[:3], y[:3], eng.post.S, 0.3, iters=5),
               "gm": eng.gradient(eng.pad_inputs(x), y.argmax(-1).int(), None, eng.post.S, _hip.LOSS_MEAN_PROB).clone()}
        # BASELINE config 4's step on ONE forward: the same results as the two calls, on every rank
        both = eng.loss_gradients_and_fgsm(x, y, eng.post.S, 0.3)
        assert torch.equal(both[0], out["lg"]) and torch.equal(both[1], out["fgsm"])
        if rank == 0:
            single = AttackEngine(full, kernels=FakeKernels())
            ref = {"probs": single.forward(x, S), "lg": single.loss_gradients(x, y, S), "fgsm": single.fgsm(x, y, S, 0.3),
                   "pgd": single.pgd(x[:3], y[:3], S, 0.3, iters=5),
                   "gm": single.gradient(single.pad_inputs(x), y.argmax(-1).int(), None, S, _hip.LOSS_MEAN_PROB).clone()}
            errs = {}
            for k in ("probs", "lg", "gm"):
                a, b = out[k].reshape(N, -1).double(), ref[k].reshape(N, -1).double()
                errs[k] = float(((a - b).abs().max(1)[0] / b.abs().max(1)[0]).max())
            safe = ref["gm"][:, :D].abs() > 1e-3 * ref["gm"].abs().max(1, keepdim=True)[0]
            errs["fgsm_bad"] = int((((out["fgsm"] - ref["fgsm"]).abs().reshape(N, -1) > 1e-6) & safe).sum())
            errs["pgd_frac"] = float(((out["pgd"] - ref["pgd"]).abs() > 1e-6).double().mean())
            q.put(errs)
        # every rank must hold the same (replicated) adversarial images
        t = out["fgsm"].clone()
        dist.broadcast(t, src=0)
        assert torch.equal(t, out["fgsm"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("arch,comm_blocks", [("fc", 2), ("fc2", 1), ("fc", 5)])
def test_sample_sharded_world2_matches_single(arch, comm_blocks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, arch, q, comm_blocks)) for r in range(2)]
    for p in procs:
        p.start()
    errs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert errs["probs"] < 1e-6 and errs["lg"] < 1e-6 and errs["gm"] < 1e-6, errs
    assert errs["fgsm_bad"] == 0 and errs["pgd_frac"] < 0.02, errs


def _run_world(world, arch, comm_blocks, dims):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, arch, q, comm_blocks, dims)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        errs = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return errs


@pytest.mark.parametrize("arch,comm_blocks", [("fc", 1), ("fc", 2)])
def test_unequal_sample_shards_world2(arch, comm_blocks):
    """S = 7 over two ranks (3 + 4 samples): the mean over samples divides by the JOB's total on every rank, so the per-sample-loss
    gradients (lossGradients.py:40), the mean-probability gradients and both attacks equal the single-process result."""
    errs = _run_world(2, arch, comm_blocks, (784, 32, 10, 7, 12, (1, 28, 28)))
    assert errs["probs"] < 1e-6 and errs["lg"] < 1e-6 and errs["gm"] < 1e-6, errs
    assert errs["fgsm_bad"] == 0 and errs["pgd_frac"] < 0.02, errs


def test_sample_sharded_world8_c4_split_on_tiny_dims():
    """BASELINE config 4's layout — the posterior sharded 8 ways, `loss_gradients` (per-sample loss) + FGSM (mean-probability loss) over
    all points — and config 5's uneven split (here S = 20 over 8 ranks: 2 / 3 samples each, as 500 -> 62 / 63) with EIGHT processes over
    gloo on tiny dimensions (4 x 4 inputs, hidden 32, 3 classes): every rank's all-reduced result equals the single-process one."""
    errs = _run_world(8, "fc", 1, (16, 32, 3, 20, 24, (1, 4, 4)))
    assert errs["probs"] < 1e-6 and errs["lg"] < 1e-6 and errs["gm"] < 1e-6, errs
    assert errs["fgsm_bad"] == 0 and errs["pgd_frac"] < 0.02, errs
