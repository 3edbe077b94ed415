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


def _worker(rank, world, port, arch, q, comm_blocks=2):
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
        D, H, C, S, N = 784, 32, 10, 6, 12
        post = O.synthetic_posterior(arch, D, H, C, S, 0.06)
        x, y = O.synthetic_inputs(N, (1, 28, 28), C, seed=5)
        full = StackedPosterior(arch, "leaky", (1, 28, 28), C, H, post, "cpu")
        eng = AttackEngine(full.shard(rank, world), kernels=FakeKernels(), group=dist.group.WORLD)
        assert eng.post.S == S // world and eng.total_samples(eng.post.S) == S
        out = {"probs": eng.forward(x, eng.post.S), "lg": eng.loss_gradients(x, y, eng.post.S),
               "fgsm": eng.fgsm(x, y, eng.post.S, 0.3), "pgd": eng.pgd(x[:3], y[:3], eng.post.S, 0.3, iters=5),
               "gm": eng.gradient(eng.pad_inputs(x), y.argmax(-1).int(), None, eng.post.S, _hip.LOSS_MEAN_PROB).clone()}
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
