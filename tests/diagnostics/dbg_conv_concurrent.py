"""Two processes on one GPU, each with its own 2-sample conv posterior and NO collectives: is the forward reproducible when they run (a) at the same
time, (b) one after the other?  Compares the forward's workspace (P1, st1, Q2, st2, P) of 40 calls with the first call's."""
import os, socket, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/diagnostics/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def run(eng, x, N, Sl, tag, reps=40):
    ws = eng.workspace(N, Sl)
    keys = [k for k in ("P1", "st1", "Q2", "st2", "P") if k in ws]
    eng.forward(x, Sl); torch.cuda.synchronize()
    ref = {k: ws[k].clone() for k in keys}
    bad = 0
    for i in range(reps):
        eng.forward(x, Sl); torch.cuda.synchronize()
        d = {k: int((ws[k] != ref[k]).sum()) for k in keys}
        if any(d.values()):
            bad += 1
            if bad <= 2: print(f"[{tag}] call {i}: {d}", flush=True)
    print(f"[{tag}] {bad} of {reps} calls differ from the first", flush=True)

def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_hip_sharded_2rank import _problem
    from robustbnns_amd.factory import make_engine
    dev = "cuda:0"
    full, part, x, y, D, S, N, NP = _problem("conv", dev)
    eng = make_engine(part(rank, world), precision="auto")            # this rank's two samples, no group: no collectives
    xd = x.to(dev)
    dist.barrier()
    run(eng, xd, N, eng.post.S, f"rank {rank}, both at once")
    dist.barrier()
    for turn in range(world):
        if turn == rank: run(eng, xd, N, eng.post.S, f"rank {rank}, alone")
        dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(300) for p in ps]
