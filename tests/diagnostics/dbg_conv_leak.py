"""Two processes on one GPU, different 2-sample conv posteriors, no collectives.  When a forward call of one process leaves a P1 (conv1 output) that
differs from its own reference: do the differing elements hold the OTHER process's values for the same (sample slot, point, channel, position)?"""
import os, socket, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/diagnostics/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_hip_sharded_2rank import _problem
    from robustbnns_amd.factory import make_engine
    dev = "cuda:0"
    full, part, x, y, D, S, N, NP = _problem("conv", dev)
    eng = make_engine(part(rank, world), precision="auto")
    xd = x.to(dev)
    Sl = eng.post.S
    ws = eng.workspace(N, Sl)
    for turn in range(world):                                              # references, one process at a time
        if turn == rank:
            eng.forward(xd, Sl); torch.cuda.synchronize()
            ref = {k: ws[k].clone() for k in ("P1", "st1")}
            torch.save(ref["P1"].cpu(), f"/tmp/p1_ref_{rank}.pt")
            print(f"[rank {rank}] P1 at {hex(ws['P1'].data_ptr())}, K1w at {hex(eng.post.t['K1w'].data_ptr()) if hasattr(eng.post, 't') else '?'}", flush=True)
        dist.barrier()
    other = torch.load(f"/tmp/p1_ref_{1 - rank}.pt").to(dev)
    found = 0
    for i in range(150):
        eng.forward(xd, Sl); torch.cuda.synchronize()
        ne = ws["P1"] != ref["P1"]
        if int(ne.sum()):
            found += 1
            idx = ne.nonzero().flatten()
            a = ws["P1"][idx]
            same_as_other = int((a == other[idx]).sum())
            blocks = sorted(set((idx // 4608).tolist()))
            # per differing (s, n) block: is EVERYTHING in the block equal to the other process's block (the whole block computed from its weights)?
            whole = sum(int(torch.equal(ws["P1"][b * 4608:(b + 1) * 4608], other[b * 4608:(b + 1) * 4608])) for b in blocks)
            chans = sorted(set(((idx % 4608) // 144).tolist()))
            print(f"[rank {rank}] call {i}: {idx.numel()} elements differ from this process's reference; {same_as_other} of them EQUAL the other process's value at the same index; "
                  f"{len(blocks)} blocks, {whole} of them entirely equal to the other process's block; channels {chans}", flush=True)
            if found >= 3: break
    print(f"[rank {rank}] {found} differing calls", flush=True)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(300) for p in ps]
