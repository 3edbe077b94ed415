"""Which stage of the conv step is not reproducible when two processes share the GPU: every rank runs forward / gradient several times on the same
inputs and reports the calls that differ from the first."""
import os, socket, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/diagnostics/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port, sharded):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_hip_sharded_2rank import _problem
    from robustbnns_amd import _hip
    from robustbnns_amd.factory import make_engine
    dev = "cuda:0"
    full, part, x, y, D, S, N, NP = _problem("conv", dev)
    eng = make_engine(part(rank, world), group=dist.group.WORLD, total_samples=S, precision="auto") if sharded else make_engine(full, precision="auto")
    Sl = eng.post.S
    lab = y.argmax(-1).int().to(dev)
    def diff(name, fn, reps=6):
        ref = fn().cpu().clone()
        bad = []
        for i in range(reps):
            d = (fn().cpu() - ref).abs()
            if float(d.max()) > 0:
                bad.append((i, int((d > 0).sum()), float(d.max()), sorted(set((d.reshape(N, -1) > 0).any(1).nonzero().flatten().tolist()))[:8]))
        print(f"[rank {rank} sharded={sharded}] {name}: {'REPRODUCIBLE' if not bad else bad}", flush=True)
    # stage by stage: the forward's workspace after every call against the first call's
    ws = eng.workspace(N, Sl)
    keys = [k for k in ("P1", "st1", "Q2", "st2", "P") if k in ws]
    xin = x.to(dev) if os.environ.get("DBG_X_ON_DEVICE") == "1" else x
    eng.forward(xin, Sl); torch.cuda.synchronize()
    ref = {k: ws[k].clone() for k in keys}
    for i in range(30):
        eng.forward(xin, Sl); torch.cuda.synchronize()
        msg = []
        for k in keys:
            a, b = ws[k], ref[k]
            ne = (a != b) & ~((a != a) & (b != b)) if a.dtype.is_floating_point else (a != b)
            if int(ne.sum()):
                idx = ne.nonzero().flatten()
                msg.append(f"{k}: {int(ne.sum())} of {a.numel()} differ (first {idx[:4].tolist()}, last {int(idx[-1])})")
                if k == "P1":
                    blocks = sorted(set((idx // 4608).tolist()))
                    chans = sorted(set(((idx % 4608) // 144).tolist()))
                    poss = sorted(set((idx % 144).tolist()))
                    samp = [(int(j), float(b[j]), float(a[j])) for j in idx[:6]]
                    msg.append(f"P1 detail: (s,n) blocks {blocks[:10]} ({len(blocks)}), channels {chans[:34]}, positions {len(poss)} distinct {poss[:20]}, max |d| {float((a - b).abs().max()):.3e}, samples {samp}")
                    break
        print(f"[rank {rank} sharded={sharded}] forward call {i}: {'same' if not msg else '; '.join(msg)}", flush=True)
    diff("forward probs", lambda: eng.forward(x, Sl))
    diff("loss_gradients", lambda: eng.loss_gradients(x, y, Sl))
    diff("mean-prob gradient", lambda: eng.gradient(eng.pad_inputs(x), lab, None, Sl, _hip.LOSS_MEAN_PROB)[:, :D])
    diff("fgsm", lambda: eng.fgsm(x, y, Sl, 0.3))
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    for sharded in (False, True):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        ps = [ctx.Process(target=worker, args=(r, 2, port, sharded)) for r in range(2)]
        [p.start() for p in ps]; [p.join(300) for p in ps]
