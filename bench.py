#!/usr/bin/env python3
"""bench.py — attack-samples/s of the Bayesian FGSM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): MNIST-shaped fc-BNN 784->512->10 (leaky), FGSM eps=0.3 on
N=10 000 test points with S=100 posterior samples PER GPU, synthetic data (X ~ U[0,1), weights ~ N(0,0.05^2)),
everything resident in HBM before the timed region.  One step = one pass of the hot path over the batch:
forward over all (point, sample) pairs, CE on the mean probabilities, hand-rolled input gradient, sign/clamp.
attack-samples = points x posterior samples x iterations (1 for FGSM).

N GPUs: one process per GPU; the posterior is SAMPLE-sharded (each rank holds its own S=100 samples, so the
job has 100*N samples: weak scaling) and each step all-reduces sum_s p_s [N,16] and the summed gradients
[N,784] over RCCL/xGMI (north star).  `--shard points` replicates the samples and
splits points instead (no collective).

Precision (`--precision`, default auto): the two GEMMs run either on the fp32 MFMA ("exact") or as error-compensated
half pairs on the f16 MFMA pipe ("split": three f16 products per fp32 product, fp32 accumulation, 2^-22 per product;
parity-tested to the same 1e-5 bar).  auto = split for this workload.  The line's top level is the mode that ran;
at N=1 the other mode is timed afterwards and reported under `exact_fp32_mode` for reference.

The JSON line carries `roofline` for the dominant kernel (the input-gradient kernel; the forward kernel is listed
beside it under roofline.kernels), timed with HIP events on the launch stream inside the timed region, and
`cpu_baseline`: the loop-structured oracle port (oracle/bnn_oracle.py::loop_attack — the reference's batch-1
autograd nest) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, = fp32 vector peak
F16_MFMA_PEAK_TFLOPS = 2516.6          # v_mfma_f32_16x16x32_f16: 16 cyc/SIMD -> 1024 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz ("~2.5 PF dense")
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (input shape, hidden, classes, arch, act, S per GPU, N, method, iters, eps)
    "c2": dict(shape=(1, 28, 28), H=512, C=10, arch="fc", act="leaky", S=100, N=10000, method="fgsm", iters=1, eps=0.3,
               desc="MNIST fc-BNN 784->512->10 (leaky), FGSM eps=0.3, N=10000 points, S=100 samples/GPU"),
    "c3": dict(shape=(1, 28, 28), H=512, C=10, arch="fc", act="leaky", S=500, N=10000, method="pgd", iters=40, eps=0.3,
               desc="F-MNIST fc-BNN 784->512->10 (leaky), PGD T=40 eps=0.3, N=10000 points, S=500 samples/GPU"),
    "conv": dict(shape=(1, 28, 28), H=512, C=10, arch="conv", act="leaky", S=16, N=2048, method="fgsm", iters=1, eps=0.3,
                 desc="MNIST conv-BNN (conv5x5x32 - pool - conv5x5x512 - pool - fc, leaky), FGSM eps=0.3, N=2048 points, S=16 samples/GPU"),
    "fc2": dict(shape=(1, 28, 28), H=512, C=10, arch="fc2", act="leaky", S=100, N=10000, method="fgsm", iters=1, eps=0.3,
                desc="MNIST fc2-BNN 784->512->512->10 (leaky; the reference's saved model_1), FGSM eps=0.3, N=10000 points, S=100 samples/GPU"),
    "c1": dict(shape=(1, 2, 1), H=64, C=2, arch="fc", act="leaky", S=10, N=100, method="fgsm", iters=1, eps=0.3,
               desc="half-moons fc-BNN 2->64->2 (leaky), FGSM eps=0.3, N=100 points, S=10 samples/GPU"),
}


def make_problem(w, rank, device):
    """Synthetic inputs (same on every rank) and this rank's posterior samples (distinct per rank)."""
    D = w["shape"][0] * w["shape"][1] * w["shape"][2]
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((w["N"],) + w["shape"], generator=g, dtype=torch.float32)
    y = torch.randint(0, w["C"], (w["N"],), generator=g)
    gw = torch.Generator().manual_seed(100 + rank)
    std = 0.5 if D < 16 else 0.05
    S, H, C = w["S"], w["H"], w["C"]
    if w["arch"] == "conv":
        r = lambda *shape: torch.randn(S, *shape, generator=gw) * 0.03
        post = {"model.0.weight": r(32, 1, 5, 5), "model.0.bias": r(32), "model.3.weight": r(H, 32, 5, 5), "model.3.bias": r(H),
                "model.7.weight": r(C, 49 * H), "model.7.bias": r(C)}
        return x, y, post
    if w["arch"] == "fc2":
        post = {"model.1.weight": torch.randn(S, H, D, generator=gw) * std, "model.1.bias": torch.randn(S, H, generator=gw) * std,
                "model.3.weight": torch.randn(S, H, H, generator=gw) * std, "model.3.bias": torch.randn(S, H, generator=gw) * std,
                "model.5.weight": torch.randn(S, C, H, generator=gw) * std, "model.5.bias": torch.randn(S, C, generator=gw) * std}
        return x, y, post
    post = {"model.1.weight": torch.randn(S, H, D, generator=gw) * std, "model.1.bias": torch.randn(S, H, generator=gw) * std,
            "model.3.weight": torch.randn(S, C, H, generator=gw) * std, "model.3.bias": torch.randn(S, C, generator=gw) * std}
    return x, y, post


def cpu_baseline(w, x, y, post, budget_s):
    """The reference's loop nest (batch 1, autograd) on this host: bounded sample, linear in points."""
    from oracle import bnn_oracle as O
    onehot = torch.nn.functional.one_hot(y, w["C"]).float()
    hyper = {"epsilon": w["eps"]}
    fn = O.loop_fgsm_attack if w["method"] == "fgsm" else O.loop_pgd_attack
    done, t0 = 0, time.perf_counter()
    fn(x[0:1], y[0:1], post, w["arch"], w["act"], w["S"], hyper)           # untimed warm-up (thread pool, caches)
    t0 = time.perf_counter()
    while done < w["N"]:
        fn(x[done:done + 1], y[done:done + 1], post, w["arch"], w["act"], w["S"], hyper)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 4:
            break
    dt = time.perf_counter() - t0
    return {"value": done * w["S"] * w["iters"] / dt, "unit": "attack-samples/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"first {done} points of the workload ({w['method']}, S={w['S']}, T={w['iters']}), "
                      f"oracle loop_attack = reference nest batch-1 autograd, {dt:.1f} s, linear in points",
            "host_cpus": os.cpu_count(), "ms_per_point": 1e3 * dt / done}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="samples", choices=["samples", "points"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--points", type=int, default=0, help="override N (debug)")
    ap.add_argument("--samples", type=int, default=0, help="override S per GPU (debug)")
    ap.add_argument("--precision", default="auto", choices=["auto", "exact", "split"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    group = None
    if world > 1 or os.environ.get("RBNN_FORCE_COLLECTIVES") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)           # nccl == RCCL on ROCm
        group = dist.group.WORLD

    from robustbnns_amd import AttackEngine, StackedPosterior, _hip

    w = dict(WORKLOADS[args.workload])
    if args.points:
        w["N"] = args.points
    if args.samples:
        w["S"] = args.samples
    x, y, post = make_problem(w, rank if args.shard == "samples" else 0, device)
    D = x[0].numel()
    from robustbnns_amd.factory import make_engine, posterior_from_stacked
    sp = posterior_from_stacked(w["arch"], w["act"], w["shape"], w["C"], w["H"], post, device)

    class TimedKernels(_hip.HipKernels):
        """HIP events around the two GEMM kernels, on the stream they are launched on (torch's current stream)."""
        def __init__(self):
            super().__init__()
            self.ev = {"fc_forward": [], "fc_input_grad": []} if w["arch"] != "conv" else {"conv_forward": [], "conv_input_grad": []}
            self.on = False

        def _timed(self, name, fn, *a):
            if not self.on:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a)
            e1.record()
            self.ev[name].append((e0, e1))
            return r

        def fc_forward(self, *a):
            return self._timed("fc_forward", super().fc_forward, *a)

        def fc_input_grad(self, *a):
            return self._timed("fc_input_grad", super().fc_input_grad, *a)

        def fc_forward_split(self, *a):
            return self._timed("fc_forward", super().fc_forward_split, *a)

        def fc_input_grad_split(self, *a):                               # includes the small dZ re-scaling kernel
            return self._timed("fc_input_grad", super().fc_input_grad_split, *a)

        def conv_forward(self, *a):
            return self._timed("conv_forward", super().conv_forward, *a)

        def conv_forward_split(self, *a):
            return self._timed("conv_forward", super().conv_forward_split, *a)

        def conv_input_grad(self, *a):
            return self._timed("conv_input_grad", super().conv_input_grad, *a)

        def conv_input_grad_split(self, *a):
            return self._timed("conv_input_grad", super().conv_input_grad_split, *a)

    if args.shard == "samples":
        xs, ys, S_job, N_job = x, y, w["S"] * world, w["N"]
    else:
        g = torch.Generator().manual_seed(4321 + rank)                # weak scaling: every rank its own N points
        xs = torch.rand((w["N"],) + w["shape"], generator=g, dtype=torch.float32) if rank else x
        ys, S_job, N_job = y, w["S"], w["N"] * world
    xs = xs.to(device)
    labels = ys.to(device=device, dtype=torch.int32)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def run(precision):
        """warmup, then EXACTLY --steps timed steps between barrier + synchronize; returns (engine precision, seconds, kernel events)."""
        kern = TimedKernels()
        if args.shard == "samples":
            eng = make_engine(sp, kernels=kern, group=group, total_samples=w["S"] * world, precision=precision)
            eng._S_total = w["S"] * world
        else:
            eng = make_engine(sp, kernels=kern, precision=precision)

        def step():
            if w["method"] == "fgsm":
                return eng.fgsm(xs, labels, w["S"], w["eps"])
            return eng.pgd(xs, labels, w["S"], w["eps"], alpha=None, iters=w["iters"])

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        kern.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern.on = False
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return getattr(eng, "precision", "exact"), dt, kern.ev

    # per-launch algorithmic flops of each GEMM kernel: half of SURVEY 8(d)'s 4*(D*H + H*C) per attack-sample
    per_launch = 2.0 * (D * w["H"] + w["H"] * w["C"]) * w["N"] * w["S"]
    if w["arch"] == "fc2":        # F_fc2 = 4*(D*H + H^2 + H*C), half per direction
        per_launch = 2.0 * (D * w["H"] + w["H"] ** 2 + w["H"] * w["C"]) * w["N"] * w["S"]
    if w["arch"] == "conv":       # 2*(460800 + 26214400*H/512 + 49*H*C) flop per (point, sample) per direction
        per_launch = 2.0 * (460800 + 51200.0 * w["H"] + 49 * w["H"] * w["C"]) * w["N"] * w["S"]
    KNAMES = {"exact": {"fc_input_grad": "fc_grad_kernel", "fc_forward": "fc_forward_kernel",
                        "conv_forward": "conv2_pool_kernel (+ conv1_pool, conv_fc)", "conv_input_grad": "conv_bwd_kernel"},
              "split": {"fc_input_grad": "fc_grad_split_kernel (+ split_dz)", "fc_forward": "fc_forward_split_kernel",
                        "conv_forward": "conv2_pool_split_kernel (+ conv1_pool_split, conv_fc)", "conv_input_grad": "conv_bwd_split_kernel (+ conv_fc_bwd, conv1_bwd)"}}
    SPLIT_KERNELS = {"fc_forward", "fc_input_grad", "conv_forward", "conv_input_grad"}
    if w["arch"] == "fc2":
        KNAMES["split"].update({"fc_forward": "fc_forward_split_kernel (x2: layer 1 -> split image, layer 2)",
                                "fc_input_grad": "fc_grad_split_kernel (x2: per-sample step through Wm, then W1; + split_dz)"})
        KNAMES["exact"].update({"fc_forward": "fc_forward_kernel (x2)", "fc_input_grad": "fc_grad_kernel (x2)"})

    def roofline(mode, evs_by_name, ms_per_step):
        kernels = {}
        for name, evs in evs_by_name.items():
            ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
            kernels[name] = {"launches": len(evs), "avg_ms": ms, "tflops": per_launch / (ms * 1e-3) / 1e12 if ms else None}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and args.workload == "c2" and not args.points and not args.samples:
            traffic = json.load(open(pmc)).get(dom + ("_split" if mode == "split" and dom in SPLIT_KERNELS else ""), {}).get("hbm_bytes_per_launch")
        fp32_eq = kernels[dom]["tflops"]
        r = {"bound": "mfma", "kernel": KNAMES[mode][dom], "unit": "TFLOP/s", "traffic": traffic,
             "flop_per_launch": per_launch, "avg_launch_ms": kernels[dom]["avg_ms"], "kernels": kernels}
        if mode == "split" and dom in SPLIT_KERNELS:
            # matrix-pipe work of the split mode: 3 f16 products per algorithmic fp32 MAC (the dA generator's MFMAs are not counted)
            r.update({"achieved": 3.0 * fp32_eq, "peak": F16_MFMA_PEAK_TFLOPS, "frac": 3.0 * fp32_eq / F16_MFMA_PEAK_TFLOPS,
                      "pipe": "v_mfma_f32_16x16x32_f16, 3 products per fp32 MAC", "fp32_equivalent_tflops": fp32_eq,
                      "fp32_mfma_peak": FP32_MFMA_PEAK_TFLOPS, "vs_fp32_mfma_peak": fp32_eq / FP32_MFMA_PEAK_TFLOPS})
        else:
            r.update({"achieved": fp32_eq, "peak": FP32_MFMA_PEAK_TFLOPS, "frac": fp32_eq / FP32_MFMA_PEAK_TFLOPS,
                      "pipe": "v_mfma_f32_16x16x4_f32"})
        r["whole_step_tflops"] = 2 * per_launch / (1e-3 * ms_per_step / w["iters"]) / 1e12
        return r

    mode, dt, evs = run(args.precision)
    other = None
    if world == 1 and mode == "split" and args.precision == "auto":
        other = run("exact")                                              # reference line: the exact-fp32 kernels on the same workload

    import ctypes
    ctypes.CDLL(None).fflush(None)          # every rank: anything RCCL left in C stdio goes out before rank 0's JSON line
    barrier()
    if rank == 0:
        units = N_job * S_job * w["iters"] * args.steps
        ms_per_step = 1e3 * dt / args.steps
        out = {
            "metric": "attack-samples/sec (test_pts x posterior_samples x PGD_iters)",
            "value": units / dt, "unit": "attack-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 carried as f16 hi+lo pairs: 3 f16 MFMA products per fp32 product, f32 accumulate (2^-22 per product)" if mode == "split" else "f32",
            "precision_mode": mode, "data": "synthetic",
            "config": {"workload": w["desc"], "name": args.workload, "points": N_job, "samples_total": S_job,
                       "iters": w["iters"], "shard": args.shard if world > 1 else "none"},
            "roofline": roofline(mode, evs, ms_per_step),
        }
        if other is not None:
            o_ms = 1e3 * other[1] / args.steps
            out["exact_fp32_mode"] = {"value": units / other[1], "ms_per_step": o_ms, "roofline": roofline("exact", other[2], o_ms)}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, x, y, post, args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        # RCCL writes its version banner through C stdio (block-buffered when stdout is a file or pipe): flush it first so
        # that the JSON line is the LAST line of the output
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
