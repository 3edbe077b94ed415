#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of tools/profile_round.sh into a short text summary + the per-launch HBM traffic record.

usage: summarize_profile.py <dir> [workload]
<dir> holds trace/ (--kernel-trace --stats) and pmc_fetch/, pmc_write/, pmc_sq/, pmc_lds/ (one --pmc pass each; FETCH_SIZE and
WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots").  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE
(KB -> bytes): on gfx950 FETCH_SIZE tallies the 128-B requests of wide streaming loads at 64 B (same guide, HBM section).
Writes <dir>/pmc_traffic.json = {bench.py kernel key: {...}} for `workload`."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
points = int(sys.argv[3]) if len(sys.argv) > 3 else None          # N and S per GPU of the profiled command (bench.py scales a conv record
samples = int(sys.argv[4]) if len(sys.argv) > 4 else None         # taken at another size by points x samples and says so)

# kernel-name fragment -> bench.py's roofline key (a key sums the kernels one C-ABI call launches).  The fp32 kernels that EVERY precision
# mode of the conv path launches (conv1, the head, their transposes) get their own "_common" keys: a profile run that times several modes
# would otherwise book them under the fp32 mode's key only.
MAIN = [("fc_forward_x3_kernel", "fc_forward_triple"), ("fc_grad_x3_kernel", "fc_input_grad_triple"), ("triple_dz_kernel", "fc_input_grad_triple"),
        ("fc_forward_split_kernel", "fc_forward_split"), ("fc_grad_split_kernel", "fc_input_grad_split"), ("split_dz_kernel", "fc_input_grad_split"),
        ("fc_forward_kernel", "fc_forward"), ("fc_grad_kernel", "fc_input_grad"),
        ("conv2_pool_x3_kernel", "conv_forward_triple"), ("conv_bwd_dense_x3_kernel", "conv_input_grad_triple"),
        ("conv1_bwd_x3_kernel", "conv_input_grad_triple"),
        ("conv2_pool_split_kernel", "conv_forward_split"), ("conv1_pool_split_kernel", "conv_forward_split"),
        ("conv_bwd_split_kernel", "conv_input_grad_split"),
        ("conv2_pool_kernel", "conv_forward"), ("conv1_pool_kernel", "conv_forward_common"), ("conv_fc_kernel", "conv_forward_common"),
        ("conv_bwd_kernel", "conv_input_grad"), ("conv_fc_bwd_kernel", "conv_input_grad_common"), ("conv1_bwd_mfma_kernel", "conv_input_grad_common"),
        ("lowdim_kernel", "lowdim"), ("lowdim2_kernel", "lowdim")]
# the streaming kernels of a step (same names in every mode; one launch each per pass, the draw once per step): their counters go into the
# record's "small" table, which bench.py adds to a pass's counter total
SMALL = ["svi_draw_flat_kernel", "svi_draw_kernel", "conv_k2_images_kernel", "triple_rows_kernel", "split_rows_kernel", "absmax_kernel", "scale_finalize_kernel", "step_tail_x3_kernel",
         "reduce_samples", "loss_dlogits", "sum_slabs_norms", "sum_slabs", "attack_step", "pgd_alpha"]
SHORT = ["conv_bwd_dense_x3_kernel"] + [k for k, _ in MAIN] + ["svi_draw_flat_kernel", "svi_draw_kernel", "conv_k2_images_kernel", "lowdim_kernel", "step_tail_x3_kernel", "attack_step_x3_kernel", "triple_rows_kernel", "split_rows_kernel", "absmax_kernel", "scale_finalize_kernel", "reduce_samples", "loss_dlogits", "sum_slabs_norms",
                                "sum_slabs", "attack_step", "pgd_alpha", "eval_metrics"]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


def short(name):
    for k in SHORT:
        if k in name:
            geo = "<3,32>" if "Geo<3, 32>" in name else ""
            return k + geo
    return name[:60]


def key_of(name):
    for k, v in MAIN:
        if k in name:
            return v
    return None


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in sorted(rows("trace/**/*kernel_stats.csv"), key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:16]:
    print(f"{short(r['Name']):32s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:10.3f} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}")

print("\n== kernel trace: per-kernel duration + resources ==")
agg = collections.defaultdict(list)
res = {}
for r in rows("trace/**/*kernel_trace.csv"):
    k = short(r["Kernel_Name"])
    agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:20]:
    v2 = sorted(v)
    print(f"{k:32s} n={len(v):4d} avg_us={sum(v)/len(v):10.2f} med_us={v2[len(v)//2]:10.2f} min_us={v2[0]:10.2f}  vgpr/agpr/sgpr/lds/scratch/grid/wg={res[k]}")


def pmc(dirname, title):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(f"{dirname}/**/*counter_collection.csv"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"\n== PMC: {title} (mean per launch) ==")
    for k, d in acc.items():
        if key_of(k) is None and not any(n in k for n in SMALL):
            continue
        print(f"{k:32s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
    return acc


f = pmc("pmc_fetch", "FETCH_SIZE (KB; gfx950 reads x2 for wide streaming loads)")
w = pmc("pmc_write", "WRITE_SIZE (KB)")
sq = pmc("pmc_sq", "SQ")
pmc("pmc_lds", "LDS / clock")
for k, d in sq.items():                                     # matrix-pipe utilisation: busy cycles vs the kernel's wall cycles
    if key_of(k) and "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        print(f"   {k}: SQ_VALU_MFMA_BUSY_CYCLES per SIMD = {busy / 1024:.4g} (1024 SIMDs)")
tr = collections.defaultdict(lambda: {"fetch_size_kb_raw": 0.0, "write_size_kb": 0.0, "kernels": []})
small = {}
for k in set(f) & set(w):
    key = key_of(k)
    fs = sum(f[k]["FETCH_SIZE"]) / len(f[k]["FETCH_SIZE"])
    wsz = sum(w[k]["WRITE_SIZE"]) / len(w[k]["WRITE_SIZE"])
    if key is None:
        if any(n in k for n in SMALL):
            small[k] = {"fetch_size_kb_raw": fs, "write_size_kb": wsz, "hbm_bytes_per_launch": (2 * fs + wsz) * 1024}
        continue
    tr[key]["fetch_size_kb_raw"] += fs
    tr[key]["write_size_kb"] += wsz
    tr[key]["kernels"].append(k)
for key, d in tr.items():
    d["hbm_bytes_per_launch"] = (2 * d["fetch_size_kb_raw"] + d["write_size_kb"]) * 1024
    d["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"
if tr:
    print("\n== traffic json ==")
    print(json.dumps({"kernels": tr, "small": small}))
    rel = "profiles/" + out.split("gpurun_out/")[-1] if "gpurun_out/" in out else out      # the copy that is committed, not the scratch path
    json.dump({workload: {"points": points, "samples": samples, "kernels": tr, "small": small},
               "source": f"{rel}/summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, {workload} workload"
                         + (f", N={points}, S={samples}" if points else "") + ")"},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
