#!/bin/bash
# usage: tools/run_ablate.sh "0 1 2 3 4 7 8 15 ..."   (run on the GPU box; builds one exe per bit set)
mkdir -p gpurun_out/abl
for b in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD $EXTRA -DRBNN_ABL=$b -o /tmp/ablate_$b tools/ablate.hip 2>/dev/null &
done
wait
for b in $1; do /tmp/ablate_$b; done 2>&1 | tee gpurun_out/abl/ablate.log
if [ -n "$2" ]; then
  export TMPDIR=/tmp
  for b in $2; do
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/abl/pmc_$b -o run -- /tmp/ablate_$b > /dev/null 2>&1
    python3 - <<EOF
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/abl/pmc_$b/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k="fwd" if "fc_forward" in r["Kernel_Name"] else "grad" if "fc_grad" in r["Kernel_Name"] else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    cyc=m.get("GRBM_GUI_ACTIVE",0)/8
    print("ABL=$b",k," ".join(f"{c}={v:.4g}" for c,v in sorted(m.items())), f"| mfma_busy_frac={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc+1e-9):.3f} waves/SIMD(avg)={m.get('SQ_WAVE_CYCLES',0)*4/(1024*cyc+1e-9):.2f}")
EOF
  done
fi
