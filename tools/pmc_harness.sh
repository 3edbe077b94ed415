#!/bin/bash
# usage: tools/pmc_harness.sh <exe> <tag>     PMC passes (matrix-pipe busy, clock, LDS conflicts) of a tools/ablate_*.hip harness (GPU box)
EXE=$1; TAG=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/abl/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o run -- $EXE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -o run -- $EXE > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for sub in ("sq","lds"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub,recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            k="fwd" if "fc_forward" in n else "grad" if "fc_grad" in n else None
            if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        m={c:sum(v)/len(v) for c,v in d.items()}
        cyc=m.get("GRBM_GUI_ACTIVE",0)/8
        extra=""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: extra=f"| wall_cycles={cyc:.4g} mfma_busy_frac={m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc+1e-9):.3f}"
        if "SQ_LDS_IDX_ACTIVE" in m: extra=f"| wall_cycles={cyc:.4g} lds_active_frac={m['SQ_LDS_IDX_ACTIVE']/(256*cyc+1e-9):.3f} conflict_frac_of_active={m.get('SQ_LDS_BANK_CONFLICT',0)/(m['SQ_LDS_IDX_ACTIVE']+1e-9):.3f}"
        print("$TAG",sub,k," ".join(f"{c}={v:.4g}" for c,v in sorted(m.items())),extra)
PY
rm -rf $OUT/sq $OUT/lds
