#!/bin/bash
# usage (GPU box): tools/r03_pmc.sh <tag> <workload> [bench args...]   two PMC passes (matrix pipe / wave states, LDS) of one bench workload, per kernel
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$WL -o run -- python3 $R/bench.py --workload $WL --cpu-seconds 0 --no-other-mode "$@" > $OUT/pmc_sq_$WL.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds_$WL -o run -- python3 $R/bench.py --workload $WL --cpu-seconds 0 --no-other-mode "$@" > $OUT/pmc_lds_$WL.log 2>&1
python3 - <<PY > $OUT/pmc_$WL.txt
import csv,glob,collections,re
for sub in ("sq","lds"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_%s_$WL/**/*counter_collection.csv" % sub,recursive=True):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(\w+_kernel)", r["Kernel_Name"])
            n=(m.group(1) if m else r["Kernel_Name"][:40]) + ("<3,32>" if "Geo<3, 32>" in r["Kernel_Name"] else "")
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n,d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE",[0]))):
        m={c:sum(v)/len(v) for c,v in d.items()}
        cyc=m.get("GRBM_GUI_ACTIVE",0)/8
        if cyc < 2e5: continue
        extra=""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: extra=f"mfma_busy_frac={m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc+1e-9):.3f} wait_frac_of_wave_cycles={m['SQ_WAIT_INST_ANY']/(m['SQ_WAVE_CYCLES']+1e-9):.3f} valu_insts={m.get('SQ_INSTS_VALU',0):.4g}"
        if "SQ_LDS_IDX_ACTIVE" in m: extra=f"lds_active_frac={m['SQ_LDS_IDX_ACTIVE']/(256*cyc+1e-9):.3f} conflict/active={m.get('SQ_LDS_BANK_CONFLICT',0)/(m['SQ_LDS_IDX_ACTIVE']+1e-9):.3f} lds_insts={m.get('SQ_INSTS_LDS',0):.4g}"
        print(f"{sub:3s} {n:36s} launches={len(d['GRBM_GUI_ACTIVE']):3d} wall_cycles={cyc:.4g} {extra}")
PY
rm -rf $OUT/pmc_sq_$WL $OUT/pmc_lds_$WL
cat $OUT/pmc_$WL.txt
