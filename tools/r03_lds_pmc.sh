#!/bin/bash
# usage (GPU box): tools/r03_lds_pmc.sh <tag> <workload> [bench args...]    LDS bank-conflict PMC pass of one bench workload -> gpurun_out/<tag>/lds_<workload>.txt
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds_$WL -o run -- python3 $R/bench.py --workload $WL --cpu-seconds 0 --no-other-mode "$@" > $OUT/pmc_lds_$WL.log 2>&1
python3 - <<PY > $OUT/lds_$WL.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_lds_$WL/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0][-60:]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE",[0]))):
    m={c:sum(v)/len(v) for c,v in d.items()}
    if m.get("SQ_LDS_IDX_ACTIVE",0) < 1e6: continue
    cyc=m.get("GRBM_GUI_ACTIVE",0)/8
    print(f"{n:62s} launches={len(d['GRBM_GUI_ACTIVE']):4d} wall_cycles={cyc:.4g} LDS_IDX_ACTIVE={m['SQ_LDS_IDX_ACTIVE']:.4g} BANK_CONFLICT={m['SQ_LDS_BANK_CONFLICT']:.4g} conflict/active={m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']:.3f} lds_active_frac_of_wall={m['SQ_LDS_IDX_ACTIVE']/(256*cyc+1e-9):.3f}")
PY
rm -rf $OUT/pmc_lds_$WL
cat $OUT/lds_$WL.txt
