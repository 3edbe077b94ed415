#!/bin/bash
# Run on the GPU box (via gpurun): tests, bench line, rocprofv3 kernel-trace stats, PMC passes.
# usage: tools/profile_bench.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
# kernel trace + stats of the same command (CPU baseline leg skipped: it launches no kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/trace.log 2>&1
ls -R $OUT/trace | head -20
# PMC passes (separate runs; no trace domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/pmc_lds.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
