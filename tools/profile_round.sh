#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command WITH the fp16-mode legs, then the PMC passes (separate
# runs, --pmc with --kernel-trace only), into gpurun_out/prof_$ROUND.  Copy what is to be judged into profiles/.
set -e
ROUND=${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd $R
if [ -z "$SKIP_STATS" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-config5 --no-phases --no-ref-style --no-configs --no-check > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $S $OUT/${ROUND}_kernel_stats.csv
python3 tools/rocprof_summarize.py $OUT/${ROUND}_kernel_stats.csv $OUT/${ROUND}_rocprof_summary.json
fi
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_probe.py > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 tools/pmc_probe.py > $OUT/pmc_mfma.log 2>&1
echo "mfma done"
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
cp $F $OUT/${ROUND}_pmc_fetch_size.csv; cp $W $OUT/${ROUND}_pmc_write_size.csv; cp $M $OUT/${ROUND}_pmc_mfma_busy.csv
python3 tools/pmc_summarize.py $F $W $M $OUT/pmc_fetch.log $OUT/${ROUND}_pmc_summary.json > $OUT/pmc_summary.log
head -14 $OUT/${ROUND}_kernel_stats.csv; tail -c 1200 $OUT/bench_profiled.json
