#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command, then the two PMC passes (separate runs), into gpurun_out/prof_r02.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-mxp --no-config5 --no-phases > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_probe.py > $OUT/pmc_write.log 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summarize.py $F $W $OUT/r02_pmc_summary.json > $OUT/pmc_summary.log
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $S $OUT/r02_kernel_stats.csv
head -12 $OUT/r02_kernel_stats.csv; tail -c 1500 $OUT/bench_profiled.json
