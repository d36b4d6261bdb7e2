#!/bin/bash
# per-kernel FETCH_SIZE / WRITE_SIZE of whole factorizations; into gpurun_out/pmc_fac_$TAG
set -e
TAG=${TAG:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fac_$TAG; mkdir -p $OUT; cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/pmc_factor_probe.py > $OUT/fetch.log 2>&1
echo fetch done; tail -3 $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 tools/pmc_factor_probe.py > $OUT/write.log 2>&1
echo write done
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); FT=$(find $OUT/fetch -name "*kernel_trace.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_factor_summarize.py $F $FT $W $OUT/fetch.log $OUT/${TAG}_pmc_nongemm_summary.json
# keep the merged artefacts small: the raw CSVs stay on the box, the summary and a per-kernel digest come back
rm -rf $OUT/fetch $OUT/write
