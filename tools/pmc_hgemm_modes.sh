#!/bin/bash
# wave-level PMC passes over tools/pmc_hgemm_modes.py; into gpurun_out/pmc_hg_$TAG
set -e
TAG=${TAG:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_hg_$TAG; mkdir -p $OUT; cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/pmc_hgemm_modes.py > $OUT/p1.log 2>&1
echo p1 done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/pmc_hgemm_modes.py > $OUT/p2.log 2>&1
echo p2 done
for p in p1 p2; do F=$(find $OUT/$p -name "*counter_collection.csv" | head -1); python3 tools/pmc_modes_table.py $F $OUT/$p.log $OUT/${TAG}_hgemm_modes_$p.json > $OUT/$p.table; cat $OUT/$p.table; done
