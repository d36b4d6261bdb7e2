#!/bin/bash
# A/B of the four dgemm variants: un-instrumented throughput, then the stamped build with the in-kernel clock
for mf in 0 1 2 3; do echo "== MPF_GEMM_MF=$mf"; MPF_GEMM_MF=$mf python tools/perf_probe.py gemm 2>&1 | grep dgemm; MPF_GEMM_MF=$mf python tools/gemm_stamp_probe.py 2>&1 | grep "m="; done
