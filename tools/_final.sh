set -e
timeout -k 10 900 bash tools/pmc_factor.sh > gpurun_out/r04_pmc_factor.log 2>&1 || true
tail -5 gpurun_out/r04_pmc_factor.log
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -c 600 gpurun_out/r04_bench_final.json
