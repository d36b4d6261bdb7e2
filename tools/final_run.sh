set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v13b -- python bench.py --steps 3 --warmup 1 --no-cpu --no-mxp --no-config5 --no-ir --no-phases > gpurun_out/bench_v13_prof.json 2> gpurun_out/bench_v13_prof.err
