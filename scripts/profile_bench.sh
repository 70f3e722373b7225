#!/bin/bash
# rocprofv3 kernel stats + counter passes of the default bench command.   scripts/profile_bench.sh TAG [bench args]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 "$@" > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
cat gpurun_out/bench_${tag}.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_${tag}.log 2>&1
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" -exec cat {} \;
scripts/pmc_run.sh ${tag} -- --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}.out 2>&1
tail -45 gpurun_out/pmc_${tag}.out
