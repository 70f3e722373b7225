#!/bin/bash
# Round-2 evidence, one GPU box call:  bench line (all configs), rocprofv3 kernel stats of the same command, counter passes.
#   scripts/profile_r2.sh          -> gpurun_out/r2prof/*
set -e
out=gpurun_out/r2prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "[1] bench line"; timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "[2] kernel stats of the same command"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
echo "[3] counters: N=512 frame (bench default, no configs)"
scripts/pmc_run.sh r2_n512 -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $out/pmc_n512.out 2>&1
echo "[4] counters: C3, C4, N=4096 (68 rows of the 1080p AA=4 frame)"
scripts/pmc_run.sh r2_c3 -- --steps 2 --warmup 1 --no-cpu-baseline --no-configs --scene demo_path --width 1280 --height 960 --aa 2 > $out/pmc_c3.out 2>&1
scripts/pmc_run.sh r2_c4 -- --steps 2 --warmup 1 --no-cpu-baseline --no-configs --scene csg_example > $out/pmc_c4.out 2>&1
scripts/pmc_run.sh r2_n4096 -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs --spheres 4096 --height 68 > $out/pmc_n4096.out 2>&1
cp gpurun_out/pmc_r2_*.json $out/
echo done
