#!/bin/bash
# Round-3 evidence in two GPU box calls (the bench line reads the counter summaries of the first for roofline.traffic / valu_issue):
#   scripts/profile_r3.sh pmc     counter passes of the bench default (one share of the N=4096 frame per launch), the same with --scan-all-tiles, C3, C4, the N=512 frame
#                                 -> gpurun_out/r3prof/pmc_*          then here: python scripts/profile_r3_collect.py pmc   -> profiles/r3/*_pmc.json
#   scripts/profile_r3.sh bench   the bench line as the driver runs it (--steps 20 --warmup 5) and rocprofv3 kernel stats of the default command
#                                 -> gpurun_out/r3prof/bench.json ...  then here: python scripts/profile_r3_collect.py bench -> profiles/r3/a_*
set -e
out=gpurun_out/r3prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$1" = pmc ]; then
  echo "[1] counters: the bench default (one share of the N=4096 frame per launch)"
  scripts/pmc_run.sh r3_n4096_share -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $out/pmc_n4096_share.out 2>&1
  scripts/pmc_run.sh r3_n4096_share_scan_all -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs --scan-all-tiles > $out/pmc_n4096_share_scan_all.out 2>&1
  echo "[2] counters: C3, C4, the N=512 frame"
  scripts/pmc_run.sh r3_c3 -- --steps 2 --warmup 1 --no-cpu-baseline --no-configs --scene demo_path --width 1280 --height 960 --aa 2 --shares 1 > $out/pmc_c3.out 2>&1
  scripts/pmc_run.sh r3_c4 -- --steps 2 --warmup 1 --no-cpu-baseline --no-configs --scene csg_example --shares 1 > $out/pmc_c4.out 2>&1
  scripts/pmc_run.sh r3_n512 -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs --spheres 512 --shares 1 > $out/pmc_n512.out 2>&1
  cp gpurun_out/pmc_r3_*.json $out/
  for t in n4096_share n4096_share_scan_all c3 c4 n512; do cp gpurun_out/pmc_r3_${t}_1.log $out/pmc_${t}_pass1.log; done
else
  echo "[1] bench line, the driver's command"; timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
  echo "[2] kernel stats of the default command"
  timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
  find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
fi
echo done
