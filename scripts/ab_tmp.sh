run() { local name=$1 w=$2; shift 2
  RT_TILE_W=$w RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so timeout -k 10 300 python3 scripts/run_configs.py --skip-samples "$@" 2>gpurun_out/r3/err_$name.txt | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('$name W=$w', j['config'], j['size'], j['kernel'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'])" || exit 1
  grep "rt phases" gpurun_out/r3/err_$name.txt | head -2
}
run bf_s8_ph 1 --synth 512 1920 1080 4 && run bf_s8_ph 4 --synth 4096 480 136 4 &&
run bf_s8_o2 1 --synth 512 1920 1080 4 && run bf_s8_o2 4 --synth 4096 480 136 4 && run bf_s8_o2 2 --synth 4096 480 136 4 &&
run bf_s16_o2 1 --synth 512 1920 1080 4 && run bf_s16_o2 4 --synth 4096 480 136 4 && run bf_s16_o2 2 --synth 4096 480 136 4
