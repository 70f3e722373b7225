#!/usr/bin/env python3
"""gpurun_out/r2prof (written by scripts/profile_r2.sh on the GPU box) -> profiles/r2: the bench line, the rocprofv3 kernel
stats of the same command, and one summary per counter run (dominant kernel, counters summed over the launches of a pass,
per-ray figures)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r2prof"), os.path.join(ROOT, "profiles", "r2")
COMMON = ("; one rocprofv3 --pmc run per counter block (SQ x8 | GRBM + SQ x7 | FETCH_SIZE | WRITE_SIZE); counters are sums over all "
          "launches of a pass; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them")


def summarise(name, launches, rays, command):
    j = json.load(open(os.path.join(SRC, f"pmc_r2_{name}.json")))
    kernel = [k for k in j if "resolve" not in k][0]
    e, c = j[kernel], j[kernel]["counters"]
    out = {"command": command + COMMON, "kernel": kernel, "launches": launches, "rays_per_launch": rays, "counters": c,
           "kernel_seconds_per_pass": e["kernel_seconds_per_pass"], "lds_bytes": e["lds_bytes"], "scratch_bytes": e["scratch_bytes"],
           "vgprs": e["vgprs"], "workgroup": e["workgroup"],
           "derived": {"valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / launches / rays, "salu_wave_instr_per_ray": c["SQ_INSTS_SALU"] / launches / rays,
                       "lds_wave_instr_per_ray": c["SQ_INSTS_LDS"] / launches / rays, "wait_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                       "issue_stall_share_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                       "hbm_bytes_per_launch": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / launches,
                       "hbm_write_bytes_per_launch": c["WRITE_SIZE"] * 1024 / launches}}
    with open(os.path.join(DST, f"{name}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(name, kernel[:60], {k: round(v, 4) for k, v in out["derived"].items()}, "vgprs", e["vgprs"], "scratch", e["scratch_bytes"])


def rays_of(log):
    for line in open(os.path.join(ROOT, "gpurun_out", log)):
        if "stats pass:" in line:
            return int(line.split("stats pass:")[1].split("rays")[0])
    raise SystemExit(f"no stats pass line in {log}")


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, "a_bench_line.json"))
    shutil.copy(os.path.join(SRC, "kernel_stats.csv"), os.path.join(DST, "a_bench_kernel_stats.csv"))
    summarise("n512", 2, rays_of("pmc_r2_n512_1.log"), "scripts/pmc_run.sh r2_n512 -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs   (bench.py defaults: 512 spheres, 1920x1080 AA=4)")
    summarise("c3", 3, rays_of("pmc_r2_c3_1.log"), "scripts/pmc_run.sh r2_c3 -- --steps 2 --warmup 1 ... --scene demo_path --width 1280 --height 960 --aa 2   (C3, path 10/3/5)")
    summarise("c4", 3, rays_of("pmc_r2_c4_1.log"), "scripts/pmc_run.sh r2_c4 -- --steps 2 --warmup 1 ... --scene csg_example   (C4: 1920x1080 AA=4, path 10/3/5)")
    summarise("n4096", 2, rays_of("pmc_r2_n4096_1.log"), "scripts/pmc_run.sh r2_n4096 -- --steps 1 --warmup 1 ... --spheres 4096 --height 68   (the 4096-sphere scene, 1920x68 AA=4: the sample count of a 1/16 slice of the 1080p frame)")
    b = json.load(open(os.path.join(DST, "a_bench_line.json")))
    print("bench:", round(b["value"], 1), "Mrays/s,", round(b["ms_per_step"], 1), "ms, frac", round(b["roofline"]["frac"], 4), b["roofline"]["kernel"])
    for c in b["configs"]:
        print("  ", c["id"], round(c["kernel_ms"], 3), "ms", round(c["Mrays_per_s"]), "Mrays/s frac", round(c["roofline"]["frac"], 3), c["kernel"])
    print(open(os.path.join(DST, "a_bench_kernel_stats.csv")).read()[:1500])


if __name__ == "__main__":
    main()
