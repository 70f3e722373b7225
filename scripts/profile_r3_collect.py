#!/usr/bin/env python3
"""gpurun_out/r3prof (written by scripts/profile_r3.sh on the GPU box) -> profiles/r3: the bench line, the rocprofv3 kernel
stats of the same command, and one summary per counter run (dominant kernel, counters summed over the launches of a pass,
per-ray figures).  bench.py reads these summaries for `roofline.traffic` and `roofline.valu_issue` and names them as the source."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r3prof"), os.path.join(ROOT, "profiles", "r3")
COMMON = ("; one rocprofv3 --pmc run per counter block (SQ x8 | GRBM + SQ x7 | FETCH_SIZE | WRITE_SIZE | VALU instruction classes | activity and "
          "LDS atomics | matrix pipe); counters are sums over all launches of a pass; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; derived.hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: the gfx950 correction of MI355X_MICROARCH.md")


def rays_of(log):
    """(launches, total rays over the launches) of one bench.py pass, from its stderr lines."""
    rays = []
    for line in open(log):
        m = re.search(r"first pass: (\d+) rays", line) or re.search(r"step \d+/\d+ done: (\d+) rays", line)
        if m:
            rays.append(int(m.group(1)))
    if not rays:
        raise SystemExit(f"no ray counts in {log}")
    return len(rays), sum(rays)


def summarise(name, command):
    j = json.load(open(os.path.join(SRC, f"pmc_r3_{name}.json")))
    kernel = [k for k in j if "resolve" not in k][0]
    e, c = j[kernel], j[kernel]["counters"]
    launches, total_rays = rays_of(os.path.join(SRC, f"pmc_{name}_pass1.log"))
    secs = list(e["kernel_seconds_per_pass"].values())[1]  # the pass that carries GRBM_GUI_ACTIVE
    clock = c["GRBM_GUI_ACTIVE"] / 8 / secs  # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
    simd_cycles = secs * clock * 1024
    d = {"clock_ghz": clock / 1e9,
         "valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / total_rays, "salu_wave_instr_per_ray": c["SQ_INSTS_SALU"] / total_rays,
         "lds_wave_instr_per_ray": c["SQ_INSTS_LDS"] / total_rays, "simd_cycles_per_ray": simd_cycles / total_rays,
         "wait_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
         "issue_stall_share_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
         "lds_bank_conflict_ratio": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
         "lds_busy_share_per_cu": c["SQ_LDS_IDX_ACTIVE"] / 256 / (secs * clock),
         "waves_resident_per_simd": c["SQ_WAVE_CYCLES"] * 4 / simd_cycles,
         "active_lanes_per_valu_instr": c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_INSTS_VALU"], 1.0),
         "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / launches,  # FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 (MI355X_MICROARCH.md)
         "hbm_write_bytes_per_launch": c["WRITE_SIZE"] * 1024 / launches}
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16), summed over the SIMDs
        secs7 = list(e["kernel_seconds_per_pass"].values())[-1]
        d["mfma_busy_share_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (secs7 * clock * 1024)
        d["mfma_instr_per_ray"] = (c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_F32", 0.0)) / total_rays
        d["mfma_busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_F32", 0.0), 1.0)
    out = {"command": command + COMMON, "kernel": kernel, "launches": launches, "rays_per_launch": total_rays / launches, "counters": c,
           "kernel_seconds_per_pass": e["kernel_seconds_per_pass"], "lds_bytes": e["lds_bytes"], "scratch_bytes": e["scratch_bytes"],
           "vgprs": e["vgprs"], "workgroup": e["workgroup"], "derived": d}
    with open(os.path.join(DST, f"{name}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(name, kernel[:60], {k: round(v, 4) for k, v in d.items()}, "vgprs", e["vgprs"], "scratch", e["scratch_bytes"])


def main():
    import sys
    os.makedirs(DST, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, "a_bench_line.json"))
        shutil.copy(os.path.join(SRC, "stats_bench.json"), os.path.join(DST, "a_bench_line_profiled_run.json"))
        shutil.copy(os.path.join(SRC, "kernel_stats.csv"), os.path.join(DST, "a_bench_kernel_stats.csv"))
        b = json.load(open(os.path.join(DST, "a_bench_line.json")))
        print("bench:", round(b["value"], 1), "Mrays/s,", round(b["ms_per_step"], 1), "ms, frac", round(b["roofline"]["frac"], 4), b["roofline"]["kernel"],
              "traffic", b["roofline"]["traffic"])
        for c in b["configs"]:
            print("  ", c["id"], round(c["kernel_ms"], 3), "ms", round(c["Mrays_per_s"]), "Mrays/s frac", round(c["roofline"]["frac"], 3), c["kernel"],
                  (c["roofline"].get("valu_issue") or {}).get("frac"))
        print(open(os.path.join(DST, "a_bench_kernel_stats.csv")).read()[:1500])
        return
    summarise("n4096_share", "scripts/pmc_run.sh r3_n4096_share -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs   (bench.py defaults: 4096 spheres, 1920x1080 AA=4, shares 0 and 1 of 16)")
    summarise("n4096_share_scan_all", "scripts/pmc_run.sh r3_n4096_share_scan_all -- --steps 1 --warmup 1 --no-cpu-baseline --no-configs --scan-all-tiles   (the same shares, every ray against every tile: k_render_path_tile)")
    summarise("c3", "scripts/pmc_run.sh r3_c3 -- --steps 2 --warmup 1 ... --scene demo_path --width 1280 --height 960 --aa 2 --shares 1   (C3, path 10/3/5)")
    summarise("c4", "scripts/pmc_run.sh r3_c4 -- --steps 2 --warmup 1 ... --scene csg_example --shares 1   (C4: 1920x1080 AA=4, path 10/3/5)")
    summarise("n512", "scripts/pmc_run.sh r3_n512 -- --steps 1 --warmup 1 ... --spheres 512 --shares 1   (the whole N=512 1920x1080 AA=4 frame per launch)")


if __name__ == "__main__":
    main()
