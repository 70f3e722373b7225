#!/usr/bin/env python3
"""bench.py — path-mode throughput of the MI355X tracing backend.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (the headline): the synthetic 4096-random-sphere + floor + sky scene (raytracer_amd/synth.py), --mode path,
1920x1080, AA=4 (16 spp), n_rays=10, roulette=3, max-depth=5.  Every object is tested for every ray (World::ray_intersection is a
linear scan, reference include/shapes.hpp:547-561); a ray = one such closest-hit query.  A whole frame is 4.04e11 rays, ~180 s
on one MI355X, so a STEP is one row-cyclic SHARE of it: 1/16 of the rows (rows k, k + 16, k + 32, ...: every share is the same
mix of sky, floor and sphere rows; shares differ by < 2 % in cost), 2.5e10 rays, ~11 s.  Step j renders share
(j * n_gpus + rank) mod 16: consecutive steps walk through the frame, a run of 16 / n_gpus steps renders all of it, and
`value` (rays / time) is the whole-frame rate up to the end of a launch, where the pools drain at low occupancy (~0.3 s per
launch: 1/32 shares measure 5 % less); `frame_ms_estimate` = 16 x ms_per_step / n_gpus.

With --gpus N every rank renders its own share of the step (per-GPU work is fixed as N grows: "scaling": "weak"; the job is
still one frame, N shares of it at a time) into HBM and rank 0 gathers the N shares over RCCL into the frame buffer.  With no
WORLD_SIZE in the environment and N > 1 the script starts its N ranks itself (a child `python -m torch.distributed.run`, decided
before anything touches the GPU) and relays rank 0's line.  Scene and buffers are resident in HBM before the timed region.

At N = 1 the line also carries `configs`: BASELINE.json's configs measured in the same run — C1 (onoff), C2 (flat and
pointlight), C3, C4 (path on the reference's sample scenes, full size), the full N=512 frame (round 2's default) with and
without the matrix-core filter, and the headline's share 0 without it — each with its kernel, ray count, kernel ms (HIP events
of the same launch) and roofline block.
configs[4] (3840x2160) is four times the headline frame: 64 shares of this cost.

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X fp32 peak, vector and matrix alike (v_mfma_f32_32x32x2_f32 runs on the VALU's fp32 lanes): 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
# (MI355X_MICROARCH.md: 157.3 TF spec for both, 155 measured on the MFMA)
FP32_VALU_PEAK_TFLOPS = 157.3
# dense bf16 MFMA peak: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz (v_mfma_f32_32x32x16_bf16 = 32768 flops in 32 cycles; the guide's "~2.5 PF dense")
BF16_MFMA_PEAK_TFLOPS = 256 * 4 * 1024 * 2.4e9 / 1e12

# fp32 flops one object test costs (FMA = 2, everything else 1), by the arithmetic form a kernel uses:
#   world-space sphere filter (wide ISO register rows, rt_wide.h): b = O.u - C.u (3 fma = 6), c = (O.O + k) - 2 O.C (1 add + 3 fma = 7),
#     disc = b^2 - c (1 fma = 2): 15 flops in 8 instructions
#   object-space sphere, FMA form (sphere_scan_test<true>): D' = M D (15), O' = M O + t (18), a (5), b (5), c (6), disc (3)
#   object-space sphere as the reference writes it (and the exact kernels execute it): 33 + 19
#   plane, FMA form: z row only (6 + 5) + t (2); as written: 33 + 2
#   the same filter on the matrix cores (rt_tile.h): four v_mfma_f32_32x32x2_f32 per 32 rays x 32 spheres = 4 x 4096 / 1024 = 16 flops per
#     test (K = 4 for b and for c: the constant terms 1 * O.u and k * 1 are multiply-adds too) + the VALU's d = b^2 - c (2): 18
F_SPHERE_TILE = 18.0
#   round 3, the same two products on the bf16 matrix cores (rt_tile.h, kernel name "...,BF16x3>"): every binary32 operand split into
#     three bf16 parts, six partial products per term -> 24 of the 32 K slots of two v_mfma_f32_32x32x16_bf16 per product: four MFMAs
#     per 32 x 32 tile = 4 x 32768 / 1024 = 128 bf16-MFMA flops per test (96 of them in slots that hold data), priced against the
#     bf16 dense peak; the VALU still does d = b^2 - c (2 fp32 flops per test) and everything else of the ray
F_SPHERE_TILE_BF16_MFMA = 128.0
F_SPHERE_WORLD, F_SPHERE_OBJECT, F_SPHERE_WRITTEN = 15.0, 52.0, 52.0
F_PLANE_FMA, F_PLANE_WRITTEN = 13.0, 35.0
F_EPILOGUE = 54.0 + 75.0  # the winner's hit record + scatter / shading, same in both counts
F_EXACT_RECORD = 59.0 + 59.0  # large-world kernels: the winner and the sphere the ray leaves, each once more as the reference writes it


# CSG trees by SURVEY.md 8d's formula: every ray pays each leaf's test (52 per sphere leaf, 35 per plane leaf as the reference
# writes them; the truth-table evaluator of the fused kernels: 58 per sphere leaf — transform 33 + a, b, c, disc 19 + both roots 6 —
# and 13 per plane leaf); a ray that meets the tree also pays, per candidate hit, one inside test per sibling leaf (23) and the
# re-mapping of point and normal per node (33).  The leaf part is counted; the per-hit part (at most +30 % on samples/csg_example.txt)
# is data dependent and is NOT counted: the CSG fractions are lower bounds by that much.
F_CSG_SPHERE_LEAF_TT, F_CSG_PLANE_LEAF_TT = 58.0, 13.0

# Issue cost of one wave64 VALU instruction on gfx950, measured (scripts/microbench/issue_bench.hip, profiles/r3/issue_bench.txt):
# fp32 add / mul / fma with vector-register operands 2.35 cycles, transcendental 8.2, everything else (compares, selects, min / max,
# conversions, shifts, integer multiplies, DPP, any VOP3-only form, any form with a scalar-register operand) 4.2.
VALU_CYCLES_FAST, VALU_CYCLES_TRANS, VALU_CYCLES_OTHER = 2.35, 8.2, 4.2
SIMDS = 256 * 4


def scene_census(scene):
    import numpy as np
    top = scene.objects[scene.objects >= 0]
    n_sph = int(np.sum(scene.shape_kind[top] == 0))
    leaves = []
    if len(scene.csg_op):
        in_tree = np.ones(len(scene.shape_kind), dtype=bool)
        in_tree[top] = False
        leaves = scene.shape_kind[in_tree]
    return {"spheres": n_sph, "planes": int(len(top)) - n_sph, "csg": int(np.sum(scene.objects < 0)),
            "csg_sphere_leaves": int(np.sum(np.asarray(leaves) == 0)), "csg_plane_leaves": int(np.sum(np.asarray(leaves) == 1)),
            "csg_nodes": int(len(scene.csg_op))}


def flops_per_ray(census, kernel: str):
    """-> (executed, algorithmic) fp32 flops of one closest-hit query + shading.  `algorithmic` is the reference's
    arithmetic as written (52 per sphere test, +7 for the one sphere that is hit, 35 per plane; CSG leaves alike); `executed`
    is what the kernel named in rt_stats.kernel issues."""
    n_sph, n_pl = census["spheres"], census["planes"]
    ls, lp = census.get("csg_sphere_leaves", 0), census.get("csg_plane_leaves", 0)
    algorithmic = F_SPHERE_WRITTEN * (n_sph + ls) + 7.0 + F_PLANE_WRITTEN * (n_pl + lp) + F_EPILOGUE
    if "path_tile<" in kernel:  # register tiles on the fp32 MFMA; the big (sky) sphere is scanned per lane
        executed = F_SPHERE_TILE * max(n_sph - 1, 0) + F_SPHERE_OBJECT * min(n_sph, 1) + F_PLANE_FMA * n_pl + F_EPILOGUE + F_EXACT_RECORD
    elif "wide<ISO" in kernel:  # register rows in the world-space form; the big (sky) sphere is scanned per lane
        executed = F_SPHERE_WORLD * max(n_sph - 1, 0) + F_SPHERE_OBJECT * min(n_sph, 1) + F_PLANE_FMA * n_pl + F_EPILOGUE + F_EXACT_RECORD
    elif "EXACT" in kernel or "simple" in kernel:  # unfused, the reference's operation order
        executed = algorithmic
    else:  # k_render_path_wave / k_render_path<FMA...>, wide<GEN>: object-space FMA form; CSG leaves by the truth-table evaluator
        executed = F_SPHERE_OBJECT * n_sph + F_PLANE_FMA * n_pl + F_CSG_SPHERE_LEAF_TT * ls + F_CSG_PLANE_LEAF_TT * lp + F_EPILOGUE
    return executed, algorithmic


def load_profile(name):
    """A committed counter summary of round 3 (profiles/r3/<name>_pmc.json, made by scripts/pmc_run.sh: separate rocprofv3 --pmc
    passes of the same configuration) or None."""
    path = os.path.join(ROOT, "profiles", "r3", name + "_pmc.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        j = json.load(f)
    j["_source"] = os.path.relpath(path, ROOT)
    return j


def valu_issue_block(profile, rays, kernel_ms):
    """The vector ALU's issue slots as a bound: VALU wave-instructions per ray by class (the profile's SQ_INSTS_VALU* counters) x the
    measured issue cost of each class, against the SIMD-cycles per ray of this very launch.  None without the counters."""
    c = profile["counters"]
    if "SQ_INSTS_VALU_FMA_F32" not in c:
        return None
    n = float(profile["rays_per_launch"]) * float(profile.get("launches", 1))
    fast = (c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) / n
    trans = c["SQ_INSTS_VALU_TRANS_F32"] / n
    mfma = (c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_F32", 0.0)) / n  # counted in SQ_INSTS_VALU, issued to the matrix pipe
    total = c["SQ_INSTS_VALU"] / n - mfma
    model = fast * VALU_CYCLES_FAST + trans * VALU_CYCLES_TRANS + (total - fast - trans) * VALU_CYCLES_OTHER
    clk = profile.get("derived", {}).get("clock_ghz", 2.3)
    measured = kernel_ms * 1e-3 * clk * 1e9 * SIMDS / rays
    return {"valu_wave_instr_per_ray": total, "fp32_add_mul_fma_per_ray": fast, "transcendental_per_ray": trans,
            "modelled_valu_cycles_per_ray": model, "simd_cycles_per_ray_this_launch": measured, "clock_ghz": clk,
            "frac": model / measured, "source": profile["_source"], "saturated": model / measured >= 0.9,
            "note": "modelled / measured issue cycles: near 1 = the vector ALU issues all the time.  'other' is priced at 4.2 cycles although the "
                    "plain moves, integer adds and logic ops in it issue at 2.3, so the model is an upper estimate and can exceed 1 by that "
                    "margin (C3 / C4: 1.07-1.08)"}


def roofline_block(census, kernel, rays, kernel_ms, profile=None, bvh=False):
    """One roofline block.  Every number can be recomputed from this line and the profile it names:
       frac_executed    = rays x flops_per_ray (what the kernel issues, FMA = 2) / kernel_ms / peak            (<= 1 by construction)
       frac_algorithmic = rays x algorithmic_flops_per_ray (the reference's arithmetic as written) / kernel_ms / peak
                          — exceeds 1 where the kernel does the same job with fewer flops (18 instead of 52 per sphere test)
       achieved / frac  = the executed figures (the contract's field names)
       traffic          = HBM-side bytes per launch from the PMC counters of `traffic_source`: 2 x FETCH_SIZE + WRITE_SIZE, each counter in
                          its own rocprofv3 --pmc pass of the same configuration, in KB as rocprofv3 reports them, FETCH_SIZE doubled as
                          MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64); the counters sit on the L2's fabric
                          side, so Infinity-Cache hits are included: an upper bound on HBM bytes.  null when no such profile is committed —
                          never measured in the timed run itself
       valu_issue       = for the kernels that are NOT flop-bound (a handful of objects per ray): the bound that applies — the
                          vector ALU's issue slots.  VALU wave-instructions per ray by class from the profile's SQ_INSTS_VALU* counters x
                          the measured issue cost of each class, against the SIMD-cycles per ray of this very launch."""
    if bvh:  # the hierarchy skips most of the tests a brute-force flop count assumes: no fraction of a roofline
        return {"bound": "valu", "achieved": None, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": kernel, "rays_per_launch": rays, "kernel_ms": kernel_ms,
                "note": "BVH mode changes the algorithmic work; see DESIGN.md 4.6 for its step counts"}
    executed, algorithmic = flops_per_ray(census, kernel)
    ach = rays * executed / (kernel_ms * 1e-3) / 1e12
    alg = rays * algorithmic / (kernel_ms * 1e-3) / 1e12
    if "tile_cull" in kernel:
        # The cluster-culled tile kernel (rt_tile_cull.h) multiplies a sphere tile only with the rays that meet its bound: the flops it
        # executes are data dependent, so they are taken from the committed counter profile of this very workload (None without one),
        # never from a formula.  It is bound by the vector ALU's ISSUE slots (valu_issue), not by flops and not by the matrix pipe:
        #   achieved / frac  = fp32 flops the vector ALU executed (2 x FMA + MUL + ADD wave-instructions x the mean number of active lanes)
        #                      against the fp32 vector peak
        #   mfma             = bf16 matrix flops (32768 per v_mfma_f32_32x32x16_bf16) against the dense bf16 peak
        #   frac_algorithmic = the reference's arithmetic as written (52 fp32 flops per sphere test, ALL spheres) against the fp32 peak
        out = {"bound": "valu", "peak_kind": "fp32 vector ALU, 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz; the kernel is bound by the vector ALU's issue slots (valu_issue), "
                                              "its bf16 matrix work is in `mfma`",
               "achieved": None, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "frac_executed": None,
               "frac_algorithmic": alg / FP32_VALU_PEAK_TFLOPS, "algorithmic_peak_kind": "fp32 vector peak, 157.3 TFLOP/s",
               "algorithmic_flops_per_ray": algorithmic, "algorithmic_tflops": alg,
               "traffic": None, "traffic_source": None, "kernel": kernel, "rays_per_launch": rays, "kernel_ms": kernel_ms,
               "note": "this kernel skips the tile products whose outcome is known (a ray's line misses the cluster's bounding sphere): the equivalent of "
                       "`mfma.tile_products_per_32_rays` of 128 are left on the headline world, the frame is bit-identical (share0_sha256), the time per ray more than halves.  What is left is list building, "
                       "LDS gathers and candidate handling - vector-ALU ISSUE work, few flops: `frac` (fp32 flops executed / fp32 peak) is small by nature, "
                       "`valu_issue.frac` says how full the issue slots are, `mfma.frac` how busy the matrix pipe is.  The kernel that does ALL the tile "
                       "products and sits on the matrix-pipe roofline is the sub-record N4096-share-scan-all (RT_FLAG_SCAN_ALL_TILES)."}
        if profile is not None and "cull" in profile.get("kernel", ""):
            c, launches = profile["counters"], float(profile.get("launches", 1))
            n = float(profile["rays_per_launch"]) * launches
            secs = kernel_ms * 1e-3
            if "SQ_INSTS_VALU_FMA_F32" in c:
                lanes = profile.get("derived", {}).get("active_lanes_per_valu_instr", 64.0)
                flops = (2.0 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) / n * lanes
                out["achieved"] = rays * flops / secs / 1e12
                out["frac"] = out["frac_executed"] = out["achieved"] / FP32_VALU_PEAK_TFLOPS
                out["valu_fp32_flops_per_ray"] = flops
                out["mean_active_lanes"] = lanes
            if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
                # the kernel mixes two shapes (bounds: 32x32x16 = 32768 flops; cluster tiles: 16x16x32 = 16384), so the flops come from the
                # MOPS counter: 512 flops per counted op (64 per 32x32x16 MFMA — profiles/r3/n4096_share_scan_all_pmc.json: 16.0 MFMAs and 1024 ops per ray)
                mfma_flops = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / n * 512.0
                tf = rays * mfma_flops / secs / 1e12
                n_tiles = -(-max(census["spheres"] - 1, 0) // 32)
                out["mfma"] = {"achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "frac": tf / BF16_MFMA_PEAK_TFLOPS,
                               "mfma_instr_per_ray": c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0) / n, "mfma_flops_per_ray": mfma_flops,
                               "tile_products_per_32_rays": mfma_flops * 32.0 / (4.0 * 32768.0), "scan_all_tile_products_per_32_rays": float(n_tiles)}
            vi = valu_issue_block(profile, rays, kernel_ms)
            if vi:
                out["valu_issue"] = vi
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                out["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / launches
                out["traffic_source"] = profile["_source"]
            out["source"] = profile["_source"]
        return out
    if "BF16x3" in kernel:
        # The dominant unit is the bf16 matrix pipe: `achieved` / `frac` / `frac_executed` count ITS flops against ITS peak.  The fp32
        # work that stays on the vector ALU (d = b^2 - c per test, the per-lane sphere, planes, hit record, exact record, shading) is
        # reported beside it against the fp32 peak; `frac_algorithmic` prices the reference's arithmetic as written (52 fp32 flops per
        # sphere test) against the fp32 vector peak, the unit the reference's own code would run on.
        n_tests = -(-max(census["spheres"] - 1, 0) // 32) * 32  # register spheres, padded to whole tiles (what the MFMAs multiply)
        mfma = rays * F_SPHERE_TILE_BF16_MFMA * n_tests / (kernel_ms * 1e-3) / 1e12
        valu = rays * (executed - (F_SPHERE_TILE - 2.0) * max(census["spheres"] - 1, 0)) / (kernel_ms * 1e-3) / 1e12
        out = {"bound": "mfma", "peak_kind": "bf16 dense MFMA (256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz); v_mfma_f32_32x32x16_bf16 on 3-way bf16 splits of the binary32 operands, binary32 accumulate",
               "achieved": mfma, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma / BF16_MFMA_PEAK_TFLOPS,
               "frac_executed": mfma / BF16_MFMA_PEAK_TFLOPS,
               "mfma_flops_per_ray": F_SPHERE_TILE_BF16_MFMA * n_tests,
               "valu_fp32": {"tflops": valu, "peak": FP32_VALU_PEAK_TFLOPS, "frac": valu / FP32_VALU_PEAK_TFLOPS,
                             "flops_per_ray": executed - (F_SPHERE_TILE - 2.0) * max(census["spheres"] - 1, 0)},
               "frac_algorithmic": alg / FP32_VALU_PEAK_TFLOPS, "algorithmic_peak_kind": "fp32 vector peak, 157.3 TFLOP/s",
               "algorithmic_flops_per_ray": algorithmic, "algorithmic_tflops": alg,
               "traffic": None, "traffic_source": None, "kernel": kernel, "rays_per_launch": rays, "kernel_ms": kernel_ms}
        if profile is not None and "FETCH_SIZE" in profile["counters"] and "WRITE_SIZE" in profile["counters"]:
            c, launches = profile["counters"], float(profile.get("launches", 1))
            out["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / launches
            out["traffic_source"] = profile["_source"]
        return out
    out = {"bound": "mfma" if "path_tile<" in kernel else "valu", "peak_kind": "fp32, vector = matrix: v_mfma_f32_32x32x2_f32 runs on the VALU's fp32 lanes (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz)",
           "achieved": ach, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_VALU_PEAK_TFLOPS,
           "frac_executed": ach / FP32_VALU_PEAK_TFLOPS, "frac_algorithmic": alg / FP32_VALU_PEAK_TFLOPS,
           "flops_per_ray": executed, "algorithmic_flops_per_ray": algorithmic, "algorithmic_tflops": alg,
           "traffic": None, "traffic_source": None, "kernel": kernel, "rays_per_launch": rays, "kernel_ms": kernel_ms}
    if census.get("csg"):
        out["csg_counted"] = "leaf tests only (SURVEY.md 8d); per-hit inside tests and node re-mapping not counted: fractions are lower bounds by at most 30 %"
    if profile is not None:
        c, launches = profile["counters"], float(profile.get("launches", 1))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / launches
            out["traffic_source"] = profile["_source"]
        vi = valu_issue_block(profile, rays, kernel_ms)
        if vi:
            out["valu_issue"] = vi
    return out


def cpu_baseline(scene, text, args):
    """Times the CPU on a bounded sample of the same workload: the synthetic scene at a small resolution.
    kind "reference": the unmodified reference (oracle/_ref/rt_ref, single-threaded by construction); the
    ray count comes from the CPU restatement run on the same configuration with the reference's serial PCG
    (it reproduces the reference bit for bit, so it traces exactly the same rays).  kind "port": only the
    restatement is timed (when the reference binary is not there)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from raytracer_amd import abi
    w, h, aa = args.cpu_width, args.cpu_height, 1  # defaults: 8x5 on the 4096-sphere world: ~5 s of the reference + ~7 s of the port on one core of the GPU box's host
    cfg = O.make_cfg(w, h, "path", spp_edge=aa, n_rays=args.n_rays, roulette=args.roulette, max_depth=args.max_depth,
                     seq_number=54, rng_mode=abi.RT_RNG_SERIAL_REFERENCE)
    t0 = time.perf_counter()
    _, stats = O.render(scene, cfg)
    port_s = time.perf_counter() - t0
    sample = f"{scene.count('n_objects')}-object synthetic scene, path {args.n_rays}/{args.roulette}/{args.max_depth}, {w}x{h} AA=1 ({stats.rays} rays)"
    out = {"value": stats.rays / port_s / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample,
           "seconds": port_s}
    if os.path.exists(O.REF_BIN):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "scene.txt")
            with open(path, "w") as f:
                f.write(text)
            res = subprocess.run([O.REF_BIN, "render", path, "-o", os.path.join(tmp, "o.pfm"), "--mode", "path", "--width", str(w),
                                  "--height", str(h), "--aa", str(aa), "--n_rays", str(args.n_rays), "--roulette", str(args.roulette),
                                  "--max-depth", str(args.max_depth), "--seq", "54"], capture_output=True, text=True, check=True)
            ref_s = json.loads(res.stdout)["elapsed_s"]
        out.update({"value": stats.rays / ref_s / 1e6, "kind": "reference", "seconds": ref_s,
                    "port_value": stats.rays / port_s / 1e6})
    return out


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as a fresh child (this process has not touched the GPU and
    never will) and pass rank 0's output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def measure_configs(device, log):
    """BASELINE.json's configs C1-C4 and the full N=512 frame, each as one sub-record (N = 1 only).  Times are HIP events around the
    very launch whose ray count is reported (rt_stats of the same call); C1-C4: best of 3 after one warm-up launch."""
    import torch
    from raytracer_amd.scene import FlatScene
    from raytracer_amd.synth import synthetic_scene
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode, RowBands
    golden = os.path.join(ROOT, "tests", "golden")
    recs = []

    def run(cid, label, scene, mode, w, h, aa, rows=None, reps=3, note=None, profile=None):
        world = DeviceWorld(scene, device.index)
        tracer = GpuImageTracer(w, h, scene.camera, aa)
        rows = rows or RowBands.whole(h)
        out = torch.empty((rows.n_local_rows, w, 3), dtype=torch.float32, device=device)
        best, st = None, None
        for k in range(reps + (1 if reps > 1 else 0)):  # reps > 1: the first launch is a warm-up
            tracer.fire_all_rays(world, mode, rows, out=out, want_stats=True)
            if (reps == 1 or k > 0) and (best is None or tracer.last_stats.kernel_ms < best):
                best, st = float(tracer.last_stats.kernel_ms), tracer.last_stats
        kernel = st.kernel.decode()
        rec = {"id": cid, "workload": label, "objects": scene.count("n_objects"), "kernel": kernel, "grid": int(st.grid), "block": int(st.block),
               "samples": int(st.samples), "rays": int(st.rays), "capped_samples": int(st.capped_samples), "kernel_ms": best,
               "Mrays_per_s": st.rays / best / 1e3,
               "roofline": roofline_block(scene_census(scene), kernel, int(st.rays), best, load_profile(profile) if profile else None)}
        if note:
            rec["note"] = note
        world.close()
        recs.append(rec)
        log(f"{cid}: {best:.3f} ms, {rec['Mrays_per_s']:.0f} Mrays/s, {kernel}")

    onoff = FlatScene.from_json(os.path.join(golden, "demo_onoff.scene.json"))
    demo = FlatScene.from_json(os.path.join(golden, "demo_path.scene.json"))
    csg = FlatScene.from_json(os.path.join(golden, "csg_example.scene.json"))
    run("C1", "samples/demo_onoff_tracing.txt, --mode onoff, 1280x960, AA=1", onoff, RenderMode("onoff"), 1280, 960, 1)
    run("C2-flat", "samples/demo_path_tracing.txt, --mode flat, 1280x960, AA=2", demo, RenderMode("flat"), 1280, 960, 2)
    run("C2-pointlight", "samples/demo_path_tracing.txt, --mode pointlight, 1280x960, AA=2", demo, RenderMode("pointlight"), 1280, 960, 2)
    run("C3", "samples/demo_path_tracing.txt, --mode path 10/3/5, 1280x960, AA=2", demo, RenderMode("path"), 1280, 960, 2, profile="c3")
    run("C4", "samples/csg_example.txt, --mode path 10/3/5, 1920x1080, AA=4", csg, RenderMode("path"), 1920, 1080, 4, profile="c4")
    small = synthetic_scene(512)[0]
    run("N512-frame", "synthetic 512-random-sphere + floor + sky, --mode path 10/3/5, 1920x1080, AA=4, the WHOLE frame in one launch (round 2's bench default)",
        small, RenderMode("path"), 1920, 1080, 4, reps=1, profile="n512")
    # The same frame WITHOUT any MFMA instruction (BASELINE.json's north_star expected no dense contraction in this path; the
    # brute-force ray x sphere filter turned out to be one): the vector-ALU filter kernel.
    run("N512-valu-only", "the same frame with RT_FLAG_VALU_ONLY: the sphere filter on the vector ALU, no matrix-core instruction",
        small, RenderMode("path", valu_only=True), 1920, 1080, 4, reps=1)
    big = synthetic_scene(4096)[0]
    # The headline's own share 0 with every ray against every tile (RT_FLAG_SCAN_ALL_TILES: round 3a's kernel, the matrix-pipe roofline)
    run("N4096-share-scan-all", "the headline's share 0 (1/16 of the 4096-sphere 1920x1080 AA=4 frame) with RT_FLAG_SCAN_ALL_TILES: the bf16 matrix filter on "
        "every tile, no cluster bounds", big, RenderMode("path", scan_all_tiles=True), 1920, 1080, 4, rows=RowBands.cyclic(1080, 0, 16, 1), reps=1,
        profile="n4096_share_scan_all")
    # The headline's own share 0 with the filter in binary32 on the vector ALU: the figure to hold against `value` if the bf16-split
    # matrix filter is to be discounted (it loses no accuracy — DESIGN.md 4.4 — but it is not binary32 arithmetic instruction by instruction).
    run("N4096-share-valu-only", "the headline's share 0 (1/16 of the 4096-sphere 1920x1080 AA=4 frame) with RT_FLAG_VALU_ONLY: binary32 filter on the vector ALU, no matrix-core instruction",
        big, RenderMode("path", valu_only=True), 1920, 1080, 4, rows=RowBands.cyclic(1080, 0, 16, 1), reps=1)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=4)
    ap.add_argument("--spheres", type=int, default=4096)
    ap.add_argument("--shares", type=int, default=16, help="a step renders 1 / SHARES of the frame's rows per rank (row-cyclic); 1 = the whole frame per step")
    ap.add_argument("--scene", default=None, choices=["demo_onoff", "demo_path", "csg_example"],
                    help="one of the reference's sample scenes (flattened fixture in tests/golden/) instead of the synthetic one")
    ap.add_argument("--n_rays", type=int, default=10)
    ap.add_argument("--roulette", type=int, default=3)
    ap.add_argument("--max-depth", dest="max_depth", type=int, default=5)
    ap.add_argument("--cpu-width", dest="cpu_width", type=int, default=8)
    ap.add_argument("--cpu-height", dest="cpu_height", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config sub-records of the N=1 line")
    ap.add_argument("--bvh", action="store_true",
                    help="RT_FLAG_ACCEL_BVH: not the reference's algorithmic work, so the line carries no roofline fraction")
    ap.add_argument("--valu-only", action="store_true", help="RT_FLAG_VALU_ONLY: the large-world sphere filter on the vector ALU (no MFMA instruction)")
    ap.add_argument("--scan-all-tiles", dest="scan_all_tiles", action="store_true",
                    help="RT_FLAG_SCAN_ALL_TILES: every ray against every sphere tile (no cluster bounds)")
    ap.add_argument("--save-pfm", default=None, help="write the frame buffer here at the end (rank 0; the rows of shares not rendered are zero)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on fewer GPUs than ranks (ranks share devices, gather staged through the host)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import gather_shares, share_rows
    from raytracer_amd.synth import synthetic_scene
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.shares < 1 or args.shares > args.height:
        raise SystemExit("--shares must be in 1 ... height")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world_size > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    staged = world_size > 1 and args.backend == "gloo"

    if args.scene:
        from raytracer_amd.scene import FlatScene
        scene, text = FlatScene.from_json(os.path.join(ROOT, "tests", "golden", f"{args.scene}.scene.json")), None
        args.no_cpu_baseline = True  # the CPU leg is defined on the synthetic scene (it needs the scene text)
    else:
        scene, text = synthetic_scene(args.spheres)
    world = DeviceWorld(scene, dev_index)
    tracer = GpuImageTracer(args.width, args.height, scene.camera, args.aa)
    mode = RenderMode("path", n_rays=args.n_rays, roulette=args.roulette, max_depth=args.max_depth, seq_number=54, bvh=args.bvh, valu_only=args.valu_only,
                      scan_all_tiles=args.scan_all_tiles)
    max_rows = -(-args.height // args.shares)
    local = torch.empty((max_rows, args.width, 3), dtype=torch.float32, device=device)
    full = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=device) if rank == 0 else None
    gather_dev = torch.device("cpu") if staged else device
    padded = torch.empty((max_rows, args.width, 3), dtype=torch.float32, device=gather_dev) if world_size > 1 else None
    parts = [torch.empty_like(padded) for _ in range(world_size)] if (world_size > 1 and rank == 0) else None

    def rows_of(step, r):
        return share_rows(args.height, step, r, world_size, args.shares)

    def render(step, want_stats):
        rows = rows_of(step, rank)
        tracer.fire_all_rays(world, mode, rows, out=local[: rows.n_local_rows], want_stats=want_stats)
        return rows

    def gather(step):
        rows = rows_of(step, rank)
        gather_shares((local.cpu() if staged else local)[: rows.n_local_rows], [rows_of(step, r) for r in range(world_size)], full, padded, parts)

    def barrier():
        if world_size > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"scene uploaded: {scene.count('n_objects')} objects; frame {args.width}x{args.height} AA={args.aa}, {args.shares} shares, {world_size} rank(s)")
    # Warm-up.  Step numbering continues into the timed steps, so the run walks through the frame's shares.
    # The first pass also names the kernel.
    render(0, True)
    gather(0)
    st = tracer.last_stats
    kernel_name = st.kernel.decode()
    log(f"first pass: {int(st.rays)} rays, {int(st.samples)} samples, kernel {st.kernel_ms:.1f} ms, {kernel_name} grid {st.grid} x {st.block}")
    import hashlib
    share0_sha = hashlib.sha256(local[: rows_of(0, rank).n_local_rows].cpu().numpy().tobytes()).hexdigest() if rank == 0 else None
    for k in range(1, args.warmup):
        render(k, False)
        gather(k)
    torch.cuda.synchronize()
    barrier()

    first = max(args.warmup, 1)
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays_local = samples_local = 0
    for k in range(args.steps):
        ev[k][0].record()  # torch's current stream == the stream rt_render launches on
        render(first + k, True)  # with the launch's tallies: shares differ slightly in rays (one 528-byte copy after the kernel, ~20 us of a step)
        rays_local += int(tracer.last_stats.rays)
        samples_local += int(tracer.last_stats.samples)
        ev[k][1].record()
        gather(first + k)
        ev[k][2].record()  # after the gather: torch makes the current stream wait for the collective
        if args.steps > 1:
            torch.cuda.current_stream().synchronize()
        log(f"step {k + 1}/{args.steps} done: {int(tracer.last_stats.rays)} rays, kernel {tracer.last_stats.kernel_ms:.1f} ms")
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / args.steps
    gather_ms = sum(b.elapsed_time(c) for _, b, c in ev) / args.steps  # on rank 0 this includes waiting for the slowest rank

    t = torch.tensor([elapsed, kernel_ms, -kernel_ms], dtype=torch.float64, device=gather_dev)
    r = torch.tensor([rays_local, samples_local], dtype=torch.int64, device=gather_dev)
    g = torch.tensor([gather_ms], dtype=torch.float64, device=gather_dev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(g, op=dist.ReduceOp.MIN)  # the rank that finished rendering last waits for nobody: the transfer itself
    elapsed, kernel_ms_max, kernel_ms_min = float(t[0]), float(t[1]), -float(t[2])
    rays_total, samples_total = int(r[0]), int(r[1])

    if rank == 0:
        census = scene_census(scene)
        default_workload = (not args.scene and args.spheres == 4096 and args.width == 1920 and args.height == 1080 and args.aa == 4 and args.shares == 16
                            and (args.n_rays, args.roulette, args.max_depth) == (10, 3, 5) and not args.bvh and not args.valu_only)
        rays_per_step_rank0 = rays_local / args.steps
        out = {
            "metric": "path-mode Mrays/s (frame ms in frame_ms_estimate), 1920x1080 AA=4",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "hit records, shading and scattering in binary32, sample sums in 64-bit fixed point; the large-world sphere FILTER (which "
                          "only ranks candidates: the winner's record is recomputed as the reference writes it) multiplies binary32 operands split "
                          "exactly into three bf16 parts on the bf16 matrix cores with binary32 accumulation — measured error 2^-22.4 of sum|terms|, "
                          "the binary32 fma chain's 2^-22.6 (scripts/microbench/bf16_probe.hip)",
            "config": {"workload": (f"samples/{args.scene} (reference sample scene)" if args.scene else
                                    f"synthetic {args.spheres}-random-sphere + floor + sky scene") + f", --mode path, "
                                   f"{args.width}x{args.height}, AA={args.aa}, n_rays={args.n_rays}, roulette={args.roulette}, "
                                   f"max-depth={args.max_depth}, seq-number=54" + (", BVH mode (RT_FLAG_ACCEL_BVH)" if args.bvh else "") + (", RT_FLAG_SCAN_ALL_TILES" if args.scan_all_tiles else "") +
                                   (f"; one step = one row-cyclic share of 1/{args.shares} of the frame per GPU (rows k, k + {args.shares}, ...), "
                                    f"consecutive steps render consecutive shares" if args.shares > 1 else "; one step = the whole frame"),
                       "objects": scene.count("n_objects"), "top_level": {k: census[k] for k in ("spheres", "planes", "csg")},
                       "shares_per_frame": args.shares, "samples_per_step": samples_total // args.steps,
                       "rays_per_step": rays_total / args.steps, "rays_per_sample": rays_total / max(samples_total, 1),
                       "parallelism": f"{world_size} ranks x 1 share per step, RCCL gather to rank 0" if world_size > 1 else "single"},
            "frame_ms_estimate": elapsed / args.steps * 1e3 * args.shares / world_size,
            "kernel_ms_per_step": {"max_over_ranks": kernel_ms_max, "min_over_ranks": kernel_ms_min},
            "gather_ms_per_step": {"rank0_including_wait_for_slowest_rank": gather_ms, "min_over_ranks": float(g[0])},
            # dominant kernel on rank 0's shares: ITS rays / ITS duration (k_resolve rides in the same interval, < 0.1 %)
            "roofline": roofline_block(census, kernel_name, rays_per_step_rank0, kernel_ms,
                                       load_profile("n4096_share_scan_all" if args.scan_all_tiles else "n4096_share") if (default_workload and world_size == 1) else None,
                                       bvh=args.bvh),
            # share 0 as rank 0 rendered it in the first pass: bit-identical for any number of ranks and any kernel shape
            "share0_sha256": share0_sha,
        }
        if args.save_pfm:
            img = full.cpu().numpy()
            with open(args.save_pfm, "wb") as f:
                f.write(b"PF\n%d %d\n-1.0\n" % (args.width, args.height))
                f.write(img[::-1].astype("<f4").tobytes())
    world.close()
    if rank == 0:
        if world_size == 1 and not args.no_configs:
            out["configs"] = measure_configs(device, log)
        if world_size == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline on a bounded sample ...")
            out["cpu_baseline"] = cpu_baseline(scene, text, args)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
