#!/usr/bin/env python3
"""bench.py — path-mode throughput of the MI355X tracing backend.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one full frame of BASELINE.json's metric: --mode path, 1920x1080, AA=4 (16 spp), n_rays=10, roulette=3,
max-depth=5, on the synthetic N-random-sphere + floor + sky scene of raytracer_amd/synth.py.  Default N = 512:
3.1e10 rays per frame, about 5 s on one MI355X.  Every object is tested for every ray (World::ray_intersection is a
linear scan, reference include/shapes.hpp:547-561); a ray = one such closest-hit query.

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its N ranks itself (a child
`python -m torch.distributed.run`, decided before anything touches the GPU) and relays rank 0's line.  The frame's rows
are dealt round-robin row by row (strong scaling: the frame is fixed), each rank renders its bands into HBM and rank 0
gathers them over RCCL.  Scene and framebuffers are resident in HBM before the timed region.

At N = 1 the line also carries `configs`: every BASELINE.json config measured in the same run — C1 (onoff), C2 (flat and
pointlight), C3, C4 (path on the reference's sample scenes, full size) and C5, the 4096-sphere scene, as ONE 1/16
row-cyclic slice of its 1920x1080 AA=4 frame (exactly one rank's share of a 16-GPU run; a whole frame is 16 of them,
~7 minutes on one GPU) — each with its kernel, ray count, kernel ms (HIP events of the same launch) and roofline block.

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

# fp32 flops one object test costs (FMA = 2, everything else 1), by the arithmetic form a kernel uses:
#   world-space sphere filter (wide ISO register rows, rt_wide.h): b = O.u - C.u (3 fma = 6), c = (O.O + k) - 2 O.C (1 add + 3 fma = 7),
#     disc = b^2 - c (1 fma = 2): 15 flops in 8 instructions
#   object-space sphere, FMA form (sphere_scan_test<true>): D' = M D (15), O' = M O + t (18), a (5), b (5), c (6), disc (3)
#   object-space sphere as the reference writes it (and the exact kernels execute it): 33 + 19
#   plane, FMA form: z row only (6 + 5) + t (2); as written: 33 + 2
#   the same filter on the matrix cores (rt_tile.h): four v_mfma_f32_32x32x2_f32 per 32 rays x 32 spheres = 4 x 4096 / 1024 = 16 flops per
#     test (K = 4 for b and for c: the constant terms 1 * O.u and k * 1 are multiply-adds too) + the VALU's d = b^2 - c (2): 18
F_SPHERE_TILE = 18.0
F_SPHERE_WORLD, F_SPHERE_OBJECT, F_SPHERE_WRITTEN = 15.0, 52.0, 52.0
F_PLANE_FMA, F_PLANE_WRITTEN = 13.0, 35.0
F_EPILOGUE = 54.0 + 75.0  # the winner's hit record + scatter / shading, same in both counts
F_EXACT_RECORD = 59.0 + 59.0  # large-world kernels: the winner and the sphere the ray leaves, each once more as the reference writes it


def scene_census(scene):
    import numpy as np
    top = scene.objects[scene.objects >= 0]
    n_sph = int(np.sum(scene.shape_kind[top] == 0))
    return {"spheres": n_sph, "planes": int(len(top)) - n_sph, "csg": int(np.sum(scene.objects < 0))}


def flops_per_ray(census, kernel: str):
    """-> (executed, algorithmic) fp32 flops of one closest-hit query + shading.  `algorithmic` is the reference's
    arithmetic as written (52 per sphere test, +7 for the one sphere that is hit, 35 per plane); `executed` is what the
    kernel named in rt_stats.kernel issues.  CSG trees are not counted in either: a lower bound where a scene has one."""
    n_sph, n_pl = census["spheres"], census["planes"]
    algorithmic = F_SPHERE_WRITTEN * n_sph + 7.0 + F_PLANE_WRITTEN * n_pl + F_EPILOGUE
    if "path_tile<" in kernel:  # register tiles on the fp32 MFMA; the big (sky) sphere is scanned per lane
        executed = F_SPHERE_TILE * max(n_sph - 1, 0) + F_SPHERE_OBJECT * min(n_sph, 1) + F_PLANE_FMA * n_pl + F_EPILOGUE + F_EXACT_RECORD
    elif "wide<ISO" in kernel:  # register rows in the world-space form; the big (sky) sphere is scanned per lane
        executed = F_SPHERE_WORLD * max(n_sph - 1, 0) + F_SPHERE_OBJECT * min(n_sph, 1) + F_PLANE_FMA * n_pl + F_EPILOGUE + F_EXACT_RECORD
    elif "EXACT" in kernel or "simple" in kernel:  # unfused, the reference's operation order
        executed = algorithmic
    else:  # k_render_path<FMA...>, wide<GEN>: object-space FMA form
        executed = F_SPHERE_OBJECT * n_sph + F_PLANE_FMA * n_pl + F_EPILOGUE
    return executed, algorithmic


def roofline_block(census, kernel, rays, kernel_ms, traffic=None, bvh=False):
    if bvh:  # the hierarchy skips most of the tests a brute-force flop count assumes: no fraction of a roofline
        return {"bound": "valu", "achieved": None, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": kernel, "rays_per_launch": rays, "kernel_ms": kernel_ms,
                "note": "BVH mode changes the algorithmic work; see DESIGN.md 4.6 for its step counts"}
    executed, algorithmic = flops_per_ray(census, kernel)
    ach = rays * executed / (kernel_ms * 1e-3) / 1e12
    # the tile kernel's flops run on the binary32 MFMA, whose dense peak equals the vector fp32 peak (it uses the same lanes)
    return {"bound": "mfma" if "path_tile<" in kernel else "valu", "achieved": ach, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_VALU_PEAK_TFLOPS,
            "traffic": traffic, "kernel": kernel, "flops_per_ray": executed, "rays_per_launch": rays, "kernel_ms": kernel_ms,
            "algorithmic_flops_per_ray": algorithmic, "algorithmic_tflops": rays * algorithmic / (kernel_ms * 1e-3) / 1e12,
            "note": "achieved / frac count the flops the kernel issues (FMA = 2); algorithmic_* count the reference's arithmetic "
                    "as written for the same rays (52 flops per sphere test) and may exceed the peak"}


def pmc_traffic(args):
    """HBM-side bytes per launch of the dominant kernel, from the committed counter passes of THIS configuration
    (scripts/pmc_run.sh -> profiles/r2/n512_pmc.json: FETCH_SIZE and WRITE_SIZE in KB, each in its own rocprofv3 --pmc
    pass, summed over the launches of one run and divided by their number).  None for any other configuration:
    counters cannot be read from inside the timed run."""
    default = (not args.bvh and not args.valu_only and args.spheres == 512 and args.width == 1920 and args.height == 1080 and args.aa == 4 and not args.scene
               and args.n_rays == 10 and args.roulette == 3 and args.max_depth == 5)
    if not default:
        return None
    for rel in (("profiles", "r2", "n512_pmc.json"), ("profiles", "r1", "g_n512_pmc.json")):
        path = os.path.join(ROOT, *rel)
        if os.path.exists(path):
            with open(path) as f:
                j = json.load(f)
            c = j["counters"]
            return (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / float(j.get("launches", 2))
    return None


def cpu_baseline(scene, text, args):
    """Times the CPU on a bounded sample of the same workload: the synthetic scene at a small resolution.
    kind "reference": the unmodified reference (oracle/_ref/rt_ref, single-threaded by construction); the
    ray count comes from the CPU restatement run on the same configuration with the reference's serial PCG
    (it reproduces the reference bit for bit, so it traces exactly the same rays).  kind "port": only the
    restatement is timed (when the reference binary is not there)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from raytracer_amd import abi
    w, h, aa = args.cpu_width, args.cpu_height, 1
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
    """BASELINE.json's five configs, each as one sub-record (N = 1 only).  Times are HIP events around the very launch
    whose ray count is reported (rt_stats of the same call); C1-C4: best of 3 after one warm-up launch."""
    import torch
    from raytracer_amd.scene import FlatScene
    from raytracer_amd.synth import synthetic_scene
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode, RowBands
    golden = os.path.join(ROOT, "tests", "golden")
    recs = []

    def run(cid, label, scene, mode, w, h, aa, rows=None, reps=3, note=None):
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
               "roofline": roofline_block(scene_census(scene), kernel, int(st.rays), best)}
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
    run("C3", "samples/demo_path_tracing.txt, --mode path 10/3/5, 1280x960, AA=2", demo, RenderMode("path"), 1280, 960, 2)
    run("C4", "samples/csg_example.txt, --mode path 10/3/5, 1920x1080, AA=4", csg, RenderMode("path"), 1920, 1080, 4,
        note="roofline counts the two top-level objects only (the CSG tree's leaf and inside tests are not counted): a lower bound")
    big = synthetic_scene(4096)[0]
    run("C5-slice", "synthetic 4096-random-sphere + floor + sky, --mode path 10/3/5, 1920x1080, AA=4: rows 0, 16, 32, ... "
        "(one 1/16 row-cyclic slice = one rank's share of 16; a whole frame is 16 such launches)", big, RenderMode("path"), 1920, 1080, 4,
        rows=RowBands.cyclic(1080, 0, 16, band_rows=1), reps=1,
        note="whole-frame estimate = 16 x kernel_ms (slices differ by < 1 %, profiles/r1/h_full_n4096_1080p_aa4.json); "
             "configs[4]'s 3840x2160 frame is 64 slices of this cost")
    recs[-1]["frame_ms_estimate_one_gpu"] = 16.0 * recs[-1]["kernel_ms"]
    # The headline workload once more WITHOUT any MFMA instruction (BASELINE.json's north_star expected no dense contraction
    # in this path; the brute-force ray x sphere filter turned out to be one): the vector-ALU filter kernel.
    run("N512-valu-only", "the bench default (synthetic 512-sphere scene, path 10/3/5, 1920x1080, AA=4) with RT_FLAG_VALU_ONLY: "
        "the sphere filter on the vector ALU, no matrix-core instruction", synthetic_scene(512)[0], RenderMode("path", valu_only=True), 1920, 1080, 4, reps=1)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=4)
    ap.add_argument("--spheres", type=int, default=512)
    ap.add_argument("--scene", default=None, choices=["demo_onoff", "demo_path", "csg_example"],
                    help="one of the reference's sample scenes (flattened fixture in tests/golden/) instead of the synthetic one")
    ap.add_argument("--n_rays", type=int, default=10)
    ap.add_argument("--roulette", type=int, default=3)
    ap.add_argument("--max-depth", dest="max_depth", type=int, default=5)
    ap.add_argument("--cpu-width", dest="cpu_width", type=int, default=96)
    ap.add_argument("--cpu-height", dest="cpu_height", type=int, default=54)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config sub-records (C1-C5) of the N=1 line")
    ap.add_argument("--bvh", action="store_true",
                    help="RT_FLAG_ACCEL_BVH: not the reference's algorithmic work, so the line carries no roofline fraction")
    ap.add_argument("--valu-only", action="store_true", help="RT_FLAG_VALU_ONLY: the large-world sphere filter on the vector ALU (no MFMA instruction)")
    ap.add_argument("--save-pfm", default=None, help="write the last frame here (rank 0)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on fewer GPUs than ranks (ranks share devices, gather staged through the host)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import gather_rows, max_local_rows, rank_rows
    from raytracer_amd.synth import synthetic_scene
    from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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
    mode = RenderMode("path", n_rays=args.n_rays, roulette=args.roulette, max_depth=args.max_depth, seq_number=54, bvh=args.bvh, valu_only=args.valu_only)
    rows = rank_rows(args.height, rank, world_size)
    local = torch.empty((rows.n_local_rows, args.width, 3), dtype=torch.float32, device=device)
    full = torch.empty((args.height, args.width, 3), dtype=torch.float32, device=device) if rank == 0 else None
    padded = parts = None
    gather_dev = torch.device("cpu") if staged else device
    full_g = torch.empty((args.height, args.width, 3), dtype=torch.float32) if (staged and rank == 0) else full
    if world_size > 1:
        padded = torch.empty((max_local_rows(args.height, world_size), args.width, 3), dtype=torch.float32, device=gather_dev)
        parts = [torch.empty_like(padded) for _ in range(world_size)] if rank == 0 else None

    def gather():
        gather_rows(local.cpu() if staged else local, args.height, full_g, padded=padded, parts=parts)
        if staged and rank == 0:
            full.copy_(full_g)

    def barrier():
        if world_size > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"scene uploaded: {scene.count('n_objects')} objects; frame {args.width}x{args.height} AA={args.aa}")
    # Warm-up.  The first pass also reads the launch's statistics: the ray count of a frame is deterministic (streams
    # depend only on the sample index), so it holds for every timed step.
    tracer.fire_all_rays(world, mode, rows, out=local, want_stats=True)
    gather()
    st = tracer.last_stats
    rays_local, samples_local = int(st.rays), int(st.samples)
    kernel_name = st.kernel.decode() if rows.n_local_rows else ""
    log(f"stats pass: {rays_local} rays, {samples_local} samples, kernel {st.kernel_ms:.1f} ms, {kernel_name} grid {st.grid} x {st.block}")
    for _ in range(max(args.warmup - 1, 0)):
        tracer.fire_all_rays(world, mode, rows, out=local, want_stats=False)
        gather()
    torch.cuda.synchronize()
    barrier()

    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()  # torch's current stream == the stream rt_render launches on
        tracer.fire_all_rays(world, mode, rows, out=local, want_stats=False)
        ev[k][1].record()
        gather()
        ev[k][2].record()  # after the gather: torch makes the current stream wait for the collective
        if args.steps > 1:
            torch.cuda.current_stream().synchronize()
            log(f"step {k + 1}/{args.steps} done")
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / args.steps
    gather_ms = sum(b.elapsed_time(c) for _, b, c in ev) / args.steps  # on rank 0 this includes waiting for the slowest rank

    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=gather_dev)
    r = torch.tensor([rays_local, samples_local], dtype=torch.int64, device=gather_dev)
    g = torch.tensor([gather_ms], dtype=torch.float64, device=gather_dev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(g, op=dist.ReduceOp.MIN)  # the rank that finished rendering last waits for nobody: the transfer itself
    elapsed, kernel_ms_max = float(t[0]), float(t[1])
    rays_total, samples_total = int(r[0]), int(r[1])

    if rank == 0:
        census = scene_census(scene)
        out = {
            "metric": "path-mode Mrays/s (frame ms in ms_per_step), 1920x1080 AA=4",
            "value": rays_total * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"samples/{args.scene} (reference sample scene)" if args.scene else
                                    f"synthetic {args.spheres}-random-sphere + floor + sky scene") + f", --mode path, "
                                   f"{args.width}x{args.height}, AA={args.aa}, n_rays={args.n_rays}, roulette={args.roulette}, "
                                   f"max-depth={args.max_depth}, seq-number=54" + (", BVH mode (RT_FLAG_ACCEL_BVH)" if args.bvh else ""),
                       "objects": scene.count("n_objects"), "top_level": census, "samples_per_frame": samples_total,
                       "rays_per_frame": rays_total, "rays_per_sample": rays_total / max(samples_total, 1),
                       "parallelism": f"rows-cyclic-x{world_size}" if world_size > 1 else "single"},
            "kernel_ms_per_step_max_rank": kernel_ms_max,
            "gather_ms_per_step": {"rank0_including_wait_for_slowest_rank": gather_ms, "min_over_ranks": float(g[0])},
            # dominant kernel on rank 0's share: ITS rays / ITS duration (k_resolve rides in the same interval, < 0.1 %)
            "roofline": roofline_block(census, kernel_name, rays_local, kernel_ms, pmc_traffic(args) if world_size == 1 else None, bvh=args.bvh),
        }
        import hashlib
        # the assembled frame is bit-identical for any number of ranks: compare this across --gpus runs
        out["frame_sha256"] = hashlib.sha256(full.cpu().numpy().tobytes()).hexdigest()
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
