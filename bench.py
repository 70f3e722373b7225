#!/usr/bin/env python3
"""bench.py — path-mode throughput of the MI355X tracing backend on the synthetic N-sphere scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one full frame of BASELINE.json's metric: --mode path, 1920x1080, AA=4 (16 spp), n_rays=10,
roulette=3, max-depth=5, on the synthetic N-random-sphere + floor + sky scene of raytracer_amd/synth.py.
Default N = 512: 3.1e10 rays per frame (927 per sample), about 7 s on one MI355X.  The 4096-sphere scene of
BASELINE configs[4] is the same generator with --spheres 4096; its 1080p frame is 3.9e11 rays x 4098 sphere
tests = 1.6e15 tests, minutes per frame on one GPU, so it is run explicitly (DESIGN.md has the numbers).
With N ranks the frame's rows are dealt round-robin row by row (strong scaling: the frame is fixed), each rank
renders its bands into HBM and rank 0 gathers them over RCCL.  The scene and the framebuffers are resident in
HBM before the timed region.  A ray = one World::ray_intersection query (closest hit over every object).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector FP32 peak: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2.4 GHz (MI355X_MICROARCH.md)


def executed_flops_per_ray(n_spheres_top: int, n_planes_top: int, iso: bool) -> float:
    """fp32 flops the wide kernel actually issues per ray (FMA = 2): the world-space sphere test is
    3 sub + 5 (b) + 6 (c) + 3 (disc) = 17, the object-space form 2*27 - 3 = 51; plus the same epilogue."""
    return (17.0 if iso else 51.0) * n_spheres_top + 35.0 * n_planes_top + 54.0 + 75.0


def flops_per_ray(n_spheres_top: int, n_planes_top: int) -> float:
    """Algorithmic fp32 flops of one closest-hit query + shading, counted on the reference's arithmetic as
    written: 52 per sphere test that misses (33 ray->object frame + 19 up to the discriminant),
    +7 for a sphere that is hit (the sky sphere always is), 35 per plane, 54 for the winner's hit record,
    ~75 for scatter + shading."""
    return 52.0 * n_spheres_top + 7.0 + 35.0 * n_planes_top + 54.0 + 75.0


def pmc_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the committed counter passes of THIS configuration
    (scripts/pmc_run.sh -> profiles/r1/g_n512_pmc.json: FETCH_SIZE and WRITE_SIZE in KB, summed over the run's two
    launches, each in its own rocprofv3 --pmc pass).  FETCH_SIZE is taken as reported (gfx950 tallies 64 B per 128-B
    request for wide streaming reads; the node-stack reads here are dword gathers, uncalibrated).  None for any other
    configuration: counters cannot be read from inside the timed run."""
    default = (not args.bvh and args.spheres == 512 and args.width == 1920 and args.height == 1080 and args.aa == 4 and not args.scene
               and args.n_rays == 10 and args.roulette == 3 and args.max_depth == 5)
    path = os.path.join(ROOT, "profiles", "r1", "g_n512_pmc.json")
    if not default or not os.path.exists(path):
        return None
    with open(path) as f:
        c = json.load(f)["counters"]
    return (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / 2.0


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
    ap.add_argument("--bvh", action="store_true",
                    help="RT_FLAG_ACCEL_BVH: not the reference's algorithmic work, so the line carries no roofline fraction")
    ap.add_argument("--save-pfm", default=None, help="write the last frame here (rank 0)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on fewer GPUs than ranks (ranks share devices, gather staged through the host)")
    args = ap.parse_args()

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
    mode = RenderMode("path", n_rays=args.n_rays, roulette=args.roulette, max_depth=args.max_depth, seq_number=54, bvh=args.bvh)
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

    def step(want_stats=False):
        tracer.fire_all_rays(world, mode, rows, out=local, want_stats=want_stats)
        gather()

    def barrier():
        if world_size > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"scene uploaded: {scene.count('n_objects')} objects; frame {args.width}x{args.height} AA={args.aa}")
    # One stats pass (untimed): the ray count of a frame is deterministic (streams depend only on the sample index)
    step(want_stats=True)
    rays_local = int(tracer.last_stats.rays)
    samples_local = int(tracer.last_stats.samples)
    log(f"stats pass: {rays_local} rays, {samples_local} samples, kernel {tracer.last_stats.kernel_ms:.1f} ms")
    for _ in range(max(args.warmup - 1, 0)):
        step()
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
        import numpy as np
        top_shapes = scene.objects[scene.objects >= 0]
        n_sph_top = int(np.sum(scene.shape_kind[top_shapes] == 0))  # RT_SHAPE_SPHERE; the sky sphere is one of them
        n_pl_top = int(len(top_shapes)) - n_sph_top
        f_ray = flops_per_ray(n_sph_top, n_pl_top)  # CSG trees are not counted: a lower bound for scenes that have them
        # dominant kernel = k_render_path on rank 0's share; achieved = algorithmic flops of ITS rays / ITS duration
        achieved = rays_local * f_ray / (kernel_ms * 1e-3) / 1e12
        executed = rays_local * executed_flops_per_ray(n_sph_top, n_pl_top, True) / (kernel_ms * 1e-3) / 1e12
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
                       "objects": scene.count("n_objects"), "samples_per_frame": samples_total, "rays_per_frame": rays_total,
                       "rays_per_sample": rays_total / max(samples_total, 1),
                       "parallelism": f"rows-cyclic-x{world_size}" if world_size > 1 else "single"},
            "kernel_ms_per_step_max_rank": kernel_ms_max,
            "gather_ms_per_step": {"rank0_including_wait_for_slowest_rank": gather_ms, "min_over_ranks": float(g[0])},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_VALU_PEAK_TFLOPS, "traffic": pmc_traffic(args),
                         "kernel": "rtd::k_render_path_wide<ISO,K,W> (+ k_resolve, <0.1 %)", "flops_per_ray": f_ray,
                         "rays_per_launch": rays_local, "kernel_ms": kernel_ms,
                         "executed_tflops": executed, "executed_frac": executed / FP32_VALU_PEAK_TFLOPS,
                         "note": "achieved = reference-as-written flops (52 per sphere test); the kernel issues 17 per test "
                                 "(world-space form, FMA), reported as executed_*"},
        }
        if args.bvh:  # the hierarchy skips most of the tests the brute-force flop count assumes: no fraction of a roofline
            out["roofline"] = {"bound": "valu", "achieved": None, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                               "traffic": None, "kernel": "rtd::k_render_path_bvh", "rays_per_launch": rays_local, "kernel_ms": kernel_ms,
                               "note": "BVH mode changes the algorithmic work; see DESIGN.md 4.6 for its step counts"}
        import hashlib
        # the assembled frame is bit-identical for any number of ranks: compare this across --gpus runs
        out["frame_sha256"] = hashlib.sha256(full.cpu().numpy().tobytes()).hexdigest()
        if world_size == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline on a bounded sample ...")
            out["cpu_baseline"] = cpu_baseline(scene, text, args)
        if args.save_pfm:
            img = full.cpu().numpy()
            with open(args.save_pfm, "wb") as f:
                f.write(b"PF\n%d %d\n-1.0\n" % (args.width, args.height))
                f.write(img[::-1].astype("<f4").tobytes())
        print(json.dumps(out), flush=True)
    world.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
