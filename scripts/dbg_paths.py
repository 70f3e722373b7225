"""Walks the first child of every bounce of chosen pixel samples on the oracle and on the device side by side (closest hit,
scatter direction, stream state) and prints both: the tool that showed path-mode divergences between the exact kernel
and the oracle to be mirror self-intersection decisions (tests/test_random_csg.py).

    python scripts/dbg_paths.py          (GPU; edit the scene / pixel list below)
"""
import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import FlatScene
from raytracer_amd.tracer import DeviceWorld, scatter_rays
s = FlatScene.from_json("/root/repo/tests/golden/random_csg_1.scene.json")
world = DeviceWorld(s)
W, H, SEQ = 60, 40, 11
def as_arr(r):
    a = np.zeros(1, dtype=abi.RAY_DTYPE)
    a["origin"][0] = r.origin[:]; a["direction"][0] = r.direction[:]; a["tmin"] = r.tmin; a["tmax"] = r.tmax
    return a
for (row, col) in [(30, 5), (30, 44), (31, 26), (34, 44), (35, 34)]:
    print("=== pixel", row, col)
    ray = as_arr(O.camera_ray(s.camera, W, H, col, row))
    st = O._u64x2() if hasattr(O, "_u64x2") else None
    pcg = O.PCG(0, SEQ)
    O.dll().rto_node_root_stream(pcg.st, SEQ, row * W + col)
    ray_d = ray.copy()
    for depth in range(0, 3):
        ho = O.trace_rays(s, ray)[0]; hd = world.ray_intersection(ray_d)[0]
        same_ray = ray.tobytes() == ray_d.tobytes()
        print(" depth", depth, "same ray", same_ray, "obj", ho["object"], hd["object"], "shape", ho["shape"], hd["shape"], "t", ho["t"], hd["t"],
              "uv", ho["uv"], hd["uv"], "uv equal", np.array_equal(ho["uv"], hd["uv"]))
        if ho["object"] < 0 or hd["object"] < 0:
            break
        m = int(s.shape_material[ho["shape"]])
        print("   material", m, "brdf", int(s.material_brdf[m]), "pig kinds", int(s.pigment_kind[s.material_brdf_pigment[m]]), int(s.pigment_kind[s.material_emission_pigment[m]]),
              "brdf colour", O.pigment(s, int(s.material_brdf_pigment[m]), float(ho["uv"][0]), float(ho["uv"][1])), O.pigment(s, int(s.material_brdf_pigment[m]), float(hd["uv"][0]), float(hd["uv"][1])))
        if depth == 2:
            break
        # child 0
        O.dll().rto_node_child_stream(pcg.st, 0)
        st_in = np.array([[pcg.st[0], pcg.st[1]]], dtype=np.uint64)
        r2 = O.scatter_ray(int(s.material_brdf[m]), pcg, ray["direction"][0], ho["point"], ho["normal"], depth + 1)
        d_dev, st_out = scatter_rays(int(s.material_brdf[m]), st_in, ray_d["direction"][0], hd["normal"])
        ray = as_arr(r2)
        ray_d = ray.copy(); ray_d["origin"][0] = hd["point"]; ray_d["direction"][0] = d_dev[0]
        print("   scatter dir", ray["direction"][0], d_dev[0], "equal", np.array_equal(ray["direction"][0], d_dev[0]), "pcg equal", int(st_out[0, 0]) == pcg.st[0])
world.close()
