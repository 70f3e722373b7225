#!/usr/bin/env python3
"""ImagePigment at render level (include/materials.hpp:102-138), made by RUNNING THE UNMODIFIED REFERENCE (oracle/_ref/rt_ref).

A 37x23 texture of seeded random texels is written as a PFM; the reference parses scenes that use it through
`image("<file>")` (include/scenefiles.hpp:685-690) on spheres (uv from atan2 / acos), on a plane (uv = fractional
part), as a BRDF pigment and as an emission pigment:

  image_pigment.scene.json        floor + sky + textured diffuse sphere + mirror + a sphere whose EMISSION is the texture
  image_pigment_many.scene.json   the same floor and sky and 40 small textured / mirror / plain spheres (>= 32 register spheres:
                                  the large-world path kernel shades from the texture)
  image_pigment_clamp_u/v.scene.json   one object and an orthogonal camera whose single ray hits where u == 1.0f / v == 1.0f
                                  exactly, so that int(u * w) == w and the clamp to w - 1 / h - 1 decides the texel
  image_pigment_images.npz        the reference's renders: flat AA=1 / AA=2, pointlight, path with its serial PCG (small exact
                                  pin) and at 36 spp (statistical target; 10/3/5 for the first scene, 3/2/3 for the second, see below), the 1x1 clamp renders with the texel they must equal

Only runnable where /root/reference exists.   python tests/golden/make_image_pigment_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import REF_BIN, read_pfm  # noqa: E402

TMP = tempfile.mkdtemp(prefix="rtimgpig")
TEX_W, TEX_H = 37, 23


def ref(*args):
    return subprocess.run([REF_BIN, *map(str, args)], check=True, capture_output=True, text=True).stdout


def write_pfm(path, img):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img[::-1].astype("<f4").tobytes())  # PFM rows run bottom to top


MATERIALS = """
material floor_material(diffuse(image("{tex}")), uniform(<0, 0, 0>))
material sky_material(diffuse(uniform(<0, 0, 0>)), uniform(<0.2, 0.3, 1>))
material tex_material(diffuse(image("{tex}")), uniform(<0, 0, 0>))
material mirror_material(specular(uniform(<0.6, 0.6, 0.6>)), uniform(<0, 0, 0>))
material lamp_material(diffuse(uniform(<0.2, 0.2, 0.2>)), image("{tex}"))
material plain_material(diffuse(uniform(<0.7, 0.4, 0.1>)), uniform(<0, 0, 0>))
"""


def scene_main(tex):
    return MATERIALS.format(tex=tex) + """
plane(translation([0, 0, -1.5]), floor_material)
sphere(scaling([50, 50, 50]), sky_material)
sphere(translation([0.3, -0.9, -0.6]) * rotation_z(25) * scaling([0.8, 0.8, 0.8]), tex_material)
sphere(translation([0.5, 1.0, -0.7]) * scaling([0.7, 0.7, 0.7]), mirror_material)
sphere(translation([-0.6, 0.1, -1.1]) * rotation_x(40) * scaling([0.35, 0.35, 0.35]), lamp_material)
point_light([-2, 2, 3], <0.9, 0.9, 0.9>, 0)
camera(perspective, translation([-3, 0, 0]), exact_asp_ratio, 1)
"""


def scene_many(tex, rng):
    lines = [MATERIALS.format(tex=tex), "plane(translation([0, 0, -1.5]), floor_material)", "sphere(scaling([50, 50, 50]), sky_material)"]
    mats = ["tex_material", "tex_material", "mirror_material", "plain_material", "lamp_material"]
    for k in range(40):
        x, y, z = rng.uniform(-1.5, 4.0), rng.uniform(-3.0, 3.0), rng.uniform(-1.3, 1.0)
        r = rng.uniform(0.12, 0.3)
        lines.append(f"sphere(translation([{x:.4f}, {y:.4f}, {z:.4f}]) * scaling([{r:.4f}, {r:.4f}, {r:.4f}]), {mats[k % len(mats)]})")
    lines.append("camera(perspective, translation([-3, 0, 0]), exact_asp_ratio, 1)")
    return "\n".join(lines) + "\n"


# One ray (orthogonal camera, 1x1 image: origin (-1, 0, 0) + the camera's translation, direction +x).
# u == 1: a sphere mirrored in x (local point = (-x, y, z)) hit at world (-1, -1e-10, 0): atan2(-1e-10, 1) / 2 pi + 1 rounds to 1.0f.
# v == 1: a plane turned to face the camera, hit at local y = -1e-10: y - floor(y) rounds to 1.0f.
def scene_clamp_u(tex):
    return MATERIALS.format(tex=tex) + """
sphere(scaling([-1, 1, 1]), tex_material)
camera(orthogonal, translation([-2, -0.0000000001, 0]), exact_asp_ratio)
"""


def scene_clamp_v(tex):
    return MATERIALS.format(tex=tex) + """
plane(translation([2, 0, 0]) * rotation_y(90), tex_material)
camera(orthogonal, translation([0, -0.0000000001, 0.3]), exact_asp_ratio)
"""


def main():
    rng = np.random.default_rng(20261111)
    tex = rng.uniform(0.05, 0.95, size=(TEX_H, TEX_W, 3)).astype(np.float32)
    tex_path = os.path.join(TMP, "texture_37x23.pfm")
    write_pfm(tex_path, tex)
    texts = {"image_pigment": scene_main(tex_path), "image_pigment_many": scene_many(tex_path, rng),
             "image_pigment_clamp_u": scene_clamp_u(tex_path), "image_pigment_clamp_v": scene_clamp_v(tex_path)}
    paths = {}
    for name, text in texts.items():
        paths[name] = os.path.join(TMP, name + ".txt")
        with open(paths[name], "w") as f:
            f.write(text)
        ref("flatten", paths[name], os.path.join(HERE, name + ".scene.json"))

    def render(name, mode, w, h, aa=1, **kw):
        pfm = os.path.join(TMP, "o.pfm")
        args = ["render", paths[name], "-o", pfm, "--mode", mode, "--width", w, "--height", h, "--aa", aa]
        for k, v in kw.items():
            args += ["--n_rays" if k == "n_rays" else "--" + k.replace("_", "-"), v]
        info = json.loads(ref(*args))
        return read_pfm(pfm), info

    images = {"texture": tex}
    meta = {}
    for name in ("image_pigment", "image_pigment_many"):
        images[f"{name}_flat_160x120_aa1"], _ = render(name, "flat", 160, 120, 1)
        images[f"{name}_flat_120x90_aa2"], _ = render(name, "flat", 120, 90, 2)
        images[f"{name}_path_48x36_aa1_serial"], _ = render(name, "path", 48, 36, 1, n_rays=3, roulette=2, max_depth=4, seq=9)
        # A hazard of the reference found here: Sphere::calculate_uv takes acos of a local z that rounding can leave a hair
        # outside [-1, 1] (about one hit in 10^8 near a pole): v = NaN, int(NaN * h) = INT_MIN, and ImagePigment::operator()
        # (include/materials.hpp:120-137, which clamps only upwards) indexes pixels[] out of bounds — an assertion failure in
        # this build, undefined behaviour in the reference's NDEBUG release build.  Seed 54 runs into it on the first scene
        # (oracle: u=0.9458 v=nan after 6.5e7 rays).  The oracle and the device kernels define that case as row 0; the
        # reference render is made with the first seed that does not meet it.
        # The 40 small spheres of the second scene meet it within every 10/3/5 render of this size (rounding relative to a
        # small radius is larger): its statistical target is rendered 3/2/3, a hundred times fewer rays.
        kw = dict(n_rays=10, roulette=3, max_depth=5) if name == "image_pigment" else dict(n_rays=3, roulette=2, max_depth=3)
        for seq in range(54, 64):
            try:
                images[f"{name}_path_64x48_aa6_serial"], info = render(name, "path", 64, 48, 6, seq=seq, **kw)
            except subprocess.CalledProcessError as e:
                print(f"{name}: seq {seq}: the reference aborted ({e.stderr.strip().splitlines()[-1] if e.stderr else e})", flush=True)
                continue
            info["seq"] = seq
            info.update(kw)
            meta[f"{name}_path_64x48_aa6_serial"] = info
            break
    images["image_pigment_pointlight_160x120_aa1"], _ = render("image_pigment", "pointlight", 160, 120, 1)
    for name in ("image_pigment_clamp_u", "image_pigment_clamp_v"):
        img, _ = render(name, "flat", 1, 1, 1)
        images[f"{name}_flat_1x1"] = img
        # which texel did the reference return?  (row 0 of the texture = top, as HdrImage stores it after reading the PFM)
        match = np.argwhere((tex == img[0, 0]).all(axis=-1))
        assert len(match) == 1, (name, img, match)
        meta[name] = {"texel_row_col": [int(match[0][0]), int(match[0][1])]}
        print(name, img[0, 0], meta[name])
    # the clamp cases must really sit on the clamp: last column (u == 1) / last row (v == 1)
    assert meta["image_pigment_clamp_u"]["texel_row_col"][1] == TEX_W - 1, meta
    assert meta["image_pigment_clamp_v"]["texel_row_col"][0] == TEX_H - 1, meta
    np.savez_compressed(os.path.join(HERE, "image_pigment_images.npz"), **images)
    with open(os.path.join(HERE, "image_pigment_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
