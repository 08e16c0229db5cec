#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libref.so, built from the
reference's own sources by oracle/Makefile).  Runs only where the reference checkout was
present at build time; the vectors it writes are data (inputs + the reference's outputs),
committed so that the CPU and GPU tests can run where the reference is absent.

Per scene file <name>.sdf the archive holds
  params            [w, h, aa_steps, max_ray_bounces] of the image vectors
  color             float32 [h, w, 3]   Renderer::color_buffer() (tone-mapped, y up)
  rgb               uint8   [h, w, 3]   the PPM's values (rows flipped as the file stores them)
  ppm_md5           md5 of the PPM file text
  hit_w, hit_h      size of the primary-hit map (aa sample 0)
  names             table of HitPoint::hit_object strings ("null" = miss)
  primary_name      int32 [hit_h*hit_w] index into names
  primary_t         float32 HitPoint::distance
  rays_o, rays_d    float32 [n, 3] seeded incoherent rays
  rays_name / rays_t / rays_pos / rays_normal / rays_raydir   the reference's HitPoint for them
"""
import hashlib
import os
import sys

import numpy as np

from oracle_lib import REPO, SDF_DIR, Reference, random_rays

OUT = os.path.join(REPO, "tests", "golden")

# (scene, image w, h, aa, bounces, hit map w, h)
CASES = [
    ("cornell.sdf", 120, 80, 2, 5, 240, 160),
    ("example.sdf", 128, 72, 2, 5, 240, 135),
    ("example_fixed.sdf", 128, 72, 1, 5, 240, 135),
    ("cornell_buster.sdf", 96, 54, 1, 5, 160, 90),
    ("superhot_colored.sdf", 96, 54, 1, 5, 160, 90),
    ("buster_closeup.sdf", 96, 54, 1, 5, 160, 90),
]
# BASELINE config 1 (cornell.sdf 480x320, 16 spp) is pinned by its md5 and 8-bit image
CONFIG1 = ("cornell.sdf", 480, 320, 4, 5)


def name_table(names):
    table = sorted(set(names) | {"null"})
    index = {n: i for i, n in enumerate(table)}
    return np.array(table), np.array([index[n] for n in names], np.int32)


def main():
    ref = Reference()
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])   # scene files named on the command line: just those (the others' archives stay as they are)
    for scene, w, h, aa, nb, hw, hh in CASES:
        if only and scene not in only:
            continue
        s = ref.load(f"{SDF_DIR}/{scene}")
        r = ref.render(s, w, h, aa, nb)
        o, d = ref.primary_rays(s, hw, hh, 1)
        ph = ref.intersect(s, o, d)
        ro, rd = random_rays(4000, 4242)
        rh = ref.intersect(s, ro, rd)
        all_names = [n if hit else "null" for n, hit in zip(ph["names"], ph["hit"])] + [n if hit else "null" for n, hit in zip(rh["names"], rh["hit"])]
        table, idx = name_table(all_names)
        np.savez_compressed(
            os.path.join(OUT, scene.replace(".sdf", ".npz")),
            params=np.array([w, h, aa, nb], np.int32), color=r["color"], rgb=r["rgb"], ppm_md5=np.array(hashlib.md5(r["ppm_text"]).hexdigest()),
            hit_w=np.int32(hw), hit_h=np.int32(hh), names=table, primary_name=idx[: hw * hh], primary_t=ph["t"],
            rays_o=ro, rays_d=rd, rays_name=idx[hw * hh:], rays_t=rh["t"], rays_pos=rh["pos"], rays_normal=rh["normal"], rays_raydir=rh["raydir"])
        print(scene, "ppm md5", hashlib.md5(r["ppm_text"]).hexdigest(), "primary hits", int(ph["hit"].sum()), "ray hits", int(rh["hit"].sum()))
        ref.free(s)
    if only:
        return
    scene, w, h, aa, nb = CONFIG1
    s = ref.load(f"{SDF_DIR}/{scene}")
    r = ref.render(s, w, h, aa, nb)
    np.savez_compressed(os.path.join(OUT, "config1_cornell_480x320_aa4.npz"), params=np.array([w, h, aa, nb], np.int32), rgb=r["rgb"],
                        ppm_md5=np.array(hashlib.md5(r["ppm_text"]).hexdigest()))
    print("config 1 md5", hashlib.md5(r["ppm_text"]).hexdigest())

    # per-primitive known answers in the style of the reference's source/tests.cpp (:76-96, :112-142,
    # :144-158, :252-259), answered by the reference's own Sphere/Box/Triangle::intersect
    o, d = random_rays(1500, 99, lo=(-6, -6, -6), hi=(6, 6, 6))
    prims = [
        (0, [0, 0, 0, 1.5], []),
        (0, [1, -2, 0.5, 2.0], [[1, 0.5, 0, -1], [3, 1.5, 0.7, 2.0]]),
        (1, [-1, -1, -1, 1, 2, 3], []),
        (1, [0, 0, 0, 2, 2, 2], [[1, 2, 1, -1], [2, 0.6, 0.2, -0.4]]),
        (2, [-2, -3, -1, 1, -3, -1, -0.5, 2, -3], []),
        (2, [-2, -3, -1, 1, -3, -1, -0.5, 2, -3], [[2, 0.3, 0.9, 0.1], [3, 2, 2, 0.5]]),
    ]
    arrays = dict(rays_o=o, rays_d=d, n_prims=np.int32(len(prims)))
    for i, (kind, p, xf) in enumerate(prims):
        r = ref.prim_intersect(kind, p, xf, o, d)
        pp = np.zeros(9, np.float32)
        pp[: len(p)] = p
        arrays[f"kind{i}"] = np.int32(kind)
        arrays[f"p{i}"] = pp
        arrays[f"xf{i}"] = np.array(xf, np.float32).reshape(-1, 4)
        for k in ("hit", "t", "pos", "normal", "raydir"):
            arrays[f"{k}{i}"] = r[k]
    # the hand-checkable cases of source/tests.cpp with the t - EPSILON values the code gives today
    kat_o = np.array([[0, 0, 10], [0, 10, 10], [0, 2, 10]], np.float32)  # sphere r=2 at origin: hit / miss / tangent
    kat_d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    r = ref.prim_intersect(0, [0, 0, 0, 2], [], kat_o, kat_d)
    arrays.update(kat_o=kat_o, kat_d=kat_d, kat_hit=r["hit"], kat_t=r["t"])
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), **arrays)
    print("primitives: sphere KAT hit", r["hit"], "t", r["t"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
