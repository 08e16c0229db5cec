#!/usr/bin/env python3
"""Per-depth test counts per ray of a library build: scripts/census_depth.py <lib | -> [scene] [aa]
With a library built with -DPTX_CENSUS_NEEDED (hipcc ... -DPTX_CENSUS_NEEDED -shared -o build/libptx_needed.so ptx.hip
ptx_bvh.cpp -ldl) the last column is the frame-0 leaf tests a lane's own ray asked for; sphere + box + tri are what
the waves executed per ray."""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
pt = importlib.import_module("path-tracer_amd")
lib = sys.argv[1]
scene = sys.argv[2] if len(sys.argv) > 2 else "cornell_buster.sdf"
aa = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if lib != "-":
    pt.set_library(lib)
hs = pt.HostScene(os.path.join(pt.SDF_DIR, scene))
ds = pt.DeviceScene(hs, 0, 8 << 30)
ds.set_profiling(2)
w, h, nb = 1920, 1080, 5
rp = pt.render_params(w, h, aa, nb)
acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
c = ds.render_accumulate(rp, acc.data_ptr(), torch.cuda.current_stream().cuda_stream, check_overflow=True, want_stats=True)
names = ("slab", "tri", "sphere", "box", "xform/needed")
for d in range(nb + 1):
    r, s = max(c["per_depth_rays"][d], 1), max(c["per_depth_shadow"][d], 1)
    print(f"depth {d}: rays {r:>11} " + " ".join(f"{n} {c['trace_tests'][d][k] / r:6.2f}" for k, n in enumerate(names)) +
          f" | shadow {s:>11} " + " ".join(f"{n} {c['shadow_tests'][d][k] / s:6.2f}" for k, n in enumerate(names)))
