#!/usr/bin/env python3
"""Scene creation / one image / destruction per queue size in a process that holds nothing but the render library
(no torch: the allocator under hipMalloc / hipFree is the runtime's own).  usage: scripts/create_destroy_timing.py [GiB ...]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pt = importlib.import_module("path-tracer_amd")
hs = pt.HostScene(os.path.join(pt.SDF_DIR, "cornell_buster.sdf"))
for g in [float(x) for x in sys.argv[1:]] or [8, 32, 64, 8, 32, 64]:
    t0 = time.perf_counter()
    ds = pt.DeviceScene(hs, 0, int(g * (1 << 30)))
    t1 = time.perf_counter()
    r = ds.render_image(1920, 1080, 32, 5)
    t2 = time.perf_counter()
    ds.close()
    t3 = time.perf_counter()
    print(f"queue {g:5.1f} GiB: create {1e3 * (t1 - t0):7.1f} ms, image incl. copy-out {1e3 * (t2 - t1):7.1f} ms (device {1e3 * r['seconds']:.1f}), destroy {1e3 * (t3 - t2):7.1f} ms", flush=True)
