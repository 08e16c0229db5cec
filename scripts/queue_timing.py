#!/usr/bin/env python3
"""What the size of the work memory costs a one-image caller: creating the scene, rendering one image, destroying
the scene, per queue size.  usage: scripts/queue_timing.py [scene] [aa] [GiB ...]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
pt = importlib.import_module("path-tracer_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_buster.sdf"
aa = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sizes = [float(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 16, 32]
hs = pt.HostScene(os.path.join(pt.SDF_DIR, scene))
w, h, nb = 1920, 1080, 5
acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
rp = pt.render_params(w, h, aa, nb)
for rep in range(2):
    for g in sizes:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds = pt.DeviceScene(hs, 0, int(g * (1 << 30)))
        t1 = time.perf_counter()
        st = ds.render_accumulate(rp, acc.data_ptr(), stream, check_overflow=True, want_stats=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = ds.render_accumulate(rp, acc.data_ptr(), stream, check_overflow=True, want_stats=True)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ds.close()
        t4 = time.perf_counter()
        print(f"{scene} aa {aa} queue {g:5.1f} GiB: create {1e3 * (t1 - t0):7.1f} ms, first image {1e3 * (t2 - t1):7.1f} ms, second image {1e3 * (t3 - t2):7.1f} ms,"
              f" destroy {1e3 * (t4 - t3):7.1f} ms, launches {st['launches']}, chunks redone {st['chunks_redone']}, one-image total {1e3 * (t2 - t0 + t4 - t3):7.1f} ms", flush=True)
