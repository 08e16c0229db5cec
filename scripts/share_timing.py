"""Time of one rank's share of the bench image as the number of ranks grows (strong scaling of the
sample-range split, rehearsed on one GPU): rank 0's range of 1024 spp for world = 1, 2, 4, 8."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("path-tracer_amd")
ptdist = importlib.import_module("path-tracer_amd.dist")
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_buster.sdf"
qg = float(sys.argv[2]) if len(sys.argv) > 2 else 32
host = pt.HostScene(os.path.join(pt.SDF_DIR, scene))
ds = pt.DeviceScene(host, 0, int(qg * (1 << 30)))
w, h, aa, nb = 1920, 1080, 32, 5
accum = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
base = None
for world in (1, 2, 4, 8):
    s0, s1 = ptdist.sample_range(0, world, aa * aa)
    rp = pt.render_params(w, h, aa, nb, s0, s1, "default")
    for rep in range(3):
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ds.render_accumulate(rp, accum.data_ptr(), stream, check_overflow=True, want_stats=False)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    base = base or t
    print(f"{scene} world {world}: samples {s0}..{s1}  {t*1e3:.1f} ms  launches {st['launches'] if st else '-'}  efficiency {base / world / t:.3f}", flush=True)
