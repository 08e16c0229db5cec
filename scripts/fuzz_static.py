"""Generated scenes (tests/fuzz_scenes.py) through ptx_render_accumulate_static against
ptx_render_accumulate: equal work counters per depth, the same resolved image (sums near the pole of
the tone mapping compared before it).  usage: fuzz_static.py FIRST COUNT"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # noqa: E402  (its HIP runtime first, as in tests/conftest.py)
from fuzz_scenes import write_scene  # noqa: E402

pt = importlib.import_module("path-tracer_amd")
KEYS = ("samples", "trace_rays", "shadow_rays", "shaded_hits", "per_depth_rays", "per_depth_shadow", "per_depth_hits")
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    sdf, obj_dir = write_scene(seed, tempfile.mkdtemp(), pt.SDF_DIR)
    ds = pt.DeviceScene(pt.HostScene(sdf, obj_dir), 0, 1 << 30)
    for w, h, aa in ((320, 200, 3), (1280, 720, 4)) if seed % 4 == 0 else ((320, 200, 3),):
        nb = (4, 6, 8)[seed % 3]
        out = []
        for static in (True, False):
            acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
            st = ds.render_accumulate(pt.render_params(w, h, aa, nb), acc.data_ptr(), None, check_overflow=True, want_stats=True, static=static)
            torch.cuda.synchronize()
            out.append((st, acc.cpu().numpy().astype(np.float64) / (aa * aa)))
        (sa, a), (sb, b) = out
        same = all(sa[k] == sb[k] for k in KEYS)
        fa, fb = np.isfinite(a), np.isfinite(b)
        both = fa & fb
        close = np.isclose(a[both], b[both], rtol=1e-5, atol=1e-6)
        if not (same and (fa != fb).sum() <= 8 and close.mean() >= 0.9995):
            bad += 1
            print(f"seed {seed} {w}x{h}: counts equal {same}, finite differ {(fa != fb).sum()}, close {close.mean():.6f}", flush=True)
    ds.close()
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} scenes, {bad} findings", flush=True)
print(f"{count} scenes from {first}: {bad} findings")
sys.exit(1 if bad else 0)
