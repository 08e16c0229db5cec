"""Generated scenes (tests/fuzz_scenes.py) at full image size: the staged pipeline in ample work
memory against the fused one and against both in the smallest work memory - the same rays, shadow
rays and hits per depth, the same image to the bar of the parity tests (pixels near the pole of
the tone mapping compared before it).  No oracle: a consistency sweep of the queue driving under
load.  usage: fuzz_big.py FIRST COUNT [WIDTH HEIGHT AA]"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # noqa: F401,E402  (its HIP runtime first, as in tests/conftest.py)
from fuzz_scenes import write_scene  # noqa: E402

pt = importlib.import_module("path-tracer_amd")
first, count = int(sys.argv[1]), int(sys.argv[2])
W, H, AA = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 4)
bad = 0
for seed in range(first, first + count):
    sdf, obj_dir = write_scene(seed, tempfile.mkdtemp(), pt.SDF_DIR)
    hs = pt.HostScene(sdf, obj_dir)
    nb = (4, 6, 8)[seed % 3]
    big, tiny = pt.DeviceScene(hs, 0, 1 << 32), pt.DeviceScene(hs, 0, 1)
    ref = big.render_image(W, H, AA, nb, pipeline="staged")
    fin = np.isfinite(ref["color"]).all(axis=2)
    rb = ref["color"].astype(np.float64)
    for ds, name, pipeline in ((big, "ample", "fused"), (tiny, "tiny", "staged"), (tiny, "tiny", "fused")):
        if pipeline == "fused" and hs.desc.contents.n_guards:
            continue        # (a split root always takes the staged kernels)
        q = ds.render_image(W, H, AA, nb, pipeline=pipeline)
        same = all(q["stats"][k] == ref["stats"][k] for k in ("trace_rays", "shadow_rays", "shaded_hits", "per_depth_rays"))
        odd = np.isfinite(q["color"]).all(axis=2) != fin
        qa = q["color"].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            close = np.isclose(qa, rb, rtol=0, atol=1e-5).all(axis=2) | ~fin | odd | np.isclose(qa / (1 - qa), rb / (1 - rb), rtol=1e-5, atol=0).all(axis=2)
        ok = same and odd.sum() <= 8 and close.mean() >= 0.997
        if not ok:
            bad += 1
            print(f"seed {seed} {name} {pipeline}: counts equal {same}, finite masks differ in {odd.sum()}, close {close.mean():.5f}", flush=True)
    big.close()
    tiny.close()
    if (seed - first) % 10 == 9:
        print(f"... {seed - first + 1} scenes, {bad} findings", flush=True)
print(f"{count} scenes from {first}: {bad} findings")
sys.exit(1 if bad else 0)
