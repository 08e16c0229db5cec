"""Exploratory GPU check (not a test): hit parity and image parity against the CPU oracle."""
import importlib, sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle'))
pt = importlib.import_module('path-tracer_amd')
from oracle_lib import Port, random_rays
port = Port()
def beq(a, b): return np.array_equal(a.view(np.uint32), b.view(np.uint32))
for name, w, h, aa in [('cornell.sdf', 240, 160, 2), ('example.sdf', 240, 135, 2), ('example_fixed.sdf', 240, 135, 1), ('cornell_buster.sdf', 160, 90, 1), ('superhot_colored.sdf', 160, 90, 1)]:
    path = f'{pt.SDF_DIR}/{name}'
    hs = pt.HostScene(path); s = port.load(path)
    ds = pt.DeviceScene(hs, 0, 1 << 30)
    print(name, ds.info(), flush=True)
    for label, (o, d) in {'primary': port.primary_rays(s, 240, 160, 1), 'random': random_rays(20000, 7)}.items():
        a = ds.intersect(o, d); b = port.intersect(s, o, d)
        m = b['hit']
        idok = np.array_equal(a['prim'], b['rank'])
        print(f'  {label:8s} n={len(o)} hits={int(m.sum())} id== {idok} (mismatch {(a["prim"]!=b["rank"]).sum()})',
              ' '.join(f'{k} bits== {beq(a[k][m], b[k][m])} (maxabs {np.nanmax(np.abs(a[k][m]-b[k][m])) if m.any() else 0:.3g})' for k in ('t','pos','normal','raydir')), flush=True)
    t0 = time.time(); r = ds.render_image(w, h, aa, 5); t1 = time.time()
    ref = port.render(s, w, h, aa, 5, counters=True)
    diff = r['color'] - ref['color']
    rmse = float(np.sqrt(np.nanmean(diff.astype(np.float64) ** 2)))
    bd = np.abs(r['rgb'].astype(int) - ref['rgb'].astype(int))
    print(f'  render {w}x{h} aa={aa}: gpu {r["seconds"]*1e3:.2f} ms (wall {t1-t0:.3f}s) rmse {rmse:.3g} maxabs {np.nanmax(np.abs(diff)):.3g} bytes differing {(bd>0).sum()} of {bd.size} (max {bd.max()})', flush=True)
    print('  gpu stats', {k: v for k, v in r['stats'].items() if k != 'per_depth_rays'}, 'per depth', r['stats']['per_depth_rays'][:7])
    print('  oracle counters trace/shadow/shaded', ref['counters'][:3])
    ds.close()
