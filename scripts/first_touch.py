"""First-render cost of a fresh scene by queue size (and by HIP runtime: `notorch` keeps torch,
and with it the HIP runtime of torch's wheel, out of the process)."""
import importlib, sys, time
sys.path.insert(0, '.')
if 'notorch' not in sys.argv:
    import torch
    torch.zeros(1, device='cuda:0')
pt = importlib.import_module('path-tracer_amd')
hs = pt.HostScene(f'{pt.SDF_DIR}/cornell_buster.sdf')
for gib in (0, 32, 8, 2, 0):
    t = time.perf_counter(); ds = pt.DeviceScene(hs, 0, gib << 30); tc = time.perf_counter() - t
    times = []
    for _ in range(3):
        t = time.perf_counter(); r = ds.render_image(1920, 1080, 32, 5); times.append(time.perf_counter() - t)
    t = time.perf_counter(); m = pt.render_image_multi([ds], 1920, 1080, 32, 5); tm = time.perf_counter() - t
    print('queue GiB', gib, 'create %.3f s' % tc, 'renders', ['%.3f' % x for x in times], 'multi %.3f' % tm, 'launches', r['stats']['launches'], flush=True)
    t = time.perf_counter(); ds.close(); print('   destroy %.3f s' % (time.perf_counter() - t), flush=True)
