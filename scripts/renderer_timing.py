import importlib, sys, time
sys.path.insert(0, '.')
pt = importlib.import_module('path-tracer_amd')
for name in ('example', 'cornell_buster'):
    hs = pt.HostScene(f'{pt.SDF_DIR}/{name}.sdf')
    for keep in (False, True):
        r = pt.HostRenderer(1920, 1080, 32, 5, ppm_path='/tmp/x.ppm', keep_scene=keep)
        for k in range(3):
            t = time.perf_counter(); o = r.render(hs); dt = time.perf_counter() - t
            print(name, 'keep', keep, 'call %.3f s' % dt, 'setup %.3f' % o['setup_seconds'], 'multi %.3f' % o['seconds'], flush=True)
        r.close()
