#!/usr/bin/env python3
"""Throughput of the render path on MI355X: path-samples per second.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts itself N times through
torch.distributed.run (one rank per GPU, RCCL) before anything touches a GPU; a launcher that
already set RANK / WORLD_SIZE is used as is.

Workload (BASELINE.json configs[2], the scene of the north star): cornell.sdf with the
buster_sword.obj triangle mesh (16 008 triangles) at 1920x1080, aa_steps = 32 (1024 samples per
pixel on the reference's regular sub-pixel grid), max_ray_bounces = 5.  One step = one complete
Renderer::render of that image: every rank renders ALL pixels for its slice of the 1024
per-pixel samples into its own fp32 accumulation buffer, the buffers are summed onto rank 0
with one RCCL reduce, rank 0 resolves (average, tone map, 8-bit rows).  The scene is resident in
HBM before the timed region starts; the image stays on the device.  configs[1] (example.sdf,
same size) is rendered once after the timed steps and reported under "also".

The JSON line also carries
  roofline      for the dominant kernel = the kernel symbol with the largest summed device time
                inside the timed steps (HIP events recorded around every launch on the launch
                stream): `achieved` = its ALGORITHMIC bytes / its time, against 8 TB/s.  The
                algorithmic bytes follow the per-ray model (ray record 48 B: written by the
                kernel that emits it, read by the closest-hit kernel, read by the shading kernel;
                hit record 16 B written by the closest-hit kernel, read by the shading kernel;
                24 B accumulator update per shaded hit or lit shadow ray), each kernel credited
                only with the records it really moves: the depth-0 closest-hit kernel generates
                its rays in registers and gets the hit records alone.  `traffic` = HBM bytes
                per launch of that kernel from the committed FETCH_SIZE / WRITE_SIZE counter
                passes (profiles/*_hbm_traffic.json) when they were taken on this workload,
                else null.  `valu` = the same kernel against the vector fp32 peak, with flops
                = 30 per slab test + 52 per triangle test + 37 per sphere test + 30 per box
                test + 33 per ray transform (+ 140 per shaded hit in the shading kernels), the
                tests counted by the kernels' counting twins in one extra, untimed step.
                `kernels` lists every kernel symbol the same way;
  cpu_baseline  the CPU oracle (oracle/cpu_ref.cpp, the bit-exact restatement of the reference
                renderer) timed on this box's host cores on a bounded sample of the same
                workload (same scene and resolution, fewer samples per pixel); when the real
                reference (oracle/_ref/libref.so, its unmodified sources compiled in the build
                container) came along, it is timed the same way and is the figure reported
                (kind "reference", with the restatement's figure beside it as port_value).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E
VALU_PEAK_TFLOPS = 157.3    # vector fp32, FMA counted as 2
RAY_B, HIT_B, ACC_B = 48, 16, 24   # the per-ray model: ray record, hit record, accumulator read-modify-write
BYTES_PER_RAY = 3 * RAY_B + 2 * HIT_B
FLOPS = {"slab": 30, "tri": 52, "sphere": 37, "box": 30, "xform": 33, "shade": 140}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell_buster.sdf")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=32)
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--queue-gib", type=float, default=64.0, help="ray queue memory per GPU (the scene stays resident over all steps: 64 GiB is where larger queues stop paying); 0 = the library's default, also 64 GiB")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration; 0 skips it")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernel launches with HIP events")
    ap.add_argument("--no-census", action="store_true", help="skip the extra, untimed step that counts the tests")
    ap.add_argument("--no-also", action="store_true", help="skip the single untimed-in-value render of configs[1]")
    ap.add_argument("--pipeline", choices=["default", "fused", "staged"], default="default")
    ap.add_argument("--lib", default=None, help="another build of libptx.so to measure")
    ap.add_argument("--save-ppm", default=None)
    ap.add_argument("--verify", action="store_true", help="rank 0 also renders all samples alone and reports the max abs difference")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script (before any GPU call) and pass their output on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(args, scene_path):
    """Oracle timed on the host cores: same scene and resolution, aa chosen for ~cpu_seconds."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_lib
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, os.cpu_count() or 1))
    w, h, nb = args.width, args.height, args.bounces

    def timed(lib, seconds, **kw):
        """about `seconds` of rendering: full resolution at reduced aa_steps when the cores manage
        that, otherwise the same view at a reduced resolution and aa_steps = 1"""
        s = lib.load(scene_path)
        pw, ph = max(16, w // 4), max(16, h // 4)
        probe = lib.render(s, pw, ph, 1, nb, **kw)
        target = seconds * pw * ph / max(probe["seconds"], 1e-6)   # samples that fit the time
        if target >= w * h:
            rw, rh, aa = w, h, max(1, min(args.aa, int((target / (w * h)) ** 0.5)))
        else:
            f = max((target / (w * h)) ** 0.5, 0.05)
            rw, rh, aa = max(16, int(w * f)), max(16, int(h * f)), 1
        r = lib.render(s, rw, rh, aa, nb, **kw)
        return rw * rh * aa * aa / r["seconds"] / 1e6, f"{rw}x{rh}, aa_steps={aa}", r["seconds"]

    v, what, sec = timed(oracle_lib.Port(), args.cpu_seconds, threads=threads)
    port = {"value": v, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{args.scene} {what} (of {w}x{h}, aa_steps={args.aa}), {nb} bounces, {sec:.1f} s"}
    if not oracle_lib.Reference.available():
        return port
    # The real reference came along as a binary: it is the baseline, the restatement is quoted beside it.
    # It prints its own timing line to the C stdout; keep this script's stdout to the one JSON line.
    import ctypes
    sys.stdout.flush()
    saved, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        rv, rwhat, rsec = timed(oracle_lib.Reference(), args.cpu_seconds)
        err = None
    except Exception as e:  # the baseline must never take the bench line down
        err = str(e)[:200]
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        os.close(null)
    if err is not None:
        port["reference_error"] = err
        return port
    # the reference spawns hardware_concurrency() threads (framework/renderer.cpp:39)
    return {"value": rv, "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "reference",
            "sample": f"{args.scene} {rwhat} (of {w}x{h}, aa_steps={args.aa}), {nb} bounces, {rsec:.1f} s incl. its ASCII PPM write",
            "cpu_share": os.environ.get("OMP_NUM_THREADS"), "port_value": port["value"], "port_cores": threads, "port_sample": port["sample"]}


def measured_counters(args, queue_gib, kernel):
    """Per-launch counter figures of `kernel` from the committed counter passes (the newest
    profiles/*_hbm_traffic.json, written by scripts/pmc_summary.py --traffic), if they were taken
    on this workload: same scene, size, spp, bounces and queue memory, hence the same launches.
    Returns {'hbm_bytes': .., 'valu_active_inst': .., 'wave_time_waiting': ..} (a key is missing
    when the record lacks that pass) or None."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_hbm_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        w = t.get("_workload", {})
        if [w.get(k) for k in ("scene", "width", "height", "aa", "bounces")] != [args.scene, args.width, args.height, args.aa, args.bounces]:
            continue
        if abs(float(w.get("queue_gib", -1)) - queue_gib) > 0.5:
            continue
        # `kernel` is a name like k_shade<false>; the symbols carry more template arguments
        # (k_shade<false, true>), of which one variant runs on a given scene
        stem = kernel[:-1] if kernel.endswith(">") else kernel
        rows = [v for name, v in t.items() if (name == kernel or name.startswith(stem + ",") or (not kernel.endswith(">") and name.startswith(stem + "<"))) and "TestCount" not in name
                and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
        if rows:
            n = sum(r["launches"] for r in rows)
            out = {"hbm_bytes": sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n, "record": os.path.basename(path)}
            if all("valu_active_inst_per_launch" in r for r in rows):
                out["valu_active_inst"] = sum(r["valu_active_inst_per_launch"] * r["launches"] for r in rows) / n
            if all("wave_time_waiting" in r for r in rows):
                out["wave_time_waiting"] = sum(r["wave_time_waiting"] * r["launches"] for r in rows) / n
            return out
    return None


HBM_COPY_GBS = 6300.0      # what a streaming copy reaches on this chip (the microarchitecture guide's measured rate)
SIMDS, CLOCK_HZ = 1024, 2.4e9


def classify(avg_launch_s, counters):
    """What bounds the dominant kernel, from its launch time (measured here) and the committed counters:
    'issue' when the vector ALUs are busy most of the time (a SIMD of gfx950 takes a 64-wide vector instruction in two
    cycles - 32 lanes a cycle - so its peak is one instruction per two cycles, which needs two waves or more: one wave
    issues one every four),
    'hbm' when its HBM traffic runs at most of the rate a copy reaches, else 'latency'.  Without counters
    for this workload: None."""
    if not counters:
        return None, None, None
    valu_busy = None
    if "valu_active_inst" in counters:
        valu_busy = counters["valu_active_inst"] * 2 / (avg_launch_s * CLOCK_HZ * SIMDS)
    traffic_gbs = counters["hbm_bytes"] / avg_launch_s / 1e9
    if valu_busy is not None and valu_busy >= 0.6:
        bound = "issue"
    elif traffic_gbs >= 0.6 * HBM_COPY_GBS:
        bound = "hbm"
    else:
        bound = "latency"
    return bound, valu_busy, traffic_gbs


def kernel_table(stats, census, nb, staged):
    """Per kernel symbol: device time and launches inside the last timed step, algorithmic bytes
    by the per-ray model, flops from the census counts (None without a census)."""
    kms, kln = stats["kernel_ms"], stats["kernel_launches"]
    rays, hits, shadow = stats["per_depth_rays"], stats["per_depth_hits"], stats["per_depth_shadow"]
    D = range(nb + 1)

    def flops(tests, depths):
        if census is None:
            return None
        t = [sum(census[tests][d][k] for d in depths) for k in range(5)]
        return FLOPS["slab"] * t[0] + FLOPS["tri"] * t[1] + FLOPS["sphere"] * t[2] + FLOPS["box"] * t[3] + FLOPS["xform"] * t[4]

    deep = [d for d in D if d >= 1]
    out = {}
    if not staged:
        for name, ds in (("k_bounce<true>", [0]), ("k_bounce<false>", deep)):
            n = sum(kln[0][d] for d in ds)
            if n:
                out[name] = {"ms": sum(kms[0][d] for d in ds), "launches": n,
                             "bytes": sum(BYTES_PER_RAY * (rays[d] + shadow[d]) + ACC_B * hits[d] for d in ds) - (RAY_B * 2 * rays[0] if ds == [0] else 0),
                             "flops": None}
        return out
    children = [rays[d + 1] if d + 1 <= nb else 0 for d in D]   # rays depth d's shading emitted
    rows = (
        # closest hit: the ray read from its queue (not at depth 0: generated in registers) + the hit record written
        ("k_trace<true>", 0, [0], lambda d: HIT_B * hits[d], "trace_tests", 0),
        ("k_trace<false>", 0, deep, lambda d: RAY_B * rays[d] + HIT_B * hits[d], "trace_tests", 0),
        # shading: hit record read, the ray read (not at depth 0: regenerated), accumulator, child and shadow rays written
        ("k_shade<true>", 1, [0], lambda d: (HIT_B + ACC_B) * hits[d] + RAY_B * (children[d] + shadow[d]), None, FLOPS["shade"]),
        ("k_shade<false>", 1, deep, lambda d: (HIT_B + ACC_B + RAY_B) * hits[d] + RAY_B * (children[d] + shadow[d]), None, FLOPS["shade"]),
        # shadow rays: the record read, the accumulator updated (counted for every ray: the model does not know which are lit)
        ("k_shadow", 2, list(D), lambda d: (RAY_B + ACC_B) * shadow[d], "shadow_tests", 0),
    )
    for name, kind, ds, bytes_of, tests, per_hit in rows:
        n = sum(kln[kind][d] for d in ds)
        if not n:
            continue
        f = flops(tests, ds) if tests else (per_hit * sum(hits[d] for d in ds) if census is not None else None)
        out[name] = {"ms": sum(kms[kind][d] for d in ds), "launches": n, "bytes": sum(bytes_of(d) for d in ds), "flops": f}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    pt = importlib.import_module("path-tracer_amd")
    ptdist = importlib.import_module("path-tracer_amd.dist")
    if args.lib:
        pt.set_library(args.lib)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PTX_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the multi-rank code
    # on a one-GPU box (RCCL refuses two ranks on one device); the real runs use nccl = RCCL.
    backend = os.environ.get("PTX_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    device_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)

    scene_path = os.path.join(pt.SDF_DIR, args.scene)
    host = pt.HostScene(scene_path)
    free_before = torch.cuda.mem_get_info()[0]
    ranks_here = max(1, world // max(torch.cuda.device_count(), 1)) if backend != "nccl" else 1
    queue_bytes = int(args.queue_gib * (1 << 30))
    if ranks_here > 1 and queue_bytes == 0:
        queue_bytes = free_before // (4 * ranks_here)   # rehearsal: several ranks share this card
    ds = pt.DeviceScene(host, device_index, queue_bytes)
    queue_gib = round((free_before - torch.cuda.mem_get_info()[0]) / (1 << 30), 1)   # what the library took
    profile = not args.no_profile
    ds.set_profiling(1 if profile else 0)

    w, h, aa, nb = args.width, args.height, args.aa, args.bounces
    spp = aa * aa
    s0, s1 = ptdist.sample_range(rank, world, spp)
    rp = pt.render_params(w, h, aa, nb, s0, s1, args.pipeline)
    accum = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    color = torch.empty(w * h * 3, dtype=torch.float32, device=dev) if rank == 0 else None
    rgb = torch.empty(w * h * 3, dtype=torch.uint8, device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        accum.zero_()
        st = ds.render_accumulate(rp, accum.data_ptr(), stream, check_overflow=True, want_stats=True)
        ptdist.reduce_accum(accum, 0)
        if rank == 0:
            ds.resolve(accum.data_ptr(), w, h, spp, color.data_ptr(), rgb.data_ptr(), stream)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        stats = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples_per_step = w * h * spp
        value = samples_per_step * args.steps / elapsed / 1e6
        out = {
            "metric": "path samples/sec", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (scene files shipped with the reference; regular sample grid, no RNG)",
            "config": {"workload": f"{args.scene} {w}x{h}, aa_steps={aa} ({spp} spp), max_ray_bounces={nb}", "queue_gib": queue_gib,
                       "samples_per_step": samples_per_step, "parallelism": f"sample-range x{world}, one RCCL sum-reduce per image"},
        }
        # ---- work done by this rank in the last step (the kernels count it themselves)
        R = (stats["trace_rays"] + stats["shadow_rays"])
        H = stats["shaded_hits"]
        my_samples = stats["samples"]
        out["work"] = {"rays_per_sample": R / my_samples, "shaded_hits_per_sample": H / my_samples,
                       "trace_rays_per_sample": stats["trace_rays"] / my_samples, "shadow_rays_per_sample": stats["shadow_rays"] / my_samples,
                       "algorithmic_bytes_per_sample": (BYTES_PER_RAY * R + ACC_B * H) / my_samples,
                       "Mrays/s": R * world * args.steps / elapsed / 1e6 if world == 1 else None,
                       "kernel_launches_per_step": stats["launches"]}
        kln = stats["kernel_launches"]
        if profile and kln[0][0]:
            staged = kln[1][0] > 0
            census = None
            if not args.no_census and staged:
                # one more step, untimed, with the counting twins of the traversal kernels
                ds.set_profiling(2)
                c = ds.render_accumulate(rp, torch.zeros_like(accum).data_ptr(), stream, check_overflow=True, want_stats=True)
                ds.set_profiling(1)
                census = {"trace_tests": c["trace_tests"], "shadow_tests": c["shadow_tests"]}
                tr = [sum(c["trace_tests"][d][k] for d in range(nb + 1)) for k in range(5)]
                sh = [sum(c["shadow_tests"][d][k] for d in range(nb + 1)) for k in range(5)]
                names = ("slab", "triangle", "sphere", "box", "transform")
                out["work"]["tests_per_trace_ray"] = {n: v / max(c["trace_rays"], 1) for n, v in zip(names, tr)}
                out["work"]["tests_per_shadow_ray"] = {n: v / max(c["shadow_rays"], 1) for n, v in zip(names, sh)}
            table = kernel_table(stats, census, nb, staged)
            tot_ms = sum(k["ms"] for k in table.values())
            kernels = {}
            for name, k in table.items():
                sec = k["ms"] * 1e-3
                gbs = k["bytes"] / sec / 1e9
                row = {"ms_per_step": round(k["ms"], 3), "share": round(k["ms"] / tot_ms, 4), "launches": k["launches"],
                       "algorithmic_GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
                if k["flops"] is not None:
                    row["TFLOPs"] = round(k["flops"] / sec / 1e12, 2)
                    row["valu_frac"] = round(k["flops"] / sec / 1e12 / VALU_PEAK_TFLOPS, 4)
                kernels[name] = row
            dom = max(table, key=lambda n: table[n]["ms"])
            k = table[dom]
            achieved = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            symbol = {"k_trace<true>": "k_trace<true, %s, NoCount>", "k_trace<false>": "k_trace<false, %s, NoCount>",
                      "k_shadow": "k_shadow<%s, NoCount>"}.get(dom)
            avg_s = k["ms"] * 1e-3 / k["launches"]
            counters = measured_counters(args, queue_gib, dom)
            bound, valu_busy, traffic_gbs = classify(avg_s, counters)
            # `achieved` / `frac`: the per-ray model's bytes against the HBM peak (the contract's figure); `traffic`:
            # the bytes the memory really moved per launch (counter passes), with the same launch time as
            # traffic_frac (of 8 TB/s) and achievable_frac (of the 6.3 TB/s a copy reaches); `bound` says which
            # resource the counters show the kernel waiting for - "issue" = vector instruction issue, in which
            # case the HBM fractions describe how little of the memory system the kernel needs, not a shortfall
            out["roofline"] = {"bound": bound if bound else "hbm (by the model: no counter record for this workload)", "kernel": dom,
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": counters["hbm_bytes"] if counters else None,
                               "traffic_frac": traffic_gbs / HBM_PEAK_GBS if counters else None,
                               "achievable_frac": traffic_gbs / HBM_COPY_GBS if counters else None,
                               "traffic_over_algorithmic": counters["hbm_bytes"] / (k["bytes"] / k["launches"]) if counters else None,
                               "valu_busy": valu_busy, "wave_time_waiting": counters.get("wave_time_waiting") if counters else None,
                               "counter_record": counters.get("record") if counters else None,
                               "launches": k["launches"], "avg_launch_ms": k["ms"] / k["launches"],
                               "algorithmic_bytes_per_launch": k["bytes"] / k["launches"],
                               "valu": None if k["flops"] is None else {
                                   "achieved": k["flops"] / (k["ms"] * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": k["flops"] / (k["ms"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS},
                               "symbol_pattern": symbol}
            out["kernels"] = kernels
            kms = stats["kernel_ms"]
            out["kernel_time"] = {"pipeline": "staged" if staged else "fused",
                                  "closest_hit_or_fused_ms_per_depth": [round(x, 3) for x in kms[0][: nb + 1]],
                                  "shade_ms_per_depth": [round(x, 3) for x in kms[1][: nb + 1]],
                                  "shadow_ms_per_depth": [round(x, 3) for x in kms[2][: nb + 1]],
                                  "per_depth_rays": stats["per_depth_rays"][: nb + 1],
                                  "per_depth_hits": stats["per_depth_hits"][: nb + 1],
                                  "per_depth_shadow_rays": stats["per_depth_shadow"][: nb + 1],
                                  "all_launches_ms": tot_ms,
                                  "all_launches_algorithmic_GBs": (BYTES_PER_RAY * R + ACC_B * H) / (tot_ms * 1e-3) / 1e9}
        else:
            out["roofline"] = None
        if args.verify:
            # the summed image of all ranks against one rank rendering every sample alone
            acc1 = torch.zeros_like(accum)
            ds.render_accumulate(pt.render_params(w, h, aa, nb, pipeline=args.pipeline), acc1.data_ptr(), stream, check_overflow=True)
            col1 = torch.empty_like(color)
            ds.resolve(acc1.data_ptr(), w, h, spp, col1.data_ptr(), None, stream)
            torch.cuda.synchronize()
            out["verify_max_abs_vs_single_rank"] = float((col1 - color).abs().max().item())
        if args.save_ppm:
            pt.write_ppm(args.save_ppm, rgb.cpu().numpy().reshape(h, w, 3))
        if world == 1 and not args.no_also and args.scene == "cornell_buster.sdf":
            # one image each, through the same calls, of BASELINE configs[1] at its full size (example.sdf) and of the
            # bench scene seen from close to the mesh (buster_closeup.sdf: the sword fills 70 % of the frame, so the
            # primary closest-hit kernel is bound by the mesh walks, which the bench view barely touches)
            ds.close()
            ds = None
            out["also"] = {}
            for other in ("example.sdf", "buster_closeup.sdf"):
                ds2 = pt.DeviceScene(pt.HostScene(os.path.join(pt.SDF_DIR, other)), device_index, queue_bytes)
                ds2.set_profiling(1)
                rp2 = pt.render_params(w, h, aa, nb)
                for k in range(2):
                    accum.zero_()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    st2 = ds2.render_accumulate(rp2, accum.data_ptr(), stream, check_overflow=True, want_stats=True)
                    ds2.resolve(accum.data_ptr(), w, h, spp, color.data_ptr(), rgb.data_ptr(), stream)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t1
                row = {"Msamples/s": samples_per_step / dt / 1e6, "ms_per_step": dt * 1e3, "rays_per_sample": (st2["trace_rays"] + st2["shadow_rays"]) / st2["samples"],
                       "kernel_ms": {"k_trace<true>": round(st2["kernel_ms"][0][0], 2), "k_trace<false>": round(sum(st2["kernel_ms"][0][1: nb + 1]), 2),
                                     "k_shade": round(sum(st2["kernel_ms"][1][: nb + 1]), 2), "k_shadow": round(sum(st2["kernel_ms"][2][: nb + 1]), 2)}}
                if other == "buster_closeup.sdf" and not args.no_census:
                    ds2.set_profiling(2)
                    accum.zero_()
                    c = ds2.render_accumulate(rp2, accum.data_ptr(), stream, check_overflow=True, want_stats=True)
                    names = ("slab", "triangle", "sphere", "box", "transform")
                    r0 = max(c["per_depth_rays"][0], 1)
                    row["tests_per_primary_ray"] = {n: c["trace_tests"][0][k] / r0 for k, n in enumerate(names)}
                    # of a primary ray's slab tests, 10 are the room (its bounds, 8 leaf filters, the mesh's bounds): the rest are BVH nodes, two per visit
                    row["bvh_node_visits_per_primary_ray"] = max(0.0, (c["trace_tests"][0][0] / r0 - 10.0) / 2.0)
                out["also"][f"{other} {w}x{h}, aa_steps={aa}, max_ray_bounces={nb}"] = row
                ds2.close()
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, scene_path)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
