#!/usr/bin/env python3
"""Throughput of the render path on MI355X: path-samples per second.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): sdf/example.sdf at 1920x1080, aa_steps = 32 (1024 samples
per pixel on the reference's regular sub-pixel grid), max_ray_bounces = 5.  One step = one
complete Renderer::render of that image: every rank renders ALL pixels for its slice of the
1024 per-pixel samples into its own fp32 accumulation buffer, the buffers are summed onto
rank 0 with one RCCL reduce, rank 0 resolves (average, tone map, 8-bit rows).  The scene is
resident in HBM before the timed region starts; the image stays on the device.

The JSON line also carries
  roofline      the dominant kernel - the one with the largest share of the step's time: the
                depth-0 closest-hit kernel k_trace<true> of the staged pipeline (or the depth-0
                k_bounce<true> with --pipeline fused): algorithmic bytes per launch / its
                average launch duration from HIP events recorded on the launch stream during
                the timed steps, against 8 TB/s.  Algorithmic bytes follow the per-ray model
                (ray record 48 B written once, read by intersect, read by shade; hit record
                16 B written, read; 24 B accumulator update per shaded hit): the closest-hit
                kernel's share is 48 + 16 B per ray it handles, the fused kernel's 176 B per
                ray segment + 24 B per hit.  `traffic` is the kernel's HBM bytes per launch
                from the FETCH_SIZE / WRITE_SIZE counter passes kept under profiles/ (taken
                with scripts/pmc.sh on this same workload), or null when no such file is there;
  cpu_baseline  the CPU oracle (oracle/cpu_ref.cpp, the bit-exact restatement of the
                reference renderer) timed on this box's host cores on a bounded sample of
                the same workload (same scene and resolution, fewer samples per pixel).
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E
VALU_PEAK_TFLOPS = 157.3    # vector fp32, FMA counted as 2
BYTES_PER_RAY = 176         # ray record written + read by intersect + read by shade (3 x 48) + hit record written + read (2 x 16)
BYTES_PER_HIT = 24          # float3 accumulator read-modify-write


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="example.sdf")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=32)
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--queue-gib", type=float, default=96.0, help="ray queue memory per GPU; larger queues mean fewer, larger launches")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration; 0 skips it")
    ap.add_argument("--cpu-kind", choices=["port", "reference"], default="port")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernel launches with HIP events")
    ap.add_argument("--pipeline", choices=["default", "fused", "staged"], default="default")
    ap.add_argument("--save-ppm", default=None)
    ap.add_argument("--verify", action="store_true", help="rank 0 also renders all samples alone and reports the max abs difference")
    return ap.parse_args()


def cpu_baseline(args, scene_path):
    """Oracle timed on the host cores: same scene and resolution, aa chosen for ~cpu_seconds."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_lib
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, os.cpu_count() or 1))
    if args.cpu_kind == "reference" and oracle_lib.Reference.available():
        lib, kind = oracle_lib.Reference(), "reference"
        threads = os.cpu_count() or 1  # the reference spawns hardware_concurrency() threads (framework/renderer.cpp:39)
    else:
        lib, kind = oracle_lib.Port(), "port"
    s = lib.load(scene_path)
    w, h, nb = args.width, args.height, args.bounces
    probe = lib.render(s, w, h, 1, nb, threads=threads)
    rate = w * h / max(probe["seconds"], 1e-6)
    aa = max(1, min(args.aa, int((args.cpu_seconds * rate / (w * h)) ** 0.5)))
    r = lib.render(s, w, h, aa, nb, threads=threads)
    samples = w * h * aa * aa
    return {"value": samples / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": f"{args.scene} {w}x{h}, aa_steps={aa} ({aa * aa} of {args.aa * args.aa} spp), {nb} bounces, {r['seconds']:.1f} s"}


def measured_traffic(args, kernel, launches):
    """HBM bytes per launch of `kernel` from the committed counter passes (the newest
    profiles/*_hbm_traffic.json, written by scripts/pmc_summary.py --traffic), if they were taken
    on this workload: same scene, size, spp, bounces and queue memory, hence the same launches."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    w = t.get("_workload", {})
    if [w.get(k) for k in ("scene", "width", "height", "aa", "bounces", "queue_gib")] != [args.scene, args.width, args.height, args.aa, args.bounces, args.queue_gib]:
        return None
    k = t.get(kernel)
    if not k or "hbm_bytes_per_launch" not in k:
        return None
    return k["hbm_bytes_per_launch"]


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    pt = importlib.import_module("path-tracer_amd")
    ptdist = importlib.import_module("path-tracer_amd.dist")

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
    local_rank = device_index

    scene_path = os.path.join(pt.SDF_DIR, args.scene)
    host = pt.HostScene(scene_path)
    # the queue block is the only large allocation; if the device cannot give that much (other
    # tenants), take half, and so on: smaller queues only mean more, smaller launches
    queue_gib = args.queue_gib
    while True:
        try:
            ds = pt.DeviceScene(host, local_rank, int(queue_gib * (1 << 30)))
            break
        except pt.PtxError as e:
            if "NOMEM" not in str(e) or queue_gib <= 1.0:
                raise
            queue_gib /= 2
            print(f"note: rank {rank}: queue allocation failed, retrying with {queue_gib} GiB", file=sys.stderr)
    args.queue_gib = queue_gib
    profile = not args.no_profile
    ds.set_profiling(profile)

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
            "dtype": "f32", "data": "synthetic (scene file shipped with the reference; regular sample grid, no RNG)",
            "config": {"workload": f"{args.scene} {w}x{h}, aa_steps={aa} ({spp} spp), max_ray_bounces={nb}", "queue_gib": queue_gib,
                       "samples_per_step": samples_per_step, "parallelism": f"sample-range x{world}, one RCCL sum-reduce per image"},
        }
        # ---- work done by this rank in the last step (the kernels count it themselves)
        R = (stats["trace_rays"] + stats["shadow_rays"])
        H = stats["shaded_hits"]
        my_samples = stats["samples"]
        out["work"] = {"rays_per_sample": R / my_samples, "shaded_hits_per_sample": H / my_samples,
                       "trace_rays_per_sample": stats["trace_rays"] / my_samples, "shadow_rays_per_sample": stats["shadow_rays"] / my_samples,
                       "algorithmic_bytes_per_sample": (BYTES_PER_RAY * R + BYTES_PER_HIT * H) / my_samples,
                       "Mrays/s": R * world * args.steps / elapsed / 1e6 if world == 1 else None,
                       "kernel_launches_per_step": stats["launches"]}
        kms, kln = stats["kernel_ms"], stats["kernel_launches"]
        if profile and kln[0][0]:
            # dominant kernel = the depth-0 launch of kernel kind 0: the fused bounce kernel, or
            # (staged pipeline) the closest-hit kernel.  Its algorithmic bytes: fused = every ray
            # segment it handles (primary + its shadow rays) and every hit it shades; staged
            # closest-hit kernel = its share of a ray segment: ray record read + hit record written.
            n0, ms0 = kln[0][0], kms[0][0]
            staged = kln[1][0] > 0
            rays0, shadow0, hits0 = stats["per_depth_rays"][0], stats["per_depth_shadow"][0], stats["per_depth_hits"][0]
            if staged:
                name = "k_trace<true, false> (depth 0: ray generation + closest hit -> hit records)"
                bytes0 = (48 + 16) * rays0
            else:
                name = "k_bounce<true> (depth 0: ray generation + closest hit + shading + shadow + emit + accumulate)"
                bytes0 = BYTES_PER_RAY * (rays0 + shadow0) + BYTES_PER_HIT * hits0
            achieved = bytes0 / (ms0 * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, "k_trace<true, false>" if staged else "k_bounce<true, true>", n0),
                               "launches": n0, "avg_launch_ms": ms0 / n0, "algorithmic_bytes_per_launch": bytes0 / n0}
            tot_ms = sum(sum(r) for r in kms)
            tot_bytes = BYTES_PER_RAY * R + BYTES_PER_HIT * H
            out["kernel_time"] = {"pipeline": "staged" if staged else "fused",
                                  "closest_hit_or_fused_ms_per_depth": [round(x, 3) for x in kms[0][: nb + 1]],
                                  "shade_ms_per_depth": [round(x, 3) for x in kms[1][: nb + 1]],
                                  "shadow_ms_per_depth": [round(x, 3) for x in kms[2][: nb + 1]],
                                  "per_depth_rays": stats["per_depth_rays"][: nb + 1],
                                  "per_depth_hits": stats["per_depth_hits"][: nb + 1],
                                  "per_depth_shadow_rays": stats["per_depth_shadow"][: nb + 1],
                                  "all_launches_ms": tot_ms, "all_launches_algorithmic_GBs": tot_bytes / (tot_ms * 1e-3) / 1e9}
        else:
            out["roofline"] = None
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, scene_path)
        else:
            out["cpu_baseline"] = None
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
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
