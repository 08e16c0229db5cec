#!/usr/bin/env python3
"""When the waves of one launch start and end, and on which compute unit: renders the bench image once with a
-DPTX_TIMELINE build of the library (every wave of the queue kernels leaves start, end, records and hardware id in a
table) and prints, for the launch asked for, how full the chip was over the launch's time.
usage: scripts/timeline.py <lib> <kind 0 closest hit | 1 shading | 2 shadow> <depth> <chunk> [scene] [aa] [GiB]
(PTX_NO_ORDER=1 in the environment: segments in the order of their numbers)"""
import importlib, os, sys, struct
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
lib, kind, depth, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene = sys.argv[5] if len(sys.argv) > 5 else "cornell_buster.sdf"
aa = int(sys.argv[6]) if len(sys.argv) > 6 else 32
gib = float(sys.argv[7]) if len(sys.argv) > 7 else 32.0
out = os.path.join(REPO, "gpurun_out", f"timeline_{kind}_{depth}_{chunk}.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["PTX_TIMELINE"] = f"{kind}:{depth}:{chunk}:{out}"
import torch
pt = importlib.import_module("path-tracer_amd")
pt.set_library(os.path.join(REPO, lib))
hs = pt.HostScene(os.path.join(pt.SDF_DIR, scene))
w, h, nb = 1920, 1080, 5
acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ds = pt.DeviceScene(hs, 0, int(gib * (1 << 30)))
ds.render_accumulate(pt.render_params(w, h, aa, nb), acc.data_ptr(), stream, check_overflow=True, want_stats=True)
torch.cuda.synchronize()
ds.close()
if kind < 0 or depth < 0:  # every kind / depth of the chunk: one file each (<out>.k<kind>.d<depth>), read by scripts/timeline_predict.py
    sys.exit(0)

raw = open(out, "rb").read()
n_seg, seg_cap, k, d = struct.unpack("4I", raw[:16])
t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(n_seg, 4)
t0, t1, rec, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64), t[:, 3]
tick_us = 0.01  # wall_clock64: 100 MHz
start, end = t0.min(), t1.max()
span = (end - start) * tick_us
dur = (t1 - t0) * tick_us
print(f"kind {k} depth {d} chunk {chunk}: {n_seg} waves, seg_cap {seg_cap}, launch spans {span:.1f} us; wave time mean {dur.mean():.1f} us,"
      f" median {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, p99 {np.percentile(dur, 99):.1f}, max {dur.max():.1f}")
print(f"records per wave: mean {rec.mean():.0f}, median {np.median(rec):.0f}, p90 {np.percentile(rec, 90):.0f}, max {rec.max()}, empty waves {(rec == 0).sum()}")
busy = dur.sum()
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
hwid = hw.astype(np.int64) & 0xffffffff
cu = (hwid >> 8) & 0xf
se = (hwid >> 13) & 0x7
sh = (hwid >> 12) & 0x1
simd = (hwid >> 4) & 0x3
cu_key = xcc * 1000 + se * 100 + sh * 20 + cu
ncu = len(np.unique(cu_key))
print(f"compute units seen {ncu}, XCCs {len(np.unique(xcc))}; wave-time summed {busy:.0f} us = {busy / span:.1f} waves resident on average"
      f" ({busy / span / max(ncu, 1):.2f} per compute unit)")
# occupancy over time: 20 slices of the launch
edges = np.linspace(start, end, 21)
occ = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
    occ.append(ov)
print("waves resident per twentieth of the launch:", " ".join(f"{o:.0f}" for o in occ))
# time from launch begin to each wave's start, in order of wave number
order = np.argsort(t0)
print("start time (us) of the wave at rank 0 / 25 / 50 / 75 / 100 % of the starts:",
      " ".join(f"{(t0[order[int(q * (n_seg - 1))]] - start) * tick_us:.1f}" for q in (0, .25, .5, .75, 1.0)))
r = np.corrcoef(rec, dur)[0, 1] if rec.std() > 0 else 0.0
print(f"correlation records ~ wave time {r:.3f}; us per 64 records: {64 * dur.sum() / max(rec.sum(), 1):.2f}")
per_cu = np.array([dur[cu_key == c].sum() for c in np.unique(cu_key)])
print(f"busy wave-time per compute unit: min {per_cu.min():.0f}, mean {per_cu.mean():.0f}, max {per_cu.max():.0f} us")
