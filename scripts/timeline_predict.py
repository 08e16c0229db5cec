#!/usr/bin/env python3
"""How well does what is known before a launch predict how long a wave will take in it?  Reads the tables a
-DPTX_TIMELINE build left for every kernel and depth of one chunk (PTX_TIMELINE=-1:-1:<chunk>:<prefix>) and prints, per
kernel kind and depth, the correlation of a segment's wave time with its record count and with its wave time one
depth earlier, and what a launch would take if segments started longest-predicted first (list scheduling).
usage: scripts/timeline_predict.py <prefix> [slots]"""
import sys, struct, heapq
import numpy as np
prefix = sys.argv[1]
def load(kind, d):
    try:
        raw = open(f"{prefix}.k{kind}.d{d}", "rb").read()
    except FileNotFoundError:
        return None
    n_seg, seg_cap, k, dd = struct.unpack("4I", raw[:16])
    t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(n_seg, 4)
    seg = (t[:, 2] >> np.uint64(32)).astype(np.int64)
    rec = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    dur = (t[:, 1].astype(np.int64) - t[:, 0].astype(np.int64)) * 0.01
    span = (t[:, 1].max() - t[:, 0].min()) * 0.01
    out_d, out_r = np.zeros(n_seg), np.zeros(n_seg, dtype=np.int64)
    out_d[seg] = dur
    out_r[seg] = rec
    return out_d, out_r, span
def makespan(dur, order, slots):
    heap = [0.0] * slots
    heapq.heapify(heap)
    end = 0.0
    for i in order:
        t = heapq.heappop(heap) + dur[i]
        end = max(end, t)
        heapq.heappush(heap, t)
    return end
names = {0: "closest hit", 1: "shading", 2: "shadow"}
for kind in (0, 1, 2):
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else (5120 if kind == 2 else 4096)
    prev = None
    for d in range(0, 8):
        cur = load(kind, d)
        if cur is None:
            break
        dur, rec, span = cur
        n = len(dur)
        line = f"{names[kind]:11s} depth {d}: span {span:7.1f} us, ideal {dur.sum() / slots:7.1f}; in order {makespan(dur, range(n), slots):7.1f}"
        line += f", by count {makespan(dur, np.argsort(-rec, kind='stable'), slots):7.1f}, by true time {makespan(dur, np.argsort(-dur), slots):7.1f}"
        line += f"; corr(count) {np.corrcoef(rec, dur)[0, 1]:.2f}"
        if prev is not None:
            pd, pr = prev
            line += f", corr(time one depth up) {np.corrcoef(pd, dur)[0, 1]:.2f}; by that {makespan(dur, np.argsort(-pd), slots):7.1f}"
            est = pd / np.maximum(pr, 1) * rec
            line += f", by time per record one depth up x count {makespan(dur, np.argsort(-est), slots):7.1f}"
        print(line)
        prev = (dur, rec)
