"""The multi-GPU split on CPU: two gloo ranks, each takes its slice of the per-pixel sample
grid, the per-rank accumulation buffers are summed onto rank 0 with path-tracer_amd.dist
(the code bench.py runs over RCCL), and the result must equal the single-process image.
Without a GPU the per-rank accumulation is produced by the CPU oracle (un-tone-mapped
per-sample colours), which is exactly what ptx_render_accumulate adds up on a device."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, aa, nb, out_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ptdist = importlib.import_module("path-tracer_amd.dist")
    from oracle_lib import Port, SDF_DIR
    oracle = Port()
    s = oracle.load(os.path.join(SDF_DIR, "cornell.sdf"))
    spp = aa * aa
    b, e = ptdist.sample_range(rank, world, spp)
    samples = oracle.sample_colors(s, w, h, aa, nb, 0, 0, w, h)  # [h, w, spp, 3], sample index = aax * aa + aay
    accum = torch.from_numpy(samples[:, :, b:e, :].astype(np.float64).sum(axis=2).astype(np.float32).reshape(-1).copy())
    ptdist.reduce_accum(accum, 0)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sample_split_and_reduce_gloo(tmp_path, port, scenes, world):
    w, h, aa, nb = 24, 16, 3, 3
    out = str(tmp_path / "accum.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, aa, nb, out), nprocs=world, join=True)
    accum = np.load(out).reshape(h, w, 3)
    # resolve as renderer.cpp:86-89 does and compare with the oracle's own image
    c = accum * np.float32(1.0 / (aa * aa))
    c = c / (c + np.float32(1.0))
    ref = port.render(scenes.oracle_scene("cornell"), w, h, aa, nb)
    assert np.max(np.abs(c - ref["color"])) <= 1e-6
