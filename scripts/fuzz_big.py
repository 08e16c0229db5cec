"""Generated scenes (tests/fuzz_scenes.py) at full image size: pipelines and work-memory sizes
against each other (fuzz_scenes.full_size_findings).  usage: fuzz_big.py FIRST COUNT [WIDTH HEIGHT AA]"""
import importlib
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # noqa: F401,E402  (its HIP runtime first, as in tests/conftest.py)
from fuzz_scenes import full_size_findings  # noqa: E402

pt = importlib.import_module("path-tracer_amd")
first, count = int(sys.argv[1]), int(sys.argv[2])
size = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 4)
bad = 0
for seed in range(first, first + count):
    for line in full_size_findings(pt, seed, tempfile.mkdtemp(), *size):
        bad += 1
        print(line, flush=True)
    if (seed - first) % 10 == 9:
        print(f"... {seed - first + 1} scenes, {bad} findings", flush=True)
print(f"{count} scenes from {first}: {bad} findings")
sys.exit(1 if bad else 0)
