"""Diagnostic: `bench.timed_mode` for several models in ONE process, e.g. `python tools/four_models.py fp16 fp16 fp16 fp16 fp16 fp16`.
Round 5 measured that the FOURTH model of a process ran 7-11 % slow whichever mode it was; round 6 found the cause (a fresh step
stream per Engine walking round HIP's ring of hardware queues until it shared one with the library's side stream:
profiles/r06/c1_fourth_model_root_cause.txt) and fixed it (one step stream per device): this tool is the check -- every model
within 1 % of the others.  "tiny" inserts a throw-away 3-image model."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = types.SimpleNamespace(windows=64, height=256, width=341, warmup=5, steps=50, no_events=True, emu=False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tiny = types.SimpleNamespace(windows=1, height=32, width=40, warmup=1, steps=1, no_events=True, emu=False)
res = []
for i, dt in enumerate(sys.argv[1:]):
    if dt == "tiny":
        bench.timed_mode(tiny, "fp32", dev, None, 1, 0, 1)
        print(i + 1, "tiny", flush=True)
        continue
    r = bench.timed_mode(args, dt, dev, None, 1, 0, 2)
    res.append((dt, min(r["region_ms_per_step"])))
    print(i + 1, dt, r["region_ms_per_step"], flush=True)
by = {}
for dt, v in res:
    by.setdefault(dt, []).append(v)
for dt, v in by.items():
    print("%s: min %.3f max %.3f ms/step over %d models: spread %.2f %%" % (dt, min(v), max(v), len(v), 100 * (max(v) / min(v) - 1)))
