"""Diagnostic for the "fourth model is 5-10 % slow" effect: where a freshly allocated work arena lands.

  python tools/arena_probe.py churn    # five fp16 models, arena freed (torch.cuda.empty_cache) and re-allocated between them (= bench.py)
  python tools/arena_probe.py keep     # five fp16 models, every arena kept alive (no memory ever goes back to the driver)
  python tools/arena_probe.py pool     # ONE allocation made first; every model's arena is a slice of it, at a sweep of base offsets

Per model: ms/step (two regions of 30 steps), the arena's virtual address, and a page probe of the fresh arena -- the time to touch one
4-byte word every 4 KiB / 64 KiB / 2 MiB of it (a TLB-reach probe: physical backing in small fragments shows up as a slower sweep)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import geomapnet_amd.engine as E  # noqa: E402

args = types.SimpleNamespace(windows=64, height=256, width=341, warmup=5, steps=30, no_events=True, emu=False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
DT = os.environ.get("DT", "fp16")


def page_probe(t, stride, reps=8):
    n = (t.numel() // stride) * stride
    v = t[:n].view(torch.int32).view(-1, stride // 4)[:, 0]
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        v.sum()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best


def stream_probe(t, reps=4):  # plain streaming read of the whole arena: GB/s
    v = t[: (t.numel() // 16) * 16].view(torch.int32)
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        v.sum()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return t.numel() / best / 1e6


LOG = []
KEEP = []
mode = sys.argv[1] if len(sys.argv) > 1 else "churn"
POOL = None
OFFSET = 0


def alloc(nbytes, device):
    if mode == "pool":
        t = POOL[OFFSET: OFFSET + nbytes]
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if mode == "keep":
            KEEP.append(t)
    LOG.append({"GB": round(nbytes / 1e9, 2), "addr": hex(t.data_ptr()), "mod_2MiB": t.data_ptr() % (2 << 20),
                "probe_us_4K_64K_2M": [round(page_probe(t, s), 1) for s in (4096, 65536, 2 << 20)],
                "stream_GBps": round(stream_probe(t))})
    return t


E.work_allocator = alloc
if mode == "pool":
    POOL = torch.empty(int(os.environ.get("POOL_GB", "24")) << 30, dtype=torch.uint8, device=dev)
    print("pool at", hex(POOL.data_ptr()), flush=True)
    offsets = [0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096, (1 << 30) + 65536, 0, (3 << 20) + 768, 0]
else:
    offsets = [0] * int(os.environ.get("MODELS", "6"))
for i, off in enumerate(offsets):
    OFFSET = off
    r = bench.timed_mode(args, DT, dev, None, 1, 0, 2)
    print(i + 1, mode, DT, "offset", off, r["region_ms_per_step"], LOG[-1], "reserved GB %.1f" % (torch.cuda.memory_reserved() / 1e9),
          flush=True)
