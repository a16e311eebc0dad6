"""profiles/rNN/pmc_conv_traffic.json from the two PMC passes of tools/gpu_prof.sh (PMC=1).

usage: python tools/pmc_conv_traffic.py gpurun_out/prof_TAG profiles/r02 "<source note>"
FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH_SIZE is doubled on gfx950 (16-byte-per-lane loads are tallied at half
their bytes, MI355X_MICROARCH.md, HBM section).  Steps of the profiled run = dispatches of the input-layout kernel."""
import csv
import json
import os
import re
import sys

CONV = re.compile(r"igemm|conv_halo|wgrad|stem_conv")


def load(path):
    rows = list(csv.DictReader(open(path)))
    steps = [int(r["dispatches"]) for r in rows if "to_padded_nhwc4" in r["kernel"]]
    return rows, (steps[0] if steps else 1)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    out = {}
    for cname, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        rows, steps = load(os.path.join(src, "pmc_%s.csv" % cname))
        key = "sum_%s" % cname
        conv = sum(float(r[key]) for r in rows if CONV.search(r["kernel"])) * 1024.0 * mul / steps
        whole = sum(float(r[key]) for r in rows) * 1024.0 * mul / steps
        out[cname] = (conv, whole, steps)
    js = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), bench.py --steps 2 --warmup 1, "
                  "MN_WGRAD_STREAM=0; " + note,
        "correction": "FETCH_SIZE x2 (gfx950: 16-byte/lane loads are tallied at half their bytes, MI355X_MICROARCH.md HBM section); "
                      "counters are KB",
        "kernels": "igemm* + conv_halo* + wgrad* (incl. the fused kernel's reduce launch) + stem_conv / stem_wgrad: the conv "
                   "launches of one step (incl. fp32 fc GEMMs)",
        "steps_profiled": out["FETCH_SIZE"][2],
        "fetch_bytes_per_step": round(out["FETCH_SIZE"][0], -7),
        "write_bytes_per_step": round(out["WRITE_SIZE"][0], -7),
        "hbm_bytes_per_step": round(out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0], -7),
        "whole_step_fetch_bytes": round(out["FETCH_SIZE"][1], -7),
        "whole_step_write_bytes": round(out["WRITE_SIZE"][1], -7),
    }
    with open(os.path.join(dst, "pmc_conv_traffic.json"), "w") as f:
        json.dump(js, f, indent=1)
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
