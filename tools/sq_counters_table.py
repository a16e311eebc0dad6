"""rocprofv3 --pmc counter_collection.csv passes -> per-kernel table of matrix-pipe busy fraction, LDS stalls and conflicts
(tools/sq_counters.sh).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): rocprofv3 reports GRBM_GUI_ACTIVE
summed over the 8 XCDs (it reads 8 x duration x ~2.1-2.4 GHz), so a per-chip cycle count is an eighth of it; beside it the same ratio
against duration x 2.4 GHz.  The first is the busy fraction at the clock the kernel actually ran at (MFMA-dense kernels run at
~2.05-2.2 GHz under the power budget), the second against the nominal peak clock the 2.5 PF figure assumes."""
import collections
import csv
import json
import sys

dt, files, trace, out = sys.argv[1], sys.argv[2:-2], sys.argv[-2], sys.argv[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    try:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mn" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    except OSError as e:
        print("missing", f, e)
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(trace)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
except OSError as e:
    print("missing", trace, e)
rows = []
for k, d in agg.items():
    per = {c: v / cnt[(k, c)] for c, v in d.items()}
    n = max(cnt[(k, c)] for c in d)
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0.0
    gui = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # per-chip cycles (the counter is summed over 8 XCDs)
    mf = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rows.append({"kernel": k[:110], "dispatches": n, "avg_us_profiled": round(us, 1), "total_us": round(us * n, 1),
                 "mfma_busy_vs_gui": round(mf / (1024.0 * gui), 4) if gui else None,
                 "mfma_busy_vs_2p4GHz": round(mf / (1024.0 * us * 2400.0), 4) if us else None,
                 "eff_clock_GHz": round(gui / us / 1e3, 3) if us and gui else None,
                 "lds_bank_conflict_frac": round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"], 4)
                 if per.get("SQ_LDS_IDX_ACTIVE") else None,
                 "wait_inst_lds_frac": round(per.get("SQ_WAIT_INST_LDS", 0.0) / per["SQ_WAVE_CYCLES"], 4)
                 if per.get("SQ_WAVE_CYCLES") else None,
                 "wait_any_frac": round(per.get("SQ_WAIT_ANY", 0.0) / per["SQ_WAVE_CYCLES"], 4)
                 if per.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in per else None,
                 "counters_per_dispatch": {c: float("%.5g" % v) for c, v in sorted(per.items())}})
rows.sort(key=lambda r: -r["total_us"])
with open("%s/sq_counters_%s.json" % (out, dt), "w") as f:
    json.dump(rows, f, indent=1)
with open("%s/sq_counters_%s.txt" % (out, dt), "w") as f:
    f.write("# %s step, serial kernels; rocprofv3 --pmc passes (tools/sq_counters.sh); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); GHz = effective clock\n" % dt)
    f.write("%-92s %5s %9s %9s %9s %7s %8s %8s\n" % ("kernel", "n", "avg us", "mfma_busy", "mfma/2.4G", "GHz", "ldsconf", "waitlds"))
    for r in rows:
        f.write("%-92s %5d %9.1f %9s %9s %7s %8s %8s\n" % (r["kernel"][:92], r["dispatches"], r["avg_us_profiled"], r["mfma_busy_vs_gui"],
                                                        r["mfma_busy_vs_2p4GHz"], r["eff_clock_GHz"], r["lds_bank_conflict_frac"],
                                                        r["wait_inst_lds_frac"]))
print(open("%s/sq_counters_%s.txt" % (out, dt)).read()[:6000])
