"""Idle time between consecutive kernels of each hardware queue in a rocprofv3 kernel trace of the bench step.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o r -- python bench.py --steps 8 --warmup 3 --repeats 1 <lean flags>
    python tools/gap_analysis.py /tmp/p/r_kernel_trace.csv [steps]

Prints, per queue: kernels, busy time, gaps (start[i+1] - end[i] where positive and < 200 us -- longer ones are the host between
steps / regions), and the largest gap contributors by (previous kernel -> next kernel) pair.
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(csv.DictReader(open(path)))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        busy = sum(e - s for s, e, _ in ks)
        gaps = collections.defaultdict(lambda: [0, 0.0])
        tot = 0.0
        n = 0
        for (s0, e0, k0), (s1, e1, k1) in zip(ks, ks[1:]):
            g = s1 - e0
            if 0 < g < 200000:
                tot += g
                n += 1
                key = (k0.split("(")[0][-48:], k1.split("(")[0][-48:])
                gaps[key][0] += 1
                gaps[key][1] += g
        print("queue %s: %d kernels, busy %.3f ms/step, %d gaps = %.3f ms/step (mean %.2f us)"
              % (q, len(ks), busy / 1e6 / steps, n, tot / 1e6 / steps, tot / max(n, 1) / 1e3))
        for key, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
            print("    %7.1f us/step  n=%5d  mean %5.2f us   %s -> %s" % (t / 1e3 / steps, c, t / c / 1e3, key[0], key[1]))


if __name__ == "__main__":
    main()
