#!/usr/bin/env python3
"""Which kernels ran at the same time?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and prints, per kernel name, the
launches, their total duration and how much of it lay under a launch of ANOTHER kernel on a different queue -- the check that the
Viterbi launches of sub-batch k really run under the conditioning / flank alignments of sub-batch k + 1.
    python tools/overlap_report.py gpurun_out/<dir>/**/kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
    rows.sort()
    t0 = rows[0][0]
    tot = defaultdict(float); under = defaultdict(float); cnt = defaultdict(int); queues = defaultdict(set)
    big = [r for r in rows if r[1] - r[0] > 200000]          # launches over 0.2 ms
    for i, (s, e, n, q) in enumerate(rows):
        tot[n] += (e - s) / 1e6; cnt[n] += 1; queues[n].add(q)
        if e - s <= 200000:
            continue
        # union of the other big launches (other queue) over [s, e]
        segs = sorted((max(s, s2), min(e, e2)) for (s2, e2, n2, q2) in big if q2 != q and s2 < e and e2 > s)
        cur = s; ov = 0
        for a, b in segs:
            a = max(a, cur)
            if b > a:
                ov += b - a; cur = b
        under[n] += ov / 1e6
    print("%-62s %6s %10s %10s  queues" % ("kernel", "calls", "total ms", "under ms"))
    for n in sorted(tot, key=lambda k: -tot[k])[:16]:
        print("%-62s %6d %10.2f %10.2f  %s" % (n, cnt[n], tot[n], under[n], ",".join(sorted(queues[n]))))
    print("wall %.1f ms, sum of launches %.1f ms" % ((max(r[1] for r in rows) - t0) / 1e6, sum(tot.values())))
    if len(sys.argv) > 2:
        # timeline of the launches over 1 ms inside the last `sys.argv[2]` ms of the trace
        end = max(r[1] for r in rows); span = float(sys.argv[2]) * 1e6
        print("timeline (ms before the end of the trace): start, end, queue, kernel")
        for s, e, n, q in rows:
            if e - s > 30000 and s > end - span:
                print("  %9.1f %9.1f  q%-3s %s" % ((s - end) / 1e6, (e - end) / 1e6, q, n))


if __name__ == "__main__":
    main(sys.argv[1])
