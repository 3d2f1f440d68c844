"""What ran while the slowest per-tick launch waited: python tools/trace_tick_stall.py <kernel_trace.csv>
(rocprofv3 --kernel-trace of bench.py with the tick probe on).  The tick kernels are paced at 20 ms; the launch whose start lies
furthest behind its predecessor + 20 ms is the stalled one; prints the kernels resident between the expected and the real start."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:80], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
allt = [r for r in rows if 'k_ingest_tick' in r[2]]
# the front lanes launch the same kernel in bursts; the probe's stream is the one paced at 20 ms
groups = collections.defaultdict(list)
for r in allt:
    groups[(r[3], r[4])].append(r)
def med_gap(lst):
    g = sorted((lst[i + 1][0] - lst[i][0]) * 1e-6 for i in range(len(lst) - 1))
    return g[len(g) // 2] if g else 0.0
key = min(groups, key=lambda k: abs(med_gap(groups[k]) - 20.0))
ticks = groups[key]
print('%d tick-kernel launches in all; the probe: queue %s stream %s, %d launches, median gap %.1f ms' % (len(allt), key[0], key[1], len(ticks), med_gap(ticks)))
gaps = sorted(((ticks[i + 1][0] - ticks[i][0]) * 1e-6, i) for i in range(len(ticks) - 1))
print('start-to-start gaps (ms): median %.1f, largest %s' % (gaps[len(gaps) // 2][0], [round(g, 1) for g, _ in gaps[-6:]]))
for g, i in gaps[-3:]:
    if g < 30:
        continue
    lo, hi = ticks[i][0] + 20_000_000, ticks[i + 1][0]
    if hi <= lo:
        continue
    print('---- tick %d started %.1f ms late; kernels overlapping the %.1f ms it waited:' % (i + 1, (hi - lo) * 1e-6, (hi - lo) * 1e-6))
    by = collections.defaultdict(lambda: [0, 0, 0])
    for s, e, k, q, sid in rows:
        if e <= lo or s >= hi:
            continue
        b = by[(k, q, sid)]
        b[0] += min(e, hi) - max(s, lo)
        b[1] += 1
        b[2] = max(b[2], e - s)
    for (k, q, sid), (d, n, mx) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print('   %7.2f ms  %5d x  longest %8.1f us  q%s s%s  %s' % (d * 1e-6, n, mx * 1e-3, q, sid, k))
    # what ran on the tick's own queue in that window
    tq = ticks[i + 1][3]
    same = [(s, e, k, sid) for s, e, k, q, sid in rows if q == tq and e > lo and s < hi]
    print('   on the tick\'s hardware queue q%s in that window: %d kernels, streams %s' % (tq, len(same), collections.Counter(x[3] for x in same).most_common(4)))
