"""Where the launch chain of the busiest stream of a rocprofv3 kernel trace waits: for that stream, the gap in front of every kernel
(previous kernel's end -> this kernel's start), summed by the kernel that follows the gap, and the distribution of the large gaps.
    python tools/trace_gaps.py <kernel_trace.csv>     (bench.py with IFH_TRACE_MARK=1: the timed region between the markers)"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    short = 'MARK' if 'spin_kernel' in name else name.split('(')[0][:60]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short, r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
marks = [r for r in rows if r[2] == 'MARK']
if len(marks) >= 2:
    gaps = [(marks[i + 1][0] - marks[i][1], i) for i in range(len(marks) - 1)]
    i = max(gaps)[1]
    lo, hi = marks[i][1], marks[i + 1][0]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
cnt = collections.Counter((r[3], r[4]) for r in rows)
(q, sid), n = cnt.most_common(1)[0]
ch = [r for r in rows if (r[3], r[4]) == (q, sid)]
print('stream (queue %s, stream %s): %d launches, window %.1f ms' % (q, sid, n, (ch[-1][1] - ch[0][0]) * 1e-6))
by = collections.defaultdict(lambda: [0, 0, 0, 0])
big = []
prev_name = None
for a, b in zip(ch, ch[1:]):
    g = max(0, b[0] - a[1])
    e = by[(a[2], b[2])]
    e[0] += g
    e[1] += 1
    if g >= 200000:
        e[2] += g
        e[3] += 1
        big.append(g)
tot = sum(v[0] for v in by.values())
busy = sum(r[1] - r[0] for r in ch)
print('kernel time %.1f ms, gaps %.1f ms (of which gaps >= 200 us: %.1f ms in %d)' % (busy * 1e-6, tot * 1e-6, sum(big) * 1e-6, len(big)))
print('  gap ms   count   mean us   big ms  big n   previous kernel -> next kernel')
for (pa, pb), v in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print('%8.1f %7d %9.1f %8.1f %6d   %s -> %s' % (v[0] * 1e-6, v[1], v[0] / v[1] * 1e-3, v[2] * 1e-6, v[3], pa, pb))
big.sort()
if big:
    print('large gaps: median %.0f us, p90 %.0f us, max %.0f us' % (big[len(big) // 2] * 1e-3, big[int(len(big) * 0.9)] * 1e-3, big[-1] * 1e-3))
