"""GPU occupancy from a rocprofv3 kernel trace: union of busy intervals, sum of kernel durations (> union = overlap), and the
top kernels by time inside [t0 + skip, t1].  python tools/trace_busy.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
f = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
cols = None
for r in csv.DictReader(open(f)):
    if cols is None:
        cols = list(r.keys())
    name = r['Kernel_Name']
    short = 'MARK spin_kernel' if 'spin_kernel' in name else name.split('(')[0][:70]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short, r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
marks = [r for r in rows if r[2].startswith('MARK')]
if len(marks) >= 2:             # bench.py with IFH_TRACE_MARK=1: the timed region lies between the first and the last marker pair
    lo, t1 = marks[-2][1], marks[-1][0]
    if len(marks) >= 4:         # (a warm-up pass was bracketed too: take the widest gap)
        gaps = [(marks[i + 1][0] - marks[i][1], i) for i in range(len(marks) - 1)]
        i = max(gaps)[1]
        lo, t1 = marks[i][1], marks[i + 1][0]
    rows = [r for r in rows if r[0] >= lo and r[1] <= t1]
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= lo]
wall = (t1 - lo) * 1e-6
busy, cur_s, cur_e = 0, None, None
for s, e, _, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _, _ in rows)
print('window %.1f ms: GPU busy (union) %.1f ms = %.1f %%; sum of kernel durations %.1f ms (x%.2f of busy); %d launches = %.0f / s'
      % (wall, busy * 1e-6, 100 * busy * 1e-6 / wall, tot * 1e-6, tot / busy, len(rows), len(rows) / wall * 1e3))
ev = sorted([(s_, 1) for s_, e_, _, _, _ in rows] + [(e_, -1) for s_, e_, _, _, _ in rows])
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(lvl, 4)] += t - last
    last, lvl = t, lvl + d
tt = sum(hist.values())
print('kernels resident: ' + '  '.join('%s: %.1f %%' % (('%d' % k) if k < 4 else '4+', 100 * hist[k] / tt) for k in sorted(hist)))
by = collections.defaultdict(lambda: [0, 0, []])
qs = collections.Counter()
for s, e, k, q, _sid in rows:
    qs[q] += e - s
    by[k][0] += e - s
    by[k][1] += 1
    by[k][2].append(e - s)
print('   total      share    launches   average   median      p99      max   kernel')
for k, (d, n, ds) in sorted(by.items(), key=lambda kv: -kv[1][0])[:28]:
    ds.sort()
    print('%6.1f ms %5.1f %% %7d x %7.1f us %7.1f %8.1f %8.1f  %s' % (d * 1e-6, 100 * d / tot, n, d / n * 1e-3, ds[len(ds) // 2] * 1e-3,
                                                                ds[min(len(ds) - 1, int(len(ds) * 0.99))] * 1e-3, ds[-1] * 1e-3, k))
print('columns:', cols)
print('busy per queue (ms):', {q: round(v * 1e-6, 1) for q, v in qs.most_common()})

# per stream: kernels, busy time, and the gaps between consecutive kernels of the stream (a dependent launch chain's gaps are
# what the command processor and the wait for CU slots add to it)
bys = collections.defaultdict(list)
for s, e, k, q, sid in rows:
    bys[(q, sid)].append((s, e, k))
print('per (queue, stream): launches, busy ms, gaps < 200 us: count / mean us / median us / sum ms; top kernels')
for key, lst in sorted(bys.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    lst.sort()
    busy_s = sum(e - s for s, e, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    g = sorted(x for x in gaps if 0 <= x < 200000)
    names = collections.Counter(k for _, _, k in lst).most_common(3)
    if len(lst) < 50:
        continue
    print('  q%s s%s: %6d launches, busy %7.1f ms, gaps %6d / %5.1f / %5.1f / %6.1f ms;  %s' % (
        key[0], key[1], len(lst), busy_s * 1e-6, len(g), (sum(g) / max(1, len(g))) * 1e-3, (g[len(g) // 2] if g else 0) * 1e-3, sum(g) * 1e-6,
        ', '.join('%s x%d' % (n[-28:], c) for n, c in names)))

# which kernels are resident ALONE (exactly one kernel on the chip), and in pairs: time by kernel name
ev2 = sorted([(s_, 1, k) for s_, e_, k, _, _ in rows] + [(e_, -1, k) for s_, e_, k, _, _ in rows])
active = collections.Counter()
alone = collections.Counter()
last = ev2[0][0]
for t, d, k in ev2:
    n = sum(active.values())
    if n == 1 and t > last:
        alone[next(iter(+active))] += t - last
    last = t
    active[k] += d
tot_alone = sum(alone.values())
print('resident alone: %.1f ms in all; by kernel:' % (tot_alone * 1e-6))
for k, v in alone.most_common(12):
    print('   %7.1f ms  %4.1f %%  %s' % (v * 1e-6, 100.0 * v / max(1, tot_alone), k))


# launches that cannot fill the chip: kernels by time spent in launches of fewer than 256 workgroups
import csv as _csv
small = collections.defaultdict(lambda: [0, 0, 0])
for r in _csv.DictReader(open(f)):
    s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s_ < lo or s_ > t1:
        continue
    try:
        wgs = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * \
              (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    except (KeyError, ValueError):
        continue
    if wgs < 256:
        k = r['Kernel_Name'].split('(')[0][:70]
        b = small[k]
        b[0] += e_ - s_
        b[1] += 1
        b[2] += wgs
print('launches of < 256 workgroups: time / launches / mean workgroups, by kernel')
for k, (d, n, w) in sorted(small.items(), key=lambda kv: -kv[1][0])[:14]:
    print('   %7.1f ms  %6d x  %5.0f wgs  %s' % (d * 1e-6, n, w / max(1, n), k))
