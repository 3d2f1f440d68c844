"""Average duration per (kernel, grid, LDS bytes) inside the timed region of a bench.py kernel trace (IFH_TRACE_MARK=1 markers).
python tools/trace_shapes.py <kernel_trace.csv> [name substring]"""
import csv, sys, collections
f = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else 'k_gemm_dec'
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), ('MARK spin_kernel' if 'spin_kernel' in r['Kernel_Name'] else r['Kernel_Name'].split('(')[0][:50]),
                 (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['LDS_Block_Size']))))
rows.sort()
marks = [r for r in rows if "spin_kernel" in r[2]]
if len(marks) >= 2:
    gaps = [(marks[i + 1][0] - marks[i][1], i) for i in range(len(marks) - 1)]
    i = max(gaps)[1]
    lo, hi = marks[i][1], marks[i + 1][0]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
by = collections.defaultdict(list)
for s, e, k, g in rows:
    if sub in k:
        by[(k, g)].append(e - s)
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:30]:
    v.sort()
    print('%-50s grid %-16s n %6d  avg %7.1f us  p10 %7.1f  p50 %7.1f  p90 %7.1f  total %7.1f ms' %
          (k, g, len(v), sum(v) / len(v) * 1e-3, v[len(v) // 10] * 1e-3, v[len(v) // 2] * 1e-3, v[len(v) * 9 // 10] * 1e-3, sum(v) * 1e-6))
