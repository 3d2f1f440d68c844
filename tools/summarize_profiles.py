"""Turn the rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/.

  python tools/summarize_profiles.py pmc    gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE <passes> profiles/r01_vocoder_pmc.json
  python tools/summarize_profiles.py stats  gpurun_out/bench_kernel_stats.csv profiles/r01_bench_kernel_stats.md "<command line>"

pmc: sums the counter over the kernels of the LAST vocoder pass (kernel-trace order), applies the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B wide read request -> x2; both counters are in KB)."""
import csv, glob, json, os, sys


def pmc_sum(d, counter, passes):
    f = max(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)   # newest run
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # keep only the vocoder's own kernels (ifh::), split evenly into passes, take the last pass
    rows = [r for r in rows if 'ifh::' in r['Kernel_Name']]
    per = len(rows) // passes
    last = rows[-per:]
    by = {}
    for r in last:
        k = r['Kernel_Name'].split('(')[0]
        by[k] = by.get(k, 0.0) + float(r['Counter_Value'])
    return per, sum(by.values()), by


def main():
    if sys.argv[1] == 'pmc':
        dF, dW, passes, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
        chunks = int(sys.argv[6]) if len(sys.argv) > 6 else 768
        what = sys.argv[7] if len(sys.argv) > 7 else 'vocoder'
        nF, sF, bF = pmc_sum(dF, 'FETCH_SIZE', passes)
        nW, sW, bW = pmc_sum(dW, 'WRITE_SIZE', passes)
        res = {
            'command': 'rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 tools/probe_%s.py %d %d  (tools/prof_r05.sh; last of %d passes)' % (what, passes, chunks, passes),
            'workload': ('one HiFi-GAN vocoder pass, %d chunks x 12 frames (bench.py roofline leg)' % chunks) if what == 'vocoder'
                        else ('one log-mel launch (ifh_logmel_run_raw), %d windows of 30 s -> raw [80,3000] f32 + window maxima (bench.py roofline_logmel leg)' % chunks),
            'chunks_per_pass': chunks,
            'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KB',
            'fetch_bytes_corrected': sF * 1024 * 2, 'write_bytes': sW * 1024,
            'hbm_bytes_per_pass': sF * 1024 * 2 + sW * 1024,
            'raw': {'FETCH_SIZE': {'kernels_per_pass': nF, 'sum_KB': sF, 'by_kernel_KB': bF},
                    'WRITE_SIZE': {'kernels_per_pass': nW, 'sum_KB': sW, 'by_kernel_KB': bW}},
        }
        json.dump(res, open(out, 'w'), indent=1)
        print(json.dumps({k: res[k] for k in ('fetch_bytes_corrected', 'write_bytes', 'hbm_bytes_per_pass')}))
    else:
        src, out, cmd = sys.argv[2], sys.argv[3], sys.argv[4]
        rows = list(csv.DictReader(open(src)))
        tot = sum(int(r['TotalDurationNs']) for r in rows)
        with open(out, 'w') as f:
            f.write('# %s\n\n1x MI355X.  Total GPU kernel time %.1f ms.\n\n| kernel | calls | total ms | avg us | %% |\n|---|---:|---:|---:|---:|\n' % (cmd, tot / 1e6))
            for r in rows[:40]:
                f.write('| `%s` | %s | %.2f | %.2f | %.1f |\n' % (r['Name'][:90], r['Calls'], int(r['TotalDurationNs']) / 1e6,
                                                                 float(r['AverageNs']) / 1e3, float(r['Percentage'])))
        print('wrote', out)


if __name__ == '__main__':
    main()
