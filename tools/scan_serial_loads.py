#!/usr/bin/env python3
"""Find loads that hipcc serialised: a global / buffer load followed within three lines by `s_waitcnt vmcnt(0)` -- the shape a load
inside a per-lane `if` takes (its own exec branch, then a full wait), which turns n independent requests into n memory round trips
in series (round 5: k_layernorm 3.7 -> 5.2 TB/s, k_beam_rowtop 229 -> 111 us per step once their loads left together).
    python tools/scan_serial_loads.py [listing.s ...]       (no argument: compiles every csrc/*.hip to /tmp and scans it)
Prints, per kernel with >= MIN (default 3) such pairs: loads, serial pairs, exec branches.  A screen, not a verdict: in k_gemm_dec the
pairs sit under the first chunk's flight and removing them measured slower (profiles/NOTES.md)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    """-> {mangled kernel name: [loads, load+vmcnt(0) pairs, exec branches]}"""
    lines = open(path).read().split('\n')
    name, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
            stats[name] = [0, 0, 0]
        if name is None:
            continue
        t = l.strip()
        if (t.startswith('global_load') or t.startswith('buffer_load')) and 'lds' not in t:
            stats[name][0] += 1
            if any('s_waitcnt' in lines[i + k] and 'vmcnt(0)' in lines[i + k] for k in range(1, 4) if i + k < len(lines)):
                stats[name][1] += 1
        if t.startswith('s_cbranch_exec'):
            stats[name][2] += 1
    return stats


def main():
    sys.path.insert(0, ROOT)
    from infernos_amd import build as b
    files = sys.argv[1:]
    if not files:
        for src in sorted(glob.glob(os.path.join(b.CSRC, '*.hip'))):
            flags = [f for f in b.flags_for(src) if f not in ('-fPIC', '-Wall')]
            out = '/tmp/scan_' + os.path.basename(src) + '.s'
            if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
                subprocess.run([b.HIPCC] + flags + ['-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
            if os.path.exists(out):
                files.append(out)
    floor = int(os.environ.get('MIN', '3'))
    for f in files:
        for n, (ld, ser, br) in scan(f).items():
            if ser >= floor:
                dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:110]
                print(f'{os.path.basename(f):24s} loads {ld:4d}  load+vmcnt(0) {ser:4d}  exec branches {br:4d}  {dn}')


if __name__ == '__main__':
    main()
