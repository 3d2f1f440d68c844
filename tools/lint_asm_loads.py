#!/usr/bin/env python3
"""Lint of a gfx950 assembly listing (hipcc -S): an inline-asm LDS read is asynchronous, but to hipcc its destination is written when
the asm statement ends -- under register pressure it may COPY or SPILL that register before the data has arrived (seen in seq.hip: a
bias register stored to scratch between its ds_read and the wait).  This walks every kernel and reports any instruction that reads a
register that a ds_read wrote while no `s_waitcnt lgkmcnt(N)` small enough has retired that read (LDS reads return in order).
    python3 tools/lint_asm_loads.py file.s [kernel-name-substring]"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def lint(name, body):
    pending = []                      # outstanding LDS ops in issue order: set of dest registers (empty for writes)
    bad = 0
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(';') or t.endswith(':') or t.startswith('.'):
            continue
        op, _, rest = t.partition(' ')
        toks = [x.strip() for x in re.split(r'[,\s]+', rest.split(';')[0]) if x.strip()]
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n < len(pending) else pending
                if n == 0:
                    pending = []
            continue
        if op in ('s_barrier',):
            continue
        live = set().union(*pending) if pending else set()
        if op.startswith('ds_read'):
            dst = regs(toks[0]) if toks else set()
            src = set().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
            if src & live:
                bad += 1
                print('%s: `%s` uses a register an LDS read has not delivered yet' % (name, t))
            pending.append(dst)
            continue
        if op.startswith('ds_write') or op.startswith('ds_'):
            src = set().union(*[regs(x) for x in toks]) if toks else set()
            if src & live:
                bad += 1
                print('%s: `%s` uses a register an LDS read has not delivered yet' % (name, t))
            pending.append(set())
            continue
        if op.startswith('s_load') or op.startswith('s_memtime') or op.startswith('s_buffer_load'):
            pending.append(set())      # scalar memory operations share lgkmcnt (they may return out of order: keep it conservative)
            continue
        used = set().union(*[regs(x) for x in toks]) if toks else set()
        if used & live:
            bad += 1
            print('%s: `%s` touches a register an LDS read has not delivered yet' % (name, t))
    return bad


def main():
    text = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    total, name, body = 0, None, []
    for ln in text:
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln)
            if ln.strip().startswith('s_endpgm'):
                if want in name:
                    total += lint(name[:90], body)
                name = None
    print('lint_asm_loads: %d hazard(s)' % total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
