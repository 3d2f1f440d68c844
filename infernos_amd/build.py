"""Build libinfernos_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build()."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libinfernos_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off',
         # the accumulator-tile loops must be fully unrolled (static register indices); LLVM's default
         # pragma-unroll budget (16K) is too small for the 16..24-tile epilogues -> tiles went to scratch
         '-mllvm', '-pragma-unroll-threshold=1000000']


# per-file additions: attn.hip's one-wave kernel (k_attn_prefill_few: 48 KB of LDS per wave, so hipcc plans for one wave per SIMD and
# 512 registers) got its MFMA results in AGPRs with a v_accvgpr copy around every vector instruction that touches them (144 in the tile
# loop); the VGPR form of the MFMAs is selected for that file (no kernel in it keeps accumulators in AGPRs on purpose)
EXTRA_FLAGS = {'attn.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def flags_for(src):
    return FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(HERE, '..', 'include', 'infernos_hip.h'), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_stamp():
    """the compile flags of every file as text: a changed flag (EXTRA_FLAGS, FLAGS) rebuilds, not only a changed source"""
    return '\n'.join('%s: %s' % (os.path.basename(s), ' '.join(flags_for(s))) for s in sources()) + '\n'


def build(force=False, verbose=True, jobs=None):
    """Compile what is out of date and link.  Safe to call from several processes at once (the ranks of a multi-GPU job all call
    it): one holds build/.lock and compiles, the others wait for it and find the library up to date."""
    import fcntl
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, jobs, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs, objdir):
    stamp_path = os.path.join(objdir, 'flags.txt')
    stamp = _flags_stamp()
    old_stamp = open(stamp_path).read() if os.path.exists(stamp_path) else None
    if old_stamp != stamp:
        # built with other flags: only on a tree whose objects are older than this check (no stamp yet) are they trusted
        force = force or old_stamp is not None
    if not force and not needs_build():
        if old_stamp is None:
            open(stamp_path, 'w').write(stamp)
        return OUT
    procs = []
    objs = []
    jobs = jobs or min(6, os.cpu_count() or 1)
    srcs = sources()
    hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h')) +
                [os.path.join(HERE, '..', 'include', 'infernos_hip.h')])

    def flush(limit):
        while len(procs) > limit:
            p, src = procs.pop(0)
            if p.wait() != 0:
                raise RuntimeError('hipcc failed for %s' % src)
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [HIPCC] + flags_for(src) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        flush(jobs - 1)
    flush(0)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp_path, 'w').write(stamp)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
