"""CPU: the kernels that read LDS through inline asm (hand-counted waits) must not have their asm destinations touched before the
data has arrived.  To hipcc an asm load's destination is written when the statement ends; under register pressure it copied or
spilled such a register right behind the read (round 5, csrc/seq.hip: a bias register and activation fragments stored to scratch
before their data had landed -- silently wrong sums on the GPU).  tools/lint_asm_loads.py walks the assembly listing for exactly
that; this test compiles the listing of csrc/seq.hip for gfx950 (hipcc cross-compiles without a GPU) and requires zero findings."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc_missing():
    from infernos_amd import build as b
    return not os.path.exists(b.HIPCC)


pytestmark = pytest.mark.skipif(_hipcc_missing(), reason='hipcc is not installed here: nothing to compile the listings with')


def test_seq_kernels_do_not_touch_asm_read_destinations_early(tmp_path):
    from infernos_amd import build as b
    out = str(tmp_path / 'seq.s')
    flags = [f for f in b.FLAGS if f not in ('-fPIC', '-Wall')]
    subprocess.check_call([b.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(b.CSRC, 'seq.hip')],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'lint_asm_loads.py'), out], capture_output=True, text=True)
    assert r.returncode == 0 and 'lint_asm_loads: 0 hazard(s)' in r.stdout, r.stdout[-2000:]


def test_lint_finds_a_planted_hazard(tmp_path):
    """the checker itself: a register spilled between its ds_read and the wait is reported, the same code with the wait first is not"""
    bad = tmp_path / 'bad.s'
    bad.write_text('_Zkern:\n\tds_read_b128 v[8:11], v20 offset:0\n\tscratch_store_dwordx4 off, v[8:11], off offset:16\n'
                   '\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[4:7], v[0:3]\n\ts_endpgm\n')
    good = tmp_path / 'good.s'
    good.write_text('_Zkern:\n\tds_read_b128 v[8:11], v20 offset:0\n\tds_read_b128 v[12:15], v20 offset:64\n\ts_waitcnt lgkmcnt(1)\n'
                    '\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[4:7], v[0:3]\n\ts_waitcnt lgkmcnt(0)\n'
                    '\tscratch_store_dwordx4 off, v[12:15], off offset:16\n\ts_endpgm\n')
    lint = os.path.join(ROOT, 'tools', 'lint_asm_loads.py')
    assert subprocess.run([sys.executable, lint, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, lint, str(good)], capture_output=True).returncode == 0


def test_dma_queue_kernels_use_no_scratch(tmp_path):
    """The kernels that keep DMA pieces in flight behind counted `s_waitcnt vmcnt` (gemm_big8 / gemm_m64d / the attention kernels of
    llm.hip) must not touch scratch: a scratch load or store is a vector-memory operation in the same in-order queue, and hipcc puts an
    `s_waitcnt vmcnt(0)` behind every scratch load -- round 5: four `uint4` structs of k_gemm_big8's residual epilogue stayed a stack
    slot (no register was spilled, `.vgpr_spill_count` read 0) and every launch with a residual waited 66 us of 186 on them.  Checked on
    the compiled listing: no kernel of these files has a private segment or a scratch instruction."""
    import re
    from infernos_amd import build as b
    flags = [f for f in b.FLAGS if f not in ('-fPIC', '-Wall')]
    for name in ('gemm_big8', 'gemm_m64d', 'llm'):
        out = str(tmp_path / (name + '.s'))
        subprocess.check_call([b.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(b.CSRC, name + '.hip')],
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
        sizes = [int(v) for v in re.findall(r'^\s+\.private_segment_fixed_size:\s+(\d+)', txt, re.M)]
        assert sizes and max(sizes) == 0, (name, sizes)
        assert 'scratch_load' not in txt and 'scratch_store' not in txt, name


def test_batched_load_kernels_stay_batched(tmp_path):
    """k_attn_prefill2 and k_beam_rowtop were rewritten so that their loads leave together (round 5: per-lane `if`s around loads had
    made hipcc wait for each one separately; tools/scan_serial_loads.py).  On the compiled listings: at most a handful of
    load + `s_waitcnt vmcnt(0)` pairs are left in k_beam_rowtop (71 before), k_attn_prefill's bias path has none of its sixteen, and
    k_attn_prefill2 -- whose DMA queue a scratch access would drain -- has neither a private segment nor AGPR copies (hipcc used 32
    AGPRs as spill space until the launch bounds said three waves per SIMD)."""
    import re
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from scan_serial_loads import scan
    from infernos_amd import build as b
    outs = {}
    for name in ('attn', 'beam'):
        outs[name] = str(tmp_path / (name + '.s'))
        flags = [f for f in b.flags_for(name + '.hip') if f not in ('-fPIC', '-Wall')]
        subprocess.check_call([b.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', outs[name], os.path.join(b.CSRC, name + '.hip')],
                              stderr=subprocess.DEVNULL)
    beam = {k: v for k, v in scan(outs['beam']).items() if 'k_beam_rowtop' in k}
    assert beam and all(v[1] <= 10 for v in beam.values()), beam
    attn = scan(outs['attn'])
    v1 = [v for k, v in attn.items() if 'k_attn_prefillENS' in k]
    assert v1 and v1[0][1] <= 4, v1
    txt = open(outs['attn']).read()
    body = txt[txt.index('_ZN3ifh15k_attn_prefill2'):]
    body = body[:body.index('s_endpgm')]
    assert 'v_accvgpr' not in body and 'scratch_' not in body
    few = txt[txt.index('_ZN3ifh18k_attn_prefill_few'):]
    few = few[:few.index('s_endpgm')]
    assert 'v_accvgpr' not in few and 'scratch_' not in few        # (the one-wave kernel: build.EXTRA_FLAGS selects the MFMAs' VGPR form)
    idx = [m.start() for m in re.finditer(r'\.name:\s+_ZN3ifh15k_attn_prefill2', txt)]
    assert idx
    m = re.search(r'\.private_segment_fixed_size:\s+(\d+)', txt[idx[0]:])       # (metadata keys are sorted: the size follows the name)
    assert m is not None and int(m.group(1)) == 0


def test_layernorm_loads_leave_together(tmp_path):
    """k_layernorm (nn.hip): every row / residual / gamma / beta load of a wave is issued before the first wait -- as per-lane
    `if (e < D)` blocks each load had an `s_waitcnt vmcnt(0)` of its own and the encoder's 192 000 x 512 stream ran at 3.7 TB/s instead
    of a copy's 5.2 (round 5).  On the listing: no instantiation has more than one load + vmcnt(0) pair."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from scan_serial_loads import scan
    from infernos_amd import build as b
    out = str(tmp_path / 'nn.s')
    flags = [f for f in b.flags_for('nn.hip') if f not in ('-fPIC', '-Wall')]
    subprocess.check_call([b.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(b.CSRC, 'nn.hip')],
                          stderr=subprocess.DEVNULL)
    ln = {k: v for k, v in scan(out).items() if 'k_layernorm' in k}
    assert len(ln) >= 8 and all(v[0] >= 6 and v[1] <= 1 for v in ln.values()), ln
