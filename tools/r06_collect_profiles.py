#!/usr/bin/env python3
"""Turn what tools/prof_r06.sh left under gpurun_out/r06p/ into the tracked profiles/r06_* summaries (kernel stats at 1 280 / 2 048 /
512 chunks, log-mel stats, HBM traffic of a vocoder pass and a log-mel launch, the log-mel kernel's SQ counters and their reading).
    python tools/r06_collect_profiles.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, 'gpurun_out', 'r06p')
P = os.path.join(ROOT, 'profiles')
S = [sys.executable, os.path.join(ROOT, 'tools', 'summarize_profiles.py')]

for src, dst, cmd in (
        ('voc1280_kernel_stats.csv', 'r06_vocoder_kernel_stats', 'rocprofv3 --kernel-trace --stats -- python3 tools/probe_vocoder.py 10 1280 (10 HiFi-GAN passes of 1280 chunks x 12 frames; tools/prof_r06.sh)'),
        ('voc2560_kernel_stats.csv', 'r06_vocoder2560_kernel_stats', 'rocprofv3 --kernel-trace --stats -- python3 tools/probe_vocoder.py 10 2560 (10 passes at the render-group size the bench line quotes; tools/prof_r06.sh)'),
        ('voc512_kernel_stats.csv', 'r06_vocoder512_kernel_stats', 'rocprofv3 --kernel-trace --stats -- python3 tools/probe_vocoder.py 10 512 (tools/prof_r06.sh)'),
        ('logmel_kernel_stats.csv', 'r06_logmel_kernel_stats', 'rocprofv3 --kernel-trace --stats -- python3 tools/probe_logmel.py 10 128 (tools/prof_r06.sh)')):
    subprocess.check_call(S + ['stats', os.path.join(O, src), os.path.join(P, dst + '.md'), cmd])
if os.path.exists(os.path.join(O, 'encoder_kernel_stats.csv')):
    subprocess.check_call(S + ['stats', os.path.join(O, 'encoder_kernel_stats.csv'), os.path.join(P, 'r06_encoder_kernel_stats.md'),
                               'rocprofv3 --kernel-trace --stats -- python3 tools/probe_encoder.py 128 whisper_base (7 encoder passes of 128 windows: '
                               'k_attn_prefill2, the k_gemm_big8 products, k_layernorm<2, 4, false>; tools/prof_encoder.sh)'])
    if os.path.exists(os.path.join(O, 'encoder_probe.txt')):
        with open(os.path.join(P, 'r06_encoder_kernel_stats.md'), 'a') as f:
            f.write('\nUnprofiled on the same box (tools/probe_encoder.py 128 whisper_base x 3; tools/probe_attn_prefill.py 128 8 1500: form 0 = '
                    'k_attn_prefill, form 1 = k_attn_prefill2; tools/probe_layernorm.py: 192 000 x 512, IFH_LN_RPW = 1 / 4, and a copy of the same bytes):\n\n```\n'
                    + open(os.path.join(O, 'encoder_probe.txt')).read() + '```\n')
subprocess.check_call(['cp', os.path.join(O, 'voc1280_kernel_stats.csv'), os.path.join(P, 'r06_vocoder_kernel_stats.csv')])
subprocess.check_call(S + ['pmc', os.path.join(O, 'pmc_FETCH_SIZE'), os.path.join(O, 'pmc_WRITE_SIZE'), '3', os.path.join(P, 'r06_vocoder_pmc.json'), '1280', 'vocoder'])
subprocess.check_call(S + ['pmc', os.path.join(O, 'pmclm_FETCH_SIZE'), os.path.join(O, 'pmclm_WRITE_SIZE'), '3', os.path.join(P, 'r06_logmel_pmc.json'), '128', 'logmel'])

p = os.path.join(P, 'r06_vocoder_pmc.json')
d = json.load(open(p))
d['note'] = ('The x2 on FETCH_SIZE is the guide\'s rule for 16-byte-per-lane streaming reads.  The residual-block kernels (k_resblock_seq / _chain / _level) '
             'read their activation rows 8 bytes per lane, a width the guide calls uncalibrated: for k_resblock_seq<64, 11> the RAW counter already equals '
             'the bytes the launch reads (x 126 MB + out 126 MB + weights), so the corrected figure over-counts those kernels by up to 2x.  '
             'Counting the 8-byte-load kernels once and the 16-byte-load ones (k_igemm, k_conv_post) twice gives ~4.3 GB per pass.')
json.dump(d, open(p, 'w'), indent=1)

p = os.path.join(P, 'r06_logmel_pmc.json')
d = json.load(open(p))
c = json.load(open(os.path.join(O, 'logmel_sq_counters.json')))
d['sq_counters_per_launch'] = c
clk = c['GRBM_GUI_ACTIVE'] / 8.0
wave = c['SQ_WAVE_CYCLES']
d['sq_reading'] = {
    'shader_clocks_per_launch (GRBM_GUI_ACTIVE / 8 XCDs)': clk,
    'valu_busy_frac (SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs / clocks)': c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / clk,
    'wave_time_split (of SQ_WAVE_CYCLES)': {'active': c['SQ_ACTIVE_INST_ANY'] / wave, 'parked (s_waitcnt / barrier)': c['SQ_WAIT_ANY'] / wave,
                                            'issue stall': c['SQ_WAIT_INST_ANY'] / wave, 'of which LDS issue stall': c['SQ_WAIT_INST_LDS'] / wave},
    'lds_bank_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'],
    'command': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY (one pass) '
               'and GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (another), --kernel-trace only, -- python3 tools/probe_logmel.py 3 128; '
               'mean over the k_logmel_fft launches (tools/prof_r06.sh)'}
d['valu_frac'] = round(d['sq_reading']['valu_busy_frac (SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs / clocks)'], 4)
json.dump(d, open(p, 'w'), indent=1)
subprocess.check_call(S + ['stats', os.path.join(O, 'bench_kernel_stats.csv'), os.path.join(P, 'r06_bench_kernel_stats.md'),
                           'IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe '
                           '(C3, LSTM VAD in the throughput path; tools/prof_r06.sh; the final build of round 6)'])
subprocess.check_call(['cp', os.path.join(O, 'bench_kernel_stats.csv'), os.path.join(P, 'r06_bench_kernel_stats.csv')])
subprocess.check_call(['cp', os.path.join(O, 'bench_busy.txt'), os.path.join(P, 'r06_bench_timed_region_busy.txt')])
print('profiles/r06_* written')
