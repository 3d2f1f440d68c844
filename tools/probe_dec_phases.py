"""Where a k_gemm_dec launch spends its time: shader-clock stamps of wave 0 of block (0, 0) (library built with -DIFH_DEC_PROF, see
profiles/NOTES.md), for the SpeechT5 decode-step launch forms at M rows.  IFH_LIB_PATH=<libprof.so> python tools/probe_dec_phases.py [M]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 384
D, FFN = 768, 3072
lib = _lib.lib()
lib.ifh_debug_dec_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, BF)
stats = torch.zeros((3, 1024, 2), dtype=torch.int64, device=dev)
att, x0, wo, bo = rnd(M, D + 8), rnd(M, D + 8), rnd(D, D, sc=1 / 28), torch.zeros(D, device=dev)
g1, b1 = torch.ones(D), torch.zeros(D)
g1d, b1d = g1.to(dev), b1.to(dev)
x256, w256, b256 = None, None, None
t1 = torch.zeros(M, D + 8, dtype=BF, device=dev)
wq, cq2, cq1 = ops.w_linear_ln(torch.randn(D, D, generator=g) / 28, torch.zeros(D), g1, b1, dev)
w1, c12, c11 = ops.w_linear_ln(torch.randn(FFN, D, generator=g) / 28, torch.zeros(FFN), g1, b1, dev)
w2 = rnd(D, FFN, sc=1 / 55)
q = torch.zeros(M, D + 8, dtype=BF, device=dev)
ff = torch.zeros(M, FFN + 8, dtype=BF, device=dev)
x256, w256, b256 = att[:, :256].contiguous(), wo[:256, :256].contiguous(), bo[:256].contiguous()
forms = {
    'wo  (resid, stats_out)': lambda: ops.linear(att, wo, bo, t1, rows=M, k=D, n=D, resid=x0, stats_out=stats, stats_off=0, ln_dim=D, lda=D + 8, ldc=D + 8, resid_ld=D + 8),
    'cq  (aln)': lambda: ops.linear(t1, wq, cq2, q, rows=M, k=D, n=D, aln=(stats, 0, cq1), ln_dim=D, lda=D + 8, ldc=D + 8),
    'ff1 (aln, gelu)': lambda: ops.linear(t1, w1, c12, ff, rows=M, k=D, n=FFN, act=2, aln=(stats, 0, c11), ln_dim=D, lda=D + 8, ldc=FFN + 8),
    'ff2 (K=3072, resid, rln, stats_out)': lambda: ops.linear(ff, w2, bo, t1, rows=M, k=FFN, n=D, resid=x0, rln=(stats, 0, g1d, b1d), stats_out=stats, stats_off=2048, ln_dim=D, lda=FFN + 8, ldc=D + 8, resid_ld=D + 8),
    'plain decode_step (K=256, relu)': lambda: ops.linear(x256, w256, b256, q, rows=M, k=256, n=256, act=1, ldc=D + 8, decode_step=True),
}
names = ['entry -> addresses', 'issue first chunk', 'epilogue operand loads', 'wait + LDS store + barrier', 'chunk loop', 'epilogue']
for name, fn in forms.items():
    fn(); torch.cuda.synchronize()
    lib.ifh_debug_dec_prof(None, 1)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(50):
            fn()
    gr.replay(); torch.cuda.synchronize()
    lib.ifh_debug_dec_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.ifh_debug_dec_prof(out, 0)
    n = max(1, out[7])
    tot = sum(out[i] for i in range(6)) / n
    print('%-38s M=%d: %.1f us per launch; wave 0 of block 0: %.0f clocks = %s' %
          (name, M, e0.elapsed_time(e1) * 1e3 / 50, tot, ' | '.join('%s %.0f' % (names[i], out[i] / n) for i in range(6))))
