#!/usr/bin/env python3
"""tests/golden/qwen2_c5.npz: what tests/test_llm_gpu.py::test_qwen2_config5_share_at_full_depth_matches_oracle used to compute on
the GPU box's host in every run (78 s of a 1 200 s step limit) -- the fp32 ORACLE (oracle/nn.py:qwen2_forward, pinned to
transformers' Qwen2ForCausalLM on the small configurations: tests/test_oracle_nn.py) at Qwen2.5-1.5B's full shape (28 layers,
seeded weights synth_state_dict('qwen2_1p5b', 4)) on 8 of the 64 ragged sessions of BASELINE configuration 5's per-GPU share:
greedy tokens, and the logits at the last prompt position and the 4 generated ones.  The full rows are 8 x 5 x 151 936 floats
(24 MB), so the fixture keeps, per (session, position): the logits at 2 048 fixed random vocabulary columns (the device's rel-L2
is taken over those columns: an unbiased estimate of the full-row figure, +-1.6 %), and the oracle's top-2 columns and values.
Also the bar's ingredients: the oracle run with weights and activations in bfloat16 (the way transformers runs the model in that
dtype) against its fp32 run on two of the sessions, over the FULL rows.   python tools/gen_golden_c5.py   (CPU, a few minutes)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infernos_amd.weights import QWEN2_CONFIGS, synth_state_dict  # noqa: E402
from oracle import nn as onn  # noqa: E402

B, N_NEW, NCOL = 64, 5, 2048
ROWS = [0, 9, 18, 27, 36, 45, 54, 63]


def prompts_of(cfg):
    g = torch.Generator().manual_seed(23)
    return [torch.randint(10, cfg['vocab'] - 10, (184 + (i * 5) % 9,), generator=g).tolist() for i in range(B)]


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    cfg = QWEN2_CONFIGS['qwen2_1p5b']
    sd = synth_state_dict('qwen2_1p5b', 4)
    prompts = prompts_of(cfg)
    cols = torch.from_numpy(np.sort(np.random.default_rng(5).choice(cfg['vocab'], NCOL, replace=False)).astype(np.int64))
    o_new, sub, top_i, top_v, full0 = [], [], [], [], {}
    with torch.no_grad():
        for j, i in enumerate(ROWS):
            caches = [{} for _ in range(cfg['layers'])]
            cur = onn.qwen2_forward(sd, cfg, torch.tensor([prompts[i]]), 0, caches)[0, -1].clone()
            lg, new = [cur], []
            for s_ in range(N_NEW):
                t = int(cur.argmax())
                new.append(t)
                if s_ + 1 == N_NEW:
                    break
                cur = onn.qwen2_forward(sd, cfg, torch.tensor([[t]]), len(prompts[i]) + s_, caches)[0, -1].clone()
                lg.append(cur)
            lg = torch.stack(lg)                                       # [N_NEW, V]
            o_new.append(new)
            sub.append(lg[:, cols])
            tk = lg.topk(2, dim=1)
            top_i.append(tk.indices)
            top_v.append(tk.values)
            if j in (0, 4):
                full0[j] = lg[0].clone()
            print('session %d: tokens %s' % (i, new), flush=True)
        sd16 = onn._cast(sd, torch.bfloat16)
        e16 = []
        for j in (0, 4):
            lg16 = onn.qwen2_forward(sd16, cfg, torch.tensor([prompts[ROWS[j]]]), 0, [{} for _ in range(cfg['layers'])])[0, -1]
            e16.append(rel_l2(lg16, full0[j]))
    out = os.path.join(ROOT, 'tests', 'golden')
    np.savez_compressed(os.path.join(out, 'qwen2_c5.npz'), cols=cols.numpy().astype(np.int32), ref_sub=torch.stack(sub).numpy().astype(np.float32),
                        top_idx=torch.stack(top_i).numpy().astype(np.int32), top_val=torch.stack(top_v).numpy().astype(np.float32),
                        o_new=np.array(o_new, np.int32))
    json.dump({'rows': ROWS, 'sessions': B, 'positions': N_NEW, 'columns': NCOL, 'oracle_bf16_rel_l2_full_rows': e16,
               'weights': "synth_state_dict('qwen2_1p5b', 4)", 'prompts': 'torch.Generator().manual_seed(23), lengths 184 + (i * 5) % 9',
               'generator': 'tools/gen_golden_c5.py'}, open(os.path.join(out, 'qwen2_c5_meta.json'), 'w'), indent=1)
    print('oracle in bf16 vs fp32 at this depth:', e16)


if __name__ == '__main__':
    main()
