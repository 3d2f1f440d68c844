"""GPU: the resident decode step (csrc/step.hip: ONE launch per SpeechT5 decoder step, clusters of workgroups walking blocks of
32 rows through every phase) against the launch chain it replaces (engines/speecht5.py:_decoder_step_ragged: ~55 launches per
step of HelloSippyRTPipe.infer's loop, HelloSippyTTSRT/HelloSippyRTPipe.py:196-229).  The launch chain is what the oracle pins
(tests/test_continuous_tts_gpu.py, test_nn_gpu.py::test_tts_infer_at_bench_batch_256_matches_oracle); the resident step has to
reproduce it BIT FOR BIT: frames, stop logits, positions, end flags and the appended K|V rows."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev):
    from infernos_amd.tts import HelloSippyRTPipe
    from infernos_amd.weights import synth_state_dict
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0), 'hifigan': synth_state_dict('hifigan', 0),
         'amendment': synth_state_dict('amendment', 0)}
    return HelloSippyRTPipe(dev, weights=W, processor=lambda **k: None, speaker_embeddings=[], output_sr=8000)


def _tensors(st):
    out = {}
    for k, v in vars(st).items():
        if torch.is_tensor(v):
            out[k] = v
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            for i, t in enumerate(v):
                out['%s[%d]' % (k, i)] = t
    return out


def _randomise(st, n, T, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    dev = st.pos.device
    R = st.R
    st.pos.copy_(torch.randint(0, 150, (R,), generator=g, dtype=torch.int32))
    st.pos[::7] = 0                                            # rows that start with this call
    st.active.copy_((torch.rand(R, generator=g) > 0.1).to(torch.uint8))
    st.enc_len.copy_(torch.randint(1, T + 1, (R,), generator=g, dtype=torch.int32))
    mm = torch.stack([torch.randint(0, 40, (R,), generator=g), torch.randint(100, 400, (R,), generator=g)], 1).to(torch.int32)
    st.minmax.copy_(mm)
    for kv in st.cross + st.self_kv:
        kv.copy_((torch.randn(kv.shape, generator=g) * 0.5).to(torch.bfloat16))
    for sp in st.spec:
        sp.copy_((torch.randn(sp.shape, generator=g) * 0.5).to(torch.bfloat16))
    masks = torch.randint(0, 2, (16, 2, 256), generator=g, dtype=torch.uint8).to(dev)
    return masks


@pytest.mark.parametrize('n,cw,wt', [(48, 16, False), (160, 8, False), (512, 16, False), (336, 5, False), (512, 16, True), (80, 32, True)])
def test_resident_step_is_bit_identical_to_the_launch_chain(built_lib, n, cw, wt):
    """wt: the write-through hand-off path (clusters spread over XCDs) forced on a cluster that shares one XCD"""
    from infernos_amd import _lib
    from infernos_amd.engines.speecht5 import TTSRaggedState, ragged_decoder_steps
    dev = _lib.require_device('cuda:0')
    pp = _model(dev)
    model = pp.model
    T = 64
    a = TTSRaggedState(model, max_rows=n, max_text=T)
    b = TTSRaggedState(model, max_rows=n, max_text=T)
    masks = _randomise(a, n, T, seed=n)
    ta, tb = _tensors(a), _tensors(b)
    for k in ta:
        tb[k].copy_(ta[k])
    model.resident_cw, model.resident_wt = cw, wt
    for call in range(2):                                      # two infer() calls: 32 steps, both frame-buffer parities
        ragged_decoder_steps(model, a, masks, n, use_graphs=False, resident=False)
        ragged_decoder_steps(model, b, masks, n, resident=True)
    torch.cuda.synchronize()
    err, _ = b.step_ctx.status()
    assert err == 0, 'a cluster of the resident step ran into its wait bound'
    assert int((a.pos[:n] != 0).sum()) > 0
    bad = [k for k in ta if not torch.equal(ta[k].view(torch.uint8), tb[k].view(torch.uint8))]
    assert not bad, 'tensors that differ from the launch chain: %s' % bad


def test_resident_step_clusters_and_xcds(built_lib):
    """diagnostic: which XCD each workgroup of a resident launch ran on (a cluster = the workgroups b with equal b % 8 within a
    group of 8 * cw: one XCD under round-robin dispatch).  Not a correctness condition -- recorded in the test output."""
    from infernos_amd import _lib
    from infernos_amd.engines.speecht5 import TTSRaggedState, ragged_decoder_steps
    dev = _lib.require_device('cuda:0')
    pp = _model(dev)
    model = pp.model
    n, cw = 256, 8
    st = TTSRaggedState(model, max_rows=n, max_text=64)
    masks = _randomise(st, n, 64, seed=3)
    model.resident_cw = cw
    ragged_decoder_steps(model, st, masks, n, resident=True)
    prog = next(iter(st.progs.values()))
    prog.run(st.step_ctx, cw=cw, debug_xcc=True)
    err, xcc = st.step_ctx.status(8 * cw)
    assert err == 0
    same = sum(1 for c in range(8) if len({xcc[c + 8 * m] for m in range(cw)}) == 1)
    print('XCC id per workgroup:', xcc, '-> clusters on one XCD: %d of 8' % same)
