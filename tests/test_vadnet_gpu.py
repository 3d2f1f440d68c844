"""GPU: the recurrent VAD network (csrc/vadnet.hip, ifh_vadnet_prob) against its oracle (oracle/nn.py:vadnet, pinned to torch.nn's
LSTM by tests/test_vadnet_oracle.py), and through the reference's state plumbing: VADIteratorB / SileroVADWorker inject the
per-channel state before a call and save it after (Core/VAD/SileroVADUtils.py:21-26,99,131), whatever sub-batch a channel lands in.
Silero itself (SileroVAD.py:44-45) is not obtainable offline: the network is shaped like it, parity against it is unpinned."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 37, 128])
def test_vadnet_matches_oracle_over_consecutive_windows(built_lib, n):
    from infernos_amd import _lib
    from infernos_amd.vad import RecurrentVADModel
    from oracle.nn import vadnet
    dev = _lib.require_device('cuda:0')
    m = RecurrentVADModel(dev, seed=2)
    assert _lib.lib().ifh_vadnet_weight_floats() == m.blob.numel()
    g = torch.Generator().manual_seed(n)
    h, c = torch.zeros(2, n, 64), torch.zeros(2, n, 64)
    for step in range(4):
        x = torch.randn(n, 768, generator=g) * (0.05 + 0.3 * step)
        p_ref, h, c = vadnet(x, m.sd, h, c)
        p = m(x.to(dev), 8000)
        assert torch.allclose(p.cpu(), p_ref, atol=2e-6), float((p.cpu() - p_ref).abs().max())
        assert torch.allclose(m._c._h.cpu(), h, atol=2e-6) and torch.allclose(m._c._c.cpu(), c, atol=2e-6)


def test_vad_worker_carries_the_network_state_per_channel(built_lib):
    """Six channels whose windows arrive in changing sub-batches (a channel twice in one batch, channels missing from a batch): every
    channel's state after the run equals the oracle run over that channel's own windows in order."""
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.vad import RecurrentVADModel, SileroVADWorker, VADChannel
    from oracle.nn import vadnet
    dev = _lib.require_device('cuda:0')
    model = RecurrentVADModel(dev, seed=5)
    w = SileroVADWorker(dev, input_sr=8000, model=model, max_channels=4)
    sink = lambda *a, **k: None
    chans = [VADChannel(sink, sink, None, dev) for _ in range(6)]
    g = torch.Generator().manual_seed(11)
    seen = {i: [] for i in range(6)}
    for rnd in range(5):
        order = torch.randperm(6, generator=g).tolist()[: 3 + rnd % 3] + ([rnd % 6] if rnd % 2 else [])
        wis = []
        for i in order:
            a = torch.randn(768, generator=g) * 0.01            # quiet: the FSM stays idle whatever the seeded network says
            seen[i].append(a)
            wis.append((chans[i], AudioChunk(a, 8000)))
        model._c._h = None                                      # force the plumbing to bring the state in
        try:
            w.process_batch(wis)
        except AssertionError:
            pytest.skip('the seeded network triggered the FSM invariant of SileroVAD.py:89 on noise')
    for i, ch in enumerate(chans):
        h, c = torch.zeros(2, 1, 64), torch.zeros(2, 1, 64)
        for a in seen[i]:
            _, h, c = vadnet(a[None], model.sd, h, c)
        if not seen[i]:
            continue
        assert torch.allclose(ch.state.model_state[0].cpu(), h[:, 0], atol=5e-6)
        assert torch.allclose(ch.state.model_state[1].cpu(), c[:, 0], atol=5e-6)
