"""CPU: the oracle of the recurrent VAD network (oracle/nn.py:vadnet, the checker of csrc/vadnet.hip) against torch.nn's own Conv1d /
LSTM / Linear modules holding the same weights -- so the stand-in network is the textbook one (gate order, bias handling, state
carry), and the weight blob the kernel reads has the size the library expects.  The network is Silero-v3.1-SHAPED, not Silero
(Core/VAD/SileroVAD.py:44-45 loads a third-party file that is not obtainable offline): parity against Silero is unpinned."""
import torch


def test_vadnet_oracle_matches_torch_modules():
    from infernos_amd.weights import pack_vadnet, synth_vadnet
    from oracle.nn import vadnet
    sd = synth_vadnet(3)
    conv1, conv2 = torch.nn.Conv1d(1, 32, 128, stride=64), torch.nn.Conv1d(32, 64, 3, stride=2, padding=1)
    lstm, out = torch.nn.LSTM(64, 64, num_layers=2, batch_first=True), torch.nn.Linear(64, 1)
    with torch.no_grad():
        conv1.weight.copy_(sd['conv1.weight']); conv1.bias.copy_(sd['conv1.bias'])
        conv2.weight.copy_(sd['conv2.weight']); conv2.bias.copy_(sd['conv2.bias'])
        for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l1', 'weight_hh_l1', 'bias_ih_l1', 'bias_hh_l1'):
            getattr(lstm, k).copy_(sd['lstm.' + k])
        out.weight.copy_(sd['out.weight']); out.bias.copy_(sd['out.bias'])
    g = torch.Generator().manual_seed(1)
    B = 5
    h, c = torch.zeros(2, B, 64), torch.zeros(2, B, 64)
    hm, cm = h.clone(), c.clone()
    for step in range(3):                                   # three consecutive windows: the state is carried
        x = torch.randn(B, 768, generator=g) * 0.3
        p, h, c = vadnet(x, sd, h, c)
        with torch.no_grad():
            f2 = torch.relu(conv2(torch.relu(conv1(x[:, None, :]))))               # [B, 64, 6]
            y, (hm, cm) = lstm(f2.transpose(1, 2), (hm, cm))
            pm = torch.sigmoid(out(y))[:, :, 0].mean(1)
        assert torch.allclose(p, pm, atol=1e-6) and torch.allclose(h, hm, atol=1e-6) and torch.allclose(c, cm, atol=1e-6)
        assert 0.0 < float(p.min()) and float(p.max()) < 1.0
    blob = pack_vadnet(sd)
    assert blob.dtype == torch.float32 and blob.numel() == 128 * 32 + 32 + 3 * 32 * 64 + 64 + 2 * (2 * 64 * 256 + 256) + 64 + 1


def test_distilled_vadnet_follows_the_energy_rule_on_call_audio():
    """The weights the throughput path runs (infernos_amd/vadnet_distilled.npz, fitted by tools/train_vadnet.py) make the network a
    usable detector: on the benchmark's own call audio, seen through G.711 as the serving path sees it and with the LSTM state carried
    window to window, its decisions at the FSM's two thresholds (0.5 and 0.5 - 0.15, SileroVADUtils.py:105-130) are those of the
    energy rule of ifh_vad_energy_prob.  (Not Silero's weights -- unobtainable offline: parity unpinned against them.)"""
    import numpy as np
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import load_vadnet_distilled
    from oracle import dsp as odsp
    from oracle.nn import vadnet
    sd = load_vadnet_distilled()
    N = 6
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    pcm = odsp.g711_decode(odsp.g711_encode(x))
    W = pcm.shape[1] // 768
    win = torch.from_numpy(pcm[:, :W * 768].reshape(N, W, 768))
    h, c = torch.zeros(2, N, 64), torch.zeros(2, N, 64)
    ps = []
    with torch.no_grad():
        for w in range(W):
            p, h, c = vadnet(win[:, w], sd, h, c)
            ps.append(p)
    ps = torch.stack(ps, 1).numpy()
    e = (win.double() ** 2).mean(2)
    pe = torch.sigmoid(0.5 * (10 * torch.log10(e + 1e-10) + 30)).numpy()
    assert ((ps > 0.5) == (pe > 0.5)).mean() >= 0.97
    assert ((ps > 0.35) == (pe > 0.35)).mean() >= 0.97
    assert (ps[:, :8] < 0.35).all() and (ps[:, -8:] < 0.35).all()        # the second of silence at either end stays silence
