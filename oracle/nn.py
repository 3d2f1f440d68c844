"""CPU restatement (plain PyTorch, fp32 by default) of the neural graphs on the speech path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function restates, from the
published architecture, what the reference executes through third-party engines:

  * SpeechT5 text encoder / speech decoder / postnet, HiFi-GAN, AmendmentNetwork1 and the
    streaming loop of HelloSippyTTSRT/HelloSippyRTPipe.py:191-259 (+ HelloSippyRT.py:163-237),
    engines = transformers (modeling_speecht5.py, v5.15.0 in the build container);
  * Whisper encoder/decoder + greedy search as reached from
    Cluster/InfernSTTWorker.py:77-107 (torch path; the default CTranslate2 int8 engine is
    absent -> "parity unpinned" at that boundary).

Pinning: tests/test_oracle_nn.py checks these against fixtures produced by running the
reference's own HelloSippyRTPipe.infer()/unbatch_and_dispatch() and
InfernSTTWorker.process_batch() on the HF modules with the same seeded weights
(tools/gen_golden_nn.py).  All functions take an HF-format state dict `sd`.
"""
import math

import torch
import torch.nn.functional as F


def _cast(sd, dtype):
    if dtype == torch.float32:
        return sd
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def linear(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def layer_norm(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.size(-1),), sd[p + '.weight'], sd[p + '.bias'], eps)


def scaled_pe(max_len, dim):
    pe = torch.zeros(max_len, dim)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.int64).float() * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# =========================================================================================
# SpeechT5
# =========================================================================================
def mha(sd, p, x, kv, nheads, key_mask=None, rel_bias_table=None, cache=None):
    """SpeechT5Attention (modeling_speecht5.py:863-985): q pre-scaled by hd^-0.5, optional
    relative-position bias q.pe_k[clip(i-j)], additive key padding mask, softmax, out_proj.
    cache: dict with 'k','v' [B,H,S,hd] to append to (self-attn) or reuse (cross-attn)."""
    B, T, D = x.shape
    hd = D // nheads
    q = linear(sd, p + '.q_proj', x) * (hd ** -0.5)
    if cache is not None and cache.get('frozen'):
        k, v = cache['k'], cache['v']
    else:
        k = linear(sd, p + '.k_proj', kv).view(B, -1, nheads, hd).transpose(1, 2)
        v = linear(sd, p + '.v_proj', kv).view(B, -1, nheads, hd).transpose(1, 2)
        if cache is not None:
            if 'k' in cache and not cache.get('cross'):
                k = torch.cat([cache['k'], k], dim=2)
                v = torch.cat([cache['v'], v], dim=2)
            cache['k'], cache['v'] = k, v
            if cache.get('cross'):
                cache['frozen'] = True
    q = q.view(B, T, nheads, hd).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if rel_bias_table is not None:
        S = k.size(2)
        pos = torch.arange(T)[:, None] - torch.arange(S)[None, :]
        half = rel_bias_table.size(0) // 2
        pos = pos.clamp(-half, half - 1) + half
        pe = rel_bias_table[pos]                       # [T,S,hd]
        w = w + torch.einsum('bhtd,tsd->bhts', q, pe)
    if key_mask is not None:
        w = w + (1.0 - key_mask[:, None, None, :].to(w.dtype)) * torch.finfo(w.dtype).min
    a = torch.softmax(w, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, T, D)
    return linear(sd, p + '.out_proj', o)


def t5_encoder(sd, input_ids, attention_mask, dtype=torch.float32):
    """SpeechT5EncoderWithTextPrenet (modeling_speecht5.py:764-780,1212-1322): embedding +
    alpha*PE, LayerNorm, 12 post-LN layers with relative position bias."""
    sd = _cast(sd, dtype)
    P = 'speecht5.encoder.'
    x = F.embedding(input_ids, sd[P + 'prenet.embed_tokens.weight'])
    pe = scaled_pe(450, x.size(-1)).to(x.dtype)
    x = x + sd[P + 'prenet.encode_positions.alpha'] * pe[: x.size(1)]
    W = P + 'wrapped_encoder.'
    x = layer_norm(sd, W + 'layer_norm', x)
    rel = sd[W + 'embed_positions.pe_k.weight']
    i = 0
    while (W + 'layers.%d.attention.q_proj.weight' % i) in sd:
        L = W + 'layers.%d.' % i
        x = layer_norm(sd, L + 'layer_norm', x + mha(sd, L + 'attention', x, x, 12, attention_mask, rel))
        h = linear(sd, L + 'feed_forward.output_dense', F.gelu(linear(sd, L + 'feed_forward.intermediate_dense', x)))
        x = layer_norm(sd, L + 'final_layer_norm', x + h)
        i += 1
    return x


def t5_prenet_last(sd, frame, t, speaker, mask1, mask2):
    """Last time row of SpeechT5SpeechDecoderPrenet (modeling_speecht5.py:646-697) at position t;
    mask1/mask2: the Bernoulli(0.5) keep-masks [256] of that row (shared across the batch)."""
    P = 'speecht5.decoder.prenet.'
    h = F.relu(linear(sd, P + 'layers.0', frame))
    h = torch.where(mask1[None, :] == 1, h, torch.zeros_like(h)) * 2.0
    h = F.relu(linear(sd, P + 'layers.1', h))
    h = torch.where(mask2[None, :] == 1, h, torch.zeros_like(h)) * 2.0
    h = linear(sd, P + 'final_layer', h)
    pe = scaled_pe(4000, h.size(-1)).to(h.dtype)
    h = h + sd[P + 'encode_positions.alpha'] * pe[t]
    spk = F.normalize(speaker)
    h = torch.cat([h, spk], dim=-1)
    return F.relu(linear(sd, P + 'speaker_embeds_layer', h))


def t5_decoder_layers(sd, x, enc, enc_mask, caches):
    """6 post-LN decoder layers on one query token (modeling_speecht5.py:1070-1160).
    caches: list per layer of {'self': {...}, 'cross': {'cross': True}}."""
    W = 'speecht5.decoder.wrapped_decoder.'
    x = x[:, None, :]
    i = 0
    while (W + 'layers.%d.self_attn.q_proj.weight' % i) in sd:
        L = W + 'layers.%d.' % i
        c = caches[i]
        x = layer_norm(sd, L + 'self_attn_layer_norm', x + mha(sd, L + 'self_attn', x, x, 12, None, None, c['self']))
        x = layer_norm(sd, L + 'encoder_attn_layer_norm',
                       x + mha(sd, L + 'encoder_attn', x, enc, 12, enc_mask, None, c['cross']))
        h = linear(sd, L + 'feed_forward.output_dense', F.gelu(linear(sd, L + 'feed_forward.intermediate_dense', x)))
        x = layer_norm(sd, L + 'final_layer_norm', x + h)
        i += 1
    return x[:, 0]


def t5_postnet(sd, mel):
    """speech_decoder_postnet.postnet (modeling_speecht5.py:700-761): 5 x (conv k5, BatchNorm eval,
    tanh except last), residual."""
    P = 'speech_decoder_postnet.layers.'
    y = mel.transpose(1, 2)
    for i in range(5):
        y = F.conv1d(y, sd[P + '%d.conv.weight' % i], None, padding=2)
        b = P + '%d.batch_norm.' % i
        y = F.batch_norm(y, sd[b + 'running_mean'], sd[b + 'running_var'], sd[b + 'weight'], sd[b + 'bias'], False, 0.0, 1e-5)
        if i < 4:
            y = torch.tanh(y)
    return mel + y.transpose(1, 2)


def hifigan(sd, mel, dtype=torch.float32):
    """SpeechT5HifiGan.forward (modeling_speecht5.py:2954-3064); mel [N,T,80] -> [N,256*T]."""
    sd = _cast(sd, dtype)
    x = ((mel.to(dtype) - sd['mean']) / sd['scale']).transpose(2, 1)
    x = F.conv1d(x, sd['conv_pre.weight'], sd['conv_pre.bias'], padding=3)
    for i in range(4):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd['upsampler.%d.weight' % i], sd['upsampler.%d.bias' % i], stride=4, padding=2)
        acc = None
        for j, k in enumerate((3, 7, 11)):
            r = x
            R = 'resblocks.%d.' % (i * 3 + j)
            for d_i, d in enumerate((1, 3, 5)):
                h = F.leaky_relu(r, 0.1)
                h = F.conv1d(h, sd[R + 'convs1.%d.weight' % d_i], sd[R + 'convs1.%d.bias' % d_i], padding=(k * d - d) // 2, dilation=d)
                h = F.leaky_relu(h, 0.1)
                h = F.conv1d(h, sd[R + 'convs2.%d.weight' % d_i], sd[R + 'convs2.%d.bias' % d_i], padding=(k - 1) // 2)
                r = h + r
            acc = r if acc is None else acc + r
        x = acc / 3
    x = F.leaky_relu(x)                     # default slope 0.01 (modeling_speecht5.py:3058)
    x = torch.tanh(F.conv1d(x, sd['conv_post.weight'], sd['conv_post.bias'], padding=3))
    return x.squeeze(1)


def amendment(sd, mel, audio, dtype=torch.float32):
    """AmendmentNetwork1.forward (HelloSippyTTSRT/HelloSippyRT.py:219-237), including the
    `.view` (not transpose) reinterpretations of mel [N,12,80]->[N,80,12] and audio
    [N,3072]->[N,256,12]."""
    sd = _cast(sd, dtype)
    N = audio.size(0)
    mel = mel.to(dtype).contiguous()
    audio = audio.to(dtype).contiguous()
    T = mel.size(-1)                                      # 80
    a = audio.view(N, 256, -1)
    m = mel.view(N, T, -1)
    xm = F.conv1d(m, sd['conv_pre_m.weight'], sd['conv_pre_m.bias'], padding=1)
    xa = F.conv1d(a, sd['conv_pre_a.weight'], sd['conv_pre_a.bias'], padding=1)
    x = torch.cat((xm, xa), dim=1)
    for i in range(2):
        x = F.leaky_relu(x, 0.01)
        x = F.conv_transpose1d(x, sd['upsampler.%d.weight' % i], sd['upsampler.%d.bias' % i], stride=4, padding=2)
    r = x
    h = F.leaky_relu(x, 0.01)
    h = F.conv1d(h, sd['resblock.conv1.weight'], sd['resblock.conv1.bias'], padding=1)
    h = F.leaky_relu(h, 0.01)
    h = F.conv1d(h, sd['resblock.conv2.weight'], sd['resblock.conv2.bias'], padding=3, dilation=3)
    x = h + r
    x = F.leaky_relu(x, 0.01)
    x = F.conv1d(x, sd['post_conv.weight'], sd['post_conv.bias'], stride=24)      # [N,256,8]
    x = F.leaky_relu(x, 0.01).reshape(N, -1)
    return torch.tanh(audio[:, 512:-512] * x)


class TTSState:
    """HelloSippyPipeStateBatched (HelloSippyRTPipe.py:81-121) restated."""

    def __init__(self, sd, input_ids, attention_mask, speakers, dtype=torch.float32):
        B = input_ids.size(0)
        self.enc_mask = attention_mask
        self.enc = t5_encoder(sd, input_ids, attention_mask, dtype)
        self.speakers = speakers.to(dtype)
        T = input_ids.size(1)
        self.maxlen = int(T * 20.0 / 2)
        self.minlen = 0
        self.last_frame = torch.zeros(B, 80, dtype=dtype)        # output_sequence[:, -1]
        self.caches = None
        self.pre_frames = torch.zeros(B, 4, 80, dtype=dtype)
        self.starts_at = torch.full((B,), 1, dtype=torch.long)   # post_nframes // reduction_factor
        self.ends_at = torch.full((B,), -1, dtype=torch.long)
        self.idx = 0
        self.audio = None


def tts_infer(sd_t5, sd_voc, sd_amd, st: TTSState, masks, dtype=torch.float32, stages=None):
    """HelloSippyRTPipe.infer (HelloSippyRTPipe.py:191-240) for output_sr == model_sr.
    masks: uint8 [16, 2, 256] keep-masks for the 16 decoder steps of this call."""
    sd = _cast(sd_t5, dtype)
    B = st.last_frame.size(0)
    if st.caches is None:
        st.caches = [{'self': {}, 'cross': {'cross': True}} for _ in range(6)]
    frames = []
    for s in range(16):
        x = t5_prenet_last(sd, st.last_frame, st.idx, st.speakers, masks[s, 0], masks[s, 1])
        h = t5_decoder_layers(sd, x, st.enc, st.enc_mask, st.caches)
        spec = linear(sd, 'speech_decoder_postnet.feat_out', h).view(B, 2, 80)
        frames.append(spec)
        st.last_frame = spec[:, -1]
        prob = torch.sigmoid(linear(sd, 'speech_decoder_postnet.prob_out', h))
        hit = (st.ends_at < 0) & (st.minlen <= st.idx) & (((prob >= 0.5).sum(1) > 0) | (st.maxlen <= st.idx))
        st.ends_at = torch.where(hit, torch.full_like(st.ends_at, st.idx + 2), st.ends_at)
        st.idx += 1
    spec = torch.cat(frames, dim=1)                                # [B,32,80]
    if stages is not None:
        stages['pre_postnet'] = spec
    spec = t5_postnet(sd, spec)
    if stages is not None:
        stages['postnet'] = spec
    S = torch.cat((st.pre_frames, spec), dim=1)                    # [B,36,80]
    st.pre_frames = S[:, -4:]
    chunks = torch.cat([S[:, 8 * i: 8 * i + 12] for i in range(4)], dim=0)     # [4B,12,80] chunk-major
    audio = hifigan(sd_voc, chunks, dtype)
    if stages is not None:
        stages['chunks'], stages['vocoder'] = chunks, audio
    audio = amendment(sd_amd, chunks, audio, dtype)
    if stages is not None:
        stages['amended'] = audio
    st.audio = torch.cat(audio.split(B, dim=0), dim=1)             # [B,8192]
    return st.audio


def tts_dispatch_offsets(idx, starts_at, ends_at, asize=8192, stepsize=512):
    """HelloSippyRTPipe.unbatch_and_dispatch arithmetic (HelloSippyRTPipe.py:242-259) for the
    rows still live: returns ([(startoff, endoff, finished)], more) """
    out = []
    end_idx = idx - 1
    for s, e in zip(starts_at, ends_at):
        startoff = max(0, asize - (idx - s) * stepsize)
        endoff = min(asize, asize - ((idx - e) * stepsize if e >= 0 else 0))
        out.append((startoff, endoff, bool(e >= 0 and e <= end_idx)))
    more = any((e < 0) or (e > end_idx) for e in ends_at)
    return out, more


# =========================================================================================
# Whisper
# =========================================================================================
def _w_attn(sd, p, x, kv, nheads, causal_cache=None, cross_cache=None):
    B, T, D = x.shape
    hd = D // nheads
    q = (linear(sd, p + '.q_proj', x) * hd ** -0.5).view(B, T, nheads, hd).transpose(1, 2)
    if cross_cache is not None and 'k' in cross_cache:
        k, v = cross_cache['k'], cross_cache['v']
    else:
        k = linear(sd, p + '.k_proj', kv).view(B, -1, nheads, hd).transpose(1, 2)
        v = linear(sd, p + '.v_proj', kv).view(B, -1, nheads, hd).transpose(1, 2)
        if cross_cache is not None:
            cross_cache['k'], cross_cache['v'] = k, v
        if causal_cache is not None:
            if 'k' in causal_cache:
                k = torch.cat([causal_cache['k'], k], 2)
                v = torch.cat([causal_cache['v'], v], 2)
            causal_cache['k'], causal_cache['v'] = k, v
    w = q @ k.transpose(-1, -2)
    if causal_cache is not None and T > 1:
        S = k.size(2)
        m = torch.ones(T, S, dtype=torch.bool).tril(S - T)
        w = w.masked_fill(~m, float('-inf'))
    o = (torch.softmax(w, -1) @ v).transpose(1, 2).reshape(B, T, D)
    return linear(sd, p + '.out_proj', o)


def whisper_encoder(sd, mel, nheads, dtype=torch.float32, stages=None):
    """WhisperEncoder.forward (modeling_whisper.py:566-650): conv k3 + GELU, conv k3 s2 + GELU,
    + sinusoid positions, pre-LN layers, final LayerNorm."""
    sd = _cast(sd, dtype)
    E = 'model.encoder.'
    x = F.gelu(F.conv1d(mel.to(dtype), sd[E + 'conv1.weight'], sd[E + 'conv1.bias'], padding=1))
    x = F.gelu(F.conv1d(x, sd[E + 'conv2.weight'], sd[E + 'conv2.bias'], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[E + 'embed_positions.weight']
    if stages is not None:
        stages['conv_out'] = x
    i = 0
    while (E + 'layers.%d.fc1.weight' % i) in sd:
        L = E + 'layers.%d.' % i
        h = layer_norm(sd, L + 'self_attn_layer_norm', x)
        x = x + _w_attn(sd, L + 'self_attn', h, h, nheads)
        h = layer_norm(sd, L + 'final_layer_norm', x)
        x = x + linear(sd, L + 'fc2', F.gelu(linear(sd, L + 'fc1', h)))
        i += 1
    return layer_norm(sd, E + 'layer_norm', x)


def whisper_decoder(sd, tokens, pos0, enc, nheads, caches):
    """WhisperDecoder.forward on `tokens` [B,T] starting at position pos0, KV-cached."""
    D = 'model.decoder.'
    x = F.embedding(tokens, sd[D + 'embed_tokens.weight']) + sd[D + 'embed_positions.weight'][pos0: pos0 + tokens.size(1)]
    i = 0
    while (D + 'layers.%d.fc1.weight' % i) in sd:
        L = D + 'layers.%d.' % i
        c = caches[i]
        h = layer_norm(sd, L + 'self_attn_layer_norm', x)
        x = x + _w_attn(sd, L + 'self_attn', h, h, nheads, causal_cache=c['self'])
        h = layer_norm(sd, L + 'encoder_attn_layer_norm', x)
        x = x + _w_attn(sd, L + 'encoder_attn', h, enc, nheads, cross_cache=c['cross'])
        h = layer_norm(sd, L + 'final_layer_norm', x)
        x = x + linear(sd, L + 'fc2', F.gelu(linear(sd, L + 'fc1', h)))
        i += 1
    x = layer_norm(sd, D + 'layer_norm', x)
    return F.linear(x, sd['proj_out.weight'])


def whisper_greedy(sd, mel, prompt, n_new, nheads, no_speech_id=None, dtype=torch.float32, suppress=None):
    """Encoder + greedy decode of exactly n_new tokens after the prompt (the torch path of
    InfernSTTWorker.infer_and_decode_torch:77-107 with a fixed length instead of EOS).
    Returns (tokens [B, n_new], first_logits [B, V] (prediction after the whole prompt),
    logits0 [B,V] at prompt position 0, enc)."""
    sd = _cast(sd, dtype)
    enc = whisper_encoder(sd, mel, nheads, dtype)
    nl = 0
    while ('model.decoder.layers.%d.fc1.weight' % nl) in sd:
        nl += 1
    caches = [{'self': {}, 'cross': {}} for _ in range(nl)]
    B = mel.size(0)
    toks = prompt.clone()
    logits = whisper_decoder(sd, toks, 0, enc, nheads, caches)
    logits0 = logits[:, 0]
    first = logits[:, -1]
    out = []
    cur = first
    for s in range(n_new):
        nxt = cur.argmax(-1)
        out.append(nxt)
        if s + 1 < n_new:
            cur = whisper_decoder(sd, nxt[:, None], prompt.size(1) + s, enc, nheads, caches)[:, -1]
    return torch.stack(out, 1), first, logits0, enc


# =========================================================================================
# Beam search (the decode CTranslate2's Whisper.generate runs by default for InfernSTTWorker.infer_and_decode_ct2,
# Cluster/InfernSTTWorker.py:61-75: beam_size 5, length_penalty 1).  ctranslate2 is a wheel the reference does not
# vendor, so the search is restated from the transformers 5.15 implementation the fixtures are generated with
# (GenerationMixin._beam_search, generation/utils.py: do_sample False, early_stopping False) and pinned against it
# in tests/golden/whisper_beam.npz.
# =========================================================================================
def beam_search(step_logits, prompt, n_new, num_beams, eos_id, length_penalty=1.0, suppress=None, begin_suppress=None):
    """step_logits(seqs [R, cur_len] int64, beam_src [R] or None) -> logits [R, V] for the next token of every row
    (rows are batch-major: row = b * num_beams + k; beam_src names the previous-step row each row continues, for a
    KV-cached model).  prompt int64 [B, P].  Returns (sequences [B][list of new tokens, eos included], scores [B]
    = sum of log-probs / generated length ** length_penalty, all_fin (sequences, scores, is_fin) per batch item)."""
    B, P = prompt.shape
    K, KK = num_beams, 2 * num_beams
    max_length = P + n_new
    NEG = -1.0e9
    run_seqs = prompt[:, None, :].repeat(1, K, 1)                       # [B, K, cur_len]
    run_scores = torch.zeros(B, K)
    run_scores[:, 1:] = NEG
    fin_seqs = [[[] for _ in range(K)] for _ in range(B)]
    fin_scores = torch.full((B, K), NEG)
    is_fin = torch.zeros(B, K, dtype=torch.bool)
    unsat = torch.ones(B, dtype=torch.bool)
    beam_src = None
    cur_len = P
    while True:
        logits = step_logits(run_seqs.reshape(B * K, cur_len), beam_src).float()
        V = logits.size(-1)
        logp = torch.log_softmax(logits, -1)
        if suppress is not None:
            logp = logp + suppress
        if begin_suppress is not None and cur_len == P:
            logp = logp + begin_suppress
        acc = (logp.view(B, K, V) + run_scores[:, :, None]).reshape(B, K * V)
        top_s, top_i = torch.topk(acc, KK)                              # sorted descending
        top_b, top_t = top_i // V, top_i % V
        cand_seqs = torch.cat([torch.gather(run_seqs, 1, top_b[:, :, None].expand(B, KK, cur_len)), top_t[:, :, None]], 2)
        hits = (top_t == eos_id) | (cur_len + 1 >= max_length)
        # running beams of the next step: the best K candidates that did not just finish
        run_pick = torch.topk(top_s + hits.float() * NEG, K)[1]
        run_seqs = torch.gather(cand_seqs, 1, run_pick[:, :, None].expand(B, K, cur_len + 1))
        run_scores = torch.gather(top_s + hits.float() * NEG, 1, run_pick)
        beam_src = (torch.gather(top_b, 1, run_pick) + torch.arange(B)[:, None] * K).reshape(-1)
        # finished hypotheses: only candidates ranked inside the first K may finish
        just = hits.clone()
        just[:, K:] = False
        fs = top_s / float(cur_len + 1 - P) ** length_penalty
        fs = fs + (~unsat)[:, None].float() * NEG + (~just).float() * NEG
        merged = torch.cat([fin_scores, fs], 1)
        pick = torch.topk(merged, K)[1]
        new_fin = []
        for b in range(B):
            row = []
            for j in pick[b].tolist():
                row.append(fin_seqs[b][j] if j < K else cand_seqs[b, j - K, P:].tolist())
            new_fin.append(row)
        fin_seqs = new_fin
        fin_scores = torch.gather(merged, 1, pick)
        is_fin = torch.gather(torch.cat([is_fin, just], 1), 1, pick)
        cur_len += 1
        best_possible = run_scores[:, 0] / float(cur_len - P) ** length_penalty
        worst = torch.where(is_fin, fin_scores.min(1, keepdim=True)[0], torch.full((B, K), NEG))
        unsat = unsat & (best_possible[:, None] > worst).any(1)
        if not bool(unsat.any()) or bool(hits.all()):
            break
    return [fin_seqs[b][0] for b in range(B)], fin_scores[:, 0].clone(), (fin_seqs, fin_scores, is_fin)


def whisper_beam(sd, mel, prompt, n_new, nheads, num_beams=5, eos_id=50257, length_penalty=1.0, suppress=None,
                 begin_suppress=None, dtype=torch.float32):
    """Encoder + beam search over the KV-cached decoder.  Returns (sequences, scores, enc)."""
    sd = _cast(sd, dtype)
    enc = whisper_encoder(sd, mel, nheads, dtype)
    nl = 0
    while ('model.decoder.layers.%d.fc1.weight' % nl) in sd:
        nl += 1
    K = num_beams
    encK = enc.repeat_interleave(K, 0)
    state = {'caches': None}

    def step(seqs, beam_src):
        if beam_src is None:
            state['caches'] = [{'self': {}, 'cross': {}} for _ in range(nl)]
            return whisper_decoder(sd, seqs, 0, encK, nheads, state['caches'])[:, -1]
        for c in state['caches']:
            c['self']['k'] = c['self']['k'].index_select(0, beam_src)
            c['self']['v'] = c['self']['v'].index_select(0, beam_src)
        return whisper_decoder(sd, seqs[:, -1:], seqs.size(1) - 1, encK, nheads, state['caches'])[:, -1]
    seqs, scores, _ = beam_search(step, prompt, n_new, K, eos_id, length_penalty, suppress, begin_suppress)
    return seqs, scores, enc


# =========================================================================================
# Qwen2 (decoder-only LLM of InfernLLMWorker, Cluster/InfernLLMWorker.py:60-119)
# transformers/models/qwen2/modeling_qwen2.py: Qwen2RMSNorm, Qwen2RotaryEmbedding + apply_rotary_pos_emb (rotate_half),
# Qwen2Attention (grouped-query heads: repeat_kv), Qwen2MLP, Qwen2ForCausalLM (tied or separate lm_head)
# =========================================================================================
def rms_norm(w, x, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rope_cos_sin(max_pos, head_dim, theta):
    """f32 [max_pos, head_dim/2] cos and sin of position * theta^(-2j/head_dim)"""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    fr = torch.arange(max_pos).float()[:, None] * inv[None, :]
    return fr.cos(), fr.sin()


def _rope(x, cos, sin):
    """x [B, H, T, hd]; cos/sin [T, hd/2]: x * cat(cos, cos) + rotate_half(x) * cat(sin, sin)"""
    h = x.size(-1) // 2
    x1, x2 = x[..., :h], x[..., h:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)


def qwen2_forward(sd, cfg, tokens, pos0, caches):
    """tokens int64 [B, T] starting at position pos0 (same for every row; ragged batches are run row by row);
    caches: list per layer of {} / {'k','v'} [B, kv_heads, S, hd].  Returns logits f32 [B, T, vocab].
    With `sd` cast to bfloat16 (_cast) the arithmetic follows transformers' Qwen2 modules run in that dtype (modeling_qwen2.py:
    RMSNorm and softmax in fp32 and cast back, rotary cos / sin cast to the activations' dtype) -- the "reference engine in bf16"
    whose error against the fp32 run sets the device bar at depths the transformers fixtures (2-3 layers) do not reach."""
    d, hd, nh, nkv = cfg['hidden'], cfg['head_dim'], cfg['heads'], cfg['kv_heads']
    B, T = tokens.shape
    dt = sd['model.embed_tokens.weight'].dtype
    if dt != torch.float32:
        return _qwen2_forward_lowp(sd, cfg, tokens, pos0, caches, dt)
    cos, sin = rope_cos_sin(pos0 + T, hd, cfg['rope_theta'])
    cos, sin = cos[pos0:], sin[pos0:]
    x = F.embedding(tokens, sd['model.embed_tokens.weight'])
    for i in range(cfg['layers']):
        L = 'model.layers.%d.' % i
        h = rms_norm(sd[L + 'input_layernorm.weight'], x, cfg['rms_eps'])
        q = linear(sd, L + 'self_attn.q_proj', h).view(B, T, nh, hd).transpose(1, 2)
        k = linear(sd, L + 'self_attn.k_proj', h).view(B, T, nkv, hd).transpose(1, 2)
        v = linear(sd, L + 'self_attn.v_proj', h).view(B, T, nkv, hd).transpose(1, 2)
        q, k = _rope(q, cos, sin), _rope(k, cos, sin)
        c = caches[i]
        if 'k' in c:
            k, v = torch.cat([c['k'], k], 2), torch.cat([c['v'], v], 2)
        c['k'], c['v'] = k, v
        S = k.size(2)
        kr, vr = k.repeat_interleave(nh // nkv, 1), v.repeat_interleave(nh // nkv, 1)
        w = (q @ kr.transpose(-1, -2)) * hd ** -0.5
        m = torch.ones(T, S, dtype=torch.bool).tril(S - T)
        w = w.masked_fill(~m, float('-inf'))
        o = (torch.softmax(w, -1) @ vr).transpose(1, 2).reshape(B, T, nh * hd)
        x = x + F.linear(o, sd[L + 'self_attn.o_proj.weight'])
        h = rms_norm(sd[L + 'post_attention_layernorm.weight'], x, cfg['rms_eps'])
        x = x + F.linear(F.silu(F.linear(h, sd[L + 'mlp.gate_proj.weight'])) * F.linear(h, sd[L + 'mlp.up_proj.weight']),
                         sd[L + 'mlp.down_proj.weight'])
    x = rms_norm(sd['model.norm.weight'], x, cfg['rms_eps'])
    return F.linear(x, sd.get('lm_head.weight', sd['model.embed_tokens.weight']))


def _qwen2_forward_lowp(sd, cfg, tokens, pos0, caches, dt):
    """qwen2_forward with weights and activations in `dt` (bfloat16), the way transformers runs the model in that dtype"""
    d, hd, nh, nkv = cfg['hidden'], cfg['head_dim'], cfg['heads'], cfg['kv_heads']
    B, T = tokens.shape
    cos, sin = rope_cos_sin(pos0 + T, hd, cfg['rope_theta'])
    cos, sin = cos[pos0:].to(dt), sin[pos0:].to(dt)

    def norm(w, x):
        xf = x.float()
        return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + cfg['rms_eps'])).to(dt)
    x = F.embedding(tokens, sd['model.embed_tokens.weight'])
    for i in range(cfg['layers']):
        L = 'model.layers.%d.' % i
        h = norm(sd[L + 'input_layernorm.weight'], x)
        q = linear(sd, L + 'self_attn.q_proj', h).view(B, T, nh, hd).transpose(1, 2)
        k = linear(sd, L + 'self_attn.k_proj', h).view(B, T, nkv, hd).transpose(1, 2)
        v = linear(sd, L + 'self_attn.v_proj', h).view(B, T, nkv, hd).transpose(1, 2)
        q, k = _rope(q, cos, sin), _rope(k, cos, sin)
        c = caches[i]
        if 'k' in c:
            k, v = torch.cat([c['k'], k], 2), torch.cat([c['v'], v], 2)
        c['k'], c['v'] = k, v
        S = k.size(2)
        kr, vr = k.repeat_interleave(nh // nkv, 1), v.repeat_interleave(nh // nkv, 1)
        w = (q @ kr.transpose(-1, -2)) * hd ** -0.5
        m = torch.ones(T, S, dtype=torch.bool).tril(S - T)
        w = w.masked_fill(~m, float('-inf'))
        o = (torch.softmax(w.float(), -1).to(dt) @ vr).transpose(1, 2).reshape(B, T, nh * hd)
        x = x + F.linear(o, sd[L + 'self_attn.o_proj.weight'])
        h = norm(sd[L + 'post_attention_layernorm.weight'], x)
        x = x + F.linear(F.silu(F.linear(h, sd[L + 'mlp.gate_proj.weight'])) * F.linear(h, sd[L + 'mlp.up_proj.weight']),
                         sd[L + 'mlp.down_proj.weight'])
    x = norm(sd['model.norm.weight'], x)
    return F.linear(x, sd.get('lm_head.weight', sd['model.embed_tokens.weight'])).float()


def qwen2_greedy(sd, cfg, prompts, n_new, eos_ids=()):
    """Greedy continuation of every prompt (list of token-id lists, decoded at their own lengths: what a left-padded
    batch through transformers' generate yields) for up to n_new tokens, stopping a row at an eos id (kept).
    Returns (list of generated id lists, list of f32 [len(prompt)+generated-1, vocab] logits per row)."""
    outs, logs = [], []
    for p in prompts:
        caches = [{} for _ in range(cfg['layers'])]
        lg = qwen2_forward(sd, cfg, torch.tensor([p]), 0, caches)[0]
        allg = [lg]
        new = []
        cur = lg[-1]
        for s in range(n_new):
            t = int(cur.argmax())
            new.append(t)
            if t in eos_ids or s + 1 == n_new:
                break
            lg = qwen2_forward(sd, cfg, torch.tensor([[t]]), len(p) + s, caches)[0]
            allg.append(lg)
            cur = lg[-1]
        outs.append(new)
        logs.append(torch.cat(allg, 0))
    return outs, logs


# =========================================================================================
# Sampling (transformers/generation/logits_process.py: RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
# TopKLogitsWarper, TopPLogitsWarper in the order generate applies them, then softmax)
# =========================================================================================
def sample_warp(logits, history, penalty=1.0, temperature=1.0, top_k=0, top_p=1.0):
    """logits f32 [V], history: token ids of the row so far -> (token ids by descending probability, their probabilities)
    of the distribution transformers' generate draws from."""
    s = logits.clone().float()
    if penalty != 1.0 and len(history):
        idx = torch.unique(torch.as_tensor(history, dtype=torch.long))
        sc = s[idx]
        s[idx] = torch.where(sc < 0, sc * penalty, sc / penalty)
    s = s / temperature
    if top_k and top_k > 0:
        kth = torch.topk(s, min(top_k, s.numel())).values[-1]
        s = s.masked_fill(s < kth, float('-inf'))
    if top_p < 1.0:
        so, si = torch.sort(s, descending=False)
        cum = torch.softmax(so, -1).cumsum(-1)
        rem = cum <= (1 - top_p)
        rem[-1:] = False
        s[si[rem]] = float('-inf')
    p = torch.softmax(s, -1)
    order = torch.argsort(p, descending=True, stable=True)
    n = int((p > 0).sum())
    return order[:n], p[order[:n]]


def sample_pick(ids, probs, u):
    """inverse-CDF draw over the candidates in descending order"""
    cdf = probs.cumsum(-1)
    i = int((cdf > u * float(cdf[-1])).nonzero()[0]) if bool((cdf > u * float(cdf[-1])).any()) else len(ids) - 1
    return int(ids[i])


# ---- the recurrent VAD network of infernos_amd/csrc/vadnet.hip ------------------------------------------------------------------
def vadnet(x, sd, h, c):
    """Speech probability of a 768-sample window per row and the new recurrent state.  The network is a Silero-v3.1-SHAPED
    stand-in (Core/VAD/SileroVAD.py:44-45 loads a third-party TorchScript file that is not in the reference tree: PARITY UNPINNED
    against it): Conv1d(1->32, k128, s64) + ReLU -> Conv1d(32->64, k3, s2, p1) + ReLU -> 2 x LSTM(64) over the 6 steps with the
    state (h, c) [2, B, 64] carried between calls (SileroVADUtils.py:21-26,99,131) -> Linear(64->1) -> sigmoid -> mean over steps.
    x f32 [B, 768]; sd = infernos_amd.weights.synth_vadnet(); returns (prob [B], h', c').  Pinned to torch.nn modules holding the same
    weights by tests/test_vadnet_oracle.py."""
    import torch
    import torch.nn.functional as F
    f1 = F.relu(F.conv1d(x[:, None, :].float(), sd['conv1.weight'], sd['conv1.bias'], stride=64))            # [B, 32, 11]
    f2 = F.relu(F.conv1d(f1, sd['conv2.weight'], sd['conv2.bias'], stride=2, padding=1))                     # [B, 64, 6]
    h, c = [h[0].clone().float(), h[1].clone().float()], [c[0].clone().float(), c[1].clone().float()]
    ys = []
    for s in range(f2.size(2)):
        inp = f2[:, :, s]
        for l in (0, 1):
            g = inp @ sd['lstm.weight_ih_l%d' % l].t() + sd['lstm.bias_ih_l%d' % l] + h[l] @ sd['lstm.weight_hh_l%d' % l].t() + \
                sd['lstm.bias_hh_l%d' % l]
            i, f, gg, o = g.chunk(4, dim=1)
            c[l] = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
            h[l] = torch.sigmoid(o) * torch.tanh(c[l])
            inp = h[l]
        ys.append(torch.sigmoid(h[1] @ sd['out.weight'].t() + sd['out.bias'])[:, 0])
    return torch.stack(ys, 1).mean(1), torch.stack(h), torch.stack(c)
