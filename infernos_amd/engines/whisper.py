"""Whisper encoder/decoder with greedy search on the HIP device.

Replaces the engines behind Cluster/InfernSTTWorker.py:61-107 (CTranslate2
`Whisper.generate`, or the HF/ipex torch path): conv front-end, pre-LN transformer encoder,
KV-cached decoder, tied output projection, greedy argmax and the no-speech probability
(softmax of the first-position logits at <|nospeech|>, InfernSTTWorker.py:89-90).
Architecture per transformers modeling_whisper.py (v5.15.0).  The decode loop runs entirely
on the device (argmax feeds the next embedding lookup; no host sync per token).
"""

import torch

from .. import _lib, ops
from ..ops import ACT_GELU, BF16

QS = 64 ** -0.5
N_CTX = 1500


def _ln(sd, p, dev):
    return sd[p + '.weight'].float().contiguous().to(dev), sd[p + '.bias'].float().contiguous().to(dev)


class Whisper:
    def __init__(self, sd, device, max_batch=64, max_tokens=448):
        self.device = dev = _lib.require_device(device)
        E, Dc = 'model.encoder.', 'model.decoder.'
        self.d = d = sd[E + 'conv1.weight'].shape[0]
        self.n_mel = sd[E + 'conv1.weight'].shape[1]
        self.h = d // 64
        self.ff = sd[E + 'layers.0.fc1.weight'].shape[0]
        self.vocab = sd[Dc + 'embed_tokens.weight'].shape[0]
        self.max_tokens = min(max_tokens, sd[Dc + 'embed_positions.weight'].shape[0])
        self.beam_cross_mfma = True       # the beams' cross-attention on k_attn_prefill (False: k_attn_decode_shared; tests/test_beam_gpu.py runs both)
        self.c1 = (ops.w_conv(sd[E + 'conv1.weight'], dev), ops.w_bias(sd[E + 'conv1.bias'], dev))
        self.c2 = (ops.w_conv(sd[E + 'conv2.weight'], dev), ops.w_bias(sd[E + 'conv2.bias'], dev))
        self.enc_pos = sd[E + 'embed_positions.weight'].to(BF16).contiguous().to(dev)
        self.enc_layers, self.dec_layers = [], []
        zero_k = torch.zeros(d)

        def attn(prefix, fused_qkv):
            q = (sd[prefix + 'q_proj.weight'].float() * QS, sd[prefix + 'q_proj.bias'].float() * QS)
            k = (sd[prefix + 'k_proj.weight'].float(), zero_k)                 # k_proj has no bias
            v = (sd[prefix + 'v_proj.weight'].float(), sd[prefix + 'v_proj.bias'].float())
            out = dict(wo=ops.w_linear(sd[prefix + 'out_proj.weight'], dev), bo=ops.w_bias(sd[prefix + 'out_proj.bias'], dev))
            if fused_qkv:
                out['wqkv'] = ops.w_linear(torch.cat([q[0], k[0], v[0]]), dev)
                out['bqkv'] = ops.w_bias(torch.cat([q[1], k[1], v[1]]), dev)
            else:
                out['wq'], out['bq'] = ops.w_linear(q[0], dev), ops.w_bias(q[1], dev)
                out['wkv'] = ops.w_linear(torch.cat([k[0], v[0]]), dev)
                out['bkv'] = ops.w_bias(torch.cat([k[1], v[1]]), dev)
            return out
        i = 0
        while (E + 'layers.%d.fc1.weight' % i) in sd:
            L = E + 'layers.%d.' % i
            lay = attn(L + 'self_attn.', True)
            lay.update(ln1=_ln(sd, L + 'self_attn_layer_norm', dev), ln2=_ln(sd, L + 'final_layer_norm', dev),
                       w1=ops.w_linear(sd[L + 'fc1.weight'], dev), b1=ops.w_bias(sd[L + 'fc1.bias'], dev),
                       w2=ops.w_linear(sd[L + 'fc2.weight'], dev), b2=ops.w_bias(sd[L + 'fc2.bias'], dev))
            self.enc_layers.append(lay)
            i += 1
        self.enc_ln = _ln(sd, E + 'layer_norm', dev)
        i = 0
        while (Dc + 'layers.%d.fc1.weight' % i) in sd:
            L = Dc + 'layers.%d.' % i
            lay = {'self': attn(L + 'self_attn.', True), 'cross': attn(L + 'encoder_attn.', False)}
            lay.update(ln1=_ln(sd, L + 'self_attn_layer_norm', dev), ln2=_ln(sd, L + 'encoder_attn_layer_norm', dev),
                       ln3=_ln(sd, L + 'final_layer_norm', dev),
                       w1=ops.w_linear(sd[L + 'fc1.weight'], dev), b1=ops.w_bias(sd[L + 'fc1.bias'], dev),
                       w2=ops.w_linear(sd[L + 'fc2.weight'], dev), b2=ops.w_bias(sd[L + 'fc2.bias'], dev))
            self.dec_layers.append(lay)
            i += 1
        self.dec_ln = _ln(sd, Dc + 'layer_norm', dev)
        self.tok = sd[Dc + 'embed_tokens.weight'].to(BF16).contiguous().to(dev)        # also the tied proj_out
        self.dec_pos = sd[Dc + 'embed_positions.weight'].to(BF16).contiguous().to(dev)
        self._enc_bufs = {}
        self._dec_bufs = {}
        # LayerNorm folded around the decode-step GEMMs (ifh_conv_desc.aln_* / stats_out; same scheme as the SpeechT5
        # decoder): 13 of the 49 launches per token disappear.  The vocabulary projection is padded to a multiple of
        # 16 rows; the padded logits are pinned to -1e30 through the folded bias so that argmax never picks them.
        self.fold_ln = True            # LayerNorms folded around the decode GEMMs (False: explicit launches, decoder_step)
        self.vpad = -(-self.vocab // 16) * 16
        self.dec_fold = []
        nl = len(self.dec_layers)
        for i in range(nl):
            L = Dc + 'layers.%d.' % i
            g = lambda nm: (sd[L + nm + '.weight'].float(), sd[L + nm + '.bias'].float())
            sa, ca = L + 'self_attn.', L + 'encoder_attn.'
            wqkv = torch.cat([sd[sa + 'q_proj.weight'].float() * QS, sd[sa + 'k_proj.weight'].float(), sd[sa + 'v_proj.weight'].float()])
            bqkv = torch.cat([sd[sa + 'q_proj.bias'].float() * QS, zero_k, sd[sa + 'v_proj.bias'].float()])
            self.dec_fold.append(dict(
                qkv=ops.w_linear_ln(wqkv, bqkv, *g('self_attn_layer_norm'), dev),
                cq=ops.w_linear_ln(sd[ca + 'q_proj.weight'].float() * QS, sd[ca + 'q_proj.bias'].float() * QS,
                                   *g('encoder_attn_layer_norm'), dev),
                ff1=ops.w_linear_ln(sd[L + 'fc1.weight'], sd[L + 'fc1.bias'], *g('final_layer_norm'), dev)))
        tokw = torch.zeros(self.vpad, d)
        tokw[:self.vocab] = sd[Dc + 'embed_tokens.weight'].float()
        w, c2, c1 = ops.w_linear_ln(tokw, None, sd[Dc + 'layer_norm.weight'].float(), sd[Dc + 'layer_norm.bias'].float(), dev)
        c2[self.vocab:] = -1e30
        self.logit_fold = (w, c2, c1)

    # ---- encoder ------------------------------------------------------------------------------
    def encode(self, mel: torch.Tensor = None, raw=None) -> torch.Tensor:
        """mel [B, n_mel, 3000] f32 or bf16 (ifh_logmel_run layout) -> bf16 [B, 1500, d].
        raw = (WhisperLogMel, raw f32 [B, n_mel, 3000], win_max int32 [B]) instead of mel: the normalisation of the
        features is applied by the layout change in front of conv1 (ifh_logmel_finish_transpose_bf16), same bits."""
        dev, d, H, FF = self.device, self.d, self.h, self.ff
        Bn = mel.size(0) if raw is None else raw[1].size(0)
        if Bn not in self._enc_bufs:
            e = lambda *s: torch.empty(s, dtype=BF16, device=dev)
            rows = Bn * N_CTX
            while len(self._enc_bufs) >= 3:                      # a few batch shapes stay resident (grouped cycles, remainders)
                self._enc_bufs.pop(next(iter(self._enc_bufs)))
            self._enc_bufs[Bn] = dict(mt=e(Bn, 3000, self.n_mel), c1=e(Bn, 3000, d), x=e(rows, d), hn=e(rows, d),
                                      qkv=e(rows, 3 * d), att=e(rows, d), ff=e(rows, FF), out=e(rows, d))
        b = self._enc_bufs[Bn]
        rows = Bn * N_CTX
        if raw is None:
            ops.transpose_to_bf16(mel.contiguous(), b['mt'], Bn, self.n_mel, 3000)
        else:
            raw[0].to_conv_input(raw[1], raw[2], out=b['mt'])
        ops.conv(b['mt'], *self.c1, b['c1'], nbatch=Bn, t_in=3000, t_out=3000, cin=self.n_mel, n=d, taps=3, pad=1, act=ACT_GELU)
        ops.conv(b['c1'], *self.c2, b['x'], nbatch=Bn, t_in=3000, t_out=N_CTX, cin=d, n=d, taps=3, stride=2, pad=1,
                 act=ACT_GELU, resid=self.enc_pos, resid_ld=d, resid_bstride=0)
        x = b['x']
        for L in self.enc_layers:
            ops.layernorm(x, *L['ln1'], b['hn'], rows, d)
            ops.linear(b['hn'], L['wqkv'], L['bqkv'], b['qkv'], rows=rows, k=d, n=3 * d)
            ops.attn_prefill(b['qkv'], b['qkv'], b['qkv'], b['att'], nbatch=Bn, nheads=H, tq=N_CTX, tk=N_CTX, k_off=d,
                             v_off=2 * d, q_ts=3 * d, k_ts=3 * d, v_ts=3 * d, o_ts=d)
            ops.linear(b['att'], L['wo'], L['bo'], x, rows=rows, k=d, n=d, resid=x)
            ops.layernorm(x, *L['ln2'], b['hn'], rows, d)
            ops.linear(b['hn'], L['w1'], L['b1'], b['ff'], rows=rows, k=d, n=FF, act=ACT_GELU)
            ops.linear(b['ff'], L['w2'], L['b2'], x, rows=rows, k=FF, n=d, resid=x)
        ops.layernorm(x, *self.enc_ln, b['out'], rows, d)
        return b['out'].view(Bn, N_CTX, d)

    # ---- decoder ------------------------------------------------------------------------------
    def _dec(self, Bn, beams=0):
        """Decode buffers for Bn rows.  beams >= 1: Bn = utterances * beams rows that share the utterances' cross-attention
        K/V (one cache row per utterance) and own a second self-attention cache set for the per-step beam gather."""
        key = Bn if beams == 0 else (Bn, beams)
        if key not in self._dec_bufs:
            dev, d = self.device, self.d
            e = lambda *s, dt=BF16: torch.empty(s, dtype=dt, device=dev)
            while len(self._dec_bufs) >= 3:
                self._dec_bufs.pop(next(iter(self._dec_bufs)))
            self._dec_bufs[key] = dict(
                cross=[e(Bn // max(1, beams) * N_CTX, 2 * d) for _ in self.dec_layers],
                kv_all=torch.zeros((len(self.dec_layers), Bn, self.max_tokens, 2 * d), dtype=BF16, device=dev),
                x=e(Bn, d), hn=e(Bn, d), q=e(Bn, d), att=e(Bn, d), ff=e(Bn, self.ff),
                logits_full=e(Bn, self.vpad, dt=torch.float32),
                stats=torch.zeros((3 * len(self.dec_layers), max(64, -(-Bn // 16) * 16), 2), dtype=torch.int64, device=dev),
                toks=torch.zeros((self.max_tokens + 1, Bn), dtype=torch.int32, device=dev),
                pos=torch.zeros(1, dtype=torch.int32, device=dev), graphs={}, eager_runs=0)
            b = self._dec_bufs[key]
            b['logits'] = b['logits_full'][:, :self.vocab]         # [Bn, vocab] view, row stride vpad
            b['kv'] = list(b['kv_all'].unbind(0))                  # per-layer views of one allocation (the beam gather moves all layers in one launch)
            if beams >= 1:
                b['cross_group'] = beams
                b['kv2_all'] = torch.zeros_like(b['kv_all'])
                b['kv2'] = list(b['kv2_all'].unbind(0))
        return self._dec_bufs[key]

    def _cross_attn(self, bufs, li, Bn):
        d, H = self.d, self.h
        ck = bufs['cross'][li]
        g = bufs.get('cross_group', 1)
        if g == 1:
            ops.attn_decode(bufs['q'], ck, ck, bufs['att'], nbatch=Bn, nheads=H, max_keys=N_CTX, q_bs=d,
                            kv_bs=N_CTX * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d)
        elif self.beam_cross_mfma:                 # beams: the g rows of an utterance are the query rows of one flash-attention tile
            # (k_attn_prefill streams the utterance's K/V once at 4.9 TB/s against 4.2 for the per-row kernel that shares the
            # loads; another summation order than k_attn_decode's -- parity is held against the transformers fixture, not against it)
            ops.attn_prefill(bufs['q'], ck, ck, bufs['att'], nbatch=Bn // g, nheads=H, tq=g, tk=N_CTX, v_off=d, q_ts=d,
                             k_ts=2 * d, v_ts=2 * d, o_ts=d)
        else:                                      # beams: g consecutive rows share one utterance's cross K/V
            ops.attn_decode_shared(bufs['q'], ck, ck, bufs['att'], nbatch=Bn, nheads=H, max_keys=N_CTX, q_bs=d,
                                   kv_bs=N_CTX * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d, kv_group=g)

    def _tail(self, bufs, Bn, mode):
        """What follows the logits of a step: nothing (False), greedy pick (True), or one beam-search step + the KV
        gather into the other cache set (('beam', parity, ...): see generate_beam).  Runs after pos was advanced."""
        if mode is True or mode is False:
            return
        bm = bufs['beam']
        ops.beam_step(bufs['logits_full'], bm['state'], bufs['toks'], bufs['pos'], vocab=self.vocab, ld=self.vpad,
                      prompt_len=bm['P'], max_length=bm['max_length'], eos_id=bm['eos'], length_penalty=bm['lp'],
                      suppress=bm['suppress'], begin_suppress=bm['begin_suppress'])
        src, dst = (bufs['kv_all'], bufs['kv2_all']) if mode[1] == 0 else (bufs['kv2_all'], bufs['kv_all'])
        ops.kv_gather(src, dst, bm['state'].beam_src, bufs['pos'], nrows=Bn, max_len=self.max_tokens, tok_elems=2 * self.d,
                      nlayers=len(self.dec_layers))

    @staticmethod
    def _kvsel(bufs, mode):
        return bufs['kv2'] if (mode is not True and mode is not False and mode[1] == 1) else bufs['kv']

    def decoder_step(self, bufs, Bn: int, argmax):
        """One token per sequence at the position held in bufs['pos'] (device scalar): embeds
        toks[pos], appends K/V at pos, attends over pos+1 keys, optionally writes
        argmax(logits) to toks[pos+1], then advances pos.  Identical launches for every position,
        hence capturable once and replayed."""
        d, H = self.d, self.h
        x, pos, toks = bufs['x'], bufs['pos'], bufs['toks']
        ops.embed(toks, self.tok, self.dec_pos, x, n=Bn, dim=d, pos0=0, seq_len=1, dyn_pos=pos, dyn_ids_mul=Bn)
        smax = self.max_tokens
        kvs = self._kvsel(bufs, argmax)
        for li, L in enumerate(self.dec_layers):
            S, C = L['self'], L['cross']
            kv = kvs[li]
            ops.layernorm(x, *L['ln1'], bufs['hn'], Bn, d)
            ops.conv(bufs['hn'], S['wqkv'], S['bqkv'], bufs['q'], nbatch=Bn, t_in=1, t_out=1, cin=d, n=3 * d, ldc=d,
                     out_bstride=d, dyn_pos=pos, n_split=d, out2=kv, out2_bstride=smax * 2 * d, ldc2=2 * d, dyn_ooff2_mul=1)
            ops.attn_decode(bufs['q'], kv, kv, bufs['att'], nbatch=Bn, nheads=H, max_keys=smax, q_bs=d,
                            kv_bs=smax * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d, dyn_len=pos, dyn_add=1)
            ops.linear(bufs['att'], S['wo'], S['bo'], x, rows=Bn, k=d, n=d, resid=x)
            ops.layernorm(x, *L['ln2'], bufs['hn'], Bn, d)
            ops.linear(bufs['hn'], C['wq'], C['bq'], bufs['q'], rows=Bn, k=d, n=d)
            self._cross_attn(bufs, li, Bn)
            ops.linear(bufs['att'], C['wo'], C['bo'], x, rows=Bn, k=d, n=d, resid=x)
            ops.layernorm(x, *L['ln3'], bufs['hn'], Bn, d)
            ops.linear(bufs['hn'], L['w1'], L['b1'], bufs['ff'], rows=Bn, k=d, n=self.ff, act=ACT_GELU)
            ops.linear(bufs['ff'], L['w2'], L['b2'], x, rows=Bn, k=self.ff, n=d, resid=x)
        ops.layernorm(x, *self.dec_ln, bufs['hn'], Bn, d)
        ops.linear(bufs['hn'], self.tok, None, bufs['logits_full'], rows=Bn, k=d, n=self.vocab, ldc=self.vpad)
        if argmax is True:
            ops.argmax_pick(bufs['logits_full'], vocab=self.vocab, nrows=Bn, ld=self.vpad, argmax_out=toks, out_off=Bn,
                            dyn_pos=pos, dyn_out_mul=Bn)
        ops.add_i32(pos, 1)
        self._tail(bufs, Bn, argmax)
        return bufs['logits']

    def decoder_step_folded(self, bufs, Bn: int, argmax: bool):
        """decoder_step with every LayerNorm but the first folded around the neighbouring GEMMs: the producers of the
        residual stream x (out_proj, fc2; resid = x) add per-row (sum, sum of squares) in their epilogue, the consumers
        of LayerNorm(x) (q|k|v, cross q, fc1, the vocabulary projection) apply mean/rstd in theirs."""
        d, H = self.d, self.h
        x, pos, toks, stats = bufs['x'], bufs['pos'], bufs['toks'], bufs['stats']
        SO = stats.size(1) * 2                 # zero on entry: cleared by the last launch of the previous token
        ops.embed(toks, self.tok, self.dec_pos, x, n=Bn, dim=d, pos0=0, seq_len=1, dyn_pos=pos, dyn_ids_mul=Bn)
        smax = self.max_tokens
        kvs = self._kvsel(bufs, argmax)
        for li, (L, F) in enumerate(zip(self.dec_layers, self.dec_fold)):
            S, C = L['self'], L['cross']
            kv = kvs[li]
            s1, s2, s3, s3p = (3 * li) * SO, (3 * li + 1) * SO, (3 * li + 2) * SO, (3 * li - 1) * SO
            kvargs = dict(nbatch=Bn, t_in=1, t_out=1, cin=d, n=3 * d, ldc=d, out_bstride=d, dyn_pos=pos, n_split=d, out2=kv,
                          out2_bstride=smax * 2 * d, ldc2=2 * d, dyn_ooff2_mul=1)
            if li == 0:                                  # x comes from the embedding kernel: no statistics yet
                ops.layernorm(x, *L['ln1'], bufs['hn'], Bn, d)
                ops.conv(bufs['hn'], S['wqkv'], S['bqkv'], bufs['q'], **kvargs)
            else:
                w, c2, c1 = F['qkv']
                ops.conv(x, w, c2, bufs['q'], aln=(stats, s3p, c1), ln_dim=d, **kvargs)
            ops.attn_decode(bufs['q'], kv, kv, bufs['att'], nbatch=Bn, nheads=H, max_keys=smax, q_bs=d,
                            kv_bs=smax * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d, dyn_len=pos, dyn_add=1)
            ops.linear(bufs['att'], S['wo'], S['bo'], x, rows=Bn, k=d, n=d, resid=x, stats_out=stats, stats_off=s1, ln_dim=d)
            w, c2, c1 = F['cq']
            ops.linear(x, w, c2, bufs['q'], rows=Bn, k=d, n=d, aln=(stats, s1, c1), ln_dim=d)
            self._cross_attn(bufs, li, Bn)
            ops.linear(bufs['att'], C['wo'], C['bo'], x, rows=Bn, k=d, n=d, resid=x, stats_out=stats, stats_off=s2, ln_dim=d)
            w, c2, c1 = F['ff1']
            ops.linear(x, w, c2, bufs['ff'], rows=Bn, k=d, n=self.ff, act=ACT_GELU, aln=(stats, s2, c1), ln_dim=d)
            ops.linear(bufs['ff'], L['w2'], L['b2'], x, rows=Bn, k=self.ff, n=d, resid=x, stats_out=stats, stats_off=s3, ln_dim=d)
        w, c2, c1 = self.logit_fold
        ops.linear(x, w, c2, bufs['logits_full'], rows=Bn, k=d, n=self.vpad, aln=(stats, (3 * len(self.dec_layers) - 1) * SO, c1),
                   ln_dim=d)
        if argmax is True:
            ops.argmax_pick(bufs['logits_full'], vocab=self.vocab, nrows=Bn, ld=self.vpad, argmax_out=toks, out_off=Bn,
                            dyn_pos=pos, dyn_out_mul=Bn)
        ops.add_i32(pos, 1, zero=stats)
        self._tail(bufs, Bn, argmax)
        return bufs['logits']

    def _step(self, bufs, Bn, argmax, use_graphs):
        if self.fold_ln and Bn <= 1024:
            step = self.decoder_step_folded
        else:
            step = self.decoder_step
        return self._step_with(step, bufs, Bn, argmax, use_graphs)

    def _step_with(self, step_fn, bufs, Bn, argmax, use_graphs):
        if not use_graphs:
            return step_fn(bufs, Bn, argmax)
        g = bufs['graphs'].get(argmax)
        if g is None:
            g = bufs['graphs'][argmax] = _lib.CountedGraph(lambda: step_fn(bufs, Bn, argmax))
        g.replay()
        return bufs['logits']

    def forced_logits(self, enc: torch.Tensor, tokens: torch.Tensor, rows=None, use_graphs=True) -> torch.Tensor:
        """Teacher-forced decode: feed `tokens` int [B,L] position by position through the same per-token step
        (and hipGraph) `generate` uses and return the f32 logits after every position for the batch rows `rows`
        (default all): [len(rows), L, vocab].  What `self.model(**inputs, decoder_input_ids=tokens).logits`
        (Cluster/InfernSTTWorker.py:83-88) returns, computed incrementally through the KV cache."""
        dev, d = self.device, self.d
        Bn, L = tokens.shape
        assert L <= self.max_tokens
        bufs = self._dec(Bn)
        use_graphs = use_graphs and bufs['eager_runs'] >= 1
        for li, Lr in enumerate(self.dec_layers):
            C = Lr['cross']
            ops.linear(enc, C['wkv'], C['bkv'], bufs['cross'][li], rows=Bn * N_CTX, k=d, n=2 * d)
        toks = bufs['toks']
        toks.zero_()
        toks[:L] = tokens.to(dev, torch.int32).t()
        bufs['pos'].zero_()
        bufs['stats'].zero_()
        sel = torch.arange(Bn, device=dev) if rows is None else torch.as_tensor(rows, device=dev)
        out = torch.empty((sel.numel(), L, self.vocab), dtype=torch.float32, device=dev)
        for pos in range(L):
            logits = self._step(bufs, Bn, False, use_graphs and pos > 0)
            out[:, pos] = logits.index_select(0, sel)
        bufs['eager_runs'] += 1
        return out

    def generate(self, enc: torch.Tensor, prompts: torch.Tensor, n_new: int, no_speech_id=None, keep_logits=False,
                 eos_id=None, check_every=16, early_exit_nsp=None, use_graphs=True):
        """Greedy decode up to n_new tokens after the prompt.
        enc bf16 [B,1500,d]; prompts int32 [B,P] -> (tokens int32 [B,n_new] on device,
        no_speech_prob f32 [B] or None, first_logits f32 [B,V] (after the whole prompt) if keep_logits).
        eos_id: stop once every row has produced it (checked every `check_every` tokens; rows are
        padded with eos).  early_exit_nsp: per-row max no-speech probabilities; if every row is
        above its limit nothing is generated (tokens None), as InfernSTTWorker.py:91-92 does.
        The per-token launch sequence is captured into two hipGraphs (with / without argmax) after
        the first eager run and replayed; the position lives in a device scalar."""
        dev, d = self.device, self.d
        Bn, P = prompts.shape
        n_new = min(n_new, self.max_tokens - P)
        bufs = self._dec(Bn)
        use_graphs = use_graphs and bufs['eager_runs'] >= 1
        for li, L in enumerate(self.dec_layers):
            C = L['cross']
            ops.linear(enc, C['wkv'], C['bkv'], bufs['cross'][li], rows=Bn * N_CTX, k=d, n=2 * d)
        # token matrix, position-major so that each step's ids are a dense int32[B] block
        toks = bufs['toks']
        toks.fill_(eos_id if eos_id is not None else 0)
        toks[:P] = prompts.to(dev, torch.int32).t()
        bufs['pos'].zero_()
        bufs['stats'].zero_()             # every token's last launch leaves it cleared for the next
        nsp = torch.empty(Bn, dtype=torch.float32, device=dev) if no_speech_id is not None else None
        first = None
        for pos in range(P + n_new - 1):
            gen = pos >= P - 1
            logits = self._step(bufs, Bn, gen, use_graphs and pos > 0)
            if pos == 0 and nsp is not None:
                ops.argmax_pick(bufs['logits_full'], vocab=self.vocab, nrows=Bn, ld=self.vpad, pick_token=no_speech_id,
                                pick_prob_out=nsp)
                if early_exit_nsp is not None:
                    lim = torch.tensor(early_exit_nsp, dtype=torch.float32)
                    if bool((nsp.cpu() > lim).all()):
                        return None, nsp, None
            if gen:
                if pos == P - 1 and keep_logits:
                    first = logits.clone()
                done = pos - (P - 1) + 1
                if eos_id is not None and done % check_every == 0 and done < n_new:
                    if bool((toks[P:P + done] == eos_id).any(0).all()):
                        break
        bufs['eager_runs'] += 1
        return toks[P:P + n_new].t().contiguous(), nsp, first

    def generate_beam(self, enc: torch.Tensor, prompts: torch.Tensor, n_new: int, beams: int = 5, eos_id: int = 50257,
                      length_penalty: float = 1.0, suppress=None, begin_suppress=None, no_speech_id=None, check_every=8,
                      use_graphs=True):
        """Beam search at the width ctranslate2's Whisper.generate defaults to (Cluster/InfernSTTWorker.py:61-75: beam_size 5,
        length_penalty 1, return_no_speech_prob), in transformers' formulation (GenerationMixin._beam_search: pinned by
        tests/golden/whisper_beam.npz; ctranslate2's own termination rule is unpinned), over the same per-token step `generate` uses, on
        B * beams decode rows that share each utterance's cross-attention K/V.  Search bookkeeping, token-matrix
        permutation and the per-step self-attention KV gather all run on the device (ifh_beam_step,
        ifh_kv_gather_bf16) inside the replayed hipGraph; the host only polls `alive` every `check_every` tokens.
        enc bf16 [B,1500,d]; prompts int [B,P] (same P for every row); suppress / begin_suppress: optional f32 [vocab]
        additive masks (0 / -inf) on the log-probs (every step / first generated position).
        -> (tokens int32 [B, n_new] padded with eos_id, lengths int32 [B] (eos included), scores f32 [B]
        = sum log p / length ** length_penalty, no_speech_prob f32 [B] or None)."""
        dev, d = self.device, self.d
        B, P = prompts.shape
        K = int(beams)
        assert 1 <= K <= 8
        n_new = min(n_new, self.max_tokens - P)
        rows = B * K
        bufs = self._dec(rows, K)
        use_graphs = use_graphs and bufs['eager_runs'] >= 1
        for li, L in enumerate(self.dec_layers):
            C = L['cross']
            ops.linear(enc, C['wkv'], C['bkv'], bufs['cross'][li], rows=B * N_CTX, k=d, n=2 * d)
        st = bufs.get('beam_state')
        if st is None or st.max_new != n_new:
            st = bufs['beam_state'] = ops.BeamState(B, K, n_new, dev)
        st.reset()
        masks = []
        for name, m in (('sup_buf', suppress), ('bsup_buf', begin_suppress)):     # fixed device buffers: graphs hold them
            if m is None:
                masks.append(None)
                continue
            if name not in bufs:
                bufs[name] = torch.zeros(self.vocab, dtype=torch.float32, device=dev)
            bufs[name].copy_(m.to(torch.float32))
            masks.append(bufs[name])
        bufs['beam'] = dict(state=st, P=P, max_length=P + n_new, eos=int(eos_id), lp=float(length_penalty),
                            suppress=masks[0], begin_suppress=masks[1])
        mkey = (P, n_new, int(eos_id), float(length_penalty), suppress is None, begin_suppress is None)
        if bufs.get('beam_key') != mkey:           # graphs bake these by value: drop the ones captured for another search
            for k in [k for k in bufs['graphs'] if isinstance(k, tuple)]:
                del bufs['graphs'][k]
            bufs['beam_key'] = mkey
        toks = bufs['toks']
        toks.fill_(int(eos_id))
        toks[:P] = prompts.to(dev, torch.int32).repeat_interleave(K, 0).t()
        bufs['pos'].zero_()
        bufs['stats'].zero_()
        nsp = torch.empty(B, dtype=torch.float32, device=dev) if no_speech_id is not None else None
        for pos in range(P + n_new - 1):
            gen = pos >= P - 1
            mode = ('beam', (pos - (P - 1)) & 1) if gen else False
            self._step(bufs, rows, mode, use_graphs and pos > 0)
            if pos == 0 and nsp is not None:       # rows b * K: the first beam of every utterance
                ops.argmax_pick(bufs['logits_full'], vocab=self.vocab, nrows=B, ld=K * self.vpad, pick_token=no_speech_id,
                                pick_prob_out=nsp)
            if gen:
                done = pos - (P - 1) + 1
                if done % check_every == 0 and done < n_new:
                    if int(st.alive[pos + 1].item()) == 0:           # this step ran at cur_len = pos + 1
                        break
        bufs['eager_runs'] += 1
        lens = st.fin_len[:, 0].clone()
        out = st.fin_seqs[:, 0, :].clone()
        out.masked_fill_(torch.arange(n_new, device=dev)[None, :] >= lens[:, None], int(eos_id))
        return out, lens, st.fin_scores[:, 0].clone(), nsp
