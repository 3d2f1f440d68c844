"""SpeechT5 text encoder, streaming speech decoder and postnet on the HIP device.

Replaces the transformers SpeechT5ForTextToSpeech modules the reference drives from
HelloSippyTTSRT/HelloSippyRTPipe.py:111-115 (encoder, once per batch) and :196-230 (16
decoder steps + postnet per infer() call).  Architecture per modeling_speecht5.py (v5.15.0):
post-LN layers, q pre-scaled by 64^-0.5 (folded into the q weights here), relative position
bias in the encoder, always-on prenet dropout whose Bernoulli(0.5) keep-masks are an explicit
input (one [2][256] pair per decoder step, shared across the batch as in the reference).
Only the last time row of the prenet is computed per step: rows are independent given the
mask row, so this equals the reference's prenet-over-history followed by [:, -1:].
"""
import math

import torch

from .. import _lib, ops
from ..ops import ACT_GELU, ACT_RELU, ACT_TANH, BF16
from ..weights import scaled_positional_table

D, H, FF = 768, 12, 3072
ROW_PAD = 8             # elements added to the row stride of decode-step activations (see TTSBatchState)
KVP = 2 * D             # K|V row stride of the caches (padding it like the activations was measured: 0.69 -> 0.72-0.76 ms per step)
QS = 64 ** -0.5


def _ln(sd, p, dev):
    return sd[p + '.weight'].float().contiguous().to(dev), sd[p + '.bias'].float().contiguous().to(dev)


class SpeechT5:
    n_dec_layers = 6

    def __init__(self, sd, device, max_steps=4000):
        self.device = dev = _lib.require_device(device)
        self.max_steps = max_steps
        self.use_graphs = True
        self._states = {}
        E = 'speecht5.encoder.'
        alpha_e = float(sd[E + 'prenet.encode_positions.alpha'])
        self.tok = sd[E + 'prenet.embed_tokens.weight'].to(BF16).contiguous().to(dev)
        self.pe_enc = (alpha_e * scaled_positional_table(450, D)).to(BF16).contiguous().to(dev)
        W = E + 'wrapped_encoder.'
        self.enc_ln = _ln(sd, W + 'layer_norm', dev)
        self.pe_k = ops.w_linear(sd[W + 'embed_positions.pe_k.weight'], dev)            # [320,64]
        self.enc_layers = []
        i = 0
        while (W + 'layers.%d.attention.q_proj.weight' % i) in sd:
            L = W + 'layers.%d.' % i
            A = L + 'attention.'
            wqkv = torch.cat([sd[A + 'q_proj.weight'].float() * QS, sd[A + 'k_proj.weight'].float(), sd[A + 'v_proj.weight'].float()])
            bqkv = torch.cat([sd[A + 'q_proj.bias'].float() * QS, sd[A + 'k_proj.bias'].float(), sd[A + 'v_proj.bias'].float()])
            self.enc_layers.append(dict(
                wqkv=ops.w_linear(wqkv, dev), bqkv=ops.w_bias(bqkv, dev),
                wo=ops.w_linear(sd[A + 'out_proj.weight'], dev), bo=ops.w_bias(sd[A + 'out_proj.bias'], dev),
                ln1=_ln(sd, L + 'layer_norm', dev),
                w1=ops.w_linear(sd[L + 'feed_forward.intermediate_dense.weight'], dev),
                b1=ops.w_bias(sd[L + 'feed_forward.intermediate_dense.bias'], dev),
                w2=ops.w_linear(sd[L + 'feed_forward.output_dense.weight'], dev),
                b2=ops.w_bias(sd[L + 'feed_forward.output_dense.bias'], dev),
                ln2=_ln(sd, L + 'final_layer_norm', dev)))
            i += 1
        P = 'speecht5.decoder.prenet.'
        self.p0 = (ops.w_linear(sd[P + 'layers.0.weight'], dev), ops.w_bias(sd[P + 'layers.0.bias'], dev))
        self.p1 = (ops.w_linear(sd[P + 'layers.1.weight'], dev), ops.w_bias(sd[P + 'layers.1.bias'], dev))
        self.pf = (ops.w_linear(sd[P + 'final_layer.weight'], dev), ops.w_bias(sd[P + 'final_layer.bias'], dev))
        self.ps = (ops.w_linear(sd[P + 'speaker_embeds_layer.weight'], dev), ops.w_bias(sd[P + 'speaker_embeds_layer.bias'], dev))
        alpha_d = float(sd[P + 'encode_positions.alpha'])
        self.pe_dec = (alpha_d * scaled_positional_table(4000, D)).to(BF16).contiguous().to(dev)
        Wd = 'speecht5.decoder.wrapped_decoder.'
        self.dec_layers = []
        for i in range(self.n_dec_layers):
            L = Wd + 'layers.%d.' % i
            S, C = L + 'self_attn.', L + 'encoder_attn.'
            self.dec_layers.append(dict(
                wqkv=ops.w_linear(torch.cat([sd[S + 'q_proj.weight'].float() * QS, sd[S + 'k_proj.weight'].float(),
                                             sd[S + 'v_proj.weight'].float()]), dev),
                bqkv=ops.w_bias(torch.cat([sd[S + 'q_proj.bias'].float() * QS, sd[S + 'k_proj.bias'].float(),
                                           sd[S + 'v_proj.bias'].float()]), dev),
                wo=ops.w_linear(sd[S + 'out_proj.weight'], dev), bo=ops.w_bias(sd[S + 'out_proj.bias'], dev),
                ln1=_ln(sd, L + 'self_attn_layer_norm', dev),
                cwq=ops.w_linear(sd[C + 'q_proj.weight'], dev, QS), cbq=ops.w_bias(sd[C + 'q_proj.bias'], dev, QS),
                cwkv=ops.w_linear(torch.cat([sd[C + 'k_proj.weight'].float(), sd[C + 'v_proj.weight'].float()]), dev),
                cbkv=ops.w_bias(torch.cat([sd[C + 'k_proj.bias'].float(), sd[C + 'v_proj.bias'].float()]), dev),
                cwo=ops.w_linear(sd[C + 'out_proj.weight'], dev), cbo=ops.w_bias(sd[C + 'out_proj.bias'], dev),
                ln2=_ln(sd, L + 'encoder_attn_layer_norm', dev),
                w1=ops.w_linear(sd[L + 'feed_forward.intermediate_dense.weight'], dev),
                b1=ops.w_bias(sd[L + 'feed_forward.intermediate_dense.bias'], dev),
                w2=ops.w_linear(sd[L + 'feed_forward.output_dense.weight'], dev),
                b2=ops.w_bias(sd[L + 'feed_forward.output_dense.bias'], dev),
                ln3=_ln(sd, L + 'final_layer_norm', dev)))
        O = 'speech_decoder_postnet.'
        self.feat = (ops.w_linear(sd[O + 'feat_out.weight'], dev), ops.w_bias(sd[O + 'feat_out.bias'], dev))
        self.prob = (ops.w_linear(sd[O + 'prob_out.weight'], dev), ops.w_bias(sd[O + 'prob_out.bias'], dev))
        # ---- LayerNorm-folded variants for the decode loop (ifh_conv_desc.aln_*/rln_*): the three LayerNorm
        # launches per layer disappear; gamma goes into the consumer's weights, mean/rstd are applied in its
        # epilogue from row statistics the producer's epilogue accumulated.
        self.fold_ln = True            # LayerNorms folded around the decode GEMMs (False: explicit launches)
        lnp = lambda name: (sd[name + '.weight'].float(), sd[name + '.bias'].float())
        self.dec_fold = []
        for i in range(self.n_dec_layers):
            L = Wd + 'layers.%d.' % i
            S, C = L + 'self_attn.', L + 'encoder_attn.'
            f = {}
            if i > 0:
                g3, b3 = lnp(Wd + 'layers.%d.final_layer_norm' % (i - 1))
                wq = torch.cat([sd[S + 'q_proj.weight'].float() * QS, sd[S + 'k_proj.weight'].float(), sd[S + 'v_proj.weight'].float()])
                bq = torch.cat([sd[S + 'q_proj.bias'].float() * QS, sd[S + 'k_proj.bias'].float(), sd[S + 'v_proj.bias'].float()])
                f['qkv'] = ops.w_linear_ln(wq, bq, g3, b3, dev)
                f['ln_prev'] = (g3.contiguous().to(dev), b3.contiguous().to(dev))
            g1, b1 = lnp(L + 'self_attn_layer_norm')
            g2, b2 = lnp(L + 'encoder_attn_layer_norm')
            f['cq'] = ops.w_linear_ln(sd[C + 'q_proj.weight'], sd[C + 'q_proj.bias'], g1, b1, dev, scale=QS)
            f['ln1'] = (g1.contiguous().to(dev), b1.contiguous().to(dev))
            f['ff1'] = ops.w_linear_ln(sd[L + 'feed_forward.intermediate_dense.weight'],
                                       sd[L + 'feed_forward.intermediate_dense.bias'], g2, b2, dev)
            f['ln2'] = (g2.contiguous().to(dev), b2.contiguous().to(dev))
            self.dec_fold.append(f)
        gl, bl = lnp(Wd + 'layers.%d.final_layer_norm' % (self.n_dec_layers - 1))
        self.feat_fold = ops.w_linear_ln(sd[O + 'feat_out.weight'], sd[O + 'feat_out.bias'], gl, bl, dev)
        self.prob_fold = ops.w_linear_ln(torch.cat([sd[O + 'prob_out.weight'].float(), torch.zeros(14, D)]),
                                         torch.cat([sd[O + 'prob_out.bias'].float(), torch.zeros(14)]), gl, bl, dev)
        self.postnet = []
        for i in range(5):                  # fold eval-mode BatchNorm into the (bias-free) conv
            b = O + 'layers.%d.batch_norm.' % i
            s = sd[b + 'weight'].float() / torch.sqrt(sd[b + 'running_var'].float() + 1e-5)
            shift = sd[b + 'bias'].float() - sd[b + 'running_mean'].float() * s
            self.postnet.append((ops.w_conv(sd[O + 'layers.%d.conv.weight' % i], dev, scale_per_out=s), shift.contiguous().to(dev)))

    # ---- text encoder (HelloSippyRTPipe.py:111-115) -------------------------------------------
    def encode(self, input_ids: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """input_ids int32 [B,T] (right-padded), lens int32 [B] -> bf16 [B,T,768]"""
        dev = self.device
        Bn, T = input_ids.shape
        rows = Bn * T
        e = lambda *s, dt=BF16: torch.empty(s, dtype=dt, device=dev)
        ids = input_ids.to(dev, torch.int32).contiguous()
        lens = lens.to(dev, torch.int32).contiguous()
        x, t1 = e(rows, D), e(rows, D)
        ops.embed(ids, self.tok, self.pe_enc, t1, n=rows, dim=D, pos0=0, seq_len=T)
        ops.layernorm(t1, *self.enc_ln, x, rows, D)
        qkv, att, ff = e(rows, 3 * D), e(rows, D), e(rows, FF)
        rel = e(rows, H, 320, dt=torch.float32)
        for L in self.enc_layers:
            ops.linear(x, L['wqkv'], L['bqkv'], qkv, rows=rows, k=D, n=3 * D)
            # relative position bias table R[b,t,h,:] = q_h . pe_k^T  (q already scaled)
            ops.conv(qkv, self.pe_k, None, rel, nbatch=rows, t_in=H, t_out=H, cin=64, n=320, lda=64, x_bstride=3 * D)
            ops.attn_prefill(qkv, qkv, qkv, att, nbatch=Bn, nheads=H, tq=T, tk=T, k_off=D, v_off=2 * D,
                             q_ts=3 * D, k_ts=3 * D, v_ts=3 * D, o_ts=D, key_len=lens, relbias=rel, nrel=320)
            ops.linear(att, L['wo'], L['bo'], t1, rows=rows, k=D, n=D, resid=x)
            ops.layernorm(t1, *L['ln1'], x, rows, D)
            ops.linear(x, L['w1'], L['b1'], ff, rows=rows, k=D, n=FF, act=ACT_GELU)
            ops.linear(ff, L['w2'], L['b2'], t1, rows=rows, k=FF, n=D, resid=x)
            ops.layernorm(t1, *L['ln2'], x, rows, D)
        return x.view(Bn, T, D)


class TTSBatchState:
    """Device-resident HelloSippyPipeStateBatched (HelloSippyRTPipe.py:81-121).  Buffers are
    allocated once per (B, T) and reused by later batches of the same shape (`acquire`), so the
    hipGraphs captured over them stay valid."""

    T_BUCKET = 16          # text lengths are padded (masked) up to a multiple of this so shapes -- and graphs -- recur
    MAX_CACHED = 4

    def __init__(self, model: 'SpeechT5', B: int, T: int):
        dev = model.device
        self.model, self.B, self.T = model, B, T            # T = allocated (bucketed) text length
        self.maxlen = int(T * 20.0 / 2)
        self.minlen = 0
        e = lambda *s, dt=BF16: torch.empty(s, dtype=dt, device=dev)
        self.enc_len = torch.zeros(B, dtype=torch.int32, device=dev)
        self.cross = [e(B * T, KVP) for _ in model.dec_layers]
        self.smax = min(model.max_steps, self.maxlen + 48)      # maxlen here is the bucket's upper bound; the loop
        # runs at most one 16-step call past maxlen before every row has ended
        self.self_kv = [torch.zeros((B, self.smax, KVP), dtype=BF16, device=dev) for _ in model.dec_layers]
        # two frame buffers (call parity): frame 0 = last frame carried over from the previous call, frames
        # 1..32 = this call.  Double-buffered so the renderer (postnet/vocoder) of call c can run on a second
        # stream while the decoder already writes call c+1.
        self.spec = [torch.zeros((B, 33, 80), dtype=BF16, device=dev) for _ in range(2)]
        self.cat = torch.zeros((B, D + 512), dtype=BF16, device=dev)
        self.pre_frames = torch.zeros((B, 4, 80), dtype=BF16, device=dev)
        self.starts_at = torch.full((B,), 1, dtype=torch.int64, device=dev)
        self.ends_at = torch.full((B,), -1, dtype=torch.int64, device=dev)
        self.post = [e(B, 32, 80), e(B, 32, 80)]
        self.ncalls = 0
        self.pos_dev = torch.zeros(1, dtype=torch.int32, device=dev)       # decoder position, read by the kernels
        # {minlen, maxlen} of the batch that currently occupies this state, read by the stop kernels: the captured
        # step graphs are shared by every batch of the (B, bucketed T) slot, whose true lengths differ
        self.lens_dev = torch.tensor([self.minlen, self.maxlen], dtype=torch.int32, device=dev)
        self.masks = torch.zeros((16, 2, 256), dtype=torch.uint8, device=dev)
        # Row stride of the step's activation buffers: D + 8 elements, i.e. 16 bytes past a multiple of 128.  The skinny
        # GEMM reads 16 rows x 64 bytes per wave-instruction; with rows a whole number of 128-byte lines apart those
        # pieces collide in the vector L1 (measured at 256 rows: qkv 13.2 -> 9.7 us, ff1 15.5 -> 11.6, ff2 18.9 -> 14.5; the whole
        # step 0.78 -> 0.66 ms).  Padding the weight rows the same way changes nothing (tools/probe_skinny.py).
        self.DP, self.FP = D + ROW_PAD, FF + ROW_PAD
        self.h1, self.h2, self.x = e(B, 256), e(B, 256), e(B, D)
        self.q, self.att, self.t1, self.ff = e(B, self.DP), e(B, self.DP), e(B, self.DP), e(B, self.FP)
        self.plog = e(B, 2, dt=torch.float32)
        self.plog16 = e(B, 16, dt=torch.float32)            # LN-folded path: stop logits in cols 0..1 of a 16-wide tile
        self.t2, self.t3, self.x0 = e(B, self.DP), e(B, self.DP), e(B, self.DP)
        self.stat_rows = max(64, -(-B // 16) * 16)
        self.stats = torch.zeros((3 * len(model.dec_layers), self.stat_rows, 2), dtype=torch.int64, device=dev)
        self.pn = [e(B, 32, 256), e(B, 32, 256)]
        self.graphs = {}
        self.eager_calls = 0
        self.idx = 0
        self.audio = None

    @classmethod
    def acquire(cls, model: 'SpeechT5', input_ids, lens, speakers):
        B, T_true = input_ids.shape
        T = -(-T_true // cls.T_BUCKET) * cls.T_BUCKET
        if T != T_true:                           # extra right padding: token 0, masked by lens like the reference's own padding
            input_ids = torch.nn.functional.pad(input_ids, (0, T - T_true))
        st = model._states.pop((B, T), None)
        if st is None:
            st = cls(model, B, T)
            while len(model._states) >= cls.MAX_CACHED:          # small LRU of state shapes (buffers + captured graphs)
                model._states.pop(next(iter(model._states)))
        model._states[(B, T)] = st
        st.reset(input_ids, lens, speakers)
        st.set_lengths(0, int(T_true * 20.0 / 2))  # HelloSippyRTPipe.py:117-118 use the batch's true padded length
        return st

    def set_lengths(self, minlen: int, maxlen: int):
        self.minlen, self.maxlen = minlen, maxlen
        self.lens_dev.copy_(torch.tensor([minlen, maxlen], dtype=torch.int32), non_blocking=False)

    def reset(self, input_ids, lens, speakers):
        model, dev, B, T = self.model, self.model.device, self.B, self.T
        self.enc_len.copy_(lens.to(torch.int32))
        self.enc = model.encode(input_ids, lens)
        for L, kv in zip(model.dec_layers, self.cross):
            ops.linear(self.enc, L['cwkv'], L['cbkv'], kv, rows=B * T, k=D, n=2 * D, ldc=KVP)
        self.spec[0].zero_()
        self.spec[1].zero_()
        self.stats.zero_()                # each decoder step leaves it cleared for the next; start from a known state
        self.ncalls = 0
        self.pre_frames.zero_()
        self.ends_at.fill_(-1)
        self.pos_dev.zero_()
        self.idx = 0
        self.audio = None
        spk = speakers.to(dev, BF16).contiguous().view(B, 512)
        _lib.check(_lib.lib().ifh_l2norm_rows_bf16(ops._addr(spk), 512, B, ops._addr(self.cat, D), D + 512,
                                                   _lib.stream_ptr(dev)), 'ifh_l2norm_rows_bf16')


def _decoder_step(model: 'SpeechT5', st: TTSBatchState, s: int, threshold: float, par: int):
    """One decoder step; everything that depends on the global position reads st.pos_dev on the
    device, so the launch sequence is identical for every step with the same in-call index s.
    (The scratch buffers are used as dense [B, D] / [B, FF] storage here; the folded step pads their rows.)"""
    dev = model.device
    B, T = st.B, st.T
    masks = st.masks
    spec = st.spec[par]
    ops.linear(spec, *model.p0, st.h1, rows=B, k=80, n=256, x_off=2 * s * 80, lda=33 * 80, act=ACT_RELU,
               colmask=masks, colmask_off=(s * 2) * 256)
    ops.linear(st.h1, *model.p1, st.h2, rows=B, k=256, n=256, act=ACT_RELU, colmask=masks, colmask_off=(s * 2 + 1) * 256)
    ops.linear(st.h2, *model.pf, st.cat, rows=B, k=256, n=D, ldc=D + 512, resid=model.pe_dec, resid_ld=0, resid_bstride=0,
               dyn_pos=st.pos_dev, dyn_resid_mul=D)
    ops.linear(st.cat, *model.ps, st.x, rows=B, k=D + 512, n=D, act=ACT_RELU)
    x = st.x
    for li, L in enumerate(model.dec_layers):
        kv = st.self_kv[li]
        # one launch: q -> scratch, K|V -> appended to the cache at the device-held position
        ops.conv(x, L['wqkv'], L['bqkv'], st.q, nbatch=B, t_in=1, t_out=1, cin=D, n=3 * D, ldc=D, out_bstride=D,
                 dyn_pos=st.pos_dev, n_split=D, out2=kv, out2_bstride=st.smax * KVP, ldc2=KVP, dyn_ooff2_mul=1)
        ops.attn_decode(st.q, kv, kv, st.att, nbatch=B, nheads=H, max_keys=st.smax, q_bs=D, kv_bs=st.smax * KVP,
                        kv_ts=KVP, o_bs=D, v_off=D, dyn_len=st.pos_dev, dyn_add=1)
        ops.linear(st.att, L['wo'], L['bo'], st.t1, rows=B, k=D, n=D, resid=x)
        ops.layernorm(st.t1, *L['ln1'], st.x, B, D)
        ops.linear(st.x, L['cwq'], L['cbq'], st.q, rows=B, k=D, n=D)
        ck = st.cross[li]
        ops.attn_decode(st.q, ck, ck, st.att, nbatch=B, nheads=H, max_keys=T, q_bs=D, kv_bs=T * KVP, kv_ts=KVP,
                        o_bs=D, v_off=D, key_len=st.enc_len)
        ops.linear(st.att, L['cwo'], L['cbo'], st.t1, rows=B, k=D, n=D, resid=st.x)
        ops.layernorm(st.t1, *L['ln2'], st.x, B, D)
        ops.linear(st.x, L['w1'], L['b1'], st.ff, rows=B, k=D, n=FF, act=ACT_GELU)
        ops.linear(st.ff, L['w2'], L['b2'], st.t1, rows=B, k=FF, n=D, resid=st.x)
        ops.layernorm(st.t1, *L['ln3'], st.x, B, D)
        x = st.x
    # two new mel frames -> frames 2s+1, 2s+2 ; stop logits ; advance the device position
    ops.linear(x, *model.feat, spec, rows=B, k=D, n=160, out_off=(2 * s + 1) * 80, ldc=33 * 80)
    ops.linear(x, *model.prob, st.plog, rows=B, k=D, n=2)
    _lib.check(_lib.lib().ifh_tts_stop_update(ops._addr(st.plog), ops._addr(st.ends_at), B, 0, st.minlen, st.maxlen,
                                              threshold, 2, ops._addr(st.pos_dev), 2, ops._addr(st.lens_dev), _lib.stream_ptr(dev)),
               'ifh_tts_stop_update')
    ops.add_i32(st.pos_dev, 1)


def _decoder_step_folded(model: 'SpeechT5', st: TTSBatchState, s: int, threshold: float, par: int):
    """Same step with every LayerNorm folded around the neighbouring GEMMs (no LayerNorm launches):
    producers accumulate row statistics in their epilogue, consumers apply mean/rstd in theirs."""
    dev = model.device
    B, T = st.B, st.T
    masks, spec, stats = st.masks, st.spec[par], st.stats
    DP, FP = st.DP, st.FP
    SO = st.stat_rows * 2                            # int64 elements per stats slot ([rows][2]); zero on entry: cleared by
                                                     # the last launch of the previous step (ifh_tts_stop_advance)
    ops.linear(spec, *model.p0, st.h1, rows=B, k=80, n=256, x_off=2 * s * 80, lda=33 * 80, act=ACT_RELU,
               colmask=masks, colmask_off=(s * 2) * 256)
    ops.linear(st.h1, *model.p1, st.h2, rows=B, k=256, n=256, act=ACT_RELU, colmask=masks, colmask_off=(s * 2 + 1) * 256)
    ops.linear(st.h2, *model.pf, st.cat, rows=B, k=256, n=D, ldc=D + 512, resid=model.pe_dec, resid_ld=0, resid_bstride=0,
               dyn_pos=st.pos_dev, dyn_resid_mul=D)
    ops.linear(st.cat, *model.ps, st.x0, rows=B, k=D + 512, n=D, act=ACT_RELU, ldc=DP)
    nl = len(model.dec_layers)
    ld = dict(lda=DP, ldc=DP, resid_ld=DP)
    for li, (L, F) in enumerate(zip(model.dec_layers, model.dec_fold)):
        kv = st.self_kv[li]
        s1, s2, s3, s3p = (3 * li) * SO, (3 * li + 1) * SO, (3 * li + 2) * SO, (3 * li - 1) * SO
        kvargs = dict(nbatch=B, t_in=1, t_out=1, cin=D, n=3 * D, lda=DP, ldc=DP, out_bstride=DP, dyn_pos=st.pos_dev, n_split=D,
                      out2=kv, out2_bstride=st.smax * KVP, ldc2=KVP, dyn_ooff2_mul=1)
        if li == 0:
            ops.conv(st.x0, L['wqkv'], L['bqkv'], st.q, **kvargs)
        else:
            w, c2, c1 = F['qkv']
            ops.conv(st.t3, w, c2, st.q, aln=(stats, s3p, c1), ln_dim=D, **kvargs)
        ops.attn_decode(st.q, kv, kv, st.att, nbatch=B, nheads=H, max_keys=st.smax, q_bs=DP, kv_bs=st.smax * KVP,
                        kv_ts=KVP, o_bs=DP, v_off=D, dyn_len=st.pos_dev, dyn_add=1)
        if li == 0:
            ops.linear(st.att, L['wo'], L['bo'], st.t1, rows=B, k=D, n=D, resid=st.x0, stats_out=stats, stats_off=s1, ln_dim=D, **ld)
        else:
            ops.linear(st.att, L['wo'], L['bo'], st.t1, rows=B, k=D, n=D, resid=st.t3, rln=(stats, s3p) + F['ln_prev'],
                       stats_out=stats, stats_off=s1, ln_dim=D, **ld)
        w, c2, c1 = F['cq']
        ops.linear(st.t1, w, c2, st.q, rows=B, k=D, n=D, aln=(stats, s1, c1), ln_dim=D, lda=DP, ldc=DP)
        ck = st.cross[li]
        ops.attn_decode(st.q, ck, ck, st.att, nbatch=B, nheads=H, max_keys=T, q_bs=DP, kv_bs=T * KVP, kv_ts=KVP,
                        o_bs=DP, v_off=D, key_len=st.enc_len)
        ops.linear(st.att, L['cwo'], L['cbo'], st.t2, rows=B, k=D, n=D, resid=st.t1, rln=(stats, s1) + F['ln1'],
                   stats_out=stats, stats_off=s2, ln_dim=D, **ld)
        w, c2, c1 = F['ff1']
        ops.linear(st.t2, w, c2, st.ff, rows=B, k=D, n=FF, act=ACT_GELU, aln=(stats, s2, c1), ln_dim=D, lda=DP, ldc=FP)
        ops.linear(st.ff, L['w2'], L['b2'], st.t3, rows=B, k=FF, n=D, resid=st.t2, rln=(stats, s2) + F['ln2'],
                   stats_out=stats, stats_off=s3, ln_dim=D, lda=FP, ldc=DP, resid_ld=DP)
    sl = (3 * nl - 1) * SO
    w, c2, c1 = model.feat_fold
    ops.linear(st.t3, w, c2, spec, rows=B, k=D, n=160, out_off=(2 * s + 1) * 80, ldc=33 * 80, aln=(stats, sl, c1), ln_dim=D, lda=DP)
    w, c2, c1 = model.prob_fold
    ops.linear(st.t3, w, c2, st.plog16, rows=B, k=D, n=16, aln=(stats, sl, c1), ln_dim=D, lda=DP)
    _lib.check(_lib.lib().ifh_tts_stop_advance(ops._addr(st.plog16), ops._addr(st.ends_at), B, st.minlen, st.maxlen,
                                               threshold, 2, ops._addr(st.pos_dev), 16, ops._addr(st.stats),
                                               st.stats.numel() * 8, ops._addr(st.lens_dev), _lib.stream_ptr(dev)), 'ifh_tts_stop_advance')


def decoder_steps(model: 'SpeechT5', st: TTSBatchState, masks: torch.Tensor, nsteps=16, threshold=0.5, use_graphs=None):
    """The while-loop of HelloSippyRTPipe.infer (:195-229).  masks uint8 [nsteps,2,256] on device.
    Each of the 16 in-call step shapes is captured once into a hipGraph (launch-bound inner loop:
    ~80 small kernels per step) and replayed for every later call on this state."""
    assert nsteps <= 16 and st.idx + nsteps <= st.smax, 'decoder step budget exceeded'
    use_graphs = model.use_graphs if use_graphs is None else use_graphs
    par = st.ncalls & 1
    st.masks[:nsteps].copy_(masks)
    st.spec[par][:, 0, :].copy_(st.spec[1 - par][:, 32, :])       # carry the last produced frame
    # the first call on a state shape runs eagerly (loads every kernel); graphs are captured from the second on
    use_graphs = use_graphs and st.eager_calls >= 2
    step_fn = _decoder_step_folded if (model.fold_ln and st.B <= 256) else _decoder_step
    for s in range(nsteps):
        if not use_graphs:
            step_fn(model, st, s, threshold, par)
        else:
            g = st.graphs.get((s, threshold, par, step_fn is _decoder_step_folded))
            if g is None:
                g = _lib.CountedGraph(lambda: step_fn(model, st, s, threshold, par))
                st.graphs[(s, threshold, par, step_fn is _decoder_step_folded)] = g
            g.replay()
        st.idx += 1
    if not use_graphs:
        st.eager_calls += 1


def postnet(model: SpeechT5, st, par: int, B: int = None):
    """speech_decoder_postnet.postnet on the 32 new frames (:230) -> st.post[par] bf16 [B,32,80] (first B rows of a ragged state)"""
    B = st.B if B is None else B
    spec = st.spec[par]
    src, off, cin = spec, 80, 80
    for i, (w, shift) in enumerate(model.postnet):
        last = i == 4
        out = st.post[par] if last else st.pn[i % 2]
        cout = 80 if last else 256
        ops.conv(src, w, shift, out, nbatch=B, t_in=32, t_out=32, cin=cin, n=cout, taps=5, pad=2, x_off=off,
                 x_bstride=(33 * 80 if i == 0 else 32 * cin), lda=cin, act=(0 if last else ACT_TANH),
                 resid=(spec if last else None), resid_off=(80 if last else 0), resid_ld=80, resid_bstride=33 * 80)
        src, off, cin = out, 0, cout
    return st.post[par]


# ---- continuous (ragged-position) batching of the decode loop ------------------------------------------------------
class TTSRaggedState:
    """Device tables of ONE running decode batch whose rows joined at different infer() calls and therefore sit at
    different decoder positions.  The reference freezes a batch at process_batch entry and loops it to the end
    (Cluster/InfernTTSWorker.py:83-92), so the GPU sees as many small launch chains as there are batches in flight;
    here every in-flight utterance is a row slot of one step whose position-dependent quantities -- KV append row,
    self-attention key count, positional-encoding row, stop-rule index and lengths -- are per-row device vectors
    (`pos`, `minmax`, `enc_len`), so one captured launch sequence per in-call step serves all of them.  Rows join at an
    infer() boundary (`pos` = 0: zero carry frame, zero pre-frames via `fresh`) and leave when their utterance ends;
    a slot without a live utterance (`active` = 0) is still computed -- as an ended row is in the reference
    (HelloSippyRTPipe.py:254-255 "still occupies its batch slot") -- with its position frozen.
    Arithmetic per row is that of TTSBatchState's LayerNorm-folded step: same kernels, same K order."""

    def __init__(self, model: 'SpeechT5', max_rows: int, max_text: int):
        dev = model.device
        R = -(-max_rows // 16) * 16
        T = -(-max_text // TTSBatchState.T_BUCKET) * TTSBatchState.T_BUCKET
        assert R <= 1024, 'the LayerNorm-folded decode GEMMs take at most 1024 rows per launch'
        self.model, self.R, self.T = model, R, T
        # a multiple of 16 (an infer() call appends 16 rows): ContinuousTTS ends a group before a call that would pass smax, so
        # no row ever appends at pos >= smax (a cache below 10 T + 48 rows cuts long utterances; the session gets its end marker)
        self.smax = min(model.max_steps, int(T * 20.0 / 2) + 48) // 16 * 16
        assert self.smax >= 32, 'SpeechT5.max_steps leaves no room for two infer() calls'
        z = lambda *s, dt=BF16: torch.zeros(s, dtype=dt, device=dev)
        self.pos = z(R, dt=torch.int32)
        self.active = z(R, dt=torch.uint8)
        self.minmax = z(R, 2, dt=torch.int32)
        self.enc_len = z(R, dt=torch.int32)
        self.ends_at = torch.full((R,), -1, dtype=torch.int64, device=dev)
        self.fresh = [z(R, dt=torch.uint8) for _ in range(2)]        # per frame-buffer parity: rows whose first call this is
        self.cross = [z(R * T, KVP) for _ in model.dec_layers]
        self.self_kv = [z(R, self.smax, KVP) for _ in model.dec_layers]
        self.spec = [z(R, 33, 80) for _ in range(2)]
        self.cat = z(R, D + 512)
        self.pre_frames = z(R, 4, 80)
        self.post = [z(R, 32, 80), z(R, 32, 80)]
        self.masks = z(16, 2, 256, dt=torch.uint8)
        self.DP, self.FP = D + ROW_PAD, FF + ROW_PAD
        self.h1, self.h2 = z(R, 256), z(R, 256)
        self.x0, self.q, self.att, self.t1, self.t2, self.t3 = (z(R, self.DP) for _ in range(6))
        self.ff = z(R, self.FP)
        self.plog16 = z(R, 16, dt=torch.float32)
        self.replay_prof = None          # [seconds, launches] of the step graphs' replays when a caller wants them (bench.py)
        self.stat_rows = max(64, R)
        self.stats = z(3 * len(model.dec_layers), self.stat_rows, 2, dt=torch.int64)
        self.pn = [z(R, 32, 256), z(R, 32, 256)]
        self.graphs, self.eager = {}, {}
        self.ncalls = 0


def _decoder_step_ragged(model: 'SpeechT5', st: TTSRaggedState, s: int, threshold: float, par: int, n: int):
    """_decoder_step_folded over the first n row slots of a ragged state: identical launches, but whatever depended on
    the batch's one position reads the row's own (`dyn_stride=1`, per-row key counts, per-row stop rule)."""
    dev = model.device
    T = st.T
    masks, spec, stats = st.masks, st.spec[par], st.stats
    DP, FP = st.DP, st.FP
    SO = st.stat_rows * 2
    # (decode_step: the un-folded launches take the K-split streaming kernel at every row count, like the folded ones)
    ops.linear(spec, *model.p0, st.h1, rows=n, k=80, n=256, x_off=2 * s * 80, lda=33 * 80, act=ACT_RELU,
               colmask=masks, colmask_off=(s * 2) * 256, decode_step=True)
    ops.linear(st.h1, *model.p1, st.h2, rows=n, k=256, n=256, act=ACT_RELU, colmask=masks, colmask_off=(s * 2 + 1) * 256,
               decode_step=True)
    ops.linear(st.h2, *model.pf, st.cat, rows=n, k=256, n=D, ldc=D + 512, resid=model.pe_dec, resid_ld=0, resid_bstride=0,
               dyn_pos=st.pos, dyn_stride=1, dyn_resid_mul=D, decode_step=True)
    ops.linear(st.cat, *model.ps, st.x0, rows=n, k=D + 512, n=D, act=ACT_RELU, ldc=DP, decode_step=True)
    nl = len(model.dec_layers)
    ld = dict(lda=DP, ldc=DP, resid_ld=DP)
    for li, (L, F) in enumerate(zip(model.dec_layers, model.dec_fold)):
        kv = st.self_kv[li]
        s1, s2, s3, s3p = (3 * li) * SO, (3 * li + 1) * SO, (3 * li + 2) * SO, (3 * li - 1) * SO
        kvargs = dict(nbatch=n, t_in=1, t_out=1, cin=D, n=3 * D, lda=DP, ldc=DP, out_bstride=DP, dyn_pos=st.pos, dyn_stride=1,
                      n_split=D, out2=kv, out2_bstride=st.smax * KVP, ldc2=KVP, dyn_ooff2_mul=1, decode_step=True)
        if li == 0:
            ops.conv(st.x0, L['wqkv'], L['bqkv'], st.q, **kvargs)
        else:
            w, c2, c1 = F['qkv']
            ops.conv(st.t3, w, c2, st.q, aln=(stats, s3p, c1), ln_dim=D, **kvargs)
        ops.attn_decode(st.q, kv, kv, st.att, nbatch=n, nheads=H, max_keys=st.smax, q_bs=DP, kv_bs=st.smax * KVP,
                        kv_ts=KVP, o_bs=DP, v_off=D, key_len=st.pos, dyn_add=1)
        if li == 0:
            ops.linear(st.att, L['wo'], L['bo'], st.t1, rows=n, k=D, n=D, resid=st.x0, stats_out=stats, stats_off=s1, ln_dim=D, **ld)
        else:
            ops.linear(st.att, L['wo'], L['bo'], st.t1, rows=n, k=D, n=D, resid=st.t3, rln=(stats, s3p) + F['ln_prev'],
                       stats_out=stats, stats_off=s1, ln_dim=D, **ld)
        w, c2, c1 = F['cq']
        ops.linear(st.t1, w, c2, st.q, rows=n, k=D, n=D, aln=(stats, s1, c1), ln_dim=D, lda=DP, ldc=DP)
        ck = st.cross[li]
        ops.attn_decode(st.q, ck, ck, st.att, nbatch=n, nheads=H, max_keys=T, q_bs=DP, kv_bs=T * KVP, kv_ts=KVP,
                        o_bs=DP, v_off=D, key_len=st.enc_len)
        ops.linear(st.att, L['cwo'], L['cbo'], st.t2, rows=n, k=D, n=D, resid=st.t1, rln=(stats, s1) + F['ln1'],
                   stats_out=stats, stats_off=s2, ln_dim=D, **ld)
        w, c2, c1 = F['ff1']
        ops.linear(st.t2, w, c2, st.ff, rows=n, k=D, n=FF, act=ACT_GELU, aln=(stats, s2, c1), ln_dim=D, lda=DP, ldc=FP)
        ops.linear(st.ff, L['w2'], L['b2'], st.t3, rows=n, k=FF, n=D, resid=st.t2, rln=(stats, s2) + F['ln2'],
                   stats_out=stats, stats_off=s3, ln_dim=D, lda=FP, ldc=DP, resid_ld=DP)
    sl = (3 * nl - 1) * SO
    w, c2, c1 = model.feat_fold
    ops.linear(st.t3, w, c2, spec, rows=n, k=D, n=160, out_off=(2 * s + 1) * 80, ldc=33 * 80, aln=(stats, sl, c1), ln_dim=D, lda=DP)
    w, c2, c1 = model.prob_fold
    ops.linear(st.t3, w, c2, st.plog16, rows=n, k=D, n=16, aln=(stats, sl, c1), ln_dim=D, lda=DP)
    _lib.check(_lib.lib().ifh_tts_stop_advance_rows(ops._addr(st.plog16), ops._addr(st.ends_at), n, threshold, 2, ops._addr(st.pos),
                                                    ops._addr(st.active), ops._addr(st.minmax), 16, ops._addr(st.stats),
                                                    st.stats.numel() * 8, _lib.stream_ptr(dev)), 'ifh_tts_stop_advance_rows')


def ragged_decoder_steps(model: 'SpeechT5', st: TTSRaggedState, masks: torch.Tensor, n: int, nsteps=16, threshold=0.5,
                         use_graphs=None, sync_every=0):
    """One infer() call's decoder steps (HelloSippyRTPipe.py:195-229) for the first n row slots of a ragged state.
    The same [nsteps,2,256] dropout keep-masks serve every row of the step, as the reference shares one mask across
    its batch (modeling_speecht5.py:671-674).  One hipGraph per (in-call step, frame-buffer parity, n)."""
    assert nsteps <= 16 and n <= st.R and n % 16 == 0
    use_graphs = model.use_graphs if use_graphs is None else use_graphs
    par = st.ncalls & 1
    st.masks[:nsteps].copy_(masks)
    _lib.check(_lib.lib().ifh_tts_carry_rows_bf16(ops._addr(st.spec[1 - par]), ops._addr(st.spec[par]), ops._addr(st.pos), n, 33,
                                                  _lib.stream_ptr(model.device)), 'ifh_tts_carry_rows_bf16')
    use_graphs = use_graphs and st.eager.get(n, 0) >= 1            # the first call at a row count runs eagerly (loads kernels)
    for s in range(nsteps):
        if not use_graphs:
            _decoder_step_ragged(model, st, s, threshold, par, n)
        else:
            key = (s, threshold, par, n)
            g = st.graphs.get(key)
            if g is None:
                g = st.graphs[key] = _lib.CountedGraph(lambda: _decoder_step_ragged(model, st, s, threshold, par, n))
            if st.replay_prof is not None:          # (statistics: host seconds inside hipGraphLaunch, and launches)
                import time as _t
                _a = _t.perf_counter()
                g.replay()
                st.replay_prof[0] += _t.perf_counter() - _a
                st.replay_prof[1] += 1
            else:
                g.replay()
            # bounded queue depth: whatever this stream has queued stands in front of a real-time tick whose launches land on the
            # same hardware queue (16 steps = ~900 kernels); waiting every few steps keeps that to a handful of milliseconds
            if sync_every and (s + 1) % sync_every == 0 and s + 1 < nsteps:
                torch.cuda.current_stream(model.device).synchronize()
    if not use_graphs:
        st.eager[n] = st.eager.get(n, 0) + 1
    st.ncalls += 1
    return par
