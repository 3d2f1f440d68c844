"""Tensor-level wrappers over the dense C-ABI entry points (ifh_conv_bf16, ifh_layernorm_bf16,
ifh_attn_*).  torch tensors are used only as device memory; every FLOP happens in
libinfernos_hip.so.  All activations are bf16, channels-last.
"""
import ctypes

import torch

from . import _lib
from ._lib import AttnDesc, ConvDesc

ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_LRELU, ACT_SIGMOID, ACT_SILU_GLU = range(7)
BF16 = torch.bfloat16


def _addr(t, off=0):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def conv(x, w, bias, out, *, nbatch, t_in, t_out, cin, n, taps=1, stride=1, dil=1, pad=0,
         x_off=0, lda=None, x_bstride=None, act=ACT_NONE, act_slope=0.0, pre_slope=1.0, colmask=None, colmask_off=0,
         resid=None, resid_off=0, resid_ld=None, resid_bstride=None, scale=1.0, accumulate=False,
         out_off=0, ldc=None, out_bstride=None, ostride=1, ooff=0, dyn_pos=None, dyn_ooff_mul=0, dyn_resid_mul=0, dyn_stride=0, decode_step=False, convt_cout=0,
         n_split=0, out2=None, out2_bstride=0, ldc2=0, ooff2=0, dyn_ooff2_mul=0,
         aln=None, rln=None, stats_out=None, stats_off=0, ln_dim=0, ln_eps=1e-5, ln_rms=False, splitk_ws=None, argmax_keys=None, whole_chip=False):
    # aln = (stats, stats_off, c1); rln = (stats, stats_off, gamma, beta); splitk_ws = f32 workspace of the caller's decode state;
    # argmax_keys = zeroed int64 [rows] (ifh_conv_desc.argmax_keys; argmax_supported(), argmax_keys_finish())
    """One implicit-GEMM launch (see ifh_conv_desc).  Strides default to dense [nbatch][t][c]."""
    d = ConvDesc()
    lda = cin if lda is None else lda
    ldc = n if ldc is None else ldc
    d.x, d.x_bstride, d.lda = _addr(x, x_off), (t_in * lda if x_bstride is None else x_bstride), lda
    d.cin, d.taps, d.stride, d.dil, d.pad = cin, taps, stride, dil, pad
    d.t_in, d.t_out, d.nbatch = t_in, t_out, nbatch
    d.w, d.n = _addr(w), n
    d.bias, d.colmask = _addr(bias), _addr(colmask, colmask_off)
    d.pre_slope, d.act, d.act_slope = pre_slope, act, act_slope
    d.resid = _addr(resid, resid_off)
    d.resid_ld = ldc if resid_ld is None else resid_ld
    d.resid_bstride = (t_out * ostride * d.resid_ld if resid_bstride is None else resid_bstride)
    d.out_scale, d.accumulate = scale, int(accumulate)
    d.out, d.out_f32 = _addr(out, out_off), int(out.dtype == torch.float32)
    d.out_bstride = (t_out * ostride * ldc if out_bstride is None else out_bstride)
    d.ldc, d.ostride, d.ooff = ldc, ostride, ooff
    d.dyn_pos, d.dyn_ooff_mul, d.dyn_resid_mul, d.dyn_stride = _addr(dyn_pos), dyn_ooff_mul, dyn_resid_mul, dyn_stride
    d.decode_step, d.convt_cout = int(decode_step), convt_cout
    d.splitk_ws, d.splitk_ws_floats = _addr(splitk_ws), (splitk_ws.numel() if splitk_ws is not None else 0)
    d.argmax_keys = _addr(argmax_keys)
    d.whole_chip = int(whole_chip)
    d.n_split, d.out2, d.out2_bstride, d.ldc2, d.ooff2, d.dyn_ooff2_mul = n_split, _addr(out2), out2_bstride, ldc2, ooff2, dyn_ooff2_mul
    if aln is not None:
        d.aln_stats, d.aln_c1 = _addr(aln[0], aln[1]), _addr(aln[2])
    if rln is not None:
        d.rln_stats, d.rln_gamma, d.rln_beta = _addr(rln[0], rln[1]), _addr(rln[2]), _addr(rln[3])
    d.stats_out, d.ln_dim, d.ln_eps, d.ln_rms = _addr(stats_out, stats_off), ln_dim, ln_eps, int(ln_rms)
    _lib.check(_lib.lib().ifh_conv_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_conv_bf16')
    return out


def resblock_pair(x, w1, b1, w2, b2, out, *, nbatch, t, c, taps, dil, slope=0.1, scale=1.0, accumulate=False):
    """out = (x + conv2(lrelu(conv1(lrelu(x); taps, dil)); taps, 1)) * scale (+ out): one launch, the
    intermediate stays in LDS (ifh_resblock_pair_bf16).  x, out dense bf16 [nbatch][t][c]."""
    d = _lib.ResblockDesc()
    d.x, d.x_bstride = _addr(x), t * c
    d.c, d.taps, d.dil, d.t, d.nbatch = c, taps, dil, t, nbatch
    d.w1, d.bias1, d.w2, d.bias2 = _addr(w1), _addr(b1), _addr(w2), _addr(b2)
    d.slope, d.out_scale, d.accumulate = slope, scale, int(accumulate)
    d.out, d.out_bstride = _addr(out), t * c
    _lib.check(_lib.lib().ifh_resblock_pair_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_resblock_pair_bf16')
    return out


def resblock_chain(x, wstream, nunits, bias, out, *, nbatch, t, c, taps, slope=0.1, scale=1.0, accumulate=False, prof=None,
                   x_bstride=None, out_bstride=None, post_slope=1.0):
    """out = chain(x) * scale (+ out): the three dilation pairs (1, 3, 5) of one HiFi-GAN residual block in ONE launch
    (ifh_resblock_chain_bf16), bit-identical to three resblock_pair launches.  wstream/nunits/bias from w_chain_pack.
    x_bstride / out_bstride: elements between sequences (default t * c); post_slope: LeakyReLU on what is stored."""
    d = _lib.ChainDesc()
    d.post_slope = post_slope
    d.x, d.x_bstride = _addr(x), (t * c if x_bstride is None else x_bstride)
    d.c, d.taps, d.t, d.nbatch = c, taps, t, nbatch
    d.wstream, d.nunits, d.bias = _addr(wstream), nunits, _addr(bias)
    d.slope, d.out_scale, d.accumulate = slope, scale, int(accumulate)
    d.out, d.out_bstride, d.debug_prof = _addr(out), (t * c if out_bstride is None else out_bstride), _addr(prof)
    _lib.check(_lib.lib().ifh_resblock_chain_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_resblock_chain_bf16')
    return out


def resblock_seq(x, wstream, nunits, bias, out, *, nbatch, t, c, taps, slope=0.1, scale=1.0, accumulate=False, prof=None,
                 x_bstride=None, out_bstride=None, post_slope=1.0):
    """out = chain(x) * scale (+ out) like resblock_chain, over whole sequences in one LDS image overwritten in place
    (ifh_resblock_seq_bf16: (c, t) = (64, 768), (128, 192), (256, 48)); bit-identical to resblock_chain.  wstream/nunits/bias from
    w_chain_pack(convs, dev, unit_bytes=seq_unit_bytes(c))."""
    d = _lib.SeqDesc()
    d.post_slope = post_slope
    d.x, d.x_bstride = _addr(x), (t * c if x_bstride is None else x_bstride)
    d.c, d.taps, d.t, d.nbatch = c, taps, t, nbatch
    d.wstream, d.nunits, d.bias = _addr(wstream), nunits, _addr(bias)
    d.slope, d.out_scale, d.accumulate = slope, scale, int(accumulate)
    d.out, d.out_bstride = _addr(out), (t * c if out_bstride is None else out_bstride)
    d.debug_prof = _addr(prof)
    _lib.check(_lib.lib().ifh_resblock_seq_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_resblock_seq_bf16')
    return out


def seq_unit_bytes(c):
    return int(_lib.lib().ifh_resblock_seq_unit_bytes(int(c)))


def seq_supported(c, t, taps):
    return bool(_lib.lib().ifh_resblock_seq_supported(int(c), int(t), int(taps)))


def level_ws_bytes():
    """bytes of the `ws` a resblock_level(post=...) launch needs (ifh_level_ws_bytes)"""
    return int(_lib.lib().ifh_level_ws_bytes())


def resblock_level(x, blocks, out, *, nbatch, t, c, slope=0.1, scale=1.0, accumulate=False, prof=None, post=None, x_bstride=None):
    """out = sum_j chain_j(x) * scale (+ out): the residual blocks of one HiFi-GAN level in ONE launch (ifh_resblock_level_bf16,
    weights stationary in registers), bit-identical to len(blocks) resblock_chain launches with accumulate.
    blocks = [(taps, wstream, bias), ...] with wstream/bias from w_chain_pack.
    post = (w f32 [7][32], bias, slope, audio bf16 [nbatch][t], ws uint8 [level_ws_bytes()]): conv_post + tanh folded in -- the
    level's mean is not written (`out` may be None), audio is; the bits of hifigan_post on the unfolded launch's `out`."""
    d = _lib.LevelDesc()
    if post is not None:
        pw, pb, ps, audio, ws = post
        d.post_w, d.post_bias, d.post_slope, d.audio = _addr(pw), float(pb), float(ps), _addr(audio)
        d.mean_ws, d.mean_ws_bytes = _addr(ws), ws.numel() * ws.element_size()
    d.x, d.x_bstride = _addr(x), (t * c if x_bstride is None else x_bstride)
    d.c, d.t, d.nbatch, d.nblocks = c, t, nbatch, len(blocks)
    for j, (taps, ws, bias) in enumerate(blocks):
        d.taps[j], d.wstream[j], d.bias[j] = taps, _addr(ws), _addr(bias)
    d.slope, d.out_scale, d.accumulate = slope, scale, int(accumulate)
    d.out, d.out_bstride, d.debug_prof = _addr(out), t * c, _addr(prof)
    _lib.check(_lib.lib().ifh_resblock_level_bf16(ctypes.byref(d), _lib.stream_ptr(x.device)), 'ifh_resblock_level_bf16')
    return out


def conv_ring256(x, wstream, bias, out, *, nbatch, t, taps, dil=1, pre_slope=1.0, resid=None, scale=1.0, accumulate=False):
    """One 256 -> 256 channel "same" convolution on sequences of t <= 48 rows (ifh_conv_ring256_bf16: two sequences per
    workgroup, weights DMA'd as fragments from w_chain_pack([(w, b)], unit_bytes=16384)); same bits as conv()."""
    d = _lib.Ring256Desc()
    d.x, d.x_bstride = _addr(x), t * 256
    d.taps, d.dil, d.t, d.nbatch = taps, dil, t, nbatch
    d.wstream, d.bias, d.pre_slope = _addr(wstream), _addr(bias), pre_slope
    d.resid, d.resid_bstride = _addr(resid), t * 256
    d.out_scale, d.accumulate = scale, int(accumulate)
    d.out, d.out_bstride = _addr(out), t * 256
    _lib.check(_lib.lib().ifh_conv_ring256_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_conv_ring256_bf16')
    return out


def linear(x, w, bias, out, *, rows, k, n, **kw):
    """out[rows, n] = epi(x[rows, k] @ w[n, k]^T + bias)"""
    return conv(x, w, bias, out, nbatch=1, t_in=rows, t_out=rows, cin=k, n=n, **kw)


def layernorm(x, gamma, beta, out, rows, dim, resid=None, eps=1e-5):
    _lib.check(_lib.lib().ifh_layernorm_bf16(_addr(x), _addr(resid), _addr(gamma), _addr(beta), _addr(out), rows, dim,
                                             eps, _lib.stream_ptr(out.device)), 'ifh_layernorm_bf16')
    return out


def transpose_to_bf16(x, out, nbatch, rows, cols):
    _lib.check(_lib.lib().ifh_transpose_to_bf16(_addr(x), int(x.dtype == torch.float32), _addr(out), nbatch, rows, cols,
                                                _lib.stream_ptr(out.device)), 'ifh_transpose_to_bf16')
    return out


def attn_prefill(q, k, v, out, *, nbatch, nheads, tq, tk, q_off=0, k_off=0, v_off=0, q_ts, k_ts, v_ts, o_ts,
                 q_bs=None, k_bs=None, v_bs=None, o_bs=None, key_len=None, relbias=None, nrel=0):
    d = AttnDesc()
    d.q, d.k, d.v, d.out = _addr(q, q_off), _addr(k, k_off), _addr(v, v_off), _addr(out)
    d.q_ts, d.k_ts, d.v_ts, d.o_ts = q_ts, k_ts, v_ts, o_ts
    d.q_bs = tq * q_ts if q_bs is None else q_bs
    d.k_bs = tk * k_ts if k_bs is None else k_bs
    d.v_bs = tk * v_ts if v_bs is None else v_bs
    d.o_bs = tq * o_ts if o_bs is None else o_bs
    d.nbatch, d.nheads, d.head_dim, d.tq, d.tk = nbatch, nheads, 64, tq, tk
    d.key_len, d.relbias, d.nrel = _addr(key_len), _addr(relbias), nrel
    _lib.check(_lib.lib().ifh_attn_prefill_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_attn_prefill_bf16')
    return out


def attn_decode(q, k, v, out, *, nbatch, nheads, max_keys, q_bs, kv_bs, kv_ts, o_bs, k_off=0, v_off=0, key_len=None,
                dyn_len=None, dyn_add=0):
    _lib.check(_lib.lib().ifh_attn_decode_bf16(_addr(q), q_bs, _addr(k, k_off), _addr(v, v_off), kv_bs, kv_ts, _addr(out),
                                               o_bs, _addr(key_len), max_keys, nbatch, nheads, 64, _addr(dyn_len), dyn_add,
                                               _lib.stream_ptr(out.device)), 'ifh_attn_decode_bf16')
    return out


def attn_decode_shared(q, k, v, out, *, nbatch, nheads, max_keys, q_bs, kv_bs, kv_ts, o_bs, kv_group, k_off=0, v_off=0):
    """attn_decode where query row b reads cache row b // kv_group (beams of one utterance over its cross-attention K/V)"""
    _lib.check(_lib.lib().ifh_attn_decode_shared_bf16(_addr(q), q_bs, _addr(k, k_off), _addr(v, v_off), kv_bs, kv_ts,
                                                      _addr(out), o_bs, max_keys, nbatch, nheads, 64, kv_group,
                                                      _lib.stream_ptr(out.device)), 'ifh_attn_decode_shared_bf16')
    return out


def rmsnorm(x, gamma, out, rows, dim, eps=1e-6):
    _lib.check(_lib.lib().ifh_rmsnorm_bf16(_addr(x), _addr(gamma), _addr(out), rows, dim, eps, _lib.stream_ptr(out.device)),
               'ifh_rmsnorm_bf16')
    return out


def rope_append(qkv, cos_sin, cache, pos0, nvalid, *, nrows, tokens_per_row, nheads, nkv, head_dim, max_pos):
    """rotary embedding of q (in place) and k, KV append at every token's own position (ifh_rope_append_bf16)"""
    ts = 2 * nkv * head_dim
    _lib.check(_lib.lib().ifh_rope_append_bf16(_addr(qkv), (nheads + 2 * nkv) * head_dim, _addr(cos_sin), max_pos, _addr(cache),
                                               max_pos * ts, ts, _addr(pos0), _addr(nvalid), nrows, tokens_per_row, nheads, nkv,
                                               head_dim, _lib.stream_ptr(qkv.device)), 'ifh_rope_append_bf16')


def attn_gqa(qkv, cache, out, key_len, *, ntokens, tokens_per_row, nheads, nkv, head_dim, max_pos, max_keys, rope_cos_sin=None):
    """grouped-query attention of single query tokens against the KV cache (ifh_attn_gqa_bf16); q = the first
    nheads*head_dim columns of the fused projection"""
    d = _lib.GqaDesc()
    d.q, d.q_ts = _addr(qkv), (nheads + 2 * nkv) * head_dim
    d.cache, d.cache_bs, d.cache_ts, d.v_off = _addr(cache), max_pos * 2 * nkv * head_dim, 2 * nkv * head_dim, nkv * head_dim
    d.out, d.o_ts, d.key_len = _addr(out), nheads * head_dim, _addr(key_len)
    d.ntokens, d.tokens_per_row, d.nheads, d.nkv, d.head_dim = ntokens, tokens_per_row, nheads, nkv, head_dim
    d.max_keys, d.scale = max_keys, head_dim ** -0.5
    d.rope_cos_sin = _addr(rope_cos_sin)         # decode step: rotary embedding + KV append inside this launch (ifh_gqa_desc)
    _lib.check(_lib.lib().ifh_attn_gqa_bf16(ctypes.byref(d), _lib.stream_ptr(out.device)), 'ifh_attn_gqa_bf16')
    return out


def silu_mul(gate_up, out, rows, ffn, interleaved=False):
    _lib.check(_lib.lib().ifh_silu_mul_bf16(_addr(gate_up), _addr(out), rows, ffn, int(interleaved), _lib.stream_ptr(out.device)),
               'ifh_silu_mul_bf16')
    return out


def add_i32_vec(values, delta, mask=None):
    _lib.check(_lib.lib().ifh_add_i32_vec(_addr(values), _addr(mask), values.numel(), delta, _lib.stream_ptr(values.device)),
               'ifh_add_i32_vec')


def repetition_penalty(logits, history, lens, *, vocab, ld, penalty):
    _lib.check(_lib.lib().ifh_repetition_penalty_f32(_addr(logits), ld, vocab, history.size(0), _addr(history), history.size(1),
                                                     _addr(lens), penalty, _lib.stream_ptr(logits.device)),
               'ifh_repetition_penalty_f32')


def sample_topk(logits, uniform, out_tokens, scratch, *, vocab, ld, nrows, temperature=1.0, top_k=0, top_p=1.0, out_cand=None,
                out_probs=None):
    _lib.check(_lib.lib().ifh_sample_topk_f32(_addr(logits), ld, vocab, nrows, temperature, top_k, top_p, _addr(uniform),
                                              _addr(out_tokens), _addr(scratch), _addr(out_cand), _addr(out_probs),
                                              _lib.stream_ptr(logits.device)), 'ifh_sample_topk_f32')


class BeamState:
    """Device-resident state of ifh_beam_step for `nbatch` utterances x `beams` (include/infernos_hip.h: ifh_beam_desc)."""

    def __init__(self, nbatch, beams, max_new, device):
        import torch
        z = lambda *s, dt: torch.zeros(s, dtype=dt, device=device)
        self.nbatch, self.beams, self.max_new = nbatch, beams, max_new
        self.run_scores = z(nbatch, beams, dt=torch.float32)
        self.fin_scores = z(nbatch, beams, dt=torch.float32)
        self.fin_seqs = z(nbatch, beams, max_new, dt=torch.int32)
        self.fin_len = z(nbatch, beams, dt=torch.int32)
        self.is_fin = z(nbatch, beams, dt=torch.uint8)
        self.unsat = z(nbatch, dt=torch.int32)
        self.beam_src = z(nbatch * beams, dt=torch.int32)
        self.alive = z(2048, dt=torch.int32)
        self.scratch = z(nbatch * beams * 132, dt=torch.uint8)

    def reset(self):
        self.run_scores.fill_(-1.0e9)
        self.run_scores[:, 0] = 0.0
        self.fin_scores.fill_(-1.0e9)
        self.fin_seqs.zero_()
        self.fin_len.zero_()
        self.is_fin.zero_()
        self.unsat.fill_(1)
        self.alive.zero_()


def beam_step(logits, st: BeamState, toks, pos, *, vocab, ld, prompt_len, max_length, eos_id, length_penalty=1.0,
              suppress=None, begin_suppress=None):
    """One beam-search step at cur_len = pos[0] (ifh_beam_step)."""
    assert max_length - prompt_len <= st.max_new and max_length <= st.alive.numel()
    assert st.fin_seqs.size(2) == max_length - prompt_len, 'fin_seqs rows are max_length - prompt_len long'
    d = _lib.BeamDesc()
    d.logits, d.ld, d.vocab, d.nbatch, d.beams = _addr(logits), ld, vocab, st.nbatch, st.beams
    d.suppress, d.begin_suppress = _addr(suppress), _addr(begin_suppress)
    d.toks, d.pos = _addr(toks), _addr(pos)
    d.prompt_len, d.max_length, d.eos_id, d.length_penalty = prompt_len, max_length, eos_id, length_penalty
    d.run_scores, d.fin_scores, d.fin_seqs, d.fin_len = _addr(st.run_scores), _addr(st.fin_scores), _addr(st.fin_seqs), _addr(st.fin_len)
    d.is_fin, d.unsat, d.beam_src, d.alive, d.scratch = _addr(st.is_fin), _addr(st.unsat), _addr(st.beam_src), _addr(st.alive), _addr(st.scratch)
    _lib.check(_lib.lib().ifh_beam_step(ctypes.byref(d), _lib.stream_ptr(logits.device)), 'ifh_beam_step')


def kv_gather(src, dst, row_src, length, *, nrows, max_len, tok_elems, nlayers=1):
    """dst[l][row] = src[l][row_src[row]] for the first length[0] tokens of a [nlayers, nrows, max_len, tok_elems] bf16 cache"""
    _lib.check(_lib.lib().ifh_kv_gather_bf16(_addr(src), _addr(dst), _addr(row_src), _addr(length), max_len, nrows,
                                             max_len * tok_elems, tok_elems, nlayers, nrows * max_len * tok_elems,
                                             _lib.stream_ptr(dst.device)),
               'ifh_kv_gather_bf16')


def embed(ids, table, pos_table, out, *, n, dim, pos0=0, seq_len=1, ids_off=0, dyn_pos=None, dyn_ids_mul=0):
    _lib.check(_lib.lib().ifh_embed_bf16(_addr(ids, ids_off), _addr(table), _addr(pos_table), pos0, seq_len, dim, n,
                                         _addr(out), _addr(dyn_pos), dyn_ids_mul, _lib.stream_ptr(out.device)),
               'ifh_embed_bf16')
    return out


def argmax_pick(logits, *, vocab, nrows, ld=None, argmax_out=None, pick_token=0, pick_prob_out=None, out_off=0,
                dyn_pos=None, dyn_out_mul=0):
    _lib.check(_lib.lib().ifh_argmax_pick_f32(_addr(logits), vocab if ld is None else ld, vocab, nrows, pick_token,
                                              _addr(argmax_out, out_off), _addr(pick_prob_out), _addr(dyn_pos), dyn_out_mul,
                                              _lib.stream_ptr(logits.device)), 'ifh_argmax_pick_f32')


def argmax_supported(rows, n, k):
    """does a rows x n x k matrix product with f32 output fill ifh_conv_desc.argmax_keys?"""
    return bool(_lib.lib().ifh_conv_argmax_supported(rows, n, k))


def argmax_keys_finish(keys, tokens, n):
    """tokens[i] = the column in keys[i]; keys[i] = 0 (ifh_argmax_keys_finish)"""
    _lib.check(_lib.lib().ifh_argmax_keys_finish(_addr(keys), _addr(tokens), n, _lib.stream_ptr(tokens.device)), 'ifh_argmax_keys_finish')


def add_i32(value, delta, zero=None):
    """value[0] += delta; `zero` (a tensor) is cleared by the same launch"""
    _lib.check(_lib.lib().ifh_add_i32(_addr(value), delta, _addr(zero), 0 if zero is None else zero.numel() * zero.element_size(),
                                      _lib.stream_ptr(value.device)), 'ifh_add_i32')


# ---- weight preparation (host side, once per model load) -----------------------------------
def w_linear_ln(w, b, gamma, beta, device, scale=None):
    """Fold a preceding LayerNorm(gamma, beta) into a Linear: returns (bf16 W*diag(gamma), f32 W*beta + b,
    f32 c1[n] = sum_k of the ROUNDED folded weights) for ifh_conv_desc.aln_*."""
    w = w.float()
    b = torch.zeros(w.size(0)) if b is None else b.float()
    if scale is not None:
        w, b = w * scale, b * scale
    wf = (w * gamma.float()[None, :]).to(BF16)
    c2 = w @ beta.float() + b
    c1 = wf.float().sum(dim=1)
    return wf.contiguous().to(device), c2.contiguous().to(device), c1.contiguous().to(device)


def w_linear(w, device, scale=None):
    w = w.float()
    if scale is not None:
        w = w * scale
    return w.to(BF16).contiguous().to(device)


def w_bias(b, device, scale=None, n=None):
    if b is None:
        return torch.zeros(n, dtype=torch.float32, device=device)
    b = b.float()
    if scale is not None:
        b = b * scale
    return b.contiguous().to(device)


def w_conv(w, device, scale_per_out=None):
    """Conv1d weight [Cout, Cin, k] -> bf16 [Cout][k][Cin]"""
    w = w.float()
    if scale_per_out is not None:
        w = w * scale_per_out[:, None, None]
    return w.permute(0, 2, 1).to(BF16).contiguous().to(device)


def w_chain_pack(convs, device, unit_bytes=8192):
    """The six convolutions of one HiFi-GAN residual block -- [(conv1_d1, b), (conv2_d1, b), (conv1_d3, b), ...], each
    weight in Conv1d layout [Cout, Cin, k] -- as the fragment stream ifh_resblock_chain_bf16 DMAs into LDS (layout in
    include/infernos_hip.h: per k-step of 32, per 16 output channels, 64 lanes x 8 bf16 in MFMA A-operand order), zero
    padded to whole 8 KB units.  -> (bf16 stream on `device`, units, f32 bias [6][C] on `device`)"""
    parts, biases = [], []
    for w, b in convs:
        w = w.float()
        cout, cin, k = w.shape
        assert cout == cin and cout % 32 == 0
        wk = w.permute(0, 2, 1).reshape(cout, k * cin).to(BF16)              # [out][tap*C + ci]
        ks = k * cin // 32
        f = wk.reshape(cout // 16, 16, ks, 4, 8).permute(2, 0, 3, 1, 4)       # [s][i][fg][fr][e]: lane = fg*16 + fr
        parts.append(f.reshape(-1))
        biases.append(torch.zeros(cout) if b is None else b.float())
    stream = torch.cat(parts)
    unit = unit_bytes // 2
    nunits = -(-stream.numel() // unit)
    pad = nunits * unit - stream.numel()
    if pad:
        stream = torch.cat([stream, torch.zeros(pad, dtype=BF16)])
    return stream.contiguous().to(device), nunits, torch.stack(biases).contiguous().to(device)


def w_convT_fused(w, bias, device):
    """ConvTranspose1d(k=8, stride=4, padding=2) as ONE 3-tap convolution with 4*Cout output channels:
    output row q of [T][4*Cout] is rows 4q..4q+3 of the [4T][Cout] result.  Phase r reads inputs
    (q-1, q) with taps (r+6, r+2) for r in {0,1} and (q, q+1) with taps (r+2, r-2) for r in {2,3}; the
    unused third tap is zero.  Returns (bf16 [4*Cout][3][Cin], f32 bias [4*Cout]); conv pad = 1."""
    w = w.float()
    cin, cout, k = w.shape
    assert k == 8
    wf = torch.zeros(4, cout, 3, cin)
    for r in range(4):
        if r < 2:
            wf[r, :, 0], wf[r, :, 1] = w[:, :, r + 6].t(), w[:, :, r + 2].t()
        else:
            wf[r, :, 1], wf[r, :, 2] = w[:, :, r + 2].t(), w[:, :, r - 2].t()
    return (wf.reshape(4 * cout, 3, cin).to(BF16).contiguous().to(device),
            bias.float().repeat(4).contiguous().to(device))


def w_convT_phases(w, device):
    """ConvTranspose1d(k=8, stride=4, padding=2) weight [Cin, Cout, 8] -> 4 phase weights
    bf16 [Cout][2][Cin] and their left pads.  Output o = 4q + r takes input q-1 (tap k=r+6) and q
    (k=r+2) for r in {0,1}; q (k=r+2) and q+1 (k=r-2) for r in {2,3}."""
    w = w.float()
    assert w.size(2) == 8
    phases = []
    for r in range(4):
        if r < 2:
            k0, k1, pad = r + 6, r + 2, 1
        else:
            k0, k1, pad = r + 2, r - 2, 0
        wr = torch.stack([w[:, :, k0].t(), w[:, :, k1].t()], dim=1)     # [Cout][2][Cin]
        phases.append((wr.to(BF16).contiguous().to(device), pad))
    return phases
