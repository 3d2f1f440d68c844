"""Qwen2 decoder-only LLM engine (InfernLLMWorker's model: Cluster/InfernLLMWorker.py:60-119 loads
Qwen/Qwen2.5-*-Instruct through transformers and calls generate with a streamer).

Layer maths: transformers/models/qwen2/modeling_qwen2.py (RMSNorm, rotary embedding with rotate_half pairing,
grouped-query attention, SiLU-gated MLP, tied or separate lm_head).  Everything runs through the C ABI: projections are
ifh_conv_bf16 GEMMs (q|k|v and gate|up fused into one weight each), the rest are the llm.hip kernels.

Batches are ragged by construction: every row keeps its own length (device int32), prompts are prefilled in one pass
over the padded [B, T] token block with per-token key counts (causal) and padding skipped, and each decode step
advances all rows by one token at their own positions -- the per-token launch sequence is identical for every step, so it
is captured into a hipGraph after the first eager step and replayed.
"""

import torch

from .. import _lib, ops

BF16 = torch.bfloat16


class Sampler:
    """The draw transformers' generate makes when the checkpoint's generation config says do_sample (Qwen2.5-Instruct:
    repetition_penalty 1.05, temperature 0.7, top_k 20, top_p 0.8): ifh_repetition_penalty_f32 over the row's prompt and
    generated tokens, then ifh_sample_topk_f32 with one uniform number per row from a seeded device generator."""

    MAX_CANDIDATES = 32          # ifh_sample_topk_f32 keeps the 32 best logits of a row

    def __init__(self, temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.05, seed=0):
        # top_k = 0 is transformers' "top-k disabled": the draw is then top-p over the 32 best candidates, which equals
        # the unrestricted nucleus whenever top_p < 1 leaves at most 32 tokens in it; a wider request cannot be served
        if not 0 <= top_k <= self.MAX_CANDIDATES:
            raise ValueError('Sampler: top_k=%r is outside 0..%d (ifh_sample_topk_f32 keeps the %d best candidates of a row); '
                             'lower the checkpoint\'s generation_config.top_k' % (top_k, self.MAX_CANDIDATES, self.MAX_CANDIDATES))
        if top_k == 0 and top_p >= 1.0:
            raise ValueError('Sampler: top_k=0 with top_p=1 asks for a draw over the whole vocabulary; the device sampler '
                             'keeps %d candidates per row -- set top_k or top_p' % self.MAX_CANDIDATES)
        self.temperature, self.top_k, self.top_p, self.repetition_penalty = temperature, top_k, top_p, repetition_penalty
        self.seed, self.gen = seed, None

    def pick(self, model, st, B):
        dev = model.device
        if self.gen is None:
            self.gen = torch.Generator(device=dev).manual_seed(self.seed)
        u = torch.rand(B, generator=self.gen, device=dev)
        if self.repetition_penalty != 1.0:
            ops.repetition_penalty(st['logits_full'], st['hist'], st['lens'][0], vocab=model.vocab, ld=model.vpad,
                                   penalty=self.repetition_penalty)
        ops.sample_topk(st['logits_full'], u, st['toks'], st['sscratch'], vocab=model.vocab, ld=model.vpad, nrows=B,
                        temperature=self.temperature, top_k=self.top_k, top_p=self.top_p)


class Qwen2:
    def __init__(self, sd, cfg, device, max_tokens=2048):
        self.device = dev = _lib.require_device(device)
        self.cfg = cfg
        self.d, self.hd, self.nh, self.nkv, self.ff = cfg['hidden'], cfg['head_dim'], cfg['heads'], cfg['kv_heads'], cfg['ffn']
        self.vocab = cfg['vocab']
        self.vpad = -(-self.vocab // 16) * 16
        self.max_tokens = min(max_tokens, cfg['max_pos'])
        self.eps = cfg['rms_eps']
        self.nq = (self.nh + 2 * self.nkv) * self.hd
        f32 = lambda t: t.float().contiguous().to(dev)
        self.tok = sd['model.embed_tokens.weight'].to(BF16).contiguous().to(dev)
        self.head = sd['lm_head.weight'].to(BF16).contiguous().to(dev) if 'lm_head.weight' in sd and not cfg['tie'] else self.tok
        self.norm = f32(sd['model.norm.weight'])
        # RMSNorm weights are folded into the projections that consume the normalised rows (W diag(gamma)), so that the
        # normalisation left to do at run time is the per-row scale rsqrt(mean(x^2) + eps): an explicit RMSNorm with unit
        # gamma in the prefill, nothing at all in the decode step (the scale is applied in the consumer's epilogue from row
        # statistics the producer of x accumulated: ifh_conv_desc.aln_stats / stats_out with ln_rms).  gate and up rows are
        # interleaved (gate_j, up_j) for the fused SiLU-gate epilogue (IFH_ACT_SILU_GLU).
        self.ones = torch.ones(self.d, dtype=torch.float32, device=dev)
        self.layers = []
        for i in range(cfg['layers']):
            L = 'model.layers.%d.' % i
            A = L + 'self_attn.'
            g1, g2 = sd[L + 'input_layernorm.weight'].float(), sd[L + 'post_attention_layernorm.weight'].float()
            wqkv = torch.cat([sd[A + 'q_proj.weight'], sd[A + 'k_proj.weight'], sd[A + 'v_proj.weight']]).float() * g1.to(sd[A + 'q_proj.weight'].device)
            wg = sd[L + 'mlp.gate_proj.weight'].float() * g2.to(sd[L + 'mlp.gate_proj.weight'].device)
            wu = sd[L + 'mlp.up_proj.weight'].float() * g2.to(sd[L + 'mlp.up_proj.weight'].device)
            wgu = torch.stack([wg, wu], 1).reshape(2 * self.ff, self.d)
            self.layers.append(dict(
                wqkv=ops.w_linear(wqkv, dev),
                bqkv=ops.w_bias(torch.cat([sd[A + 'q_proj.bias'], sd[A + 'k_proj.bias'], sd[A + 'v_proj.bias']]), dev),
                wo=ops.w_linear(sd[A + 'o_proj.weight'], dev),
                wgu=ops.w_linear(wgu, dev),
                wd=ops.w_linear(sd[L + 'mlp.down_proj.weight'], dev)))
            del wqkv, wg, wu, wgu
        # cos/sin of position * theta^(-2j/hd), f32 [max_tokens][hd/2][2] (Qwen2RotaryEmbedding, computed in fp32)
        inv = 1.0 / (cfg['rope_theta'] ** (torch.arange(0, self.hd, 2, dtype=torch.int64).float() / self.hd))
        fr = torch.arange(self.max_tokens).float()[:, None] * inv[None, :]
        self.cos_sin = torch.stack([fr.cos(), fr.sin()], -1).contiguous().to(dev)
        self._bufs = {}
        self.fuse = True       # False: explicit RMSNorm / SiLU launches (tests/test_llm_gpu.py compares the two)
        self.glu_prefill = True                                      # prefill's gate|up with the SiLU-gate epilogue, until the library declines it
        self.bucket_batches = True

    # ---- buffers ------------------------------------------------------------------------------
    def _state(self, B):
        if B not in self._bufs:
            dev, d = self.device, self.d
            e = lambda *s, dt=BF16: torch.empty(s, dtype=dt, device=dev)
            while len(self._bufs) >= 3:
                self._bufs.pop(next(iter(self._bufs)))
            self._bufs[B] = dict(
                kv=[torch.zeros((B, self.max_tokens, 2 * self.nkv * self.hd), dtype=BF16, device=dev) for _ in self.layers],
                x=e(B, d), h=e(B, d), qkv=e(B, self.nq), att=e(B, self.nh * self.hd), gu=e(B, 2 * self.ff), ff=e(B, self.ff),
                logits_full=e(B, self.vpad, dt=torch.float32),
                lens=torch.zeros((2, B), dtype=torch.int32, device=dev),        # [0]: tokens in the cache, [1]: that + 1
                ones=torch.ones(B, dtype=torch.int32, device=dev),
                hist=torch.zeros((B, self.max_tokens), dtype=torch.int32, device=dev),     # prompt + chosen tokens per row
                sscratch=torch.zeros(B * 260, dtype=torch.uint8, device=dev),
                stats=torch.zeros((2 * len(self.layers), max(64, -(-B // 16) * 16), 2), dtype=torch.int64, device=dev),
                tick=torch.zeros(1, dtype=torch.int32, device=dev),
                # f32 workspace of the down projection's split-K form (ifh_conv_desc.splitk_ws): part of THIS decode state, so that the
                # captured step graphs of different states / engines never share one
                skws=torch.empty(24 * max(B, 16) * d, dtype=torch.float32, device=dev),
                amax=torch.zeros(B, dtype=torch.int64, device=dev),        # arg-max keys of the head (zero between launches)
                toks=torch.zeros(B, dtype=torch.int32, device=dev), graph=None, eager_steps=0)
            b = self._bufs[B]
            b['logits'] = b['logits_full'][:, :self.vocab]
        return self._bufs[B]

    # ---- one layer stack over `rows` tokens -----------------------------------------------------
    def _layers(self, x, h, qkv, att, gu, ff, kv, rows, nrows, T, pos0, nvalid, key_len, max_keys):
        """the layer stack with explicit (unit-gamma) RMSNorm launches: prefill, and decode batches outside 17..64 rows"""
        d = self.d
        pws, big = None, rows >= 4096              # a prompt pass: the GEMMs take every CU (ifh_conv_desc.whole_chip) and a split workspace
        if rows >= 4096:
            # 64 MB, prompt passes only; one per STREAM (include/infernos_hip.h: a split workspace belongs to one stream / graph):
            # two prompt passes of this model on different streams must not share the partial tiles
            key = torch.cuda.current_stream(self.device).cuda_stream
            if not hasattr(self, '_pws'):
                self._pws = {}
            if key not in self._pws:
                self._pws[key] = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=self.device)
            pws = self._pws[key]
        for L, cache in zip(self.layers, kv):
            ops.rmsnorm(x, self.ones, h, rows, d, self.eps)
            ops.linear(h, L['wqkv'], L['bqkv'], qkv, rows=rows, k=d, n=self.nq, whole_chip=big)
            ops.rope_append(qkv, self.cos_sin, cache, pos0, nvalid, nrows=nrows, tokens_per_row=T, nheads=self.nh, nkv=self.nkv,
                            head_dim=self.hd, max_pos=self.max_tokens)
            ops.attn_gqa(qkv, cache, att, key_len, ntokens=rows, tokens_per_row=T, nheads=self.nh, nkv=self.nkv,
                         head_dim=self.hd, max_pos=self.max_tokens, max_keys=max_keys)
            # (pws: a workspace for the last, partial round of tiles of the 256 x 256 GEMM -- 288 tiles of o / down on 256 CUs at 12 288 rows)
            ops.linear(att, L['wo'], None, x, rows=rows, k=self.nh * self.hd, n=d, resid=x, splitk_ws=pws, whole_chip=big)
            ops.rmsnorm(x, self.ones, h, rows, d, self.eps)
            fused = self.glu_prefill and rows >= 4096 and rows % 128 == 0 and (2 * self.ff) % 256 == 0
            if fused:
                # prefill: SiLU(gate) * up in the epilogue of the DMA-ring GEMM -- the [rows, 2 ffn] product is never written.  The library
                # may decline the shape (its own row threshold IFH_GEMM_BIG_ROWS, view alignment, the 4 GiB DMA-offset limit): it says so
                # before launching anything, and the two-launch form below takes over for good.
                try:
                    ops.linear(h, L['wgu'], None, ff, rows=rows, k=d, n=2 * self.ff, ldc=self.ff, act=ops.ACT_SILU_GLU, splitk_ws=pws, whole_chip=big)
                except _lib.InfernosHipError as e:
                    if e.code != _lib.IFH_EINVAL:        # a HIP error is an error, not a declined shape
                        raise
                    self.glu_prefill = fused = False
            if not fused:
                ops.linear(h, L['wgu'], None, gu, rows=rows, k=d, n=2 * self.ff)
                ops.silu_mul(gu, ff, rows, self.ff, interleaved=True)
            ops.linear(ff, L['wd'], None, x, rows=rows, k=self.ff, n=d, resid=x, splitk_ws=pws, whole_chip=big)

    def _layers_fused(self, st, B):
        """decode step at 17..64 rows: six launches per layer.  o-proj and down-proj leave the (sum, sum of squares) of the
        residual-stream rows they store; q|k|v and gate|up apply rsqrt(mean(x^2) + eps) in their epilogues; gate|up also
        applies SiLU(gate) * up.  Layer 0 reads the embedding rows, which nobody produced statistics for."""
        d = self.d
        x, h, qkv, att, ff, stats = st['x'], st['h'], st['qkv'], st['att'], st['ff'], st['stats']
        SO = stats.size(1) * 2
        pos0, ones, key_len = st['lens'][0], st['ones'], st['lens'][1]
        for li, (L, cache) in enumerate(zip(self.layers, st['kv'])):
            s1, s2 = (2 * li) * SO, (2 * li + 1) * SO
            if li == 0:
                ops.rmsnorm(x, self.ones, h, B, d, self.eps)
                ops.linear(h, L['wqkv'], L['bqkv'], qkv, rows=B, k=d, n=self.nq)
            else:
                ops.linear(x, L['wqkv'], L['bqkv'], qkv, rows=B, k=d, n=self.nq, aln=(stats, s1, None), ln_dim=d, ln_eps=self.eps,
                           ln_rms=True)
            if self.hd == 128:
                # rotary embedding + KV append inside the attention launch (ifh_gqa_desc.rope_cos_sin: the same bits, one launch less)
                ops.attn_gqa(qkv, cache, att, key_len, ntokens=B, tokens_per_row=1, nheads=self.nh, nkv=self.nkv, head_dim=self.hd,
                             max_pos=self.max_tokens, max_keys=self.max_tokens, rope_cos_sin=self.cos_sin)
            else:
                ops.rope_append(qkv, self.cos_sin, cache, pos0, ones, nrows=B, tokens_per_row=1, nheads=self.nh, nkv=self.nkv,
                                head_dim=self.hd, max_pos=self.max_tokens)
                ops.attn_gqa(qkv, cache, att, key_len, ntokens=B, tokens_per_row=1, nheads=self.nh, nkv=self.nkv, head_dim=self.hd,
                             max_pos=self.max_tokens, max_keys=self.max_tokens)
            ops.linear(att, L['wo'], None, x, rows=B, k=self.nh * self.hd, n=d, resid=x, stats_out=stats, stats_off=s2, ln_dim=d,
                       ln_eps=self.eps, ln_rms=True)
            ops.linear(x, L['wgu'], None, ff, rows=B, k=d, n=2 * self.ff, ldc=self.ff, act=ops.ACT_SILU_GLU, aln=(stats, s2, None),
                       ln_dim=d, ln_eps=self.eps, ln_rms=True)
            if li + 1 < len(self.layers):
                ops.linear(ff, L['wd'], None, x, rows=B, k=self.ff, n=d, resid=x, stats_out=stats, stats_off=s1 + 2 * SO, ln_dim=d,
                           ln_eps=self.eps, ln_rms=True, splitk_ws=st['skws'])
            else:
                ops.linear(ff, L['wd'], None, x, rows=B, k=self.ff, n=d, resid=x, splitk_ws=st['skws'])

    def _head(self, st, x, B, argmax):
        ops.rmsnorm(x, self.norm, st['h'], B, self.d, self.eps)
        if argmax and ops.argmax_supported(B, self.vocab, self.d):
            # the greedy pick inside the head's epilogue (per-row keys by atomic max; ifh_conv_desc.argmax_keys) + a 64-thread launch
            # that turns the keys into token ids and re-arms them: no second pass over the [B, vocab] logits
            ops.linear(st['h'], self.head, None, st['logits_full'], rows=B, k=self.d, n=self.vocab, ldc=self.vpad, argmax_keys=st['amax'])
            ops.argmax_keys_finish(st['amax'], st['toks'], B)
            return
        ops.linear(st['h'], self.head, None, st['logits_full'], rows=B, k=self.d, n=self.vocab, ldc=self.vpad)
        if argmax:
            ops.argmax_pick(st['logits_full'], vocab=self.vocab, nrows=B, ld=self.vpad, argmax_out=st['toks'])

    # ---- prefill ------------------------------------------------------------------------------
    def prefill(self, prompts, all_logits=False, argmax=True):
        """prompts: list of B token-id lists (1 <= len <= max_tokens - 1).  Fills the KV caches and leaves the logits of
        every row's last prompt token in state['logits'] (and their argmax in state['toks'] if `argmax`).  all_logits:
        also return f32 [B, T, vocab] logits of every prompt position (padding rows undefined) -- a test hook."""
        dev, d = self.device, self.d
        B = len(prompts)
        lens = [len(p) for p in prompts]
        assert min(lens) >= 1 and max(lens) < self.max_tokens
        T = max(lens)
        st = self._state(B)
        ids = torch.zeros((B, T), dtype=torch.int32)
        for i, p in enumerate(prompts):
            ids[i, :len(p)] = torch.as_tensor(p, dtype=torch.int32)
        ids = ids.to(dev)
        lens_t = torch.tensor(lens, dtype=torch.int32, device=dev)
        pos0 = torch.zeros(B, dtype=torch.int32, device=dev)
        t_idx = torch.arange(T, dtype=torch.int32, device=dev)[None, :].expand(B, T)
        key_len = torch.where(t_idx < lens_t[:, None], t_idx + 1, torch.ones_like(t_idx)).contiguous()
        rows = B * T
        e = lambda *s: torch.empty(s, dtype=BF16, device=dev)
        x, h, qkv, att, gu, ff = e(rows, d), e(rows, d), e(rows, self.nq), e(rows, self.nh * self.hd), e(rows, 2 * self.ff), e(rows, self.ff)
        ops.embed(ids, self.tok, None, x, n=rows, dim=d)
        self._layers(x, h, qkv, att, gu, ff, st['kv'], rows, B, T, pos0, lens_t, key_len, T)
        last = (torch.arange(B, device=dev) * T + (lens_t.long() - 1))
        st['x'].copy_(x.index_select(0, last))
        st['lens'][0].copy_(lens_t)
        st['lens'][1].copy_(lens_t + 1)
        st['hist'][:, :T].copy_(ids)
        self._head(st, st['x'], B, argmax)
        out = None
        if all_logits:
            out = torch.empty((rows, self.vocab), dtype=torch.float32, device=dev)
            ops.rmsnorm(x, self.norm, h, rows, d, self.eps)
            ops.linear(h, self.head, None, out, rows=rows, k=d, n=self.vocab)
            out = out.view(B, T, self.vocab)
        return st, out

    # ---- decode -------------------------------------------------------------------------------
    def _step_launches(self, st, B, argmax):
        ops.embed(st['toks'], self.tok, None, st['x'], n=B, dim=self.d)
        fused = self.fuse and 16 < B <= 64 and 2 * self.ff >= 8192 and self.d % 16 == 0
        if fused:
            self._layers_fused(st, B)
        else:
            self._layers(st['x'], st['h'], st['qkv'], st['att'], st['gu'], st['ff'], st['kv'], B, B, 1, st['lens'][0], st['ones'],
                         st['lens'][1], self.max_tokens)
        self._head(st, st['x'], B, argmax)
        ops.add_i32_vec(st['lens'], 1)
        if fused:
            ops.add_i32(st['tick'], 1, zero=st['stats'])          # the row statistics of this step are spent

    def step(self, st, B, argmax=True, use_graphs=True):
        """Feed state['toks'] (one token per row) at every row's own position; logits of the next token land in
        state['logits'], their argmax in state['toks'] when `argmax` (greedy chaining without a host round trip)."""
        if not use_graphs or st['eager_steps'] < 1:
            st['eager_steps'] += 1
            return self._step_launches(st, B, argmax)
        key = 'graph%d' % int(argmax)
        if st.get(key) is None:
            st[key] = _lib.CountedGraph(lambda: self._step_launches(st, B, argmax))
        st[key].replay()

    def generate(self, prompts, max_new_tokens, eos_ids=(), pad_id=0, on_tokens=None, sampler=None, use_graphs=True,
                 keep_logits=False):
        """Continue every prompt for up to max_new_tokens tokens; a row stops at one of eos_ids (kept in its output) and is
        fed pad_id afterwards, as transformers' generate does; the loop ends when every row has stopped or a row's
        cache is full.  on_tokens(int64 [B] CPU tensor) is called once per step with the tokens just chosen (pad_id for
        stopped rows): the streamer hook of InfernLLMWorker.py:113-118.  sampler (a Sampler) replaces the greedy pick.
        Returns (list of per-row generated id lists, per-step logits list
        if keep_logits)."""
        nreal = len(prompts)
        # batch sizes are bucketed (1, 2, 4, ... rows; a bucket = one set of KV caches, work buffers and captured step graphs):
        # a worker that is handed 1..max_batch_size requests per batch would otherwise allocate and capture per size.
        # Padding rows hold one pad token, count as stopped from the start and never reach the caller.
        B = 1
        while B < nreal:
            B *= 2
        if not self.bucket_batches or keep_logits:
            B = nreal
        prompts = list(prompts) + [[pad_id]] * (B - nreal)
        st, _ = self.prefill(prompts, argmax=sampler is None)
        done = [i >= nreal for i in range(B)]
        out = [[] for _ in range(B)]
        eos = set(int(e) for e in eos_ids)
        budget = min(max_new_tokens, self.max_tokens - max(len(p) for p in prompts))
        kept = []
        # The token loop runs one step ahead of the host: the chosen tokens chain on the device (argmax / sampler -> next
        # embedding), so step s + 1 is launched before the host has seen the tokens of step s; those arrive through a small
        # pinned ring and are handed to `on_tokens` one step late.  A row that has stopped keeps decoding its own continuation
        # on the device (transformers feeds it the pad token instead): rows are independent and its outputs are dropped, so
        # only the host-side view (pad_id for stopped rows) matters.  The loop ends one step after the host sees the last stop.
        ring = st.get('tok_ring')
        if ring is None:
            ring = st['tok_ring'] = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(2)]
            st['tok_ev'] = [torch.cuda.Event() for _ in range(2)]
        evs = st['tok_ev']

        def deliver(s):
            evs[s & 1].synchronize()
            toks = ring[s & 1].tolist()
            step_toks = []
            for i, t in enumerate(toks):
                if done[i]:
                    step_toks.append(pad_id)
                    continue
                out[i].append(t)
                step_toks.append(t)
                if t in eos:
                    done[i] = True
            if on_tokens is not None:
                on_tokens(torch.tensor(step_toks[:nreal], dtype=torch.long))

        for s in range(budget):
            if keep_logits:
                kept.append(st['logits'].clone())
            if sampler is not None:
                sampler.pick(self, st, B)
            st['hist'].scatter_(1, st['lens'][0].long()[:, None], st['toks'][:, None])      # the chosen token's position
            ring[s & 1].copy_(st['toks'], non_blocking=True)
            evs[s & 1].record()
            if s + 1 < budget:
                self.step(st, B, argmax=sampler is None, use_graphs=use_graphs)          # step s + 1, ahead of the host
            if s >= 1:
                deliver(s - 1)
                if all(done):
                    break                      # (the step already in flight is never looked at)
        else:
            if budget > 0 and not all(done):
                deliver(budget - 1)
        out = out[:nreal]
        return out, kept
