"""HiFi-GAN vocoder + AmendmentNetwork1 on the HIP device.

Replaces `self.vocoder(spectrogram)` / `self.chunker(spectrogram, audio)` at
HelloSippyTTSRT/HelloSippyRTPipe.py:236-237 (transformers SpeechT5HifiGan,
modeling_speecht5.py:2954-3064; AmendmentNetwork1, HelloSippyRT.py:181-237).  Every
convolution is an implicit GEMM on the matrix cores with LeakyReLU fused on the operand load and
bias/residual/3-way-mean fused in the epilogue: the residual pairs (conv k,d -> conv k,1 -> + x) are one
ifh_resblock_pair_bf16 launch each with the intermediate held in LDS, the rest one ifh_conv_bf16 launch
each; the transposed convolutions run as 4 two-tap phases.
"""
import torch

from .. import _lib, ops
from ..ops import ACT_LRELU, ACT_NONE, BF16


class HifiGan:
    def __init__(self, sd, device):
        self.device = dev = _lib.require_device(device)
        self.mean = sd['mean'].float().to(dev)
        self.scale = sd['scale'].float().to(dev)
        self.pre_w, self.pre_b = ops.w_conv(sd['conv_pre.weight'], dev), ops.w_bias(sd['conv_pre.bias'], dev)
        self.up = [(ops.w_convT_phases(sd['upsampler.%d.weight' % i], dev), ops.w_bias(sd['upsampler.%d.bias' % i], dev))
                   for i in range(4)]
        self.upf = [ops.w_convT_fused(sd['upsampler.%d.weight' % i], sd['upsampler.%d.bias' % i], dev) for i in range(4)]
        self.fused_up = True             # (attributes: the unfused forms are what the parity tests compare against)
        self.res = []
        for i in range(4):
            lvl = []
            for j, k in enumerate((3, 7, 11)):
                R = 'resblocks.%d.' % (i * 3 + j)
                lvl.append([(ops.w_conv(sd[R + 'convs1.%d.weight' % d], dev), ops.w_bias(sd[R + 'convs1.%d.bias' % d], dev),
                             ops.w_conv(sd[R + 'convs2.%d.weight' % d], dev), ops.w_bias(sd[R + 'convs2.%d.bias' % d], dev))
                            for d in range(3)])
            self.res.append(lvl)
        # whole residual blocks as one launch each (csrc/chain.hip) where the channel count allows: the six convolutions'
        # weights as one pre-packed fragment stream per block
        self.chain = {}
        self.fused_chain = True
        # the C = 32 level: its three residual blocks as ONE launch with the weights stationary in registers (csrc/level.hip)
        self.fused_level = True
        self.fused_post = True           # conv_post + tanh inside that launch: the level's [n, t, 32] mean never crosses HBM
        # upsamplers 1-3 as PLAIN matrix products (12-frame chunks on the default kernels): a zero guard row between sequences makes the
        # 3-tap window of channels-last rows one contiguous K = 3 Cin vector at row stride Cin, and the LeakyReLU in front of the
        # upsampler is taken by the producing level's last launch (post_slope) -- the same bits, 377 -> 251 us per 1 280 chunks
        self.plain_up = True
        for i in range(1, 4):
            for j, k in enumerate((3, 7, 11)):
                R = 'resblocks.%d.' % (i * 3 + j)
                convs = []
                for d in range(3):
                    convs.append((sd[R + 'convs1.%d.weight' % d], sd[R + 'convs1.%d.bias' % d]))
                    convs.append((sd[R + 'convs2.%d.weight' % d], sd[R + 'convs2.%d.bias' % d]))
                self.chain[(i, j)] = ops.w_chain_pack(convs, dev)
        # whole-sequence blocks (csrc/seq.hip: one LDS image overwritten in place, nothing recomputed) where the level's shape is one
        # of (c, t) = (256, 48), (128, 192), (64, 768); seq_levels = the channel counts that take it (64 and
        # 256: at C = 128 the chain kernel's one-chunk tiles measure faster, 943 against 1 095 us per level at 1 280 chunks)
        self.seq = {}
        self.seq_levels = (64, 256)
        for i, c in ((0, 256), (1, 128), (2, 64)):
            for j, k in enumerate((3, 7, 11)):
                if ops.seq_unit_bytes(c) == 8192:
                    self.seq[(i, j)] = self.chain[(i, j)]                       # the same fragment stream
                else:
                    R = 'resblocks.%d.' % (i * 3 + j)
                    convs = []
                    for d in range(3):
                        convs.append((sd[R + 'convs1.%d.weight' % d], sd[R + 'convs1.%d.bias' % d]))
                        convs.append((sd[R + 'convs2.%d.weight' % d], sd[R + 'convs2.%d.bias' % d]))
                    self.seq[(i, j)] = ops.w_chain_pack(convs, dev, unit_bytes=ops.seq_unit_bytes(c))
        # the C = 256 level: every convolution as its own fragment stream for ifh_conv_ring256_bf16 (two chunks per workgroup)
        self.ring = {}
        self.fused_ring = True
        for j in range(3):
            R = 'resblocks.%d.' % j
            for d in range(3):
                for which in (1, 2):
                    ws, _, b = ops.w_chain_pack([(sd[R + 'convs%d.%d.weight' % (which, d)], sd[R + 'convs%d.%d.bias' % (which, d)])],
                                                dev, unit_bytes=16384)
                    self.ring[(j, d, which)] = (ws, b.reshape(-1))
        self.post_w = sd['conv_post.weight'].float()[0].t().contiguous().to(dev)     # [7][32]
        self.post_b = float(sd['conv_post.bias'].float()[0])
        self._bufs = {}
        self.fused_pairs = True

    def _buffers(self, n, t0, cache=None):
        """Activation buffers of one batch shape.  `cache` is a dict owned by the caller (a TTS batch state, so that
        the buffers live exactly as long as the hipGraphs captured over them); without it one shape stays resident here."""
        key = (n, t0)
        if cache is not None:
            if key not in cache:
                cache[key] = self._alloc(n, t0)
            return cache[key]
        if key not in self._bufs:
            self._bufs = {key: self._alloc(n, t0)}          # keep one shape resident
        return self._bufs[key]

    def _alloc(self, n, t0):
        dev = self.device
        b = {'x0': torch.empty((n, t0, 512), dtype=BF16, device=dev)}
        t, c = t0, 512
        for i in range(4):
            t, c = t * 4, c // 2
            for nm in ('u', 'h', 'r0', 'r1', 'xn'):
                b['%s%d' % (nm, i)] = torch.empty((n, t, c), dtype=BF16, device=dev)
        b['audio'] = torch.empty((n, t), dtype=BF16, device=dev)
        # the C = 32 level's running mean, one tile per workgroup (ifh_level_desc.mean_ws: scratch of this buffer set, so of one stream / graph)
        b['lvl_ws'] = torch.empty(ops.level_ws_bytes(), dtype=torch.uint8, device=dev)
        return b

    def _alloc_plain(self, n):
        """buffers of the guard-row layout (t0 = 12): xg<i> = level i's mean with a zero row in front of and behind every sequence (+ 2
        rows: the matrix product reads three rows from every row); ug<i> = upsampler i's output, four rows per row of xg<i-1>"""
        dev = self.device
        b = {'x0': torch.empty((n, 12, 512), dtype=BF16, device=dev), 'u0': torch.empty((n, 48, 256), dtype=BF16, device=dev),
             'audio': torch.empty((n, 3072), dtype=BF16, device=dev),
             'lvl_ws': torch.empty(ops.level_ws_bytes(), dtype=torch.uint8, device=dev)}
        t, c = 48, 256
        for i in range(3):
            rows = self._plain_rows(n, t)
            b['xg%d' % i] = torch.zeros((rows + 2, c), dtype=BF16, device=dev)                 # guard rows stay zero: nothing writes them
            b['ug%d' % (i + 1)] = torch.empty((rows + 2, 2 * c), dtype=BF16, device=dev)
            t, c = t * 4, c // 2
        return b

    @staticmethod
    def _plain_rows(n, t):
        """rows of an upsampler's matrix product: n (t + 2), and never so few that ifh_conv_bf16 would take its decode-step kernels
        (<= 256 rows: other accumulation chains) -- a chunk's audio must not depend on how many chunks share its launch"""
        return max(n * (t + 2), 264)

    def _call_plain(self, voc_in, cache):
        n = voc_in.size(0)
        key = (n, 12, 'plain')
        store = self._bufs if cache is None else cache
        if key not in store:
            if cache is None:
                store.clear()
            store[key] = self._alloc_plain(n)
        B = store[key]
        ops.conv(voc_in, self.pre_w, self.pre_b, B['x0'], nbatch=n, t_in=12, t_out=12, cin=80, n=512, taps=7, pad=3)
        wf, bf = self.upf[0]
        ops.conv(B['x0'], wf, bf, B['u0'], nbatch=n, t_in=12, t_out=12, cin=512, n=1024, taps=3, pad=1, pre_slope=0.1, convt_cout=256)
        x, xbs, t, c = B['u0'], 48 * 256, 48, 256
        for i in range(3):
            out, obs = B['xg%d' % i][1:], (t + 2) * c                  # sequence b's rows start at row b (t + 2) + 1
            for j, k in enumerate((3, 7, 11)):
                kw = dict(nbatch=n, t=t, c=c, taps=k, slope=0.1, scale=1.0 / 3.0, accumulate=(j > 0), x_bstride=xbs, out_bstride=obs,
                          post_slope=(0.1 if j == 2 else 1.0))
                if c in self.seq_levels and ops.seq_supported(c, t, k):
                    ws, nunits, bias = self.seq[(i, j)]
                    ops.resblock_seq(x, ws, nunits, bias, out, **kw)
                else:
                    ws, nunits, bias = self.chain[(i, j)]
                    ops.resblock_chain(x, ws, nunits, bias, out, **kw)
            # upsampler i + 1: rows m = 0 .. n (t + 2) - 1 of xg<i> x [3 c] -> row m + 1 of ug<i+1> (4 output rows of c / 2 channels)
            wf, bf = self.upf[i + 1]
            ug = B['ug%d' % (i + 1)]
            ops.linear(B['xg%d' % i], wf, bf, ug[1:], rows=self._plain_rows(n, t), k=3 * c, n=2 * c, lda=c)
            x, xbs = ug[1:], (t + 2) * 2 * c
            t, c = t * 4, c // 2
        blocks = [(k, self.chain[(3, j)][0], self.chain[(3, j)][2]) for j, k in enumerate((3, 7, 11))]
        ops.resblock_level(x, blocks, None, nbatch=n, t=t, c=c, slope=0.1, scale=1.0 / 3.0, x_bstride=xbs,
                           post=(self.post_w, self.post_b, 0.01, B['audio'], B['lvl_ws']))
        return B['audio']

    def level(self, i, u, B, n, t, c, post=False):
        """The three residual blocks (k = 3, 7, 11) of upsampling level i over u bf16 [n, t, c] -> their mean (B['xn%d' % i]);
        post (the last level, on the level kernel): -> conv_post + tanh of that mean, B['audio'], with the mean left on the chip."""
        h, xn = B['h%d' % i], B['xn%d' % i]
        rbuf = (B['r0%d' % i], B['r1%d' % i])
        if self.fused_level and self.fused_chain and c == 32:
            blocks = [(k, self.chain[(i, j)][0], self.chain[(i, j)][2]) for j, k in enumerate((3, 7, 11))]
            if post:
                ops.resblock_level(u, blocks, None, nbatch=n, t=t, c=c, slope=0.1, scale=1.0 / 3.0,
                                   post=(self.post_w, self.post_b, 0.01, B['audio'], B['lvl_ws']))
                return B['audio']
            ops.resblock_level(u, blocks, xn, nbatch=n, t=t, c=c, slope=0.1, scale=1.0 / 3.0)
            return xn
        for j, k in enumerate((3, 7, 11)):
            cur = u
            if c in self.seq_levels and ops.seq_supported(c, t, k):
                ws, nunits, bias = self.seq[(i, j)]
                ops.resblock_seq(u, ws, nunits, bias, xn, nbatch=n, t=t, c=c, taps=k, slope=0.1, scale=1.0 / 3.0, accumulate=(j > 0))
                continue
            if self.fused_chain and (i, j) in self.chain and (c < 128 or t <= 192):
                ws, nunits, bias = self.chain[(i, j)]
                ops.resblock_chain(u, ws, nunits, bias, xn, nbatch=n, t=t, c=c, taps=k, slope=0.1, scale=1.0 / 3.0,
                                   accumulate=(j > 0))
                continue
            for di, d in enumerate((1, 3, 5)):
                w1, b1, w2, b2 = self.res[i][j][di]
                if self.fused_ring and c == 256 and t <= 48 and i == 0:
                    last = di == 2
                    nxt = xn if last else rbuf[di]
                    (ws1, rb1), (ws2, rb2) = self.ring[(j, di, 1)], self.ring[(j, di, 2)]
                    ops.conv_ring256(cur, ws1, rb1, h, nbatch=n, t=t, taps=k, dil=d, pre_slope=0.1)
                    ops.conv_ring256(h, ws2, rb2, nxt, nbatch=n, t=t, taps=k, dil=1, pre_slope=0.1, resid=cur,
                                     scale=(1.0 / 3.0 if last else 1.0), accumulate=(last and j > 0))
                    cur = nxt
                    continue
                # both convolutions in one launch, intermediate kept in LDS (same bits) -- except at C = 256 with
                # >= 512 batch entries, where two launches measure 1.2-1.35x faster (tools/probe_resblock.py 768)
                if self.fused_pairs and not (c == 256 and n >= 512):
                    last = di == 2
                    nxt = xn if last else rbuf[di]
                    ops.resblock_pair(cur, w1, b1, w2, b2, nxt, nbatch=n, t=t, c=c, taps=k, dil=d, slope=0.1,
                                      scale=(1.0 / 3.0 if last else 1.0), accumulate=(last and j > 0))
                    cur = nxt
                    continue
                ops.conv(cur, w1, b1, h, nbatch=n, t_in=t, t_out=t, cin=c, n=c, taps=k, dil=d, pad=(k * d - d) // 2,
                         pre_slope=0.1)
                if di < 2:
                    nxt = rbuf[di]
                    ops.conv(h, w2, b2, nxt, nbatch=n, t_in=t, t_out=t, cin=c, n=c, taps=k, pad=(k - 1) // 2,
                             pre_slope=0.1, resid=cur)
                    cur = nxt
                else:       # last dilation: fold the /3 mean over the three resblocks into the epilogue
                    ops.conv(h, w2, b2, xn, nbatch=n, t_in=t, t_out=t, cin=c, n=c, taps=k, pad=(k - 1) // 2,
                             pre_slope=0.1, resid=cur, scale=1.0 / 3.0, accumulate=(j > 0))
        return xn

    def __call__(self, voc_in: torch.Tensor, cache=None) -> torch.Tensor:
        """voc_in bf16 [N, T, 80], already (x-mean)/scale normalised -> bf16 [N, 256*T]"""
        n, t0, _ = voc_in.shape
        if (self.plain_up and t0 == 12 and self.fused_up and self.fused_chain and self.fused_level and self.fused_post
                and 64 in self.seq_levels and 256 in self.seq_levels):
            return self._call_plain(voc_in, cache)
        B = self._buffers(n, t0, cache)
        ops.conv(voc_in, self.pre_w, self.pre_b, B['x0'], nbatch=n, t_in=t0, t_out=t0, cin=80, n=512, taps=7, pad=3)
        prev, t, c = B['x0'], t0, 512
        for i in range(4):
            u = B['u%d' % i]
            if self.fused_up:          # the 4 output phases as one 3-tap conv with 4*Cout channels: x read once, rows written whole
                wf, bf = self.upf[i]
                ops.conv(prev, wf, bf, u, nbatch=n, t_in=t, t_out=t, cin=c, n=2 * c, taps=3, pad=1, pre_slope=0.1, convt_cout=c // 2)
            else:
                phases, ub = self.up[i]
                for r, (w, pad) in enumerate(phases):
                    ops.conv(prev, w, ub, u, nbatch=n, t_in=t, t_out=t, cin=c, n=c // 2, taps=2, pad=pad, pre_slope=0.1,
                             ostride=4, ooff=r)
            t, c = t * 4, c // 2
            post = i == 3 and c == 32 and self.fused_post and self.fused_level and self.fused_chain
            prev = self.level(i, u, B, n, t, c, post=post)
        audio = B['audio']
        if post:
            return audio
        _lib.check(_lib.lib().ifh_hifigan_post_bf16(ops._addr(prev), ops._addr(self.post_w), self.post_b, ops._addr(audio),
                                                    n, t, 0.01, _lib.stream_ptr(self.device)), 'ifh_hifigan_post_bf16')
        return audio


class Amendment:
    """AmendmentNetwork1 (HelloSippyRT.py:181-237) for 12-frame chunks."""

    def __init__(self, sd, device):
        self.device = dev = _lib.require_device(device)
        g = lambda k: sd[k]
        self.pm_w, self.pm_b = ops.w_conv(g('conv_pre_m.weight'), dev), ops.w_bias(g('conv_pre_m.bias'), dev)
        self.pa_w, self.pa_b = ops.w_conv(g('conv_pre_a.weight'), dev), ops.w_bias(g('conv_pre_a.bias'), dev)
        self.upf = [ops.w_convT_fused(g('upsampler.%d.weight' % i), g('upsampler.%d.bias' % i), dev) for i in range(2)]
        self.r1_w, self.r1_b = ops.w_conv(g('resblock.conv1.weight'), dev), ops.w_bias(g('resblock.conv1.bias'), dev)
        self.r2_w, self.r2_b = ops.w_conv(g('resblock.conv2.weight'), dev), ops.w_bias(g('resblock.conv2.bias'), dev)
        self.po_w, self.po_b = ops.w_conv(g('post_conv.weight'), dev), ops.w_bias(g('post_conv.bias'), dev)
        self._bufs = {}

    def _buffers(self, n, cache=None):
        store = self._bufs if cache is None else cache
        if n not in store:
            dev = self.device
            e = lambda *s: torch.empty(s, dtype=BF16, device=dev)
            if cache is None:
                store.clear()                          # keep one shape resident
            store[n] = dict(a_cl=e(n, 12, 256), cat=e(n, 12, 192), u0=e(n, 48, 128), u1=e(n, 192, 64),
                            h=e(n, 192, 64), r=e(n, 192, 64), post=e(n, 8, 256))
        return store[n]

    def __call__(self, amd_mel: torch.Tensor, audio: torch.Tensor, out: torch.Tensor, nbatch: int, cache=None):
        """amd_mel bf16 [4B,12,80] (ifh_tts_chunks_bf16 output), audio bf16 [4B,3072] -> out bf16 [B,8192]"""
        n = audio.size(0)
        assert n == 4 * nbatch and audio.size(1) == 3072
        B = self._buffers(n, cache)
        ops.transpose_to_bf16(audio, B['a_cl'], n, 256, 12)                      # audio.view(N,256,12) -> [N,12,256]
        ops.conv(amd_mel, self.pm_w, self.pm_b, B['cat'], nbatch=n, t_in=12, t_out=12, cin=80, n=32, taps=3, pad=1, ldc=192)
        ops.conv(B['a_cl'], self.pa_w, self.pa_b, B['cat'], nbatch=n, t_in=12, t_out=12, cin=256, n=160, taps=3, pad=1,
                 ldc=192, out_off=32)
        prev, t, c = B['cat'], 12, 192
        for i, co in enumerate((128, 64)):
            wf, bf = self.upf[i]
            u = B['u%d' % i]
            ops.conv(prev, wf, bf, u, nbatch=n, t_in=t, t_out=t, cin=c, n=4 * co, taps=3, pad=1, pre_slope=0.01,
                     convt_cout=(co if c % 32 == 0 else 0))
            prev, t, c = u, t * 4, co
        ops.conv(prev, self.r1_w, self.r1_b, B['h'], nbatch=n, t_in=192, t_out=192, cin=64, n=64, taps=3, pad=1, pre_slope=0.01)
        ops.conv(B['h'], self.r2_w, self.r2_b, B['r'], nbatch=n, t_in=192, t_out=192, cin=64, n=64, taps=3, dil=3, pad=3,
                 pre_slope=0.01, resid=prev)
        ops.conv(B['r'], self.po_w, self.po_b, B['post'], nbatch=n, t_in=192, t_out=8, cin=64, n=256, taps=8, stride=24,
                 pad=0, pre_slope=0.01, act=ACT_LRELU, act_slope=0.01)
        _lib.check(_lib.lib().ifh_amend_final_bf16(ops._addr(B['post']), ops._addr(audio), ops._addr(out), nbatch,
                                                   _lib.stream_ptr(self.device)), 'ifh_amend_final_bf16')
        return out
