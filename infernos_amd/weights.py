"""Model weights for the speech engines: HF-format state dicts in, device tensors out.

The engines consume state dicts with the key names/shapes of the checkpoints the reference
loads (HelloSippyRTPipe.py:164-178: microsoft/speecht5_tts, microsoft/speecht5_hifigan,
sobomax/speecht5-rt.post_vocoder.v2; InfernSTTWorker.py:25: openai/whisper-*).  No
checkpoint is reachable offline, so benchmarks and tests use `synth_state_dict`: seeded
random tensors with exactly those names and shapes (schema captured from the reference's
model classes into tests/golden/nn_schema.json and mirrored in SCHEMA_FILE).
"""
import json
import math
import os
import zlib

import torch

SCHEMA_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nn_schema.json')

FAMILIES = ('speecht5_tts', 'hifigan', 'amendment', 'whisper_tiny', 'whisper_base')

# Qwen2 decoder-only configurations (InfernLLMWorker, Cluster/InfernLLMWorker.py:60: Qwen/Qwen2.5-*-Instruct).  Key names
# and shapes are a pure function of the configuration (qwen2_schema); 'qwen2_tiny' is the parity-test size, whose schema
# is also captured from transformers' Qwen2ForCausalLM into nn_schema.json and compared in the tests; 'qwen2_1p5b' is
# Qwen2.5-1.5B-Instruct's config.json (BASELINE configuration 5).
QWEN2_CONFIGS = {
    'qwen2_tiny': dict(vocab=1000, hidden=512, ffn=1024, layers=2, heads=4, kv_heads=2, head_dim=128, rope_theta=1.0e6,
                       rms_eps=1e-6, tie=True, max_pos=512),
    'qwen2_tiny64': dict(vocab=777, hidden=256, ffn=640, layers=3, heads=4, kv_heads=1, head_dim=64, rope_theta=1.0e4,
                         rms_eps=1e-5, tie=False, max_pos=512),
    # wide MLP: the smallest shape whose decode step takes the fused path (row statistics instead of RMSNorm launches,
    # SiLU-gate epilogue on the weight-streaming GEMM: gate|up >= 8192 columns); checked against the oracle directly
    'qwen2_wide': dict(vocab=1000, hidden=512, ffn=4096, layers=3, heads=4, kv_heads=2, head_dim=128, rope_theta=1.0e6,
                       rms_eps=1e-6, tie=True, max_pos=512),
    # the reference's default checkpoint (Cluster/InfernLLMWorker.py:69: Qwen/Qwen2.5-14B-Instruct), for shape / capacity probes
    'qwen2_14b': dict(vocab=152064, hidden=5120, ffn=13824, layers=48, heads=40, kv_heads=8, head_dim=128, rope_theta=1.0e6,
                      rms_eps=1e-6, tie=False, max_pos=32768),
    'qwen2_1p5b': dict(vocab=151936, hidden=1536, ffn=8960, layers=28, heads=12, kv_heads=2, head_dim=128, rope_theta=1.0e6,
                       rms_eps=1e-6, tie=True, max_pos=32768),
    # Qwen2.5-1.5B's layer and vocabulary dimensions with 2 of its 28 layers: every kernel selection of BASELINE configuration 5's
    # LLM (k_gemm_m64<4,2> on gate|up 17 920 x 1 536 and on the 151 936-row tied head) at a size the fp32 oracle finishes in seconds
    'qwen2_1p5b_2l': dict(vocab=151936, hidden=1536, ffn=8960, layers=2, heads=12, kv_heads=2, head_dim=128, rope_theta=1.0e6,
                          rms_eps=1e-6, tie=True, max_pos=32768),
}


def qwen2_schema(cfg):
    """state-dict names -> (shape, dtype) of transformers' Qwen2ForCausalLM for a QWEN2_CONFIGS entry"""
    d, hd, nh, nkv, ffn = cfg['hidden'], cfg['head_dim'], cfg['heads'], cfg['kv_heads'], cfg['ffn']
    sc = {'model.embed_tokens.weight': ([cfg['vocab'], d], 'float32'), 'model.norm.weight': ([d], 'float32')}
    if not cfg['tie']:
        sc['lm_head.weight'] = ([cfg['vocab'], d], 'float32')
    for i in range(cfg['layers']):
        L = 'model.layers.%d.' % i
        sc[L + 'input_layernorm.weight'] = ([d], 'float32')
        sc[L + 'post_attention_layernorm.weight'] = ([d], 'float32')
        for n, rows in (('q_proj', nh * hd), ('k_proj', nkv * hd), ('v_proj', nkv * hd)):
            sc[L + 'self_attn.%s.weight' % n] = ([rows, d], 'float32')
            sc[L + 'self_attn.%s.bias' % n] = ([rows], 'float32')
        sc[L + 'self_attn.o_proj.weight'] = ([d, nh * hd], 'float32')
        sc[L + 'mlp.gate_proj.weight'] = ([ffn, d], 'float32')
        sc[L + 'mlp.up_proj.weight'] = ([ffn, d], 'float32')
        sc[L + 'mlp.down_proj.weight'] = ([d, ffn], 'float32')
    return sc


# Whisper shapes beyond the two families of the schema file, derived from whisper_base's entries (same tensor names; d, ffn, mel bins,
# vocabulary and layer counts substituted).  whisper_large_v3_2l = the layer shapes of openai/whisper-large-v3 -- the default model of
# Cluster/InfernSTTWorker.py:25 -- with two encoder and two decoder layers instead of 32 + 32, for parity tests at its widths.
# whisper_tiny_en = openai/whisper-tiny.en, the model BASELINE configuration 1 names (English-only vocabulary of 51 864).
WHISPER_CONFIGS = {'whisper_large_v3_2l': dict(d=1280, ffn=5120, n_mels=128, vocab=51866, enc_layers=2, dec_layers=2),
                   'whisper_tiny_en': dict(d=384, ffn=1536, n_mels=80, vocab=51864, enc_layers=4, dec_layers=4)}


def whisper_schema(cfg):
    base = load_schema()['whisper_base']
    sub = {512: cfg['d'], 2048: cfg['ffn'], 80: cfg['n_mels'], 51865: cfg['vocab']}
    sc = {}
    for name, (shape, dtype) in base.items():
        shape = [sub.get(v, v) for v in shape]
        if '.layers.' in name:
            pre, rest = name.split('.layers.')
            idx, tail = rest.split('.', 1)
            if idx != '0':
                continue
            n = cfg['enc_layers'] if pre.endswith('encoder') else cfg['dec_layers']
            for li in range(n):
                sc['%s.layers.%d.%s' % (pre, li, tail)] = (shape, dtype)
        else:
            sc[name] = (shape, dtype)
    return sc


def load_schema():
    with open(SCHEMA_FILE) as f:
        return json.load(f)


def _std_for(name, shape):
    """Initialisation scale keeping activations O(1) through the deep stacks."""
    if len(shape) < 2:
        return None
    if name.endswith('pe_k.weight'):
        return 0.1
    if 'embed_tokens' in name or 'embed_positions' in name:
        return 0.5 if 'embed_tokens' in name else 0.1
    if name.startswith('upsampler.') or '.upsampler.' in name:      # ConvTranspose1d [in, out, k], stride 4
        fan = shape[0] * shape[2] / 4.0
        return 1.0 / math.sqrt(fan)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    gain = 1.0
    if 'resblocks' in name or name.startswith('resblock.'):
        gain = 0.6
    if 'post_conv' in name:
        gain = 2.0
    return gain / math.sqrt(fan_in)


def synth_tensor(name, shape, dtype, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    shape = tuple(shape)
    if dtype == 'int64':
        return torch.zeros(shape, dtype=torch.int64)
    leaf = name.rsplit('.', 1)[-1]
    norm_like = ('layer_norm' in name or 'batch_norm' in name or 'layernorm' in name or name == 'model.norm.weight')
    if leaf == 'alpha':
        return torch.ones(shape)
    if leaf == 'running_var':
        return 1.0 + 0.2 * torch.rand(shape, generator=g)
    if leaf == 'running_mean':
        return 0.1 * torch.randn(shape, generator=g)
    if name == 'mean':
        return 0.1 * torch.randn(shape, generator=g)
    if name == 'scale':
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if norm_like and leaf == 'weight':
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == 'bias' or len(shape) < 2:
        return 0.05 * torch.randn(shape, generator=g)
    return _std_for(name, shape) * torch.randn(shape, generator=g)


def synth_state_dict(family: str, seed: int = 0, stop_bias=None):
    """Seeded random HF-format state dict (CPU, float32) for one model family.
    stop_bias: value for speech_decoder_postnet.prob_out.bias (e.g. -20 disables the stop
    head)."""
    schema = (qwen2_schema(QWEN2_CONFIGS[family]) if family in QWEN2_CONFIGS else
              whisper_schema(WHISPER_CONFIGS[family]) if family in WHISPER_CONFIGS else load_schema()[family])
    sd = {}
    for name in sorted(schema):
        shape, dtype = schema[name]
        sd[name] = synth_tensor(name, shape, dtype, seed)
    if family.startswith('whisper'):
        sd['proj_out.weight'] = sd['model.decoder.embed_tokens.weight']      # tied
        sd['model.encoder.embed_positions.weight'] = whisper_sinusoids(*schema['model.encoder.embed_positions.weight'][0])
    if stop_bias is not None and 'speech_decoder_postnet.prob_out.bias' in sd:
        sd['speech_decoder_postnet.prob_out.bias'] = torch.full_like(sd['speech_decoder_postnet.prob_out.bias'], stop_bias)
    return sd


def whisper_sinusoids(length, channels, max_timescale=10000.0):
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    st = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([st.sin(), st.cos()], dim=1)


def scaled_positional_table(max_len, dim):
    """SpeechT5ScaledPositionalEncoding.pe (non-persistent buffer, so not in checkpoints)."""
    pe = torch.zeros(max_len, dim)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.int64).float() * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def qwen2_random_on_device(cfg, device, seed=0):
    """Random bf16/f32 tensors of a Qwen2 configuration's shapes generated ON the device (1.5 B parameters take minutes
    through the seeded CPU generator): for throughput/latency measurements only -- parity tests use synth_state_dict."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, (shape, _) in qwen2_schema(cfg).items():
        if name.endswith('norm.weight') or 'layernorm' in name:
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif len(shape) == 1:
            sd[name] = 0.05 * torch.randn(shape, generator=g, device=device)
        elif 'embed_tokens' in name:
            sd[name] = 0.5 * torch.randn(shape, generator=g, device=device, dtype=torch.bfloat16)
        else:
            sd[name] = torch.randn(shape, generator=g, device=device, dtype=torch.bfloat16) / shape[1] ** 0.5
    return sd


# ---- the recurrent VAD network of csrc/vadnet.hip (Silero-v3.1-SHAPED stand-in: the reference's model file is not obtainable) ----
VADNET_SHAPES = {
    'conv1.weight': (32, 1, 128), 'conv1.bias': (32,), 'conv2.weight': (64, 32, 3), 'conv2.bias': (64,),
    'lstm.weight_ih_l0': (256, 64), 'lstm.weight_hh_l0': (256, 64), 'lstm.bias_ih_l0': (256,), 'lstm.bias_hh_l0': (256,),
    'lstm.weight_ih_l1': (256, 64), 'lstm.weight_hh_l1': (256, 64), 'lstm.bias_ih_l1': (256,), 'lstm.bias_hh_l1': (256,),
    'out.weight': (1, 64), 'out.bias': (1,),
}


def synth_vadnet(seed: int = 0):
    """Seeded fp32 weights in torch's own module layouts (Conv1d / LSTM / Linear, uniform +-1/sqrt(fan_in) as torch initialises)."""
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for name, shape in VADNET_SHAPES.items():
        fan = {'conv1': 128, 'conv2': 96, 'lstm': 64, 'out': 64}[name.split('.')[0]]
        sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / (fan ** 0.5)
    return sd


def load_vadnet_distilled():
    """The weights `tools/train_vadnet.py` fitted to the energy rule of ifh_vad_energy_prob on synthetic call audio (torch module
    layouts, fp32; infernos_amd/vadnet_distilled.npz): a detector of the reference's architecture whose decisions can be used.  Not
    Silero's weights (Core/VAD/SileroVAD.py:44 -- the model file is not obtainable offline): PARITY UNPINNED against it."""
    import os
    import numpy as np
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vadnet_distilled.npz'))
    sd = {k: torch.from_numpy(np.ascontiguousarray(z[k])).float() for k in VADNET_SHAPES}
    assert all(tuple(sd[k].shape) == v for k, v in VADNET_SHAPES.items())
    return sd


def pack_vadnet(sd):
    """The weight blob ifh_vadnet_prob reads: every matrix transposed to [k][out] (consecutive lanes read consecutive outputs), the two
    LSTM biases of a layer added up."""
    import torch
    parts = [sd['conv1.weight'][:, 0, :].t().contiguous().flatten(), sd['conv1.bias'],
             sd['conv2.weight'].permute(2, 1, 0).contiguous().flatten(), sd['conv2.bias']]
    for l in (0, 1):
        parts += [sd['lstm.weight_ih_l%d' % l].t().contiguous().flatten(), sd['lstm.weight_hh_l%d' % l].t().contiguous().flatten(),
                  sd['lstm.bias_ih_l%d' % l] + sd['lstm.bias_hh_l%d' % l]]
    parts += [sd['out.weight'].flatten(), sd['out.bias']]
    return torch.cat([p.float().flatten() for p in parts]).contiguous()
