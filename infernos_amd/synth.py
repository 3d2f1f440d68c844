"""Synthetic call audio used by benchmarks and tests: a gated harmonic
tone plus noise, 1 s of leading/trailing silence, mu-law framed into 20 ms packets."""
import numpy as np


def synth_utterance(seed: int, seconds: float = 10.0, sr: int = 8000) -> np.ndarray:
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    f0 = rng.uniform(100, 300)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * t)
    env[(t < 1.0) | (t > seconds - 1.0)] = 0.0
    x = 0.3 * env * sum(a * np.sin(2 * np.pi * k * f0 * t) for k, a in ((1, 1.0), (2, 0.5), (3, 0.25)))
    x = x + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)


class CharChatTokenizer:
    """Stand-in for the Qwen chat tokenizer (its vocabulary files are not reachable offline): one token per character
    over a small alphabet, ChatML-style template, left padding.  Used by the tests, the fixture generator
    (tools/gen_golden_nn.py) and the LLM bench leg; ids beyond the alphabet (what a random-weight model emits) decode
    through a modulo so that any id stream yields text with sentence boundaries."""
    ALPHABET = 'etaoin shrdlu.?!\n,cmfwyp'
    pad_token_id, eos_token_id = 0, 1
    IM_START, IM_END = 2, 3
    padding_side = 'left'

    def __init__(self, vocab=777):
        self.vocab = vocab
        self.c2i = {c: 4 + i for i, c in enumerate(self.ALPHABET)}

    def apply_chat_template(self, context, tokenize=False, add_generation_prompt=True):
        assert not tokenize
        s = ''.join('<|im_start|>%s\n%s<|im_end|>\n' % (m.get('role', 'user'), m.get('content', '')) for m in context)
        return s + ('<|im_start|>assistant\n' if add_generation_prompt else '')

    def encode(self, text):
        ids, i = [], 0
        while i < len(text):
            if text.startswith('<|im_start|>', i):
                ids.append(self.IM_START)
                i += 12
            elif text.startswith('<|im_end|>', i):
                ids.append(self.IM_END)
                i += 10
            else:
                ids.append(self.c2i.get(text[i].lower(), 4 + (ord(text[i]) % len(self.ALPHABET))))
                i += 1
        return ids

    def __call__(self, messages, return_tensors='pt', padding=True):
        import torch
        rows = [self.encode(m) for m in messages]
        T = max(len(r) for r in rows)
        ids = torch.full((len(rows), T), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), T), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, T - len(r):] = torch.tensor(r)
            mask[i, T - len(r):] = 1

        class Enc(dict):
            def to(self, device):
                return Enc({k: v.to(device) for k, v in self.items()})
            __getattr__ = dict.__getitem__
        return Enc(input_ids=ids, attention_mask=mask)

    def decode_ids(self, ids, skip_special_tokens=True):
        out = []
        for t in ids:
            t = int(t)
            if t < 4:
                if not skip_special_tokens:
                    out.append(('<pad>', '<eos>', '<|im_start|>', '<|im_end|>')[t])
                continue
            out.append(self.ALPHABET[(t - 4) % len(self.ALPHABET)])
        return ''.join(out)

    def batch_decode(self, token_ids, skip_special_tokens=True):
        return [self.decode_ids(r, skip_special_tokens) for r in (token_ids.tolist() if hasattr(token_ids, 'tolist') else token_ids)]
