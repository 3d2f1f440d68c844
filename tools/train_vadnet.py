#!/usr/bin/env python3
"""Fits the recurrent VAD network of csrc/vadnet.hip (conv front end + 2 x LSTM(64) + linear: the SHAPE of the detector the reference
loads at Core/VAD/SileroVAD.py:44-45) to the energy rule `ifh_vad_energy_prob` computes, on synthetic call audio, so that the
network's decisions can be USED by the throughput path (with seeded weights its probabilities mean nothing).  Silero's own weights
are not obtainable offline: this is a detector of its architecture and cost class, taught by the stand-in rule -- not Silero.

    python tools/train_vadnet.py [iterations]      -> infernos_amd/vadnet_distilled.npz (fp32, torch module layouts)

CPU only (a few minutes on 8 cores).  The teacher is p = sigmoid(0.5 * (10 log10(mean(x^2) + 1e-10) + 30)) per 768-sample window;
the student sees the same windows in call order with its LSTM state carried from window to window, as the serving loop runs it."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infernos_amd.weights import VADNET_SHAPES  # noqa: E402

WIN = 768


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(1, 32, 128, stride=64)
        self.conv2 = nn.Conv1d(32, 64, 3, stride=2, padding=1)
        self.lstm = nn.LSTM(64, 64, num_layers=2, batch_first=True)
        self.out = nn.Linear(64, 1)

    def forward(self, x, state=None):
        """x [B, W, 768] consecutive windows -> prob [B, W] (mean over a window's 6 steps of sigmoid(linear(h)))"""
        B, W, _ = x.shape
        f1 = F.relu(self.conv1(x.reshape(B * W, 1, WIN)))
        f2 = F.relu(self.conv2(f1))                                   # [B*W, 64, 6]
        seq = f2.permute(0, 2, 1).reshape(B, W * 6, 64)
        y, state = self.lstm(seq, state)
        p = torch.sigmoid(self.out(y))[..., 0].reshape(B, W, 6).mean(2)
        return p, state


def mulaw_roundtrip(x):
    """G.711 quantisation as the serving path sees it (closed form, 14-bit magnitude; close enough for a training input)"""
    mu = 255.0
    y = np.sign(x) * np.log1p(mu * np.abs(np.clip(x, -1, 1))) / np.log1p(mu)
    q = np.round((y + 1) / 2 * 255) / 255 * 2 - 1
    return (np.sign(q) * (np.power(1 + mu, np.abs(q)) - 1) / mu).astype(np.float32)


def make_batch(rng, B, W):
    """B call excerpts of W windows: gated harmonic tones (infernos_amd/synth.py's family, random pitch, gate rate, level, onset),
    noise floors of random level, silences and steady tones"""
    n = W * WIN
    t = np.arange(n) / 8000.0
    xs = np.zeros((B, n), dtype=np.float32)
    for b in range(B):
        kind = rng.integers(0, 10)
        noise = 10 ** rng.uniform(-4.0, -1.3) * rng.standard_normal(n)
        if kind == 0:
            x = noise
        else:
            f0 = rng.uniform(80, 400)
            rate = rng.uniform(1.0, 8.0)
            env = 0.5 - 0.5 * np.cos(2 * np.pi * rate * t + rng.uniform(0, 2 * np.pi))
            if kind == 1:
                env[:] = 1.0
            on, off = sorted(rng.uniform(0, n / 8000.0, 2))
            if kind >= 4:
                env[(t < on) | (t > off)] = 0.0
            lvl = 10 ** rng.uniform(-2.5, -0.2)
            x = lvl * env * sum(a * np.sin(2 * np.pi * k * f0 * t) for k, a in ((1, 1.0), (2, 0.5), (3, 0.25))) + noise
        xs[b] = mulaw_roundtrip(x)
    xw = torch.from_numpy(xs).reshape(B, W, WIN)
    e = (xw.double() ** 2).mean(2)
    logit = 0.5 * (10.0 * torch.log10(e + 1e-10) + 30.0)
    return xw, torch.sigmoid(logit).float()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    net = Net()
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, iters, eta_min=1e-4)
    for it in range(iters):
        x, p = make_batch(rng, 48, 24)
        q, _ = net(x)
        loss = F.binary_cross_entropy(q.clamp(1e-6, 1 - 1e-6), p) - F.binary_cross_entropy(p.clamp(1e-6, 1 - 1e-6), p)
        opt.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        sched.step()
        if it % 100 == 0 or it == iters - 1:
            with torch.no_grad():
                agree = ((q > 0.5) == (p > 0.5)).float().mean()
            print('iter %4d  excess BCE %.5f  max |dp| %.3f  decision agreement %.4f' % (it, float(loss), float((q - p).abs().max()), float(agree)), flush=True)
    sd = {k: v.detach().float().numpy() for k, v in net.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == VADNET_SHAPES, 'module layout differs from weights.VADNET_SHAPES'
    out = os.path.join(ROOT, 'infernos_amd', 'vadnet_distilled.npz')
    np.savez(out, **sd)
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
