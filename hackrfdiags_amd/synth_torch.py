"""Synthetic IQ batches generated ON the device with torch: bench.py's workload and the
full-size parity tests' distinct-per-channel inputs (hackrfdiags_amd/synth.py holds the numpy generators of the
small test vectors).  Plumbing: torch only makes the bytes; nothing here is on the measured path."""
import numpy as np
import torch

BLOCK = 262144


def make_fm_batch(channels, blocks, device, first_channel=0):
    """FM test signal of SURVEY.md 8(d), generated on the GPU: carrier at -64 kHz,
    +-30 kHz deviation by a (300 + 100*(c mod 32)) Hz tone, amplitude 100, uniform
    noise in [-3,4].  (Same signal family as hackrfdiags_amd/synth.py; the noise
    comes from torch's generator here, so the bytes differ from the test vectors.)
    Thirty-two channels per pass (round 4; one channel per pass before): a 1024-channel batch is ~400 torch dispatches
    instead of ~12 000 -- a counter pass of rocprofv3 segfaulted inside this function at the larger number."""
    n = blocks * (BLOCK // 2)
    out = torch.empty((channels, blocks, BLOCK), dtype=torch.int8, device=device)
    k = torch.arange(n, dtype=torch.float64, device=device)
    gen = torch.Generator(device=device)
    G = 32
    for g0 in range(0, channels, G):
        g = min(G, channels - g0)
        ch = first_channel + g0 + torch.arange(g, device=device)
        f_c = (300.0 + 100.0 * (ch % 32)).to(torch.float64)[:, None]
        beta = 30000.0 / f_c
        phi = (2.0 * np.pi * (-64000.0) / 2048000.0) * k[None, :] - beta * (torch.cos((2.0 * np.pi / 2048000.0) * f_c * k[None, :]) - 1.0)
        gen.manual_seed(12345 + first_channel + g0)
        noise = torch.randint(-3, 5, (g, 2, n), device=device, generator=gen, dtype=torch.int32)
        i = torch.round(100.0 * torch.cos(phi)).to(torch.int32) + noise[:, 0]
        q = torch.round(100.0 * torch.sin(phi)).to(torch.int32) + noise[:, 1]
        del phi, noise
        out[g0:g0 + g] = torch.stack([i, q], dim=2).to(torch.int8).reshape(g, blocks, BLOCK)   # [g, n, 2] interleaved
        del i, q
    return out


def make_random_batch(channels, blocks, device, first_channel=0):
    gen = torch.Generator(device=device)
    gen.manual_seed(1 + first_channel)
    return torch.randint(-128, 128, (channels, blocks, BLOCK), dtype=torch.int8, device=device, generator=gen)
