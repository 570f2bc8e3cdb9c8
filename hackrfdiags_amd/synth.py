"""Deterministic synthetic IQ / PCM generators.

Used by bench.py (workload), tests (parity inputs) and tests/golden/make_golden.py.
Everything is seeded; the LCG family is pure integer arithmetic so the same
bytes come out on every machine.

  lcg_bytes      uniform random int8 IQ (worst case for LUT locality)
  fm_tone_iq     FM test signal at 2.048 MS/s: carrier -64 kHz, +-30 kHz deviation
  dc_iq / impulse_iq / zeros_iq   quirk cases (full-scale DC wrap)
  lcg_pcm        int16 PCM for the transmit path
"""
from __future__ import annotations

import numpy as np

LCG_A = np.uint32(1664525)
LCG_C = np.uint32(1013904223)
BLOCK_BYTES = 262144          # one libhackrf transfer (hackRf/hackrf.c:101)
BLOCK_IQ = BLOCK_BYTES // 2   # 131072 complex samples = 64 ms at 2.048 MS/s
PCM_PER_BLOCK = 512           # 8 kS/s * 64 ms


def lcg_states(seed: int, n: int) -> np.ndarray:
    """s[k+1] = s[k]*1664525 + 1013904223 (mod 2^32); returns s[1..n] as uint32.

    Vectorised by doubling: s[k+L] = A_L*s[k] + C_L.
    """
    out = np.empty(max(n, 1), dtype=np.uint32)
    with np.errstate(over="ignore"):
        s0 = np.uint32(seed & 0xFFFFFFFF)
        out[0] = s0 * LCG_A + LCG_C
        a_l, c_l = LCG_A, LCG_C          # advance-by-L multiplier / increment
        filled = 1
        while filled < n:
            take = min(filled, n - filled)
            out[filled:filled + take] = out[:take] * a_l + c_l
            # (A_L, C_L) -> (A_2L, C_2L)
            c_l = a_l * c_l + c_l
            a_l = a_l * a_l
            filled += take
    return out[:n]


def lcg_bytes(seed: int, n: int) -> np.ndarray:
    """n int8 values, byte = state >> 24."""
    return (lcg_states(seed, n) >> np.uint32(24)).astype(np.uint8).view(np.int8)


def lcg_pcm(seed: int, n: int) -> np.ndarray:
    """n int16 values, sample = state >> 16."""
    return (lcg_states(seed, n) >> np.uint32(16)).astype(np.uint16).view(np.int16)


def fm_tone_iq(channel: int, n_iq: int, start_iq: int = 0) -> np.ndarray:
    """Interleaved int8 IQ of an FM test signal (2*n_iq bytes).

    I = round(100 cos phi) + n1, Q = round(100 sin phi) + n2,
    phi[k+1] = phi[k] + 2*pi*(-64000 + 30000 sin(2*pi*f_c*t_k))/2048000,
    f_c = 300 + 100*(channel mod 32) Hz, noise in [-3,4] from an LCG seeded
    12345+channel (top three bits of successive states).  `start_iq` lets a
    long stream be produced in pieces with identical bytes.
    """
    fs = 2048000.0
    f_c = 300.0 + 100.0 * (channel % 32)
    k = np.arange(start_iq, start_iq + n_iq, dtype=np.float64)
    # closed-form phase (no cumulative rounding drift between pieces)
    beta = 30000.0 / f_c
    phi = 2.0 * np.pi * (-64000.0) * k / fs - beta * (np.cos(2.0 * np.pi * f_c * k / fs) - 1.0)
    states = lcg_states(12345 + channel, 2 * (start_iq + n_iq))[2 * start_iq:]
    noise = (states >> np.uint32(29)).astype(np.int32) - 3
    i = np.rint(100.0 * np.cos(phi)).astype(np.int32) + noise[0::2]
    q = np.rint(100.0 * np.sin(phi)).astype(np.int32) + noise[1::2]
    out = np.empty(2 * n_iq, dtype=np.int8)
    out[0::2] = i.astype(np.int8)
    out[1::2] = q.astype(np.int8)
    return out


def am_tone_iq(channel: int, n_iq: int, start_iq: int = 0) -> np.ndarray:
    """AM test signal: carrier at -64 kHz (so the Fs/4 mix brings it to DC),
    80 % modulation by a (300+100*(c mod 32)) Hz tone, amplitude 60."""
    fs = 2048000.0
    f_c = 300.0 + 100.0 * (channel % 32)
    k = np.arange(start_iq, start_iq + n_iq, dtype=np.float64)
    env = 60.0 * (1.0 + 0.8 * np.sin(2.0 * np.pi * f_c * k / fs))
    phi = 2.0 * np.pi * (-64000.0) * k / fs
    out = np.empty(2 * n_iq, dtype=np.int8)
    out[0::2] = np.rint(env * np.cos(phi)).astype(np.int32).astype(np.int8)
    out[1::2] = np.rint(env * np.sin(phi)).astype(np.int32).astype(np.int8)
    return out


def dc_iq(n_iq: int, i_val: int = 127, q_val: int = 127) -> np.ndarray:
    out = np.empty(2 * n_iq, dtype=np.int8)
    out[0::2] = np.int8(i_val)
    out[1::2] = np.int8(q_val)
    return out


def impulse_iq(n_iq: int, at: int = 0, i_val: int = 127, q_val: int = -128) -> np.ndarray:
    out = np.zeros(2 * n_iq, dtype=np.int8)
    out[2 * at] = np.int8(i_val)
    out[2 * at + 1] = np.int8(q_val)
    return out


def zeros_iq(n_iq: int) -> np.ndarray:
    return np.zeros(2 * n_iq, dtype=np.int8)


def make_input(kind: str, seed: int, n_blocks: int) -> np.ndarray:
    """One channel's stream of n_blocks x 262144 int8, by kind name."""
    n_iq = n_blocks * BLOCK_IQ
    if kind == "lcg":
        return lcg_bytes(seed, 2 * n_iq)
    if kind == "fmtone":
        return fm_tone_iq(seed, n_iq)
    if kind == "amtone":
        return am_tone_iq(seed, n_iq)
    if kind == "dc_pos":
        return dc_iq(n_iq, 127, 127)
    if kind == "dc_neg":
        return dc_iq(n_iq, -128, -128)
    if kind == "impulse":
        return impulse_iq(n_iq, at=seed % 64)
    if kind == "zeros":
        return zeros_iq(n_iq)
    raise ValueError(f"unknown input kind {kind!r}")


def digest(data: np.ndarray) -> str:
    """sha256 of the raw bytes (long-run golden hashes)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(data).view(np.uint8).tobytes()).hexdigest()
