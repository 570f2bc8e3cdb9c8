"""What the full-size parity tests share (round 5).

Until round 4 every test at BASELINE's sizes fed its channels inputs with period 8 -- the number of XCDs -- and checked
"channel c equals channel c mod 8": an addressing or state-slot error that maps channel c onto c +- 8k (the shape an
XCD-interleaved workgroup -> channel map or a per-XCD stride bug would have) was invisible.  Now:

  * every channel gets an input OF ITS OWN, made on the device (`distinct_batch`: the bench's own generators,
    hackrfdiags_amd/synth_torch.py -- FM test signal seeded per channel, every fifth channel uniform random bytes);
  * `pick_channels` draws >= 64 channels from the WHOLE range -- always 0, C - 1 and at least one per residue mod 8
    (and per residue mod 8 of c // 8: the second level of such a map) -- and only THOSE go through the sequential CPU
    oracle (Radio.cc:164-237: one state chain per radio), block after block, launch after launch;
  * the old all-channel property ("equal input => equal output") is kept as a second check with a period COPRIME to 8
    (`PERIOD` = 7), so that no two channels of one residue class share an input."""
import numpy as np

from hackrfdiags_amd import synth

BLK = synth.BLOCK_BYTES
PERIOD = 7                      # coprime to the 8 XCDs (and to 256 CUs, 16 waves, 64 lanes)


def pick_channels(C: int, n: int = 64, seed: int = 2026) -> list:
    """n channels of range(C), sorted: 0, C - 1, one per residue mod 8, one per residue of (c // 8) mod 8, the rest
    drawn from the whole range by a seeded generator."""
    rng = np.random.default_rng(seed + C)
    chosen = {0, C - 1}
    for r in range(8):
        cand = np.arange(r, C, 8)
        if len(cand):
            chosen.add(int(rng.choice(cand)))
        cand = np.array([c for c in range(C) if (c // 8) % 8 == r][:4096])
        if len(cand):
            chosen.add(int(rng.choice(cand)))
    n = min(n, C)
    while len(chosen) < n:
        chosen.add(int(rng.integers(0, C)))
    return sorted(chosen)


def distinct_batch(C: int, B: int, device, first_channel: int = 0):
    """[C][B][262144] int8 on the device, every channel different from every other: the FM test signal of SURVEY 8(d)
    with the channel's own noise seed and tone, every fifth channel uniform random bytes instead."""
    import torch
    from hackrfdiags_amd.synth_torch import make_fm_batch, make_random_batch
    x = make_fm_batch(C, B, device, first_channel=first_channel)
    rnd = torch.arange(3, C, 5, device=device)
    if len(rnd):
        x[rnd] = make_random_batch(len(rnd), B, device, first_channel=first_channel + 77)
    return x


def assert_all_distinct(x) -> None:
    """no two channels of the batch carry the same bytes (a cheap 64-bit fold per channel, on the device)"""
    import torch
    C = x.shape[0]
    w = x.reshape(C, -1)[:, :1 << 16].to(torch.int64)
    k = torch.arange(1, w.shape[1] + 1, device=x.device, dtype=torch.int64)
    sig = ((w + 129) * (k * 2654435761 % 1000003)).sum(dim=1)
    assert len(torch.unique(sig)) == C, "two channels carry the same input"


def oracle_rx_stream(oracle, mode, blocks_iter, threshold=None):
    """one channel through the sequential oracle: yields (pcm, magnitude, allowed) per block"""
    o = oracle.rx()
    o.set_mode(mode)
    if threshold is not None:
        o.set_threshold(threshold)
    for xb in blocks_iter:
        p, m, a, _ = o.process(xb)
        yield p, m, a


def _set_modes(rx, C, mode_of):
    modes = {mode_of(c) for c in range(C)}
    if len(modes) == 1:
        rx.set_mode(modes.pop())
        return
    for c in range(C):
        rx.set_mode(mode_of(c), channel=c)


def check_rx_bank_distinct(oracle, api, C, B, mode_of, launches=2, n_check=64, twin=None):
    """A bank of C channels (mode_of(c) each) x B blocks per launch, `launches` consecutive launches (the streams
    continue), EVERY channel fed an input of its own.  PCM, magnitude, n_pcm and the gate of the `pick_channels` sample
    against the sequential oracle; every launch committed, nothing replayed.  `twin(rx2)`: configure a second handle
    that must agree on ALL channels (another kernel over the same input: crc of the whole PCM).  Returns the handle."""
    import zlib
    import torch
    dev = torch.device("cuda:0")
    x = distinct_batch(C, launches * B, dev)
    assert_all_distinct(x)
    sel = pick_channels(C, n_check)
    tsel = torch.tensor(sel, device=dev)
    rx = api.Rx(C)
    other = None
    _set_modes(rx, C, mode_of)
    from tests.hooks import HOOKS_ON
    if twin is not None and HOOKS_ON:                    # (the second kernel is chosen through a hook: left out in the shipped state)
        other = api.Rx(C)
        _set_modes(other, C, mode_of)
        twin(other)
    got = []
    for k in range(launches):
        xs = x[:, k * B:(k + 1) * B].contiguous()
        out = torch.full((C, B, 512), 77, dtype=torch.int16, device=dev)
        mag = torch.zeros((C, B), dtype=torch.int32, device=dev)
        npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
        alw = torch.zeros((C, B), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()                         # torch fills on its own stream, the handle runs on another
        rx.process_device(xs.data_ptr(), B * BLK, BLK, B, out.data_ptr(), d_n_pcm=npcm.data_ptr(), d_magnitude=mag.data_ptr(),
                          d_allowed=alw.data_ptr())
        assert rx.sync() == 0, k
        assert int(npcm.sum().item()) == C * B * 512
        got.append((out[tsel].cpu().numpy(), mag[tsel].cpu().numpy(), alw[tsel].cpu().numpy()))
        if other is not None:
            out2 = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()
            other.process_device(xs.data_ptr(), B * BLK, BLK, B, out2.data_ptr())
            assert other.sync() == 0, k
            assert zlib.crc32(out2.cpu().numpy().tobytes()) == zlib.crc32(out.cpu().numpy().tobytes()), k
        del xs, out, mag, npcm, alw
    xsel = x[tsel].cpu().numpy()                         # [n_check][launches * B][262144]
    del x
    torch.cuda.empty_cache()
    for i, c in enumerate(sel):
        for b, (p, m, a) in enumerate(oracle_rx_stream(oracle, mode_of(c), xsel[i])):
            pcm, mag, alw = got[b // B]
            assert (pcm[i, b % B] == p).all(), (c, b)
            assert int(mag[i, b % B]) == m and bool(alw[i, b % B]) == a, (c, b)
    assert rx.debug_counters()[5] == 0, "a launch was replayed"
    if other is not None:
        assert other.debug_counters()[5] == 0
    return rx


def check_rx_bank_period(oracle, api, C, B, mode_of, launches=2, seed=200):
    """The all-channel property: inputs with period PERIOD = 7 (coprime to 8); the PERIOD base inputs of every mode
    against the oracle, and every channel equal to the first channel of its mode that was fed its input."""
    import torch
    dev = torch.device("cuda:0")
    base = [synth.make_input("fmtone" if k % 2 else "lcg", seed + k, launches * B).reshape(launches * B, BLK) for k in range(PERIOD)]
    bdev = torch.from_numpy(np.stack(base)).to(dev)
    idx = torch.arange(C, device=dev) % PERIOD
    rx = api.Rx(C)
    _set_modes(rx, C, mode_of)
    # first channel of each (mode, base input) pair
    firsts = {}
    for c in range(C):
        firsts.setdefault((mode_of(c), c % PERIOD), c)
    want = {key: list(oracle_rx_stream(oracle, key[0], base[key[1]])) for key in firsts}
    for k in range(launches):
        xs = bdev[idx, k * B:(k + 1) * B].contiguous()
        out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
        mag = torch.zeros((C, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        rx.process_device(xs.data_ptr(), B * BLK, BLK, B, out.data_ptr(), d_magnitude=mag.data_ptr())
        assert rx.sync() == 0, k
        got, gmag = out.cpu().numpy(), mag.cpu().numpy()
        for (mode, r), c0 in firsts.items():
            for b in range(B):
                p, m, _ = want[(mode, r)][k * B + b]
                assert (got[c0, b] == p).all() and int(gmag[c0, b]) == m, (k, mode, r, b)
        for c in range(C):
            c0 = firsts[(mode_of(c), c % PERIOD)]
            if c != c0:
                assert (got[c] == got[c0]).all() and (gmag[c] == gmag[c0]).all(), (k, c, c0)
        del xs, out, mag
    assert rx.debug_counters()[5] == 0
    del bdev
    torch.cuda.empty_cache()
