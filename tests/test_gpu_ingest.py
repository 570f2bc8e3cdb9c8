"""hrfd_ingest_*: the pinned, double-buffered block transport in front of hrfd_rx (SURVEY 8f
rank 2) must deliver exactly what the sequential CPU oracle delivers, batch after batch -- also when
channels fail their speculation in a batch and are replayed together with what is in flight behind it."""
import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests.reflib import WBFM, AM

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES


def _run_pipeline(rx, xs, B, n_slots, gain_db=0):
    """xs [C, n_batches*B, BLK] -> list of per-batch results, with up to n_slots batches in flight"""
    C, total = xs.shape[0], xs.shape[1]
    ing = api.Ingest(rx, BLK, B, n_slots)
    out, submitted, k = [], 0, 0
    n_batches = total // B
    while len(out) < n_batches:
        while submitted < n_batches and submitted - len(out) < n_slots:
            slot = ing.acquire()
            slot[...] = xs[:, submitted * B:(submitted + 1) * B]
            ing.submit(gain_db)
            submitted += 1
        out.append([np.array(a) for a in ing.collect()])
    replayed = ing.replayed()
    ing.close()
    return out, replayed


def _check_against_oracle(oracle, mode, xs, got, B, threshold=None):
    C, total = xs.shape[0], xs.shape[1]
    closed = 0
    for c in range(C):
        o = oracle.rx()
        o.set_mode(mode)
        if threshold is not None:
            o.set_threshold(threshold)
        for t in range(total):
            p, m, a, _ = o.process(xs[c, t])
            k, b = divmod(t, B)
            pcm, n_pcm, mag, allowed = got[k][0], got[k][1], got[k][2], got[k][3]
            assert n_pcm[c, b] == len(p) and bool(allowed[c, b]) == a and int(mag[c, b]) == m, (c, t)
            assert (pcm[c, b, :len(p)] == p).all(), (c, t)
            assert (pcm[c, b, len(p):] == 0).all(), (c, t)      # squelched units: zeros
            closed += 0 if a else 1
    return closed


@pytest.mark.parametrize("mode,n_slots", [(WBFM, 2), (WBFM, 3), (AM, 2)])
def test_ingest_pipeline_equals_oracle(oracle, mode, n_slots):
    C, B, NB = 4, 3, 5
    xs = np.stack([synth.make_input("fmtone" if c % 2 else "lcg", 120 + c, B * NB) for c in range(C)]).reshape(C, B * NB, BLK)
    rx = api.Rx(C); rx.set_mode(mode)
    got, replayed = _run_pipeline(rx, xs, B, n_slots)
    _check_against_oracle(oracle, mode, xs, got, B)
    assert replayed == 0


@pytest.mark.parametrize("gated", [True, False], ids=["gated_pass_on_the_device", "host_replay"])
def test_ingest_replays_failed_channels_in_order(oracle, gated):
    """a squelch gate that closes inside a batch breaks that channel's "all gates open" speculation.  With the gated
    second pass (the default) the device redoes the channel behind the batch launch, before the batch in flight behind
    it starts: nothing is replayed by the host.  Without it (test hook) the channel must be replayed in that batch
    and in the one already in flight behind it.  Either way every batch of every channel equals the sequential oracle"""
    C, B, NB = 3, 3, 5
    xs = np.stack([synth.make_input("fmtone", 130 + c, B * NB) for c in range(C)]).reshape(C, B * NB, BLK)
    xs[:2, 4:6] = 0                                  # silence in the middle of batch 1 (blocks 3..5), channels 0 and 1
    xs[0, 10] = 0                                    # and one silent block in batch 3 of channel 0; channel 2 never fails
    rx = api.Rx(C); rx.set_mode(WBFM); rx.set_threshold(-30)
    rx.debug_set_gated(gated)
    got, replayed = _run_pipeline(rx, xs, B, 2)
    closed = _check_against_oracle(oracle, WBFM, xs, got, B, threshold=-30)
    assert closed > 0 and (replayed == 0 if gated else replayed >= 2)
