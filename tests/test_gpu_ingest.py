"""hrfd_ingest_*: the pinned, double-buffered block transport in front of hrfd_rx (SURVEY 8f
rank 2) must deliver exactly what the blocking entry delivers, batch after batch -- also when a
batch fails its speculation and is replayed together with the one in flight behind it."""
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
        out.append(ing.collect())
    replayed = ing.replayed()
    ing.close()
    return out, replayed


@pytest.mark.parametrize("mode,n_slots", [(WBFM, 2), (WBFM, 3), (AM, 2)])
def test_ingest_pipeline_equals_blocking_entry(mode, n_slots):
    C, B, NB = 4, 3, 5
    xs = np.stack([synth.make_input("fmtone" if c % 2 else "lcg", 120 + c, B * NB) for c in range(C)]).reshape(C, B * NB, BLK)
    rx = api.Rx(C); rx.set_mode(mode)
    got, replayed = _run_pipeline(rx, xs, B, n_slots)
    ref = api.Rx(C); ref.set_mode(mode)
    for k in range(NB):
        pcm, n_pcm, mag, allowed, _ = ref.process_block(xs[:, k * B:(k + 1) * B], B)
        assert (got[k][0] == pcm).all() and (got[k][1] == n_pcm).all(), k
        assert (got[k][2] == mag).all() and (got[k][3] == allowed).all(), k
    assert replayed == 0


def test_ingest_replays_failed_batches_in_order():
    """a squelch gate that closes inside a batch breaks the batch's "all gates open" speculation:
    that batch and the one already in flight behind it must be replayed, and every batch must
    still equal the sequential result"""
    C, B, NB = 2, 3, 5
    xs = np.stack([synth.make_input("fmtone", 130 + c, B * NB) for c in range(C)]).reshape(C, B * NB, BLK)
    xs[:, 4:6] = 0                                   # silence in the middle of batch 1 (blocks 3..5)
    xs[0, 10] = 0                                    # and one silent block in batch 3
    rx = api.Rx(C); rx.set_mode(WBFM); rx.set_threshold(-30)
    got, replayed = _run_pipeline(rx, xs, B, 2)
    ref = api.Rx(C); ref.set_mode(WBFM); ref.set_threshold(-30)
    closed = 0
    for k in range(NB):
        pcm, n_pcm, mag, allowed, _ = ref.process_block(xs[:, k * B:(k + 1) * B], B)
        assert (got[k][1] == n_pcm).all() and (got[k][3] == allowed).all(), k
        assert (got[k][0] == pcm).all() and (got[k][2] == mag).all(), k
        closed += int((allowed == 0).sum())
    assert closed > 0 and replayed >= 2
