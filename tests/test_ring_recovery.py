"""A producer that dies between claiming a ring slot and publishing it must not wedge the ingestion ring
(parallel/csrc/shm_ring.cpp: the claim records its pid; the consumer skips a blocked slot whose claimant is gone)."""
import os
import time

import numpy as np

from pytorch_distributed_amd.parallel import hostlib
from pytorch_distributed_amd.parallel.ingest import ShmRing
from pytorch_distributed_amd.utils import mpctx as mp


def _claim_and_die(ring):
    pos = hostlib.lib().pda_ring_claim_and_abandon(ring.base)
    os._exit(0 if pos >= 0 else 3)          # killed "mid-copy": the slot stays claimed and unpublished


def _rec(v, n=64):
    return np.full(n, v, dtype=np.uint8)


def test_dead_producer_slot_is_skipped_and_later_records_arrive():
    ring = ShmRing(8, 64)
    L = hostlib.lib()
    L.pda_ring_set_stuck_ms(ring.base, 200)
    assert ring.push_segments([_rec(1)])
    p = mp.Process(target=_claim_and_die, args=(ring,))
    p.start()
    p.join(30)
    assert p.exitcode == 0
    for v in (3, 4, 5):                     # producers behind the dead one keep publishing
        assert ring.push_segments([_rec(v)])
    out = np.zeros((8, 64), dtype=np.uint8)
    assert ring.pop_many(out, 8) == 1 and out[0, 0] == 1        # blocked behind the abandoned slot ...
    assert ring.pop_many(out, 8) == 0
    time.sleep(0.3)
    assert ring.pop_many(out, 8) == 3                           # ... until it has been stuck long enough
    assert out[:3, 0].tolist() == [3, 4, 5]
    st = ring.stats()
    assert st["skipped"] == 1 and st["backlog"] == 0
    # the ring is fully usable again: fill it, drain it, twice around
    for rnd in range(2):
        for v in range(8):
            assert ring.push_segments([_rec(10 * rnd + v)])
        assert not ring.push_segments([_rec(99)], timeout_s=0.05)
        assert ring.pop_many(out, 8) == 8
        assert out[:, 0].tolist() == [10 * rnd + v for v in range(8)]


def test_live_slow_producer_is_never_skipped():
    ring = ShmRing(8, 64)
    L = hostlib.lib()
    L.pda_ring_set_stuck_ms(ring.base, 100)
    assert L.pda_ring_claim_and_abandon(ring.base) == 0          # THIS process holds the claim and stays alive
    assert ring.push_segments([_rec(7)])
    out = np.zeros((8, 64), dtype=np.uint8)
    assert ring.pop_many(out, 8) == 0
    time.sleep(0.25)
    assert ring.pop_many(out, 8) == 0 and ring.stats()["skipped"] == 0
