"""The load policy of the resident solver service (mvo_ba_set_mode THROUGHPUT): windows go to the resident
grid only while the offered load keeps most of its 16 slots busy.  The estimate is a plain state machine over submission
times; mvo_debug_ba_demand_replay runs it over synthetic arrival patterns (no device involved)."""
import ctypes as C

import numpy as np


def _replay(mvo, times):
    lib = mvo.load_library()
    t = np.ascontiguousarray(times, np.float64)
    out = np.zeros(len(t), np.uint8)
    lib.mvo_debug_ba_demand_replay.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    flips = lib.mvo_debug_ba_demand_replay(t.ctypes.data_as(C.c_void_p), len(t), out.ctypes.data_as(C.c_void_p))
    assert flips >= 0
    return out.astype(bool), flips


def _bursts(period, n, steps, t0=0.0, jitter=0.002, seed=0):
    rng = np.random.RandomState(seed)
    return np.sort(np.concatenate([t0 + s * period + rng.uniform(0, jitter, n) for s in range(steps)]))


def test_the_saturated_launch_path_brings_the_grid_up(mvo):
    """32 sequences in a closed loop on the launch path deliver ~2500 windows/s (9.6 slots' worth): the grid must come up from
    there -- with the threshold at 10 slots it never did when a run's first steps were slow, and the run stayed at half speed."""
    on, flips = _replay(mvo, np.sort(np.random.RandomState(7).uniform(0, 0.5, 1250)))
    assert flips == 1 and on[-1]
    on, flips = _replay(mvo, _bursts(0.0128, 32, 40))   # the same rate in lock step
    assert flips == 1 and on[-1]


def test_saturating_load_brings_the_grid_up_within_two_steps_and_keeps_it(mvo):
    # 24 sequences, one window each per 7-ms step = 3400 windows/s = 13 slots
    on, flips = _replay(mvo, _bursts(0.007, 24, 60))
    first = int(np.argmax(on))
    assert flips == 1 and on[-1] and first < 6 * 24 and on[first:].all()        # (128 submissions in a row at that rate)
    # the same rate without the lock step of the sequences
    t = np.sort(np.random.RandomState(1).uniform(0, 0.42, 1440))
    on, flips = _replay(mvo, t)
    assert flips == 1 and on[-1] and t[int(np.argmax(on))] < 0.05


def test_partial_load_stays_on_the_launch_path(mvo):
    # tracking rows in the loop: 24 windows per 16 ms = 1500 windows/s = 5.7 slots, in bursts
    on, flips = _replay(mvo, _bursts(0.016, 24, 80))
    assert flips == 0 and not on.any()
    # independent arrivals at half the rate that brings the grid up (8 of 16 slots = 2100 windows/s): no chance cluster does
    on, flips = _replay(mvo, np.sort(np.random.RandomState(2).uniform(0, 1.0, 1050)))
    assert flips == 0 and not on.any()
    # independent arrivals at the rate of the tracking-rows loop (1500 windows/s = 5.7 slots), 20 s of them: the grid (it would
    # hold 208 CUs for slots that are two thirds empty: 1800 frames/s with it, 2470 without) stays off, or -- when a chance
    # cluster of 128 arrivals at the 8-slot rate brings it up -- leaves again (it leaves below 7 slots; ADVICE r04: with the
    # leave threshold at 5 and 64 arrivals such a load flipped it on for good)
    for seed in range(4):
        on, flips = _replay(mvo, np.sort(np.random.RandomState(100 + seed).uniform(0, 20.0, 30000)))
        assert flips <= 8 and on.mean() < 0.03 and (flips % 2 == 0) == (not on[-1]), (seed, flips, on.mean())
    # one sequence at 400 frames/s
    on, flips = _replay(mvo, np.arange(400) * 0.0025)
    assert flips == 0 and not on.any()


def test_pauses_and_the_end_of_a_run_do_not_look_like_low_load(mvo):
    warm = _bursts(0.007, 24, 8)
    for pause in (0.03, 0.3, 3.0):                       # barrier between warm-up and timed region, a long stop
        t = np.concatenate([warm, _bursts(0.007, 24, 20, t0=warm[-1] + pause, seed=3)])
        on, flips = _replay(mvo, t)
        assert flips == 1 and on[len(warm):].all(), pause
    # the sequences finish one after the other: the rate of the last 40 ms tapers off
    steady = _bursts(0.007, 24, 30)
    tail = np.concatenate([_bursts(0.007, 24 - 3 * k, 1, t0=steady[-1] + 0.007 * (k + 1), seed=k) for k in range(7)])
    on, flips = _replay(mvo, np.concatenate([steady, tail]))
    assert flips == 1 and on[-1]


def test_sustained_low_load_takes_the_grid_off(mvo):
    high = _bursts(0.007, 24, 30)
    low = _bursts(0.024, 24, 40, t0=high[-1] + 0.2, seed=5)     # 1000 windows/s = 3.8 slots (the grid leaves below 5)
    on, flips = _replay(mvo, np.concatenate([high, low]))
    assert flips == 2 and on[len(high) - 1] and not on[-1]
    off_at = low[int(np.argmin(on[len(high):]))] - low[0]
    assert 0.08 < off_at < 0.3                            # after 80 ms of low load in a row, not at once
