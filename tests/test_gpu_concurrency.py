"""Several host threads, each with its own ctx, run the whole hot path at the same time (what bench.py's shards and a
multi-sequence caller do): extraction + matching on their own streams, bundle adjustments meeting in the library's launch
thread where they are batched into shared grids.  Every thread must get exactly what a lone ctx gets."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame_work(mvo, ctx, seed):
    img0 = mvo.synth.small_test_image(seed, 320, 240)
    img1 = np.roll(img0, 2, axis=1)
    out = []
    for img in (img0, img1):
        k = ctx.calc_keypoints(img)
        k, d = ctx.calc_descriptors(img, k, reuse_pyramid=True)
        out.append((k, d))
    m = ctx.match_features(out[0][1], out[1][1], 2, lowe_ratio=0.8)
    pb = mvo.synth.ba_problem(4, 500 + 50 * (seed % 3), 40 + seed)
    args = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    P, X, st = ctx.bundle_adjustment(*args, fix_points=False, max_iterations=20)
    return out[0][0].tobytes(), out[0][1].tobytes(), out[1][1].tobytes(), m.tobytes(), P.tobytes(), X.tobytes(), st["trials"]


def test_concurrent_contexts_reproduce_the_serial_results(mvo):
    n_threads, rounds = 6, 3
    ref_ctx = mvo.Context(0, max_keypoints=800)
    expected = {s: _frame_work(mvo, ref_ctx, s) for s in range(n_threads)}
    ref_ctx.close()
    errors, results = [], {}

    def worker(s):
        try:
            ctx = mvo.Context(0, max_keypoints=800)
            got = [_frame_work(mvo, ctx, s) for _ in range(rounds)]
            ctx.close()
            results[s] = got
        except Exception as e:  # noqa: BLE001
            errors.append((s, repr(e)))

    th = [threading.Thread(target=worker, args=(s,)) for s in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for s in range(n_threads):
        for r in range(rounds):
            assert results[s][r] == expected[s], "thread %d round %d differs from the lone-ctx result" % (s, r)
