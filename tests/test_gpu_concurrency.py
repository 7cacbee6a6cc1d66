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


def test_throughput_mode_follows_the_load(mvo, O):
    """mvo_ba_set_mode(THROUGHPUT) = "many sequences share this GPU": a lone caller's windows keep the launch path and the
    latency cut (the resident grid would hold 2 x 14 CUs of every XCD for nothing); 24 callers submitting back to back
    bring the resident solver service up, and their windows move to its slots.  Whatever the route, a result equals the ORACLE's
    result for one of the two cuts bit for bit."""
    pb = mvo.synth.ba_problem(5, 2000, 7)
    args = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    ref = mvo.Context(0)
    lat = ref.bundle_adjustment(*args, fix_points=False)
    plan_lat = ref.ba_plan()
    assert plan_lat["wgs"] == 28
    ref.ba_set_mode("throughput")
    mvo.debug_set("ba_service", 2)
    try:
        svc = ref.bundle_adjustment(*args, fix_points=False)
        plan_svc = ref.ba_plan()
        assert plan_svc["wgs"] == 14
    finally:
        mvo.debug_set("ba_service", 1)
    # what a result may be is said by the ORACLE (its blocked twin with the plan of either cut), not by an earlier GPU run
    olat = O.bundle_adjustment_blocked(*args, plan=plan_lat, fix_points=False)
    osvc = O.bundle_adjustment_blocked(*args, plan=plan_svc, fix_points=False)
    assert lat[0].tobytes() == olat[0].tobytes() and lat[1].tobytes() == olat[1].tobytes()
    assert svc[0].tobytes() == osvc[0].tobytes() and svc[1].tobytes() == osvc[1].tobytes()
    ref.synchronize()
    ref.ba_launch_stats(reset=True)
    lone = ref.bundle_adjustment(*args, fix_points=False)                       # default policy, no load
    assert ref.ba_plan()["wgs"] == 28 and lone[0].tobytes() == lat[0].tobytes()
    assert ref.ba_launch_stats()["resident_windows"] == 0
    ok = {(r[0].tobytes(), r[1].tobytes()) for r in (olat, osvc)}
    errors, routes = [], []

    def worker(k):
        try:
            c = mvo.Context(0)
            c.ba_set_mode("throughput")
            for _ in range(40):
                P, X, st = c.bundle_adjustment(*args, fix_points=False)
                assert (P.tobytes(), X.tobytes()) in ok
                routes.append(c.ba_plan()["wgs"])
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(24)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    stats = ref.ba_launch_stats()
    assert stats["resident_windows"] > 200 and routes.count(14) > 200, (stats, routes.count(14), routes.count(28))
    ref.close()


def test_sibling_context_shares_the_stream_and_nothing_else(mvo):
    """mvo_create_sibling: the second context of a sequence works on its parent's stream (no hardware queue of its own) but
    keeps its own workspaces and mode; results equal those of independent contexts."""
    img = mvo.synth.small_test_image(3, 320, 240)
    pb = mvo.synth.ba_problem(4, 500, 40)
    args = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
    ref = mvo.Context(0, max_keypoints=800)
    k_ref = ref.calc_keypoints(img)
    P_ref, X_ref, _ = ref.bundle_adjustment(*args, fix_points=False, max_iterations=10)
    ref.close()
    parent = mvo.Context(0, max_keypoints=800)
    sib = parent.sibling()
    for _ in range(3):
        P, X, st = sib.bundle_adjustment(*args, fix_points=False, max_iterations=10)
        k = parent.calc_keypoints(img)
        assert k.tobytes() == k_ref.tobytes() and P.tobytes() == P_ref.tobytes() and X.tobytes() == X_ref.tobytes()
    sib.ba_set_mode("throughput")                            # the mode is the sibling's own
    assert parent.calc_keypoints(img).tobytes() == k_ref.tobytes()
    sib.close()
    k = parent.calc_keypoints(img)                           # the parent (and its stream) outlive the sibling
    assert k.tobytes() == k_ref.tobytes()
    parent.close()
