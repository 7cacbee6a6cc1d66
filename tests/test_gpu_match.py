"""HIP matcher vs the oracle through the C-ABI: indices and distances bit-exact, including ties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("kind", ["uniform", "perturbed", "ties"])
@pytest.mark.parametrize("nq,nt", [(2000, 2000), (1, 1), (63, 65), (64, 16), (129, 15), (1000, 2500), (5, 1), (4000, 4001), (17, 255),
                                   (33, 257), (300, 5000)])
def test_knn2_bit_exact(mvo, O, ctx, kind, nq, nt, mfma):
    """Both implementations -- the i8 Gram on the matrix cores (default) and the vector-ALU kernel -- against the oracle."""
    mvo.debug_set("match_mfma", mfma)
    try:
        q, t = mvo.synth.match_inputs(kind, nq, nt)
        idx, dist = ctx.match_knn2(q, t)
    finally:
        mvo.debug_set("match_mfma", 1)
    io, do = O.match_knn2(q, t)
    assert np.array_equal(idx, io), "%d index rows differ" % (idx != io).any(1).sum()
    assert np.array_equal(dist, do)


def test_knn2_mfma_on_structured_descriptors(mvo, O, ctx):
    """Descriptors whose bits are NOT exchangeable (single bits, bytes, halves set; all zero; all one): a wrong bit -> byte
    placement or a swapped operand role in the MFMA shows here, on random descriptors it could hide."""
    rows = [np.zeros(32, np.uint8), np.full(32, 255, np.uint8)]
    for b in range(0, 256, 7):
        r = np.zeros(32, np.uint8)
        r[b // 8] = 1 << (b % 8)
        rows.append(r)
    for k in range(32):
        r = np.zeros(32, np.uint8)
        r[k] = 255
        rows.append(r)
        r2 = np.zeros(32, np.uint8)
        r2[:k] = 0xA5
        rows.append(r2)
    t = np.array(rows, np.uint8)
    rng = np.random.RandomState(5)
    q = t[rng.permutation(len(t))[:60]].copy()
    q[::3, 5] ^= 0x10                                       # near-duplicates: distance 1 to exactly one train row
    idx, dist = ctx.match_knn2(q, t)
    io, do = O.match_knn2(q, t)
    assert np.array_equal(idx, io) and np.array_equal(dist, do)


def test_knn2_empty_sets(mvo, O, ctx):
    q, t = mvo.synth.match_inputs("uniform", 10, 10)
    idx, dist = ctx.match_knn2(q, t[:0])
    assert (idx == -1).all() and (dist == np.iinfo(np.int32).max).all()
    idx, dist = ctx.match_knn2(q[:0], t)
    assert idx.shape == (0, 2)


@pytest.mark.parametrize("nq,nt,r", [(500, 700, 50.0), (64, 64, 0.0), (100, 3, 1e6), (257, 1000, 12.5)])
def test_radius_l1_bit_exact(mvo, O, ctx, nq, nt, r):
    rng = np.random.RandomState(nq + nt)
    q, t = mvo.synth.match_inputs("perturbed", nq, nt, seed=nq)
    qxy = rng.uniform(0, 640, (nq, 2)).astype(np.float32)
    txy = (qxy[rng.randint(0, nq, nt)] + rng.normal(0, 20, (nt, 2))).astype(np.float32)
    txy[: min(nq, nt)][::7] = qxy[: min(nq, nt)][::7]       # exact coincidences (distance 0, '<=' gate)
    idx, s = ctx.match_radius_l1(q, qxy, t, txy, r)
    io, so = O.match_radius_l1(q, qxy, t, txy, r)
    assert np.array_equal(idx, io) and np.array_equal(s, so)


@pytest.mark.parametrize("method", [1, 2, 3])
@pytest.mark.parametrize("kind", ["perturbed", "ties"])
def test_match_features_bit_exact(mvo, O, ctx, method, kind):
    q, t = mvo.synth.match_inputs(kind, 1200, 1500)
    rng = np.random.RandomState(3)
    xy1 = rng.uniform(0, 640, (1200, 2)).astype(np.float32)
    xy2 = rng.uniform(0, 640, (1500, 2)).astype(np.float32)
    for lowe in (1.0, 0.8):
        m = ctx.match_features(q, t, method, 2.0, lowe, xy1, xy2, 100.0)
        mo = O.match_features(q, t, method, 2.0, lowe, xy1, xy2, 100.0)
        assert m.tobytes() == mo.astype(m.dtype).tobytes(), (method, kind, lowe, len(m), len(mo))
    assert (np.diff(m["trainIdx"]) > 0).all()              # sorted by trainIdx, unique


def test_reference_style_free_functions(mvo, O):
    """matchFeatures / calcKeyPoints with the reference's names, default arguments and error behaviour."""
    mvo.reset_default_context()
    q, t = mvo.synth.match_inputs("perturbed", 300, 300)
    m = mvo.matchFeatures(q, t)                            # method_index = 1 (feature_match.h:22)
    mo = O.match_features(q, t, 1, 2.0, 1.0)
    assert m.tobytes() == mo.astype(m.dtype).tobytes()
    m2 = mvo.matchFeatures(q, t, 2)                        # lowe ratio latched as int(0.8 -> 1)
    assert m2.tobytes() == O.match_features(q, t, 2, 2.0, 1.0).astype(m.dtype).tobytes()
    with pytest.raises(RuntimeError, match="wrong method index"):
        mvo.matchFeatures(q, t, 7)
    img = mvo.synth.small_test_image(2, 320, 240)
    k = mvo.calcKeyPoints(img)
    k, d = mvo.calcDescriptors(img, k)
    ko = O.calc_keypoints(img, O.default_params())
    ko, do = O.calc_descriptors(img, ko, O.default_params())
    assert np.array_equal(d, do)
    mvo.reset_default_context()
