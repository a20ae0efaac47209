"""The oracle against every fixture the reference offers for the hot path
(SURVEY.md section 8c): outputs of the reference's NumPy-only modules captured by
tests/golden/make_golden.py, and the literal of its only test."""
import numpy as np
import pytest

from oracle import similarity as osim, distance as odist, tensor_ops, math_utils, cosine as ocos

SIM_CASES = ["n6_h8", "n6_h64", "n4_h2500", "n5_h32_params", "n5_p7_h16"]


def sim_dataset(g, name):
    if name + "/dataset" in g:
        return g[name + "/dataset"]
    seed, n, p, h = g[name + "/dataset_seed_uniform01"]
    return np.random.RandomState(int(seed)).uniform(0.0, 1.0, size=(int(n), int(p), int(h)))


@pytest.mark.parametrize("name", SIM_CASES)
def test_similarity_pieces_match_reference(golden, name):
    g = golden("similarity.npz")
    ds = sim_dataset(g, name)
    mu, sigma, a, b = g[name + "/params"]
    avg = osim.average_response(ds)
    np.testing.assert_array_equal(avg, g[name + "/average_response"])
    s = osim.distinctive_score(avg, mu, sigma)
    np.testing.assert_array_equal(s, g[name + "/distinctive_score"])
    n = ds.shape[0]
    for i in range(n):
        for j in range(n):
            idx = osim.match_features(ds[i], ds[j])
            np.testing.assert_array_equal(idx, g[name + "/argmin"][i, j])
            d = osim.weighted_distances(ds[i], ds[j], idx, s)
            np.testing.assert_array_equal(d, g[name + "/weighted_distances"][i, j])
            got = osim.similarity_score(ds, ds[i], ds[j], mu, sigma, a, b)
            want = g[name + "/scores"][i, j]
            assert (np.isinf(want) and got == want) or got == want


def test_similarity_is_asymmetric_and_inf_on_identical(golden):
    g = golden("similarity.npz")
    sc = g["n6_h64/scores"]
    assert np.isposinf(sc[1, 3]) and np.isposinf(sc[3, 1])        # identical frames
    assert np.isposinf(sc[0, 0])                                  # self pair
    assert not np.allclose(sc[0, 1], sc[1, 0])                    # asymmetric


@pytest.mark.parametrize("name", ["n6_h8", "n6_h64", "n5_h32_params", "n5_p7_h16"])
def test_similarity_matrix_loop_semantics(golden, name):
    """create_similarity_matrix.py:31-38: i<j only, mirrored, int64 truncation, diag -1."""
    g = golden("similarity.npz")
    ds = sim_dataset(g, name)
    mu, sigma, a, b = g[name + "/params"]
    m = osim.similarity_matrix(ds, mu=mu, sigma=sigma, a=a, b=b)
    sc = g[name + "/scores"]
    n = ds.shape[0]
    assert m.dtype == np.int64
    for i in range(n):
        assert m[i, i] == -1
        for j in range(i + 1, n):
            want = osim.INT64_MIN if not np.isfinite(sc[i, j]) else int(np.trunc(sc[i, j]))
            assert m[i, j] == want and m[j, i] == want


def test_distance_matches_reference(golden):
    g = golden("distance.npz")
    assert odist.calculate_distance(g["probe/a"], g["probe/b"]) == g["probe/distance"] == 6
    np.testing.assert_array_equal(odist.bitwise_diff(g["all/a"], np.zeros(256, np.int8)), g["all/per_element"])
    for name in ("n7_d2243", "n9_d37", "n3_d1"):
        np.testing.assert_array_equal(odist.distance_matrix(g[name + "/desc"]), g[name + "/matrix"])


def test_compressed_size_matches_reference(golden):
    g = golden("mathutils.npz")
    got = [math_utils.compressed_size(int(v), float(g["compression"])) for v in g["values"]]
    np.testing.assert_array_equal(got, g["sizes"])
    got50 = [math_utils.compressed_size(int(v), 50.0) for v in g["values"]]
    np.testing.assert_array_equal(got50, g["sizes_50"])
    assert math_utils.compressed_size(279936, 99.59) == 1148


def test_tw_matmul_reference_known_answer(golden):
    """test/TensorflowWrapperTest.py:11-21 (the reference's only test)."""
    g = golden("tensorwrapper_test_example.npz")
    got = tensor_ops.tw_matmul(g["x"], g["w"])
    assert got.dtype == np.float64 and np.array_equal(got, g["expected"])


def test_cosine_topk_tie_break_and_merge():
    db = np.eye(8)[[0, 1, 1, 2, 1, 3]]            # rows 1,2,4 identical
    q = np.eye(8)[[1]]
    s, i = ocos.cosine_topk(q, db, 3)
    assert i.tolist() == [[1, 2, 4]] and np.allclose(s, 1.0)
    # sharded merge == global
    rng = np.random.RandomState(0)
    db = ocos.l2_normalize(rng.standard_normal((500, 32)))
    q = ocos.l2_normalize(rng.standard_normal((7, 32)))
    gs, gi = ocos.cosine_topk(q, db, 10)
    parts = [ocos.cosine_topk(q, db[lo:lo + 125], 10, row_offset=lo) for lo in range(0, 500, 125)]
    ms, mi = ocos.merge_topk(np.concatenate([p[0] for p in parts], 1), np.concatenate([p[1] for p in parts], 1), 10)
    np.testing.assert_array_equal(mi, gi)
    np.testing.assert_allclose(ms, gs, rtol=0, atol=1e-14)


def test_config1_plumbing_on_cpu():
    """BASELINE configs[0] without a GPU: the reference's own frames -> grey -> key-points ->
    30 patches -> SDAV forward -> cosine matrix on the flattened descriptors + the reference-semantics
    similarity matrix, all through the oracle (hidden width reduced so it runs in seconds)."""
    import glob
    import os
    from conftest import GOLDEN
    from oracle import cosine as ocos, keypoints as okp, patches as opatch, sdav as osdav, similarity as osim
    frames = sorted(glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")))
    assert len(frames) == 3
    x = []
    for f in frames:
        gray = opatch.bgr2gray_opencv(opatch.read_ppm(f))
        pts, resp, count = okp.key_points(gray, 30)
        assert count == 30 and np.all(np.diff(resp) <= 0)
        x.append(opatch.parse(gray, [tuple(p) for p in pts], 41))
    x = np.stack(x)
    assert x.shape == (3, 30, 1681) and 0.0 <= x.min() and x.max() <= 1.0
    ws, bs = osdav.init_weights(7, hidden_units=[48, 40, 32, 24, 16], scale="fan_in")
    h = osdav.transform(x, ws, bs)
    assert h.shape == (90, 16) and np.all((h > 0) & (h < 1))
    desc = h.reshape(3, 30, 16)
    place = ocos.l2_normalize(desc.reshape(3, -1), center=True)
    s = ocos.scores(place, place)
    assert np.allclose(np.diag(s), 1.0) and np.allclose(s, s.T) and np.all(s <= 1 + 1e-12)
    top_s, top_i = ocos.cosine_topk(place, place, 2)
    assert top_i[:, 0].tolist() == [0, 1, 2]                                # every frame's best match is itself
    m = osim.similarity_matrix(desc)
    assert m.dtype == np.int64 and m.shape == (3, 3) and np.all(np.diag(m) == -1) and np.array_equal(m, m.T)
