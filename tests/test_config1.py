"""BASELINE configs[0] at its stated size: the 20 frames of the reference's datasets/test, frames -> grey ->
key-points -> patches -> SDAV descriptors -> 20 x 20 cosine matrix + 20 x 20 SDAV-similarity matrix.

CPU (no GPU): the oracle end to end; its similarity matrix == what the REFERENCE's own
SimilarityCalculator returned for the same descriptors (tests/golden/config1.npz, captured by
tests/golden/make_config1_golden.py by importing the reference in the build container).
GPU: the same pipeline through the HIP path against the oracle and against that golden matrix."""
import numpy as np
import pytest

from conftest import load_golden
import config1_common as c1
from oracle import cosine as ocos
from oracle import similarity as osim

N = 20


@pytest.fixture(scope="module")
def patches():
    paths = c1.frame_paths()
    assert len(paths) == N and [p[-10:-4] for p in paths] == ["%06d" % i for i in range(N)]
    x = c1.oracle_patches(paths)
    assert x.shape == (N, 30, 1681) and x.min() >= 0.0 and x.max() <= 1.0
    return paths, x


@pytest.mark.parametrize("scale", ["reference", "fan_in"])
def test_config1_oracle_end_to_end_vs_reference_similarity(patches, scale):
    g = load_golden("config1.npz")
    _, x = patches
    h = c1.oracle_descriptors(x, scale)
    assert h.shape == (N * 30, 2500) and abs(h.sum() - float(g["descriptor_sum_" + scale])) < 1e-6 * h.sum()
    ds = h.reshape(N, 30, 2500)
    m = osim.similarity_matrix_f64(ds)
    want = g["similarity_f64_" + scale]                                   # the reference module's own outputs
    assert m.shape == (N, N) and np.array_equal(np.isfinite(m), np.isfinite(want))
    assert np.abs(m - want).max() <= 1e-9 * np.abs(want).max()
    mi = osim.similarity_matrix(ds)
    assert mi.dtype == np.int64 and np.all(np.diag(mi) == -1) and np.array_equal(mi, mi.T)
    assert np.array_equal(mi, osim.truncate_to_int64(want))               # create_similarity_matrix.py:31,36-37
    c = c1.oracle_cosine(h, N)
    assert c.shape == (N, N) and np.allclose(np.diag(c), 1.0) and np.allclose(c, c.T)
    _, idx = ocos.topk_from_scores(c, 3)
    assert np.array_equal(idx[:, 0], np.arange(N))                        # each frame's best match is itself


@pytest.mark.gpu
@pytest.mark.parametrize("scale", ["reference", "fan_in"])
def test_config1_gpu_end_to_end_vs_oracle(patches, scale):
    import torch
    import deeploopcloser_amd as dlc
    g = load_golden("config1.npz")
    paths, x = patches
    parser = dlc.CvInputParser(30, 41)
    rgb = np.stack([dlc.read_ppm(p) for p in paths])
    xg = parser.parse_batch(rgb).cpu().numpy()                            # grey, Harris, patch gather: all on the GPU
    assert np.array_equal(xg, x)                                          # byte work: bit-exact
    assert np.array_equal(np.stack([parser.parse_from_path(p) for p in paths]), x)
    net = dlc.SDAV(seed=c1.SEED, weight_scale=scale)
    h = dlc.encode(xg, net)
    ho = c1.oracle_descriptors(x, scale)
    assert h.shape == (N * 30, 2500) and np.abs(h - ho).max() < 1e-10
    # SDAV similarity 20 x 20: the reference's int64 matrix
    calc = dlc.SimilarityCalculator(h.reshape(N, 30, 2500))
    mf = calc.similarity_matrix(as_int64=False)
    want = g["similarity_f64_" + scale]
    assert np.abs(mf - want).max() <= 1e-9 * np.abs(want).max()
    assert np.array_equal(calc.similarity_matrix(), osim.truncate_to_int64(want))
    assert np.array_equal(calc.similarity_matrix(), osim.similarity_matrix(ho.reshape(N, 30, 2500)))
    assert calc.similarity_score(h[:30], h[30:60]) == mf[0, 1]            # the per-pair entry point
    # cosine 20 x 20 over the flattened place descriptors + top-k
    place = dlc.flatten_frame_descriptors(h)
    db = dlc.KeyframeDatabase(place, dtype="bf16", center=True)
    s = db.match(db.rows).cpu().numpy()
    stored = db.rows.float().cpu().numpy().astype(np.float64)
    assert s.shape == (N, N) and np.abs(s - ocos.scores(stored, stored)).max() < 2e-5
    assert np.abs(s - c1.oracle_cosine(ho, N)).max() < 6e-3              # + bf16 rounding of the stored rows
    ts, ti = db.match_topk(db.rows, 5)
    full = ocos.scores(stored, stored)
    es, ei = ocos.topk_from_scores(full, 5)
    # 75 008-d rows of real frames under an untrained encoder: every frame looks alike, scores crowd together --
    # the order is decided on fp64 scores, so the indices are the oracle's, slot for slot
    assert np.array_equal(ti.cpu().numpy(), ei)
    assert np.abs(ts.cpu().numpy() - es).max() < 1.2e-7
    torch.cuda.synchronize()


@pytest.fixture(scope="module")
def cnn_descriptors(patches):
    paths, _ = patches
    return c1.oracle_cnn_descriptors(paths)


def test_config3_oracle_distance_matrix_vs_reference(cnn_descriptors):
    """configs[2] on the same 20 frames: oracle CnnVtl descriptors -> oracle distance matrix == what the REFERENCE's
    DistanceCalculator returned for them in the loop of create_distance_matrix.py:30-36 (tests/golden/config1.npz)."""
    from oracle import distance as odist
    g = load_golden("config1.npz")
    d8 = cnn_descriptors
    assert d8.dtype == np.int8 and d8.shape == (N, int(g["cnn_descriptor_width"])) and d8.shape[1] <= 2243
    assert int(d8.astype(np.int64).sum()) == int(g["cnn_descriptor_sum"])
    m = odist.distance_matrix(d8)
    assert m.dtype == np.int64 and np.array_equal(m, g["distance_i64"])
    assert np.array_equal(m, m.T) and np.all(np.diag(m) == 0) and m.max() > 0


@pytest.mark.gpu
def test_config3_gpu_end_to_end_vs_oracle(patches, cnn_descriptors):
    import deeploopcloser_amd as dlc
    g = load_golden("config1.npz")
    paths, _ = patches
    net = dlc.CnnVtl(input_shape=[N, 192, 240, 3], seed=3, mask_seed=4)
    d = net.transform(c1.bgr_frames(paths))
    assert np.array_equal(d, cnn_descriptors)                             # every int8 byte
    m = dlc.DistanceCalculator.distance_matrix(d)
    assert np.array_equal(m, g["distance_i64"])                           # the reference's own integers
    assert dlc.DistanceCalculator.calculate_distance(d[3], d[17]) == g["distance_i64"][3, 17]
