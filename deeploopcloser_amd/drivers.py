"""Counterparts of the reference's matrix drivers:
src/sdav/create_similarity_matrix.py and src/cnn_vtl/create_distance_matrix.py, with the O(N^2)
Python loops replaced by one GPU call each and the hard-coded paths by arguments.
Images are written with PIL (cv2 is not required)."""
import glob
import os

import numpy as np

from .cnn_vtl import CnnVtl
from .input import CvInputParser, read_ppm
from .sdav import SDAV
from .similarity import SimilarityCalculator


def _write_gray(path, img):
    from PIL import Image
    Image.fromarray(np.clip(np.rint(img), 0, 255).astype(np.uint8)).save(path)   # cv2.imwrite saturates + rounds


def similarity_image(similarity_matrix):
    """create_similarity_matrix.py:41-45: shift to 0, divide by the range, scale to 0..255."""
    m = np.asarray(similarity_matrix, dtype=np.float64)
    move = 0 - m.min()
    return 255 * ((m + move) / (m.max() + move))


def distance_image(distance_matrix):
    """create_distance_matrix.py:40: 255 - d / d.max() * 255."""
    d = np.asarray(distance_matrix, dtype=np.float64)
    return 255 - d / d.max() * 255


def _read_frames(files):
    """The dataset's frames as ONE uint8 array [N, H, W, 3] (RGB) -- or a list when their sizes differ."""
    frames = [read_ppm(f) for f in files]
    if all(fr.shape == frames[0].shape for fr in frames):
        return np.stack(frames)
    return frames


def create_similarity_matrix(dataset_path, out_png=None, network=None, key_points_fn=None, pattern="*"):
    """Frames of `dataset_path` -> patches -> SDAV descriptors -> int64 similarity matrix
    (create_similarity_matrix.py:23-38) [-> PNG].  key_points_fn(gray_shape) supplies the patch
    centres (the reference uses SURF; default: the build's Harris detector).
    Device-resident (pipeline.py): the uint8 frames go up once, in chunks that overlap the first kernels; patches and
    descriptors never leave HBM; the matrix comes down once."""
    from . import pipeline
    files = sorted(glob.glob(os.path.join(dataset_path, pattern)))
    if not files:
        raise ValueError("Specified dataset is empty or could not find dataset")        # InputGenerator.py:21-23
    network = network or SDAV()
    parser = CvInputParser(network.input_shape[0], int(round(np.sqrt(network.input_shape[1]))))
    frames = _read_frames(files)
    if isinstance(frames, list):
        # frames of several sizes (the reference parses them one by one, CvInputParser.py:30-33): each frame's patches are
        # gathered on the device, the stack never visits the host
        import torch
        x = torch.stack([parser.parse_tensor(fr, key_points_fn(fr.shape[:2]) if key_points_fn else None) for fr in frames])
        desc = network.transform_tensor(x).view(len(files), network.input_shape[0], -1)
        matrix = SimilarityCalculator(desc).similarity_matrix()
    else:
        kp = None
        if key_points_fn:
            kp = pipeline.key_point_array([key_points_fn(fr.shape[:2]) for fr in frames], network.input_shape[0], network.engine)
        matrix = pipeline.sdav_similarity_matrix_from_frames(frames, network, parser, key_points=kp)
    if out_png:
        finite = matrix.astype(np.float64)
        finite[matrix == np.iinfo(np.int64).min] = finite[matrix != np.iinfo(np.int64).min].max()
        _write_gray(out_png, similarity_image(finite))
    return matrix


def create_distance_matrix(dataset_path, out_png=None, network=None, pattern="*"):
    """Frames -> CnnVtl int8 descriptors -> int64 N x N distance matrix
    (create_distance_matrix.py:14-36) [-> PNG].  Device-resident (pipeline.py): uint8 frames up (chunks overlapping the
    convolutions), descriptors stay in HBM, the matrix comes down once."""
    from . import pipeline
    files = sorted(glob.glob(os.path.join(dataset_path, pattern)))
    if not files:
        raise ValueError("Specified dataset is empty or could not find dataset")
    frames = np.stack([read_ppm(f)[..., ::-1] for f in files])           # cv2.imread gives BGR (:23)
    network = network or CnnVtl(input_shape=[len(files)] + list(frames.shape[1:]))
    matrix = pipeline.cnn_vtl_distance_matrix_from_frames(frames, network)
    if out_png:
        _write_gray(out_png, distance_image(matrix))
    return matrix
