"""NumPy restatement of the patch front-end AFTER key-point detection
(src/sdav/input/CvInputParser.py:19-33, 49-123).  Test infrastructure only.

SURF (CvInputParser.py:36-46) is OpenCV-contrib non-free code that is not
available here and is NOT cloned: key-points are an input.  PARITY UNPINNED by
the reference (cv2 does not import), restated from the source text, including
its quirk: ``kp.pt`` is (x, y) but is used as ``img[x.., y..]`` with
``img.shape`` (rows, cols) as the (x, y) extents (:111-119)."""
import math

import numpy as np


def round_half_even(v):
    """Python 3 ``round`` on a float (CvInputParser.py:111)."""
    return int(round(float(v)))


def get_1d_boundaries(rect_shape, center_points, patch_size, axis):
    """CvInputParser.py:49-89: [lo, hi] of a patch_size window around each centre, shifted
    forward if lo < 0 and back if hi > dim - 1."""
    if patch_size % 2 == 0:
        raise ValueError("Invalid patch size. Patch size must be an odd number")
    if len(rect_shape) != 2:
        raise ValueError("Invalid rect shape. It must be a list of two integers")
    center_points = np.asarray(center_points)
    if center_points.ndim != 2:
        raise ValueError("Invalid center points. center_points must be a numpy array of 2D coordinates")
    if center_points.shape[1] != 2:
        raise ValueError("Invalid center points. Coordinates must be in 2D")
    shift = np.full(len(center_points), math.floor(patch_size / 2))
    c = center_points.transpose()[axis]
    lo, hi = c - shift, c + shift
    shift_forward = (lo < 0) * lo * -1
    aux = hi - rect_shape[axis] + 1
    shift_back = (aux > 0) * aux
    return lo - shift_back + shift_forward, hi - shift_back + shift_forward


def vectorized_patches(img, coordinates, patch_size):
    """CvInputParser.py:100-123 with the key-points already rounded to integer
    (x, y) pairs: int [n, patch_size^2]."""
    coordinates = np.asarray(coordinates, dtype=np.int64)
    x_lo, x_hi = get_1d_boundaries(img.shape, coordinates, patch_size, 0)
    y_lo, y_hi = get_1d_boundaries(img.shape, coordinates, patch_size, 1)
    out = np.empty((len(coordinates), patch_size ** 2), dtype=int)
    for i in range(len(coordinates)):
        out[i] = img[x_lo[i]:x_hi[i] + 1, y_lo[i]:y_hi[i] + 1].reshape(1, patch_size ** 2)
    return out


def parse(img, key_points_xy, patch_size=41):
    """CvInputParser.parse (:19-28) given the top-n key-point centres: float64 [n, ps^2] in [0,1]."""
    coords = np.array([[round_half_even(x), round_half_even(y)] for x, y in key_points_xy], dtype=np.int64)
    return vectorized_patches(np.asarray(img), coords, patch_size) / 255.0


def bgr2gray_opencv(rgb):
    """cv2.imread(..., IMREAD_GRAYSCALE) of a colour file (CvInputParser.py:32): OpenCV's
    fixed-point BT.601, (R*4899 + G*9617 + B*1868 + 8192) >> 14, on uint8 RGB [H,W,3]."""
    rgb = np.asarray(rgb, dtype=np.int64)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def read_ppm(path):
    """Binary P6 PPM (the datasets/ frames: 240x192, maxval 255) -> uint8 RGB [H,W,3]."""
    with open(path, "rb") as f:
        data = f.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    pos += 1
    assert tokens[0] == b"P6" and int(tokens[3]) == 255
    w, h = int(tokens[1]), int(tokens[2])
    return np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos).reshape(h, w, 3)
