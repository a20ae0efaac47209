"""SDAV patch front-end with the reference's call surface
(src/sdav/input/CvInputParser.py:14-33) on MI355X.

The reference picks the patch centres with OpenCV's SURF detector (:36-46), which is
non-free opencv-contrib code, unavailable here and deliberately NOT re-implemented.
Key-points are an argument: any sequence of objects with ``.pt`` (and optionally
``.response``, as cv2.KeyPoint has) or of (x, y) pairs.  With responses the reference's
ordering is applied (descending response, top n); everything after that -- rounding, the
clamp-inside-the-image window, flattening, /255.0 -- runs in the HIP kernel.  Without
key-points the build's own detector supplies them: Harris corners in exact integer arithmetic
(``harris_key_points``, ``CvInputParser.parse_batch``; HIP kernels, oracle/keypoints.py).
``grid_key_points`` is a deterministic stand-in for plumbing tests, not a detector.
"""
from collections import namedtuple

import numpy as np
import torch

from .engine import default_engine


def read_ppm(path):
    """Binary P6 PPM (the reference's datasets/*.ppm) -> uint8 RGB [H, W, 3]."""
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
    if tokens[0] != b"P6" or int(tokens[3]) != 255:
        raise ValueError("%s: only binary P6 PPM with maxval 255 is supported" % path)
    w, h = int(tokens[1]), int(tokens[2])
    return np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos).reshape(h, w, 3)


def grid_key_points(shape, n):
    """n deterministic (x, y) centres on a regular grid over an image of `shape` (rows, cols)."""
    rows = int(np.ceil(np.sqrt(n * shape[0] / shape[1])))
    cols = int(np.ceil(n / rows))
    xs = (np.arange(rows) + 0.5) * shape[0] / rows
    ys = (np.arange(cols) + 0.5) * shape[1] / cols
    pts = [(float(x), float(y)) for x in xs for y in ys]
    return pts[:n]


def _centres(key_points, n):
    kps = list(key_points)
    if kps and hasattr(kps[0], "response"):
        kps.sort(key=lambda kp: -kp.response)                       # CvInputParser.py:45
    kps = kps[:n]                                                    # :46
    pts = [kp.pt if hasattr(kp, "pt") else kp for kp in kps]
    return np.array([[int(round(float(p[0]))), int(round(float(p[1])))] for p in pts], dtype=np.int32).reshape(-1, 2)


KeyPoint = namedtuple("KeyPoint", "pt response")      # the two cv2.KeyPoint fields the reference reads


def _gray(engine, image):
    img = engine.to_device(image)
    if img.dim() == 3 and img.shape[-1] == 3:
        img = engine.rgb_to_gray(img.to(torch.uint8))
    if img.dim() != 2:
        raise ValueError("image must be [H, W] grey or [H, W, 3] RGB")
    return img.to(torch.uint8)


def harris_key_points(image, n, device=None):
    """The n strongest Harris corners of one image as cv2-style KeyPoint(pt=(x, y), response), strongest
    first: the build's stand-in for the reference's SURF detector (include/dlc.h, dlc_harris_keypoints_u8)."""
    e = default_engine(device)
    pts, resp, cnt = e.harris_keypoints(_gray(e, image).unsqueeze(0), n)
    c = int(cnt[0].item())
    pts, resp = pts[0, :c].cpu().numpy(), resp[0, :c].cpu().numpy()
    return [KeyPoint((float(x), float(y)), float(r)) for (x, y), r in zip(pts, resp)]


class CvInputParser:
    def __init__(self, n_patches: int = 30, patch_size: int = 41, device=None):
        self.n_patches = n_patches
        self.patch_size = patch_size
        self.engine = default_engine(device)
        self._grid = {}

    def parse_batch(self, frames):
        """frames uint8 [F,H,W,3] RGB or [F,H,W] grey -> [F, n_patches, patch_size^2] float64 on the GPU:
        grey conversion, Harris key-points, patch gather, nothing through the host.  A frame with
        fewer than n_patches corners is topped up with grid points (the network needs the shape)."""
        e = self.engine
        fr = e.to_device(frames)
        gray = e.rgb_to_gray(fr.to(torch.uint8)) if fr.dim() == 4 else fr.to(torch.uint8)
        if gray.dim() != 3:
            raise ValueError("frames must be [F, H, W, 3] RGB or [F, H, W] grey")
        pts, _, _ = e.harris_keypoints(gray, self.n_patches)
        shape = tuple(gray.shape[1:])
        if shape not in self._grid:                        # resident top-up points per frame size
            self._grid[shape] = torch.from_numpy(_centres(grid_key_points(shape, self.n_patches), self.n_patches)).to(e.device)
        pts = torch.where(pts < 0, self._grid[shape].unsqueeze(0).expand_as(pts), pts)
        return e.extract_patches(gray, pts, self.patch_size)

    def parse_tensor(self, image, key_points=None):
        e = self.engine
        img = _gray(e, image)
        if key_points is None:                 # the build's detector instead of SURF
            return self.parse_batch(img.unsqueeze(0))[0]
        kp = _centres(key_points, self.n_patches)
        if kp.shape[0] == 0:
            return torch.empty((0, self.patch_size ** 2), dtype=torch.float64, device=e.device)
        kpt = torch.from_numpy(kp).to(e.device).unsqueeze(0)
        return e.extract_patches(img.to(torch.uint8).unsqueeze(0), kpt, self.patch_size)[0]

    def parse(self, image, key_points=None):
        """CvInputParser.parse (:19-28): float64 [number of key-points (<= n_patches), patch_size^2].
        key_points=None: the n_patches strongest Harris corners (the reference: SURF)."""
        return self.parse_tensor(image, key_points).cpu().numpy()

    def parse_from_path(self, image_path, key_points=None):
        """CvInputParser.parse_from_path (:30-33) for PPM frames; grey conversion as cv2.imread does it."""
        return self.parse(read_ppm(str(image_path)), key_points)
