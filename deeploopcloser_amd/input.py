"""SDAV patch front-end with the reference's call surface
(src/sdav/input/CvInputParser.py:14-33) on MI355X.

The reference picks the patch centres with OpenCV's SURF detector (:36-46), which is
non-free opencv-contrib code, unavailable here and deliberately NOT re-implemented.
Key-points are therefore an argument: any sequence of objects with ``.pt`` (and optionally
``.response``, as cv2.KeyPoint has) or of (x, y) pairs.  With responses the reference's
ordering is applied (descending response, top n); everything after that -- rounding, the
clamp-inside-the-image window, flattening, /255.0 -- runs in the HIP kernel.
``grid_key_points`` is a deterministic stand-in for plumbing tests, not a detector.
"""
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


class CvInputParser:
    def __init__(self, n_patches: int = 30, patch_size: int = 41, device=None):
        self.n_patches = n_patches
        self.patch_size = patch_size
        self.engine = default_engine(device)

    def parse_tensor(self, image, key_points):
        e = self.engine
        img = e.to_device(image)
        if img.dim() == 3 and img.shape[-1] == 3:
            img = e.rgb_to_gray(img.to(torch.uint8))
        if img.dim() != 2:
            raise ValueError("image must be [H, W] grey or [H, W, 3] RGB")
        kp = _centres(key_points, self.n_patches)
        if kp.shape[0] == 0:
            return torch.empty((0, self.patch_size ** 2), dtype=torch.float64, device=e.device)
        kpt = torch.from_numpy(kp).to(e.device).unsqueeze(0)
        return e.extract_patches(img.to(torch.uint8).unsqueeze(0), kpt, self.patch_size)[0]

    def parse(self, image, key_points):
        """CvInputParser.parse (:19-28): float64 [number of key-points (<= n_patches), patch_size^2]."""
        return self.parse_tensor(image, key_points).cpu().numpy()

    def parse_from_path(self, image_path, key_points):
        """CvInputParser.parse_from_path (:30-33) for PPM frames; grey conversion as cv2.imread does it."""
        return self.parse(read_ppm(str(image_path)), key_points)
