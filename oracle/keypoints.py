"""NumPy restatement of the build's Harris key-point detector (test infrastructure only).

NOT IN THE REFERENCE: it picks patch centres with OpenCV-contrib's non-free SURF
(src/sdav/input/CvInputParser.py:36-46), which is neither available nor re-implemented
("key-points supplied by caller or a simple GPU detector") -- parity
unpinned by construction.  Definition (include/dlc.h, dlc_harris_keypoints_u8), all in exact
integer arithmetic: Sobel 3x3 gradients, structure tensor over the 5x5 window, response
16*det - trace^2 for pixels >= 3 from the border (0 elsewhere), 3x3 non-maximum suppression
(equal responses: the lower row-major index survives), the n largest responses (ties: lower
index), reported as cv2.KeyPoint-style (x = column, y = row)."""
import numpy as np


def response(gray):
    g = np.asarray(gray, dtype=np.int64)
    h, w = g.shape
    ix = np.zeros((h, w), dtype=np.int64)
    iy = np.zeros((h, w), dtype=np.int64)
    ix[1:-1, 1:-1] = (g[:-2, 2:] + 2 * g[1:-1, 2:] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[1:-1, :-2] + g[2:, :-2])
    iy[1:-1, 1:-1] = (g[2:, :-2] + 2 * g[2:, 1:-1] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[:-2, 1:-1] + g[:-2, 2:])
    r = np.zeros((h, w), dtype=np.int64)
    sxx = np.zeros((h - 6, w - 6), dtype=np.int64)
    syy = np.zeros_like(sxx)
    sxy = np.zeros_like(sxx)
    for dr in range(-2, 3):
        for dc in range(-2, 3):
            a = ix[3 + dr:h - 3 + dr, 3 + dc:w - 3 + dc]
            b = iy[3 + dr:h - 3 + dr, 3 + dc:w - 3 + dc]
            sxx += a * a
            syy += b * b
            sxy += a * b
    r[3:-3, 3:-3] = 16 * (sxx * syy - sxy * sxy) - (sxx + syy) ** 2
    return r


def key_points(gray, n):
    """-> (points int32 [n, 2] as (x = column, y = row), responses int64 [n], count); (-1,-1)/0 past count."""
    r = response(gray)
    h, w = r.shape
    cand = []
    for p in np.flatnonzero(r > 0):
        y, x = divmod(int(p), w)
        v = r[y, x]
        keep = True
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy == 0 and dx == 0:
                    continue
                u = r[y + dy, x + dx]
                q = (y + dy) * w + (x + dx)
                if u > v or (u == v and q < p):
                    keep = False
        if keep:
            cand.append((-int(v), int(p)))
    cand.sort()
    pts = np.full((n, 2), -1, dtype=np.int32)
    resp = np.zeros(n, dtype=np.int64)
    count = min(n, len(cand))
    for j in range(count):
        pts[j] = (cand[j][1] % w, cand[j][1] // w)
        resp[j] = -cand[j][0]
    return pts, resp, count
