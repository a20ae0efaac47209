"""deeploopcloser_amd -- MI355X-native loop-closure descriptor-and-match engine.

The hot path of nschejtman/deepLoopCloser (encode -> all-vs-all similarity ->
top-k match) as hand-written HIP kernels for gfx950 behind the C ABI of
include/dlc.h, with the reference's Python call surface on top:

    SDAV, DA                      (src/sdav/network)
    CnnVtl                        (src/cnn_vtl/network)
    CvInputParser                 (src/sdav/input; key-points supplied by the caller)
    SimilarityCalculator          (src/sdav/similarity; + SimilarityStream: one new frame against the resident ones)
    DistanceCalculator            (src/cnn_vtl/similarity)
    MathUtils                     (src/utils/MathUtils.py)
    tensor_wrapper (tw)           (src/utils/TensorflowWrapper.py)
    encode / match / match_topk   (BASELINE.json north_star; new)

Importing the package is cheap and works without a GPU; constructing any of
the classes needs libdlc_hip.so and a visible MI355X and raises otherwise.
"""
from . import _lib
from .math_utils import MathUtils
from .engine import Engine, default_engine
from .sdav import SDAV, DA
from .cnn_vtl import CnnVtl
from .similarity import SimilarityCalculator, SimilarityStream
from .distance import DistanceCalculator
from .matching import encode, match, match_topk, KeyframeDatabase, MatchPipeline, flatten_frame_descriptors
from .dist import ShardedKeyframeDatabase, shard_bounds, merge_topk_torch
from .input import CvInputParser, KeyPoint, grid_key_points, harris_key_points, read_ppm
from . import tensor_wrapper
from .loop_closure import LoopClosureDetector, SdavLoopClosureDetector

__all__ = ["LoopClosureDetector", "SdavLoopClosureDetector", "SimilarityStream", "SDAV", "DA", "CnnVtl", "SimilarityCalculator", "DistanceCalculator", "MathUtils", "tensor_wrapper", "CvInputParser",
           "grid_key_points", "harris_key_points", "KeyPoint", "read_ppm",
           "encode", "match", "match_topk", "KeyframeDatabase", "MatchPipeline", "ShardedKeyframeDatabase", "Engine",
           "default_engine", "shard_bounds", "merge_topk_torch", "flatten_frame_descriptors"]
