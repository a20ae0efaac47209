"""CPU oracle for the deepLoopCloser hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain NumPy / C fp64 restatement of the reference's
algorithms for the path  encode (SDAV / DA / CnnVtl forward) -> all-vs-all
similarity / distance -> top-k match.  It exists so that the HIP product path
in ``deeploopcloser_amd`` can be checked against the reference's arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.  The product package never does:
it fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * similarity.py, distance.py, math_utils  -- PINNED: checked against outputs
    of the reference's own NumPy-only modules, generated in the build container
    by ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``.
  * tensor_ops.tw_matmul                    -- PINNED by the literal of the
    reference's only test (test/TensorflowWrapperTest.py:12-14).
  * sdav.py, cnn_vtl.py (encoder forward)   -- PARITY UNPINNED by the reference:
    TensorFlow is not installable here and the reference ships no golden
    vectors or weights for these; the restatement follows the source text.
  * cosine.py (cosine + top-k)              -- PARITY UNPINNED: the function does
    not exist in the reference; it is defined by BASELINE.json's north_star.
"""
from . import tensor_ops, sdav, cnn_vtl, similarity, distance, cosine, math_utils, patches  # noqa: F401
