"""Restatement of src/utils/MathUtils.py:1-4 (test infrastructure only)."""


def compressed_size(value: int, compression: float) -> int:
    """MathUtils.compressed_size (src/utils/MathUtils.py:3-4).

    ``int(round(value * ((100 - compression) / 100)))`` -- Python 3 ``round``
    is banker's rounding, the float expression is kept in the same order.
    """
    return int(round(value * ((100 - compression) / 100)))
