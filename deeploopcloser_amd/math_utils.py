"""Host-side integer helper mirroring src/utils/MathUtils.py:1-4."""


class MathUtils:
    @staticmethod
    def compressed_size(value: int, compression: float):
        # same expression, same (banker's) rounding as the reference
        return int(round(value * ((100 - compression) / 100)))
