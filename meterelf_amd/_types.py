"""Small value types of the path (reference: meterelf/_types.py:5-31)."""
from typing import NamedTuple, Tuple

Point = Tuple[int, int]
FloatPoint = Tuple[float, float]
Size = Tuple[int, int]


class DialCenter(NamedTuple):
    center: FloatPoint
    diameter: int


class Rect(NamedTuple):
    top_left: Point
    bottom_right: Point


class HlsColor(NamedTuple):
    """An (h, l, s) triple of ints in 0..255 (reference: meterelf/_colors.py:6-50)."""
    hue: int
    lightness: int
    saturation: int

    def get_range(self, color_range: 'HlsColor') -> Tuple['HlsColor', 'HlsColor']:
        lo = HlsColor(*(max(c - r, 0) for (c, r) in zip(self, color_range)))
        hi = HlsColor(*(min(c + r, 255) for (c, r) in zip(self, color_range)))
        return (lo, hi)
