"""params.yml -> Params (reference: meterelf/_params.py:17-155).

Same file format, same attribute names, same strict type checks and LoadError
messages.  Host-only: the GPU sees the scalars through melf_params
(include/meterelf_hip.h), built by Params.to_c().
"""
import os
from typing import Any, Dict, List, Optional

import yaml

from . import _hip
from ._types import DialCenter, HlsColor, Rect


class LoadError(Exception):
    pass


class _Getter:
    """Type-checked access to a YAML mapping (reference TypeCheckedGetter, :84-155)."""

    def __init__(self, data: Dict[Any, Any], base_dir: Optional[str] = None) -> None:
        self.data = data
        self.base_dir = base_dir

    def value(self, tp: type, name: str) -> Any:
        v = self.data[name]
        if not isinstance(v, tp):
            raise LoadError('{} is not {}'.format(name, tp.__name__))
        return v

    def items(self, name: str, tp: type, length: Optional[int] = None) -> List[Any]:
        seq = self.value(list, name)
        for (n, item) in enumerate(seq):
            if not isinstance(item, tp):
                raise LoadError('Item {} in {} is not {}'.format(n, name, tp.__name__))
        if length is not None and len(seq) != length:
            raise LoadError('{} must have exactly {} items'.format(name, length))
        return seq

    def path(self, name: str) -> str:
        bn = self.value(str, name)
        return os.path.join(self.base_dir, bn) if self.base_dir else bn

    def existing_file(self, name: str) -> str:
        fn = self.path(name)
        if not os.path.exists(fn):
            raise LoadError('File not found: {}'.format(fn))
        return fn

    def rect(self, name: str) -> Rect:
        sub = _Getter(self.data[name])
        (x0, y0) = sub.items('top_left', int, 2)
        (x1, y1) = sub.items('bottom_right', int, 2)
        return Rect(top_left=(x0, y0), bottom_right=(x1, y1))

    def hls(self, name: str) -> HlsColor:
        sub = _Getter(self.data[name])
        color = HlsColor(sub.value(int, 'h'), sub.value(int, 'l'), sub.value(int, 's'))
        for c in color:
            assert 0 <= c < 256  # HlsColor.__new__ asserts, meterelf/_colors.py:13-15
        return color


class Params:
    @classmethod
    def load(cls, filename: str) -> 'Params':
        try:
            with open(filename, 'rt') as fp:
                data = yaml.load(fp, Loader=yaml.SafeLoader)
        except Exception as error:
            raise LoadError('Cannot load YAML data from {}'.format(filename)) from error
        if not isinstance(data, dict):
            raise LoadError('Not a valid parameters file: {}'.format(filename))
        return cls(os.path.dirname(filename), data)

    def __init__(self, base_dir: str, data: Dict[Any, Any]) -> None:
        d = _Getter(data, base_dir=base_dir)
        self.image_glob: str = d.path('image_glob')
        self.meter_rect: Rect = d.rect('meter_rect')
        self.dials_file: str = d.existing_file('dials_template')
        self.dials_match_threshold: int = d.value(int, 'dials_template_match_threshold')
        (w, h) = d.items('dials_template_size', int, 2)
        self.dials_template_size = (h, w)  # (rows, cols), as the reference swaps it (:136-138)
        self.hue_shift: int = d.value(int, 'hue_shift')
        self.needle_color = d.hls('needle_color')
        self.needle_color_range = d.hls('needle_color_range')

        needles = d.items('needle_data', dict)
        if not needles:
            raise LoadError('Must have data of at least one needle')
        self.dial_color_range: Dict[str, HlsColor] = {}
        self.needle_dists_from_dial_center: Dict[str, int] = {}
        self.needle_circle_mask_thickness: Dict[str, int] = {}
        self.needle_angles_of_zero: Dict[str, float] = {}
        self.negative_momentum_dials = set()
        self.dial_centers: Dict[str, DialCenter] = {}
        for nd in needles:
            g = _Getter(nd)
            name = g.value(str, 'name')
            self.dial_color_range[name] = g.hls('color_range')
            self.needle_dists_from_dial_center[name] = g.value(int, 'dist_from_center')
            self.needle_circle_mask_thickness[name] = g.value(int, 'circle_thickness')
            self.needle_angles_of_zero[name] = g.value(float, 'angle_of_zero')
            (cx, cy) = g.items('center', float, 2)
            self.dial_centers[name] = DialCenter((cx, cy), g.value(int, 'diameter'))
            if g.value(bool, 'negative_momentum'):
                self.negative_momentum_dials.add(name)

    @property
    def dial_names(self) -> List[str]:
        return list(self.dial_centers.keys())

    def to_c(self, meter_rect: Optional[Rect] = None) -> '_hip.MelfParams':
        """The scalars the GPU needs, as melf_params."""
        names = self.dial_names
        if len(names) > _hip.MAX_DIALS:
            raise LoadError('At most {} needles are supported'.format(_hip.MAX_DIALS))
        p = _hip.MelfParams()
        p.abi_version = _hip.ABI_VERSION
        rect = meter_rect or self.meter_rect
        (p.rect_x0, p.rect_y0), (p.rect_x1, p.rect_y1) = rect.top_left, rect.bottom_right
        (p.th, p.tw) = self.dials_template_size
        p.hue_shift = self.hue_shift
        p.ndials = len(names)
        (lo, hi) = self.needle_color.get_range(self.needle_color_range)
        for c in range(3):
            p.needle_lo[c], p.needle_hi[c] = lo[c], hi[c]
        for (k, i) in enumerate(sorted(range(len(names)), key=lambda i: names[i])):
            p.name_order[k] = i
        p.match_threshold = float(self.dials_match_threshold)
        for (i, name) in enumerate(names):
            dl = p.dial[i]
            (dl.cx, dl.cy) = self.dial_centers[name].center
            dl.diameter = self.dial_centers[name].diameter
            dl.angle_of_zero = self.needle_angles_of_zero[name]
            (dl.range_h, dl.range_l, dl.range_s) = self.dial_color_range[name]
            dl.negative_momentum = 1 if name in self.negative_momentum_dials else 0
            dl.dist_from_center = self.needle_dists_from_dial_center[name]
            dl.circle_thickness = self.needle_circle_mask_thickness[name]
        return p


def load(filename: str) -> Params:
    return Params.load(filename)
