"""Image loading for the path (reference: meterelf/_image.py:12-55).

Baseline JPEG files are decoded on the GPU (melf_jpeg_process_files, the default of get_meter_values); this module
is the HOST decoder behind cv2.imread's other cases -- formats the GPU decoder does not take (PNG, progressive or
CMYK JPEG ...), files it reports as corrupt, and METERELF_DECODE=host -- with Pillow's libjpeg-turbo (ISLOW IDCT +
fancy upsampling, cv2's defaults).  Everything after the decode -- crop, HLS, dial finding -- happens on the GPU.
"""
from typing import Optional

import numpy as np

from ._params import Params as _Params
from .exceptions import ImageLoadingError


def imread_bgr(filename: str) -> Optional[np.ndarray]:
    """cv2.imread(filename): H x W x 3 uint8 BGR, or None if unreadable.

    Like cv2.imread (OpenCV 3.4) the EXIF orientation tag is applied, and a JPEG file that ends early yields the part
    that could be decoded with the rest grey: libjpeg's data source answers a premature end of file with a fake EOI
    marker, after which every remaining block decodes as all-zero coefficients.  The same marker is appended here for
    the same effect (Pillow's own LOAD_TRUNCATED_IMAGES switch is process-global and leaves the missing rows black)."""
    import io

    from PIL import Image, ImageOps
    try:
        with open(filename, 'rb') as fp:
            data = fp.read()
        if data[:2] == b'\xff\xd8' and data[-2:] != b'\xff\xd9':
            data += b'\xff\xd9'
        with Image.open(io.BytesIO(data)) as im:
            im = ImageOps.exif_transpose(im)
            rgb = np.asarray(im.convert('RGB'), dtype=np.uint8)
    except Exception:
        return None
    if rgb.ndim != 3 or rgb.shape[0] == 0 or rgb.shape[1] == 0:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


class ImageFile:
    """Same constructor as the reference's ImageFile (meterelf/_image.py:13-21).

    `bgr_image`, when given, is used instead of reading `filename` and -- as in
    the reference (:47-48) -- is taken to be the meter_rect crop already.
    """

    def __init__(self, filename: str, params: _Params, bgr_image: Optional[np.ndarray] = None) -> None:
        self.filename = filename
        self.params = params
        self.bgr_image = bgr_image

    @property
    def is_cropped(self) -> bool:
        return self.bgr_image is not None

    def get_frame(self) -> np.ndarray:
        """The array handed to the GPU: the full decoded frame, or the injected crop."""
        if self.bgr_image is not None:
            return self.bgr_image
        img = imread_bgr(self.filename)
        if img is None:
            raise ImageLoadingError(self.filename)
        return img
