"""Error types of the reading path.

Same class names, hierarchy, constructor and message text as the reference's
meterelf/exceptions.py:4-52 -- the messages are part of the golden stdout
(`UNKNOWN Dials not found (match val = ...)`), so they are boundary, not style.
"""
from typing import Any, Dict, Optional


class ImageProcessingError(Exception):
    default_message = "Unable to process image"

    def __init__(self, filename: str = '', message: Optional[str] = None,
                 extra_info: Optional[Dict[str, Any]] = None) -> None:
        super().__init__()
        self.filename = filename
        self.message = message if message else self.default_message
        self.extra_info = extra_info

    def get_message(self, *, with_filename: bool = False, with_extra_info: bool = True) -> str:
        parts = [self.message]
        if with_filename and self.filename:
            parts.append(' from file: ' + self.filename)
        if with_extra_info and self.extra_info:
            details = ', '.join('{} = {}'.format(k, v) for (k, v) in self.extra_info.items())
            if details:
                parts.append(' (' + details + ')')
        return ''.join(parts)

    def __str__(self) -> str:
        return self.get_message(with_filename=True, with_extra_info=True)


class ImageLoadingError(ImageProcessingError, IOError):
    default_message = "Unable to load image"


class ImageAnalyzingError(ImageProcessingError, ValueError):
    default_message = "Failed to analyze image"


class DialsNotFoundError(ImageAnalyzingError):
    default_message = "Dials not found"


class DialAngleDeterminingError(ImageAnalyzingError):
    default_message = "Cannot determine angle of a dial"


class NeedleContoursNotFoundError(ImageAnalyzingError):
    default_message = "Cannot find needle contours of a dial"
