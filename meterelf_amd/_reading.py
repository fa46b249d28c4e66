"""get_meter_value(imgf) with the reference's signature and error behaviour
(reference: meterelf/_reading.py:19-115), computed on the GPU."""
from typing import Dict

from ._engine import get_reader, result_to_python
from ._image import ImageFile


def get_meter_value(imgf: ImageFile) -> Dict[str, float]:
    reader = get_reader(imgf.params)
    # the reference's digit combine asserts exactly four dials (_reading.py:166)
    assert len(reader.dial_names) == 4
    frame = imgf.get_frame()  # may raise ImageLoadingError
    rec = reader.read_many([frame], [imgf.is_cropped])[0]
    (values, error) = result_to_python(rec, reader.dial_names, imgf.filename)
    if error is not None:
        raise error
    return values
