"""CLI: `python -m meterelf_amd PARAMETERS_FILE [IMAGE_FILE...]`, one output line
per image in the reference's format (meterelf/_main.py:8-22)."""
import sys
from typing import Sequence

from . import _debug
from ._api import get_meter_values


def main(argv: Sequence[str] = sys.argv) -> None:
    if len(argv) < 2:
        raise SystemExit('Usage: {} PARAMETERS_FILE [IMAGE_FILE...]'.format(argv[0] if argv else 'meterelf'))
    for data in get_meter_values(argv[1], argv[2:]):
        line = data.filename + ': '
        if data.value:
            line += '{:07.3f}'.format(data.value)
        if data.error:
            line += 'UNKNOWN {}'.format(data.error.get_message())
        if _debug.DEBUG:
            line += ' {!r}'.format(data.meter_values)
        print(line)  # noqa
