"""DEBUG environment switch (reference: meterelf/_debug.py:3-14).  Only the
re-raise behaviour and the stdout suffix exist here; the GUI drawing of the
reference is out of scope."""
import os

_FALSY = {'0', 'no', 'off', 'false'}
DEBUG = {word for word in os.getenv('DEBUG', '').replace(',', ' ').split() if word.lower() not in _FALSY}
if 'all' in DEBUG:
    DEBUG = {'masks'}


def reraise_if_debug_on() -> None:
    """Call inside an `except` block: re-raises the active exception when DEBUG is set."""
    if DEBUG:
        raise
