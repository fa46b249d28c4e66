"""Public API with the reference's signature (meterelf/_api.py:9-33):

    get_meter_values(params_file, filenames) -> Iterator[MeterImageData]

Lazy, yields in input order, per-image ImageProcessingError becomes the `error`
field (re-raised only when DEBUG is set).  Unlike the reference's one-at-a-time
loop, files go to the GPU in chunks (METERELF_BATCH, default 1024) and the library already
works on the next two chunks while the current chunk's results are consumed, so up to three chunks
are read ahead of the consumer.

cv2.imread of the reference (meterelf/_image.py:49): baseline JPEG files are read (on threads, inside the
library) and decoded on the GPU (METERELF_DECODE=gpu, the default; bit-identical to libjpeg's defaults); any other file, and
everything when METERELF_DECODE=host, is decoded on the host by Pillow on a small thread pool
(METERELF_DECODE_THREADS, default min(8, cpu count); Pillow releases the GIL).
"""
import hashlib
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional

from . import _debug, _params
from ._engine import MeterReader, make_blob, records_to_items, records_to_python, result_to_python
from ._image import ImageFile
from .exceptions import ImageProcessingError


class MeterImageData(NamedTuple):
    filename: str
    value: Optional[float]
    error: Optional[ImageProcessingError]
    meter_values: Dict[str, float]


# ---- GPU contexts kept between calls ------------------------------------------------------------------------------
# Creating a context (device tables, streams) and growing its workspaces on the first chunk cost ~30 ms per call -- more
# than the 1024 files of a chunk take to read.  The reference keeps its per-Params state in module-level caches too
# (meterelf/_image.py:69-81, meterelf/_dial_data.py:11-19, keyed by id(params)); here the key is the calibration itself
# (digest of the blob = params + template + masks, plus the dial names), so a params.yml edited between two calls gets a
# new context.  An idle context keeps its device workspaces (about 3 GB after 1024-file chunks of 640 x 480 frames);
# METERELF_CTX_CACHE=0 turns the cache off, release_cached_contexts() empties it.
_idle_readers: Dict[bytes, MeterReader] = {}
_idle_lock = threading.Lock()
_IDLE_MAX = 2


def _acquire_reader(params) -> MeterReader:
    blob = make_blob(params)
    key = hashlib.sha1(blob.tobytes() + repr(list(params.dial_names)).encode()).digest()
    with _idle_lock:
        reader = _idle_readers.pop(key, None)
    if reader is None:
        reader = MeterReader(params, blob=blob)
    reader._cache_key = key
    return reader


def _release_reader(reader: MeterReader) -> None:
    key = getattr(reader, '_cache_key', None)
    if key is not None and os.getenv('METERELF_CTX_CACHE', '1') != '0' and type(reader) is MeterReader:
        with _idle_lock:
            if key not in _idle_readers and len(_idle_readers) < _IDLE_MAX:
                _idle_readers[key] = reader
                return
    reader.close()


def release_cached_contexts() -> None:
    """Closes the GPU contexts get_meter_values keeps between calls (and frees their device memory)."""
    with _idle_lock:
        readers = list(_idle_readers.values())
        _idle_readers.clear()
    for r in readers:
        r.close()


import atexit  # noqa: E402

atexit.register(release_cached_contexts)   # before the interpreter tears the library binding down


def _chunks(items: Iterable[str], size: int) -> Iterator[List[str]]:
    chunk: List[str] = []
    for item in items:
        chunk.append(item)
        if len(chunk) == size:
            yield chunk
            chunk = []
    if chunk:
        yield chunk


_REAL_READER = MeterReader   # tests substitute MeterReader with a CPU stand-in: such readers are never cached


def get_meter_values(params_file: str, filenames: Iterable[str]) -> Iterator[MeterImageData]:
    params = _params.load(params_file)
    gpu_decode = os.getenv('METERELF_DECODE', 'gpu') != 'host'
    batch = max(1, int(os.getenv('METERELF_BATCH', '1024' if gpu_decode else '64')))
    if _debug.DEBUG:
        batch = 1  # DEBUG re-raises at the failing file, before any later file is touched
    reader: Optional[MeterReader] = None
    nthreads = max(1, int(os.getenv('METERELF_DECODE_THREADS', str(min(8, os.cpu_count() or 1)))))
    pool = ThreadPoolExecutor(max_workers=nthreads) if (nthreads > 1 and batch > 1) else None

    def _decode(filename: str):
        try:
            return ImageFile(filename, params).get_frame()
        except ImageProcessingError as e:
            return e

    # GPU decode: the library works on the chunks k + 1 and k + 2 on its own threads (melf_jpeg_process_files_begin / _end,
    # two calls in flight: chunk k + 2's files are read while chunk k + 1 decodes) while this thread turns chunk k's
    # records into Python objects and the consumer handles them.
    clean = False  # the generator ran to its end (or was closed between chunks with nothing in flight)
    DEPTH = 2

    def _gpu_read(chunk: List[str]):
        if hasattr(reader, 'read_jpeg_paths_batch'):
            (records, ok) = reader.read_jpeg_paths_batch(chunk)
            return (records, ok.tolist())
        recs = reader.read_jpeg_paths(chunk)
        return [None if rec is None else result_to_python(rec, reader.dial_names, f) for (rec, f) in zip(recs, chunk)]

    try:
        chunks = _chunks(filenames, batch)
        begun: List[List[str]] = []  # chunks handed to the library, oldest first

        def _begin_more() -> None:
            while len(begun) < DEPTH:
                nxt = next(chunks, None)
                if nxt is None:
                    return
                reader.read_jpeg_paths_begin(nxt)
                begun.append(nxt)

        chunk = next(chunks, None)
        while chunk is not None:
            if reader is None:
                reader = _acquire_reader(params) if MeterReader is _REAL_READER else MeterReader(params)
            assert len(reader.dial_names) == 4  # meterelf/_reading.py:166
            pipelined = gpu_decode and batch > 1 and hasattr(reader, 'read_jpeg_paths_begin')
            errors: Dict[int, ImageProcessingError] = {}
            by_index: Dict[int, object] = {}
            converted: list = [None] * len(chunk)  # (meter_values, error) of the files the GPU decoded
            raw = None
            if gpu_decode:  # files are read inside the library
                if begun:
                    assert begun[0] is chunk
                    begun.pop(0)
                    raw = reader.read_jpeg_paths_end()
                    raw = (raw[0], raw[1].tolist())
                else:
                    raw = _gpu_read(chunk)
            if pipelined:
                _begin_more()  # before this chunk's records are touched: the library has DEPTH chunks to work on
                following = begun[0] if begun else None
            else:
                following = next(chunks, None)
            items = None  # the chunk's result objects, when every file went through the GPU decoder
            if isinstance(raw, tuple):
                if not _debug.DEBUG and all(raw[1]):
                    items = records_to_items(raw[0], raw[1], reader.dial_names, chunk, MeterImageData)
                else:
                    converted = records_to_python(raw[0], raw[1], reader.dial_names, chunk)
            elif raw is not None:
                converted = raw
            if items is not None:
                yield from items
                chunk = following
                continue
            on_host = [i for i in range(len(chunk)) if converted[i] is None]
            if on_host:
                frames, where = [], []
                host_files = [chunk[i] for i in on_host]
                decoded = list(pool.map(_decode, host_files)) if pool is not None else [_decode(f) for f in host_files]
                for (i, item) in zip(on_host, decoded):
                    if isinstance(item, ImageProcessingError):
                        errors[i] = item
                        if _debug.DEBUG:
                            raise item
                    else:
                        frames.append(item)
                        where.append(i)
                if frames and pipelined:
                    reader.drain_jpeg_paths()  # host-decoded frames go through the context: nothing may be in flight on it
                records = reader.read_many(frames) if frames else []
                by_index.update(zip(where, records))
            for (i, filename) in enumerate(chunk):
                done = converted[i]
                if done is not None:
                    (meter_values, error) = done
                else:
                    meter_values = {}
                    error = errors.get(i)
                    if error is None:
                        (meter_values, error) = result_to_python(by_index[i], reader.dial_names, filename)
                if error is not None and _debug.DEBUG and i not in errors:
                    raise error
                yield MeterImageData(filename, meter_values.get('value'), error, meter_values)
            chunk = following
        clean = True
    except GeneratorExit:
        clean = True   # dropped by the consumer half way: whatever is in flight is collected below
        raise
    finally:
        if reader is not None and hasattr(reader, 'jpeg_paths_in_flight') and reader.jpeg_paths_in_flight():
            try:
                reader.discard_jpeg_paths()  # the context must be idle before it is closed or handed on
            except Exception:
                clean = False
        if pool is not None:
            pool.shutdown(wait=False)
        if reader is not None:
            if clean and MeterReader is _REAL_READER:
                _release_reader(reader)   # idle and in a known state: the next call with this calibration takes it over
            else:
                reader.close()
