"""Public API with the reference's signature (meterelf/_api.py:9-33):

    get_meter_values(params_file, filenames) -> Iterator[MeterImageData]

Lazy, yields in input order, per-image ImageProcessingError becomes the `error`
field (re-raised only when DEBUG is set).  Unlike the reference's one-at-a-time
loop, files go to the GPU in chunks: 32 names first, then 64, 128 ... up to METERELF_BATCH (default 1024;
METERELF_BATCH_FIRST sets the first size, 0 = no ramp), so that a slow or unbounded `filenames` iterator has its first
results after 32 names; nothing is pulled from the iterator while a short chunk's results are outstanding.  Once the
chunks are full-sized the library already works on the next chunks while the current chunk's results are consumed, so up
to three chunks are read ahead of the consumer.

Several GPUs in one process (round 4): the chunks are dealt round-robin over METERELF_DEVICES (a comma list of
device indices; "all" = every visible device; "0,0" = two contexts on one GPU), one MeterReader and one
begin / end pipeline per device, each on its own thread; the results are yielded in input order.  Unset, ONE device is
used -- LOCAL_RANK's under torchrun (one process per GPU), else device 0: a rank of a
process-per-GPU job must not open contexts on its neighbours' GPUs, so fanning out is always an explicit choice.  Frames are
independent (meterelf/_api.py:22-33), so there is no collective: the process holds the calibration blob and every
device gets its own context from it.  With one device nothing changes (no threads).

cv2.imread of the reference (meterelf/_image.py:49): baseline JPEG files are read (on threads, inside the
library) and decoded on the GPU (METERELF_DECODE=gpu, the default; bit-identical to libjpeg's defaults); any other file, and
everything when METERELF_DECODE=host, is decoded on the host by Pillow on a small thread pool
(METERELF_DECODE_THREADS, default min(8, cpu count); Pillow releases the GIL).
"""
import collections
import hashlib
import os
import queue
import sys
import threading
from concurrent.futures import ThreadPoolExecutor
from time import perf_counter
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional

from . import _debug, _hip, _params
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
# (digest of the blob = params + template + masks, plus the dial names), the device, and the environment switches the
# library reads when it creates a context (MELF_MATCH, MELF_JPEG_*, ...: a changed switch gets a new context instead of a
# stale one), so a params.yml edited between two calls gets a new context.  An idle context keeps its device workspaces
# (about 3 GB after 1024-file chunks of 640 x 480 frames, of the GPU's 288 GB); at most METERELF_CTX_CACHE_MAX are kept
# (default: two per device used), METERELF_CTX_CACHE=0 turns the cache off, release_cached_contexts() empties it.  A
# fork()ed child forgets the parent's contexts without touching HIP (it must create its own).
_idle_readers: Dict[bytes, MeterReader] = {}
_idle_lock = threading.Lock()
_IDLE_MAX = 2
_devices_used = 1
_ENV_IN_KEY = ('MELF_MATCH', 'MELF_MATCH_LAYOUT', 'MELF_GEN_SHAPE', 'MELF_FORCE_GENERIC_MASK', 'MELF_FUSED_VARIANT', 'MELF_JPEG_CHUNK',
               'MELF_IO_THREADS', 'MELF_HOST_THREADS', 'MELF_LIB_PATH')


def _reader_key(params, blob, device: int) -> bytes:
    env = ';'.join('%s=%s' % (k, os.environ[k]) for k in _ENV_IN_KEY if k in os.environ)
    return hashlib.sha1(blob.tobytes() + repr(list(params.dial_names)).encode() + b'|dev%d|' % device + env.encode()).digest()


def _acquire_reader(params, device: int = 0) -> MeterReader:
    blob = make_blob(params)
    key = _reader_key(params, blob, device)
    with _idle_lock:
        reader = _idle_readers.pop(key, None)
    if reader is None:
        reader = MeterReader(params, device=device, blob=blob)
    reader._cache_key = key
    return reader


def _acquire_readers_bcast(params, devices: List[int]) -> Dict[int, MeterReader]:
    """The fan-out's readers, one per DISTINCT device: idle ones from the cache, the others created together -- the calibration
    blob uploaded to the first missing device and broadcast to the rest by RCCL over xGMI (melf_ctx_create_bcast; north_star:
    "RCCL broadcast of the template/params").  {} when that is not possible (a device listed twice, no RCCL, an error): the
    workers then create their contexts one by one from the host blob, which is equivalent."""
    if len(set(devices)) != len(devices) or len(devices) < 2 or os.getenv('METERELF_BCAST', '1') == '0':
        return {}
    blob = make_blob(params)
    out: Dict[int, MeterReader] = {}
    missing = []
    for (w, d) in enumerate(devices):
        key = _reader_key(params, blob, d)
        with _idle_lock:
            r = _idle_readers.pop(key, None)
        if r is not None:
            r._cache_key = key
            out[w] = r
        else:
            missing.append(w)
    if len(missing) == 1:     # one context to make: a plain upload (a communicator for a broadcast to oneself costs more than it moves)
        w = missing[0]
        try:
            r = MeterReader(params, device=devices[w], blob=blob)
        except Exception:
            for r2 in out.values():
                _release_reader(r2)
            return {}
        r._cache_key = _reader_key(params, blob, devices[w])
        out[w] = r
    elif missing:
        try:
            ctxs = _hip.Context.create_bcast(blob, [devices[w] for w in missing])
        except Exception:
            for r in out.values():
                _release_reader(r)
            return {}
        try:
            for (w, ctx) in zip(missing, ctxs):
                r = MeterReader(params, device=devices[w], blob=blob, ctx=ctx)
                r._cache_key = _reader_key(params, blob, devices[w])
                out[w] = r
        except Exception:     # nothing of a half-made set survives: the workers start from scratch
            for ctx in ctxs:
                ctx.close()
            for r in out.values():
                if getattr(r, 'ctx', None) not in ctxs:
                    _release_reader(r)
            return {}
    return out


def _release_reader(reader: MeterReader) -> None:
    key = getattr(reader, '_cache_key', None)
    limit = int(os.getenv('METERELF_CTX_CACHE_MAX', str(max(_IDLE_MAX, 2 * _devices_used))))
    if key is not None and os.getenv('METERELF_CTX_CACHE', '1') != '0' and type(reader) is MeterReader:
        with _idle_lock:
            if key not in _idle_readers and len(_idle_readers) < limit:
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


def _forget_contexts_in_child() -> None:
    # after fork() the parent's HIP state is not usable in the child: drop the handles WITHOUT calling into the library
    for r in list(_idle_readers.values()):
        for ctx in [getattr(r, 'ctx', None)] + list(getattr(r, '_crop_ctx', {}).values()):
            if ctx is not None:
                ctx._h = None
    _idle_readers.clear()


import atexit  # noqa: E402

atexit.register(release_cached_contexts)   # before the interpreter tears the library binding down
if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_forget_contexts_in_child)


_noted_one_device = False


def _device_list() -> List[int]:
    """METERELF_DEVICES: comma list of device indices (an index may repeat: that many contexts on that GPU), or "all".
    Unset or empty: one device -- LOCAL_RANK's (torchrun: one process per GPU), else device 0."""
    global _noted_one_device
    spec = os.getenv('METERELF_DEVICES', '').strip()
    if spec == '':
        rank = os.getenv('LOCAL_RANK', '').strip()
        ndev = max(1, _hip.device_count())
        if ndev > 1 and not rank.isdigit() and not _noted_one_device:
            _noted_one_device = True      # once per process: before round 5 an unset METERELF_DEVICES meant every visible GPU
            print('meterelf_amd: %d GPUs visible, using device 0 only (set METERELF_DEVICES=all or a comma list to fan out)' % ndev, file=sys.stderr)
        return [int(rank) % ndev] if rank.isdigit() else [0]
    if spec == 'all':
        return list(range(max(1, _hip.device_count())))
    return [int(x) for x in spec.split(',') if x.strip() != ''] or [0]


def _chunks(items: Iterable[str], size: int, first: int = 0) -> Iterator[List[str]]:
    """Lists of `size` names; with `first` > 0 the sizes ramp up geometrically from it (first, 2 first, 4 first ... size): a slow
    or unbounded `filenames` iterator gets its first results after `first` names instead of after a whole chunk (the reference
    pulls ONE name per result: meterelf/_api.py:22-33)."""
    want = min(size, first) if first > 0 else size
    chunk: List[str] = []
    for item in items:
        chunk.append(item)
        if len(chunk) == want:
            yield chunk
            chunk = []
            want = min(size, 2 * want)
    if chunk:
        yield chunk


_REAL_READER = MeterReader   # tests substitute MeterReader with a CPU stand-in: such readers are never cached

# Where the Python side of get_meter_values spends its time, per device pipeline (bench.py's jpeg_decode.get_meter_values.host):
# seconds summed over the chunks since the last api_stats(reset=True).  A few perf_counter() calls per 1024-file chunk.
_stats_lock = threading.Lock()
_LIB_STATS_KEPT = 64   # the newest per-pipeline library records api_stats() hands out (a long-running service never calls it)
_stats = {'chunks': 0, 'files': 0, 's_begin': 0.0, 's_end_wait': 0.0, 's_convert': 0.0, 'library': collections.deque(maxlen=_LIB_STATS_KEPT)}


def api_stats(reset: bool = True) -> dict:
    """{'chunks', 'files', 's_begin' (marshalling + melf_jpeg_process_files_begin), 's_end_wait' (blocked in _end: the library
    had not finished the chunk), 's_convert' (records -> MeterImageData objects), 'library': [melf_ctx_files_stats of every
    context released since the last reset, the newest 64]}."""
    with _stats_lock:
        out = dict(_stats, library=list(_stats['library']))
        if reset:
            _stats.update({'chunks': 0, 'files': 0, 's_begin': 0.0, 's_end_wait': 0.0, 's_convert': 0.0, 'library': collections.deque(maxlen=_LIB_STATS_KEPT)})
    return out


class _LazyItems:
    """A chunk's records, not yet turned into MeterImageData objects: the fan-out's workers hand these to the consumer thread,
    which converts them itself (round 5).  A worker then holds the interpreter lock for ~0.1 ms per chunk (collect the call, begin
    the next) instead of ~0.5 ms: with several device threads the lock's hand-overs, not the conversion, were what the consumer
    waited for."""
    __slots__ = ('records', 'ok', 'dial_names', 'chunk')

    def __init__(self, records, ok, dial_names, chunk):
        (self.records, self.ok, self.dial_names, self.chunk) = (records, ok, dial_names, chunk)

    def materialize(self) -> List['MeterImageData']:
        t0 = perf_counter()
        items = records_to_items(self.records, self.ok, self.dial_names, self.chunk, MeterImageData)
        with _stats_lock:
            _stats['s_convert'] += perf_counter() - t0
        return items


def _process_chunks(params, chunks: Iterator[List[str]], device: int, gpu_decode: bool, batch: int,
                    reader: Optional[MeterReader] = None, lazy: bool = False) -> Iterator[List[MeterImageData]]:
    """One device's pipeline: chunk lists in, one list of MeterImageData per chunk out, in order.  `reader`: made by the caller
    (the fan-out's broadcast-created contexts); handed back to the cache or closed here like one made here.  `lazy`: chunks whose
    files all went through the GPU decoder are yielded as _LazyItems (the caller converts)."""
    nthreads = max(1, int(os.getenv('METERELF_DECODE_THREADS', str(min(8, os.cpu_count() or 1)))))
    pool = ThreadPoolExecutor(max_workers=nthreads) if (nthreads > 1 and batch > 1) else None

    def _decode(filename: str):
        try:
            return ImageFile(filename, params).get_frame()
        except ImageProcessingError as e:
            return e

    # GPU decode: the library works on the chunks k + 1 .. k + 3 on its own threads (melf_jpeg_process_files_begin / _end,
    # three calls in flight: one reading its files, one preparing and enqueueing, one waiting for its kernels) while this
    # thread turns chunk k's records into Python objects and the consumer handles them.
    clean = False  # the generator ran to its end (or was closed between chunks with nothing in flight)
    DEPTH = _hip.FILES_IN_FLIGHT_MAX
    local = {'chunks': 0, 'files': 0, 's_begin': 0.0, 's_end_wait': 0.0, 's_convert': 0.0}

    def _gpu_read(chunk: List[str]):
        if hasattr(reader, 'read_jpeg_paths_batch'):
            (records, ok) = reader.read_jpeg_paths_batch(chunk)
            return (records, ok.tolist())
        recs = reader.read_jpeg_paths(chunk)
        return [None if rec is None else result_to_python(rec, reader.dial_names, f) for (rec, f) in zip(recs, chunk)]

    try:
        begun: List[List[str]] = []  # chunks handed to the library, oldest first

        def _begin_more() -> None:
            while len(begun) < DEPTH:
                nxt = next(chunks, None)
                if nxt is None:
                    return
                t0 = perf_counter()
                reader.read_jpeg_paths_begin(nxt)
                local['s_begin'] += perf_counter() - t0
                begun.append(nxt)

        chunk = next(chunks, None)
        while chunk is not None:
            if reader is None:
                reader = _acquire_reader(params, device) if MeterReader is _REAL_READER else MeterReader(params, device=device)
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
                    t0 = perf_counter()
                    raw = reader.read_jpeg_paths_end()
                    local['s_end_wait'] += perf_counter() - t0
                    raw = (raw[0], raw[1].tolist())
                else:
                    raw = _gpu_read(chunk)
            # Look-ahead: before this chunk's records are touched the library gets DEPTH chunks to work on -- but only once the
            # chunks have reached full size.  While they ramp up (a short chunk: the start of the list, or its end) nothing is pulled
            # from `filenames` until the chunk's results have been handed out, so a slow iterator sees its first results at once.
            ahead = pipelined and (len(chunk) >= batch or bool(begun))
            if ahead:
                _begin_more()
            following = begun[0] if begun else None
            items = None  # the chunk's result objects, when every file went through the GPU decoder
            t0 = perf_counter()
            if isinstance(raw, tuple):
                if not _debug.DEBUG and all(raw[1]):
                    items = (_LazyItems(raw[0], raw[1], reader.dial_names, chunk) if lazy else
                             records_to_items(raw[0], raw[1], reader.dial_names, chunk, MeterImageData))
                else:
                    converted = records_to_python(raw[0], raw[1], reader.dial_names, chunk)
            elif raw is not None:
                converted = raw
            local['s_convert'] += perf_counter() - t0
            local['chunks'] += 1
            local['files'] += len(chunk)
            if items is not None:
                yield items
                chunk = following if following is not None else next(chunks, None)
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
            out: List[MeterImageData] = []
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
                out.append(MeterImageData(filename, meter_values.get('value'), error, meter_values))
            yield out
            chunk = following if following is not None else next(chunks, None)
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
        lib_stats = None
        if reader is not None and clean and hasattr(getattr(reader, 'ctx', None), 'files_stats'):
            try:
                lib_stats = reader.ctx.files_stats(reset=True)
            except Exception:
                lib_stats = None
        with _stats_lock:
            for k in ('chunks', 'files', 's_begin', 's_end_wait', 's_convert'):
                _stats[k] += local[k]
            if lib_stats is not None and lib_stats.get('calls'):
                _stats['library'].append(lib_stats)
        if reader is not None:
            if clean and MeterReader is _REAL_READER:
                _release_reader(reader)   # idle and in a known state: the next call with this calibration takes it over
            else:
                reader.close()


def get_meter_values(params_file: str, filenames: Iterable[str]) -> Iterator[MeterImageData]:
    global _devices_used
    params = _params.load(params_file)
    gpu_decode = os.getenv('METERELF_DECODE', 'gpu') != 'host'
    batch = max(1, int(os.getenv('METERELF_BATCH', '1024' if gpu_decode else '64')))
    devices = _device_list()
    if _debug.DEBUG:
        batch = 1  # DEBUG re-raises at the failing file, before any later file is touched
        devices = devices[:1]
    # METERELF_BATCH_FIRST: size of the first chunk (default 32; 0 = every chunk METERELF_BATCH names); the sizes double up to METERELF_BATCH
    chunks = _chunks(filenames, batch, max(0, int(os.getenv('METERELF_BATCH_FIRST', '32'))))
    if len(devices) == 1:
        for items in _process_chunks(params, chunks, devices[0], gpu_decode, batch):
            yield from items
        return
    _devices_used = max(_devices_used, len(set(devices)))
    yield from _fan_out(params, chunks, devices, gpu_decode, batch)


_STOP = object()

# The interpreter's thread switch interval is process-wide: overlapping fan-outs share ONE override, counted under a lock,
# and the value found by the first one is put back when the last one ends (METERELF_SWITCH_INTERVAL=0: never touched).
_switch_lock = threading.Lock()
_switch_users = 0
_switch_saved = 0.0


def _switch_interval_enter() -> bool:
    global _switch_users, _switch_saved
    import sys
    want = float(os.getenv('METERELF_SWITCH_INTERVAL', '0.0005'))
    if want <= 0:
        return False
    with _switch_lock:
        if _switch_users == 0:
            _switch_saved = sys.getswitchinterval()
            if want < _switch_saved:
                sys.setswitchinterval(want)
        _switch_users += 1
    return True


def _switch_interval_exit() -> None:
    global _switch_users
    import sys
    with _switch_lock:
        _switch_users -= 1
        if _switch_users == 0 and sys.getswitchinterval() != _switch_saved:
            sys.setswitchinterval(_switch_saved)


def _fan_out(params, chunks: Iterator[List[str]], devices: List[int], gpu_decode: bool, batch: int) -> Iterator[MeterImageData]:
    """Chunk k goes to worker k mod D (one per entry of `devices`: its own thread, MeterReader and begin / end pipeline); the
    workers' chunk results come back through per-worker queues and are yielded in input order.  Every worker may hold
    AHEAD chunks (those its library works on, the one being converted, one waiting), so the consumer is at most
    D x AHEAD chunks behind the file list.  A worker's exception is re-raised here, at the position of its chunk."""
    AHEAD = _hip.FILES_IN_FLIGHT_MAX + 2
    nw = len(devices)
    inq = [queue.Queue() for _ in range(nw)]
    outq = [queue.Queue() for _ in range(nw)]

    def work(w: int) -> None:
        def feed() -> Iterator[List[str]]:
            while True:
                item = inq[w].get()
                if item is _STOP:
                    return
                yield item
        gen = _process_chunks(params, feed(), devices[w], gpu_decode, batch, reader=pre.get(w), lazy=True)
        try:
            for items in gen:
                outq[w].put(items)
        except BaseException as e:     # handed to the consumer thread
            outq[w].put(e)
        finally:
            gen.close()
            outq[w].put(_STOP)

    pre = _acquire_readers_bcast(params, devices) if MeterReader is _REAL_READER else {}
    threads = [threading.Thread(target=work, args=(w,), name='meterelf-dev%d-%d' % (devices[w], w), daemon=True) for w in range(nw)]
    # The workers and this thread hand the interpreter lock to each other once per chunk (a worker converts a chunk's records,
    # this thread yields them); a thread that wants the lock while another runs Python code waits up to one switch interval --
    # 5 ms by default, several chunks' worth.  A shorter interval for as long as any fan-out runs (counted: overlapping
    # generators share one override, the last one out restores; METERELF_SWITCH_INTERVAL=seconds, 0 = leave it alone).
    switched = _switch_interval_enter()
    for t in threads:
        t.start()
    dealt = 0        # chunks handed out
    taken = 0        # chunks yielded
    exhausted = False
    try:
        while True:
            while not exhausted and dealt - taken < nw * AHEAD:
                chunk = next(chunks, None)
                if chunk is None:
                    exhausted = True
                    for q in inq:
                        q.put(_STOP)
                    break
                inq[dealt % nw].put(chunk)
                dealt += 1
            if taken == dealt:
                break
            items = outq[taken % nw].get()
            if isinstance(items, BaseException):
                raise items
            if items is _STOP:      # a worker ended early without an exception: cannot happen
                raise RuntimeError('meterelf_amd: a device worker stopped before its chunks were done')
            taken += 1
            if isinstance(items, _LazyItems):
                items = items.materialize()
            yield from items
    finally:
        if not exhausted:
            for q in inq:        # a consumer that stopped early (or an error): chunks not yet started are dropped
                try:
                    while True:
                        q.get_nowait()
                except queue.Empty:
                    pass
                q.put(_STOP)
        # a consumer that stopped early: the workers finish the chunk they are in (their pipelines are collected by
        # _process_chunks' own clean-up) and end; nothing of theirs is yielded any more
        for t in threads:
            t.join()
        if switched:
            _switch_interval_exit()
