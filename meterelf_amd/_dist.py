"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

The path shards naturally: frames are independent (reference:
meterelf/_api.py:22-33) and the only shared state is the read-only calibration
(Params, template, dial masks: meterelf/_image.py:69-81, _dial_data.py:11-19).
So there is exactly ONE collective on the set-up path -- rank 0 packs the
calibration blob (~0.2 MB) and broadcasts it -- and none on the data path: each
rank reads a contiguous shard with its own context.  Results can optionally be
all-gathered (fixed-size records) when every rank wants the full list.
"""
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _hip

ProcessFn = Callable[[np.ndarray], np.ndarray]  # (n, H, W, 3) u8 -> RESULT_DTYPE records


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ceil(n / world)-sized shards; trailing ranks may get fewer (or none)."""
    per = (n + world - 1) // world
    start = min(rank * per, n)
    return start, min(start + per, n)


def init_process_group(backend: Optional[str] = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torchrun sets them."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(backend=backend)
    return dist


def broadcast_blob(blob: Optional[np.ndarray], dial_names: Optional[List[str]], src: int = 0, device=None):
    """Rank `src` passes the packed blob and dial names, the others pass None.
    Returns (blob as numpy uint8 on the host, device tensor or None, dial names).

    With the nccl backend the payload travels GPU-to-GPU (RCCL broadcast over
    xGMI) and the returned device tensor can seed melf_ctx_create directly
    (blob_on_device=1)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    use_cuda = dist.get_backend() == 'nccl'
    dev = device if device is not None else (torch.device('cuda', torch.cuda.current_device()) if use_cuda else torch.device('cpu'))
    meta = [int(blob.nbytes), list(dial_names)] if rank == src else [None, None]
    dist.broadcast_object_list(meta, src=src)
    (nbytes, names) = meta
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)).to(dev)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    host = t.cpu().numpy().copy()
    return host, (t if use_cuda else None), names


def all_gather_records(local: np.ndarray, counts: Sequence[int], device=None) -> np.ndarray:
    """Concatenates every rank's records (rank order = frame order).  Fixed-size
    byte records; ranks pad to the largest shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    use_cuda = dist.get_backend() == 'nccl'
    dev = device if device is not None else (torch.device('cuda', torch.cuda.current_device()) if use_cuda else torch.device('cpu'))
    per = max(counts) if len(counts) else 0
    item = _hip.RESULT_DTYPE.itemsize
    buf = np.zeros(per * item, np.uint8)
    raw = local.view(np.uint8).reshape(-1)
    buf[:raw.size] = raw
    mine = torch.from_numpy(buf).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    if per:
        dist.all_gather(gathered, mine)
    parts = []
    for (r, g) in enumerate(gathered):
        a = g.cpu().numpy()[:counts[r] * item]
        parts.append(a.view(_hip.RESULT_DTYPE))
    return np.concatenate(parts) if parts else np.zeros(0, _hip.RESULT_DTYPE)


class ShardedMeterReader:
    """Data-parallel reader.  Construct on every rank after init_process_group().

    params_file is read by rank 0 only; the other ranks build their context from
    the broadcast blob.  `process_factory(blob, names)` may replace the GPU
    context (CPU tests inject a checker there); by default it is the HIP context
    and fails loudly without a GPU.
    """

    def __init__(self, params_file: Optional[str] = None, process_factory=None, src: int = 0) -> None:
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        (blob, names) = (None, None)
        if self.rank == src:
            from . import _engine, _params
            params = _params.load(params_file)
            blob = _engine.make_blob(params)
            names = params.dial_names
        (self.blob, dev_tensor, self.dial_names) = broadcast_blob(blob, names, src=src)
        self._ctx = None
        if process_factory is not None:
            self._process: ProcessFn = process_factory(self.blob, self.dial_names)
        else:
            if dev_tensor is not None:  # RCCL path: the blob already sits on this rank's GPU
                self._ctx = _hip.Context(self.blob, dev_tensor.device.index, blob_device_ptr=dev_tensor.data_ptr())
            else:  # host collective (gloo): no torch.cuda involved, the library picks the rank's GPU itself
                self._ctx = _hip.Context(self.blob, int(os.environ.get('LOCAL_RANK', '0')))
            self._process = self._ctx.process_batch

    @property
    def ctx(self):
        return self._ctx

    def close(self) -> None:
        if self._ctx is not None:
            self._ctx.close()

    def my_range(self, n: int) -> Tuple[int, int]:
        return shard_range(n, self.rank, self.world)

    def read_local(self, frames: np.ndarray) -> np.ndarray:
        """This rank's shard only; no communication."""
        return self._process(frames)

    def read_files_local(self, filenames: Sequence[str]) -> np.ndarray:
        """This rank's files -> records.  JPEG files are decoded on the rank's GPU
        (melf_jpeg_process_batch); whatever the GPU decoder does not take, and every file when
        the compute is injected (CPU tests), is decoded on the host.  An unreadable file gives a
        record with status -1."""
        from ._image import imread_bgr
        n = len(filenames)
        out = np.zeros(n, _hip.RESULT_DTYPE)
        todo = list(range(n))
        if self._ctx is not None:  # files are read and decoded inside the library (melf_jpeg_process_files)
            done = set()
            pending = list(range(n))
            while pending:
                (recs, status, _hw) = self._ctx.jpeg_process_files([filenames[i] for i in pending])
                again = []
                for (k, i) in enumerate(pending):
                    if status[k] == _hip.JPEG_OK:
                        out[i] = recs[k]
                        done.add(i)
                    elif status[k] == _hip.JPEG_SIZE_MISMATCH:
                        again.append(i)
                if len(again) == len(pending):
                    break
                pending = again
            todo = [i for i in range(n) if i not in done]
        by_shape = {}
        for i in todo:
            img = imread_bgr(filenames[i])
            if img is None:
                out[i]['status'] = -1
            else:
                by_shape.setdefault(img.shape, []).append((i, img))
        for items in by_shape.values():
            recs = self._process(np.stack([img for (_, img) in items]))
            for ((i, _), r) in zip(items, recs):
                out[i] = r
        return out

    def read_files_global(self, filenames: Sequence[str], gather: bool = True) -> np.ndarray:
        """Every rank gets the same list of file names, reads its contiguous shard of the FILES
        (no rank touches another rank's files) and, if `gather`, receives all records."""
        n = len(filenames)
        (a, b) = self.my_range(n)
        local = self.read_files_local(filenames[a:b]) if b > a else np.zeros(0, _hip.RESULT_DTYPE)
        if not gather:
            return local
        counts = [shard_range(n, r, self.world)[1] - shard_range(n, r, self.world)[0] for r in range(self.world)]
        return all_gather_records(local, counts)

    def read_global(self, frames: np.ndarray, gather: bool = True) -> np.ndarray:
        """Every rank holds the same (n, H, W, 3) array (or a memory map of it),
        reads its contiguous shard and, if `gather`, receives all records."""
        n = len(frames)
        (a, b) = self.my_range(n)
        local = self._process(frames[a:b]) if b > a else np.zeros(0, _hip.RESULT_DTYPE)
        if not gather:
            return local
        counts = [shard_range(n, r, self.world)[1] - shard_range(n, r, self.world)[0] for r in range(self.world)]
        return all_gather_records(local, counts)
