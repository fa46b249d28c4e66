"""N>1 path on CPU: world_size-2 gloo run of meterelf_amd._dist.  Covers the
blob broadcast from rank 0, contiguous sharding and the result all-gather; the
per-shard compute is injected (the oracle, in tests only) because there is no
GPU here -- on the GPU box the same class drives one melf_ctx per rank."""
import glob
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from meterelf_amd import _dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

WORKER = r'''
import os, sys, glob
import numpy as np
sys.path.insert(0, {root!r})
from meterelf_amd import _dist, _hip
from meterelf_amd._image import imread_bgr
from oracle import pyoracle as po
from tests import helpers

dist = _dist.init_process_group('gloo')
rank = dist.get_rank()
pfile = os.path.join({golden!r}, 'sample-images2', 'params.yml')
files = sorted(glob.glob(os.path.join({golden!r}, 'sample-images2', '*.jpg')))[:9]
frames = np.stack([imread_bgr(f) for f in files])

def factory(blob, names):
    # every rank must have received rank 0's bytes: rebuild params from the blob alone
    cp = _hip.blob_params(blob)
    assert (cp.th, cp.tw, cp.ndials) == (119, 188, 4) and names == ['0.0001', '0.001', '0.01', '0.1']
    op = po.Params(pfile)
    return lambda fr: helpers.orc_to_records(po.process_frames(fr, op), len(fr))

reader = _dist.ShardedMeterReader(pfile if rank == 0 else None, process_factory=factory)
(a, b) = reader.my_range(len(frames))
local = reader.read_local(frames[a:b])
allrec = reader.read_global(frames, gather=True)
np.save(os.path.join({out!r}, 'rank%d.npy' % rank), allrec)
byfile = reader.read_files_global(files + ['/nonexistent/file.jpg'], gather=True)   # sharded over FILES
np.save(os.path.join({out!r}, 'files%d.npy' % rank), byfile)
np.save(os.path.join({out!r}, 'blob%d.npy' % rank), reader.blob)
assert np.array_equal(allrec[a:b], local)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range():
    assert [_dist.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [_dist.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert [_dist.shard_range(8192, r, 8) for r in range(8)] == [(1024 * r, 1024 * (r + 1)) for r in range(8)]
    assert _dist.shard_range(0, 0, 2) == (0, 0)


def test_world_size_2_gloo(tmp_path):
    from meterelf_amd._image import imread_bgr
    from oracle import pyoracle as po
    from tests import helpers
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, golden=GOLDEN, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        (out, _) = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-3000:]
    r0 = np.load(tmp_path / 'rank0.npy')
    r1 = np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1) and len(r0) == 9
    assert np.array_equal(np.load(tmp_path / 'blob0.npy'), np.load(tmp_path / 'blob1.npy'))
    pfile = os.path.join(GOLDEN, 'sample-images2', 'params.yml')
    files = sorted(glob.glob(os.path.join(GOLDEN, 'sample-images2', '*.jpg')))[:9]
    frames = np.stack([imread_bgr(f) for f in files])
    single = helpers.orc_to_records(po.process_frames(frames, po.Params(pfile)), 9)
    assert np.array_equal(r0, single)
    f0 = np.load(tmp_path / 'files0.npy')
    assert np.array_equal(f0, np.load(tmp_path / 'files1.npy')) and len(f0) == 10
    assert np.array_equal(f0[:9], single) and f0[9]['status'] == -1
    with open(os.path.join(GOLDEN, 'sample-images2_stdout.txt')) as fp:
        expected = dict(line.split(': ', 1) for line in fp.read().splitlines())
    for (f, r) in zip(files, r0):
        assert '{:07.3f}'.format(float(r['value'])) == expected[os.path.basename(f)]
