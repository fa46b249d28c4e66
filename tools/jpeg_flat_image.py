"""k_jpeg_huff on a flat (dark) frame against a normal one (GPU box): blocks of "DC difference 0 + end of block" only never
throw a wrongly-phased decoder out of step, so the truth crawls one segment per synchronisation round.
    python3 tools/jpeg_flat_image.py"""
import glob, os, sys, time
sys.path.insert(0, os.getcwd())
from meterelf_amd import MeterReader, _hip, _params
d='tests/golden/sample-images1'
reader = MeterReader(_params.load(os.path.join(d, 'params.yml')))
files=sorted(glob.glob(os.path.join(d,'*.jpg')))
dark=open(os.path.join(d,'20180814021309-01-e01.jpg'),'rb').read()
norm=open(files[5],'rb').read()
for (name, blob, mix) in (('normal', norm, None), ('dark', dark, None), ('511 normal + 1 dark', norm, dark)):
    (H,W,ok,_)=_hip.jpeg_probe(blob)
    batch=[blob]*512
    if mix is not None and _hip.jpeg_probe(mix)[:2]==(H,W): batch[100]=mix
    elif mix is not None:
        print('dark file has another size', _hip.jpeg_probe(mix)[:2], (H,W)); continue
    reader.ctx.jpeg_process_batch(batch,H,W); reader.ctx.jpeg_process_batch(batch,H,W)
    t0=time.perf_counter()
    for _ in range(5): (r,s)=reader.ctx.jpeg_process_batch(batch,H,W)
    dt=(time.perf_counter()-t0)/5
    reader.ctx.set_profiling(1); reader.ctx.timings(); reader.ctx.jpeg_process_batch(batch,H,W); t=reader.ctx.timings(); reader.ctx.set_profiling(0)
    print('%-22s %dx%d: %.2f ms per 512-file call, k_jpeg_huff %.3f ms, status ok %d' % (name, W, H, dt*1e3, t['k_jpeg_huff'][0]/max(t['k_jpeg_huff'][1],1), int((s==0).sum())))

# a flat frame of the NORMAL frames' size among 1023 normal ones: the pipelined call decodes the files with very few bits
# per block in a first chunk of their own (MELF_JPEG_NO_REORDER=1: where they happen to be)
import io
import numpy as np
from PIL import Image
(H, W, ok, _) = _hip.jpeg_probe(norm)
buf = io.BytesIO(); Image.fromarray(np.full((H, W, 3), 9, np.uint8)).save(buf, 'JPEG', quality=90)
normal = [open(f, 'rb').read() for f in files]
normal = [b for b in normal if _hip.jpeg_probe(b)[:3] == (H, W, True)]
for (label, nflat) in (('1024 normal', 0), ('1021 normal + 3 flat', 3), ('960 normal + 64 flat', 64)):
    batch = [normal[i % len(normal)] for i in range(1024)]
    for k in range(nflat): batch[(k * 331 + 100) % 1024] = buf.getvalue()
    for env in ('', '1'):
        if env: os.environ['MELF_JPEG_NO_REORDER'] = '1'
        else: os.environ.pop('MELF_JPEG_NO_REORDER', None)
        reader.ctx.jpeg_process_batch(batch, H, W); reader.ctx.jpeg_process_batch(batch, H, W)
        t0 = time.perf_counter()
        for _ in range(5): (r, s_) = reader.ctx.jpeg_process_batch(batch, H, W)
        dt = (time.perf_counter() - t0) / 5
        print('%-24s %-28s %.2f ms per 1024-file call' % (label, 'in place (no reorder)' if env else 'flat files first', dt * 1e3))
        if not nflat: break
os.environ.pop('MELF_JPEG_NO_REORDER', None)
