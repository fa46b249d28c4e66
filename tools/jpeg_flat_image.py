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
