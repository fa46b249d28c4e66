"""Step time of the full pipeline with and without per-kernel event profiling (each hipEventRecord is a
barrier packet in the queue): why bench.py times with events around the dominant kernel only."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from meterelf_amd import _engine, _hip, _params
from meterelf_amd._image import imread_bgr
pfile = os.path.join(ROOT, 'tests', 'golden', 'sample-images1', 'params.yml')
ctx = _hip.Context(_engine.make_blob(_params.load(pfile)), 0)
dev = torch.device('cuda', 0)
files = [f for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg'))) if os.path.basename(f) not in bench.REJECTED]
imgs = [imread_bgr(f) for f in files]
base = np.stack([im for im in imgs if im.shape == imgs[-1].shape])
B = 1024
frames = bench.synth_frames_gpu(torch, torch.from_numpy(base).to(dev), B, 2024, dev)
(H, W) = base.shape[1:3]
stream = torch.cuda.current_stream().cuda_stream
d_results = torch.empty(B * _hip.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    for _ in range(5):
        ctx.process_batch_dev(frames.data_ptr(), B, H, W, d_results_ptr=d_results.data_ptr(), want_host=False, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.process_batch_dev(frames.data_ptr(), B, H, W, d_results_ptr=d_results.data_ptr(), want_host=False, stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    ctx.timings()
    print('profiling %-5s: %.4f ms/step' % (prof, dt * 1e3))
