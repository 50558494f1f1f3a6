"""Where does the first batch of a process spend its time?  (tools/; run on the GPU box)"""
import io, sys, time, os
sys.path.insert(0, os.getcwd())
os.environ.setdefault('GPU_MAX_HW_QUEUES', '40')
import torch, bench
from badread_amd.engine import HipEngine
wl = bench.build_workload(io.StringIO(), 'human', bench.default_ref_dir())
t0 = time.perf_counter(); e = HipEngine(0, scratch_bytes=40 << 30); bench.configure(e, wl); torch.cuda.synchronize(); print('engine create+configure %.2fs' % (time.perf_counter() - t0))
for n in (64, 64, 4096, 65536, 65536):
    t0 = time.perf_counter(); e.simulate_batch_device(42, 0, n, expected_bytes=n * 36000); torch.cuda.synchronize()
    print('batch of %5d reads: %.2fs   device stages %s' % (n, time.perf_counter() - t0, {k: round(v) for k, v in e.stage_ms().items()}))
