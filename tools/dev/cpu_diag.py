"""CPU baseline diagnostics on the GPU box: core counts, affinity, cgroup quota, torch-CPU W48 forward time vs thread count."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import hrnet_ref as hr
try:
    import psutil
    print('physical', psutil.cpu_count(logical=False), 'logical', psutil.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
except Exception as e:
    print('psutil', e)
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    if os.path.exists(f):
        print(f, open(f).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|MHz' | head -8")
cfg = hr.load_config('hrnet_w48')
sd = bench.seeded_weights('hrnet_w48', 1)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    for b in (1, 8):
        x = torch.rand((b, 3, 540, 960))
        with torch.no_grad():
            hr.forward(sd, x, cfg)
            t0 = time.time(); hr.forward(sd, x, cfg); dt = time.time() - t0
        print(f'threads {nt:3d} batch {b}: {dt / b:.3f} s/frame', flush=True)
