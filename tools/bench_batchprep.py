"""Batch preparation for the headline workload (HR 512^2 fields -> LR 128^2 block means + HR targets, batch 64):
the reference-style host loop (numpy port) vs the device gather kernels.
   python tools/bench_batchprep.py [interpolation] [upsampling]      e.g.  bicubic pin"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.dataloader import DataGenerator, DeviceDataGenerator

lib = L.lib()
N, H, B, S = 256, 512, 64, 4
hr = np.random.default_rng(0).random((N, H, H, 1)).astype(np.float32)
INTERP = sys.argv[1] if len(sys.argv) > 1 else 'inter_area'
UPS = sys.argv[2] if len(sys.argv) > 2 else 'spc'
kw = dict(backbone='resnet', upsampling=UPS, scale=S, batch_size=B, seed=1, interpolation=INTERP)
print('interpolation', INTERP, 'upsampling', UPS)
host, dev = DataGenerator(hr, None, **kw), DeviceDataGenerator(hr, None, **kw)
t0 = time.perf_counter()
for i in range(len(host)):
    host[i]
t_host = (time.perf_counter() - t0) / len(host)
for i in range(len(dev)):
    dev[i]
L.check(lib.dl4ds_sync())
t0 = time.perf_counter()
reps = 10
for r in range(reps):
    for i in range(len(dev)):
        dev[i]
L.check(lib.dl4ds_sync())
t_dev = (time.perf_counter() - t0) / (reps * len(dev))
L.check(lib.dl4ds_profile_enable(1))
for i in range(len(dev)):
    dev[i]
buf = ctypes.create_string_buffer(1 << 14)
L.check(lib.dl4ds_profile_report(buf, len(buf)))
rep = json.loads(buf.value.decode())
print(f'host numpy loop : {1e3 * t_host:8.2f} ms/batch  {B / t_host:9.0f} samples/s')
print(f'device gather   : {1e3 * t_dev:8.3f} ms/batch  {B / t_dev:9.0f} samples/s   ({t_host / t_dev:.0f}x)')
for k, v in rep.items():
    print(f"   {k:20s} {v['ms'] / v['n'] * 1e3:8.1f} us/launch  {v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f} GB/s algorithmic")
