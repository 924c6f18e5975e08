"""Forward-only (model.predict, inference.py:238) throughput of the headline model on one GPU: device-resident inputs and
output, batch 64, + per-kernel breakdown.   python tools/bench_predict.py [batch] [steps]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
import dl4ds_amd.models as PM

lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (128, 128), seed=7)
rng = np.random.default_rng(0)
x = DeviceArray.from_numpy(rng.random((B, 128, 128, 1)).astype(np.float32))
y = DeviceArray.zeros((B, 512, 512, 1))
ptrs = (ctypes.c_void_p * 1)(x.ptr)
fwd = lambda: L.check(lib.dl4ds_graph_forward(model.graph.h, ptrs, 1, B, 0, 0, y.ptr))
for _ in range(3):
    fwd()
L.check(lib.dl4ds_sync())
t0 = time.perf_counter()
for _ in range(steps):
    fwd()
L.check(lib.dl4ds_sync())
dt = (time.perf_counter() - t0) / steps
print(f'predict: batch {B}  {1e3 * dt:.3f} ms/batch  {B / dt:.0f} HR samples/s  (folding {"off" if os.environ.get("DL4DS_NO_FOLD") else "on"})')
L.check(lib.dl4ds_profile_enable(1))
for _ in range(3):
    fwd()
buf = ctypes.create_string_buffer(1 << 16)
L.check(lib.dl4ds_profile_report(buf, len(buf)))
rep = json.loads(buf.value.decode())
tot = sum(v['ms'] for v in rep.values()) / 3
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]['ms'])[:10]:
    tf = (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0.0
    print(f"   {k:30s} n={v['n'] / 3:4.1f} ms={v['ms'] / 3:7.3f} ({100 * v['ms'] / 3 / tot:4.1f} %) {tf:6.1f} TFLOP/s {v['bytes'] / (v['ms'] * 1e-3) / 1e9:7.0f} GB/s")
