"""Train-step throughput + per-kernel breakdown of BASELINE configs[3] (cfg4: recurrent dense + attention + LCB,
resize-conv x4, T=8, 64^2 -> 256^2, one HR aux) and configs[4] (cfg5: U-Net(dc) generator + residual discriminator CGAN
at 512^2, 5 + 1 input channels) at the BASELINE sizes on one GPU.   python tools/bench_configs.py [cfg4|cfg5] [batch]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
import dl4ds_amd.models as PM
from dl4ds_amd.training import SupervisedEngine, CGANEngine

lib = L.lib()
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
rng = np.random.default_rng(0)


def timed(step, B, steps=5, warmup=2, nprof=2):
    for _ in range(warmup):
        step()
    L.check(lib.dl4ds_sync())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    L.check(lib.dl4ds_sync())
    dt = (time.perf_counter() - t0) / steps
    L.check(lib.dl4ds_profile_enable(1))
    for _ in range(nprof):
        step()
    buf = ctypes.create_string_buffer(1 << 17)
    L.check(lib.dl4ds_profile_report(buf, len(buf)))
    L.check(lib.dl4ds_profile_enable(0))
    rep = json.loads(buf.value.decode())
    tot = sum(v['ms'] for v in rep.values()) / nprof
    print(f'{which}: batch {B}  {1e3 * dt:.2f} ms/step  {B / dt:.1f} samples/s   (sum of kernels {tot:.2f} ms)')
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]['ms'])[:18]:
        tf = (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0.0
        print(f"   {k:34s} n/step={v['n'] / nprof:6.1f} ms/step={v['ms'] / nprof:8.3f} ({100 * v['ms'] / nprof / tot:4.1f} %)"
              f"  {tf:6.1f} TFLOP/s {v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f} GB/s")


if which == 'cfg4':
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    T, h, s = 8, 64, 4
    model = PM.recnet_postupsampling('densenet', 'rc', s, 1, 1, (h, h), time_window=T, attention=True, localcon_layer=True,
                                     seed=3)
    print('params', model.count_params())
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    x = DeviceArray.from_numpy(rng.standard_normal((B, T, h, h, 1)).astype(np.float32))
    aux = DeviceArray.from_numpy(rng.standard_normal((B, h * s, h * s, 1)).astype(np.float32))
    y = DeviceArray.from_numpy(rng.standard_normal((B, T, h * s, h * s, 1)).astype(np.float32))
    timed(lambda: eng.step_device([x.ptr, aux.ptr], y.ptr, B), B)
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    H = 512
    gen = PM.unet_pin('unet', 5, 1, hr_size=(H, H), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=3)
    disc = PM.residual_discriminator(5, 'pin', False, 8, (H // 8, H // 8), n_filters=8, hr_size=(H, H), seed=4)
    print('params G', gen.count_params(), 'D', disc.count_params())
    eng = CGANEngine(gen, disc, loss='mae')
    lr = DeviceArray.from_numpy(rng.standard_normal((B, H, H, 5)).astype(np.float32))
    st = DeviceArray.from_numpy(rng.standard_normal((B, H, H, 1)).astype(np.float32))
    hr = DeviceArray.from_numpy(rng.standard_normal((B, H, H, 1)).astype(np.float32))
    timed(lambda: eng.step_device([lr.ptr, st.ptr], hr.ptr, B), B)
