import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['DL4DS_TEST_HOOKS'] = '1'
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def run(off):
    if off: os.environ['DL4DS_NO_TWO_ADD_INPLACE'] = '1'
    else: os.environ.pop('DL4DS_NO_TWO_ADD_INPLACE', None)
    from dl4ds_amd.graph import GraphBuilder, Model
    from dl4ds_amd.training import SupervisedEngine
    from dl4ds_amd.models.blocks import residual_block
    g = GraphBuilder()
    x = g.input(H, H, 3)
    r = b = g.conv2d(x, 'in', 8, 3)
    for i in range(2):
        b = residual_block(g, f'rb{i}', b, 8)
    y = g.conv2d(b, 'out', 8, 3, add=r)
    y = g.conv2d(y, 'last', 1, 3)
    g.finalize(y, seed=1)
    m = Model(g, 'dbg', [(H, H, 3)])
    rng = np.random.default_rng(0)
    xs = rng.standard_normal((4, H, H, 3)).astype(np.float32)
    ys = rng.standard_normal((4, H, H, 1)).astype(np.float32)
    e = SupervisedEngine(m, loss='mse', learning_rate=1e-3)
    l, gr = e.loss_and_grads([xs], ys)
    return l, gr
# separate processes would be cleaner (static env caches): run the two variants in children
if len(sys.argv) > 2:
    l, gr = run(sys.argv[2] == 'off')
    np.savez(f'/tmp/two_add_{sys.argv[2]}.npz', **gr)
    print(sys.argv[2], l)
else:
    import subprocess
    for v in ('on', 'off'):
        subprocess.run([sys.executable, __file__, str(H), v], check=True)
    a, b = np.load('/tmp/two_add_on.npz'), np.load('/tmp/two_add_off.npz')
    for k in a.files:
        print(k, float(np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)))
