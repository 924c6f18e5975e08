"""profiles/traffic.json from the round's PMC passes: {config: {bench tag: HBM bytes per CALL}, '_round': tag}.

    python tools/build_traffic_json.py r03 [dir = gpurun_out/prof_r03]

bench.py's roofline block looks its dominant kernel up here (keyed by config, then by the library profiler's tag).  A tag of
the library profiler is one CALL of a layer; the Winograd layers with more than 48 input channels run as several kernel
launches per call (and one small filter-transform launch), so per-launch PMC averages are scaled by launches / calls, both
counted per step (PMC pass: --steps 3 --warmup 2; calls: the bench line's breakdown is not needed -- the kernel-trace
statistics of the same command give launches per step for the tag's kernels, the bench line its calls per step)."""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'prof_' + tag) if len(sys.argv) < 3 else sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
from bench import csrc_sha
try:
    commit = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True,
                            cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip()
    dirty = bool(subprocess.run(['git', 'status', '--porcelain', 'dl4ds_amd/csrc', 'include'], capture_output=True, text=True,
                                cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip())
except Exception:
    commit, dirty = None, None
# the fingerprint of the kernel sources the PMC passes ran on: the collection script writes it next to the counters
# (gpurun_out/prof_<tag>/csrc_sha.txt); bench.py reports `traffic` only when its own build has the same one
shaf = os.path.join(root, 'csrc_sha.txt')
sha = open(shaf).read().strip() if os.path.exists(shaf) else csrc_sha()
out = {'_round': tag, '_csrc_sha': sha, '_commit': (commit or '?') + ('+uncommitted kernel changes' if dirty else ''), '_unit': 'HBM bytes per call of the tagged layer kernel(s): (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction'}
PMC_STEPS = 5
for cfg in ('cfg2', 'cfg4', 'cfg5'):
    f = os.path.join(root, f'pmc_traffic_{cfg}_{tag}.json')
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    sfx = '' if cfg == 'cfg2' else '_' + cfg
    line = json.loads(open(os.path.join(root, f'bench_line{sfx}_{tag}.json')).read().strip().splitlines()[-1])
    dom = line['roofline']['kernel']
    rf = line['roofline']
    calls_per_step = rf.get('calls_per_step') or rf['launches'] / line['steps']
    # steps the PMC pass actually ran (bench.py adds a steady-state window of >= 2 s): the loss kernel launches once per step
    pmc_steps = next((v['launches'] for k, v in t.items() if 'pixel_loss' in k), PMC_STEPS)
    # (files written by an older tools/pmc_traffic.py carry the split kernel's epilogue forms apart: one bench tag, weighted by launches)
    forms = {k: v for k, v in t.items() if k.startswith('conv_split<') and k != 'conv_split<3,3>'}
    if forms:
        n = sum(v['launches'] for v in forms.values())
        t['conv_split<3,3>'] = {'launches': n, 'hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in forms.values()) / n}
    # calls per step of EVERY tag from the library profiler's breakdown of the same command (profiles/bench_breakdown[_cfg]_<tag>.json,
    # DL4DS_BENCH_BREAKDOWN=1), where there is one: a tag's traffic is per CALL of the layer whichever tag a run finds dominant (round 6:
    # cfg2's two leading tags, conv_wino<3,3> and conv_split<3,3>, are within 3 % of each other and take turns)
    bfile = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', f'bench_breakdown{sfx}_{tag}.json')
    calls = {}
    if os.path.exists(bfile):
        bl = json.loads(open(bfile).read().strip().splitlines()[-1])
        calls = {k: v['launches_per_step'] for k, v in (bl.get('breakdown') or {}).items()}
    d = {}
    for k, v in t.items():
        if not re.match(r'^[a-z_0-9]+(<[0-9,]*>)?$', k):
            continue
        per_call = v['hbm_bytes_per_launch']
        if k in calls and calls[k] > 0:
            per_call *= (v['launches'] / pmc_steps) / calls[k]
        elif k == dom:
            per_call *= (v['launches'] / pmc_steps) / calls_per_step
        d[k] = per_call
    out[cfg] = d
json.dump(out, open(os.path.join(os.path.dirname(root.rstrip('/')), '..', 'profiles', 'traffic.json') if False else
                    os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'traffic.json'), 'w'), indent=1)
print({c: len(v) for c, v in out.items() if isinstance(v, dict)})
