"""Build libdl4ds_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python dl4ds_amd/csrc/build.py [--force]

Objects go to dl4ds_amd/csrc/_build/ (git-ignored), the library to dl4ds_amd/libdl4ds_hip.so (in-tree so it
travels to the GPU box).  Only sources newer than their object are recompiled.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
# DL4DS_BUILD_EXPERIMENTS=1: the library with the measured-and-dropped variants' and the diagnostics' run-time switches compiled in
# (common.h: exp_env) -> dl4ds_amd/libdl4ds_hip_exp.so, objects in _build_exp/; load it with DL4DS_HIP_LIB for A/B runs.  The product
# library (the default, what __graft_entry__.build() makes and the tests / bench load) has them compiled out.
EXPERIMENTS = os.environ.get('DL4DS_BUILD_EXPERIMENTS', '') == '1'
OUT = os.path.join(PKG, 'libdl4ds_hip_exp.so' if EXPERIMENTS else 'libdl4ds_hip.so')
OBJ = os.path.join(HERE, '_build_exp' if EXPERIMENTS else '_build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-I/opt/rocm/include', '-x', 'hip'] + (['-DDL4DS_EXPERIMENTS'] if EXPERIMENTS else [])


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(('.hip', '.cpp')))


def headers_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(PKG), 'include', 'dl4ds_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force):
    obj = os.path.join(OBJ, src + '.o')
    sp = os.path.join(HERE, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(sp)
            and os.path.getmtime(obj) > headers_mtime()):
        return obj, None
    cmd = [HIPCC] + FLAGS + ['-c', sp, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return obj, f'{" ".join(cmd)}\n{r.stdout}\n{r.stderr}'
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, None


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: compile_one(s, force), srcs))
    errs = [e for _, e in res if e]
    if errs:
        raise RuntimeError('hipcc failed:\n' + '\n'.join(errs))
    objs = [o for o, _ in res]
    need_link = force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs)
    if need_link:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + \
              ['-L/opt/rocm/lib', '-lrccl', '-Wl,-rpath,/opt/rocm/lib']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
    if verbose:
        print(f'built {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB)')
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
