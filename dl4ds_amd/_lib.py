"""ctypes binding of libdl4ds_hip.so (the gfx950 HIP library; C ABI in include/dl4ds_hip.h).

There is NO CPU fallback: importing this module without the built library, or calling into it
without a visible MI355X, raises.  Prototypes are parsed from the header so the binding cannot
drift from the ABI.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# DL4DS_HIP_LIB: alternative build of the same library (kernel-variant experiments, tools/variant_build.sh)
LIB_PATH = os.environ.get('DL4DS_HIP_LIB') or os.path.join(_HERE, 'libdl4ds_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dl4ds_hip.h')


class Dl4dsHipError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int)\s+(dl4ds_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        at = []
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a or '[' in a:
                    at.append(ctypes.c_void_p)
                elif 'size_t' in a:
                    at.append(ctypes.c_size_t)
                elif re.search(r'\bdouble\b', a):
                    at.append(ctypes.c_double)
                elif re.search(r'\bfloat\b', a):
                    at.append(ctypes.c_float)
                elif re.search(r'\blong\b', a):
                    at.append(ctypes.c_long)
                else:
                    at.append(ctypes.c_int)
        protos[name] = (ctypes.c_char_p if 'char' in ret else ctypes.c_int, at)
    return protos


_lib = None
_inited = False


def load():
    """dlopen the library (no GPU needed) and attach prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Dl4dsHipError(
            f'{LIB_PATH} is missing: build it with `python dl4ds_amd/csrc/build.py` '
            '(or __graft_entry__.build()).  dl4ds_amd has no CPU fallback.')
    # torch (plumbing: torch.distributed rendezvous, CPU oracle in tests) ships its own ROCm runtime with the
    # same SONAMEs; import it first so one HIP/RCCL runtime serves the whole process.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (ret, at) in parse_header().items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = ret
        fn.argtypes = at
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().dl4ds_last_error()
        raise Dl4dsHipError(msg.decode() if msg else f'libdl4ds_hip call failed with status {status}')


def lib():
    """Loaded library with the device initialised (raises if no GPU is visible)."""
    global _inited
    l = load()
    if not _inited:
        n = ctypes.c_int(0)
        st = l.dl4ds_device_count(ctypes.byref(n))
        if st != 0 or n.value < 1:
            raise Dl4dsHipError('no HIP device visible: dl4ds_amd runs on MI355X (gfx950) only, '
                                'there is no CPU fallback')
        dev = int(os.environ.get('LOCAL_RANK', '0')) % n.value
        check(l.dl4ds_init(dev))
        _inited = True
    return l


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(lib().dl4ds_device_name(buf, 256))
    return buf.value.decode()
