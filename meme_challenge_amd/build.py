"""Build libuniter_hip.so (HIP kernels + C ABI).  hipcc cross-compiles gfx950
without a GPU present.

    python -m meme_challenge_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get('UNITER_CSRC_DIR') or os.path.join(PKG, 'csrc')     # override: variant builds of another source tree
OBJ = os.path.join(PKG, 'build')
LIB = os.path.join(PKG, 'libuniter_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall',
         '-Wno-unused-function', '-ffp-contract=off'] + os.environ.get('UNITER_EXTRA_HIPCC_FLAGS', '').split()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith('.hip') or f.endswith('.cpp'))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(PKG), 'include', 'uniter_hip.h'))
    return sorted(hs)


def _digest(paths):
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, hdr_digest, force, objdir=None):
    objdir = objdir or OBJ
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, os.path.basename(src) + '.o')
    stamp = obj + '.sha'
    dig = _digest([src]) + hdr_digest
    if (not force and os.path.exists(obj) and os.path.exists(stamp)
            and open(stamp).read() == dig):
        return obj, False
    cmd = [HIPCC] + FLAGS + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    with open(stamp, 'w') as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=True, variant=None):
    """variant: name of an experimental build (A/B kernel measurements): objects go to
    build/<variant>/, the library to libuniter_hip_<variant>.so, loaded when UNITER_LIB_VARIANT
    names it (see _lib.py).  Compile flags come from UNITER_EXTRA_HIPCC_FLAGS as usual."""
    srcs = _sources()
    hd = _digest(_headers())
    objdir = os.path.join(OBJ, variant) if variant else OBJ
    LIB = os.path.join(PKG, 'libuniter_hip_%s.so' % variant) if variant else globals()['LIB']
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, hd, force, objdir), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--no-undefined', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date:', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
