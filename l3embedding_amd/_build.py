"""Builds libl3hip.so (hipcc, gfx950) in-tree.  Called by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libl3hip.so')
SOURCES = ['conv.hip', 'conv_wino.hip', 'conv_bf16.hip', 'elementwise.hip', 'bn_fused.hip', 'frontend.hip', 'engine.hip', 'ops.hip']


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, '..', 'include', 'l3hip.h'))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC -> l3embedding_amd/lib/libl3hip.so"""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIBPATH]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIBPATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
