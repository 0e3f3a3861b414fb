"""Builds libl3hip.so (hipcc, gfx950) in-tree.  Called by __graft_entry__.build().

Every HIP source is compiled to its own object (in parallel, rebuilt only when it or a header changed)
and the objects are linked into l3embedding_amd/lib/libl3hip.so."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIBPATH = os.path.join(LIBDIR, 'libl3hip.so')
# The product library.  One file per kernel family of DESIGN.md section 4: fp32 convolutions (F(4x4,3x3) forward / data gradient,
# F(2x2,3x3) = l3_config.fp32_conv F2X2 + the dispatch, F(3x3,2x2) weight gradient, direct implicit GEMM for the DFT and every
# geometry Winograd does not take), mixed precision (halo forward / data gradient, transpose-read weight gradient, the dispatch +
# fp32-tensor entry points), first layers, BatchNorm / pool, head / loss / Adam, front-end, engine, operator entry points, RCCL.
SOURCES = ['conv.hip', 'conv_wino.hip', 'conv_wino4.hip', 'conv_bf16.hip', 'conv_bf16_halo.hip', 'conv_wgrad_bf16.hip', 'conv_wgrad_wino.hip',
           'conv_first.hip', 'elementwise.hip', 'bn_fused.hip', 'frontend.hip', 'engine.hip', 'ops.hip', 'comm.hip']
# Measured-and-rejected kernel variants (split-bf16 fp32 convolutions, flat-tile MODE 5, tap-split bf16 weight gradient; round 6: the
# filter-in-registers 64-channel halo kernel, the split-bf16 first convolution): records of negative results (profiles/r05_bx6_ablations.txt,
# r05_bf16_conv_notes.txt, r06_halo64_regfilter.txt, r06_first_conv_mfma.txt), NOT product paths.  L3_BUILD_EXPERIMENTS=1 compiles them
# in (-DL3_EXPERIMENTS; l3_build_experiments() == 1) and their tests run; the default library does not carry them.
EXPERIMENT_SOURCES = ['conv_wino_bx6.hip', 'conv_wgrad_bx6.hip']
EXPERIMENTS = os.environ.get('L3_BUILD_EXPERIMENTS') == '1'
if EXPERIMENTS:
    SOURCES = SOURCES + EXPERIMENT_SOURCES
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DL3_EXPERIMENTS'] if EXPERIMENTS else [])
# conv_wino4.hip: its input transform runs in the gaps between MFMAs, where plain fp32 VALU is cheaper than packed
# conv_first.hip: the weight gradient's 32 / 48 accumulator registers in arch VGPRs (-amdgpu-mfma-vgpr-form): left to choose, the
# compiler kept them half in AGPRs and permuted the whole set through v_accvgpr_read / _write at every loop end (16 of 79 VALU
# instructions per 4-pixel unit, round 6)
FILE_FLAGS = {'conv_wino4.hip': ['-fno-slp-vectorize'], 'conv_wino_bx6.hip': ['-fno-slp-vectorize'], 'conv_wgrad_bx6.hip': ['-fno-slp-vectorize'],
              'conv_first.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', 'include', 'l3hip.h'))
    return [h for h in hs if os.path.exists(h)]


def source_hash(files=None):
    """sha256 (first 16 hex digits) over the HIP sources and headers, or over the named ones: the staleness key of the
    builder-side PMC summaries under profiles/ (bench.py compares it with the tree it runs from)."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(files) if files else sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    for f in names:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers() + [os.path.abspath(__file__)]
    return _stale(LIBPATH, deps)


def _hipcc():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    return hipcc if os.path.exists(hipcc) else 'hipcc'


def build(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 -O3 -c each source, then -shared -> l3embedding_amd/lib/libl3hip.so"""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc, headers = _hipcc(), _headers()
    flags = FLAGS + list(extra_flags)
    stamp = os.path.join(OBJDIR, 'flags.txt')
    stamp_text = ' '.join(flags) + ' | ' + repr(sorted(FILE_FLAGS.items()))
    if not os.path.exists(stamp) or open(stamp).read() != stamp_text:
        force = True

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(OBJDIR, src.replace('.hip', '.o'))
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + flags + FILE_FLAGS.get(src, []) + ['-c', path, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stdout.decode(errors='replace')))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBPATH] + objs + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as fh:
        fh.write(stamp_text)
    return LIBPATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
