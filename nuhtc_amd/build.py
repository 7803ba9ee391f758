"""Builds libnuhtc_hip.so (hand-written HIP for gfx950, no torch linkage) in-tree with hipcc.

    python -m nuhtc_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libnuhtc_hip.so')
# box / IoU arithmetic must follow the reference's separate float32 mul and add steps (no fused multiply-add)
NO_CONTRACT = ('proposals.hip', 'roi.hip')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _flags():
    return ' '.join(f'{k}={os.environ[k]}' for k in sorted(os.environ) if k.startswith('NUHTC_EXTRA_CFLAGS'))


def needs_build():
    if not os.path.exists(LIB):
        return True
    stamp = os.path.join(HERE, 'build', 'flags.txt')       # a library built with other extra flags (a -DNUHTC_DEV build) is rebuilt
    if (open(stamp).read() if os.path.exists(stamp) else '') != _flags():
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'nuhtc_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-c', src, '-o', obj]
        cmd[4:4] = os.environ.get('NUHTC_EXTRA_CFLAGS', '').split()          # dev probes (-D...)
        cmd[4:4] = os.environ.get('NUHTC_EXTRA_CFLAGS_' + os.path.basename(src).split('.')[0].upper(), '').split()   # ... per file
        if os.path.basename(src) in NO_CONTRACT:
            cmd.insert(4, '-ffp-contract=off')
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + out.decode())
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    subprocess.check_call(cmd)
    with open(os.path.join(objdir, 'flags.txt'), 'w') as f:
        f.write(_flags())
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
