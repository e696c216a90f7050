"""Builds libspeecht_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libspeecht_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def sources():
  return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


HASH_PATH = os.path.join(LIB_DIR, 'libspeecht_hip.sha256')


def source_digest():
  """sha256 over the flags and the bytes of every source/header the library is built from.  The shipped .so is
  reused only when the digest recorded next to it at link time matches -- file times say nothing after a copy."""
  import hashlib
  h = hashlib.sha256(' '.join(FLAGS).replace(ROOT, '').encode())
  deps = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
  inc = os.path.join(ROOT, 'include')
  deps += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h'))
  for d in deps:
    h.update(os.path.basename(d).encode())
    with open(d, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def _stale():
  if not (os.path.exists(LIB_PATH) and os.path.exists(HASH_PATH)):
    return True
  with open(HASH_PATH) as f:
    return f.read().strip() != source_digest()


def build_library(force=False, verbose=True):
  """Compile every .hip under csrc/ to objects (in parallel) and link the shared library."""
  if not force and not _stale():
    return LIB_PATH
  os.makedirs(LIB_DIR, exist_ok=True)
  obj_dir = os.path.join(LIB_DIR, 'obj')
  os.makedirs(obj_dir, exist_ok=True)
  procs, objs = [], []
  for src in sources():
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + '.o')
    objs.append(obj)
    cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
    if verbose:
      print(' '.join(cmd), file=sys.stderr)
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
  for src, p in procs:
    out, _ = p.communicate()
    if p.returncode != 0:
      raise RuntimeError('hipcc failed on {}:\n{}'.format(src, out.decode()))
  cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
  if verbose:
    print(' '.join(cmd), file=sys.stderr)
  subprocess.check_call(cmd)
  with open(HASH_PATH, 'w') as f:
    f.write(source_digest() + '\n')
  return LIB_PATH


if __name__ == '__main__':
  print(build_library(force='--force' in sys.argv))
