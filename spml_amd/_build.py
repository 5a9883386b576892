"""Builds libspml_hip.so (gfx950) from spml_amd/csrc/*.hip with hipcc, in-tree.

`python -m spml_amd._build` or `spml_amd._build.build()`; a no-op when the
library is newer than every source.  hipcc cross-compiles without a GPU."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libspml_hip.so')
OBJ_DIR = os.path.join(HERE, 'build')

# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs (gfx90a+ unified register file);
# without it the epilogues pay one v_accvgpr_read/write per accumulator element.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE,
         '-Wno-unused-result', '-Wno-pass-failed', '-Wno-inline-asm', '-mllvm', '-amdgpu-mfma-vgpr-form']


def _hipcc():
  for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (needed to build libspml_hip.so)')


def sources():
  return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _deps():
  return sources() + glob.glob(os.path.join(CSRC, '*.hpp')) + glob.glob(os.path.join(CSRC, '*.inc')) + \
      glob.glob(os.path.join(INCLUDE, '*.h'))


# Sources built WITHOUT -amdgpu-mfma-vgpr-form: kernels that use the whole 512-entry register file (one wave
# per SIMD) need their accumulators in the accumulation half, or the allocator spills operands through it.
AGPR_FORM = set()


STAMP = os.path.join(OBJ_DIR, 'build_flags.txt')


def _experiment_flags():
  """The -D set of the profiling variants (they overwrite outputs and skip MFMAs: never a product build).  Recorded
  next to the objects and inside the library (spml_build_experiment): a library built with any of them is rebuilt by the
  next plain build and refused by spml_amd._ffi unless the same variable is still set."""
  extra = ['-DSPML_TRACE'] if os.environ.get('SPML_TRACE') else []
  if os.environ.get('SPML_CONV_EXP'):                    # ... of conv.hip
    extra.append('-DSPML_CONV_EXP=' + os.environ['SPML_CONV_EXP'])
  if os.environ.get('SPML_P64_EXP'):                     # experiment switches of kmeans64.hip (profiling builds)
    extra.append('-DSPML_P64_EXP=' + os.environ['SPML_P64_EXP'])
  code = (int(os.environ.get('SPML_CONV_EXP') or 0) & 0xffff) | ((int(os.environ.get('SPML_P64_EXP') or 0) & 0xffff) << 16)
  return extra, code


def is_fresh():
  if not os.path.exists(LIB_PATH):
    return False
  stamp = open(STAMP).read() if os.path.exists(STAMP) else ''
  if stamp != ' '.join(_experiment_flags()[0]):
    return False
  t = os.path.getmtime(LIB_PATH)
  return all(os.path.getmtime(s) <= t for s in _deps())


def build(force=False, verbose=True):
  """Compile every .hip for gfx950 and link the shared library."""
  if not force and is_fresh():
    return LIB_PATH
  extra, exp_code = _experiment_flags()
  stamp = open(STAMP).read() if os.path.exists(STAMP) else ''
  if stamp != ' '.join(extra):
    force = True                                           # objects of another -D set: none of them is reusable
  hipcc = _hipcc()
  os.makedirs(LIB_DIR, exist_ok=True)
  os.makedirs(OBJ_DIR, exist_ok=True)
  hdr_t = max(os.path.getmtime(p) for p in _deps() if not p.endswith('.hip'))

  def compile_one(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o')
    if (not force and os.path.exists(obj) and
        os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t)):
      return obj
    flags = FLAGS[:-2] if os.path.basename(src) in AGPR_FORM else FLAGS
    cmd = [hipcc] + flags + extra + ['-DSPML_BUILD_EXPERIMENT=%d' % exp_code, '-c', src, '-o', obj]
    if verbose:
      print('[spml_amd] hipcc', os.path.basename(src), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    return obj

  with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
    objs = list(ex.map(compile_one, sources()))
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError('link failed:\n' + r.stderr[-4000:])
  with open(STAMP, 'w') as f:
    f.write(' '.join(extra))
  if verbose:
    print('[spml_amd] built', LIB_PATH, flush=True)
  return LIB_PATH


if __name__ == '__main__':
  build(force='--force' in sys.argv)
