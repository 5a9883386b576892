"""The C-ABI shared library loads without a GPU and exports every symbol that
include/spml_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'spml_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(spml_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib_path():
  from spml_amd import _build
  return _build.build(verbose=False)


def test_header_declares_the_hot_path():
  names = declared_symbols()
  for must in ('spml_normalize_concat_loc_f32', 'spml_kmeans_run_f32', 'spml_kmeans_assign_f32',
               'spml_segment_sum_normalize_f32', 'spml_segsort_nll_fwd_f32',
               'spml_segsort_nll_bwd_f32', 'spml_topk_affinity_f32', 'spml_status_string'):
    assert must in names


def test_library_exports_every_declared_symbol(lib_path):
  handle = ctypes.CDLL(lib_path)
  missing = [n for n in declared_symbols() if not hasattr(handle, n)]
  assert not missing, missing


def test_ffi_table_matches_header(lib_path):
  from spml_amd import _ffi
  assert sorted(_ffi.EXPORTS) == declared_symbols()
  lib = _ffi.lib()
  import re
  hdr = open(os.path.join(ROOT, 'include', 'spml_hip.h')).read()
  want = int(re.search(r'#define SPML_ABI_VERSION (\d+)', hdr).group(1))
  assert lib.spml_abi_version() == want == _ffi.ABI_VERSION
  assert lib.spml_status_string(0) == b'ok'
  assert b'workspace' in lib.spml_status_string(-3)
  # host-only size queries
  assert lib.spml_kmeans_workspace_bytes(1000, 66, 36, 2, 500) > 1000 * 4
  assert lib.spml_segsort_nll_workspace_bytes(1000, 100, 64) > 0
  assert lib.spml_topk_workspace_bytes(100, 100, 64, 5) > 0
  assert lib.spml_kmeans_workspace_bytes(-1, 66, 36, 2, 500) == 0


def test_no_cpu_fallback():
  """A CPU tensor must raise, never silently compute."""
  import torch
  from spml_amd import _ffi
  import spml_amd.utils.segsort.common as sc
  import spml_amd.utils.general.common as gc
  x = torch.randn(10, 8)
  with pytest.raises(_ffi.SpmlHipError):
    gc.normalize_embedding(x)
  with pytest.raises(_ffi.SpmlHipError):
    sc.calculate_prototypes_from_labels(x, torch.zeros(10, dtype=torch.long), 1)
  with pytest.raises(_ffi.SpmlHipError):
    sc.kmeans_with_initial_labels(x, torch.zeros(10, dtype=torch.long), 1, 2)


def test_product_does_not_import_the_oracle():
  """Nothing under spml_amd/ may reference oracle/ (it is test infrastructure)."""
  bad = []
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'spml_amd')):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
          bad.append(os.path.join(dirpath, f))
  assert not bad, bad


def test_bench_refuses_to_run_without_a_gpu():
  """bench.py measures the HIP path only: without a GPU it must stop, not fall back."""
  import subprocess
  import sys
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is present')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '0'],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode != 0
  assert 'no CPU fallback' in (r.stderr + r.stdout)


@pytest.mark.parametrize('source,kernel', [('nll_de3.hip', 'nll_bwd_de3'), ('nll_dp3.hip', 'nll_bwd_dp3')])
def test_hand_assigned_registers_of_the_pipelined_nll_kernels_are_the_kernels_alone(source, kernel):
  """csrc/nll_de3.hip / nll_dp3.hip name their pipeline registers (v96..v255, a0..a255) in the text of single-instruction
  asm statements and confine the compiler to v0..v95: no compiler-generated instruction may touch a
  hand-assigned register, every inline-asm statement must be one instruction, and the generated register map
  must be the committed one (tools/gen_nll_de3.py)."""
  import subprocess
  import tempfile
  from spml_amd import _build
  inc = os.path.join(ROOT, 'spml_amd', 'csrc', 'nll_de3_regs.inc')
  with tempfile.TemporaryDirectory() as tmp:
    fresh = os.path.join(tmp, 'regs.inc')                  # (written next to nothing: the source tree stays as it is)
    subprocess.run(['python', os.path.join(ROOT, 'tools', 'gen_nll_de3.py'), fresh], check=True, capture_output=True)
    assert open(fresh).read() == open(inc).read(), 'nll_de3_regs.inc is not what tools/gen_nll_de3.py writes'
  with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'de3.s')
    cmd = [_build._hipcc()] + _build.FLAGS + ['-S', '--cuda-device-only', os.path.join(_build.CSRC, source),
                                              '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
  kernels = re.findall(r'^(_ZN4spml\S*' + kernel + r'\S*):[^\n]*\n(.*?)s_endpgm', text, flags=re.S | re.M)
  assert len(kernels) == 2, [k for k, _ in kernels]
  for name, body in kernels:
    in_asm, n_asm = False, 0
    for line in body.splitlines():
      code = line.split(';')[0].strip() if not line.strip().startswith(';;#') else line.strip()
      if code.startswith(';;#ASMSTART'):
        in_asm, n_asm = True, 0
        continue
      if code.startswith(';;#ASMEND'):
        in_asm = False
        continue
      if not code or code.endswith(':') or code.startswith('.'):
        continue
      if in_asm:
        n_asm += 1
        assert n_asm == 1 or code.startswith(('v_accvgpr_write_b32', 'v_mov_b32', 's_nop', 'global_load_lds', 's_mov_b32 m0', 's_barrier', 's_waitcnt')), \
            '%s: multi-instruction asm statement: %s' % (name, code)
        continue
      regs = re.findall(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]|\ba(\d+)\b|\ba\[(\d+):(\d+)\]', code)
      for v1, v2, v3, a1, a2, a3 in regs:
        assert not (a1 or a2), '%s: compiler-generated accumulation-register access: %s' % (name, code)
        hi = int(v1) if v1 else int(v3)
        assert hi < 96, '%s: compiler-generated access to a hand-assigned register: %s' % (name, code)
    assert body.count('v_mfma_f32_32x32x16_f16') >= 2 * 4 * 24, name      # two step versions x four steps x 24 slots
    assert 'scratch_' not in body, name + ' spills'
  assert '.amdhsa_accum_offset 256' in text and '.amdhsa_next_free_vgpr 512' in text


def test_hand_issued_instructions_of_the_kmeans_passes_keep_their_distances():
  """csrc/kmeans64.hip and kmeans64k.hip issue their LDS reads and every MFMA from inline asm, which hipcc does not see
  into: no wait states are inserted for them.  tools/check_asm_hazards.py re-derives the hazards from the compiled
  kernels (a register read while an LDS read into it is outstanding; a vector-ALU write straight before an MFMA that
  reads it; an MFMA result read too early; a dependent MFMA straight behind its producer -- the hardware interlocks
  none of them: tools/hw_probes/mfma_chain.hip, mfma_valu_raw.hip) for every instantiation (48 of kmeans_pass64, 16 of
  kmeans_assign64k + 16 of kmeans_accum64k); nothing may spill."""
  import subprocess
  import sys
  import tempfile
  from spml_amd import _build
  for src, kern, count in (('kmeans64.hip', 'kmeans_pass64', 48), ('kmeans64k.hip', '64k', 32)):
    with tempfile.TemporaryDirectory() as tmp:
      out = os.path.join(tmp, 'k.s')
      cmd = [_build._hipcc()] + _build.FLAGS + ['-S', '--cuda-device-only', os.path.join(_build.CSRC, src), '-o', out]
      r = subprocess.run(cmd, capture_output=True, text=True)
      assert r.returncode == 0, r.stderr[-2000:]
      assert 'scratch_' not in open(out).read(), src
      chk = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_asm_hazards.py'), out, kern],
                           capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-3000:]
    assert chk.stdout.count(' 0 hazards') == count, (src, chk.stdout[-1000:])


def test_profiling_builds_are_stamped_and_refused(monkeypatch):
  """a library compiled with SPML_CONV_EXP / SPML_P64_EXP (kernels that skip work and overwrite outputs)
  must not pass for a product build: the -D set is recorded next to the objects (a plain build after it rebuilds),
  compiled into the library (`spml_build_experiment`) and checked when the library is loaded."""
  from spml_amd import _build, _ffi
  assert _build._experiment_flags() == ([], 0)
  assert _build.is_fresh()                                   # (build() ran in this container: plain stamp)
  assert _ffi.lib().spml_build_experiment() == 0
  monkeypatch.setenv('SPML_CONV_EXP', '3')
  monkeypatch.setenv('SPML_P64_EXP', '16')
  flags, code = _build._experiment_flags()
  assert flags == ['-DSPML_CONV_EXP=3', '-DSPML_P64_EXP=16'] and code == (3 | (16 << 16))
  assert not _build.is_fresh()                               # the objects on disk are of another -D set
  # a fresh load of the plain library under those variables is refused
  monkeypatch.setattr(_ffi, '_lib', None)
  with pytest.raises(_ffi.SpmlHipError, match='profiling build'):
    _ffi.lib()
  monkeypatch.delenv('SPML_CONV_EXP')
  monkeypatch.delenv('SPML_P64_EXP')
  assert _ffi.lib().spml_build_experiment() == 0
