"""tools/check_asm_hazards.py on hand-made listings: every rule fires on the pattern it is for and stays silent on the
repaired one (the checker guards the hand-issued MFMA / LDS streams of csrc/kmeans64.hip and kmeans64k.hip)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('check_asm_hazards', os.path.join(ROOT, 'tools', 'check_asm_hazards.py'))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def run(text):
  lines = [(i + 1, ln) for i, ln in enumerate(text.strip('\n').split('\n'))]
  return [why.split()[0] for _, _, why, _ in chk.check_function('f', lines)]


MFMA = '\tv_mfma_f32_16x16x32_f16 v[0:3], a[0:3], v[8:11], v[0:3]'


def test_read_of_a_register_an_outstanding_lds_read_will_write():
  bad = '\tds_read_b128 v[8:11], v20\n\tv_mov_b32_e32 v30, v8\n\ts_waitcnt lgkmcnt(0)'
  good = '\tds_read_b128 v[8:11], v20\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v30, v8'
  assert run(bad) == ['R1'] and run(good) == []
  # counted wait: the OLDER read has landed, the younger one has not
  two = '\tds_read_b128 v[8:11], v20\n\tds_read_b128 v[12:15], v20 offset:1024\n\ts_waitcnt lgkmcnt(1)\n'
  assert run(two + '\tv_mov_b32_e32 v30, v9') == [] and run(two + '\tv_mov_b32_e32 v30, v13') == ['R1']


def test_vector_alu_write_straight_before_a_hand_written_mfma():
  bad = '\tv_mov_b32_e32 v8, v40\n\t;;#ASMSTART\n' + MFMA + '\n\t;;#ASMEND'
  good = '\tv_mov_b32_e32 v8, v40\n\t;;#ASMSTART\n\ts_nop 1\n' + MFMA + '\n\t;;#ASMEND'
  assert run(bad) == ['R2'] and run(good) == []
  # the compiler's own MFMAs get their wait states from the compiler: not flagged
  assert run('\tv_mov_b32_e32 v8, v40\n' + MFMA) == []


def test_mfma_result_read_too_early():
  asm = '\t;;#ASMSTART\n\ts_nop 1\n' + MFMA + '\n\t;;#ASMEND\n'
  assert run(asm + '\tv_add_f32_e32 v40, v0, v41') == ['R3']
  assert run(asm + '\ts_nop 15\n\ts_nop 7\n\tv_add_f32_e32 v40, v0, v41') == []
  assert run(asm + '\tglobal_store_dwordx4 v[50:51], v[0:3], off') == ['R3']
  # another MFMA accumulating in place into the same registers is fine two slots later (R4 is about distance 1)
  other = '\t;;#ASMSTART\n\ts_nop 1\n\tv_mfma_f32_16x16x32_f16 v[4:7], a[0:3], v[8:11], v[4:7]\n\t;;#ASMEND\n'
  assert run(asm + other + asm) == []


def test_dependent_mfma_straight_behind_its_producer():
  one = '\t;;#ASMSTART\n' + MFMA + '\n\t;;#ASMEND\n'
  assert run(one + one) == ['R4']
  other = '\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[4:7], a[0:3], v[8:11], v[4:7]\n\t;;#ASMEND\n'
  assert run(one + other + one) == []


def test_main_reads_a_listing_per_function(tmp_path):
  text = ('_ZN4spml1fEv:\n\t;;#ASMSTART\n' + MFMA + '\n\t;;#ASMEND\n\t;;#ASMSTART\n' + MFMA + '\n\t;;#ASMEND\n\ts_endpgm\n'
          '.Lfunc_end0:\n_ZN4spml1gEv:\n\ts_nop 0\n\ts_endpgm\n.Lfunc_end1:\n')
  p = tmp_path / 'k.s'
  p.write_text(text)
  import subprocess
  import sys
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_asm_hazards.py'), str(p)], capture_output=True,
                     text=True)
  assert r.returncode == 1 and '1 hazards' in r.stdout and ' 0 hazards' in r.stdout
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_asm_hazards.py'), str(p), '1gEv'],
                     capture_output=True, text=True)
  assert r.returncode == 0
