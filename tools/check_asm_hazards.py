#!/usr/bin/env python3
"""Static hazard check of hand-issued instructions in a device assembly listing (hipcc -S --offload-device-only).

hipcc inserts the wait states gfx950 needs around instructions it knows; it does not look inside inline asm.  The
k-means pass kernels issue LDS reads and MFMAs from inline asm, so this script re-checks, per function, linearly:

 R1  no instruction reads or overwrites a register an outstanding hand-issued `ds_read*` will still write (the counted
     `s_waitcnt lgkmcnt(N)` statements only name the registers as "+v" operands: the register allocator may still place a
     copy before the wait);
 R2  no asm MFMA reads (srcA / srcB / srcC) a register the vector ALU wrote less than 2 issue slots earlier (measured:
     tools/hw_probes/mfma_valu_raw.hip -- no interlock);
 R3  nothing but an MFMA accumulating in place reads a register an asm MFMA wrote less than 20 issue slots earlier
     (`s_nop N` = N + 1 slots, an MFMA = 4, anything else = 1);
 R4  an asm MFMA accumulating in place (srcC == vdst) is not issued straight behind the MFMA that produced its srcC
     (tools/hw_probes/mfma_chain.hip: wrong sums at distance 1, exact from distance 2).

usage: check_asm_hazards.py file.s [function-substring ...]   -> exit status 1 when a hazard is found
"""
import re
import sys

REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs(text):
  out = set()
  for m in REG.finditer(text):
    if m.group(1):
      out.add((m.group(1), int(m.group(2))))
    else:
      for i in range(int(m.group(4)), int(m.group(5)) + 1):
        out.add((m.group(3), i))
  return out


def split_args(args):
  return [p.strip() for p in args.split(',')]


def check_function(name, lines):
  pending = []            # R1: in-order queue of outstanding DS ops: set of destination registers (may be empty)
  valu_age = {}           # R2: register -> issue slots since a vector-ALU write
  mfma_age = {}           # R3: register -> issue slots since an asm-MFMA write
  last_mfma_dst = None    # R4
  bad = []
  in_asm = False
  for no, raw in lines:
    if '#ASMSTART' in raw:
      in_asm = True
      continue
    if '#ASMEND' in raw:
      in_asm = False
      continue
    ln = raw.split(';')[0].strip()
    if not ln or ln.endswith(':') or ln.startswith('.'):
      continue
    op = ln.split()[0]
    args = ln[len(op):]
    slots = 1
    if op == 's_nop':
      slots = int(args.strip()) + 1
    elif op.startswith('v_mfma'):
      slots = 4

    def age(d):
      for k in list(d):
        d[k] += slots
        if d[k] > 64:
          del d[k]

    if op == 's_waitcnt':
      m = re.search(r'lgkmcnt\((\d+)\)', ln)
      if m:
        n = int(m.group(1))
        while len(pending) > n:
          pending.pop(0)
      age(valu_age); age(mfma_age)
      continue
    if op == 's_nop' or op == 's_barrier' or op.startswith('s_cbranch') or op in ('s_branch', 's_endpgm'):
      age(valu_age); age(mfma_age)
      if op != 's_nop':
        last_mfma_dst = None
      continue
    busy = set().union(*pending) if pending else set()
    parts = split_args(args)
    if op.startswith('v_mfma'):
      dst, src = regs(parts[0]), regs(','.join(parts[1:]))
      inplace = regs(parts[3]) == dst if len(parts) > 3 else False
      if (dst | src) & busy:
        bad.append((no, raw.strip(), 'R1 pending LDS read', sorted((dst | src) & busy)))
      if in_asm:
        young = {r for r in src if valu_age.get(r, 99) < 2}
        if young:
          bad.append((no, raw.strip(), 'R2 vector-ALU write straight before', sorted(young)))
        if inplace and last_mfma_dst is not None and dst == last_mfma_dst:
          bad.append((no, raw.strip(), 'R4 dependent MFMA back to back', sorted(dst)[:1]))
        # a source other than the in-place accumulator that an asm MFMA wrote recently
        fresh = {r for r in (src if not inplace else regs(','.join(parts[1:3]))) if mfma_age.get(r, 99) < 20}
        if fresh:
          bad.append((no, raw.strip(), 'R3 MFMA result read too early', sorted(fresh)))
      age(valu_age); age(mfma_age)
      for r in dst:
        valu_age.pop(r, None)
        if in_asm:
          mfma_age[r] = 0
        else:
          mfma_age.pop(r, None)
      last_mfma_dst = dst
      continue
    last_mfma_dst = None
    if op.startswith('ds_'):
      if op.startswith(('ds_read', 'ds_bpermute', 'ds_swizzle', 'ds_permute')):
        dst, src = regs(parts[0]), regs(','.join(parts[1:]))
      else:
        dst, src = set(), regs(args)
      if (src | dst) & busy:
        bad.append((no, raw.strip(), 'R1 pending LDS read', sorted((src | dst) & busy)))
      fresh = {r for r in src if mfma_age.get(r, 99) < 20}
      if fresh:
        bad.append((no, raw.strip(), 'R3 MFMA result read too early', sorted(fresh)))
      pending.append(dst)
      age(valu_age); age(mfma_age)
      continue
    if op.startswith(('s_load', 's_buffer_load')):
      pending.append(set())
      age(valu_age); age(mfma_age)
      continue
    touched = regs(args)
    if touched & busy:
      bad.append((no, raw.strip(), 'R1 pending LDS read', sorted(touched & busy)))
    if op.startswith(('v_', 'global_', 'buffer_', 'scratch_', 'flat_')):
      if op.startswith('v_') and not op.startswith('v_cmp'):
        dst, src = regs(parts[0]), regs(','.join(parts[1:]))
      elif op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')) and 'lds' not in op:
        dst, src = regs(parts[0]), regs(','.join(parts[1:]))
      else:
        dst, src = set(), regs(args)
      fresh = {r for r in src if mfma_age.get(r, 99) < 20}
      if fresh:
        bad.append((no, raw.strip(), 'R3 MFMA result read too early', sorted(fresh)))
      age(valu_age); age(mfma_age)
      if op.startswith('v_'):
        for r in dst:
          valu_age[r] = 0
          mfma_age.pop(r, None)
      continue
    age(valu_age); age(mfma_age)
  return bad


def main():
  path = sys.argv[1]
  wanted = sys.argv[2:]
  funcs, cur, name = {}, None, None
  for no, raw in enumerate(open(path), 1):
    m = re.match(r'^(_Z\w+):', raw)
    if m:
      name, cur = m.group(1), []
      funcs[name] = cur
      continue
    if raw.startswith('\t.amdhsa_kernel') or raw.startswith('.Lfunc_end'):
      cur = None
    if cur is not None:
      cur.append((no, raw))
  rc = 0
  for name, lines in funcs.items():
    if wanted and not any(w in name for w in wanted):
      continue
    bad = check_function(name, lines)
    n_ds = sum(1 for _, l in lines if l.strip().startswith('ds_read'))
    n_mf = sum(1 for _, l in lines if l.strip().startswith('v_mfma'))
    print('%s: %d ds_read, %d mfma, %d hazards' % (name, n_ds, n_mf, len(bad)))
    for no, text, why, rs in bad[:12]:
      print('   line %d: %-90s %s %s' % (no, text[:90], why, rs[:6]))
    rc |= bool(bad)
  return rc


if __name__ == '__main__':
  sys.exit(main())
