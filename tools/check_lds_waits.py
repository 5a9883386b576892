#!/usr/bin/env python3
"""Static check of hand-issued LDS reads in a device assembly listing (hipcc -S --offload-device-only).

The pass kernels issue `ds_read_*` from inline asm and wait for them with counted `s_waitcnt lgkmcnt(N)`
statements that only name the destination registers as "+v" operands.  That keeps the ORDER, but it does not
stop the register allocator from copying a destination register BEFORE the wait (at a control-flow join, or when
it re-colours a value): the copy then reads the register before the LDS data has landed.  This script walks every
function linearly, keeps the in-order queue of outstanding DS operations, and reports any instruction that reads
(or overwrites) a register an outstanding `ds_read` will still write.

usage: check_lds_waits.py file.s [function-substring ...]   -> exit status 1 when a hazard is found
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


def check_function(name, lines):
  """-> list of (line number, text, registers) hazards"""
  pending = []            # in-order queue of outstanding DS ops: set of destination registers (may be empty)
  bad = []
  for no, raw in lines:
    ln = raw.split(';')[0].strip()
    if not ln or ln.endswith(':') or ln.startswith('.'):
      continue
    op = ln.split()[0]
    args = ln[len(op):]
    if op == 's_waitcnt':
      m = re.search(r'lgkmcnt\((\d+)\)', ln)
      if m:
        n = int(m.group(1))
        while len(pending) > n:
          pending.pop(0)
      continue
    busy = set().union(*pending) if pending else set()
    if op.startswith('ds_'):
      parts = [p.strip() for p in args.split(',')]
      if op.startswith('ds_read') or op.startswith('ds_bpermute') or op.startswith('ds_swizzle') or \
          op.startswith('ds_permute'):
        dst = regs(parts[0])
        src = regs(','.join(parts[1:]))
      else:                                       # stores
        dst, src = set(), regs(args)
      if (src | dst) & busy:
        bad.append((no, raw.strip(), sorted((src | dst) & busy)))
      pending.append(dst)
      continue
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
      pending.append(set())                       # scalar loads share the counter (out of order: conservative)
      continue
    if op in ('s_barrier', 's_nop', 's_endpgm') or op.startswith('s_cbranch') or op == 's_branch':
      continue
    touched = regs(args)
    if touched & busy:
      bad.append((no, raw.strip(), sorted(touched & busy)))
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
    print('%s: %d ds_read, %d hazards' % (name, n_ds, len(bad)))
    for no, text, rs in bad[:20]:
      print('   line %d: %s   <- pending %s' % (no, text, rs))
    rc |= bool(bad)
  return rc


if __name__ == '__main__':
  sys.exit(main())
