"""Socket power, shader clock and per-wave phase cycles of the matrix-core convolutions.

One convolution (forward GEMM and weight gradient of the res4 / res5 shapes) runs in a loop for a few seconds
while `rocm-smi --showpower --showclocks` is sampled: is the kernel at the 1400 W socket limit, and what clock does the
firmware leave it?  With a SPML_CONV_EXP & 48 build (tools/run_conv_power.sh builds the variants) the kernel also
leaves the shader clocks of one tile and the cycles its four waves spent waiting for DMA data, in the barrier,
issuing the next stage's DMA and in fragment reads + MFMAs.
profiles/r05_conv_power.md is made from this.
"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from spml_amd import _ffi

EXP = int(os.environ.get('SPML_CONV_EXP', '0'))


class Sampler(object):
  def __init__(self):
    self.samples, self.stop = [], False
    self.th = threading.Thread(target=self.run)

  def run(self):
    while not self.stop:
      try:
        o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True,
                           timeout=10).stdout
      except Exception as e:       # noqa
        o = ''
      pw = re.findall(r'Power \(W\):\s*([0-9.]+)', o)
      sc = re.findall(r'sclk clock level:.*\((\d+)Mhz\)', o)
      if pw and sc:
        self.samples.append((float(pw[0]), float(sc[0])))
      time.sleep(0.25)

  def __enter__(self):
    self.th.start()
    return self

  def __exit__(self, *a):
    self.stop = True
    self.th.join()

  def mean(self):
    s = self.samples[2:] or self.samples
    if not s:
      return float('nan'), float('nan')
    return sum(p for p, _ in s) / len(s), sum(c for _, c in s) / len(s)


def loop(fn, seconds):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  with Sampler() as sm:
    t0, it = time.time(), 0
    while time.time() - t0 < seconds:
      for _ in range(50):
        out = fn()
      torch.cuda.synchronize()
      it += 50
    el = time.time() - t0
  w, mhz = sm.mean()
  return el / it * 1e6, w, mhz, out


def main():
  n, h, w = 16, 65, 65
  secs = float(os.environ.get('SECONDS_PER_CASE', '3'))
  idle = subprocess.run(['rocm-smi', '--showpower'], capture_output=True, text=True).stdout
  m = re.findall(r'Power \(W\):\s*([0-9.]+)', idle)
  print('SPML_CONV_EXP=%d   idle socket power %s W' % (EXP, m[0] if m else '?'), flush=True)
  for cin, cout, k, dil in [(2048, 512, 1, 1), (512, 512, 3, 4), (256, 1024, 1, 1), (1024, 256, 1, 1)]:
    x = torch.randn(n, cin, h, w, device='cuda').clamp_min(0).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, device='cuda') * 0.02).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, w, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-4
    xa, dya = _ffi.hl8_from_f32(x), _ffi.hl8_from_f32(dy)
    wf, wtr = _ffi.hl8_weight(wt)
    flops = 2.0 * n * h * w * cin * cout * k * k * 3
    us, watt, mhz, out = loop(lambda: _ffi.conv_hl8(xa, wf, n, h, w, k * k, dil), secs)
    line = '%4d->%4d k%d fwd  : %7.1f us  %.2f of 2.5 PF  %6.0f W  %4.0f MHz  %.3f J/launch' % (
        cin, cout, k, us, flops / us / 1e6 / 2500., watt, mhz, watt * us * 1e-6)
    if EXP & 16:
      o = out.permute(0, 2, 3, 1).reshape(-1)[:24].tolist()
      line += '   tile: %.0f cycles at %.0f MHz' % (o[0], o[0] / max(o[1], 1.) * 100)
      print(line, flush=True)
      if EXP & 32:
        stages = cin // 16 * k * k
        for wv in range(4):
          print('      wave %d, cycles per stage: DMA wait %5.0f  barrier %5.0f  DMA issue %5.0f  reads+MFMA %5.0f' % (
              (wv,) + tuple(v / stages for v in o[8 + 4 * wv:12 + 4 * wv])), flush=True)
    else:
      print(line, flush=True)
    if not EXP:
      us, watt, mhz, _ = loop(lambda: _ffi.conv_hl8(dya, wtr, n, h, w, k * k, dil), secs)
      print('%4d->%4d k%d dgrad: %7.1f us  %.2f of 2.5 PF  %6.0f W  %4.0f MHz  %.3f J/launch' % (
          cin, cout, k, us, flops / us / 1e6 / 2500., watt, mhz, watt * us * 1e-6), flush=True)
      us, watt, mhz, _ = loop(lambda: _ffi.conv_wgrad_hl8(dya, xa, n, h, w, k * k, dil), secs)
      print('%4d->%4d k%d wgrad: %7.1f us  %.2f of 2.5 PF  %6.0f W  %4.0f MHz  %.3f J/launch' % (
          cin, cout, k, us, flops / us / 1e6 / 2500., watt, mhz, watt * us * 1e-6), flush=True)
      xf = x
      us, watt, mhz, _ = loop(lambda: torch.nn.functional.conv2d(xf, wt, padding=dil * (k // 2), dilation=dil), secs)
      print('%4d->%4d k%d library fp32 fwd: %7.1f us  %6.0f W  %4.0f MHz  %.3f J/launch' % (
          cin, cout, k, us, watt, mhz, watt * us * 1e-6), flush=True)


if __name__ == '__main__':
  main()
