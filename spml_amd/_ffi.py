"""ctypes binding of libspml_hip.so (include/spml_hip.h).

Torch is plumbing only: it owns device memory and the HIP stream; every call
below passes raw `data_ptr()`s and `torch.cuda.current_stream().cuda_stream` to
the C-ABI.  There is no CPU fallback: if the library is missing or a tensor is
not on a GPU the call raises."""
import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPML_HIP_LIB: load another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get('SPML_HIP_LIB') or os.path.join(_HERE, 'lib', 'libspml_hip.so')

_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); mirrors include/spml_hip.h one to one.
_P = c_void_p
_SIGNATURES = {
    'spml_status_string': (c_char_p, [c_int]),
    'spml_abi_version': (c_int, []),
    'spml_normalize_concat_loc_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'spml_normalize_concat_loc_bwd_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'spml_normalize_concat_local_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    'spml_normalize_concat_local_nhwc_supported': (c_int, [c_int, c_int]),
    'spml_normalize_concat_local_nhwc_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    'spml_normalize_concat_local_nhwc_bwd_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P,
                                                         _P, _P]),
    'spml_normalize_concat_local_bwd_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P,
                                                    _P, _P]),
    'spml_normalize_rows_f32': (c_int, [_P, c_int64, c_int, _P, _P]),
    'spml_normalize_rows_bwd_f32': (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    'spml_kmeans_init_grid_i64': (c_int, [c_int, c_int, c_int, c_int, _P, _P]),
    'spml_relabel_unique_workspace_bytes': (c_size_t, [c_int64]),
    'spml_relabel_unique_i64': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_size_t, _P]),
    'spml_kmeans_workspace_bytes': (c_size_t, [c_int64, c_int, c_int, c_int, c_int64]),
    'spml_kmeans_run_f32': (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, c_int, _P, _P,
                                    c_int, _P, c_size_t, _P]),
    'spml_kmeans_assign_f32': (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, _P, c_int,
                                       _P, c_size_t, _P]),
    'spml_kmeans_fused_pass_f32': (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, _P, _P,
                                           c_int, _P, c_size_t, _P]),
    'spml_kmeans_preconvert_f32': (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, c_size_t,
                                           _P]),
    'spml_kmeans_path_name': (c_char_p, [c_int64, c_int, c_int, c_int, c_int64, c_int, c_int, c_int]),
    'spml_kmeans_profile_layout': (c_int, [c_int64, c_int, c_int, c_int, c_int64, c_int, _P, _P]),
    'spml_kmeans_run_profiled_f32': (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, c_int,
                                             _P, c_int, _P, c_size_t, _P, c_size_t, _P]),
    'spml_segment_sum_normalize_f32': (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P, _P]),
    'spml_segment_sum_normalize_bwd_f32': (c_int, [_P, _P, _P, c_int64, c_int, c_int64, _P, _P, c_int, _P]),
    'spml_segment_sum_det_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'spml_segment_sum_normalize_det_f32': (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    'spml_set_deterministic': (c_int, [c_int]),
    'spml_get_deterministic': (c_int, []),
    'spml_build_experiment': (c_int, []),
    'spml_segsort_nll_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int]),
    'spml_segsort_nll_fwd_f32': (c_int, [_P, _P, _P, c_int64, _P, _P, c_int64, c_int, c_float, c_int,
                                         _P, _P, _P, c_size_t, _P]),
    'spml_segsort_nll_bwd_f32': (c_int, [_P, _P, _P, c_int64, _P, _P, c_int64, c_int, c_float, c_int,
                                         _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    'spml_topk_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int, c_int]),
    'spml_topk_affinity_f32': (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, _P, _P, _P, c_float,
                                       _P, _P, _P, c_size_t, _P]),
    'spml_affinity_workspace_bytes': (c_size_t, [c_int, c_int, c_int64]),
    'spml_affinity_transition_f32': (c_int, [_P, c_int, c_int, c_int64, c_float, c_int, _P, _P, c_size_t,
                                             _P]),
    'spml_bn_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'spml_bn_act_fwd_f32': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P,
                                    _P, c_size_t, _P]),
    'spml_bn_act_bwd_f32': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'spml_hl8_from_f32': (c_int, [_P, c_int64, c_int, _P, c_int, _P, _P]),
    'spml_hl8_weight_transposed_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    'spml_conv_hl8_supported': (c_int, [c_int, c_int, c_int]),
    'spml_conv_hl8_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'spml_conv_hl8_stats_layout': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'spml_conv_hl8_stats_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'spml_bn_fwd_hl8_chunks_f32': (c_int, [_P, _P, c_int, c_int, _P, _P, c_int64, c_int, _P, _P, _P, _P, c_float, c_float,
                                           c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'spml_conv_wgrad_hl8_supported': (c_int, [c_int, c_int, c_int]),
    'spml_conv_wgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'spml_conv_wgrad_hl8_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                        c_size_t, _P]),
    'spml_conv_wgrad_pyramid_hl8_supported': (c_int, [c_int, c_int, c_int]),
    'spml_conv_tap_gather_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'spml_conv_wgrad_pyramid_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'spml_conv_wgrad_pyramid_hl8_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P,
                                                c_size_t, _P]),
    'spml_bn_stats_ext_f32': (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'spml_bn_stats_ext_chunks_f32': (c_int, [_P, c_int, c_int, c_int64, c_int, _P, _P, _P, _P, _P]),
    'spml_bn_finalize_f32': (c_int, [_P, _P, c_int, c_double, c_float, c_float, _P, _P, _P, _P]),
    'spml_bn_act_apply_hl8_f32': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P,
                                          _P]),
    'spml_bn_act_bwd_reduce_ext_f32': (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'spml_bn_act_bwd_apply_hl8_f32': (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                              c_double, _P, _P, _P, _P, _P, _P]),
    'spml_bn_fwd_hl8_f32': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P,
                                    _P, _P, _P, _P, _P, c_size_t, _P]),
    'spml_bn_bwd_hl8_f32': (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                    c_size_t, _P]),
    'spml_conv_hl8_pyramid_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                          _P]),
    'spml_hl8_weight_transposed_into_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, _P]),
    'spml_absmax_bound_f32': (c_int, [_P, c_int64, _P, c_int, _P]),
    'spml_hl8_weight_set_f32': (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'spml_bn_finalize_ranks_f32': (c_int, [_P, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'spml_conv_hl8_affine_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, _P]),
    'spml_upsample_ce_supported': (c_int, [c_int]),
    'spml_upsample_ce_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'spml_upsample_ce_fwd_f32': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, _P, _P, _P,
                                         c_size_t, _P]),
    'spml_upsample_ce_bwd_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, _P, _P, _P]),
    'spml_bn_stats_f32': (c_int, [_P, c_int64, c_int, _P, _P, _P, c_size_t, _P]),
    'spml_bn_act_apply_f32': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    'spml_bn_act_bwd_reduce_f32': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'spml_bn_act_bwd_apply_f32': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, c_double, _P, _P, _P,
                                          _P]),
    'spml_clock_probe': (c_int, [_P, c_int, _P]),
    'spml_maxpool3x3s2_nhwc_f32': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    'spml_window_accumulate_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int,
                                           _P]),
}

EXPORTS = tuple(_SIGNATURES)


class SpmlHipError(RuntimeError):
  pass


ABI_VERSION = 4            # = SPML_ABI_VERSION of include/spml_hip.h (tests/test_cabi_exports.py compares the two)


def lib():
  """Loads libspml_hip.so once; raises if it has not been built."""
  global _lib
  if _lib is None:
    with _lock:
      if _lib is None:
        if not os.path.exists(LIB_PATH):
          raise SpmlHipError(
              'libspml_hip.so not found at %s -- build it with `python -m spml_amd._build` '
              '(there is no CPU fallback for the HIP path)' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
          try:
            fn = getattr(handle, name)
          except AttributeError:
            raise SpmlHipError('libspml_hip.so does not export %s (stale build? run '
                               '`python -m spml_amd._build --force`)' % name)
          fn.restype = res
          fn.argtypes = args
        got = handle.spml_abi_version()
        if got != ABI_VERSION:
          raise SpmlHipError('libspml_hip.so has ABI version %d, this wrapper was written against %d (stale build? '
                             'run `python -m spml_amd._build --force`)' % (got, ABI_VERSION))
        exp = handle.spml_build_experiment()
        want = (int(os.environ.get('SPML_CONV_EXP') or 0) & 0xffff) | ((int(os.environ.get('SPML_P64_EXP') or 0) & 0xffff) << 16)
        if exp != want:
          raise SpmlHipError('libspml_hip.so is a profiling build (SPML_CONV_EXP=%d, SPML_P64_EXP=%d: kernels that skip work '
                             'and overwrite outputs) but this process asks for (%d, %d) -- rebuild with '
                             '`python -m spml_amd._build`' % (exp & 0xffff, exp >> 16, want & 0xffff, want >> 16))
        if os.environ.get('SPML_DETERMINISTIC', '0') not in ('', '0'):
          handle.spml_set_deterministic(1)
        _lib = handle
  return _lib


def set_deterministic(on):
  """Deterministic mode of the library (include/spml_hip.h, spml_set_deterministic; also switched on by
  SPML_DETERMINISTIC=1 in the environment when the library is loaded): the segment sums and the prototype gradient
  of the NLL backward are accumulated in 64-bit fixed point instead of through fp32 atomics.  Returns the previous
  setting."""
  return bool(lib().spml_set_deterministic(1 if on else 0))


def deterministic():
  return bool(lib().spml_get_deterministic())


def check(rc, what):
  if rc != 0:
    raise SpmlHipError('%s failed: %s (status %d)' % (
        what, lib().spml_status_string(rc).decode(), rc))


def stream_ptr():
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, allow_none=False):
  """Device pointer of a contiguous GPU tensor (or NULL)."""
  if t is None:
    if allow_none:
      return c_void_p(0)
    raise SpmlHipError('missing tensor argument')
  if not t.is_cuda:
    raise SpmlHipError('the HIP path needs GPU tensors (got %s); there is no CPU fallback'
                       % t.device)
  if t.device.index != torch.cuda.current_device():
    # kernels are launched on the CURRENT device's current stream with raw pointers
    raise SpmlHipError('tensor lives on %s but the current device is cuda:%d -- wrap the call in '
                       '`torch.cuda.device(tensor.device)`' % (t.device, torch.cuda.current_device()))
  if not t.is_contiguous():
    raise SpmlHipError('tensor must be contiguous')
  if dtype is not None and t.dtype != dtype:
    raise SpmlHipError('expected dtype %s, got %s' % (dtype, t.dtype))
  return c_void_p(t.data_ptr())


def workspace(nbytes, device):
  """A scratch buffer of at least `nbytes` for one library call.  Large sizes are rounded up to 1/8 .. 1/16 of
  themselves: workspaces that follow the prototype / pixel counts of a step drift by a few per cent from step to
  step, and a request a few MB above every cached block made the caching allocator map a new multi-GB segment in the
  middle of a run (config 5: 18 GB in the 7th step, a 480-ms stall; `BENCH_MEM_TRACE=1 python bench.py --recipe
  stress`) -- rounded sizes hit the block of the step before."""
  n = max(int(nbytes), 16)
  if n > (64 << 20):
    g = min(1 << (n.bit_length() - 4), 256 << 20)      # (at most 256 MB of slack: config 5's 18-GB workspace grew by 2 GB)
    n = (n + g - 1) // g * g
  return torch.empty((n,), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------
# thin typed wrappers (no autograd here; see spml_amd/ops.py)
# ---------------------------------------------------------------------------

def k1_channels_last(emb, nl):
  """True when K1 can stream `emb` as it is: channels-last storage and a channel count the row-wise
  kernels take (no NHWC -> NCHW copy in front of K1, none behind its backward)."""
  return (emb.dim() == 4 and emb.dtype == torch.float32 and not emb.is_contiguous() and
          emb.is_contiguous(memory_format=torch.channels_last) and
          bool(lib().spml_normalize_concat_local_nhwc_supported(int(emb.shape[1]), int(nl))))


def normalize_concat_loc(emb, loc=None, row_map=None, num_rows=None, want_emb=True,
                         want_loc=True):
  n, c, h, w = emb.shape
  nl = 2 if loc is None else int(loc.shape[-1])       # local-feature channels ((y, x) = 2)
  rows = n * h * w if num_rows is None else int(num_rows)
  out_emb = torch.empty((rows, c), dtype=torch.float32, device=emb.device) if want_emb else None
  out_loc = torch.empty((rows, c + nl), dtype=torch.float32, device=emb.device) if want_loc else None
  if k1_channels_last(emb, nl):
    check(lib().spml_normalize_concat_local_nhwc_f32(
        _ptr_any(emb), n, c, h, w, ptr(loc, torch.float32, True), nl,
        ptr(row_map, torch.int64, True), ptr(out_emb, None, True), ptr(out_loc, None, True),
        stream_ptr()), 'spml_normalize_concat_local_nhwc_f32')
    return out_emb, out_loc
  check(lib().spml_normalize_concat_local_f32(
      ptr(emb, torch.float32), n, c, h, w, ptr(loc, torch.float32, True), nl,
      ptr(row_map, torch.int64, True), ptr(out_emb, None, True), ptr(out_loc, None, True),
      stream_ptr()), 'spml_normalize_concat_local_f32')
  return out_emb, out_loc


def normalize_concat_loc_bwd(emb, loc, row_map, d_out_emb, d_out_loc):
  n, c, h, w = emb.shape
  nl = 2 if loc is None else int(loc.shape[-1])
  d_emb = torch.empty_like(emb)                        # (keeps the memory format of emb)
  if k1_channels_last(emb, nl):
    check(lib().spml_normalize_concat_local_nhwc_bwd_f32(
        _ptr_any(emb), n, c, h, w, ptr(loc, torch.float32, True), nl,
        ptr(row_map, torch.int64, True), ptr(d_out_emb, torch.float32, True),
        ptr(d_out_loc, torch.float32, True), _ptr_any(d_emb), stream_ptr()),
          'spml_normalize_concat_local_nhwc_bwd_f32')
    return d_emb
  check(lib().spml_normalize_concat_local_bwd_f32(
      ptr(emb, torch.float32), n, c, h, w, ptr(loc, torch.float32, True), nl,
      ptr(row_map, torch.int64, True), ptr(d_out_emb, torch.float32, True),
      ptr(d_out_loc, torch.float32, True), ptr(d_emb), stream_ptr()),
        'spml_normalize_concat_local_bwd_f32')
  return d_emb


def normalize_rows(x):
  d = x.shape[-1]
  x2 = x.reshape(-1, d)
  y = torch.empty_like(x2)
  check(lib().spml_normalize_rows_f32(ptr(x2, torch.float32), x2.shape[0], d, ptr(y),
                                      stream_ptr()), 'spml_normalize_rows_f32')
  return y.view(x.shape)


def normalize_rows_bwd(x, dy):
  d = x.shape[-1]
  x2, g2 = x.reshape(-1, d), dy.reshape(-1, d).contiguous()
  dx = torch.empty_like(x2)
  check(lib().spml_normalize_rows_bwd_f32(ptr(x2, torch.float32), ptr(g2, torch.float32),
                                          x2.shape[0], d, ptr(dx), stream_ptr()),
        'spml_normalize_rows_bwd_f32')
  return dx.view(x.shape)


def kmeans_init_grid(h, w, ky, kx, device):
  out = torch.empty((h, w), dtype=torch.int64, device=device)
  check(lib().spml_kmeans_init_grid_i64(h, w, ky, kx, ptr(out), stream_ptr()),
        'spml_kmeans_init_grid_i64')
  return out


_last_kmeans = None        # arguments of this thread's last k-means call (for kmeans_last_path)


def relabel_unique(keys, with_uniq=True, padded=False):
  """-> (uniq, inv [P], count [1] device tensor).  with_uniq: `uniq` = the U sorted distinct keys (the
  count is read on the host: one sync); padded: `uniq` has P entries, the sorted distinct keys followed
  by INT64_MAX (still sorted; no sync); neither: `uniq` is None and nothing synchronises.
  Keys: any int64 except INT64_MIN (the empty-slot sentinel of the hash set; the callers build keys from
  non-negative labels).  Cost ~ P + U^2 / 2048 * 11 compares: meant for U << P (segments, not pixels)."""
  keys = keys.reshape(-1)
  if keys.dtype != torch.int64 or not keys.is_contiguous():
    keys = keys.long().contiguous()
  p = keys.shape[0]
  inv = torch.empty((p,), dtype=torch.int64, device=keys.device)
  count = torch.empty((1,), dtype=torch.int64, device=keys.device)
  if padded:
    uniq = torch.full((p,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=keys.device)
  else:
    uniq = torch.empty((p if with_uniq else 0,), dtype=torch.int64, device=keys.device)
  ws = workspace(lib().spml_relabel_unique_workspace_bytes(p), keys.device)
  check(lib().spml_relabel_unique_i64(ptr(keys, torch.int64, p == 0), p, ptr(inv, allow_none=p == 0),
                                      ptr(uniq, allow_none=uniq.numel() == 0), uniq.numel(), ptr(count), ptr(ws),
                                      ws.numel(), stream_ptr()), 'spml_relabel_unique_i64')
  if padded:
    return uniq, inv, count
  if not with_uniq:
    return None, inv, count
  return uniq[:int(count.item())], inv, count


def _note_kmeans(p, d, k, n_img, max_seg_len, iterations, given, flags):
  global _last_kmeans
  _last_kmeans = (int(p), int(d), int(k), int(n_img), int(max_seg_len), int(iterations), int(given),
                  int(flags))


def kmeans_path_name(p, d, k, n_img, max_seg_len, iterations, given_centroids=False, flags=0):
  """Code path a k-means call with these arguments takes (a pure function of the library)."""
  return lib().spml_kmeans_path_name(int(p), int(d), int(k), int(n_img), int(max_seg_len),
                                     int(iterations), int(bool(given_centroids)), int(flags)).decode()


def kmeans_last_path():
  """Path name of the last k-means call made through this module (Python-side bookkeeping
  over spml_kmeans_path_name; the library keeps no state)."""
  return 'none' if _last_kmeans is None else lib().spml_kmeans_path_name(*_last_kmeans).decode()


def kmeans_run(x, seg_offsets, max_seg_len, k, labels_init, iterations, want_centroids=False,
               flags=0):
  p, d = x.shape
  n_img = seg_offsets.shape[0] - 1
  labels = torch.empty((p,), dtype=torch.int64, device=x.device)
  cent = (torch.zeros((n_img, k, d), dtype=torch.float32, device=x.device)
          if want_centroids else None)
  nbytes = lib().spml_kmeans_workspace_bytes(p, d, k, n_img, max_seg_len)
  ws = workspace(nbytes, x.device)
  check(lib().spml_kmeans_run_f32(
      ptr(x, torch.float32), p, d, ptr(seg_offsets, torch.int64), n_img, int(max_seg_len), k,
      ptr(labels_init, torch.int64), int(iterations), ptr(labels), ptr(cent, None, True),
      int(flags), ptr(ws), ws.numel(), stream_ptr()), 'spml_kmeans_run_f32')
  _note_kmeans(p, d, k, n_img, max_seg_len, iterations, 0, flags)
  return (labels, cent) if want_centroids else labels


def kmeans_run_profiled(x, seg_offsets, max_seg_len, k, labels_init, iterations, flags=0):
  """-> (labels, durations_us [iterations + 1]): every pass kernel's duration from the
  per-workgroup device time stamps (max end - min start, 100-MHz s_memrealtime)."""
  import ctypes
  p, d = x.shape
  n_img = seg_offsets.shape[0] - 1
  n_pass, wgs = ctypes.c_int(0), ctypes.c_int(0)
  check(lib().spml_kmeans_profile_layout(p, d, k, n_img, int(max_seg_len), int(iterations),
                                         ctypes.byref(n_pass), ctypes.byref(wgs)),
        'spml_kmeans_profile_layout')
  clocks = torch.zeros((n_pass.value, wgs.value, 2), dtype=torch.int64, device=x.device)
  labels = torch.empty((p,), dtype=torch.int64, device=x.device)
  ws = workspace(lib().spml_kmeans_workspace_bytes(p, d, k, n_img, max_seg_len), x.device)
  check(lib().spml_kmeans_run_profiled_f32(
      ptr(x, torch.float32), p, d, ptr(seg_offsets, torch.int64), n_img, int(max_seg_len), k,
      ptr(labels_init, torch.int64), int(iterations), ptr(labels), int(flags), ptr(ws), ws.numel(),
      ptr(clocks), clocks.numel(), stream_ptr()), 'spml_kmeans_run_profiled_f32')
  _note_kmeans(p, d, k, n_img, max_seg_len, iterations, 0, flags)
  c = clocks.cpu()
  # (passes may run fewer workgroups than the layout's maximum: their stamps stay zero)
  start = torch.where(c[:, :, 0] == 0, torch.full_like(c[:, :, 0], torch.iinfo(torch.int64).max), c[:, :, 0])
  dur = (c[:, :, 1].max(dim=1).values - start.min(dim=1).values).double() * 0.01
  return labels, dur


def kmeans_assign(x, seg_offsets, max_seg_len, centroids, flags=0):
  p, d = x.shape
  n_img = seg_offsets.shape[0] - 1
  k = centroids.shape[-2]
  labels = torch.empty((p,), dtype=torch.int64, device=x.device)
  nbytes = lib().spml_kmeans_workspace_bytes(p, d, k, n_img, max_seg_len)
  ws = workspace(nbytes, x.device)
  check(lib().spml_kmeans_assign_f32(
      ptr(x, torch.float32), p, d, ptr(seg_offsets, torch.int64), n_img, int(max_seg_len), k,
      ptr(centroids, torch.float32), ptr(labels), int(flags), ptr(ws), ws.numel(),
      stream_ptr()), 'spml_kmeans_assign_f32')
  _note_kmeans(p, d, k, n_img, max_seg_len, 1, 1, flags)
  return labels


def kmeans_workspace(x, seg_offsets, max_seg_len, k):
  p, d = x.shape
  return workspace(lib().spml_kmeans_workspace_bytes(p, d, k, seg_offsets.shape[0] - 1, max_seg_len),
                   x.device)


def kmeans_preconvert(x, seg_offsets, max_seg_len, k, ws):
  """X -> split-f16 tiles inside `ws` (for kmeans_fused_pass(..., preconverted=True))."""
  p, d = x.shape
  check(lib().spml_kmeans_preconvert_f32(
      ptr(x, torch.float32), p, d, ptr(seg_offsets, torch.int64), seg_offsets.shape[0] - 1,
      int(max_seg_len), int(k), ptr(ws), ws.numel(), stream_ptr()), 'spml_kmeans_preconvert_f32')


def kmeans_fused_pass(x, seg_offsets, max_seg_len, centroids, ws=None, preconverted=False, flags=0,
                      out=None):
  """One fused pass: (labels [P] int64, raw sums [n_img,K,D] of X by the new labels)."""
  p, d = x.shape
  n_img = seg_offsets.shape[0] - 1
  k = centroids.shape[-2]
  if ws is None:
    ws = kmeans_workspace(x, seg_offsets, max_seg_len, k)
  if out is None:
    out = (torch.empty((p,), dtype=torch.int64, device=x.device),
           torch.empty((n_img, k, d), dtype=torch.float32, device=x.device))
  labels, sums = out
  flags = int(flags) | (32 if preconverted else 0)
  check(lib().spml_kmeans_fused_pass_f32(
      ptr(x, torch.float32), p, d, ptr(seg_offsets, torch.int64), n_img, int(max_seg_len), k,
      ptr(centroids, torch.float32), ptr(labels, torch.int64), ptr(sums, torch.float32), flags,
      ptr(ws), ws.numel(), stream_ptr()), 'spml_kmeans_fused_pass_f32')
  _note_kmeans(p, d, k, n_img, max_seg_len, 1, 1, flags)
  return labels, sums


def clock_probe(device, spin_us, stream):
  """Launches the clock probe on `stream` (a torch.cuda.Stream); returns the device tensor [cycles, 100-MHz ticks]
  (read it after synchronising): shader clock in MHz = 100 * cycles / ticks."""
  out = torch.zeros((2,), dtype=torch.int64, device=device)
  # the probe starts where the current stream stands now (behind the zero fill of `out` and everything queued in front
  # of it), not at once: it measures the clock of what the caller launches NEXT on the current stream
  stream.wait_stream(torch.cuda.current_stream(device))
  check(lib().spml_clock_probe(ptr(out), int(spin_us), c_void_p(stream.cuda_stream)), 'spml_clock_probe')
  return out


def maxpool3x3s2_nhwc(x):
  """nn.MaxPool2d(3, 2, 1) of a channels-last fp32 map (forward only) -> channels-last map."""
  n, c, h, w = x.shape
  oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
  y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
  check(lib().spml_maxpool3x3s2_nhwc_f32(_ptr_any(x), n, h, w, c, _ptr_any(y), stream_ptr()), 'spml_maxpool3x3s2_nhwc_f32')
  return y


def segment_sum_normalize(x, ids, m):
  p, d = x.shape
  sums = torch.empty((m, d), dtype=torch.float32, device=x.device)
  protos = torch.empty((m, d), dtype=torch.float32, device=x.device)
  if deterministic():
    nbytes = lib().spml_segment_sum_det_workspace_bytes(int(m), d)
    ws = workspace(nbytes, x.device)
    check(lib().spml_segment_sum_normalize_det_f32(ptr(x, torch.float32), ptr(ids, torch.int64), p, d, int(m),
                                                   ptr(sums), ptr(protos), ptr(ws), ws.numel(), stream_ptr()),
          'spml_segment_sum_normalize_det_f32')
    return protos, sums
  check(lib().spml_segment_sum_normalize_f32(ptr(x, torch.float32), ptr(ids, torch.int64), p, d,
                                             int(m), ptr(sums), ptr(protos), stream_ptr()),
        'spml_segment_sum_normalize_f32')
  return protos, sums


def segment_sum_normalize_bwd(d_protos, sums, ids, p):
  m, d = sums.shape
  scratch = torch.empty_like(sums)
  dx = torch.empty((p, d), dtype=torch.float32, device=sums.device)
  check(lib().spml_segment_sum_normalize_bwd_f32(
      ptr(d_protos, torch.float32), ptr(sums, torch.float32), ptr(ids, torch.int64), p, d, m,
      ptr(scratch), ptr(dx), 0, stream_ptr()), 'spml_segment_sum_normalize_bwd_f32')
  return dx


def segsort_nll_fwd(emb, own, px_code, protos, pr_code, kappa, mode):
  p, d = emb.shape
  m = protos.shape[0]
  nll = torch.empty((p,), dtype=torch.float32, device=emb.device)
  stats = torch.empty((p, 4), dtype=torch.float32, device=emb.device)
  ws = workspace(lib().spml_segsort_nll_workspace_bytes(p, m, d), emb.device)
  check(lib().spml_segsort_nll_fwd_f32(
      ptr(emb, torch.float32), ptr(own, torch.int64), ptr(px_code, torch.int64), p,
      ptr(protos, torch.float32), ptr(pr_code, torch.int64), m, d, float(kappa), int(mode),
      ptr(nll), ptr(stats), ptr(ws), ws.numel(), stream_ptr()), 'spml_segsort_nll_fwd_f32')
  return nll, stats


def segsort_nll_bwd(emb, own, px_code, protos, pr_code, kappa, mode, stats, d_nll, m_grad=-1):
  p, d = emb.shape
  m = protos.shape[0]
  d_emb = torch.empty_like(emb)
  d_protos = torch.zeros_like(protos)
  ws = workspace(lib().spml_segsort_nll_workspace_bytes(p, m, d), emb.device)
  check(lib().spml_segsort_nll_bwd_f32(
      ptr(emb, torch.float32), ptr(own, torch.int64), ptr(px_code, torch.int64), p,
      ptr(protos, torch.float32), ptr(pr_code, torch.int64), m, d, float(kappa), int(mode),
      ptr(stats, torch.float32), ptr(d_nll, torch.float32), ptr(d_emb), ptr(d_protos),
      int(m_grad), ptr(ws), ws.numel(), stream_ptr()), 'spml_segsort_nll_bwd_f32')
  return d_emb, d_protos


def topk_affinity(q, protos, k, q_group=None, pr_group=None, pr_valid=None, masked_value=-2.0):
  nq, d = q.shape
  m = protos.shape[0]
  idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
  val = torch.empty((nq, k), dtype=torch.float32, device=q.device)
  ws = workspace(lib().spml_topk_workspace_bytes(nq, m, d, k), q.device)
  check(lib().spml_topk_affinity_f32(
      ptr(q, torch.float32), nq, ptr(protos, torch.float32), m, d, int(k),
      ptr(q_group, torch.int64, True), ptr(pr_group, torch.int64, True),
      ptr(pr_valid, torch.uint8, True), float(masked_value), ptr(idx), ptr(val), ptr(ws),
      ws.numel(), stream_ptr()), 'spml_topk_affinity_f32')
  return idx, val


def window_accumulate(patch, acc, counts, sh, sw):
  """acc[:, sh:sh+h, sw:sw+w] += patch / |patch|_channels ; counts[window] += 1 (in place)."""
  c, h, w = patch.shape
  big_c, big_h, big_w = acc.shape
  if big_c != c or tuple(counts.shape) != (big_h, big_w):
    raise SpmlHipError('window_accumulate: shape mismatch')
  check(lib().spml_window_accumulate_f32(
      ptr(patch, torch.float32), c, h, w, ptr(acc, torch.float32), ptr(counts, torch.float32),
      big_h, big_w, int(sh), int(sw), stream_ptr()), 'spml_window_accumulate_f32')


def affinity_transition(emb, scale=5.0, power=20):
  """emb [B,C,n] (unit columns) -> column-stochastic transition matrix [n,n]."""
  b, c, n = emb.shape
  trans = torch.empty((n, n), dtype=torch.float32, device=emb.device)
  ws = workspace(lib().spml_affinity_workspace_bytes(b, c, n), emb.device)
  check(lib().spml_affinity_transition_f32(
      ptr(emb, torch.float32), b, c, n, float(scale), int(power), ptr(trans), ptr(ws), ws.numel(),
      stream_ptr()), 'spml_affinity_transition_f32')
  return trans


# ---------------------------------------------------------------------------
# fused batch norm + ReLU + residual (channels-last [R, C] views of NHWC tensors)
def _nhwc_rows(t):
  """(R, C) of a 4-D channels-last tensor (or a 2-D [R, C] one)."""
  if t.dim() == 2:
    return t.shape[0], t.shape[1]
  n, c, h, w = t.shape
  if not t.is_contiguous(memory_format=torch.channels_last):
    raise SpmlHipError('fused batch norm needs channels-last (NHWC) tensors')
  return n * h * w, c


def _ptr_any(t, allow_none=False):
  """data_ptr of a CUDA fp32 tensor that is dense in memory (NCHW-contiguous or channels-last)."""
  if t is None:
    if allow_none:
      return c_void_p(0)
    raise SpmlHipError('missing tensor argument')
  if not t.is_cuda or t.dtype != torch.float32:
    raise SpmlHipError('expected a float32 GPU tensor (got %s on %s)' % (t.dtype, t.device))
  if t.device.index != torch.cuda.current_device():
    raise SpmlHipError('tensor lives on %s but the current device is cuda:%d' % (t.device, torch.cuda.current_device()))
  return c_void_p(t.data_ptr())


_bn_ws = {}


def _bn_workspace(r, c, device):
  """Cached scratch for the batch-norm partial sums (stream-ordered reuse: every call consumes its
  partials before the next one on the same stream overwrites them)."""
  need = lib().spml_bn_workspace_bytes(r, c)
  key = (device.index, torch.cuda.current_stream().cuda_stream)
  ws = _bn_ws.get(key)
  if ws is None or ws.numel() < need:
    ws = workspace(need, device)
    _bn_ws[key] = ws
  return ws


def bn_act_fwd(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu):
  """Single-rank fused forward -> (y, mean, invstd); updates the running statistics in place."""
  r, c = _nhwc_rows(x)
  y = torch.empty_like(x)
  stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
  ws = _bn_workspace(r, c, x.device)
  check(lib().spml_bn_act_fwd_f32(
      _ptr_any(x), _ptr_any(residual, True), r, c, ptr(gamma, torch.float32), ptr(beta, torch.float32),
      ptr(running_mean, None, True), ptr(running_var, None, True), float(momentum), float(eps),
      int(bool(relu)), _ptr_any(y), c_void_p(stats[0].data_ptr()), c_void_p(stats[1].data_ptr()), ptr(ws),
      ws.numel(), stream_ptr()), 'spml_bn_act_fwd_f32')
  return y, stats[0], stats[1]


def bn_act_bwd(dy, y, x, mean, invstd, gamma, want_dx=True, want_dres=False):
  """Single-rank fused backward -> (dx, d_residual, d_gamma, d_beta)."""
  r, c = _nhwc_rows(dy)
  dx = torch.empty_like(dy) if want_dx else None
  dres = torch.empty_like(dy) if want_dres else None
  dgb = torch.empty((2, c), dtype=torch.float32, device=dy.device)
  ws = _bn_workspace(r, c, dy.device)
  check(lib().spml_bn_act_bwd_f32(
      _ptr_any(dy), _ptr_any(y, True), _ptr_any(x), r, c, c_void_p(mean.data_ptr()), c_void_p(invstd.data_ptr()),
      ptr(gamma, torch.float32), c_void_p(dgb[0].data_ptr()), c_void_p(dgb[1].data_ptr()), _ptr_any(dx, True),
      _ptr_any(dres, True), ptr(ws), ws.numel(), stream_ptr()), 'spml_bn_act_bwd_f32')
  return dx, dres, dgb[0], dgb[1]


def bn_stats(x):
  r, c = _nhwc_rows(x)
  mean = torch.empty((c,), dtype=torch.float32, device=x.device)
  m2 = torch.empty((c,), dtype=torch.float32, device=x.device)
  ws = workspace(lib().spml_bn_workspace_bytes(r, c), x.device)
  check(lib().spml_bn_stats_f32(_ptr_any(x), r, c, ptr(mean), ptr(m2), ptr(ws), ws.numel(), stream_ptr()),
        'spml_bn_stats_f32')
  return mean, m2


def bn_act_apply(x, residual, mean, invstd, gamma, beta, relu):
  r, c = _nhwc_rows(x)
  y = torch.empty_like(x)
  check(lib().spml_bn_act_apply_f32(_ptr_any(x), _ptr_any(residual, True), r, c, ptr(mean), ptr(invstd),
                                    ptr(gamma, torch.float32), ptr(beta, torch.float32), int(bool(relu)),
                                    _ptr_any(y), stream_ptr()), 'spml_bn_act_apply_f32')
  return y


def bn_act_bwd_reduce(dy, y, x, mean, invstd):
  r, c = _nhwc_rows(x)
  s0 = torch.empty((c,), dtype=torch.float32, device=x.device)
  s1 = torch.empty((c,), dtype=torch.float32, device=x.device)
  ws = workspace(lib().spml_bn_workspace_bytes(r, c), x.device)
  check(lib().spml_bn_act_bwd_reduce_f32(_ptr_any(dy), _ptr_any(y, True), _ptr_any(x), r, c, ptr(mean),
                                         ptr(invstd), ptr(s0), ptr(s1), ptr(ws), ws.numel(), stream_ptr()),
        'spml_bn_act_bwd_reduce_f32')
  return s0, s1


def bn_act_bwd_apply(dy, y, x, mean, invstd, gamma, sum_dz, sum_dz_xhat, count, want_dx=True,
                     want_dres=False):
  """count: rows the statistics were pooled over -- a number, or the device float [1] that
  bn_finalize_ranks wrote (SyncBatchNorm: the ranks' row counts may differ)."""
  r, c = _nhwc_rows(dy)
  count, count_dev = _count_args(count)
  dx = torch.empty_like(dy) if want_dx else None
  dres = torch.empty_like(dy) if want_dres else None
  check(lib().spml_bn_act_bwd_apply_f32(
      _ptr_any(dy), _ptr_any(y, True), _ptr_any(x, True), r, c, ptr(mean, None, True), ptr(invstd, None, True),
      ptr(gamma, None, True), ptr(sum_dz, None, True), ptr(sum_dz_xhat, None, True), count, count_dev,
      _ptr_any(dx, True), _ptr_any(dres, True), stream_ptr()), 'spml_bn_act_bwd_apply_f32')
  return dx, dres


# ---------------------------------------------------------------------------
# split-f16 ("hl8") tensors and the matrix-core convolutions (csrc/conv.hip)
# ---------------------------------------------------------------------------

class Hl8(object):
  """Split-f16 copy of an fp32 tensor [rows, C]: `data` (uint8, rows*C*4 bytes) + the device
  float `bound` (>= max|v|, fixes the power-of-two scale) -- see include/spml_hip.h."""
  __slots__ = ('data', 'bound', 'rows', 'channels')

  def __init__(self, data, bound, rows, channels):
    self.data, self.bound, self.rows, self.channels = data, bound, rows, channels


def hl8_from_f32(x, rows=None, channels=None, bound=None):
  """x: dense fp32 GPU tensor read as [rows, channels] (default: channels-last 4-D or 2-D)."""
  if rows is None:
    rows, channels = _nhwc_rows(x)
  compute = bound is None
  if compute:
    bound = torch.empty((1,), dtype=torch.float32, device=x.device)
  out = torch.empty((rows * channels * 4,), dtype=torch.uint8, device=x.device)
  check(lib().spml_hl8_from_f32(_ptr_any(x), rows, channels, c_void_p(bound.data_ptr()), int(compute),
                                ptr(out), stream_ptr()), 'spml_hl8_from_f32')
  return Hl8(out, bound, rows, channels)


def hl8_weight(w):
  """Conv weight [Cout, Cin, kh, kw] -> (forward operand [Cout][taps*Cin], data-gradient operand
  [Cin][taps*Cout] with mirrored taps); both share one scale."""
  cout, cin, kh, kw = w.shape
  taps = kh * kw
  wl = w.detach().contiguous(memory_format=torch.channels_last)          # [Cout][kh][kw][Cin] storage
  fwd = hl8_from_f32(wl, cout, taps * cin)
  tr = torch.empty((cout * taps * cin * 4,), dtype=torch.uint8, device=w.device)
  check(lib().spml_hl8_weight_transposed_f32(_ptr_any(wl), cout, taps, cin, c_void_p(fwd.bound.data_ptr()),
                                             ptr(tr), stream_ptr()), 'spml_hl8_weight_transposed_f32')
  return fwd, Hl8(tr, fwd.bound, cin, taps * cout)


def conv_hl8_supported(k, n, taps):
  return bool(lib().spml_conv_hl8_supported(int(k), int(n), int(taps)))


def conv_hl8(a, b, n_img, h, w, taps, dilation=1, addend=None, addend_mask=None):
  """out [n_img, N, h, w] (channels-last fp32) = conv(a, b): a Hl8 [n_img*h*w, K], b Hl8 [N, taps*K]."""
  k, n = a.channels, b.rows
  if a.rows != n_img * h * w or b.channels != taps * k:
    raise SpmlHipError('conv_hl8: operand shapes do not match')
  out = torch.empty((n_img, n, h, w), dtype=torch.float32, device=a.data.device,
                    memory_format=torch.channels_last)
  check(lib().spml_conv_hl8_f32(ptr(a.data), c_void_p(a.bound.data_ptr()), ptr(b.data),
                                c_void_p(b.bound.data_ptr()), _ptr_any(addend, True), _dp(addend_mask), _ptr_any(out),
                                n_img, h, w,
                                k, n, taps, dilation, stream_ptr()), 'spml_conv_hl8_f32')
  return out


class ChunkStats(object):
  """Per-row-tile column statistics a convolution's epilogue left for the batch norm that follows:
  data [4, chunks, N] (mean, M2, max, min), chunk geometry, and the bound slot the launch reset."""

  def __init__(self, data, chunks, chunk_rows, bound):
    self.data, self.chunks, self.chunk_rows, self.bound = data, chunks, chunk_rows, bound


def conv_hl8_stats(a, b, n_img, h, w, taps, dilation=1):
  """conv_hl8 whose epilogue also leaves the chunk statistics of the output -> (out, ChunkStats), or
  (out, None) where the launch has no per-tile column owner (narrow outputs).  SPML_CONV_BN_STATS=0: never."""
  import ctypes
  k, n = a.channels, b.rows
  chunks, rows = ctypes.c_int(0), ctypes.c_int(0)
  if os.environ.get('SPML_CONV_BN_STATS') == '0' or lib().spml_conv_hl8_stats_layout(
      n_img, h, w, k, n, taps, ctypes.byref(chunks), ctypes.byref(rows)) != 0:
    return conv_hl8(a, b, n_img, h, w, taps, dilation), None
  if a.rows != n_img * h * w or b.channels != taps * k:
    raise SpmlHipError('conv_hl8_stats: operand shapes do not match')
  dev = a.data.device
  out = torch.empty((n_img, n, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
  st = torch.empty((4, chunks.value, n), dtype=torch.float32, device=dev)
  bound = _f32(1, dev)
  check(lib().spml_conv_hl8_stats_f32(ptr(a.data), c_void_p(a.bound.data_ptr()), ptr(b.data),
                                      c_void_p(b.bound.data_ptr()), _ptr_any(out), _dp(st), _dp(bound), n_img, h, w,
                                      k, n, taps, dilation, stream_ptr()), 'spml_conv_hl8_stats_f32')
  return out, ChunkStats(st, chunks.value, rows.value, bound)


def conv_wgrad_hl8_supported(k, n, taps):
  return bool(lib().spml_conv_wgrad_hl8_supported(int(k), int(n), int(taps)))


_wgrad_ws = {}


def conv_wgrad_hl8(dy, x, n_img, h, w, taps, dilation=1):
  """dW [N, K, kh, kw] (channels-last storage [N][taps][K]) from dy Hl8 [R, N] and x Hl8 [R, K]."""
  n, k = dy.channels, x.channels
  if dy.rows != n_img * h * w or x.rows != dy.rows:
    raise SpmlHipError('conv_wgrad_hl8: operand shapes do not match')
  need = lib().spml_conv_wgrad_workspace_bytes(n_img, h, w, k, n, taps)
  key = (dy.data.device.index, torch.cuda.current_stream().cuda_stream)
  ws = _wgrad_ws.get(key)
  if ws is None or ws.numel() < need:
    ws = workspace(need, dy.data.device)
    _wgrad_ws[key] = ws
  side = 3 if taps == 9 else 1
  dw = torch.empty((n, k, side, side), dtype=torch.float32, device=dy.data.device,
                   memory_format=torch.channels_last)
  check(lib().spml_conv_wgrad_hl8_f32(ptr(dy.data), c_void_p(dy.bound.data_ptr()), ptr(x.data),
                                      c_void_p(x.bound.data_ptr()), _ptr_any(dw), n_img, h, w, k, n, taps,
                                      dilation, ptr(ws), ws.numel(), stream_ptr()), 'spml_conv_wgrad_hl8_f32')
  return dw


def conv_wgrad_pyramid_hl8_supported(k, n, branches):
  return bool(lib().spml_conv_wgrad_pyramid_hl8_supported(int(k), int(n), int(branches)))


def conv_wgrad_pyramid_hl8(dy, x, n_img, h, w, dilations):
  """The weight gradients of up to four dilated 3x3 branches (64 output channels each) that share the output
  gradient dy Hl8 [R, 64], from x Hl8 [R, K]: a list of dW [64, K, 3, 3] (channels-last storage), views of one
  [branches, 64, 9, K] buffer written by one launch."""
  n, k, nb = dy.channels, x.channels, len(dilations)
  if dy.rows != n_img * h * w or x.rows != dy.rows:
    raise SpmlHipError('conv_wgrad_pyramid_hl8: operand shapes do not match')
  need = lib().spml_conv_wgrad_pyramid_workspace_bytes(n_img, h, w, k, n, nb)
  if need == 0:
    raise SpmlHipError('conv_wgrad_pyramid_hl8: unsupported shape (K %d, N %d, %d branches)' % (k, n, nb))
  key = (dy.data.device.index, torch.cuda.current_stream().cuda_stream)
  ws = _wgrad_ws.get(key)
  if ws is None or ws.numel() < need:
    ws = workspace(need, dy.data.device)
    _wgrad_ws[key] = ws
  dw = torch.empty((nb, n, 3, 3, k), dtype=torch.float32, device=dy.data.device)
  dil = (ctypes.c_int * nb)(*[int(d) for d in dilations])
  check(lib().spml_conv_wgrad_pyramid_hl8_f32(ptr(dy.data), c_void_p(dy.bound.data_ptr()), ptr(x.data),
                                              c_void_p(x.bound.data_ptr()), _ptr_any(dw), n_img, h, w, k, n, nb,
                                              ctypes.cast(dil, c_void_p), ptr(ws), ws.numel(), stream_ptr()),
        'spml_conv_wgrad_pyramid_hl8_f32')
  return [dw[b].permute(0, 3, 1, 2) for b in range(nb)]


def _f32(n, device):
  return torch.empty((n,), dtype=torch.float32, device=device)


def _dp(t):
  return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def bn_stats_ext(x, rows, channels, chunk_stats=None):
  """Local statistics of x read as [rows, channels] fp32 -> [5, C] = (count, mean, M2, channel max,
  channel min); rows 0..2 are what the ranks exchange.  chunk_stats: pooled from the statistics the
  producing convolution's epilogue left instead of reading x."""
  st = torch.empty((5, channels), dtype=torch.float32, device=x.device)
  st[0].fill_(float(rows))
  if chunk_stats is not None:
    check(lib().spml_bn_stats_ext_chunks_f32(_dp(chunk_stats.data), chunk_stats.chunks, chunk_stats.chunk_rows, rows,
                                             channels, _dp(st[1]), _dp(st[2]), _dp(st[3]), _dp(st[4]), stream_ptr()),
          'spml_bn_stats_ext_chunks_f32')
    return st
  ws = _bn_workspace(rows, channels, x.device)
  check(lib().spml_bn_stats_ext_f32(_ptr_any(x), rows, channels, _dp(st[1]), _dp(st[2]), _dp(st[3]), _dp(st[4]),
                                    ptr(ws), ws.numel(), stream_ptr()), 'spml_bn_stats_ext_f32')
  return st


def _count_args(count):
  """(host count, device count pointer) of the batch-norm backward apply calls."""
  if torch.is_tensor(count):
    return 0.0, ptr(count, torch.float32)
  return float(count), None


def bn_finalize_ranks(gathered, eps, momentum, running_mean, running_var):
  """gathered [world, 3, C] (count, mean, M2 per rank) -> pooled (mean, invstd, total row count [1], a
  device float: the ranks' counts are never read on the host); running statistics updated."""
  world, _, c = gathered.shape
  out = torch.empty((2 * c + 4,), dtype=torch.float32, device=gathered.device)
  mean, invstd, total = out[:c], out[c:2 * c], out[2 * c:2 * c + 1]
  check(lib().spml_bn_finalize_ranks_f32(ptr(gathered, torch.float32), world, c, float(eps), float(momentum),
                                         _dp(running_mean), _dp(running_var), _dp(mean), _dp(invstd), _dp(total),
                                         stream_ptr()),
        'spml_bn_finalize_ranks_f32')
  return mean, invstd, total


def bn_finalize(mean, m2, count, eps, momentum, running_mean, running_var):
  invstd = torch.empty_like(mean)
  check(lib().spml_bn_finalize_f32(_dp(mean), _dp(m2), mean.numel(), float(count), float(eps), float(momentum),
                                   _dp(running_mean), _dp(running_var), _dp(invstd), stream_ptr()),
        'spml_bn_finalize_f32')
  return invstd


def bn_act_apply_hl8(x, rows, channels, residual, residual_bound, mean, invstd, gamma, beta, cmax, cmin, relu,
                     want_f32, want_hl8, like=None, want_mask=False):
  """-> (y fp32 or None, Hl8 or None, bound, ReLU mask bytes or None)"""
  y = torch.empty_like(x if like is None else like) if want_f32 else None
  yh = torch.empty((rows * channels * 4,), dtype=torch.uint8, device=x.device) if want_hl8 else None
  bound = _f32(1, x.device)
  mask = torch.empty((rows * (channels // 4),), dtype=torch.uint8, device=x.device) if want_mask else None
  check(lib().spml_bn_act_apply_hl8_f32(
      _ptr_any(x), _ptr_any(residual, True), _dp(residual_bound), rows, channels, _dp(mean), _dp(invstd),
      ptr(gamma, torch.float32), ptr(beta, torch.float32), _dp(cmax), _dp(cmin), int(bool(relu)), _ptr_any(y, True),
      _dp(yh), _dp(bound), _dp(mask), stream_ptr()), 'spml_bn_act_apply_hl8_f32')
  return y, (Hl8(yh, bound, rows, channels) if want_hl8 else None), bound, mask


def bn_act_bwd_reduce_ext(dy, y, relu_mask, x, rows, channels, mean, invstd):
  """-> (sum dz, sum dz*xhat, max|dz| per channel); ReLU mask from y (fp32), the mask bytes or none."""
  st = torch.empty((3, channels), dtype=torch.float32, device=dy.device)     # rows 0, 1 are all-reduced together
  ws = _bn_workspace(rows, channels, dy.device)
  check(lib().spml_bn_act_bwd_reduce_ext_f32(
      _ptr_any(dy), _ptr_any(y, True), _dp(relu_mask), _ptr_any(x), rows, channels,
      _dp(mean), _dp(invstd), _dp(st[0]), _dp(st[1]), _dp(st[2]), ptr(ws), ws.numel(), stream_ptr()),
        'spml_bn_act_bwd_reduce_ext_f32')
  return st


def bn_act_bwd_apply_hl8(dy, y, relu_mask, x, rows, channels, mean, invstd, gamma, s0, s1, max_dz, cmax, cmin, count,
                         want_dx_f32=False, want_dx_hl8=True, want_dres=False):
  """-> (dx fp32 or None, dx Hl8 or None, d_residual fp32 or None)"""
  dx = torch.empty_like(dy) if want_dx_f32 else None
  dxh = torch.empty((rows * channels * 4,), dtype=torch.uint8, device=dy.device) if want_dx_hl8 else None
  bound = _f32(1, dy.device) if want_dx_hl8 else None
  dres = torch.empty_like(dy) if want_dres else None
  count, count_dev = _count_args(count)
  check(lib().spml_bn_act_bwd_apply_hl8_f32(
      _ptr_any(dy), _ptr_any(y, True), _dp(relu_mask), _ptr_any(x), rows, channels,
      _dp(mean), _dp(invstd), ptr(gamma, torch.float32), _dp(s0), _dp(s1), _dp(max_dz), _dp(cmax), _dp(cmin),
      count, count_dev, _ptr_any(dx, True), _dp(dxh), _dp(bound), _ptr_any(dres, True), stream_ptr()),
        'spml_bn_act_bwd_apply_hl8_f32')
  return dx, (Hl8(dxh, bound, rows, channels) if want_dx_hl8 else None), dres


def bn_fwd_hl8(x, rows, channels, residual, residual_bound, gamma, beta, running_mean, running_var, momentum, eps,
               relu, want_f32, want_hl8, want_mask, chunk_stats=None):
  """Single-rank batch norm forward -> (y fp32|None, Hl8|None, bound, mask|None, saved (mean, invstd, cmax, cmin))."""
  y = torch.empty_like(x) if want_f32 else None
  yh = torch.empty((rows * channels * 4,), dtype=torch.uint8, device=x.device) if want_hl8 else None
  bound = _f32(1, x.device) if chunk_stats is None else chunk_stats.bound
  mask = torch.empty((rows * (channels // 4),), dtype=torch.uint8, device=x.device) if want_mask else None
  st = torch.empty((4, channels), dtype=torch.float32, device=x.device)
  if chunk_stats is not None:              # the producing convolution already took the statistics
    check(lib().spml_bn_fwd_hl8_chunks_f32(
        _ptr_any(x), _dp(chunk_stats.data), chunk_stats.chunks, chunk_stats.chunk_rows, _ptr_any(residual, True),
        _dp(residual_bound), rows, channels, ptr(gamma, torch.float32), ptr(beta, torch.float32), _dp(running_mean),
        _dp(running_var), float(momentum), float(eps), int(bool(relu)), _ptr_any(y, True), _dp(yh), _dp(bound),
        _dp(mask), _dp(st[0]), _dp(st[1]), _dp(st[2]), _dp(st[3]), stream_ptr()), 'spml_bn_fwd_hl8_chunks_f32')
    return y, (Hl8(yh, bound, rows, channels) if want_hl8 else None), bound, mask, (st[0], st[1], st[2], st[3])
  ws = _bn_workspace(rows, channels, x.device)
  check(lib().spml_bn_fwd_hl8_f32(
      _ptr_any(x), _ptr_any(residual, True), _dp(residual_bound), rows, channels, ptr(gamma, torch.float32),
      ptr(beta, torch.float32), _dp(running_mean), _dp(running_var), float(momentum), float(eps), int(bool(relu)),
      _ptr_any(y, True), _dp(yh), _dp(bound), _dp(mask), _dp(st[0]), _dp(st[1]), _dp(st[2]), _dp(st[3]), ptr(ws),
      ws.numel(), stream_ptr()), 'spml_bn_fwd_hl8_f32')
  return y, (Hl8(yh, bound, rows, channels) if want_hl8 else None), bound, mask, (st[0], st[1], st[2], st[3])


def bn_bwd_hl8(dy, relu_mask, x, rows, channels, saved, gamma, want_dres=False):
  """Single-rank batch norm backward -> (dx Hl8, d_residual|None, d_gamma, d_beta)."""
  mean, invstd, cmax, cmin = saved
  dxh = torch.empty((rows * channels * 4,), dtype=torch.uint8, device=dy.device)
  bound = _f32(1, dy.device)
  dres = torch.empty_like(dy) if want_dres else None
  dgb = torch.empty((2, channels), dtype=torch.float32, device=dy.device)
  ws = _bn_workspace(rows, channels, dy.device)
  check(lib().spml_bn_bwd_hl8_f32(
      _ptr_any(dy), c_void_p(0), _dp(relu_mask), _ptr_any(x), rows, channels, _dp(mean), _dp(invstd),
      ptr(gamma, torch.float32), _dp(cmax), _dp(cmin), _dp(dgb[0]), _dp(dgb[1]), c_void_p(0), _dp(dxh), _dp(bound),
      _ptr_any(dres, True), ptr(ws), ws.numel(), stream_ptr()), 'spml_bn_bwd_hl8_f32')
  return Hl8(dxh, bound, rows, channels), dres, dgb[0], dgb[1]


def conv_hl8_pyramid_dgrad(dy, weights, dilations, n_img, h, w):
  """dX of `sum_g conv2d(x, weights[g], dilation=dilations[g], padding=dilations[g])` for 3x3 weights
  [Cout, Cin, 3, 3] sharing the output gradient dy (Hl8 [R, Cout]) -> fp32 [n_img, Cin, h, w]."""
  groups = len(weights)
  cout, cin = weights[0].shape[0], weights[0].shape[1]
  dev = dy.data.device
  bound = torch.empty((1,), dtype=torch.float32, device=dev)
  wl = [wt.detach().contiguous(memory_format=torch.channels_last) for wt in weights]
  for g, wt in enumerate(wl):
    check(lib().spml_absmax_bound_f32(_ptr_any(wt), wt.numel(), _dp(bound), int(g == 0), stream_ptr()),
          'spml_absmax_bound_f32')
  b = torch.empty((cin * 9 * groups * cout * 4,), dtype=torch.uint8, device=dev)
  for g, wt in enumerate(wl):
    check(lib().spml_hl8_weight_transposed_into_f32(_ptr_any(wt), cout, 9, cin, _dp(bound), ptr(b), 9 * groups, 9 * g,
                                                    stream_ptr()), 'spml_hl8_weight_transposed_into_f32')
  out = torch.empty((n_img, cin, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
  import ctypes
  dil = (ctypes.c_int * 4)(*([int(d) for d in dilations] + [1] * (4 - groups)))
  check(lib().spml_conv_hl8_pyramid_f32(ptr(dy.data), _dp(dy.bound), ptr(b), _dp(bound), c_void_p(0), c_void_p(0),
                                        _ptr_any(out), n_img, h, w, cout, cin, groups, dil, stream_ptr()),
        'spml_conv_hl8_pyramid_f32')
  return out


def conv_hl8_pyramid_forward(x, weights, biases, dilations, n_img, h, w):
  """sum_g conv2d(x, weights[g], biases[g], dilation = padding = dilations[g]) for 3x3 weights
  [Cout, Cin, 3, 3] in one launch (x: Hl8 [R, Cin]) -> fp32 [n_img, Cout, h, w]."""
  import ctypes
  groups = len(weights)
  cout, cin = weights[0].shape[0], weights[0].shape[1]
  dev = x.data.device
  # forward operand [Cout][36 taps * Cin]: the branches' channels-last weights side by side
  cat = torch.cat([wt.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin) for wt in weights], dim=1).contiguous()
  b = hl8_from_f32(cat, cout, 9 * groups * cin)
  bias = None
  if any(bb is not None for bb in biases):
    bias = sum(bb.detach() for bb in biases if bb is not None).contiguous()
  out = torch.empty((n_img, cout, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
  dil = (ctypes.c_int * 4)(*([int(d) for d in dilations] + [1] * (4 - groups)))
  check(lib().spml_conv_hl8_pyramid_f32(ptr(x.data), _dp(x.bound), ptr(b.data), _dp(b.bound), _dp(bias), c_void_p(0),
                                        _ptr_any(out), n_img, h, w, cin, cout, groups, dil, stream_ptr()),
        'spml_conv_hl8_pyramid_f32')
  return out


_pyramid_operand_cache = {}     # the last packed operand of conv_hl8_pyramid_forward_gemm (one head per model)


def conv_hl8_pyramid_forward_gemm_supported(cin, cout, groups):
  return (16 <= cout <= 1024 and cout & (cout - 1) == 0 and 1 <= groups <= 4 and
          conv_hl8_supported(cin, 9 * groups * cout, 1))


def conv_hl8_pyramid_forward_gemm(x, weights, biases, dilations, n_img, h, w):
  """The same sum of dilated 3x3 branches as `conv_hl8_pyramid_forward`, for narrow heads: ONE 1x1 convolution
  with 9 * branches * Cout columns (z[r][tap * Cout + n] = x[r] . w_tap[n]: x is streamed once per 256 columns
  instead of once per tap) + `spml_conv_tap_gather_f32`, which adds every output pixel's taps from their shifted
  rows of z in a fixed order."""
  import ctypes
  groups = len(weights)
  cout, cin = weights[0].shape[0], weights[0].shape[1]
  dev = x.data.device
  # 1x1 operand [(branch, kh, kw, n)][Cin]: packed once per version of the weights (inference re-uses it; a training
  # step changes the weights in place -- `_version` moves -- and packs again)
  # (the key holds WEAK references to the weight tensors themselves: a pointer + version pair alone can recur for a
  # different tensor once the first one has been freed -- it did, in the test suite)
  import weakref
  refs, vers = _pyramid_operand_cache.get('refs', ()), _pyramid_operand_cache.get('versions', ())
  if (len(refs) == groups and all(r() is wt for r, wt in zip(refs, weights)) and
      vers == tuple(wt._version for wt in weights)):
    operand = _pyramid_operand_cache['operand']
  else:
    cat = torch.stack([wt.detach().permute(2, 3, 0, 1).reshape(9 * cout, cin) for wt in weights]).reshape(
        9 * groups * cout, cin).contiguous()
    operand = hl8_from_f32(cat, 9 * groups * cout, cin)
    _pyramid_operand_cache.update(refs=tuple(weakref.ref(wt) for wt in weights),
                                  versions=tuple(wt._version for wt in weights), operand=operand)
  # (z is [rows, 9 * branches * Cout] fp32 -- 623 MB at batch 16, 65 x 65, 2304 columns -- alive until the gather below)
  z = conv_hl8(x, operand, n_img, h, w, 1)
  bias = None
  if any(bb is not None for bb in biases):
    bias = sum(bb.detach() for bb in biases if bb is not None).contiguous()
  out = torch.empty((n_img, cout, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
  dil = (ctypes.c_int * groups)(*[int(d) for d in dilations])
  check(lib().spml_conv_tap_gather_f32(_ptr_any(z), _dp(bias), _ptr_any(out), n_img, h, w, cout, groups,
                                       ctypes.cast(dil, c_void_p), stream_ptr()), 'spml_conv_tap_gather_f32')
  return out


def hl8_weight_set(weights):
  """[(forward operand, data-gradient operand)] of up to four conv weights [Cout, Cin, kh, kw] in two
  launches (one unit's weights); every weight has its own bound / scale."""
  import ctypes
  n = len(weights)
  dev = weights[0].device
  wl = [wt.detach().contiguous(memory_format=torch.channels_last) for wt in weights]
  bounds = torch.empty((4,), dtype=torch.float32, device=dev)
  shapes = [(wt.shape[0], wt.shape[1], wt.shape[2] * wt.shape[3]) for wt in wl]
  fwd = [torch.empty((co * ci * t * 4,), dtype=torch.uint8, device=dev) for co, ci, t in shapes]
  tr = [torch.empty((co * ci * t * 4,), dtype=torch.uint8, device=dev) for co, ci, t in shapes]
  pv, iv = ctypes.c_void_p * n, ctypes.c_int * n
  check(lib().spml_hl8_weight_set_f32(
      pv(*[wt.data_ptr() for wt in wl]), iv(*[sh[0] for sh in shapes]), iv(*[sh[1] for sh in shapes]),
      iv(*[sh[2] for sh in shapes]), n, _dp(bounds), pv(*[t.data_ptr() for t in fwd]),
      pv(*[t.data_ptr() for t in tr]), stream_ptr()), 'spml_hl8_weight_set_f32')
  out = []
  for i, (co, ci, t) in enumerate(shapes):
    b = bounds[i:i + 1]
    out.append((Hl8(fwd[i], b, co, t * ci), Hl8(tr[i], b, ci, t * co)))
  return out


def conv_hl8_affine(a, b, bias, n_img, h, w, taps, dilation=1, addend=None, relu=True, want_hl8=True):
  """Inference: act(conv(a, b) + bias [+ addend]) -> (out fp32 channels-last, Hl8 of it or None)."""
  k, n = a.channels, b.rows
  if a.rows != n_img * h * w or b.channels != taps * k:
    raise SpmlHipError('conv_hl8_affine: operand shapes do not match')
  dev = a.data.device
  out = torch.empty((n_img, n, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
  bound = _f32(1, dev) if want_hl8 else None
  check(lib().spml_conv_hl8_affine_f32(ptr(a.data), _dp(a.bound), ptr(b.data), _dp(b.bound), ptr(bias, torch.float32),
                                       _ptr_any(addend, True), int(bool(relu)), _ptr_any(out), _dp(bound), n_img, h, w,
                                       k, n, taps, dilation, stream_ptr()), 'spml_conv_hl8_affine_f32')
  return out, (hl8_from_f32(out, bound=bound) if want_hl8 else None)


# ---------------------------------------------------------------------------
# softmax head: cross-entropy of bilinearly up-sampled logits
def upsample_ce_supported(c):
  return bool(lib().spml_upsample_ce_supported(int(c)))


def upsample_ce_fwd(logits_nhwc, labels, ignore_index):
  """logits_nhwc fp32 [N, h, w, C] contiguous, labels int64 [N, H, W] -> (result [3] = sum, count, mean;
  lse [N, H, W])."""
  n, h, w, c = logits_nhwc.shape
  hh, ww = labels.shape[-2:]
  lse = torch.empty((n, hh, ww), dtype=torch.float32, device=logits_nhwc.device)
  result = torch.empty((3,), dtype=torch.float32, device=logits_nhwc.device)
  ws = workspace(lib().spml_upsample_ce_workspace_bytes(n, hh, ww), logits_nhwc.device)
  check(lib().spml_upsample_ce_fwd_f32(ptr(logits_nhwc, torch.float32), ptr(labels, torch.int64), n, c, h, w, hh, ww,
                                       int(ignore_index), ptr(lse), ptr(result), ptr(ws), ws.numel(), stream_ptr()),
        'spml_upsample_ce_fwd_f32')
  return result, lse


def upsample_ce_bwd(logits_nhwc, labels, lse, ignore_index, scale):
  """-> d_logits [N, h, w, C] = scale[0] * d(sum of the pixel losses) / d logits."""
  n, h, w, c = logits_nhwc.shape
  hh, ww = labels.shape[-2:]
  d = torch.empty_like(logits_nhwc)
  check(lib().spml_upsample_ce_bwd_f32(ptr(logits_nhwc, torch.float32), ptr(labels, torch.int64),
                                       ptr(lse, torch.float32), n, c, h, w, hh, ww, int(ignore_index),
                                       ptr(scale, torch.float32), ptr(d), stream_ptr()), 'spml_upsample_ce_bwd_f32')
  return d
