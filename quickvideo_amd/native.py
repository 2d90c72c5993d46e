"""ctypes binding of libquickprefill.so (include/quickprefill.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every operator of the hot
path is a hand-written gfx950 kernel behind the C ABI.  There is NO fallback: if the library is
missing, or no MI355X is visible, importing the ops fails loudly (``QuickPrefillUnavailable``).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUICKPREFILL_LIB") or os.path.join(_HERE, "libquickprefill.so")   # override: kernel A/B builds (tools/)

QP_OK, QP_ERR_INVALID, QP_ERR_UNSUPPORTED, QP_ERR_HIP, QP_ERR_WORKSPACE, QP_ERR_TIMEOUT = 0, -1, -2, -3, -4, -5


class QuickPrefillUnavailable(RuntimeError):
    pass


class QuickPrefillError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libquickprefill status {status}: {msg}")
        self.status = status


_c = ctypes
_vp, _i64, _i32, _f32, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float, _c.c_size_t

class QpLayer(_c.Structure):
    """struct qp_layer (include/quickprefill.h): one decoder layer's weight and cache pointers."""
    _fields_ = [(n, _vp) for n in ("ln1", "w_qkv", "b_qkv", "w_o", "ln2", "w_gate_up", "w_down", "k_cache", "v_cache")]


class QpSegment(_c.Structure):
    """struct qp_segment (include/quickprefill.h)."""
    _fields_ = ([(n, _c.c_int32) for n in ("n_layers", "hidden", "n_q_heads", "n_kv_heads", "head_dim", "intermediate")] +
                [("rms_eps", _f32), ("attn_scale", _f32), ("n", _i64), ("cache_capacity", _i64), ("prune_mode", _c.c_int32),
                 ("attend_prefix", _c.c_int32), ("split_qkv", _i64), ("split_o", _i64), ("split_gate_up", _i64), ("split_down", _i64),
                 ("gate_up_two_gemms", _c.c_int32), ("reserved_", _c.c_int32)] +
                [(n, _vp) for n in ("h", "x", "qkv", "q", "att", "o", "gate_up", "act", "down", "k_stage", "v_stage", "norm_keys", "kept_idx")] +
                [("kept_idx_stride", _i64), ("cos", _vp), ("sin", _vp), ("attn_ws", _vp), ("attn_ws_bytes", _sz), ("gemm_ws", _vp),
                 ("gemm_ws_bytes", _sz), ("attn_events", _c.POINTER(_vp)), ("prune_events", _c.POINTER(_vp))])


class QpVitBlock(_c.Structure):
    """struct qp_vit_block (include/quickprefill.h)."""
    _fields_ = [(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_bias_scaled", "fc2_w", "fc2_b")]


# name -> (restype, argtypes): exactly the declarations of include/quickprefill.h
SIGNATURES = {
    "qp_vit_blocks": (_i32, [_vp, _c.POINTER(QpVitBlock), _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _sz, _vp]),
    "qp_prefill_segment": (_i32, [_vp, _c.POINTER(QpSegment), _c.POINTER(QpLayer), _c.POINTER(_i64), _c.POINTER(_i64), _vp]),
    "qp_create": (_i32, [_c.POINTER(_vp), _i32]),
    "qp_destroy": (None, [_vp]),
    "qp_last_error": (_c.c_char_p, []),
    "qp_version": (_c.c_char_p, []),
    "qp_device_cus": (_i32, [_vp]),
    "qp_host_memcpy": (_i32, [_vp, _vp, _sz, _i32]),
    "qp_mrope_table": (_i32, [_vp, _vp, _i64, _c.POINTER(_c.c_int32), _f32, _i32, _vp, _vp, _vp]),
    "qp_rope_append": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "qp_rope_append_keys": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    "qp_query_scores_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "qp_query_head_sums": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "qp_query_scores_from_head_sums": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "qp_query_scores": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "qp_norm_keys": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _vp]),
    "qp_prune_keys": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i64, _vp, _vp]),
    "qp_prefill_attn": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "qp_prefill_attn_rows": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "qp_attn_workspace_bytes": (_sz, [_vp, _i64, _i64, _i32, _i32]),
    "qp_key_sumsq": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "qp_select_workspace_bytes": (_sz, [_i64]),
    "qp_select_k_smallest": (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "qp_select_keys": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "qp_gather_kv": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i64, _vp]),
    "qp_prune_staged": (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    "qp_prune_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "qp_prune_tail_workspace_bytes": (_sz, [_vp, _i64, _i64, _i32, _i32, _vp]),
    "qp_prune_tail": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _sz, _vp]),
    "qp_gather_rows": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "qp_sp_unpack": (_i32, [_vp, _vp, _i32, _i32, _i64, _i32, _i64, _vp, _vp, _i64, _vp, _vp]),
    "qp_add_rmsnorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "qp_add_inplace": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "qp_swiglu": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "qp_swiglu_split": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "qp_gemv": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _i64, _i64, _i32, _vp]),
    "qp_decode_rope_append": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "qp_decode_attn_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "qp_decode_attn": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "qp_decode_attn_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "qp_decode_advance": (_i32, [_vp, _vp, _i64, _vp]),
    "qp_vit_rope": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "qp_vit_attn": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp]),
    "qp_vit_attn_varlen": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp]),
    "qp_quick_gelu": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "qp_patchify": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "qp_linear_tune": (_i32, [_vp, _vp, _c.POINTER(_vp), _i32, _vp, _i32, _f32, _vp, _i64, _i64, _i64, _i32, _vp, _sz, _vp]),
    "qp_linear_act": (_i32, [_vp, _vp, _vp, _vp, _i32, _f32, _vp, _i64, _i64, _i64, _i32, _vp, _sz, _vp]),
    "qp_linear_plan_choice": (_i32, [_vp, _i64, _i64, _i64, _i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32)]),
    "qp_dev_switch": (_i32, [_c.c_char_p, _i32]),
    "qp_add_layernorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    # frame ring of the overlap producer (ring.py)
    "qp_frame_ring_create": (_i32, [_vp, _i32, _sz, _c.POINTER(_vp), _c.POINTER(_vp), _vp, _c.POINTER(_vp)]),
    "qp_frame_ring_start": (_i32, [_vp, _vp, _vp, _i64]),
    "qp_frame_ring_start_file": (_i32, [_vp, _c.c_char_p, _i64, _i64, _c.POINTER(_i64), _i64, _i32, _i32]),
    "qp_frame_ring_set_origin": (_i32, [_vp, _vp]),
    "qp_frame_ring_acquire": (_i32, [_vp, _i64, _vp, _c.POINTER(_vp), _c.POINTER(_sz)]),
    "qp_frame_ring_acquire_for": (_i32, [_vp, _i64, _vp, _i64, _c.POINTER(_vp), _c.POINTER(_sz)]),
    "qp_frame_ring_mark_read": (_i32, [_vp, _i64, _vp]),
    "qp_frame_ring_release": (_i32, [_vp, _i64, _vp]),
    "qp_frame_ring_stop": (_i32, [_vp]),
    "qp_frame_ring_stats": (_i32, [_vp, _c.POINTER(_c.c_double), _i32]),
    "qp_frame_ring_h2d_ms": (_i32, [_vp, _c.POINTER(_f32), _i64]),
    "qp_frame_ring_destroy": (None, [_vp]),
}
FRAME_SOURCE_FN = _c.CFUNCTYPE(_i64, _vp, _i64, _vp, _sz)      # qp_frame_source_fn


def load_library(path: str = LIB_PATH, hold_gil: bool = False) -> ctypes.CDLL:
    """dlopen the C-ABI library and bind every symbol the header declares (no compute, no GPU needed).
    hold_gil: bind through ctypes.PyDLL — the interpreter lock is KEPT across the call.  The operator object does that for its
    kernel-launch entry points: a launch returns in microseconds, whereas dropping the lock around each of the ~350 calls of a frame
    group hands it to any other Python thread that wants it, and getting it back costs up to a switch interval (5 ms) EACH time
    (measured with one busy Python thread beside the group loop: bench.py host_contention).  Long host-side calls (qp_host_memcpy in the
    producer thread) stay on the lock-free binding."""
    if not os.path.exists(path):
        raise QuickPrefillUnavailable(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C quickvideo_amd/csrc`). There is no CPU fallback for the QuickPrefill hot path.")
    lib = (ctypes.PyDLL if hold_gil else ctypes.CDLL)(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


_HOST_LIB = None


def host_memcpy(dst: torch.Tensor, src: torch.Tensor, threads: int = 4):
    """dst[...] = src[...] for contiguous CPU tensors of equal byte size, through qp_host_memcpy: native and GIL-free (needs only
    the shared library, no GPU)."""
    global _HOST_LIB
    if _HOST_LIB is None:
        _HOST_LIB = load_library()
    nbytes = src.numel() * src.element_size()
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == nbytes and not dst.is_cuda and not src.is_cuda
    rc = _HOST_LIB.qp_host_memcpy(dst.data_ptr(), src.data_ptr(), nbytes, threads)
    if rc != QP_OK:
        raise QuickPrefillError(rc, _HOST_LIB.qp_last_error().decode())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class QuickPrefillOps:
    """Stream-ordered operators on torch CUDA(HIP) tensors.  One instance per device."""

    def __init__(self, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise QuickPrefillUnavailable("no HIP device visible to torch: the QuickPrefill engine needs an MI355X (gfx950); "
                                          "there is no CPU fallback")
        self.lib = load_library(hold_gil=os.environ.get("QP_CTYPES_RELEASE_GIL") != "1")
        # entry points that SYNCHRONISE the stream (qp_linear_tune: a timing loop, hundreds of ms per shape) go through a second,
        # lock-releasing handle of the same .so: the producer thread, RCCL/gloo progress and Ctrl-C keep running meanwhile
        self.lib_blocking = load_library(hold_gil=False)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = _vp()
        self._check(self.lib.qp_create(ctypes.byref(h), self.device.index or 0))
        self.ctx = h
        self._select_ws = torch.empty(int(self.lib.qp_select_workspace_bytes(65536)), dtype=torch.uint8, device=self.device)
        self._attn_ws = None
        self._lt_ws = {}                          # hipBLASLt scratch, one buffer per stream (see _lt_workspace)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.qp_destroy(self.ctx)
        except Exception:
            pass

    # -- helpers
    def _check(self, status: int):
        if status != QP_OK:
            msg = self.lib.qp_last_error().decode()
            if status == QP_ERR_INVALID:
                raise ValueError(f"libquickprefill: {msg}")
            raise QuickPrefillError(status, msg)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    @property
    def cus(self) -> int:
        return int(self.lib.qp_device_cus(self.ctx))

    # -- seam 2
    def mrope_table(self, pos: torch.Tensor, sections, theta: float, head_dim: int):
        """pos int64 [3, n] (device) -> cos, sin bf16 [n, head_dim//2]."""
        assert pos.dtype == torch.int64 and pos.dim() == 2 and pos.shape[0] == 3 and pos.is_contiguous()
        n = pos.shape[1]
        cos = torch.empty(n, head_dim // 2, dtype=torch.bfloat16, device=pos.device)
        sin = torch.empty_like(cos)
        sec = (ctypes.c_int32 * 3)(*sections)
        self._check(self.lib.qp_mrope_table(self.ctx, pos.data_ptr(), n, sec, float(theta), head_dim, cos.data_ptr(),
                                            sin.data_ptr(), self._stream()))
        return cos, sin

    def rope_append(self, qkv, cos, sin, n_q, n_kv, head_dim, q_out, k_dst, v_dst, dst_head_stride, dst_row0, head_sumsq):
        n = qkv.shape[0]
        self._check(self.lib.qp_rope_append(self.ctx, qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), n, n_q, n_kv, head_dim,
                                            q_out.data_ptr(), k_dst.data_ptr(), v_dst.data_ptr(), dst_head_stride, dst_row0,
                                            _ptr(head_sumsq), self._stream()))

    # prune_mode argument of the seam-1 entry points (enum qp_prune_mode): bit 0 = keep the k LARGEST norms, bit 1 = score the VALUE rows
    PRUNE_KEY_NORMS_SMALL, PRUNE_KEY_NORMS, PRUNE_VECTOR_NORMS_SMALL, PRUNE_VECTOR_NORMS = 0, 1, 2, 3

    @staticmethod
    def prune_mode(norm_source: int, order: int) -> int:
        """(norm_source, order) of lvu_config.NORM_PRUNE_MODES -> the C ABI's prune_mode (utils.py:117-136)."""
        return (int(norm_source) << 1) | int(order)

    def rope_append_keys(self, qkv, cos, sin, n_q, n_kv, head_dim, q_out, k_dst, v_dst, dst_head_stride, dst_row0, head_sumsq, norm_keys,
                         mode: int = 0):
        """rope_append that also emits the layer's 16-bit norm keys (raises QuickPrefillError, status QP_ERR_UNSUPPORTED, when the
        head layout cannot be fused: see can_fuse_keys).  mode: one of the two key-row prune modes."""
        n = qkv.shape[0]
        self._check(self.lib.qp_rope_append_keys(self.ctx, qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), n, n_q, n_kv, head_dim,
                                                 q_out.data_ptr(), k_dst.data_ptr(), v_dst.data_ptr(), dst_head_stride, dst_row0,
                                                 _ptr(head_sumsq), norm_keys.data_ptr(), int(mode), self._stream()))

    @staticmethod
    def can_fuse_keys(n_q, n_kv) -> bool:
        rpt = n_q + 2 * n_kv
        return n_kv == 1 or (n_kv == 2 and rpt % 2 == 0 and n_q % 2 == 0) or (n_kv == 4 and rpt % 4 == 0 and n_q % 4 == 0)

    PRUNE_KEYS_MAX_N = 8192

    def query_scores(self, q_prompt, k_group, k_head_stride, n, n_q, n_kv, head_dim, norm_keys, value_sumsq=None, scores=None):
        """Query-based scoring (lvu_cache.py:97-117): q_prompt [m, n_q, D], the group's keys -> complemented score keys for prune_keys."""
        m = q_prompt.shape[0]
        need = int(self.lib.qp_query_scores_workspace_bytes(n, m, n_q))
        if getattr(self, "_qs_ws", None) is None or self._qs_ws.numel() < need:
            self._qs_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._check(self.lib.qp_query_scores(self.ctx, q_prompt.data_ptr(), k_group.data_ptr(), k_head_stride, n, m, n_q, n_kv, head_dim,
                                             _ptr(value_sumsq), norm_keys.data_ptr(), _ptr(scores), self._qs_ws.data_ptr(), self._qs_ws.numel(),
                                             self._stream()))

    def query_head_sums(self, q_prompt, k_group, k_head_stride, n, n_q, n_kv, head_dim, head_sums):
        """Step 1 of the query-based scoring for sharded heads: bf16 per-head sums [n_q, n] of the LOCAL heads (int16 tensor)."""
        m = q_prompt.shape[0]
        need = int(self.lib.qp_query_scores_workspace_bytes(n, m, n_q))
        if getattr(self, "_qs_ws", None) is None or self._qs_ws.numel() < need:
            self._qs_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._check(self.lib.qp_query_head_sums(self.ctx, q_prompt.data_ptr(), k_group.data_ptr(), k_head_stride, n, m, n_q, n_kv, head_dim,
                                                head_sums.data_ptr(), self._qs_ws.data_ptr(), self._qs_ws.numel(), self._stream()))

    def query_scores_from_head_sums(self, head_sums, n_heads_total, n, norm_keys, value_sumsq=None, n_kv_total=0, scores=None):
        """Step 2: mean over ALL heads (rows of head_sums in ascending head order) -> complemented score keys for prune_keys."""
        self._check(self.lib.qp_query_scores_from_head_sums(self.ctx, head_sums.data_ptr(), n_heads_total, n, _ptr(value_sumsq), n_kv_total,
                                                            norm_keys.data_ptr(), _ptr(scores), self._stream()))

    def norm_keys(self, head_sumsq, n_heads_total, n, norm_keys, mode: int = 0):
        self._check(self.lib.qp_norm_keys(self.ctx, head_sumsq.data_ptr(), n_heads_total, n, norm_keys.data_ptr(), int(mode), self._stream()))

    def prune_keys(self, norm_keys, n, k, k_src, v_src, src_head_stride, n_kv, head_dim, k_dst, v_dst, dst_head_stride, dst_row0, kept_idx):
        self._check(self.lib.qp_prune_keys(self.ctx, norm_keys.data_ptr(), n, k, k_src.data_ptr(), v_src.data_ptr(), src_head_stride, n_kv,
                                           head_dim, k_dst.data_ptr(), v_dst.data_ptr(), dst_head_stride, dst_row0, kept_idx.data_ptr(),
                                           self._stream()))

    # -- seam 3
    def prefill_attn(self, q, k_prefix, v_prefix, prefix_head_stride, prefix_len, k_new, v_new, new_head_stride, n, n_q, n_kv,
                     head_dim, scale, out, q_row0=0, nq=None):
        """q/out rows = the group's new tokens [q_row0, q_row0+nq) (default: all n)."""
        nq = n if nq is None else nq
        need = int(self.lib.qp_attn_workspace_bytes(self.ctx, nq, prefix_len + q_row0, n_q, n_kv))
        if self._attn_ws is None or self._attn_ws.numel() < need:          # caller-owned scratch, grown on demand
            self._attn_ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        self._check(self.lib.qp_prefill_attn_rows(self.ctx, q.data_ptr(), _ptr(k_prefix), _ptr(v_prefix), prefix_head_stride,
                                                  prefix_len, k_new.data_ptr(), v_new.data_ptr(), new_head_stride, n, q_row0, nq,
                                                  n_q, n_kv, head_dim, float(scale), out.data_ptr(), self._attn_ws.data_ptr(),
                                                  self._attn_ws.numel(), self._stream()))

    # -- seam 1
    def key_sumsq(self, k, head_stride, row0, n, n_kv, head_dim, head_sumsq):
        self._check(self.lib.qp_key_sumsq(self.ctx, k.data_ptr(), head_stride, row0, n, n_kv, head_dim, head_sumsq.data_ptr(),
                                          self._stream()))

    def select_k_smallest(self, head_sumsq, n_heads_total, n, k, kept_idx, norm_bits=None, mode: int = 0):
        need = int(self.lib.qp_select_workspace_bytes(n))
        if self._select_ws.numel() < need:
            self._select_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._check(self.lib.qp_select_k_smallest(self.ctx, head_sumsq.data_ptr(), n_heads_total, n, k, kept_idx.data_ptr(),
                                                  _ptr(norm_bits), int(mode), self._select_ws.data_ptr(), self._select_ws.numel(),
                                                  self._stream()))

    def select_keys(self, norm_keys, n, k, kept_idx):
        """k smallest of n ready-made 16-bit sort keys (ties -> lowest index), ascending index list; any n."""
        self._check(self.lib.qp_select_keys(self.ctx, norm_keys.data_ptr(), n, k, kept_idx.data_ptr(), self._stream()))

    def gather_kv(self, k_src, v_src, src_head_stride, idx, k, n_kv, head_dim, k_dst, v_dst, dst_head_stride, dst_row0):
        self._check(self.lib.qp_gather_kv(self.ctx, k_src.data_ptr(), v_src.data_ptr(), src_head_stride, idx.data_ptr(), k, n_kv,
                                          head_dim, k_dst.data_ptr(), v_dst.data_ptr(), dst_head_stride, dst_row0,
                                          self._stream()))

    def prune_staged(self, head_sumsq, n_heads_total, n, k, k_src, v_src, src_head_stride, n_kv, head_dim, k_dst, v_dst,
                     dst_head_stride, dst_row0, kept_idx, norm_bits=None, mode: int = 0):
        self._check(self.lib.qp_prune_staged(self.ctx, head_sumsq.data_ptr(), n_heads_total, n, k, k_src.data_ptr(), v_src.data_ptr(),
                                             src_head_stride, n_kv, head_dim, k_dst.data_ptr(), v_dst.data_ptr(), dst_head_stride,
                                             dst_row0, kept_idx.data_ptr(), _ptr(norm_bits), int(mode), self._stream()))

    def prune_workspace_bytes(self, n, k, n_kv, head_dim) -> int:
        """Upper bound: enough for either form of qp_prune_tail on any device / stream."""
        return int(self.lib.qp_prune_workspace_bytes(n, k, n_kv, head_dim))

    def prune_tail_workspace_bytes(self, n, k, n_kv, head_dim) -> int:
        """Exact size on this context and the current stream (the 2.25 B/token in-place figure only when that form will run)."""
        return int(self.lib.qp_prune_tail_workspace_bytes(self.ctx, n, k, n_kv, head_dim, self._stream()))

    def prune_tail(self, k_cache, v_cache, head_stride, past_len, n, k, n_kv, head_dim, kept_idx, workspace, mode: int = 0):
        self._check(self.lib.qp_prune_tail(self.ctx, k_cache.data_ptr(), v_cache.data_ptr(), head_stride, past_len, n, k, n_kv,
                                           head_dim, kept_idx.data_ptr(), int(mode), workspace.data_ptr(),
                                           workspace.numel() * workspace.element_size(), self._stream()))

    # -- one segment through all layers in one call (qp_prefill_segment)
    def attn_workspace(self, n: int, prefix_lens, n_q: int, n_kv: int) -> torch.Tensor:
        """Scratch large enough for the attention launch of n new rows over ANY of the given prefix lengths (the launch plan, and with it the
        size, is chosen per shape)."""
        need = max(int(self.lib.qp_attn_workspace_bytes(self.ctx, n, int(p), n_q, n_kv)) for p in set(prefix_lens))
        if self._attn_ws is None or self._attn_ws.numel() < need:
            self._attn_ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._attn_ws

    def prefill_segment(self, seg: "QpSegment", layers, cache_len, k_keep):
        """seg: a filled QpSegment; layers: (QpLayer * L) array; cache_len / k_keep: (c_int64 * L) arrays (cache_len is updated in place).
        The hipBLASLt scratch of the current stream and the stream itself are filled in here."""
        ws = self._lt_workspace()
        seg.gemm_ws, seg.gemm_ws_bytes = ws.data_ptr(), ws.numel()
        self._check(self.lib.qp_prefill_segment(self.ctx, ctypes.byref(seg), layers, cache_len, k_keep, self._stream()))

    def vit_blocks(self, blocks, n_blocks, n_seq, seq_len, dim, heads, mlp_dim, x, y, qkv, att, pending, z, cos, sin, eps):
        """All blocks of the Qwen2-VL tower in one call (qp_vit_blocks); `blocks`: (QpVitBlock * n_blocks) array."""
        ws = self._lt_workspace()
        self._check(self.lib.qp_vit_blocks(self.ctx, blocks, n_blocks, n_seq, seq_len, dim, heads, mlp_dim, x.data_ptr(), y.data_ptr(), qkv.data_ptr(),
                                           att.data_ptr(), pending.data_ptr(), z.data_ptr(), cos.data_ptr(), sin.data_ptr(), float(eps),
                                           ws.data_ptr(), ws.numel(), self._stream()))

    def sp_unpack(self, gathered, world, n_kv, m2, head_dim, n, k_stage, v_stage, stage_head_stride, sumsq_out):
        self._check(self.lib.qp_sp_unpack(self.ctx, gathered.data_ptr(), world, n_kv, m2, head_dim, n, k_stage.data_ptr(), v_stage.data_ptr(),
                                          stage_head_stride, sumsq_out.data_ptr(), self._stream()))

    def gather_rows(self, src, idx, k, row_bytes, dst):
        self._check(self.lib.qp_gather_rows(self.ctx, src.data_ptr(), idx.data_ptr(), k, row_bytes, dst.data_ptr(), self._stream()))

    # -- glue
    def add_rmsnorm(self, h, delta, w, out, eps):
        n, hidden = h.shape
        self._check(self.lib.qp_add_rmsnorm(self.ctx, h.data_ptr(), _ptr(delta), w.data_ptr(), out.data_ptr(), n, hidden, float(eps),
                                            self._stream()))

    def add_inplace(self, h, delta):
        self._check(self.lib.qp_add_inplace(self.ctx, h.data_ptr(), delta.data_ptr(), h.numel(), self._stream()))

    def swiglu(self, gate_up, out):
        n, two_i = gate_up.shape
        self._check(self.lib.qp_swiglu(self.ctx, gate_up.data_ptr(), n, two_i // 2, out.data_ptr(), self._stream()))

    def swiglu_split(self, gate, up, out):
        n, inter = gate.shape
        self._check(self.lib.qp_swiglu_split(self.ctx, gate.data_ptr(), up.data_ptr(), n, inter, out.data_ptr(), self._stream()))

    # -- decode step (device-resident state int64[2] = {kv_len, rope_pos}; graph-capturable)
    GEMV_BIAS, GEMV_SWIGLU, GEMV_RESIDUAL = 0, 1, 2

    def gemv(self, w, x, out, mode=0, bias=None, norm_w=None, eps=0.0):
        """out[n_out] = epilogue(w[n_out(*2 for SwiGLU)][k] . x[k]); x = RMSNorm(x) * norm_w first when norm_w is given."""
        k = w.shape[1]
        n_out = w.shape[0] // 2 if mode == self.GEMV_SWIGLU else w.shape[0]
        assert w.is_contiguous() and x.numel() == k and out.numel() == n_out
        self._check(self.lib.qp_gemv(self.ctx, w.data_ptr(), x.data_ptr(), _ptr(norm_w), float(eps), _ptr(bias), out.data_ptr(),
                                     n_out, k, mode, self._stream()))

    def decode_rope_append(self, qkv, state, theta, n_q, n_kv, D, q_out, k_cache, v_cache, head_stride, cos=None, sin=None):
        self._check(self.lib.qp_decode_rope_append(self.ctx, qkv.data_ptr(), state.data_ptr(), _ptr(cos), _ptr(sin), float(theta), n_q, n_kv, D,
                                                   q_out.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), head_stride, self._stream()))

    def decode_attn_workspace(self, n_q, n_kv):
        nbytes = int(self.lib.qp_decode_attn_workspace_bytes(self.ctx, n_q, n_kv))
        return torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)

    def decode_attn(self, q, k_cache, v_cache, head_stride, state, n_q, n_kv, D, scale, out, workspace):
        self._check(self.lib.qp_decode_attn(self.ctx, q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), head_stride, state.data_ptr(),
                                            n_q, n_kv, D, float(scale), out.data_ptr(), workspace.data_ptr(), workspace.numel() * 4,
                                            self._stream()))

    def decode_attn_fused(self, qkv, cos, sin, state, k_cache, v_cache, head_stride, n_q, n_kv, D, scale, out, workspace):
        self._check(self.lib.qp_decode_attn_fused(self.ctx, qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), state.data_ptr(), k_cache.data_ptr(),
                                                  v_cache.data_ptr(), head_stride, n_q, n_kv, D, float(scale), out.data_ptr(),
                                                  workspace.data_ptr(), workspace.numel() * 4, self._stream()))

    def decode_advance(self, state):
        self._check(self.lib.qp_decode_advance(self.ctx, state.data_ptr(), state.numel(), self._stream()))

    # -- vision front end
    def vit_rope(self, qkv, cos, sin, heads, head_dim):
        self._check(self.lib.qp_vit_rope(self.ctx, qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), qkv.shape[0], heads, head_dim, self._stream()))

    def vit_attn(self, qkv, n_seq, seq_len, heads, head_dim, scale, out):
        self._check(self.lib.qp_vit_attn(self.ctx, qkv.data_ptr(), n_seq, seq_len, heads, head_dim, float(scale), out.data_ptr(), self._stream()))

    def vit_attn_varlen(self, qkv, cu_seqlens, max_seq_len, heads, head_dim, scale, out):
        """Ragged batch (window attention): cu_seqlens int32 [n_seq+1] on the device."""
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_cuda
        self._check(self.lib.qp_vit_attn_varlen(self.ctx, qkv.data_ptr(), cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, max_seq_len, heads,
                                                head_dim, float(scale), out.data_ptr(), self._stream()))

    def add_layernorm(self, x, delta, w, b, out, eps):
        n, hidden = x.shape
        self._check(self.lib.qp_add_layernorm(self.ctx, x.data_ptr(), _ptr(delta), w.data_ptr(), b.data_ptr(), out.data_ptr(), n, hidden,
                                              float(eps), self._stream()))

    ACT_NONE, ACT_SWISH = 0, 1

    def _lt_workspace(self):
        """hipBLASLt scratch of the CURRENT stream.  One buffer per stream: stream-K / split-K GEMMs keep partial tiles and flags in it,
        and the ViT stream's GEMMs run concurrently with the prefill's (pipeline.py) — sharing one buffer between them corrupts both and
        can leave a stream-K workgroup spinning on a flag forever (seen as a stuck device in the video -> first-token leg)."""
        key = self._stream()
        ws = self._lt_ws.get(key)
        if ws is None:
            ws = self._lt_ws[key] = torch.empty(128 << 20, dtype=torch.uint8, device=self.device)
        return ws

    def linear_act(self, x, w, bias, out, act, alpha=1.0):
        """out = act(alpha * x w^T + bias) as one hipBLASLt GEMM with the activation in the epilogue (bias bf16 or fp32)."""
        m, k = x.shape
        n = w.shape[0]
        ws = self._lt_workspace()
        f32 = 1 if (bias is not None and bias.dtype == torch.float32) else 0
        self._check(self.lib.qp_linear_act(self.ctx, x.data_ptr(), w.data_ptr(), _ptr(bias), f32, float(alpha), out.data_ptr(), m, n, k, act,
                                           ws.data_ptr(), ws.numel(), self._stream()))

    def linear_tune(self, x, weights, bias, out, act=0, alpha=1.0):
        """Time hipBLASLt's candidates for out = act(alpha x w^T + bias) over the given weight tensors (cold, round-robin); keep the best."""
        m, k = x.shape
        n = weights[0].shape[0]
        ws = self._lt_workspace()
        arr = (_vp * len(weights))(*[w.data_ptr() for w in weights])
        f32 = 1 if (bias is not None and bias.dtype == torch.float32) else 0
        self._check(self.lib_blocking.qp_linear_tune(self.ctx, x.data_ptr(), arr, len(weights), _ptr(bias), f32, float(alpha), out.data_ptr(), m, n, k,
                                                     act, ws.data_ptr(), ws.numel(), self._stream()))

    def linear_plan_choice(self, m, n, k, act=0, bias=None):
        """-> (index of the hipBLASLt heuristic candidate THIS context runs for the problem, or -1 if it has not met it;
        whether the process holds a stopwatch decision for it on this device).  bias: None | a bf16/fp32 bias tensor."""
        kind = 0 if bias is None else (2 if bias.dtype == torch.float32 else 1)
        choice, tuned = _i32(-1), _i32(0)
        self._check(self.lib.qp_linear_plan_choice(self.ctx, m, n, k, act, kind, ctypes.byref(choice), ctypes.byref(tuned)))
        return choice.value, bool(tuned.value)

    def dev_switch(self, name: str, value: int):
        """Developer A/B switch of the launch paths (include/quickprefill.h: qp_dev_switch) — process-wide; the library does not read the
        environment at launch time."""
        self._check(self.lib.qp_dev_switch(name.encode(), int(value)))

    def patchify(self, frames_u8, patch, temporal_patch, merge, lut, out):
        """uint8 frames [F, 3, H, W] -> out [gt*gh*gw, row_elems] bf16 (HF patch order; columns behind the patch zeroed); lut: [3, 256] bf16."""
        f, c, h, w = frames_u8.shape
        assert c == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and out.is_contiguous() and lut.is_contiguous()
        self._check(self.lib.qp_patchify(self.ctx, frames_u8.data_ptr(), f, h, w, patch, temporal_patch, merge, lut.data_ptr(), out.data_ptr(),
                                         out.shape[1], self._stream()))

    def quick_gelu(self, x, out):
        self._check(self.lib.qp_quick_gelu(self.ctx, x.data_ptr(), out.data_ptr(), x.numel(), self._stream()))
