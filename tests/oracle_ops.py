"""Test double with the interface of quickvideo_amd.native.QuickPrefillOps, computing on CPU tensors with the
oracle.  It lets the engine's HOST logic (planner, layer loop, KV arena bookkeeping, tensor-parallel sharding and
collectives over gloo) run in the CPU test suite.  It lives under tests/ — product code never imports it."""
import numpy as np
import torch

from oracle import qp_oracle as O


class OracleOps:
    cus = 0

    def mrope_table(self, pos, sections, theta, head_dim):
        spec = O.TextSpec(hidden=head_dim, n_heads=1, n_kv_heads=1, head_dim=head_dim, intermediate=8, n_layers=1, vocab=8,
                          rope_theta=theta, mrope_section=tuple(sections))
        cos, sin = O.mrope_cos_sin(pos.cpu(), spec, torch.bfloat16)
        return cos[:, :head_dim // 2].contiguous(), sin[:, :head_dim // 2].contiguous()

    def rope_append(self, qkv, cos, sin, n_q, n_kv, D, q_out, k_dst, v_dst, dst_head_stride, dst_row0, head_sumsq):
        n = qkv.shape[0]
        c, s = torch.cat([cos, cos], -1), torch.cat([sin, sin], -1)
        q = qkv[:, :n_q * D].view(n, n_q, D).transpose(0, 1)
        k = qkv[:, n_q * D:(n_q + n_kv) * D].view(n, n_kv, D).transpose(0, 1)
        v = qkv[:, (n_q + n_kv) * D:].view(n, n_kv, D).transpose(0, 1)
        q_out[:n].copy_(O.apply_rope(q, c, s).transpose(0, 1))
        kr = O.apply_rope(k, c, s)
        kd = k_dst.reshape(-1) if k_dst.is_contiguous() else None
        for h in range(n_kv):
            self._rows(k_dst, h, dst_head_stride, dst_row0, n, D).copy_(kr[h])
            self._rows(v_dst, h, dst_head_stride, dst_row0, n, D).copy_(v[h])
        if head_sumsq is not None:
            ss = O.key_sumsq_heads(O.torch_bf16_to_bits(kr.contiguous()))
            head_sumsq.view(-1)[: n_kv * n].copy_(torch.from_numpy(ss).view(-1))

    @staticmethod
    def _rows(t, h, head_stride, row0, n, D):
        """view of rows [row0,row0+n) of head h given a base tensor whose storage starts at element 0 of head 0"""
        base = t.as_strided((t.untyped_storage().nbytes() // t.element_size() - t.storage_offset(),), (1,), t.storage_offset())
        return base[h * head_stride + row0 * D: h * head_stride + (row0 + n) * D].view(n, D)

    def prefill_attn(self, q, k_prefix, v_prefix, prefix_head_stride, P, k_new, v_new, new_head_stride, n, n_q, n_kv, D, scale, out,
                     q_row0=0, nq=None):
        nq = n if nq is None else nq
        if (q_row0, nq) != (0, n):        # query sub-range: the keys after the last local query are invisible anyway
            n_vis = q_row0 + nq
            qfull = torch.zeros(n_vis, n_q, D, dtype=q.dtype)
            qfull[q_row0:] = q[:nq]
            tmp = torch.empty(n_vis, n_q, D, dtype=q.dtype)
            self.prefill_attn(qfull, k_prefix, v_prefix, prefix_head_stride, P, k_new, v_new, new_head_stride, n_vis, n_q, n_kv, D, scale, tmp)
            out[:nq].copy_(tmp[q_row0:])
            return
        ks, vs = [], []
        for h in range(n_kv):
            parts_k, parts_v = [], []
            if P > 0:
                parts_k.append(self._rows(k_prefix, h, prefix_head_stride, 0, P, D)); parts_v.append(self._rows(v_prefix, h, prefix_head_stride, 0, P, D))
            parts_k.append(self._rows(k_new, h, new_head_stride, 0, n, D)); parts_v.append(self._rows(v_new, h, new_head_stride, 0, n, D))
            ks.append(torch.cat(parts_k)); vs.append(torch.cat(parts_v))
        out[:n].copy_(O.attention_bottom_right(q[:n].transpose(0, 1), torch.stack(ks), torch.stack(vs), scale))

    def key_sumsq(self, k, head_stride, row0, n, n_kv, head_dim, head_sumsq):
        rows = torch.stack([self._rows(k, h, head_stride, row0, n, head_dim) for h in range(n_kv)])
        ss = O.key_sumsq_heads(O.torch_bf16_to_bits(rows.contiguous()))
        head_sumsq.view(-1)[: n_kv * n].copy_(torch.from_numpy(ss).view(-1))

    def select_k_smallest(self, head_sumsq, n_heads_total, n, k, kept_idx, norm_bits=None, mode=0):
        """mode = the C ABI's prune_mode: bit 0 = keep the k largest norms"""
        ss = head_sumsq.reshape(-1)[: n_heads_total * n].view(n_heads_total, n).numpy()
        nb = O.key_norms_bf16(ss)
        kept_idx[:k].copy_(torch.from_numpy(O.select_k_largest(nb, k) if (mode & 1) else O.select_k_smallest(nb, k)))

    def select_keys(self, norm_keys, n, k, kept_idx):
        kept_idx[:k].copy_(torch.from_numpy(O.select_k_smallest(norm_keys[:n].numpy().view(np.uint16), k)))

    def gather_kv(self, k_src, v_src, src_head_stride, idx, k, n_kv, D, k_dst, v_dst, dst_head_stride, dst_row0):
        ii = idx[:k].long()
        n_src = int(ii.max()) + 1
        for h in range(n_kv):
            self._rows(k_dst, h, dst_head_stride, dst_row0, k, D).copy_(self._rows(k_src, h, src_head_stride, 0, n_src, D)[ii])
            self._rows(v_dst, h, dst_head_stride, dst_row0, k, D).copy_(self._rows(v_src, h, src_head_stride, 0, n_src, D)[ii])

    def prune_staged(self, head_sumsq, n_heads_total, n, k, k_src, v_src, src_head_stride, n_kv, D, k_dst, v_dst, dst_head_stride,
                     dst_row0, kept_idx, norm_bits=None, mode=0):
        self.select_k_smallest(head_sumsq, n_heads_total, n, k, kept_idx, mode=mode)
        self.gather_kv(k_src, v_src, src_head_stride, kept_idx, k, n_kv, D, k_dst, v_dst, dst_head_stride, dst_row0)

    PRUNE_KEYS_MAX_N = 8192

    def query_scores(self, q_prompt, k_group, k_head_stride, n, n_q, n_kv, D, norm_keys, value_sumsq=None, scores=None):
        kg = torch.stack([self._rows(k_group, h, k_head_stride, 0, n, D) for h in range(n_kv)])
        sc = O.query_attention_scores(q_prompt.transpose(0, 1).contiguous(), kg)
        if value_sumsq is not None:                      # * bf16 value norm (utils.py:58-62)
            vn = O.bits_to_torch_bf16(O.key_norms_bf16(value_sumsq.reshape(-1)[: n_kv * n].view(n_kv, n).numpy()))
            sc = sc * vn
        bits = O.torch_bf16_to_bits(sc)
        norm_keys[:n].copy_(torch.from_numpy((~bits).astype(np.uint16).view(np.int16)))

    def query_head_sums(self, q_prompt, k_group, k_head_stride, n, n_q, n_kv, D, head_sums):
        kg = torch.stack([self._rows(k_group, h, k_head_stride, 0, n, D) for h in range(n_kv)])
        qp = q_prompt.transpose(0, 1).contiguous()                                   # [Hq_local, m, D]
        kr = kg.repeat_interleave(n_q // n_kv, dim=0)
        a = torch.einsum("hqd,hkd->hqk", qp, kr) / (D ** 0.5)
        p = torch.softmax(a, dim=-1, dtype=torch.float32).to(qp.dtype)
        head_sums.view(-1)[: n_q * n].copy_(torch.from_numpy(O.torch_bf16_to_bits(p.sum(-2)).view(np.int16)).view(-1))

    def query_scores_from_head_sums(self, head_sums, n_heads_total, n, norm_keys, value_sumsq=None, n_kv_total=0, scores=None):
        s1 = O.bits_to_torch_bf16(head_sums.view(-1)[: n_heads_total * n].view(n_heads_total, n).numpy().view(np.uint16))
        sc = s1.mean(0)
        if value_sumsq is not None:
            vn = O.bits_to_torch_bf16(O.key_norms_bf16(value_sumsq.reshape(-1)[: n_kv_total * n].view(n_kv_total, n).numpy()))
            sc = sc * vn
        norm_keys[:n].copy_(torch.from_numpy((~O.torch_bf16_to_bits(sc)).astype(np.uint16).view(np.int16)))

    def prune_keys(self, norm_keys, n, k, k_src, v_src, src_head_stride, n_kv, D, k_dst, v_dst, dst_head_stride, dst_row0, kept_idx):
        keys = norm_keys[:n].numpy().view(np.uint16)
        kept_idx[:k].copy_(torch.from_numpy(O.select_k_smallest(keys, k)))
        self.gather_kv(k_src, v_src, src_head_stride, kept_idx, k, n_kv, D, k_dst, v_dst, dst_head_stride, dst_row0)

    def sp_unpack(self, gathered, world, n_kv, m2, head_dim, n, k_stage, v_stage, stage_head_stride, sumsq_out):
        m = 2 * m2
        kv_bytes = n_kv * m * head_dim * 2
        chunk = 2 * kv_bytes + n_kv * m * 4
        g = gathered.view(torch.uint8)[: world * chunk].view(world, chunk)
        X = g[:, : 2 * kv_bytes].view(k_stage.dtype).view(world, 2, n_kv, 2, m2, head_dim)
        S = g[:, 2 * kv_bytes:].view(torch.float32).view(world, n_kv, 2, m2)
        rev = torch.arange(world - 1, -1, -1)
        full = torch.cat([X[:, :, :, 0].permute(1, 2, 0, 3, 4), X[rev][:, :, :, 1].permute(1, 2, 0, 3, 4)], 2).reshape(2, n_kv, 2 * world * m2, head_dim)
        ks = k_stage.view(-1)[: n_kv * stage_head_stride].view(n_kv, stage_head_stride // head_dim, head_dim)
        vs = v_stage.view(-1)[: n_kv * stage_head_stride].view(n_kv, stage_head_stride // head_dim, head_dim)
        ks[:, :n].copy_(full[0, :, :n]); vs[:, :n].copy_(full[1, :, :n])
        ss = torch.cat([S[:, :, 0].permute(1, 0, 2), S[rev][:, :, 1].permute(1, 0, 2)], 1).reshape(n_kv, 2 * world * m2)[:, :n]
        sumsq_out.view(-1)[: n_kv * n].view(n_kv, n).copy_(ss)

    def gather_rows(self, src, idx, k, row_bytes, dst):
        dst[:k].copy_(src[idx[:k].long()])

    def add_rmsnorm(self, h, delta, w, out, eps):
        if delta is not None:
            h.copy_(h + delta)
        out.copy_(O.rmsnorm(h, w, eps))

    def add_inplace(self, h, delta):
        h.copy_(h + delta)

    def swiglu_split(self, gate, up, out):
        out.copy_(torch.nn.functional.silu(gate) * up)

    def swiglu(self, gate_up, out):
        i = gate_up.shape[1] // 2
        out.copy_(torch.nn.functional.silu(gate_up[:, :i]) * gate_up[:, i:])
