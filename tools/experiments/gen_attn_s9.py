#!/usr/bin/env python3
"""Generates quickvideo_amd/csrc/qp_attn_s9_iter.inc: the tile step of attn_fwd_kernel_s9 — the s6 pipeline (tools/gen_attn_s6.py) with the
softmax moved one tile further away from the matrix products that consume it.

s6, step t:   Q part  S(t+1) = K(t+1).Q^T   ||  softmax(t), elements 0..~22 (7 of them before the part's first MFMA)
              P part  O^T += V(t)^T.P(t)    ||  softmax(t), the rest; row max of S(t+1)
P(t) must be complete when the P part needs it, so the VALU work piles up in the Q part (measured: 1355-1775 clk for 1024 clk of MFMA,
then 920 clk for the P part).  s9 keeps a second packed-P buffer (16 VGPRs) and runs
s9, step t:   Q part  S(t+2) = K(t+2).Q^T   ||  softmax(t+1), elements 0..17
              P part  O^T += V(t)^T.P(t)    ||  softmax(t+1), elements 18..31; row max of S(t+2)
with P(t) finished during step t-1: one score element per MFMA gap in both parts.  Element e: A (x = s*c - m*c) in gap e-2, B (p = 2^x)
one gap later, C (row sum, bf16 pack) one gap after that; gaps 0 and 1 start two elements each so that element 31 retires in gap 31 and
nothing of a tile's softmax crosses the step boundary (the running row sum is rescaled there).
The reference switch of the lazy rescale is decided at the end of step t from max S(t+2) and used by softmax(t+2) in step t+1; the O
accumulators, which still receive P(t+1) (old reference) in step t+1, are rescaled one step later (`alpha_pend` in qp_attn_s9.hip).
K(t+2) is read from K[b], K(t+3) lands in K[1-b]; V as in s6.
RESULT (round 2): correct, 4-7 % SLOWER than s6 — not part of the product build.  To rebuild: copy qp_attn_s9.hip into
quickvideo_amd/csrc/, `python tools/experiments/gen_attn_s9.py > quickvideo_amd/csrc/qp_attn_s9_iter.inc`, add it to the Makefile and call
qp_launch_attn_s9 from qp_launch_prefill_attn.
"""
K_BASE = [0, 16384]
RING = 4


class Part:
    def __init__(self):
        self.out, self.lds_seq, self.issued_at = [], 0, {}

    def emit(self, s):
        self.out.append("  " + s)

    def lds_op(self, tag=None, n=1):
        self.lds_seq += n
        if tag is not None:
            self.issued_at[tag] = self.lds_seq - 1

    def wait_for(self, tag):
        return self.lds_seq - 1 - self.issued_at[tag]


def g_a(e):
    return e // 2 if e < 4 else e - 2


def el_ops(p, s_cons, pw_w, gap):
    """softmax element stages of `gap` (0..31), order C, B, A so that a register is read before the same gap rewrites it."""
    ops = []
    for e in range(32):
        if g_a(e) + 2 == gap:
            ops.append(f"S9_ELC({e & 3});" + (f" S9_PACK({pw_w}, {e >> 1}, {(e - 1) & 3}, {e & 3});" if e & 1 else ""))
    for e in range(32):
        if g_a(e) + 1 == gap:
            ops.append(f"S9_ELB({e & 1}, {e & 3});")
    for e in range(32):
        if g_a(e) == gap:
            ops.append(f"S9_ELA({s_cons}, {e}, {e & 1});")
    for o in ops:
        p.emit(o)
    if ops:
        p.emit("S9_ELKEEP();")


def vread(p, m, var):
    c, db = m >> 2, m & 3
    off = (((c >> 1) * 8 + (c & 1) * 4) * 4 + db) << 8
    p.emit(f"S9_VREAD({m % RING}, {var}, {off}, {off + (2 * 4 << 8)});")
    p.lds_op(("v", m), 2)


def gen_q(b):
    """Q part of step parity b: S(t+2) -> s{b} from K[b]; softmax(t+1) on s{1-b} -> pw{1-b}; ends with V pairs 0..3 in flight."""
    p = Part()
    s_prod, s_cons, pw_w = f"s{b}", f"s{1 - b}", f"pw{1 - b}"
    kb_r, kb_w = K_BASE[b], K_BASE[1 - b]

    def kread(j):
        kk, h = j >> 1, j & 1
        p.emit(f"S9_KREAD({j % RING}, {kk}, {kb_r + h * 8192});")
        p.lds_op(("k", j))

    p.emit(f"{s_prod}[0] = (f32x16_t){{0}}; {s_prod}[1] = (f32x16_t){{0}};")
    for j in range(RING):
        kread(j)
    p.emit("PIN();")
    for g in range(16):
        if g % 2 == 0:
            p.emit(f"S9_WAIT({p.wait_for(('k', g + 1))}); PIN();")
        p.emit(f"S9_QK({g % RING}, {g >> 1}, {s_prod}[{g & 1}]); PIN();")
        if g + RING < 16:
            kread(g + RING)
        else:
            vread(p, g + RING - 16, "vrd_pref")
        if g < 8:
            p.emit(f"S9_DMA_K({g}, {kb_w});" if g < 4 else f"S9_DMA_V({g - 4});")
        el_ops(p, s_cons, pw_w, g)
        p.emit("PIN();")
    p.emit("S9_STAMP(1);")
    p.emit(f"S9_MASK_NEXT({s_prod});")
    return p.out


def gen_p(b):
    """P part: O^T += V(t)^T.P(t) with pw{b}; softmax(t+1) continues on s{1-b} -> pw{1-b}; row max of s{b} = S(t+2)."""
    p = Part()
    s_prod, s_cons, pw_r, pw_w = f"s{b}", f"s{1 - b}", f"pw{b}", f"pw{1 - b}"
    for m in range(RING):
        p.lds_op(("v", m), 2)
    for g in range(16, 32):
        m = g - 16
        if m % 2 == 0:
            p.emit(f"S9_WAIT({p.wait_for(('v', m + 1))}); PIN();")
        p.emit(f"S9_PV({m % RING}, {m >> 2}, {m & 3}, {pw_r}); PIN();")
        if m + RING < 16:
            vread(p, m + RING, "vrd_main")
        el_ops(p, s_cons, pw_w, g)
        if g in (19, 23, 27, 31):
            p.emit(f"S9_MAX4({s_prod}, {(g - 19) // 4});")
        p.emit("PIN();")
    return p.out


def main():
    print("// GENERATED by tools/gen_attn_s9.py -- do not edit.  Parts of one tile step of attn_fwd_kernel_s9 (macros: qp_attn_s9.hip).")
    print("// S9_PART: 0/1 = Q part of step parity 0/1, 2/3 = P part.")
    parts = [gen_q(0), gen_q(1), gen_p(0), gen_p(1)]
    for i, body in enumerate(parts):
        print(("#if" if i == 0 else "#elif") + f" S9_PART == {i}")
        print("\n".join(body))
    print("#endif")


if __name__ == "__main__":
    main()
