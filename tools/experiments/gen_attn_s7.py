#!/usr/bin/env python3
"""Generates quickvideo_amd/csrc/qp_attn_s7_iter.inc: the tile step of attn_fwd_kernel_s7 (one wave per SIMD, 64 query rows =
two 32-row q blocks A and B per wave, MFMA accumulators owned through inline-asm constraints).

Same software pipeline as tools/gen_attn_s6.py (Q part: scores of the NEXT tile, P part: P.V of the current tile, softmax
elements / fragment reads / DMA pieces placed in the MFMA gaps), but every K / V fragment read from LDS now feeds TWO MFMAs
(q blocks A and B), so a step is 64 gaps with half the LDS reads, waits, DMA pieces and barriers per MFMA:
  Q part, gaps  0..31   frag j = g/2:  S_next_A[j&1] += K_j.QA_{j/2}   (g even),   S_next_B[j&1] += K_j.QB_{j/2}   (g odd)
  P part, gaps 32..63   pair m = (g-32)/2 (chunk c = m/4, d block db = m%4):  O_A[db] += V_m.P_A[c]  (even),  O_B[db] += V_m.P_B[c]  (odd)
Softmax elements: 64 per step (32 of A, 32 of B, interleaved A0 B0 A1 B1 ...), three-stage pipeline, ~0.95 gaps per element so
that chunk c of either block is packed before gap 32 + 8c; the row maxima of the next tile fill gaps 56..63.
usage: python tools/gen_attn_s7.py > quickvideo_amd/csrc/qp_attn_s7_iter.inc
"""
K_BASE = [0, 16384]
RING = 4
# asm-owned accumulator file (never visible to hipcc as variables): O of q block A / B, Q fragments of A / B
O_BASE = {"A": 0, "B": 64}          # a[O_BASE + 16*db .. +15]
Q_BASE = {"A": 128, "B": 160}       # a[Q_BASE + 4*kk .. +3]


def areg(lo, n):
    return f"a[{lo}:{lo + n - 1}]"


def clob(lo, n):
    return ", ".join(f'"a{r}"' for r in range(lo, lo + n))


def mfma_qk(acc, slot, st, kk, first):
    q = areg(Q_BASE[st] + 4 * kk, 4)
    if first:
        return f'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, {q}, 0" : "=&v"({acc}) : "v"(kr[{slot}]));'
    return f'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, {q}, %0" : "+v"({acc}) : "v"(kr[{slot}]));'


def mfma_pv(st, db, slot, c):
    o = areg(O_BASE[st] + 16 * db, 16)
    return (f'{{ S7_PV_OPERANDS({slot}, pw{st}, {c}); asm volatile("v_mfma_f32_32x32x16_bf16 {o}, %0, %1, {o}" : : "v"(av_), "v"(pc_) : '
            f'{clob(O_BASE[st] + 16 * db, 16)}); }}')


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


def element_plan():
    """gap -> list of (stage, stream, element).  k-th element overall (stream A if k even) finishes (stage C) in gap floor(0.95 k) - 5;
    stage B two gaps earlier, stage A three gaps earlier (no stage reads a result produced in its own gap)."""
    plan = {}
    for k in range(64):
        st, e = "AB"[k & 1], k >> 1
        gc = int(k * 0.95) - 5
        for stage, g in (("A", gc - 3), ("B", gc - 2), ("C", gc)):
            plan.setdefault(g, []).append((stage, st, e))
    return plan


PLAN = element_plan()
assert min(PLAN) >= -8 and max(g for g, ops in PLAN.items() if any(s == "C" for s, _, _ in ops)) < 56
for c in range(4):           # chunk c of A (B) must be packed before gap 32 + 8c (33 + 8c)
    for st in "AB":
        last = max(g for g, ops in PLAN.items() for s, t, e in ops if s == "C" and t == st and e == 8 * c + 7)
        assert last < 32 + 8 * c, (c, st, last)


def el_ops(p, b, gap):
    ops = sorted(PLAN.get(gap, []), key=lambda o: "CBA".index(o[0]))      # C first (oldest data), then B, then A
    for stage, st, e in ops:
        if stage == "C":
            p.emit(f"S7_ELC({st}, {e & 3});" + (f" S7_PACK({st}, {e >> 1}, {(e - 1) & 3}, {e & 3});" if e & 1 else ""))
        elif stage == "B":
            p.emit(f"S7_ELB({st}, {e & 1}, {e & 3});")
        else:
            p.emit(f"S7_ELA({st}, s{b}{st}, {e}, {e & 1});")
    if ops:
        p.emit("S7_ELKEEP();")


def vread(p, m, var):
    c, db = m >> 2, m & 3
    off = (((c >> 1) * 8 + (c & 1) * 4) * 4 + db) << 8
    p.emit(f"S7_VREAD({m % RING}, {var}, {off}, {off + (2 * 4 << 8)});")
    p.lds_op(("v", m), 2)


def gen_q(b):
    p = Part()
    sn = f"s{1 - b}"
    kb_next, kb_w = K_BASE[1 - b], K_BASE[b]

    def kread(j):
        p.emit(f"S7_KREAD({j % RING}, {j >> 1}, {kb_next + (j & 1) * 8192});")
        p.lds_op(("k", j))

    for j in range(RING):
        kread(j)
    for gp in range(-8, 0):
        el_ops(p, b, gp)
    p.emit("PIN();")
    for g in range(32):
        j, st = g >> 1, "AB"[g & 1]
        if g % 4 == 0:                          # one counted wait releases two fragments
            p.emit(f"S7_WAIT({p.wait_for(('k', j + 1))}); PIN();")
        p.emit(mfma_qk(f"{sn}{st}[{j & 1}]", j % RING, st, j >> 1, j < 2) + " PIN();")   # first MFMA of an accumulator: C = 0
        if g & 1:                               # both q blocks have consumed fragment j: its ring slot is free
            if j + RING < 16:
                kread(j + RING)
            else:
                vread(p, j + RING - 16, "vrd")
        if g < 8:
            p.emit(f"S7_DMA_K({g}, {kb_w});" if g < 4 else f"S7_DMA_V({g - 4});")
        el_ops(p, b, g)
        p.emit("PIN();")
    p.emit(f"S7_MASK_NEXT({sn});")
    return p.out


def gen_p(b):
    p = Part()
    sn = f"s{1 - b}"
    for m in range(RING):
        p.lds_op(("v", m), 2)
    for g in range(32, 64):
        m, st = (g - 32) >> 1, "AB"[g & 1]
        if (g - 32) % 4 == 0:
            p.emit(f"S7_WAIT({p.wait_for(('v', m + 1))}); PIN();")
        p.emit(mfma_pv(st, m & 3, m % RING, m >> 2) + " PIN();")
        if g & 1 and m + RING < 16:
            vread(p, m + RING, "vrd")
        el_ops(p, b, g)
        if g >= 56:
            q = g - 56                           # 8 calls: A0..A3, B0..B3
            p.emit(f"S7_MAX4({'AB'[q >> 2]}, {sn}{'AB'[q >> 2]}, {q & 3});")
        p.emit("PIN();")
    return p.out


def gen_init():
    """zero O_A, O_B; load the Q fragments (uint4 qa[8], qb[8]: this lane's 8 x 16 B of the two query rows)."""
    out = []
    z = "\\n\\t".join(f"v_accvgpr_write_b32 a{r}, 0" for r in range(128))
    out.append(f'  asm volatile("{z}" : : : {clob(0, 128)});')
    for st, var in (("A", "qa"), ("B", "qb")):
        for kk in range(8):
            lo = Q_BASE[st] + 4 * kk
            txt = "\\n\\t".join(f"v_accvgpr_write_b32 a{lo + i}, %{i}" for i in range(4))
            src = f"{var}[{kk}]"
            out.append(f'  asm volatile("{txt}" : : "v"({src}.x), "v"({src}.y), "v"({src}.z), "v"({src}.w) : {clob(lo, 4)});')
    return out


def gen_rescale(st):
    """O_st *= alpha (rare branch; caller has fenced).  One temporary VGPR."""
    lo = O_BASE[st]
    txt = "\\n\\t".join(f"v_accvgpr_read_b32 %0, a{r}\\n\\ts_nop 0\\n\\tv_mul_f32 %0, %0, %1\\n\\ts_nop 0\\n\\tv_accvgpr_write_b32 a{r}, %0" for r in range(lo, lo + 64))
    return [f'  {{ float t_; asm volatile("{txt}" : "=&v"(t_) : "v"(alpha) : {clob(lo, 64)}); }}']


def gen_readout(st):
    """f32x16_t o[4] <- O_st (epilogue; caller has fenced)."""
    out = []
    for db in range(4):
        lo = O_BASE[st] + 16 * db
        txt = "\\n\\t".join(f"v_accvgpr_read_b32 %{i}, a{lo + i}" for i in range(16))
        outs = ", ".join(f'"=v"(r_[{i}])' for i in range(16))
        vals = ", ".join(f"r_[{i}]" for i in range(16))
        out.append(f'  {{ float r_[16]; asm volatile("{txt}" : {outs}); o[{db}] = (f32x16_t){{{vals}}}; }}')
    return out


def gen_prologue():
    """S(lo) of both q blocks the plain way (K tile in buffer 0): per accumulator the k-steps in ascending order, as everywhere."""
    out = []
    for kk in range(8):
        out.append(f"  S7_KREAD(0, {kk}, {K_BASE[0]}); S7_KREAD(1, {kk}, {K_BASE[0] + 8192}); S7_WAIT(0); PIN();")
        for h in (0, 1):
            for st in "AB":
                out.append("  " + mfma_qk(f"s0{st}[{h}]", h, st, kk, kk == 0) + " PIN();")
    return out


def main():
    print("// GENERATED by tools/gen_attn_s7.py -- do not edit.  S7_PART: 0/1 = Q part (S parity 0/1), 2/3 = P part, 4 = accumulator-file")
    print("// init (zero O, load Q), 5/6 = rescale O_A / O_B by `alpha`, 7/8 = read O_A / O_B into f32x16_t o[4], 9 = S(first tile).  Macros: qp_attn_s7.hip.")
    for i, body in enumerate([gen_q(0), gen_q(1), gen_p(0), gen_p(1), gen_init(), gen_rescale("A"), gen_rescale("B"),
                              gen_readout("A"), gen_readout("B"), gen_prologue()]):
        print(("#if" if i == 0 else "#elif") + f" S7_PART == {i}")
        print("\n".join(body))
    print("#endif")


if __name__ == "__main__":
    main()
