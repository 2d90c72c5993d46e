#!/usr/bin/env python3
"""Generates quickvideo_amd/csrc/qp_attn_s8_iter.inc: the tile step of attn_fwd_kernel_s8 — the s7 register scheme (one wave per
SIMD, q blocks A and B, asm-owned AGPR accumulators) with a THREE-stage software pipeline, so that all three kinds of work are
spread evenly over the 64 MFMA gaps of a step instead of a K-read/DMA/softmax-heavy Q half and a light P half (s7: 61 vs 38 clk/gap):
    step t:   S(t+2) = K(t+2).Q^T   (32 MFMAs)   |   P(t+1) = softmax(S(t+1))   (64 elements, one per gap)   |   O += V(t)^T.P(t)   (32 MFMAs)
MFMA order inside the step: quads  QK_A(frag q), QK_B(frag q), PV_A(pair q), PV_B(pair q),  q = 0..15 (frag q: k-step q/2, key
half q%2; pair q: 16-key chunk q/4, d block q%4), so K and V fragment reads alternate as well.  The row maxima of S(t+1) are taken
at the start of the step (they also cover the latency of the first fragment reads); the softmax reference is switched there and
the rescale of O, if any, is deferred to the end of the step, after P.V(t) — P(t) was formed against the old reference.
S7_PART-style selector S8_PART: 0/1 = step with parity 0/1; 4 = accumulator-file init, 5/6 = rescale O_A/O_B, 7/8 = read-out,
9/10 = S(first tile) / S(second tile) the plain way.    usage: python tools/gen_attn_s8.py > quickvideo_amd/csrc/qp_attn_s8_iter.inc
"""
from gen_attn_s7 import Part, areg, clob, gen_init, gen_rescale, gen_readout, K_BASE, O_BASE, Q_BASE

RING = 4


def mfma_qk(acc, slot, st, kk, first):
    q = areg(Q_BASE[st] + 4 * kk, 4)
    if first:
        return f'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, {q}, 0" : "=&v"({acc}) : "v"(kr[{slot}]));'
    return f'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, {q}, %0" : "+v"({acc}) : "v"(kr[{slot}]));'


def mfma_pv(st, db, slot, pw, c):
    o = areg(O_BASE[st] + 16 * db, 16)
    return (f'{{ S8_PV_OPERANDS({slot}, {pw}, {c}); asm volatile("v_mfma_f32_32x32x16_bf16 {o}, %0, %1, {o}" : : "v"(av_), "v"(pc_) : '
            f'{clob(O_BASE[st] + 16 * db, 16)}); }}')


def element_plan():
    """gap -> stages.  Element k (stream A if k even, element k/2 of the 32 of its q block) finishes (stage C) in gap k; stage B two gaps
    earlier, stage A three gaps earlier."""
    plan = {}
    for k in range(64):
        st, e = "AB"[k & 1], k >> 1
        for stage, g in (("A", k - 3), ("B", k - 2), ("C", k)):
            plan.setdefault(g, []).append((stage, st, e))
    return plan


PLAN = element_plan()


def el_ops(p, b, gap):
    sc, pwn = f"s{1 - b}", f"pw{1 - b}"                   # consumed scores S(t+1), produced probabilities P(t+1)
    ops = sorted(PLAN.get(gap, []), key=lambda o: "CBA".index(o[0]))
    for stage, st, e in ops:
        if stage == "C":
            p.emit(f"S8_ELC({st}, {e & 3});" + (f" S8_PACK({st}, {pwn}{st}, {e >> 1}, {(e - 1) & 3}, {e & 3});" if e & 1 else ""))
        elif stage == "B":
            p.emit(f"S8_ELB({st}, {e & 1}, {e & 3});")
        else:
            p.emit(f"S8_ELA({st}, {sc}{st}, {e}, {e & 1});")
    if ops:
        p.emit("S8_ELKEEP();")


def gen_step(b):
    p = Part()
    sp, sc = f"s{b}", f"s{1 - b}"                          # produced S(t+2), consumed S(t+1)
    kb_read, kb_w = K_BASE[b], K_BASE[1 - b]

    def kread(q):
        p.emit(f"S8_KREAD({q % RING}, {q >> 1}, {kb_read + (q & 1) * 8192});")
        p.lds_op(("k", q))

    def vread(q):
        c, db = q >> 2, q & 3
        off = (((c >> 1) * 8 + (c & 1) * 4) * 4 + db) << 8
        p.emit(f"S8_VREAD({q % RING}, vrd, {off}, {off + (2 * 4 << 8)});")
        p.lds_op(("v", q), 2)

    for q in range(RING):
        kread(q)
    for q in range(RING):
        vread(q)
    for q in range(8):                                     # row maxima of S(t+1): 4 calls per q block
        p.emit(f"S8_MAX4({'AB'[q >> 2]}, {sc}{'AB'[q >> 2]}, {q & 3});")
    p.emit("PIN();")
    p.emit("S8_DECIDE(A) S8_DECIDE(B)")
    for gp in range(-3, 0):
        el_ops(p, b, gp)
    p.emit("PIN();")
    for g in range(64):
        q, r = g >> 2, g & 3
        if r == 0:
            p.emit(f"S8_WAIT({min(15, p.wait_for(('k', q)))}); PIN();")   # lgkmcnt is a 4-bit counter
            p.emit(mfma_qk(f"{sp}A[{q & 1}]", q % RING, "A", q >> 1, q < 2) + " PIN();")
        elif r == 1:
            p.emit(mfma_qk(f"{sp}B[{q & 1}]", q % RING, "B", q >> 1, q < 2) + " PIN();")
            if q + RING < 16:
                kread(q + RING)
        elif r == 2:
            p.emit(f"S8_WAIT({min(15, p.wait_for(('v', q)))}); PIN();")
            p.emit(mfma_pv("A", q & 3, q % RING, f"pw{b}A", q >> 2) + " PIN();")
        else:
            p.emit(mfma_pv("B", q & 3, q % RING, f"pw{b}B", q >> 2) + " PIN();")
            if q + RING < 16:
                vread(q + RING)
        if g < 16 and g % 2 == 1:                          # the step's 8 LDS-DMA pieces, one every other gap
            k = g >> 1
            p.emit(f"S8_DMA_K({k}, {kb_w});" if k < 4 else f"S8_DMA_V({k - 4});")
        el_ops(p, b, g)
        p.emit("PIN();")
    p.emit(f"S8_MASK_NEXT({sp});")
    return p.out


def gen_plain_qk(dst, kbase):
    out = []
    for kk in range(8):
        out.append(f"  S8_KREAD(0, {kk}, {kbase}); S8_KREAD(1, {kk}, {kbase + 8192}); S8_WAIT(0); PIN();")
        for h in (0, 1):
            for st in "AB":
                out.append("  " + mfma_qk(f"{dst}{st}[{h}]", h, st, kk, kk == 0) + " PIN();")
    return out


def main():
    print("// GENERATED by tools/gen_attn_s8.py -- do not edit.  S8_PART: 0/1 = step (parity), 4 = accumulator-file init, 5/6 = rescale O_A/O_B,")
    print("// 7/8 = read O_A/O_B into f32x16_t o[4], 9/10 = S(first)/S(second tile) the plain way.  Macros: qp_attn_s8.hip.")
    bodies = {0: gen_step(0), 1: gen_step(1), 4: gen_init(), 5: gen_rescale("A"), 6: gen_rescale("B"), 7: gen_readout("A"),
              8: gen_readout("B"), 9: gen_plain_qk("s0", K_BASE[0]), 10: gen_plain_qk("s1", K_BASE[1])}
    for i, (k, body) in enumerate(sorted(bodies.items())):
        print(("#if" if i == 0 else "#elif") + f" S8_PART == {k}")
        print("\n".join(body))
    print("#endif")


if __name__ == "__main__":
    main()
