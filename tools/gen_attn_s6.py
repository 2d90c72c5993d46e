#!/usr/bin/env python3
"""Generates quickvideo_amd/csrc/qp_attn_s6_iter.inc: the software-pipelined tile step of attn_fwd_kernel_s6.

One step of a wave is 32 MFMA "gaps" in two parts:
  Q part, gaps  0..15   S(next) = K(next).Q^T      (16 MFMAs, the two 32-key halves alternating)
  P part, gaps 16..31   O^T   += V(cur)^T.P(cur)   (16 MFMAs, 16-key chunk major, d block minor)
Every gap carries a few single-issue fillers placed by this script (hipcc never interleaves MFMA and VALU by itself;
`sched_barrier(0)` pins the order):
  * one softmax element of the current tile per gap as a three-stage VALU pipeline (A: x = s*c - m*c, B: p = 2^x one gap later,
    C: row sum / bf16 pack two gaps after that), so no instruction consumes a result of its own gap and P chunk c is complete
    one gap before P.V needs it; the pipeline runs across the Q part and the first 12 gaps of the P part;
  * the LDS fragment reads of the MFMAs four gaps ahead (K ring, then V ring), hand-issued, with counted `s_waitcnt lgkmcnt(N)`
    (LDS operations retire in order, so N = operations issued after the producer: that bookkeeping is why this is generated);
  * the LDS-DMA pieces of the coming K/V tiles (they retire on vmcnt, waited before the step's barrier);
  * the row max of S(next) in the last four gaps of the P part.
The Q part starts with no LDS read in flight and ends with the first four V fragment pairs of the following P part in flight;
the P part consumes them and ends with none.  Every wave runs the same cyclic sequence Q(t) P(t) Q(t+1) P(t+1) ...; what differs
between the two wave groups of an 8-wave workgroup is where their step (barrier to barrier) starts:
  order A (waves 0-3):  | Q(t) P(t) |  Q(t+1) P(t+1) | ...        barrier after every P part
  order B (waves 4-7):    Q(t) | P(t) Q(t+1) | P(t+1) ...          barrier after every Q part: half a step behind
so the two waves that share a SIMD are always in complementary parts (one MFMA + K-read heavy, one MFMA + V-read heavy) instead
of meeting in the same part after every barrier.  The part that opens a wave's step carries that step's DMA (dma_q / dma_p).
usage: python tools/gen_attn_s6.py > quickvideo_amd/csrc/qp_attn_s6_iter.inc
"""
K_BASE = [0, 16384]          # LDS byte offsets of the two K tile buffers (immediates); V buffers sit behind, addressed at run time
RING = 4


class Part:
    def __init__(self):
        self.out = []
        self.lds_seq = 0
        self.issued_at = {}

    def emit(self, s):
        self.out.append("  " + s)

    def lds_op(self, tag=None, n=1):
        self.lds_seq += n
        if tag is not None:
            self.issued_at[tag] = self.lds_seq - 1

    def wait_for(self, tag):
        return self.lds_seq - 1 - self.issued_at[tag]


def el_stage(p, sc, gap):
    """Element pipeline stages that run in `gap` (-7..27; negative = pre-phase of the Q part)."""
    ops = []
    e = gap + 4
    if 0 <= e <= 31:
        ops.append(f"S6_ELC({e & 3});" + (f" S6_PACK({e >> 1}, {(e - 1) & 3}, {e & 3});" if e & 1 else ""))
    e = gap + 6
    if 0 <= e <= 31:
        ops.append(f"S6_ELB({sc}, {e}, {e & 1}, {e & 3});")
    e = gap + 7
    if 0 <= e <= 31:
        ops.append(f"S6_ELA({sc}, {e}, {e & 1});")
    for o in ops:
        p.emit(o)
    if ops:
        p.emit("S6_ELKEEP();")


def vread(p, m, var):
    c, db = m >> 2, m & 3
    off = (((c >> 1) * 8 + (c & 1) * 4) * 4 + db) << 8
    p.emit(f"S6_VREAD({m % RING}, {var}, {off}, {off + (2 * 4 << 8)});")
    p.lds_op(("v", m), 2)


def gen_q(b):
    """Q part: S[1-b] = K[1-b].Q^T; elements 0..19 (+ ramp) of S[b]; ends with V pairs 0..3 (address `vrd_pref`) in flight."""
    p = Part()
    sc, sn = f"s{b}", f"s{1 - b}"
    kb_next, kb_w = K_BASE[1 - b], K_BASE[b]

    def kread(j):
        kk, h = j >> 1, j & 1
        p.emit(f"S6_KREAD({j % RING}, {kk}, {kb_next + h * 8192});")
        p.lds_op(("k", j))

    p.emit(f"{sn}[0] = (f32x16_t){{0}}; {sn}[1] = (f32x16_t){{0}};")
    for j in range(RING):
        kread(j)
    for gp in range(-7, 0):
        el_stage(p, sc, gp)
    p.emit("PIN();")
    for g in range(16):
        if g % 2 == 0:                        # one counted wait releases two fragments (they were issued 4 and 3 gaps ago)
            p.emit(f"S6_WAIT({p.wait_for(('k', g + 1))}); PIN();")
        p.emit(f"S6_QK({g % RING}, {g >> 1}, {sn}[{g & 1}]); PIN();")
        if g + RING < 16:
            kread(g + RING)
        else:
            vread(p, g + RING - 16, "vrd_pref")
        if g < 8:
            p.emit(f"if (dma_q) S6_DMA_K({g}, {kb_w});" if g < 4 else f"if (dma_q) S6_DMA_V({g - 4});")
        el_stage(p, sc, g)
        p.emit("PIN();")
    p.emit("S6_STAMP(1);")
    p.emit(f"S6_MASK_NEXT({sn});")
    return p.out


def gen_p(b):
    """P part: O^T += V^T.P with the probabilities of S[b] (elements 20..31 finish here); row max of S[1-b] in the last gaps.
    Enters with V pairs 0..3 in flight; the remaining pairs are read from `vrd_main`."""
    p = Part()
    sc, sn = f"s{b}", f"s{1 - b}"
    for m in range(RING):                     # the producer (previous Q part) issued these 8 reads last
        p.lds_op(("v", m), 2)
    kb_w = K_BASE[1 - b]                      # order B: this part opens the NEXT step (tile t+1, parity 1-b): its K DMA goes to K[1-b]
    for g in range(16, 32):
        m = g - 16
        if m % 2 == 0:
            p.emit(f"S6_WAIT({p.wait_for(('v', m + 1))}); PIN();")
        p.emit(f"S6_PV({m % RING}, {m >> 2}, {m & 3}); PIN();")
        if (m & 3) == 3:                      # hook after the four d blocks of a 16-key chunk (experiment: row sum on the matrix pipe)
            p.emit(f"S6_SUMP({m >> 2});")
        if m + RING < 16:
            vread(p, m + RING, "vrd_main")
        if m < 8:
            p.emit(f"if (dma_p) S6_DMA_K({m}, {kb_w});" if m < 4 else f"if (dma_p) S6_DMA_V({m - 4});")
        el_stage(p, sc, g)
        if g >= 28:
            p.emit(f"S6_MAX4({sn}, {g - 28});")
        p.emit("PIN();")
    return p.out


def main():
    print("// GENERATED by tools/gen_attn_s6.py -- do not edit.  Parts of one software-pipelined tile step of attn_fwd_kernel_s6.")
    print("// S6_PART selects: 0/1 = Q part with S parity 0/1, 2/3 = P part.  Both carry the LDS-DMA pieces of a step under the wave-")
    print("// uniform predicates dma_q / dma_p (the part that opens the wave's step issues them).  Macros: qp_attn_s6.hip.")
    parts = [gen_q(0), gen_q(1), gen_p(0), gen_p(1)]
    for i, body in enumerate(parts):
        print(("#if" if i == 0 else "#elif") + f" S6_PART == {i}")
        print("\n".join(body))
    print("#endif")


if __name__ == "__main__":
    main()
