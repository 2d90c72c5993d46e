"""Discrete-event model of the layer pipeline's schedule (no GPU): what the ViT run-ahead gate does to a pipe of pp stages.

Events per group g: V(g) = the group's features are on every rank (scatter + data-parallel tower + all-gather; needs EVERY rank's gate
open), S_s(g) / E_s(g) = stage s starts / ends its layers of group g.  Dependencies:
    S_0(g) >= V(g), E_0(g-1)            S_s(g) >= E_{s-1}(g) + hop, E_s(g-1)            E_s(g) = S_s(g) + tau
    V(g)   >= gate + vit, where gate = S_r(g-2) over the ranks r that gate (run ahead at most two groups)
  old rule: every stage gates on ITS OWN S_s(g-2)   -> V(g) >= max_s S_s(g-2) + vit
  new rule: only rank 0 (stage 0) gates              -> V(g) >= S_0(g-2) + vit
Prints the steady-state rate relative to one group per tau.  An analytic illustration of DESIGN 7.1's claim, not a measurement."""
import sys


def rate(pp, G=400, tau=1.0, vit=0.15, hop=0.01, every_stage_gates=False):
    S = [[0.0] * G for _ in range(pp)]
    E = [[0.0] * G for _ in range(pp)]
    for g in range(G):
        gate = 0.0
        if g >= 2:
            gate = max(S[s][g - 2] for s in range(pp)) if every_stage_gates else S[0][g - 2]
        V = gate + vit
        for s in range(pp):
            ready = V if s == 0 else E[s - 1][g] + hop
            S[s][g] = max(ready, E[s][g - 1] if g else 0.0)
            E[s][g] = S[s][g] + tau
    span = E[pp - 1][G - 1] - E[pp - 1][G // 2]
    return (G - 1 - G // 2) * tau / span


if __name__ == "__main__":
    print("stages  every stage gates (rounds 4a)   rank 0 gates (now)   2/(pp-1)")
    for pp in (2, 3, 4, 8):
        print(f"{pp:6d}  {rate(pp, every_stage_gates=True):28.3f}   {rate(pp):18.3f}   {min(1.0, 2 / (pp - 1)):8.3f}")
