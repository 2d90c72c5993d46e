import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import qp_oracle as O
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0")); D = 128
n, k, hkv = 300000, 60000, 2
rs = np.random.RandomState(n + k)
keys = torch.from_numpy(rs.standard_normal((hkv, n, D)).astype(np.float32)).to(torch.bfloat16)
kd = keys.cuda().contiguous()
ss = torch.empty(hkv, n, dtype=torch.float32, device="cuda")
ops.key_sumsq(kd, n * D, 0, n, hkv, D, ss)
idx = torch.full((k,), -1, dtype=torch.int32, device="cuda"); nb = torch.zeros(n, dtype=torch.int16, device="cuda")
ops.select_k_smallest(ss, hkv, n, k, idx, nb)
torch.cuda.synchronize()
ss_ref = O.key_sumsq_heads(O.torch_bf16_to_bits(keys))
ssg = ss.cpu().numpy()
print("ss mismatches", int((ssg.view(np.uint32) != ss_ref.view(np.uint32)).sum()))
nb_ref = O.key_norms_bf16(ss_ref); nbg = nb.cpu().numpy().view(np.uint16)
bad = np.nonzero(nbg != nb_ref)[0]
print("nb mismatches", len(bad), bad[:10], nbg[bad[:10]], nb_ref[bad[:10]])
if len(bad):
    t = bad[0]; s = np.float32(ss_ref[0, t]) + np.float32(ss_ref[1, t]); print("sum", s, np.sqrt(s), float(np.sqrt(np.float64(s))))
print("idx equal", np.array_equal(idx.cpu().numpy(), O.select_k_smallest(nbg, k)))
