#!/bin/bash
# round 6: the driver's N > 1 command on ONE GPU (all ranks share it, gloo moves the bytes): functional records on the round's code
set -u
mkdir -p gpurun_out
export QP_BENCH_SINGLE_DEVICE=1
for N in 2 4; do
  QP_BENCH_FULL_RECORD=r6t_functional_n${N}_one_gpu_cfg2_full.json timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
    bench.py --gpus $N --steps 2 --warmup 1 --config cfg2 > gpurun_out/r6t_functional_n${N}_one_gpu_cfg2.json 2> gpurun_out/r6t_functional_n${N}.err
  echo "N=$N rc=$? $(wc -c < gpurun_out/r6t_functional_n${N}_one_gpu_cfg2.json) bytes"; grep "\[bench" gpurun_out/r6t_functional_n${N}.err | tail -3 | cut -c1-200
done
QP_BENCH_FULL_RECORD=r6t_functional_n8_one_gpu_cfg5_full.json timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 \
    bench.py --gpus 8 --steps 2 --warmup 1 --config cfg5 --no-pipeline > gpurun_out/r6t_functional_n8_one_gpu_cfg5.json 2> gpurun_out/r6t_functional_n8_cfg5.err
echo "cfg5 N=8 rc=$?"; grep "\[bench" gpurun_out/r6t_functional_n8_cfg5.err | tail -4 | cut -c1-200
python - <<'PY'
import json
for f in ("r6t_functional_n2_one_gpu_cfg2", "r6t_functional_n4_one_gpu_cfg2", "r6t_functional_n8_one_gpu_cfg5"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["config"]["parallelism"], d["first_token"], d.get("first_token_matches_record"), d.get("rccl_ranks"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
