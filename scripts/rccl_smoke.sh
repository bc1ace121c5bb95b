#!/bin/bash
# RCCL smoke for a box with >= 2 GPUs (the builder's box has one: this has never run there -- DESIGN.md section 7).
#   bash scripts/rccl_smoke.sh [N=2] [workload=chr20]
# Runs the sharded chunk-scale workload on 1 GPU and on N GPUs with the nccl (= RCCL) backend, one process per GPU over xGMI, and asserts that
# the N-rank run gathered the SAME bytes: its paf_md5 equals the 1-GPU run's and every chunk pair equals its oracle digest (tests/golden/).
set -e
N=${1:-2}; WL=${2:-chr20}
cd "$(dirname "$0")/.."
OUT=${GRAFT_OUT:-gpurun_out/rccl_smoke}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIBLAST_BENCH_BACKEND=nccl
python bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/one_full.json > $OUT/one.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${PORT:-29517} bench.py --gpus $N --workload $WL --steps 2 --warmup 1 \
    --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/many_full.json > $OUT/many.json
python - "$OUT" "$N" <<'PY'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
one, many = json.load(open(out + "/one_full.json")), json.load(open(out + "/many_full.json"))
assert many["n_gpus"] == n and many["config"]["collective_backend"] == "nccl", many["config"]
assert one["parity"]["same_bytes"] and many["parity"]["same_bytes"], (one["parity"], many["parity"])
assert one["config"]["paf_md5"] == many["config"]["paf_md5"], (one["config"]["paf_md5"], many["config"]["paf_md5"])
print("rccl smoke ok: %d ranks over RCCL gathered the bytes of the 1-GPU run (md5 %s), %.1f -> %.1f ms per step" % (n, one["config"]["paf_md5"], one["ms_per_step"], many["ms_per_step"]))
PY
