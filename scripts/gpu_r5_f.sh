#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5f; mkdir -p gpurun_out/$TAG
( time timeout 1000 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/$TAG/pytest.log
( time timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ); echo "bench rc=$?"
python scripts/bench_summary.py gpurun_out/$TAG/bench.json | cut -c1-700
# RCCL: two ranks of the nccl backend on ONE device (the gather of SURVEY 8e), if the library permits it
MIBLAST_BENCH_SINGLE_DEVICE=1 MIBLAST_BENCH_BACKEND=nccl timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/nccl2.json 2> gpurun_out/$TAG/nccl2.err; echo "nccl two ranks on one device rc=$?"
tail -c 600 gpurun_out/$TAG/nccl2.err; head -c 300 gpurun_out/$TAG/nccl2.json
bash scripts/gpu_profile_r05.sh r05 > gpurun_out/$TAG/profile.log 2>&1; echo "profile rc=$?"; tail -8 gpurun_out/$TAG/profile.log
