#!/bin/bash
# round 5: randomised differential runs of the final build (GPU vs oracle, byte for byte): the default path, forced rejections inside the launch,
# small pieces (many hand-overs), and a few chunk-scale cases
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5fuzz; mkdir -p gpurun_out/$TAG
( timeout 280 python scripts/gpu_fuzz.py 250 5000 ) > gpurun_out/$TAG/a.log 2>&1; tail -2 gpurun_out/$TAG/a.log
( MIBLAST_RELAY_S0=64 MIBLAST_RELAY_S=256 MIBLAST_RELAY_W=64 MIBLAST_RELAY_INLINE_FORCE_REJECT=2 timeout 250 python scripts/gpu_fuzz.py 150 6000 ) > gpurun_out/$TAG/b.log 2>&1; tail -2 gpurun_out/$TAG/b.log
( MIBLAST_RELAY_S0=64 MIBLAST_RELAY_S=300 MIBLAST_RELAY_W=48 MIBLAST_RELAY_INLINE_FORCE_REJECT=-2 MIBLAST_RELAY_FORCE_REJECT=5 timeout 250 python scripts/gpu_fuzz.py 150 7000 ) > gpurun_out/$TAG/c.log 2>&1; tail -2 gpurun_out/$TAG/c.log
( FUZZ_NMIN=1700000 FUZZ_NMAX=2600000 timeout 400 python scripts/gpu_fuzz.py 5 8000 ) > gpurun_out/$TAG/d.log 2>&1; tail -2 gpurun_out/$TAG/d.log
