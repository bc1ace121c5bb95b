#!/bin/bash
# round 6: randomised differential runs of the HEAD build (scripts/gpu_fuzz.py: random structures x random lastz options, GPU vs oracle, byte for byte --
# PAF, HSP list, alignments, edit ops, twelve counters): the default path, and the round's new paths forced onto every case whatever its size -- the
# dense seed stage with the packed extension windows through the level-synchronous kernels, q batches with extent[] in the keys' scrambled order and
# batches through the bins, many small bins -- and a few chunk-scale cases (where the dense path is the default)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r6fuzz; mkdir -p gpurun_out/$TAG; rm -f gpurun_out/$TAG/*
run() { tag=$1; secs=$2; n=$3; seed=$4; ( timeout $secs python scripts/gpu_fuzz.py $n $seed ) > gpurun_out/$TAG/$tag.log 2>&1; echo "$tag rc=$? $(tail -1 gpurun_out/$TAG/$tag.log)"; grep -c MISMATCH gpurun_out/$TAG/$tag.log; }
run a_default 200 200 16000
MIBLAST_SEED_BATCHED=0 MIBLAST_SEED_PACKED=2 MIBLAST_UX_PACKED=2 MIBLAST_UNGAPPED=ux run b_packed_windows 200 200 17000
MIBLAST_SEED_BATCHED=0 MIBLAST_SEED_PACKED=2 MIBLAST_UX_PACKED=2 MIBLAST_UNGAPPED=ux MIBLAST_HIT_CAP=3000 run c_q_batches 200 150 18000
MIBLAST_SEED_BATCHED=0 MIBLAST_SEED_PACKED=2 MIBLAST_UX_PACKED=2 MIBLAST_UNGAPPED=ux MIBLAST_HIT_CAP=20000 MIBLAST_BIN_MEAN=40 run d_q_batches_bins 200 150 19000
MIBLAST_SEED_BATCHED=0 MIBLAST_SEED_PACKED=2 MIBLAST_UX_PACKED=2 MIBLAST_BIN_MEAN=40 run e_small_bins 200 150 20000
FUZZ_NMIN=1700000 FUZZ_NMAX=2600000 run f_chunk_scale 420 5 21000
