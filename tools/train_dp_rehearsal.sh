#!/usr/bin/env bash
# Runs ON THE GPU BOX: the data-parallel training entry with TWO ranks sharing the one GPU of the box (gradients over gloo:
# ADAISP_DP_REHEARSAL=1 — a correctness rehearsal of the N > 1 path with the real kernels, not a measurement), then the
# single-rank run that IS a measurement. Usage: gpurun -- 'bash tools/train_dp_rehearsal.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
echo "== 2 ranks on one device (rehearsal)"
ADAISP_DP_REHEARSAL=1 timeout 600 python -m adaptiveisp_amd.train --gpus 2 --iters 8 --warmup 3 --batch 8 --size 512 2>&1 | grep '^{'
echo "== 1 rank"
python -m adaptiveisp_amd.train --iters 30 --warmup 5 2>&1 | grep '^{'
