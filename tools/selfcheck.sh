#!/bin/bash
# Runs ON THE GPU BOX: the maintained tools once each with short settings, against the in-tree libraries — "does it still run
# against this ABI". One line per tool (ok / FAILED + the last lines of its output).  usage: gpurun -- 'bash tools/selfcheck.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { name=$1; shift; out=$(timeout 300 "$@" 2>&1); rc=$?; if [ $rc -eq 0 ]; then echo "ok      $name"; else echo "FAILED  $name (rc $rc)"; echo "$out" | tail -5; fi; }
run engine_profile python tools/engine_profile.py
run chain_ab python tools/chain_ab.py --rounds 2 --reps 2
run kernel_times python tools/kernel_times.py
run nlm_ab python tools/nlm_ab.py
run conv_bench python tools/conv_bench.py
run train_bench python tools/train_bench.py
run eval_bench python tools/eval_bench.py
run isp_step_ab python tools/isp_step_ab.py --ops=0,5 --pairs=3
run engine_env_ab python tools/engine_env_ab.py "ADAYOLO_BNECK_WS=0" "ADAYOLO_BNECK_WS=1" --rounds 2 --reps 2
run eval_graph_prof python tools/eval_graph_prof.py 12
run k1_bench python tools/k1_bench.py --reps 8
run train_graph_ab python tools/train_graph_ab.py --iters 10 --rounds 1
run train_graph_hist python tools/train_graph_hist.py
run graph_launch_gap python tools/graph_launch_gap.py
run pipeline_graphs_ab python tools/pipeline_graphs_ab.py 10 1
run train_soak python tools/train_soak.py 300
run train_det_breakdown python tools/train_det_breakdown.py 8 512 5
run bneck_ws_lib_ab python tools/bneck_ws_lib_ab.py
