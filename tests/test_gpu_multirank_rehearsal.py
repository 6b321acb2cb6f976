"""The N > 1 launch paths with the REAL kernels, on the one GPU of the test box: two ranks
share device 0 and talk over gloo (RCCL refuses two ranks on one device) — a rehearsal of launch / barrier / max-over-ranks /
gradient bucket / rank-0-only printing, so that the first multi-GPU lease is a one-shot. Throughput of these runs means
nothing and the lines say so. Both commands run as CHILD processes of pytest (this process has initialised the GPU and must
never exec); the `--gpus 2` parent they start is GPU-free and starts the ranks as its own children."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_line(cmd, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, *cmd], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]                 # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_device():
    line = _one_line(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-detail", "--no-cpu-baseline"],
                     {"BENCH_REHEARSAL": "1"})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 and line["config"]["per_gpu_batch"] == 8
    assert line["config"]["parallelism"] == "replicas x2" and "REHEARSAL" in line["data"]
    assert line["metric"].startswith("ISP+YOLO forward images/sec @1280x720 bs8") and line["value"] > 0
    assert line["launch_mode"] in ("pipelined", "graph", "eager") and line["dtype"] == "bf16"
    assert abs(line["value"] - 16 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.02 * line["value"]      # whole-job aggregate


def test_train_two_ranks_on_one_device():
    # (6 iterations: three ordinary ones, then the iteration as two hipGraphs around the gradient all-reduce — train.Trainer)
    line = _one_line(["-m", "adaptiveisp_amd.train", "--gpus", "2", "--iters", "6", "--warmup", "2", "--batch", "8", "--size", "512"],
                     {"ADAISP_DP_REHEARSAL": "1"})
    assert line["n_gpus"] == 2 and line["global_batch"] == 16 and line["per_gpu_batch"] == 8 and line["iters"] == 4      # timed = iters - warmup
    assert line["graph"] == "split"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "config"):
        assert k in line, k
    assert line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak" and "REHEARSAL" in line["data"]
    # ONE flattened bucket per model carries every gradient; the collective was issued and timed
    assert line["grad_buckets"] == 1 and line["grad_bucket_bytes"] > 30e6 and line["all_reduce_ms"] > 0
    assert line["value"] > 0
