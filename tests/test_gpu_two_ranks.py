"""Two processes on the box run the graphed multi-rank step (the capability of ``train_kpcn.py:256-271``).

No multi-GPU node has been available to the builder or the driver, so the step exactly as rank k of N runs it -- graph A (forward,
backward, gradient hand-over, guard flag) | three asynchronous all-reduces of the gradient buckets | graph B (global guard, loss sums,
scale -> clip -> Adam) -- has only ever run with N = 1 on a GPU.  Here a CHILD process runs ``bench.py --gpus 2 --backend gloo
--share-gpu``: bench.py starts its two workers itself before anything touches the GPU (one rank per process, both on cuda:0;
RCCL refuses two ranks on one device, so the collective is gloo's: the printed throughput is meaningless and not checked).
What is checked is the structure: both ranks step, the launch form is the split tail, the losses are finite, and the replicas
hold bit-identical parameters after the steps (the same summed gradients went into the same Adam state on both).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_run_the_graphed_split_tail_step_on_one_gpu():
    env = dict(os.environ)
    env.pop("WCMC_DEBUG_LIB", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, "bench.py --gpus 2 failed:\n%s\n%s" % (proc.stdout[-3000:], proc.stderr[-3000:])
    start = proc.stdout.index('{"metric"')
    line, _ = json.JSONDecoder().raw_decode(proc.stdout[start:])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    assert line["collective_backend"].startswith("gloo")
    assert line["rccl_ranks"] == 0                                   # (nothing here claims RCCL traffic)
    # both ranks stepped, and by the split tail (two graph replays around the eager all-reduces)
    rk = line["rank_ms_per_step"]
    assert rk["min"] > 0 and rk["max"] >= rk["min"]
    assert "two hipGraph replays per step around three eager asynchronous all-reduces" in line["config"]["launch"], line["config"]["launch"]
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    for k, v in line["losses_last_step"].items():
        assert v == v and abs(v) < 1e6, (k, v)
    assert line["allreduce"]["ranks"] == 2 and line["allreduce"]["messages"] == 3 and line["allreduce"]["bytes_per_rank"] > 40e6
    assert line["params_identical_across_ranks"] is True
