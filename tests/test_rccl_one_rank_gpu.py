"""The RCCL leg of the batched step, executed on the ONE GPU of the test box: a one-rank communicator (backend "nccl" = RCCL).

Two ranks need two GPUs with RCCL, so the N > 1 tests of tests/test_multirank_gpu.py move the records with gloo through host
memory.  Here the collective itself is RCCL: communicator set-up, the pack kernel on the handle's side stream,
all_gather_into_tensor(async_op=True) from the front end's collective stream, work.wait() - everything except a second GPU.
Both tests run their GPU work in a child process (its own process group, a timeout of its own).
Reference: SURVEY.md section 8(e); src/Frame.cc:61-120 (frames are independent)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert p.returncode == 0, "%s failed (%d)\n%s\n%s" % (cmd[1], p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_pipelined_step_all_gather_over_rccl_one_rank_rotating_batches():
    out = _run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")])
    assert out["backend"] == "nccl" and out["world"] == 1
    for mode, r in out["modes"].items():
        assert not r["bad"], "%s: %s" % (mode, r["bad"])
        assert r["frames_checked"] == 18, (mode, r)          # 3 buffer sets x 6 frames
    assert out["modes"]["pipelined_late"]["collective_stream_used"] and out["modes"]["pipelined"]["collective_stream_used"]


def test_bench_force_gather_line():
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--batch", "16", "--steps", "7", "--warmup", "2",
                "--ramp-steps", "0", "--no-cpu-baseline", "--no-other-workloads", "--no-tracking", "--no-end-to-end", "--gen-workers", "6",
                "--verify-all-gathered"])
    cfg = out["config"]
    assert out["n_gpus"] == 1 and "RCCL" in cfg["parallelism"] and "one-rank" in cfg["parallelism"], cfg["parallelism"]
    assert out["verified"] is True, out["verified_note"]
    assert out["gathered_verified"] is True, out["gathered_verified_note"]
    assert out["gathered_verified_note"].startswith("16 frames"), out["gathered_verified_note"]
    g = out["gather_leg"]
    assert g["ranks"] == 1 and g["ms_per_step_alone"] > 0 and g["bytes_per_step_per_rank"] == 16 * g["record_bytes_per_frame"]
