"""N > 1 ranks of bench.py on the ONE GPU of the test box (gloo moves the records through host memory: RCCL needs one GPU per
rank), launched exactly as the driver launches a multi-GPU run.  What is checked is what RCCL would deliver on a node: rank 0
compares the ALL-GATHERED records of the last timed step - its own frames and the other rank's - with the CPU oracle of the
owning rank's seeds, every frame (bench.py --verify-all-gathered).

* BASELINE config 4 as written, at two ranks: --workload mono_1920x1080_4000feat --total-frames 128 (strong scaling, 64 frames
  of 1920x1080 / 4000 features per rank = the per-GPU share of 512 frames over 8 GPUs; batching.shard_range -> seeds f0..f1).
* The headline stereo step with the all-gather: the software-pipelined step (stereo matcher of step i-1 and its pack + all-gather
  on the side stream behind the FAST stage of step i) is the same for N = 1 and N > 1.
Reference: src/Frame.cc:61-120 (one Frame per image pair; frames are independent)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(nranks, extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--backend", "gloo",
           "--no-cpu-baseline", "--ramp-steps", "0", "--gen-workers", "6", "--verify-all-gathered"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert p.returncode == 0, "bench.py failed (%d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_config4_strong_preset_two_ranks_gathered_records():
    out = _run_bench(2, ["--workload", "mono_1920x1080_4000feat", "--total-frames", "128", "--steps", "3", "--warmup", "1"])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and cfg["world_size_observed"] == 2
    assert cfg["total_frames_per_step"] == 128 and cfg["frames_per_step_per_gpu"] == 64
    assert [r["first_seed"] for r in cfg["ranks"]] == [0, 64] and [r["frames_per_step"] for r in cfg["ranks"]] == [64, 64]
    assert out["verified"] is True, out["verified_note"]
    assert out["gathered_verified"] is True, out["gathered_verified_note"]
    assert out["gathered_verified_note"].startswith("128 frames"), out["gathered_verified_note"]
    assert cfg["avg_keypoints_per_image"] > 3800


def test_headline_step_two_ranks_same_pipeline_as_one():
    out = _run_bench(2, ["--workload", "kitti_stereo_1241x376_1000feat", "--batch", "16", "--steps", "7", "--warmup", "2"])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and cfg["world_size_observed"] == 2
    assert "stereo matcher of step i-1" in cfg["pipelining"], cfg["pipelining"]      # the N = 1 pipeline, not a reduced one
    assert [r["first_seed"] for r in cfg["ranks"]] == [0, 1000]
    assert out["verified"] is True, out["verified_note"]
    assert out["gathered_verified"] is True, out["gathered_verified_note"]
    assert out["gathered_verified_note"].startswith("32 frames"), out["gathered_verified_note"]


def test_bench_self_launch_two_ranks():
    """`python3 bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's one-GPU command): bench.py starts its
    own ranks as children before anything touches the GPU and relays rank 0's single JSON line.  The batch of 33 frames does not
    divide by two: blocks of 17 and 16 frames travel through the fixed-shape collective as 17 rows per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--ramp-steps", "0", "--gen-workers", "6"]
    p = subprocess.run(cmd + ["--batch", "8"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert p.returncode == 0, "bench.py failed (%d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_observed"] == 2
    assert out["verified"] is True and out["gathered_verified"] is True, (out["verified_note"], out["gathered_verified_note"])
    # uneven strong-scaling batch through the same self-launch
    p = subprocess.run(cmd + ["--workload", "mono_640x480_1000feat", "--total-frames", "33", "--verify-all-gathered"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert p.returncode == 0, "bench.py failed (%d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert [r["frames_per_step"] for r in out["config"]["ranks"]] == [17, 16] and out["config"]["total_frames_per_step"] == 33
    assert out["gathered_verified"] is True and out["gathered_verified_note"].startswith("33 frames"), out["gathered_verified_note"]
