"""Worker of tests/test_rccl_one_rank_gpu.py (run as a child process: the process group is this process's own).

The software-pipelined step of pipeline.FrontEnd with the result all-gather over RCCL in a ONE-rank communicator - a box with one
GPU cannot hold two RCCL ranks, but a one-rank communicator is valid and takes the same route through the code: communicator
set-up, pack kernel on the handle's side stream, all_gather_into_tensor(async_op=True) issued from the front end's collective
stream behind an event of the side stream, the work handle waited for when the buffer set comes round again.  THREE different
resident batches rotate through the handle (as in test_bench_step_rotating_batches); every gathered record still held at the end
must equal the CPU oracle of ITS batch.  Prints one JSON line."""
import importlib
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import oracle
    oracle.build()
    ref = importlib.import_module("oracle.reference_frames")
    w, h, nf, B, nsets, steps = 752, 480, 600, 6, 3, 8
    mbf, fx = 386.1448, 718.856
    mb = float(np.float32(mbf) / np.float32(fx))
    # the oracle's worker pool forks: before this process touches the GPU
    exps = [ref.run_pool(ref.stereo_frame, [(w, h, nf, 2700 + 50 * s + i, mbf, mb) for i in range(B)]) for s in range(nsets)]
    import torch
    import torch.distributed as dist
    pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
    pkg = importlib.import_module("orb_slam2v2-1_amd")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "modes": {}}
    for mode in ("pipelined_late", "pipelined", "plain"):
        fe = pl.FrontEnd(w, h, nf, True, B, prefetch=(mode != "plain"), lag_stereo=mode.startswith("pipelined"),
                         stereo_late=(mode == "pipelined_late"), world=1, gather=True, force_gather=True)
        assert abs(fe.mb - mb) < 1e-12
        fe.upload(np.stack([e["left"] for e in exps[0]]), np.stack([e["right"] for e in exps[0]]))
        for k in range(1, nsets):
            fe.upload_more(np.stack([e["left"] for e in exps[k]]), np.stack([e["right"] for e in exps[k]]))
        assert fe.ring.gather and not fe.ring.via_host
        for i in range(steps):
            fe.step(i)
        fe.drain()
        bad, checked = [], 0
        for i in range(steps - fe.ring.nbuf, steps):
            j = i % fe.ring.nbuf
            if fe.ring.gathered_steps[j] != i:
                bad.append("set %d holds the gathered records of step %s, expected %d" % (j, fe.ring.gathered_steps[j], i))
                continue
            g = {k: v.cpu().numpy() for k, v in fe.ring.gathered(j).items()}
            exp = exps[i % nsets]
            for b in range(B):
                e, n = exp[b], int(g["counts"][b])
                if n != len(e["kl"]):
                    bad.append("step %d frame %d: count %d != %d" % (i, b, n, len(e["kl"])))
                    continue
                k = np.frombuffer(g["kps"][b, :n].tobytes(), pkg.KP_DTYPE)
                m = ref.image_mismatch(k, g["desc"][b, :n], e["kl"], e["dl"])
                if m:
                    bad.append("step %d frame %d: %s" % (i, b, m))
                if g["uright"][b, :n].tobytes() != e["uright"].tobytes() or g["depth"][b, :n].tobytes() != e["depth"].tobytes():
                    bad.append("step %d frame %d: gathered mvuRight / mvDepth differ" % (i, b))
                checked += 1
        out["modes"][mode] = {"bad": bad[:10], "frames_checked": checked, "collective_stream_used": fe._coll is not None}
        del fe
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
