#!/usr/bin/env python3
"""bench.py — ORB extract+match throughput on MI355X (the BASELINE.json metric).

One step = one pass of the hot path over one batch of synthetic STEREO frames that are
already resident in HBM: ORB extraction of the left and right 1241x376 images (8 levels,
1000 features per image) + Frame::ComputeStereoMatches.  A "frame" is one stereo frame (two
images), the conservative reading of "frames/s ORB extract+match".  Each rank (one process per
GPU) owns its own frames (weak scaling); with N > 1 every step's results (left keypoints,
descriptors, counts, mvuRight, mvDepth) are all-gathered over RCCL/xGMI, overlapped with the
next step's compute.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, nfeatures, stereo)
    "kitti_stereo_1241x376_1000feat": (1241, 376, 1000, True),
    "kitti_stereo_1241x376_2000feat": (1241, 376, 2000, True),
    "euroc_stereo_752x480_1000feat": (752, 480, 1000, True),
    "mono_1241x376_1000feat": (1241, 376, 1000, False),
    "mono_640x480_1000feat": (640, 480, 1000, False),
    "mono_1920x1080_4000feat": (1920, 1080, 4000, False),
}
KITTI_FX, KITTI_BF = 718.856, 386.1448  # KITTI-00 calibration (fx, baseline*fx)


def level_pixels(w, h, nlevels=8, sf=1.2):
    """P = sum of inner level pixels (reference: src/ORBextractor.cc:1111-1112)."""
    s = np.float32(1.0)
    tot = 0
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        tot += int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv))
        s = np.float32(s * np.float32(sf))
    return tot


def _gen_pair(args):
    w, h, k, stereo = args
    synth = importlib.import_module("orb_slam2v2-1_amd.synth")
    if stereo:
        l, r = synth.stereo_pair_blocky(w, h, k)
        return l, r
    return synth.frame(w, h, k), None


def make_frames(w, h, n, k0, stereo, workers):
    import multiprocessing as mp
    jobs = [(w, h, k0 + i, stereo) for i in range(n)]
    if workers > 1 and n > 2:
        with mp.get_context("fork").Pool(min(workers, n)) as pool:
            out = pool.map(_gen_pair, jobs)
    else:
        out = [_gen_pair(j) for j in jobs]
    left = np.stack([o[0] for o in out])
    right = np.stack([o[1] for o in out]) if stereo else None
    return left, right


def _cpu_frame(args):
    """CPU baseline worker: the ORACLE (restated CPU path) on one frame."""
    w, h, nf, k, stereo = args
    import oracle
    synth = importlib.import_module("orb_slam2v2-1_amd.synth")
    t0 = time.perf_counter()
    if stereo:
        l, r = synth.stereo_pair_blocky(w, h, k)
    else:
        l, r = synth.frame(w, h, k), None
    tgen = time.perf_counter() - t0
    t0 = time.perf_counter()
    exl = oracle.Extractor(nf, 1.2, 8, 20, 7)
    kl, dl = exl.extract(l)
    stages = exl.stage_seconds.copy()
    if stereo:
        exr = oracle.Extractor(nf, 1.2, 8, 20, 7)
        kr, dr = exr.extract(r)
        stages += exr.stage_seconds
        pl = [exl.pyramid_level(i) for i in range(8)]
        pr = [exr.pyramid_level(i) for i in range(8)]
        mb = np.float32(KITTI_BF) / np.float32(KITTI_FX)
        oracle.stereo_match(kl, dl, kr, dr, pl, pr, exl.scale_factors, exl.inv_scale_factors, KITTI_BF, mb)
    return time.perf_counter() - t0, tgen, stages


def cpu_baseline(w, h, nf, stereo, budget_s=20.0):
    """Times the oracle on the host cores over a bounded sample of the same workload."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    cores = max(1, min(os.cpu_count() or 1, 16))
    t1, _, _ = _cpu_frame((w, h, nf, 0, stereo))  # one frame, one core: sizes the sample
    nframes = int(max(cores, min(8 * cores, budget_s * cores / max(t1, 1e-3))))
    jobs = [(w, h, nf, 100 + i, stereo) for i in range(nframes)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_frame, jobs)
    wall = time.perf_counter() - t0
    gen = sum(r[1] for r in res)
    work = sum(r[0] for r in res)
    # frame generation happens inside the workers too; remove its share of the wall time
    wall_work = wall * work / max(work + gen, 1e-9)
    st = np.sum([r[2] for r in res], 0)
    # (i) one thread, mono extract; (ii) two threads, stereo extract + ComputeStereoMatches, the way
    # the reference runs its two extractors (src/Frame.cc:78-81; ctypes releases the GIL during the oracle calls)
    import threading
    synth = importlib.import_module("orb_slam2v2-1_amd.synth")
    one_thread_mono = two_thread_stereo = None
    try:
        reps = 3
        img = synth.frame(w, h, 7)
        e1 = oracle.Extractor(nf, 1.2, 8, 20, 7)
        t0 = time.perf_counter()
        for _ in range(reps):
            e1.extract(img)
        one_thread_mono = reps / (time.perf_counter() - t0)
        if stereo:
            l, r = synth.stereo_pair_blocky(w, h, 7)
            exs = [oracle.Extractor(nf, 1.2, 8, 20, 7) for _ in range(2)]
            out = [None, None]

            def _work(i, im):
                out[i] = exs[i].extract(im)
            mb = np.float32(KITTI_BF) / np.float32(KITTI_FX)
            t0 = time.perf_counter()
            for _ in range(reps):
                ts = [threading.Thread(target=_work, args=(i, im)) for i, im in enumerate((l, r))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                oracle.stereo_match(out[0][0], out[0][1], out[1][0], out[1][1], [exs[0].pyramid_level(i) for i in range(8)],
                                    [exs[1].pyramid_level(i) for i in range(8)], exs[0].scale_factors, exs[0].inv_scale_factors,
                                    KITTI_BF, mb)
            two_thread_stereo = reps / (time.perf_counter() - t0)
    except Exception:
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(nframes / wall_work, 3), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": "%d synthetic %s frames %dx%d, %d features, CPU oracle (restated CPU path, gcc -O3 -march=native, "
                  "scalar FAST) on %d processes; single-core %.3f s/frame" % (
                      nframes, "stereo" if stereo else "mono", w, h, nf, cores, t1),
        "stage_share": {k: round(float(v / max(st.sum(), 1e-9)), 3) for k, v in
                        zip(["pyramid", "fast", "quadtree", "orientation", "blur", "descriptor"], st)},
        "cpu_model": model, "host_cores_visible": os.cpu_count(),
        "one_thread_mono_images_per_s": None if one_thread_mono is None else round(one_thread_mono, 2),
        "two_thread_stereo_frames_per_s": None if two_thread_stereo is None else round(two_thread_stereo, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-steps", type=int, default=200,
                    help="untimed steps BEFORE the W warm-up steps: the GPU clocks need ~0.2 s of load to settle after the "
                         "CPU-only set-up phase (3 warm-up steps measured 2.5 %% low)")
    ap.add_argument("--workload", default="kitti_stereo_1241x376_1000feat", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the result all-gather when N > 1")
    ap.add_argument("--streams", type=int, default=1,
                    help="S > 1: consecutive steps alternate over S extractor handles on S streams (independent steps overlap; "
                         "the default 1 keeps every kernel alone on the GPU so that its measured duration is its own)")
    ap.add_argument("--overlap-pass", action="store_true",
                    help="after the measurement, an extra pass of K steps alternating over 3 handles on 3 streams; its throughput "
                         "is reported beside the headline value (\"overlapped\").  Off by default so that the kernel launches of "
                         "the default command are all single-stream (rocprof averages = the reported kernel duration)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N > 1 on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run "
                  "--nproc-per-node %d" % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)

    w, h, nf, stereo = WORKLOADS[args.workload]
    B = args.batch
    nimg = 2 * B if stereo else B
    # host-side work that forks worker processes happens BEFORE this process touches the GPU
    workers = max(1, min((os.cpu_count() or 1) // max(world, 1), 16))
    left, right = make_frames(w, h, B, 1000 * rank, stereo, workers)
    imgs = np.concatenate([left, right]) if stereo else left        # slots [0,B) left, [B,2B) right
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, h, nf, stereo)

    pkg = importlib.import_module("orb_slam2v2-1_amd")
    batching = importlib.import_module("orb_slam2v2-1_amd.batching")
    pkg.lib()  # fails loudly if the HIP library is missing
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        print("bench.py: %d ranks but %d GPU(s): RCCL needs one GPU per rank" % (world, ndev), file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    d_imgs = torch.from_numpy(imgs).to(dev)

    S = max(1, args.streams)
    exs = [pkg.ORBextractor(nf, 1.2, 8, 20, 7, device=dev_index) for _ in range(S)]
    ex = exs[0]
    for e in exs:
        e(imgs[0])                  # plan for this image size; cap is now exact
    cap = ex.max_keypoints()
    mbf = KITTI_BF
    mb = float(np.float32(KITTI_BF) / np.float32(KITTI_FX))

    nbuf = max(3, S)          # >= 3: the overlapped pass below alternates over three streams
    kps = [torch.zeros((nimg, cap, 7), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    desc = [torch.zeros((nimg, cap, 32), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    cnt = [torch.zeros(nimg, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    ur = [torch.zeros((B, cap), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    dp = [torch.zeros((B, cap), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    nm = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    gather = world > 1 and not args.no_gather
    if gather:
        # one packed record per frame so that a step is ONE all-gather: kps | desc | uright | depth | count
        rec_bytes = batching.record_bytes(cap)
        pack = [torch.zeros((B, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        gath = [torch.zeros((world * B, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        works = [None] * nbuf
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
    ev_m0 = [torch.cuda.Event(enable_timing=True) for _ in range(max(1, min(args.steps, 10)))]
    ev_m1 = [torch.cuda.Event(enable_timing=True) for _ in range(max(1, min(args.steps, 10)))]

    def step(i, timed_idx=None):
        j = i % nbuf
        exi, stream = exs[i % S], streams[i % S]
        st = stream.cuda_stream
        if gather and works[j] is not None:
            works[j].wait()         # buffer j is free again (its all-gather finished)
            works[j] = None
        exi.extract_batch_device(d_imgs.data_ptr(), nimg, w, h, w, w * h, kps[j].data_ptr(), desc[j].data_ptr(),
                                 cnt[j].data_ptr(), cap, st)
        if stereo:
            if timed_idx is not None:
                ev_m0[timed_idx].record(stream)
            pkg.stereo_batch_device(exi, exi, B, 0, B,
                                    kps[j].data_ptr(), desc[j].data_ptr(), cnt[j].data_ptr(),
                                    kps[j][B:].data_ptr(), desc[j][B:].data_ptr(), cnt[j][B:].data_ptr(),
                                    cap, mbf, mb, ur[j].data_ptr(), dp[j].data_ptr(), nm[j].data_ptr(), st)
            if timed_idx is not None:
                ev_m1[timed_idx].record(stream)
        if gather:
            with torch.cuda.stream(stream):     # pack + collective are ordered behind this step's kernels
                batching.pack_records(kps[j][:B], desc[j][:B], ur[j], dp[j], cnt[j][:B], out=pack[j])
                if args.backend == "nccl":
                    _, works[j] = batching.all_gather_records(pack[j], gath[j], async_op=True)
                else:  # rehearsal: gloo moves host memory
                    g, _ = batching.all_gather_records(pack[j].cpu())
                    gath[j].copy_(g)

    def drain():
        if gather:
            for j in range(nbuf):
                if works[j] is not None:
                    works[j].wait()
                    works[j] = None
        torch.cuda.synchronize()

    for i in range(args.ramp_steps):    # clock ramp (untimed, not counted as warm-up steps)
        step(i)
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    # Timed region: only the dominant kernel (k_fast_cells) is bracketed by HIP events on its launch stream.  Every
    # recorded event idles the GPU for ~4.5 us, so the full stage breakdown is taken in a separate untimed pass below.
    ex.set_profiling(2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fast_timed_ms = float(ex.stage_ms()[0][1])        # k_fast_cells, averaged over the K timed steps
    nprof = max(1, min(args.steps, 10))               # untimed pass: events at every stage boundary
    ex.set_profiling(1)
    for i in range(nprof):
        step(args.warmup + args.steps + i, i)
    drain()
    stage_ms, ncalls = ex.stage_ms()
    ex.set_profiling(0)
    match_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_m0[:nprof], ev_m1[:nprof])])) if stereo else 0.0

    # Extra pass (--overlap-pass, 1 GPU): the same K steps alternating over THREE extractor handles on three streams.
    # Steps are independent, so the latency-bound kernels of one step (upper pyramid levels, quad-tree, stereo bins /
    # median) overlap the VALU-bound ones of another.  Reported beside `value`, not as `value`: with kernels sharing the
    # GPU a kernel's own duration cannot be measured, and the roofline above is about kernels measured alone.
    overlapped = None
    if world == 1 and S == 1 and args.overlap_pass:
        S3 = 3
        exs3 = [ex] + [pkg.ORBextractor(nf, 1.2, 8, 20, 7, device=dev_index) for _ in range(S3 - 1)]
        for e in exs3[1:]:
            e(imgs[0])
        st3 = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S3 - 1)]

        def step3(i):
            j, e, st = i % nbuf, exs3[i % S3], st3[i % S3].cuda_stream
            e.extract_batch_device(d_imgs.data_ptr(), nimg, w, h, w, w * h, kps[j].data_ptr(), desc[j].data_ptr(),
                                   cnt[j].data_ptr(), cap, st)
            if stereo:
                pkg.stereo_batch_device(e, e, B, 0, B, kps[j].data_ptr(), desc[j].data_ptr(), cnt[j].data_ptr(),
                                        kps[j][B:].data_ptr(), desc[j][B:].data_ptr(), cnt[j][B:].data_ptr(),
                                        cap, mbf, mb, ur[j].data_ptr(), dp[j].data_ptr(), nm[j].data_ptr(), st)
        for i in range(30):
            step3(i)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(args.steps):
            step3(i)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        overlapped = {"streams": S3, "value": round(B * args.steps / dt3, 2), "unit": "frames/s",
                      "ms_per_step": round(dt3 / args.steps * 1e3, 4), "steps": args.steps,
                      "note": "independent steps alternate over 3 handles on 3 streams; kernels share the GPU, so "
                              "per-kernel durations are not isolated - informational, not the headline value"}

    if rank == 0:
        frames = world * B * args.steps
        value = frames / dt
        P = level_pixels(w, h)
        counts = cnt[(args.warmup + args.steps - 1) % nbuf].cpu().numpy()
        navg = float(counts.mean())
        bytes_img = 3 * P + 60 * navg
        ncand_img = sum(len(ex.debug_level_points(l, 0, b=0)) for l in range(8))   # FAST candidates of image 0
        bytes_frame = (2 * bytes_img + 64 * navg) if stereo else bytes_img
        # dominant single kernel of the step (HIP events on the launch stream, averaged over the
        # timed region).  Algorithmic bytes per image: FAST+NMS reads every level once = P;
        # quad-tree reads its candidates; describe reads P + writes 60 N (SURVEY §8(d) split).
        kern = {
            "k_fast_cells": (fast_timed_ms, P * nimg),
            "k_octree": (float(stage_ms[2]), 8.0 * ncand_img * nimg),   # 4 B key + 2 B node index read, 2 B written
            "k_describe": (float(stage_ms[3]), (P + 60 * navg) * nimg),
        }
        dom = max(kern, key=lambda k: kern[k][0])
        dom_ms, dom_bytes = kern[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                pm = json.load(open(pmc_path))
                inst = [k for k in pm.get("kernels", {}) if k.split("<")[0] == dom]   # template instances: k_fast_cells<44>
                if pm.get("workload") == args.workload and pm.get("batch") == B and inst:
                    traffic = max(pm["kernels"][k]["hbm_bytes_per_launch"] for k in inst)
            except Exception:
                traffic = None
        out = {
            "metric": "frames/s ORB extract+match @1241x376 8-lvl 1000-feat" if args.workload.startswith("kitti_stereo_1241x376_1000")
            else "frames/s ORB extract+match (%s)" % args.workload,
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload, "frame": "stereo pair (2 images)" if stereo else "mono image",
                       "width": w, "height": h, "nlevels": 8, "scale_factor": 1.2, "nfeatures": nf,
                       "ini_th_fast": 20, "min_th_fast": 7, "frames_per_step_per_gpu": B, "clock_ramp_steps": args.ramp_steps,
                       "match": "Frame::ComputeStereoMatches" if stereo else "none",
                       "parallelism": "frames sharded over %d GPU(s)%s" % (world, ", results all-gathered (%s)" % ("RCCL" if args.backend == "nccl" else "gloo rehearsal") if gather else ""),
                       "avg_keypoints_per_image": round(navg, 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes),
                         "pipeline_GBps": round(bytes_frame * value / world / 1e9, 2),
                         "pipeline_frac": round(bytes_frame * value / world / 1e9 / 8000.0, 5),
                         "algorithmic_bytes_per_frame": int(bytes_frame)},
            "stage_ms_per_call": {"pyramid": round(float(stage_ms[0]), 4), "fast": round(float(stage_ms[1]), 4),
                                  "quadtree": round(float(stage_ms[2]), 4), "describe": round(float(stage_ms[3]), 4),
                                  "extract_total": round(float(stage_ms[4]), 4), "stereo_match": round(match_ms, 4),
                                  "images_per_call": nimg, "calls_averaged": ncalls,
                                  "fast_timed_region": round(fast_timed_ms, 4),
                                  "note": "stage breakdown from an untimed pass of %d steps after the timed region; "
                                          "roofline.kernel_ms is k_fast_cells over the %d timed steps" % (nprof, args.steps)},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if overlapped is not None:
            out["overlapped"] = overlapped
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
