"""The batched many-frame step that bench.py times and tests/test_bench_step_gpu.py checks against the oracle:

    B frames already resident in HBM  ->  ORBextractor::operator() on every image (ONE orbx_extract_batch_device call:
    left images in slots [0,B), right images in [B,2B) of the same handle)  ->  Frame::ComputeStereoMatches for the B
    pairs (orbm_stereo_batch_device with hl == hr)  ->  with N > 1 ranks: one packed record per frame, one all-gather.

Software pipelining across steps (one handle, results identical, each step still does one of everything): after the FAST stage
of step i the GPU has three independent things to do - gather / quad-tree / descriptors of step i (main stream), the stereo
matcher of step i-1 and the pyramid of step i+1 (both on a side stream, in that order).  The latency-bound kernels of one fill
the idle issue slots of the others; FAST itself always runs alone.  `prefetch` / `lag_stereo` switch the two halves off.

Reference call sites: src/Frame.cc:78-81 (two extractors), :84 ComputeStereoMatches, src/ORBextractor.cc:1043-1105.
Host-side plumbing only (torch = device memory, streams, torch.distributed); every computation is a kernel of the HIP
library, and there is no CPU fallback: constructing a FrontEnd without a GPU raises.
"""
import numpy as np
import torch

from . import KP_DTYPE, ORBextractor, stereo_batch_device
import sys as _sys
_pkg = _sys.modules[__package__]
from .batching import ResultRing

KITTI_FX, KITTI_BF = 718.856, 386.1448  # KITTI-00 calibration (fx, baseline*fx)


def independent_stream(dev, busy, ncand=8):
    """A new torch stream whose commands do not queue behind those of the streams in `busy`.  HIP deals streams to a handful of
    hardware queues round-robin, and a copy submitted on a stream that shares its queue with the compute stream starts only
    when the kernels queued in front of it are done (seen in the kernel / copy trace of the end-to-end mode: every upload
    started right behind the FAST kernel of the main stream).  There is no API that tells the queue of a stream, so candidates
    are probed: a few ms of work goes to the busy stream, a tiny copy to the candidate - if the copy lands while the work is still
    running, the two do not share a queue."""
    a = torch.randn((4096, 4096), device=dev)
    small_h = torch.zeros(64, dtype=torch.uint8).pin_memory()
    small_d = torch.zeros(64, dtype=torch.uint8, device=dev)
    cands = [torch.cuda.Stream(dev) for _ in range(ncand)]
    for c in cands:
        ok = True
        for b in busy:
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(b):
                y = a @ a
                for _ in range(3):
                    y = y @ a
                evb = torch.cuda.Event()
                evb.record(b)
            with torch.cuda.stream(c):
                small_d.copy_(small_h, non_blocking=True)
                evc = torch.cuda.Event()
                evc.record(c)
            evc.synchronize()
            if evb.query():          # the work was over before the copy landed: it queued behind it (or nothing can be told)
                ok = False
                break
        if ok:
            torch.cuda.synchronize(dev)
            return c
    torch.cuda.synchronize(dev)
    return cands[0]


class FrontEnd:
    def __init__(self, w, h, nfeatures, stereo, B, device_index=0, nbuf=3, streams=1, world=1, gather=False,
                 gather_via_host=False, mbf=KITTI_BF, fx=KITTI_FX, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, prefetch=True, lag_stereo=True,
                 stereo_late=None, gather_B=None, two_side=None, fast_alone=None, force_gather=False):
        if not torch.cuda.is_available():
            raise RuntimeError("orb_slam2v2-1_amd.pipeline.FrontEnd needs a GPU: the HIP path has no CPU fallback")
        self.w, self.h, self.nf, self.stereo, self.B = w, h, nfeatures, stereo, B
        self.nimg = 2 * B if stereo else B
        self.dev_index = device_index
        self.dev = torch.device("cuda", device_index)
        self.mbf = float(mbf)
        self.mb = float(np.float32(mbf) / np.float32(fx))
        self.prefetch = prefetch       # build the next step's pyramid beside this step's latency-bound kernels (same handle)
        self.S = max(1, streams)
        self.exs = [ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th, device=device_index) for _ in range(self.S)]
        self.ex = self.exs[0]
        # Where the pyramid built ahead starts (ORBX_OPT_PREFETCH_GATE).  Default: behind FAST(i), beside the quad-tree - a stage of serial
        # chains that leaves most of the GPU idle, so the pyramid rides free.  When the quad-tree launch holds more workgroups than the GPU takes at
        # once (1024-thread build: one per CU; 512-thread build: three) nothing rides free there, and for MONO frames nothing else runs beside
        # FAST (stereo: the matcher of the previous step does): the pyramid then starts with FAST(i).  Measured, 64 images 1920x1080: 1.145 ->
        # 1.124 ms per step; the same choice at 32 images (quad-tree not saturated) 0.601 -> 0.630, at 640x480 x 64 0.251 -> 0.259, for stereo
        # workloads +-0.5 % (HISTORY.md, round 5).
        wide = ((w - 26) // 30) * ((h - 26) // 30) >= 600         # a level of >= 600 FAST cells: the 1024-thread quad-tree kernels
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        if prefetch and not stereo and self.S == 1 and B * nlevels > cus * (1 if wide else 3) and 10 not in _pkg._default_options:      # (an A/B run that sets the option itself keeps its value)
            self.ex.set_option(10, 3)
        # the matcher of step i-1 is issued after the extraction of step i, on the side stream, behind that extraction's FAST stage;
        # with N > 1 ranks the pack kernel + all-gather of step i-1 follow it on that stream, so N = 1 and N > 1 run the same pipeline
        self.lag = bool(lag_stereo and prefetch and stereo and self.S == 1)
        # the handle's own side stream, probed against the main stream: two streams that share a hardware queue overlap nothing
        main0 = torch.cuda.current_stream(self.dev)
        self._side_raw = self.ex.side_stream_for(main0.cuda_stream) if (prefetch and self.S == 1) else self.ex.side_stream()
        self.side = torch.cuda.ExternalStream(self._side_raw, device=self.dev) if self.lag else None
        # Order of the two side-stream jobs behind FAST(i).  stereo_late (the default since round 4): pyramid(i+1) first, then matcher(i-1)
        # (three pyramid buffers) - the pyramid starts beside the quad-tree, the matcher runs beside the descriptor kernel, the next FAST
        # does NOT wait for it (below).  stereo_late = False: matcher(i-1), then pyramid(i+1) - the matcher beside the quad-tree, the pyramid beside the
        # descriptors (two buffers).  Measured per 64-frame step, round 4 (descriptor kernel 133 us alone): 1000 features 0.613 -> 0.607 ms,
        # round 3 (165 us): 2000 features 76.6 -> 77.6 k frames/s, 1000 features no difference.
        self.late = bool(self.lag and (True if stereo_late is None else stereo_late))
        if self.late:
            self.ex.set_pyramid_buffers(3)
        # two_side: the matcher of step i-1 on a SECOND side stream, started behind FAST(i) like the pyramid of step i+1 (three pyramid
        # buffers): the matcher beside the quad-tree, the pyramid beside quad-tree + descriptors, neither behind the other
        import os as _os
        # fast_alone (rounds 2-4 until this change: always): FAST(i+1) waits for matcher(i-1), the last job of the side stream, so that the
        # issue-bound FAST kernel has the GPU to itself.  It has no data to wait for (the matcher reads buffer set i-1 and the pyramid of
        # step i-1, FAST(i+1) writes the handle's cell lists and reads the pyramid of step i+1), and the wait left the main stream idle
        # from the end of the descriptors to the end of the matcher: ~50 us of a 612-us step.  Without it the matcher runs beside FAST
        # (62 -> 285 us, hidden; FAST 265 -> 277 us) and the step takes 0.573 ms: 104.8 -> 111.8 k frames/s (three runs each, +-0.1).
        self.fast_alone = bool(int(_os.environ.get("ORBX_FAST_ALONE", "0")) if fast_alone is None else fast_alone)
        self.two_side = bool(self.late and (int(_os.environ.get("ORBX_TWO_SIDE", "0")) if two_side is None else two_side))
        if self.two_side:
            self.fast_alone = True     # the pyramid of step i+2 (side) overwrites what matcher(i-1) (side2) reads: ordered through FAST(i+1)'s wait
        self.side2 = independent_stream(self.dev, [main0, self.side]) if self.two_side else None
        self._coll = None              # torch-native stream the all-gather of the pipelined mode is issued from (N > 1)
        self._ev_late = None           # event behind the last late matcher: with fast_alone the next FAST waits for it
        self._pend = None              # (buffer set, step) whose matcher has not been issued yet
        self._ev_side = {}             # buffer set -> event behind its matcher on the side stream
        self.streams = [torch.cuda.current_stream(self.dev)] + [torch.cuda.Stream(self.dev) for _ in range(self.S - 1)]
        self.d_imgs = None
        self.cap = None
        self.ring = None
        self._ring_args = (max(nbuf, self.S), world, gather, gather_via_host, gather_B, force_gather)
        self._hs = None                # host-streaming state (enable_host_streaming)

    def upload(self, left, right=None):
        """left / right: uint8 [B, h, w] host arrays -> HBM (slots [0,B) left, [B,2B) right); plans every handle."""
        imgs = np.concatenate([left, right]) if self.stereo else left
        assert imgs.shape == (self.nimg, self.h, self.w) and imgs.dtype == np.uint8
        for e in self.exs:
            e(imgs[0])                  # plan for this image size; max_keypoints() is now exact
        self.cap = self.ex.max_keypoints()
        self.d_imgs = torch.from_numpy(np.ascontiguousarray(imgs)).to(self.dev)
        self.d_sets = [self.d_imgs]
        nbuf, world, gather, via_host, gB, force = self._ring_args
        self.ring = ResultRing(nbuf, self.B, self.nimg, self.cap, self.dev, world=world, gather=gather,
                               blocking_via_host=via_host, gather_B=gB, force_gather=force)
        return self

    def upload_more(self, left, right=None):
        """One more resident batch of the same shape: step i then works on batch i % (number of batches)."""
        imgs = np.concatenate([left, right]) if self.stereo else left
        assert imgs.shape == (self.nimg, self.h, self.w) and imgs.dtype == np.uint8
        self.d_sets.append(torch.from_numpy(np.ascontiguousarray(imgs)).to(self.dev))
        return self

    def _imgs(self, i):
        return self.d_sets[i % len(self.d_sets)].data_ptr()

    def _match(self, exi, j, st, prev=False):
        r, B, cap = self.ring, self.B, self.cap
        stereo_batch_device(exi, exi, B, 0, B, r.kps[j].data_ptr(), r.desc[j].data_ptr(), r.cnt[j].data_ptr(),
                            r.kps[j][B:].data_ptr(), r.desc[j][B:].data_ptr(), r.cnt[j][B:].data_ptr(),
                            cap, self.mbf, self.mb, r.ur[j].data_ptr(), r.dp[j].data_ptr(), r.nm[j].data_ptr(), st, prev=prev)

    def _flush(self):
        """Issue the matcher still owed (the last step's): its pyramid is the handle's current one."""
        if self._pend is None:
            return
        j, step = self._pend
        self._pend = None
        main = self.streams[0]
        self.side.wait_stream(main)
        if self.two_side:
            self.side.wait_stream(self.side2)     # the matcher of the step before ran there and uses the same stereo scratch of the handle
        self._match(self.ex, j, self.side.cuda_stream)
        self._publish_side(j, step)
        main.wait_stream(self.side)

    def _publish_side(self, j, step, side=None):
        """N > 1: records of buffer set j packed and all-gathered right behind its matcher, in side-stream order (the collective
        itself runs on the backend's own stream and overlaps whatever follows)."""
        side = self.side if side is None else side
        if self.ring.gather:
            with torch.cuda.stream(side):
                self.ring.pack_set(j)
                ev = torch.cuda.Event()
                ev.record(side)
            # the collective is issued from a stream torch created itself (the side stream is the handle's, wrapped as an external
            # stream: fine for kernels, but the backend's stream bookkeeping is best left to a native one)
            if self._coll is None:
                self._coll = torch.cuda.Stream(self.dev)
            self._coll.wait_event(ev)
            with torch.cuda.stream(self._coll):
                self.ring.gather_set(j, step)

    # ---- end to end from host memory (src/ros_stereo.cc:133, src/Frame.cc:78-81: the reference receives its images on the host)
    def enable_host_streaming(self, nimgbuf=3):
        """Frames arrive in PINNED host memory and results leave to pinned host memory, both by asynchronous copies on their own
        streams, overlapped with the kernels of other batches:
            submit(i, left, right)   batch i: one hipMemcpyAsync host -> HBM on the upload stream into image buffer i % nimgbuf
            step(i)                  the usual step; its kernels (and the pyramid built ahead for it) wait for that upload only
            fetch(i)                 results of step i -> pinned host arrays on the download stream (call it after step(i + 1) has
                                     been issued when the pipelined matcher is on: that is when step i's matcher is in the queue)
            wait(i)                  host blocks until fetch(i) has landed; returns the pinned arrays
        Three image buffers: the upload of batch i + 2 must not wait for the pyramid of batch i (two buffers would chain the copy
        engine - the bottleneck, 60 MB per 64 stereo frames - to the compute queue).  Call after upload() (plans, result ring)."""
        assert self.ring is not None and self.S == 1
        # with the frames arriving over PCIe the step is upload-bound: the pyramid of batch i + 1 waits for its upload, and in the
        # pyramid-first order of the side stream that wait sits in front of the matcher of batch i - 1, which the next FAST waits for
        # (measured: 51.8 k frames/s against 54.1 k with the matcher first) - host streaming takes the matcher-first order
        self.drain()
        self.late = False
        self._ev_late = None
        # (the second side stream only exists in the late order: with it on and `late` off, nothing would order the pyramid built
        # ahead on `side` behind the matcher on `side2` that still reads the buffer it overwrites)
        self.two_side = False
        hs = type("HostStream", (), {})()
        hs.n = nimgbuf
        hs.d = [torch.empty((self.nimg, self.h, self.w), dtype=torch.uint8, device=self.dev) for _ in range(nimgbuf)]
        hs.side = torch.cuda.ExternalStream(self._side_raw, device=self.dev)
        hs.up = independent_stream(self.dev, [self.streams[0], hs.side])            # copies must not queue behind kernels
        hs.down = independent_stream(self.dev, [self.streams[0], hs.side, hs.up])
        hs.ev_up = [None] * nimgbuf          # upload of the batch in buffer k complete
        hs.ev_free = [None] * nimgbuf        # pyramid of the batch in buffer k built: the buffer may be overwritten
        r, nb = self.ring, self.ring.nbuf
        pin = lambda t: torch.empty(t.shape, dtype=t.dtype).pin_memory()
        hs.out = [{k: pin(getattr(r, k)[j]) for k in ("kps", "desc", "cnt", "ur", "dp", "nm")} for j in range(nb)]
        hs.ev_out = [None] * nb
        hs.ev_main = [None] * nb             # step's own kernels on the caller's stream done (buffer set j)
        self.d_sets = hs.d
        self._hs = hs
        return self

    def submit(self, i, left, right=None):
        """left / right: pinned uint8 tensors [B, h, w] (torch.Tensor.pin_memory()) holding batch i."""
        hs, B = self._hs, self.B
        k = i % hs.n
        if hs.ev_free[k] is not None:
            hs.up.wait_event(hs.ev_free[k])
        with torch.cuda.stream(hs.up):
            hs.d[k][:B].copy_(left, non_blocking=True)
            if self.stereo:
                hs.d[k][B:].copy_(right, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(hs.up)
        ev._batch = i
        hs.ev_up[k] = ev

    def fetch(self, i):
        hs, r = self._hs, self.ring
        j = i % r.nbuf
        if self.lag:
            if self._pend is not None and self._pend[0] == j:
                self._flush()
            ev = self._ev_side.get(j)
            if ev is not None:
                hs.down.wait_event(ev)
        hs.down.wait_event(hs.ev_main[j])
        with torch.cuda.stream(hs.down):
            for k in ("cnt", "kps", "desc", "ur", "dp", "nm"):
                if k in ("ur", "dp", "nm") and not self.stereo:
                    continue
                hs.out[j][k].copy_(getattr(r, k)[j], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(hs.down)
        hs.ev_out[j] = ev

    def wait(self, i):
        hs = self._hs
        j = i % self.ring.nbuf
        hs.ev_out[j].synchronize()
        return hs.out[j]

    def host_results(self, i):
        """wait(i) in the shape of results(): per-image (keypoints, descriptors) + per-frame (uright, depth, nmatch)."""
        o = self.wait(i)
        cnt = o["cnt"].numpy()
        kps = o["kps"].numpy().view(np.uint8).reshape(self.nimg, self.cap, 28)
        desc = o["desc"].numpy()
        imgs = [(np.frombuffer(kps[b, :int(cnt[b])].tobytes(), KP_DTYPE).copy(), desc[b, :int(cnt[b])].copy()) for b in range(self.nimg)]
        frames = []
        if self.stereo:
            ur, dp, nm = o["ur"].numpy(), o["dp"].numpy(), o["nm"].numpy()
            frames = [(ur[b, :int(cnt[b])].copy(), dp[b, :int(cnt[b])].copy(), int(nm[b])) for b in range(self.B)]
        return imgs, frames

    def step(self, i, ev_before_match=None, ev_after_match=None):
        r = self.ring
        j = r.acquire(i)
        exi, stream = self.exs[i % self.S], self.streams[i % self.S]
        st = stream.cuda_stream
        B, cap, w, h = self.B, self.cap, self.w, self.h
        hs = self._hs
        if hs is not None:
            stream.wait_event(hs.ev_up[i % hs.n])            # this batch is in HBM (long since, when its pyramid was built ahead)
            if hs.ev_out[j] is not None:
                stream.wait_event(hs.ev_out[j])              # the results this buffer set held have left for the host
        lag = self.lag and self.prefetch and ev_before_match is None
        if not lag:
            self._flush() if self.lag else None
        elif j in self._ev_side:
            stream.wait_event(self._ev_side.pop(j))     # the matcher that last wrote buffer set j (nbuf steps ago) is long done
        if lag and self.fast_alone and self._ev_late is not None:
            stream.wait_event(self._ev_late)            # late order: the previous step's matcher is the last job of the side stream
        exi.extract_batch_device(self._imgs(i), self.nimg, w, h, w, w * h, r.kps[j].data_ptr(), r.desc[j].data_ptr(),
                                 r.cnt[j].data_ptr(), cap, st)
        if hs is not None:
            k = i % hs.n
            if getattr(hs.ev_free[k], "_batch", None) != i:     # pyramid not built ahead: the buffer is free behind this extraction
                fr = torch.cuda.Event()
                fr.record(stream)
                fr._batch = i
                hs.ev_free[k] = fr
            if lag:                                              # keypoints / descriptors / counts of step i (its matcher: _ev_side)
                hs.ev_main[j] = torch.cuda.Event()
                hs.ev_main[j].record(stream)
        if lag:
            sd = self.side.cuda_stream
            if self.late:
                exi.stream_wait_fast_stage(sd)
                self._prefetch_next(exi, i, self.side)      # three buffers: the previous pyramid survives this one
            if self._pend is not None:
                pj, pstep = self._pend
                ms = self.side2 if self.two_side else self.side
                if not self.late or self.two_side:
                    exi.stream_wait_fast_stage(ms.cuda_stream)
                self._match(exi, pj, ms.cuda_stream, prev=True)       # step i-1: its pyramid is the buffer the call above swapped out
                self._publish_side(pj, pstep, ms)
                ev = torch.cuda.Event()
                ev.record(ms)
                self._ev_side[pj] = ev
                self._ev_late = ev if self.late else None
            if not self.late:
                self._prefetch_next(exi, i, self.side)      # behind the matcher: it reads that buffer
            self._pend = (j, i)
            return j
        if self.prefetch:
            # the next step of this handle reads the same resident images: its pyramid starts behind this step's FAST stage
            self._prefetch_next(exi, i, hs.side if hs is not None else None)
        if self.stereo:
            if ev_before_match is not None:
                ev_before_match.record(stream)
            self._match(exi, j, st)
            if ev_after_match is not None:
                ev_after_match.record(stream)
        if r.gather:
            with torch.cuda.stream(stream):     # pack + collective are ordered behind this step's kernels
                r.publish(j, i)
        if hs is not None:
            hs.ev_main[j] = torch.cuda.Event()
            hs.ev_main[j].record(stream)
        return j

    def _prefetch_next(self, exi, i, side):
        """Pyramid of step i + S built ahead on `side` (None: the handle's own side stream).  Host streaming: only once that batch
        has been submitted (its upload is then waited for on the side stream) - and the buffer is marked free behind the pyramid."""
        hs = self._hs
        sd = side.cuda_stream if side is not None else None
        if hs is None:
            exi.prefetch_batch_device(self._imgs(i + self.S), self.nimg, self.w, self.h, self.w, self.w * self.h, sd)
            return
        k = (i + 1) % hs.n
        ev = hs.ev_up[k]
        if ev is None or getattr(ev, "_batch", None) != i + 1:
            return                       # batch i + 1 has not been submitted yet: its own step builds the pyramid
        side.wait_event(ev)
        exi.prefetch_batch_device(self._imgs(i + 1), self.nimg, self.w, self.h, self.w, self.w * self.h, sd)
        fr = torch.cuda.Event()
        fr.record(side)
        fr._batch = i + 1
        hs.ev_free[k] = fr

    def drain(self):
        if self.lag:
            self._flush()
        self.ring.drain()
        torch.cuda.synchronize(self.dev)

    def results(self, j):
        """Host copies of buffer set j: list of per-image (keypoints, descriptors) + per-frame (uright, depth, nmatch).
        In the pipelined mode the stereo outputs of step i are produced on the side stream one step later: a matcher still owed for
        set j is issued here, and the copies wait for the side stream, so the caller never reads a set whose matcher is pending or
        running (drain() first is cheaper when several sets are read)."""
        r = self.ring
        if self.lag:
            if self._pend is not None and self._pend[0] == j:
                self._flush()
            ev = self._ev_side.get(j)
            if ev is not None:
                self.streams[0].wait_event(ev)      # (kept: the next step that refills set j waits for it as well)
        cnt = r.cnt[j].cpu().numpy()
        kps = r.kps[j].cpu().numpy().view(np.uint8).reshape(self.nimg, self.cap, 28)
        desc = r.desc[j].cpu().numpy()
        ur, dp, nm = r.ur[j].cpu().numpy(), r.dp[j].cpu().numpy(), r.nm[j].cpu().numpy()
        imgs = []
        for b in range(self.nimg):
            n = int(cnt[b])
            imgs.append((np.frombuffer(kps[b, :n].tobytes(), KP_DTYPE).copy(), desc[b, :n].copy()))
        frames = []
        if self.stereo:
            for b in range(self.B):
                n = int(cnt[b])
                frames.append((ur[b, :n].copy(), dp[b, :n].copy(), int(nm[b])))
        return imgs, frames
