"""Batched many-frame mode across GPUs: frame sharding + the result all-gather.

Frames are independent (ORBextractor::operator() reads only its image and ctor constants;
a stereo pair stays on one GPU because Frame::ComputeStereoMatches needs both images), so a
batch of F frames is cut into contiguous blocks of F/G frames per GPU.
There is no data-path collective during compute; ONE all-gather per step gives every rank the
features of all frames.  Variable keypoint counts travel through the fixed-shape collective as
one fixed-capacity record per frame:

    [ cap x 28 B keypoints | cap x 32 B descriptors | cap x 4 B mvuRight | cap x 4 B mvDepth
      | int32 count | 12 B pad ]

Works on any torch device/backend (RCCL on GPUs, gloo on CPU for the tests).
"""
import torch

KP_BYTES, DESC_BYTES = 28, 32
TAIL_BYTES = 16


def shard_range(nframes, rank, world):
    """Contiguous block of frames owned by `rank`: sizes differ by at most one."""
    base, rem = divmod(nframes, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def record_bytes(cap):
    return cap * (KP_BYTES + DESC_BYTES + 4 + 4) + TAIL_BYTES


def pack_records(kps, desc, uright, depth, counts, out=None):
    """kps [B,cap,7] f32 (bit pattern of orbx_keypoint_t), desc [B,cap,32] u8, uright/depth
    [B,cap] f32, counts [B] i32 -> uint8 [B, record_bytes(cap)]."""
    B, cap = kps.shape[0], kps.shape[1]
    rb = record_bytes(cap)
    if out is None:
        out = torch.zeros((B, rb), dtype=torch.uint8, device=kps.device)
    if kps.is_cuda:   # ONE kernel of the HIP library on the current stream (the torch slicing below is five strided copies)
        from . import pack_records_device
        assert kps.is_contiguous() and desc.is_contiguous() and uright.is_contiguous() and depth.is_contiguous()
        assert counts.dtype == torch.int32 and counts.is_contiguous() and out.is_contiguous() and out.shape == (B, rb)
        pack_records_device(kps.data_ptr(), desc.data_ptr(), uright.data_ptr(), depth.data_ptr(), counts.data_ptr(), B, cap,
                            out.data_ptr(), torch.cuda.current_stream(kps.device).cuda_stream)
        return out
    o = 0
    for t, nb in ((kps, cap * KP_BYTES), (desc, cap * DESC_BYTES), (uright, cap * 4), (depth, cap * 4)):
        out[:, o:o + nb] = t.contiguous().reshape(B, -1).view(torch.uint8)
        o += nb
    out[:, o:o + 4] = counts.to(torch.int32).contiguous().reshape(B, 1).view(torch.uint8)
    return out


def unpack_records(rec, cap):
    """Inverse of pack_records -> dict(kps, desc, uright, depth, counts)."""
    F = rec.shape[0]
    o = 0
    res = {}
    for name, nb, dt, shape in (("kps", cap * KP_BYTES, torch.float32, (F, cap, 7)),
                                ("desc", cap * DESC_BYTES, torch.uint8, (F, cap, 32)),
                                ("uright", cap * 4, torch.float32, (F, cap)),
                                ("depth", cap * 4, torch.float32, (F, cap))):
        res[name] = rec[:, o:o + nb].contiguous().view(dt).reshape(shape)
        o += nb
    res["counts"] = rec[:, o:o + 4].contiguous().view(torch.int32).reshape(F)
    return res


def all_gather_records(rec, gathered=None, group=None, async_op=False):
    """One all-gather of the per-frame records of every rank (equal B on every rank).
    Returns (gathered [world*B, rb], work-or-None); rank r's frames sit at [r*B, (r+1)*B)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=torch.uint8, device=rec.device)
    work = dist.all_gather_into_tensor(gathered, rec, group=group, async_op=async_op)
    return gathered, work


class ResultRing:
    """The result buffers of the batched mode and their life cycle, shared by bench.py, the GPU tests and the gloo CPU tests.

    `nbuf` buffer sets; step i uses set j = i % nbuf:
        j = ring.acquire(i)      # waits until set j's previous all-gather (step i - nbuf) has finished: its pack[j] /
                                 # gath[j] may be overwritten only then
        ... the caller's kernels fill ring.kps[j], desc[j], cnt[j] (nimg images) and ur[j], dp[j], nm[j] (B frames) ...
        ring.publish(j)          # N > 1: ONE pack kernel + ONE all-gather of fixed-capacity records, asynchronous, so that
                                 # it overlaps the next step's kernels (on CPU tensors / gloo: torch slicing + gloo)
        ring.drain()             # wait for every outstanding all-gather
    Frames are the first B images of a set (a stereo pair's right image sits at slot B + b and is not exchanged)."""

    def __init__(self, nbuf, B, nimg, cap, device, world=1, gather=False, blocking_via_host=False, gather_B=None, force_gather=False):
        self.nbuf, self.B, self.nimg, self.cap, self.world = nbuf, B, nimg, cap, world
        # force_gather: pack + all-gather even in a one-rank communicator (valid for RCCL): the collective leg - communicator, its
        # stream, the async work handle, the hand-off from the handle's side stream - then runs on a box with ONE GPU
        self.gather = gather and (world > 1 or force_gather)
        # rows every rank contributes to the fixed-shape collective: the LARGEST block of any rank when a batch does not divide
        # evenly (shard_range: sizes differ by at most one); a smaller block leaves its last row zero (count 0)
        self.gB = B if gather_B is None else int(gather_B)
        assert self.gB >= B
        self.via_host = blocking_via_host   # rehearsal on a box with fewer GPUs than ranks: gloo moves host memory
        z = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=device) for _ in range(nbuf)]
        self.kps, self.desc, self.cnt = z((nimg, cap, 7), torch.float32), z((nimg, cap, 32), torch.uint8), z((nimg,), torch.int32)
        self.ur, self.dp, self.nm = z((B, cap), torch.float32), z((B, cap), torch.float32), z((B,), torch.int32)
        self.works = [None] * nbuf
        self.gathered_steps = [None] * nbuf     # step whose records gath[j] holds (set by acquire / drain once complete)
        self._pending = [None] * nbuf
        if self.gather:
            rb = record_bytes(cap)
            self.pack = z((self.gB, rb), torch.uint8)
            self.gath = z((world * self.gB, rb), torch.uint8)

    def acquire(self, i):
        j = i % self.nbuf
        self._finish(j)
        return j

    def _finish(self, j):
        if self.works[j] is not None:
            self.works[j].wait()
            self.works[j] = None
        if self._pending[j] is not None:
            self.gathered_steps[j], self._pending[j] = self._pending[j], None

    def publish(self, j, step=None):
        if not self.gather:
            return
        self.pack_set(j)
        self.gather_set(j, step)

    def pack_set(self, j):
        """ONE pack kernel (current stream): the B frame records of set j -> pack[j]."""
        if self.gather:
            pack_records(self.kps[j][:self.B], self.desc[j][:self.B], self.ur[j], self.dp[j], self.cnt[j][:self.B], out=self.pack[j][:self.B])

    def gather_set(self, j, step=None):
        """ONE all-gather of pack[j] (asynchronous; ordered behind the current stream's work)."""
        if not self.gather:
            return
        if self.via_host:
            g, _ = all_gather_records(self.pack[j].cpu())
            self.gath[j].copy_(g)
        else:
            _, self.works[j] = all_gather_records(self.pack[j], self.gath[j], async_op=True)
        self._pending[j] = step

    def drain(self):
        for j in range(self.nbuf):
            self._finish(j)

    def gathered(self, j):
        """dict(kps, desc, uright, depth, counts) of ALL ranks' frames of the step held by set j (after acquire/drain); rank r's
        frames are rows [r * gB, r * gB + its block size)."""
        return unpack_records(self.gath[j], self.cap)
