"""Frame-sharded data parallelism: one process per GPU, every rank holds the whole
Gaussian cloud + Adam state and renders its own camera; ONE all-reduce(SUM) per step over RCCL/xGMI -- of the
compact [P,14] gradient with the HIP step driver (all_reduce_compact), of the packed [59 P] gradient
buffer under autograd (GradBucket / sync_gradients) (backend "nccl" on ROCm; "gloo" in the CPU tests), plus the
densification statistics (SUM, SUM, MAX).  The reference has no distributed code at all."""
import os

import torch
import torch.distributed as dist

from .model import PARAM_NAMES


# True: issue the collectives even in a 1-rank group.  Only the tests set it: a 1-rank "nccl" group on the one test GPU is
# the closest thing to the multi-GPU run a one-GPU box offers (RCCL kernels, their streams and events are all exercised;
# RCCL refuses two ranks on one device, so the 2-rank tests there run over gloo).
FORCE_COLLECTIVES = False


def _live():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for 1 rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if os.environ.get("FSGS_DIST_ONE_GPU") == "1":
        # smoke-testing the N > 1 code path on a box with ONE GPU: every rank on device 0, gloo carries the tensors
        # (RCCL refuses two ranks on one device).  Never set in production.
        local, backend = 0, "gloo"
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket:
    """Flat [P*59] fp32 buffer holding every Gaussian gradient; params' .grad are VIEWS into it so the
    all-reduce needs no pack/unpack copies (70.8 MB at P = 300k)."""

    def __init__(self, pc):
        self.shapes = {k: tuple(pc.params[k].shape) for k in PARAM_NAMES}
        self.sizes = {k: int(pc.params[k].numel()) for k in PARAM_NAMES}
        n = sum(self.sizes.values())
        any_p = pc.params["_xyz"]
        self.flat = torch.zeros((n,), dtype=torch.float32, device=any_p.device)
        self.views = {}
        off = 0
        for k in PARAM_NAMES:
            self.views[k] = self.flat[off:off + self.sizes[k]].view(self.shapes[k])
            off += self.sizes[k]

    def matches(self, pc):
        return all(tuple(pc.params[k].shape) == self.shapes[k] for k in PARAM_NAMES)

    def attach(self, pc, zero=True):
        """Point every parameter's .grad at its view of the bucket.  zero=True for autograd (which accumulates);
        the autograd-free stepper overwrites every element each step and skips the 70 MB memset."""
        if zero:
            self.flat.zero_()
        for k in PARAM_NAMES:
            pc.params[k].grad = self.views[k]

    def all_reduce(self):
        if _live():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


def all_reduce_compact(gc):
    """all-reduce(SUM) of the compact [P,14] gradient of fast_step.FastStepper (56 B per Gaussian instead of 236 B:
    the SH gradients are rank-1 in a rank-independent basis, see csrc/render.hip OUT_COMPACT).  16.8 MB at
    P = 300 k -- what actually has to cross the point-to-point xGMI links."""
    if _live():
        dist.all_reduce(gc, op=dist.ReduceOp.SUM)


class DirectAllReduce:
    """all-reduce(SUM) as a DIRECT reduce-scatter + all-gather: xGMI is point-to-point -- every GPU has
    a link to every other -- so rank r sends shard j of its gradient straight to rank j (ONE all-to-all: all N - 1 links of
    a GPU carry 1/N of the buffer at once, one hop each), sums the N shards it holds, and the reduced shards are
    all-gathered the same way: 2 exchange steps of (N - 1)/N of the bytes in total per GPU, against the 2 (N - 1)
    dependent steps of a ring.  Whether RCCL's own all_reduce already does as well on a given node is for the first
    multi-GPU run to say: `bench.py --gpus N` times both (`comm.ms`, `comm.direct_ms`); this class is the explicit form,
    usable as `reduce_compact=` of FastStepper.mapping_step (`bench.py --ar-algo direct`).  In place; any element count
    (a padded staging buffer when it is not a multiple of 4 N floats).  gloo runs the same calls in the CPU tests."""

    def __init__(self):
        self._recv = self._red = self._stage = None

    def _buffers(self, like, n, world):
        shard = -(-n // (4 * world)) * 4  # floats per rank, 16-byte granular
        if self._recv is None or self._recv.numel() != world * shard or self._recv.device != like.device:
            self._recv = torch.empty((world, shard), dtype=like.dtype, device=like.device)
            self._red = torch.empty((shard,), dtype=like.dtype, device=like.device)
            self._stage = torch.zeros((world * shard,), dtype=like.dtype, device=like.device)
        return shard

    def __call__(self, gc):
        if not _live():
            return
        world = dist.get_world_size()
        flat = gc.reshape(-1)
        if flat.data_ptr() != gc.data_ptr():
            raise ValueError("DirectAllReduce reduces in place: the gradient buffer must be contiguous")
        n = flat.numel()
        shard = self._buffers(flat, n, world)
        padded = n != world * shard
        send = flat
        if padded:  # the tail of the last shard is padding (zeros)
            self._stage[:n].copy_(flat)
            send = self._stage
        dist.all_to_all_single(self._recv.view(-1), send)         # shard j of every rank -> rank j
        torch.sum(self._recv, dim=0, out=self._red)               # this rank's shard of the sum
        dist.all_gather_into_tensor(send, self._red)              # every reduced shard -> everybody (in place of `send`)
        if padded:
            flat.copy_(self._stage[:n])


class PipelinedCompactReducer:
    """all-reduce of the compact gradient in `nchunks` row ranges on a second stream, the Adam kernel of chunk i
    running beside the all-reduce of chunk i+1:  t = AR + Adam / nchunks instead of AR + Adam.  Chunk boundaries are
    multiples of 256 Gaussians (the Adam kernel's workgroup; keeps every 16-byte access aligned)."""

    pipelined = True

    def __init__(self, nchunks=4):
        self.nchunks = max(1, int(nchunks))
        self.side = None

    def __call__(self, gc, adam_rows):
        P = int(gc.shape[0])
        if not _live():
            adam_rows(0, P)
            return
        per = -(-P // self.nchunks)
        per = max(256, -(-per // 256) * 256)
        bounds = [(lo, min(P, lo + per)) for lo in range(0, P, per)]
        if not gc.is_cuda:
            for lo, hi in bounds:
                dist.all_reduce(gc[lo:hi], op=dist.ReduceOp.SUM)
                adam_rows(lo, hi)
            return
        main = torch.cuda.current_stream(gc.device)
        if self.side is None or self.side.device != gc.device:
            self.side = torch.cuda.Stream(device=gc.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        done = []
        with torch.cuda.stream(self.side):
            for lo, hi in bounds:
                dist.all_reduce(gc[lo:hi], op=dist.ReduceOp.SUM)
                e = torch.cuda.Event()
                e.record(self.side)
                done.append(e)
        gc.record_stream(self.side)
        for (lo, hi), e in zip(bounds, done):
            main.wait_event(e)
            adam_rows(lo, hi)


class ProducerPipelinedReducer:
    """Producer-side pipelining of the step's one exchange: the per-Gaussian backward is launched in
    `nchunks` row chunks (fsgs_render_backward_compact_rows) and the all-reduce of chunk i starts on a second stream as
    soon as chunk i is produced -- beside the production of chunks i+1.. -- and Adam consumes chunk i as soon as its
    all-reduce is done, beside the all-reduces still in flight:

        main : blend_bwd | pre_bwd 0 | pre_bwd 1 | pre_bwd 2 | pre_bwd 3 |      Adam 0 | Adam 1 | Adam 2 | Adam 3
        comm :                       |   AR 0    |   AR 1    |   AR 2    |   AR 3   |

    instead of  blend_bwd | pre_bwd | AR | Adam.  What it can hide is the per-Gaussian backward (~50 us at C2) and all but
    the last chunk's Adam (~60 of 78 us); the price is nchunks - 1 extra collective launches.  Chunk boundaries are
    multiples of 256 Gaussians (the workgroup of both kernels).  With CPU tensors (gloo tests) the same calls run in
    order without streams."""

    producer = True
    pipelined = True

    def __init__(self, nchunks=4):
        self.nchunks = max(1, int(nchunks))
        self.side = None
        self._pending = []

    def bounds(self, P):
        """Row chunks of this step's production.  Called once at the start of a step's production: whatever an aborted
        earlier step left queued (an exception between produced() and finish()) is dropped here -- its rows may lie past
        the cloud's new size after a densification.  An empty cloud still gets ONE (0, 0) chunk, so that the backward
        call that also writes the step's scalar loss is made."""
        self._pending = []
        P = int(P)
        if P <= 0:
            return [(0, 0)]
        per = -(-P // self.nchunks)
        per = max(256, -(-per // 256) * 256)
        return [(lo, min(P, lo + per)) for lo in range(0, P, per)]

    def produced(self, gc, lo, hi):
        """rows [lo, hi) of gc are final on the current stream: start their all-reduce."""
        live = _live() and hi > lo
        if not gc.is_cuda or hi <= lo:
            if live:
                dist.all_reduce(gc[lo:hi], op=dist.ReduceOp.SUM)
            self._pending.append((lo, hi, None))
            return
        main = torch.cuda.current_stream(gc.device)
        if self.side is None or self.side.device != gc.device:
            self.side = torch.cuda.Stream(device=gc.device)
        ready = torch.cuda.Event()
        ready.record(main)
        done = torch.cuda.Event()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if live:
                dist.all_reduce(gc[lo:hi], op=dist.ReduceOp.SUM)
            done.record(self.side)
        gc.record_stream(self.side)
        self._pending.append((lo, hi, done))

    def finish(self, adam_rows):
        """Adam for every chunk, each as soon as its all-reduce has landed (in production order)."""
        pending, self._pending = self._pending, []
        for lo, hi, done in pending:
            if done is not None:
                torch.cuda.current_stream().wait_event(done)
            if hi > lo:
                adam_rows(lo, hi)

    def __call__(self, gc, adam_rows):
        """consumer-side fallback (several views per step: the gradient is only final after the last view)."""
        for lo, hi in self.bounds(gc.shape[0]):
            self.produced(gc, lo, hi)
        self.finish(adam_rows)


def sync_gradients(pc, bucket=None):
    """all-reduce(SUM) of the Gaussian gradients.  With a bucket attached before backward this is a
    single collective on one contiguous buffer; otherwise gradients are packed first."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    if bucket is not None and all(pc.params[k].grad is not None and
                                  pc.params[k].grad.data_ptr() == bucket.views[k].data_ptr() for k in PARAM_NAMES):
        bucket.all_reduce()
        return
    grads = [pc.params[k].grad if pc.params[k].grad is not None else torch.zeros_like(pc.params[k])
             for k in PARAM_NAMES]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for k, g in zip(PARAM_NAMES, grads):
        n = g.numel()
        pc.params[k].grad = flat[off:off + n].view_as(g)
        off += n


def sync_densification_stats(pc):
    """xyz_gradient_accum SUM, denom SUM, max_radii2D MAX (scene/gaussian_model.py:678-681,
    train.py:299-303) so every rank takes identical clone/split/prune decisions."""
    if not _live():
        return
    v = pc.variables
    packed = torch.cat([v["xyz_gradient_accum"].reshape(-1), v["denom"].reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    P = v["denom"].numel()
    v["xyz_gradient_accum"].copy_(packed[:P].view_as(v["xyz_gradient_accum"]))
    v["denom"].copy_(packed[P:].view_as(v["denom"]))
    dist.all_reduce(v["max_radii2D"], op=dist.ReduceOp.MAX)
