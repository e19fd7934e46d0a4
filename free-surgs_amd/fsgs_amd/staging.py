"""Frame staging for sequences that do not fit in HBM: the per-frame inputs of
PoseModel.record_data -- colours, mono-depths, forward flows (scene/pose_optimizer.py:441-460) -- live in PINNED host
memory, a bounded set of frames is resident on the device, and the next frame is copied on a copy stream while the
current one is being optimised.  The reference moves every frame's tensors host -> device inside every iteration
(`.cuda()` at train.py:174,253,255,390-393); the resident `FrameData` (trainer.py) removes those copies altogether and
stays the default -- 288 GB hold ~9 000 frames at 1280x1024 -- this is the path beyond that.

`StagedLane` is list-like (len / [] / iteration), so it stands in for the lists of a FrameData without the step drivers
noticing: lane[i] returns a device tensor that is valid on the CURRENT stream (it waits for the copy's event); a miss
copies in order on the current stream.  Device buffers come from a fixed pool; when a buffer is taken from the least
recently used frame, the copy stream first waits for an event recorded on the current stream AT THAT MOMENT -- every kernel
that reads the old frame was enqueued before the eviction was decided (the step drivers join their view streams into
the calling stream before they return), so the copy cannot overtake a reader.  On CPU tensors (the logic tests) the same
bookkeeping runs with plain copies.
"""
import collections

import numpy as np
import torch

from .trainer import FrameData


class Fence:
    """`stream` must not start what follows before everything enqueued on `cur` up to the FIRST hold() has finished;
    recorded once, waited for once per stream"""

    def __init__(self):
        self.event, self.held = None, set()

    def hold(self, cur, stream):
        if self.event is None:
            self.event = torch.cuda.Event()
            self.event.record(cur)
        if stream.cuda_stream not in self.held:
            self.held.add(stream.cuda_stream)
            stream.wait_event(self.event)


class StagedLane:
    def __init__(self, host_items, device, capacity, copy_stream=None):
        """host_items: list of equally shaped float32 CPU tensors (pinned when the device is a GPU) or None entries"""
        self.host = list(host_items)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # "cuda" -> the device tensors will report
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.capacity = max(1, int(capacity))
        self.copy_stream = copy_stream
        self.cache = collections.OrderedDict()  # frame -> (device tensor, copy-done event or None), least recently used first
        # frames whose copy was started by prefetch(protect=True) and that nobody has looked up yet: not evicted (ADVICE r3:
        # progressive_run asks for frame t+1 a whole frame cycle ahead, and the ~30 random keyframes of the mapping
        # iterations in between would otherwise push it out of a 4-buffer lane before it is read)
        # A protection is released by the frame's first lookup, by clear_protected() (Runner calls it when a phase ends), or
        # by AGE: a prefetch that is never read (an early exit, an exception, a caller asking the wrong lane) would otherwise
        # pin one buffer for the life of the lane, and in a small lane the frame in use becomes the only eviction victim
        # (ADVICE r4).  protected: frame -> value of the lane's load counter when it was protected; older than
        # PROTECT_LOADS loads = expired.  A lane needs >= 4 buffers for Runner's access pattern (frame t, keyframe, the two
        # look-aheads): StagedFrames clamps to that, a bare StagedLane with less simply protects less (see prefetch).
        self.protected = {}
        self.loads = 0
        self.free = []
        self.hits = self.misses = self.prefetched = 0
        first = next((h for h in self.host if h is not None), None)
        if first is not None:
            for h in self.host:
                if h is not None and (h.dtype != torch.float32 or tuple(h.shape) != tuple(first.shape) or not h.is_contiguous()):
                    raise ValueError("a staged lane holds contiguous float32 tensors of one shape")
            self.free = [torch.empty(first.shape, dtype=torch.float32, device=self.device)
                         for _ in range(min(self.capacity, len(self.host)))]
        self.shape = None if first is None else tuple(first.shape)

    # ---- list protocol ----
    def __len__(self):
        return len(self.host)

    def __iter__(self):
        for i in range(len(self.host)):
            yield self[i]

    def _cuda(self):
        return self.device.type == "cuda"

    def _victim(self):
        """least recently used resident frame that is not protected (the oldest protected one when all are: capacity
        smaller than the number of outstanding prefetches)"""
        for k in [k for k, born in self.protected.items() if self.loads - born > self.PROTECT_LOADS]:
            del self.protected[k]
        for k in self.cache:
            if k not in self.protected:
                return k
        k = next(iter(self.cache))
        self.protected.pop(k, None)
        return k

    PROTECT_LOADS = 64  # loads a never-read prefetch stays protected for (a frame cycle is ~30 keyframe loads)

    def clear_protected(self):
        """release every prefetch protection (the frames stay resident and age out in least-recently-used order)"""
        self.protected.clear()

    def _load(self, i, asynchronous, fence=None):
        evicted = not self.free
        old_done = None
        if self.free:
            buf = self.free.pop()
        else:
            buf, old_done = self.cache.pop(self._victim())
        done = None
        if self._cuda():
            cur = torch.cuda.current_stream(self.device)
            stream = self.copy_stream if (asynchronous and self.copy_stream is not None) else cur
            if evicted and stream is not cur:
                # everything enqueued so far may still read the evicted frame (one fence serves all the copies a caller
                # starts together: StagedFrames.prefetch hands the same one to its lanes)
                (fence if fence is not None else Fence()).hold(cur, stream)
            if old_done is not None:
                # the evicted frame's own copy may still be in flight on the copy stream (prefetched, never read): the new
                # copy into the same buffer is ordered behind it whichever stream it runs on
                stream.wait_event(old_done)
            with torch.cuda.stream(stream):
                buf.copy_(self.host[i], non_blocking=True)
                done = torch.cuda.Event()
                done.record(stream)
        else:
            buf.copy_(self.host[i])
        self.cache[i] = (buf, done)
        self.loads += 1

    def prefetch(self, i, fence=None, protect=True):
        """start copying frame i (no-op when it is resident, out of range or absent).  A prefetched frame stays resident
        until its first lookup (protect=False: it ages out like any other) -- least-recently-USED order alone would evict
        exactly the frames that were fetched ahead and not read yet: they are older than everything touched since"""
        if i is None or i < 0 or i >= len(self.host) or self.host[i] is None:
            return
        if protect and len(self.protected) < self.capacity - 1:  # (at least one buffer stays evictable)
            self.protected[int(i)] = self.loads
        if i in self.cache:
            self.cache.move_to_end(i)
            return
        self._load(i, asynchronous=True, fence=fence)
        self.prefetched += 1

    def ready(self, i):
        """the copy-done event of resident frame i (None on CPU or when it is not resident) for FURTHER streams that read
        the tensor lane[i] returned -- lane[i] itself only makes the current stream wait"""
        hit = self.cache.get(int(i))
        return None if hit is None else hit[1]

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += len(self.host)
        if self.host[i] is None:
            return None
        self.protected.pop(i, None)
        if i in self.cache:
            self.hits += 1
            self.cache.move_to_end(i)
        else:
            self.misses += 1
            self._load(i, asynchronous=False)
        buf, done = self.cache[i]
        if done is not None:
            torch.cuda.current_stream(self.device).wait_event(done)
        return buf


class RecentWindow:
    """list-like store of per-frame device tensors that keeps the `keep` most recently WRITTEN frames: the predicted depths
    of record_data (train.py:216-222), of which tracking frame t reads frame t-1 only."""

    def __init__(self, n, keep=4):
        self.n, self.keep = int(n), max(1, int(keep))
        self.items = collections.OrderedDict()

    def __len__(self):
        return self.n

    def __setitem__(self, i, v):
        i = int(i)
        self.items.pop(i, None)
        self.items[i] = v
        while len(self.items) > self.keep:
            self.items.popitem(last=False)

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self.items.get(i)  # None = not written yet or already dropped, as a fresh record_data entry


class StagedFrames(FrameData):
    """FrameData whose colours / mono-depths / forward flows are StagedLanes over pinned host memory.  `capacity` frames
    per lane stay on the device (>= 4: a tracking iteration reads frame t and the flows t-2, t-1; a two-view mapping
    iteration a keyframe as well); `prefetch(t)` starts the copies frame t will need."""

    def __init__(self, colors, monodeps, flows_fw=None, K=None, gt_w2c=None, device="cuda", capacity=8):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.copy_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        pin = (lambda a: a.pin_memory()) if dev.type == "cuda" else (lambda a: a)
        host = lambda seq: [None if a is None else pin(torch.as_tensor(np.ascontiguousarray(a) if not torch.is_tensor(a) else a)
                                                       .detach().to("cpu", torch.float32).contiguous()) for a in seq]
        cap = max(4, int(capacity))
        lanes = [StagedLane(host(colors), dev, cap, self.copy_stream), StagedLane(host(monodeps), dev, cap, self.copy_stream)]
        fw = None if flows_fw is None else StagedLane(host(flows_fw), dev, cap, self.copy_stream)
        super().__init__(lanes[0], lanes[1], flows_fw=fw, K=K, gt_w2c=gt_w2c)
        self.pred_depths = RecentWindow(len(lanes[0]), keep=cap)
        self.device = dev

    def prefetch(self, t, flows=True, protect=True, colors=True, monodeps=True):
        """frame t's colours and mono-depth and, with `flows`, the flows its tracking reads (t-1 -> t for the flow loss,
        t-2 -> t-1 for the rigid mask; trainer.Runner.tracking) -- a mapping view needs the first two only, a tracking
        frame no mono-depth.  What is fetched stays resident until it is first read (the next FRAME is asked for a whole
        frame cycle ahead, the next keyframe one iteration ahead), so only ask for what WILL be read: a lane whose
        prefetched frame is never looked up keeps one buffer less for everything else."""
        if t is None:
            return
        fence = Fence()
        if colors:
            self.colors.prefetch(t, fence, protect)
        if monodeps:
            self.monodeps.prefetch(t, fence, protect)
        if flows and self.flows_fw is not None:
            self.flows_fw.prefetch(t - 1, fence, protect)
            self.flows_fw.prefetch(t - 2, fence, protect)

    def clear_protected(self):
        """release the prefetch protections of every lane (end of a phase: whatever was fetched ahead and not read may go)"""
        for lane in (self.colors, self.monodeps, self.flows_fw):
            if lane is not None:
                lane.clear_protected()

    def stats(self):
        lanes = {"colors": self.colors, "monodeps": self.monodeps, "flows_fw": self.flows_fw}
        return {k: {"hits": v.hits, "misses": v.misses, "prefetched": v.prefetched} for k, v in lanes.items() if v is not None}
