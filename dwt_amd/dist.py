"""Multi-GPU plumbing: frames shard across ranks with no data-path collective; the one exchange
step is the gather of the encoded streams to rank 0 (frames are independent files:
fresh VLI order and run counter each, vli.h:33, rle.h:33).

Backend-agnostic: RCCL ("nccl") moves device tensors over xGMI; "gloo" (CPU tests, single-GPU
rehearsals) stages through host memory.
"""


def shard_frames(total, rank, world):
    """Contiguous block of frame indices [first, first+count) owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def _round8(v):
    return (int(v) + 7) // 8 * 8


class StreamGather:
    """Gather of every step's variable-length streams to rank `dst`, one step behind the encoder.

    Step k calls post(k, streams, lens) right after its encode: the byte lengths of all ranks are
    exchanged with one all_gather (8 bytes per frame) and copied to page-locked host memory without
    waiting.  collect(k) — called a step later, or at the end of the run — reads those lengths (the
    copy finished long ago, so the host does not stall on the device inside a step) and starts ONE
    group of point-to-point transfers: every stream travels with its own length straight from the
    encoder's output buffer into a row of `dst`'s slot (rows as wide as the step's longest stream);
    it overlaps whatever the caller runs next.  wait(k) orders the caller's stream behind that
    gather, after which slot k % slots may be reused.

    Rank `dst` keeps `slots` receive buffers of world x n x width bytes (grow-only): memory on the
    root is bounded by the step size, not by the length of the job — a consumer drains
    result(k) (device -> host / file / network) while later steps run.
    """

    def __init__(self, n, device, dst=0, group=None, slots=2):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.n, self.device, self.dst, self.group, self.slots = n, torch.device(device), dst, group, slots
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) != "nccl"   # gloo moves host memory only
        cdev = torch.device("cpu") if self.host_staged else self.device
        self.cdev = cdev
        pin = self.device.type == "cuda" and not self.host_staged
        self.all_lens = [torch.zeros((self.world * n,), dtype=torch.int64, device=cdev) for _ in range(slots)]
        self.all_lens_host = [torch.zeros((self.world * n,), dtype=torch.int64, pin_memory=pin) for _ in range(slots)]
        self.lens_ready = [None] * slots
        self.streams = [None] * slots
        self.send = [None] * slots
        self.recv = [[None] * self.world for _ in range(slots)] if self.rank == dst else None
        self.width = [0] * slots
        self.work = [None] * slots
        self.bytes_gathered = 0   # payload bytes that arrived on dst (its own rows included)
        self.collected = -1       # last step whose gather was started

    def post(self, k, streams, lens):
        """Exchange the lengths of step k.  streams: uint8 [n, stride] (kept by reference until collect(k)),
        lens: int64 [n] on the same device."""
        torch, dist = self.torch, self.dist
        s = k % self.slots
        self.wait(k - self.slots)
        self.streams[s] = streams
        mine = lens.to(self.cdev).contiguous()
        dist.all_gather_into_tensor(self.all_lens[s], mine, group=self.group)
        if self.host_staged:
            self.all_lens_host[s].copy_(self.all_lens[s])
            self.lens_ready[s] = None
        else:
            self.all_lens_host[s].copy_(self.all_lens[s], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.lens_ready[s] = ev

    def collect(self, k):
        """Start the gather of step k's streams (post(k) must have been called): every rank sends each of its
        streams straight out of the encoder's buffer (row i, its own length rounded up to 8 bytes — a contiguous
        view, nothing is copied or padded on the sender), rank `dst` receives them into rows of its slot; all
        sends and receives of the step go out as ONE group (grouped ncclSend/ncclRecv on RCCL: the peers' links into
        `dst` work side by side)."""
        torch, dist = self.torch, self.dist
        if k <= self.collected:
            return
        self.collected = k
        s = k % self.slots
        if self.lens_ready[s] is not None:
            self.lens_ready[s].synchronize()
        streams = self.streams[s]
        lens_all = self.all_lens_host[s].tolist()
        n = self.n
        width = min(streams.shape[1], _round8(max(lens_all)))
        self.width[s] = width
        ops, keep = [], []
        if self.rank == self.dst:
            need = n * width
            for r in range(self.world):
                flat = self.recv[s][r]
                if flat is None or flat.numel() < need:
                    flat = torch.empty((need + need // 8,), dtype=torch.uint8, device=self.cdev)
                    self.recv[s][r] = flat
                rows = flat[:need].view(n, width)
                if r == self.rank:
                    rows.copy_(streams[:, :width])   # dst's own streams: one strided copy into its slot
                    continue
                for i in range(n):
                    length = min(_round8(lens_all[r * n + i]), width)
                    if length:
                        ops.append(dist.P2POp(dist.irecv, rows[i, :length], r, self.group))
            self.bytes_gathered += int(sum(lens_all))
        else:
            for i in range(n):
                length = min(_round8(lens_all[self.rank * n + i]), width)
                if length:
                    row = streams[i, :length]
                    if self.host_staged:   # gloo moves host memory only
                        row = row.cpu()
                        keep.append(row)
                    ops.append(dist.P2POp(dist.isend, row, self.dst, self.group))
        self.send[s] = keep
        self.work[s] = dist.batch_isend_irecv(ops) if ops else []

    def wait(self, k):
        if k < 0:
            return
        s = k % self.slots
        if self.work[s] is not None:
            for w in self.work[s]:
                w.wait()
            self.work[s] = None
            self.send[s] = None

    def result(self, k):
        """On dst: (list of per-rank uint8 [n, width] tensors — views of slot k % slots, valid until post(k + slots) —
        and an int64 host tensor [world*n] of byte lengths, a copy) of step k."""
        s = k % self.slots
        self.wait(k)
        lens = self.all_lens_host[s].clone()
        if self.rank != self.dst:
            return None, lens
        need = self.n * self.width[s]
        return [self.recv[s][r][:need].view(self.n, self.width[s]) for r in range(self.world)], lens


def gather_streams(streams, lens, dst=0, group=None):
    """One-shot form: gather variable-length byte streams to rank `dst`.

    streams: uint8 [n, stride] (row i holds lens[i] valid bytes), lens: int64 [n], same n on every
    rank.  Returns (list of per-rank uint8 tensors [n, width], int64 tensor [world*n]) on dst and
    (None, lens_all) elsewhere."""
    g = StreamGather(lens.numel(), streams.device, dst=dst, group=group, slots=1)
    g.post(0, streams, lens)
    g.collect(0)
    return g.result(0)
