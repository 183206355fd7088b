"""Multi-GPU plumbing: frames shard across ranks with no data-path collective; the one exchange
step is the gather of the encoded streams to rank 0 (frames are independent files:
fresh VLI order and run counter each, vli.h:33, rle.h:33).

Backend-agnostic: RCCL ("nccl") moves device tensors over xGMI; "gloo" (CPU tests, single-GPU
rehearsals) stages through host memory.  The RCCL path has not run on more than one GPU yet (the
build container has one GPU per call): what is tested is the same call sequence over gloo.
"""


def shard_frames(total, rank, world):
    """Contiguous block of frame indices [first, first+count) owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def _round8(v):
    return (int(v) + 7) // 8 * 8


def packed_offsets(lens, stride=None):
    """Byte offset of every stream inside a packed message (include/dwtx.h dwtx_pack_streams): stream i starts at the sum of
    the 8-byte-rounded lengths before it; returns n + 1 offsets, the last one the message's size."""
    off = [0]
    for v in lens:
        v = int(v)
        if stride is not None and v > stride:
            v = stride
        off.append(off[-1] + _round8(v))
    return off


def torch_pack(streams, lens_host, out):
    """The packing of dwtx_pack_streams with plain tensor copies (CPU tensors / no library at hand): one slice copy per stream."""
    off = packed_offsets(lens_host, streams.shape[1])
    for i in range(streams.shape[0]):
        n8 = off[i + 1] - off[i]
        if n8:
            out[off[i]:off[i + 1]].copy_(streams[i, :n8])
    return out


class Gathered:
    """Step k's streams on rank `dst`: stream(r, i) is frame i of rank r (a view, valid until post(k + slots)), lens the byte
    lengths of all world * n frames (a copy).  On the other ranks only `lens` is set."""

    def __init__(self, mode, n, lens, bufs=None, offsets=None, width=0):
        self.mode, self.n, self.lens, self.bufs, self.offsets, self.width = mode, n, lens, bufs, offsets, width

    def stream(self, r, i):
        length = int(self.lens[r * self.n + i])
        if self.mode == "packed":
            o = self.offsets[r][i]
            return self.bufs[r][o:o + length]
        return self.bufs[r][i, :length]

    def rank_bytes(self, r):
        return int(self.lens[r * self.n:(r + 1) * self.n].sum())


class StreamGather:
    """Gather of every step's variable-length streams to rank `dst`, one step behind the encoder.

    Step k calls post(k, streams, lens) right after its encode: the byte lengths of all ranks are
    exchanged with one all_gather (8 bytes per frame) and copied to page-locked host memory without
    waiting.  collect(k) — called a step later, or at the end of the run — reads those lengths (the
    copy finished long ago, so the host does not stall on the device inside a step) and starts the
    transfers; they overlap whatever the caller runs next.  wait(k) orders the caller behind that
    gather, after which slot k % slots may be reused.

    mode "packed" (default): ONE message per peer and step.  The sender moves its streams together
    (each rounded up to 8 bytes, dwtx_pack_streams: one kernel, `packer`) and sends the buffer; `dst` posts
    world - 1 receives of exactly the announced sizes.  A step of 64 frames on 8 ranks is 7 receives on
    rank 0 instead of 448, and no byte travels that is not stream.
    mode "rows": zero-copy on the sender — every stream travels with its own length straight out of
    the encoder's output buffer into a row of `dst`'s slot, all sends and receives of the step as one group
    (n * (world - 1) operations on `dst`).  Bytes of a row beyond round8(length) are NOT written: they hold
    whatever an earlier step left there.

    Rank `dst` keeps `slots` receive buffers (grow-only): memory on the root is bounded by the step
    size, not by the length of the job — a consumer drains result(k) while later steps run.
    """

    def __init__(self, n, device, dst=0, group=None, slots=2, mode="packed", packer=None):
        import torch
        import torch.distributed as dist

        assert mode in ("packed", "rows")
        self.torch, self.dist = torch, dist
        self.n, self.device, self.dst, self.group, self.slots, self.mode = n, torch.device(device), dst, group, slots, mode
        self.packer = packer      # packer(streams, lens_device, out) on the streams' device; None: torch_pack
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
        self.lens_dev = [None] * slots
        self.send = [None] * slots          # what must stay alive until the step's transfers are over
        self.sendbuf = [None] * slots       # packed mode: the sender's message (grow-only)
        self.recv = [[None] * self.world for _ in range(slots)] if self.rank == dst else None
        self.width = [0] * slots
        self.offsets = [None] * slots
        self.work = [None] * slots
        self.bytes_gathered = 0   # payload bytes that arrived on dst (its own rows included)
        self.messages_posted = 0  # point-to-point operations this rank has posted
        self.collected = -1       # last step whose gather was started

    def post(self, k, streams, lens):
        """Exchange the lengths of step k.  streams: uint8 [n, stride] (kept by reference until collect(k)),
        lens: int64 [n] on the same device."""
        torch, dist = self.torch, self.dist
        s = k % self.slots
        self.wait(k - self.slots)
        self.streams[s] = streams
        self.lens_dev[s] = lens
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

    def _grow(self, buf, need, device):
        if buf is None or buf.numel() < need:
            buf = self.torch.empty((need + need // 8 + 8,), dtype=self.torch.uint8, device=device)
        return buf

    def _pack(self, s, lens_mine_host, out):
        """this rank's streams of slot s, packed, in `out` (on the streams' device)"""
        streams = self.streams[s]
        if self.packer is not None and streams.device.type == "cuda":
            self.packer(streams, self.lens_dev[s], out)
        else:
            torch_pack(streams, lens_mine_host, out)
        return out

    def collect(self, k):
        """Start the gather of step k's streams (post(k) must have been called)."""
        torch, dist = self.torch, self.dist
        if k <= self.collected:
            return
        self.collected = k
        s = k % self.slots
        if self.lens_ready[s] is not None:
            self.lens_ready[s].synchronize()
        streams = self.streams[s]
        lens_all = self.all_lens_host[s].tolist()
        n, stride = self.n, streams.shape[1]
        ops, keep = [], []
        if self.mode == "packed":
            offs = [packed_offsets(lens_all[r * n:(r + 1) * n], stride) for r in range(self.world)]
            self.offsets[s] = offs
            mine = lens_all[self.rank * n:(self.rank + 1) * n]
            total = offs[self.rank][n]
            if self.rank == self.dst:
                for r in range(self.world):
                    self.recv[s][r] = self._grow(self.recv[s][r], offs[r][n], self.cdev)
                    if r == self.rank:
                        if total:   # dst's own streams: packed into its slot like everybody else's
                            if self.host_staged and streams.device.type == "cuda":
                                tmp = self._pack(s, mine, torch.empty((total,), dtype=torch.uint8, device=streams.device))
                                self.recv[s][r][:total].copy_(tmp)
                            else:
                                self._pack(s, mine, self.recv[s][r])
                    elif offs[r][n]:
                        ops.append(dist.P2POp(dist.irecv, self.recv[s][r][:offs[r][n]], r, self.group))
                self.bytes_gathered += int(sum(lens_all))
            elif total:
                self.sendbuf[s] = self._grow(self.sendbuf[s], total, streams.device)
                msg = self._pack(s, mine, self.sendbuf[s])[:total]
                if self.host_staged and msg.device.type != "cpu":
                    msg = msg.cpu()
                keep.append(msg)
                ops.append(dist.P2POp(dist.isend, msg, self.dst, self.group))
        else:
            width = min(stride, _round8(max(lens_all)))
            self.width[s] = width
            if self.rank == self.dst:
                need = n * width
                for r in range(self.world):
                    self.recv[s][r] = self._grow(self.recv[s][r], need, self.cdev)
                    rows = self.recv[s][r][:need].view(n, width)
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
        self.messages_posted += len(ops)
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
        """Step k's Gathered (see there): the streams on dst — views of slot k % slots, valid until post(k + slots) —
        and everywhere a copy of the byte lengths."""
        s = k % self.slots
        self.wait(k)
        lens = self.all_lens_host[s].clone()
        if self.rank != self.dst:
            return Gathered(self.mode, self.n, lens)
        if self.mode == "packed":
            return Gathered("packed", self.n, lens, bufs=list(self.recv[s]), offsets=self.offsets[s])
        need = self.n * self.width[s]
        return Gathered("rows", self.n, lens, bufs=[self.recv[s][r][:need].view(self.n, self.width[s]) for r in range(self.world)],
                        width=self.width[s])


def gather_streams(streams, lens, dst=0, group=None, mode="packed", packer=None):
    """One-shot form: gather variable-length byte streams to rank `dst`.

    streams: uint8 [n, stride] (row i holds lens[i] valid bytes), lens: int64 [n], same n on every
    rank.  Returns a Gathered: on dst with the streams, elsewhere with the lengths only."""
    g = StreamGather(lens.numel(), streams.device, dst=dst, group=group, slots=1, mode=mode, packer=packer)
    g.post(0, streams, lens)
    g.collect(0)
    return g.result(0)
