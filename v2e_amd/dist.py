"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The hot path shards over *clips*: each clip owns its pixel state, so clips are
independent units and the compute needs no collective.  The one
exchange step BASELINE.json names is the all-gather(v) of the ranks' event streams, which
`EventStreamGatherer` runs on a side stream so that it overlaps the next step's kernels.
xGMI is point-to-point, so the payload is kept as one large message per rank per step
(a 300-frame step is ~10 M events per rank) rather than one per frame, and it travels in the
lossless 8-byte wire format of include/v2e_amd.h (float32 bits of t | x | y | polarity) instead of
float32[4] rows: one GPU emits ~100 GB/s of rows, eight of them more than a GPU's xGMI links can take in.
"""
import ctypes as C

import torch
import torch.distributed as dist


def clips_of_rank(n_clips, world, rank):
    """clip c runs on rank c mod world (round-robin; clips are independent)."""
    return [c for c in range(n_clips) if c % world == rank]


def pack_events64(ev, out=None):
    """[n,4] float32 rows (t, x, y, p) -> [n] int64 words, lossless (include/v2e_amd.h v2e_events_pack64)."""
    n = int(ev.shape[0])
    if out is None:
        out = torch.empty((n,), dtype=torch.int64, device=ev.device)
    if n == 0:
        return out
    if ev.is_cuda:
        from . import _capi
        s = C.c_void_p(torch.cuda.current_stream(ev.device).cuda_stream)
        _capi.check(_capi.lib().v2e_events_pack64(C.c_void_p(ev.data_ptr()), C.c_void_p(out.data_ptr()), n, s), "v2e_events_pack64")
    else:  # CPU tensors (gloo tests): the same bit layout with torch ops
        e = ev.contiguous()
        t = e[:, 0].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        out[:n] = (t << 32) | ((e[:, 1].to(torch.int64) & 0x3FFF) << 18) | ((e[:, 2].to(torch.int64) & 0x3FFF) << 4) | \
            (e[:, 3] > 0).to(torch.int64)
    return out


def unpack_events64(words):
    """Inverse of pack_events64: [n] int64 -> [n,4] float32 rows, bit for bit what was packed."""
    n = int(words.shape[0])
    ev = torch.empty((n, 4), dtype=torch.float32, device=words.device)
    if n == 0:
        return ev
    if words.is_cuda:
        from . import _capi
        s = C.c_void_p(torch.cuda.current_stream(words.device).cuda_stream)
        w = words.contiguous()
        _capi.check(_capi.lib().v2e_events_unpack64(C.c_void_p(w.data_ptr()), C.c_void_p(ev.data_ptr()), n, s), "v2e_events_unpack64")
    else:
        w = words.contiguous()
        ev[:, 0] = (w >> 32).to(torch.int32).view(torch.float32)  # arithmetic shift keeps the sign bit of t
        ev[:, 1] = ((w >> 18) & 0x3FFF).to(torch.float32)
        ev[:, 2] = ((w >> 4) & 0x3FFF).to(torch.float32)
        ev[:, 3] = (w & 1).to(torch.float32) * 2 - 1
    return ev


class EventStreamGatherer:
    """all-gather(v) of per-rank event lists [n_r,4] float32 -> every rank gets all of them.

    submit(ev, n) packs the first n rows of `ev` (device tensor) into a staging buffer (8 bytes per event) and
    enqueues, on a side stream: all-gather of the counts (read back on that stream: every rank pads to the same
    maximum), all-gather of the padded payload.  result() returns the per-rank [n_r,4] float32 lists (unpacked on demand) of the most recent
    completed submit, in rank order.  On CPU tensors (gloo, tests) everything runs inline.
    """

    def __init__(self, device, world, group=None):
        self.device = torch.device(device)
        self.world = world
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self.staging = [None, None]
        self.flip = 0
        self.done_evt = [None, None]
        self.out = None
        self.counts = None
        self.nmax = 0
        self.bytes_gathered = 0

    def _ensure(self, slot, rows):
        s = self.staging[slot]
        if s is None or s.shape[0] < rows:
            self.staging[slot] = torch.empty((max(rows, 1),), dtype=torch.int64, device=self.device)
        return self.staging[slot]

    def submit(self, ev, n):
        slot = self.flip
        self.flip ^= 1
        n = int(n)
        nt = torch.tensor([n], dtype=torch.int64, device=self.device)
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            if self.done_evt[slot] is not None:
                main.wait_event(self.done_evt[slot])  # previous gather out of this staging slot
            ready = torch.cuda.Event()
            # nt is allocated on the main stream and read by collectives on the side stream: tell the caching
            # allocator, or the block could be handed out again on the main stream before the side stream read it
            nt.record_stream(self.side)
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                counts = torch.empty(self.world, dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(counts, nt, group=self.group)
                # read back ON the stream the collective ran on (a read on the main stream would not be ordered
                # after it, and every rank must size the payload gather from the same numbers)
                counts_host = counts.cpu()
            nmax = max(int(counts_host.max()), 1)
            st = self._ensure(slot, nmax)
            pack_events64(ev[:n], st)  # main stream: the event buffer may be overwritten by the next step
            ready.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                out = self._out_buffer(nmax)
                dist.all_gather_into_tensor(out[:self.world * nmax], st[:nmax], group=self.group)
                self.done_evt[slot] = torch.cuda.Event()
                self.done_evt[slot].record(self.side)
        else:
            cl = [torch.zeros_like(nt) for _ in range(self.world)]
            dist.all_gather(cl, nt, group=self.group)
            counts_host = torch.cat(cl)
            nmax = max(int(counts_host.max()), 1)
            st = self._ensure(slot, nmax)
            pack_events64(ev[:n], st)
            pl = [torch.empty_like(st[:nmax]) for _ in range(self.world)]
            dist.all_gather(pl, st[:nmax].contiguous(), group=self.group)
            out = torch.cat(pl)
        self.out, self.counts, self.nmax = out, [int(c) for c in counts_host.tolist()], nmax
        self.bytes_gathered += self.world * nmax * 8

    def _out_buffer(self, nmax):
        out = self.out
        if out is None or not out.is_cuda or out.shape[0] < self.world * nmax:
            out = torch.empty((self.world * nmax,), dtype=torch.int64, device=self.device)
            out.record_stream(self.side)
        return out

    def wait(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)

    def result(self):
        """Per-rank event lists [n_r,4] float32 of the last submit, in rank order (call wait() first)."""
        self.wait()
        c = self.counts
        return [unpack_events64(self.out[r * self.nmax: r * self.nmax + c[r]]) for r in range(self.world)]
