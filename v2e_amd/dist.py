"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The hot path shards over *clips*: each clip owns its pixel state, so clips are independent units
and the compute needs no collective.  The one exchange step BASELINE.json names is the all-gather(v) of the ranks' event
streams, which `EventStreamGatherer` runs on a side stream so that it overlaps the next step's kernels.  xGMI is
point-to-point, so the payload is one large message per rank per step (a 300-frame step is ~10 M events per rank), and it
travels in a lossless wire format instead of float32[4] rows (one GPU emits ~140 GB/s of rows, eight of them far more
than a GPU's xGMI links can take in):

  pack32 (sensors up to 2048 x 1024)  4 bytes per event (x | y << 11 | p << 21) + a run table: the rows of a run come in
         blocks of one time stamp -- all events of one (frame, iteration) share it -- so the float32 t travels once per block
  pack64 (any sensor)                 8 bytes per event (float32 bits of t | x | y | polarity)

Nothing in submit() waits for the device: the event counts are host integers already (the emulator read its records) and
are exchanged over a host-side gloo group; the side stream waits only for the event that marks the run's completion, not
for whatever the caller has enqueued since (the next step); the pack runs on the side stream and the main stream is only
made to wait for the PACK before the event buffer is reused.

Two exchange algorithms: "allgather" (ncclAllGather on payloads padded to the largest rank: a ring, bounded by one xGMI
link per direction) and "p2p" (one grouped ncclSend/ncclRecv per peer, exact sizes: each peer's stream on its own link).
"""
import ctypes as C

import torch
import torch.distributed as dist


def clips_of_rank(n_clips, world, rank):
    """clip c runs on rank c mod world (round-robin; clips are independent)."""
    return [c for c in range(n_clips) if c % world == rank]


def _stream_ptr(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_events64(ev, out=None):
    """[n,4] float32 rows (t, x, y, p) -> [n] int64 words, lossless (include/v2e_amd.h v2e_events_pack64)."""
    n = int(ev.shape[0])
    if out is None:
        out = torch.empty((n,), dtype=torch.int64, device=ev.device)
    if n == 0:
        return out
    if ev.is_cuda:
        from . import _capi
        _capi.check(_capi.lib().v2e_events_pack64(C.c_void_p(ev.data_ptr()), C.c_void_p(out.data_ptr()), n, _stream_ptr(ev.device)),
                    "v2e_events_pack64")
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
        w = words.contiguous()
        _capi.check(_capi.lib().v2e_events_unpack64(C.c_void_p(w.data_ptr()), C.c_void_p(ev.data_ptr()), n, _stream_ptr(words.device)),
                    "v2e_events_unpack64")
    else:
        w = words.contiguous()
        ev[:, 0] = (w >> 32).to(torch.int32).view(torch.float32)  # arithmetic shift keeps the sign bit of t
        ev[:, 1] = ((w >> 18) & 0x3FFF).to(torch.float32)
        ev[:, 2] = ((w >> 4) & 0x3FFF).to(torch.float32)
        ev[:, 3] = (w & 1).to(torch.float32) * 2 - 1
    return ev


def pack_events32(ev, payload=None, runs=None, cap_runs=None, scratch=None):
    """[n,4] float32 rows -> (payload int32 [n], runs int64 [1 + R]) in the 4-byte wire format (include/v2e_amd.h
    v2e_events_pack32): runs[0] = R (low 32 bits; bits 62 / 63 = the overflow flags, so that whoever receives the table sees
    them), runs[1 + r] = float32 bits of t << 32 | index of the block's first event.  On the
    device the outputs may be preallocated (runs with room for cap_runs blocks); returns (payload, runs, flags) where
    flags is a device int32 [>=1] whose word 0 is non-zero if a coordinate did not fit or the run table was too small."""
    n = int(ev.shape[0])
    dev = ev.device
    if ev.is_cuda:
        from . import _capi
        lib = _capi.lib()
        if cap_runs is None:
            cap_runs = n
        if payload is None:
            payload = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        if runs is None:
            runs = torch.empty((cap_runs + 1,), dtype=torch.int64, device=dev)
        words = int(lib.v2e_events_pack32_scratch_words(n))
        if scratch is None or scratch.numel() < words:
            scratch = torch.empty((words,), dtype=torch.int32, device=dev)
        _capi.check(lib.v2e_events_pack32(C.c_void_p(ev.data_ptr()) if n else None, n, C.c_void_p(payload.data_ptr()),
                                          C.c_void_p(runs.data_ptr()), int(min(cap_runs, runs.numel() - 1)),
                                          C.c_void_p(scratch.data_ptr()), _stream_ptr(dev)), "v2e_events_pack32")
        return payload, runs, scratch
    e = ev.contiguous()
    tb = e[:, 0].contiguous().view(torch.int32)
    x, y = e[:, 1].to(torch.int64), e[:, 2].to(torch.int64)
    bad = bool(((x >= 2048) | (y >= 1024)).any()) if n else False
    pl = ((x & 0x7FF) | ((y & 0x3FF) << 11) | ((e[:, 3] > 0).to(torch.int64) << 21)).to(torch.int32)
    if n:
        start = torch.ones(n, dtype=torch.bool)
        start[1:] = tb[1:] != tb[:-1]
        idx = torch.nonzero(start).flatten()
        tbl = ((tb[idx].to(torch.int64) & 0xFFFFFFFF) << 32) | idx
    else:
        tbl = torch.zeros((0,), dtype=torch.int64)
    fl = 1 if bad else 0
    if cap_runs is not None and tbl.numel() > int(cap_runs):  # as the device kernels: the table keeps what it holds, flagged
        tbl, fl = tbl[:int(cap_runs)], fl | 2
    head = int(tbl.numel()) | (fl << 62)
    if head >= 1 << 63:
        head -= 1 << 64  # bit 63 in a signed word
    runs_t = torch.cat([torch.tensor([head], dtype=torch.int64), tbl])
    return pl, runs_t, torch.tensor([fl], dtype=torch.int32)


def unpack_events32(payload, runs, n=None):
    """Inverse of pack_events32: float32 rows, bit for bit what was packed."""
    n = int(payload.shape[0]) if n is None else int(n)
    ev = torch.empty((n, 4), dtype=torch.float32, device=payload.device)
    if n == 0:
        return ev
    if payload.is_cuda:
        from . import _capi
        _capi.check(_capi.lib().v2e_events_unpack32(C.c_void_p(payload.data_ptr()), n, C.c_void_p(runs.data_ptr()),
                                                    C.c_void_p(ev.data_ptr()), _stream_ptr(payload.device)), "v2e_events_unpack32")
        return ev
    R = int(runs[0]) & 0xFFFFFFFF
    tbl = runs[1:1 + R]
    starts = tbl & 0xFFFFFFFF
    k = torch.bucketize(torch.arange(n, dtype=torch.int64), starts, right=True) - 1
    w = payload[:n].to(torch.int64)
    ev[:, 0] = (tbl[k] >> 32).to(torch.int32).view(torch.float32)
    ev[:, 1] = (w & 0x7FF).to(torch.float32)
    ev[:, 2] = ((w >> 11) & 0x3FF).to(torch.float32)
    ev[:, 3] = ((w >> 21) & 1).to(torch.float32) * 2 - 1
    return ev


class EventStreamGatherer:
    """all-gather(v) of per-rank event lists [n_r,4] float32 -> every rank gets all of them.

    submit(ev, n, ready_event=None, run_bound=None) packs the first n rows of `ev` (device tensor) and enqueues the
    exchange on a side stream; result() returns the per-rank [n_r,4] float32 lists (unpacked on demand) of the most recent
    submit, in rank order.  `ready_event`: a torch.cuda.Event behind which the rows are final (the emulator run's
    completion); without it the side stream waits for the whole main stream.  `run_bound`: an upper bound of the number of
    distinct-time-stamp blocks of the stream (sum over frames of max(iterations, 1)); with it, and a sensor that fits
    2048 x 1024, the stream travels in the 4-byte format, otherwise in the 8-byte one.
    On CPU tensors (gloo, tests) everything runs inline.
    """

    def __init__(self, device, world, group=None, wire="auto", algo="allgather", sensor=None):
        self.device = torch.device(device)
        self.world = world
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        if wire not in ("auto", "pack32", "pack64"):
            raise ValueError("wire must be auto, pack32 or pack64")
        if algo not in ("allgather", "p2p"):
            raise ValueError("algo must be allgather or p2p")
        # pack32 holds x < 2048, y < 1024: it is chosen only for a sensor that is KNOWN to fit (`sensor` = (H, W)); without one
        # "auto" sends the 8-byte format, and an explicit "pack32" is taken at the caller's word (a coordinate that does not
        # fit is then flagged inside the run table and result() raises on EVERY rank)
        fits32 = sensor is not None and int(sensor[0]) <= 1024 and int(sensor[1]) <= 2048
        if wire == "pack32" and sensor is not None and not fits32:
            raise ValueError("pack32 holds x < 2048, y < 1024")
        self.wire = "pack32" if (wire == "pack32" or (wire == "auto" and fits32)) else "pack64"
        self.algo = algo
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # event counts / run bounds are host integers: they travel over a host-side group of the SAME ranks, the device is
        # never asked
        self.host_group = group
        self._global = None  # group-local rank -> global rank (P2POp peers are global ranks)
        if dist.is_initialized() and group is not None:
            self._global = list(dist.get_process_group_ranks(group))
        if self.cuda and dist.is_initialized() and dist.get_backend(group) != "gloo":
            self.host_group = dist.new_group(ranks=self._global, backend="gloo")  # (every rank of the job must construct it)
        self.slots = [dict(), dict()]
        self.flip = 0
        self.last = None
        self.bytes_gathered = 0

    # ------------------------------------------------------------------ helpers
    def _buf(self, slot, key, numel, dtype):
        b = self.slots[slot].get(key)
        if b is None or b.numel() < numel or b.dtype != dtype:
            b = torch.empty((max(int(numel), 1),), dtype=dtype, device=self.device)
            if self.cuda:
                b.record_stream(self.side)
            self.slots[slot][key] = b
        return b

    def _host_counts(self, n, rb):
        """[world][2] (events, run bound; -1: this rank has none) of all ranks, exchanged on the host."""
        mine = torch.tensor([int(n), int(rb)], dtype=torch.int64)
        if self.world == 1 or not dist.is_initialized():
            return mine.view(1, 2)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.host_group)
        return torch.stack(parts)

    def _exchange(self, send, recv_len, out, elem_max):
        """send[: recv_len[rank]] of every rank into out[r * elem_max : ...] on the current stream."""
        if self.algo == "allgather" or self.world == 1:
            if self.cuda:
                dist.all_gather_into_tensor(out[:self.world * elem_max], send[:elem_max], group=self.group)
            else:
                parts = [torch.empty_like(send[:elem_max]) for _ in range(self.world)]
                dist.all_gather(parts, send[:elem_max].contiguous(), group=self.group)
                out[:self.world * elem_max] = torch.cat(parts)
            return self.world * elem_max
        ops = []
        me = self.rank
        out[me * elem_max: me * elem_max + recv_len[me]] = send[:recv_len[me]]
        for r in range(self.world):
            if r == me:
                continue
            peer = r if self._global is None else self._global[r]
            if recv_len[me] > 0:
                ops.append(dist.P2POp(dist.isend, send[:recv_len[me]], peer, group=self.group))
            if recv_len[r] > 0:
                ops.append(dist.P2POp(dist.irecv, out[r * elem_max: r * elem_max + recv_len[r]], peer, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):  # one ncclGroupStart / End: every peer's stream on its own link
                w.wait()
        return sum(recv_len)

    # ------------------------------------------------------------------ API
    def submit(self, ev, n, ready_event=None, run_bound=None):
        slot = self.flip
        self.flip ^= 1
        n = int(n)
        # the wire format is agreed on by ALL ranks (they must issue the same collectives): the 4-byte format only if this
        # gatherer is set up for it and EVERY rank supplied a run bound for this step
        rb = int(run_bound) if (self.wire == "pack32" and run_bound is not None) else -1
        hc = self._host_counts(n, rb)
        counts = [int(v) for v in hc[:, 0].tolist()]
        nmax = max(max(counts), 1)
        wire = "pack32" if int(hc[:, 1].min()) >= 0 else "pack64"
        rmax = max(int(hc[:, 1].max()), 1)
        st = self.slots[slot]
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            if ready_event is not None:
                self.side.wait_event(ready_event)
            else:
                self.side.wait_stream(main)
            if st.get("done") is not None:
                self.side.wait_event(st["done"])  # the previous exchange out of this staging slot
            with torch.cuda.stream(self.side):
                moved = self._pack_and_exchange(slot, ev, n, wire, counts, nmax, rmax)
                packed = st["packed"]
                st["done"] = torch.cuda.Event()
                st["done"].record(self.side)
            main.wait_event(packed)  # the event buffer may be reused once the pack (not the exchange) has read it
        else:
            moved = self._pack_and_exchange(slot, ev, n, wire, counts, nmax, rmax)
        self.bytes_gathered += moved
        self.last = dict(slot=slot, wire=wire, counts=counts, nmax=nmax, rmax=rmax)

    def _pack_and_exchange(self, slot, ev, n, wire, counts, nmax, rmax):
        st = self.slots[slot]
        if wire == "pack64":
            send = self._buf(slot, "w64", nmax, torch.int64)
            pack_events64(ev[:n], send)
            if self.cuda:
                st["packed"] = torch.cuda.Event()
                st["packed"].record(self.side)
            out = self._buf(slot, "o64", self.world * nmax, torch.int64)
            return 8 * self._exchange(send, counts, out, nmax)
        # pack32: payload (4 B per event) + run table (8 B per block of one time stamp; entry 0 = number of blocks)
        if self.cuda:
            payload = self._buf(slot, "p32", nmax, torch.int32)
            runs = self._buf(slot, "r32", rmax + 1, torch.int64)
            scratch = self._buf(slot, "s32", 2 + (n + 1023) // 1024, torch.int32)
            pack_events32(ev[:n], payload, runs, rmax, scratch)
            st["flags"] = scratch
            st["packed"] = torch.cuda.Event()
            st["packed"].record(self.side)
        else:
            pl, rt, fl = pack_events32(ev[:n], cap_runs=rmax)
            payload = self._buf(slot, "p32", nmax, torch.int32)
            runs = self._buf(slot, "r32", rmax + 1, torch.int64)
            payload[:n] = pl
            runs[:rt.numel()] = rt
        outp = self._buf(slot, "op32", self.world * nmax, torch.int32)
        outr = self._buf(slot, "or32", self.world * (rmax + 1), torch.int64)
        moved = 4 * self._exchange(payload, counts, outp, nmax)
        moved += 8 * self._exchange(runs, [rmax + 1] * self.world, outr, rmax + 1)
        return moved

    def wait(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)

    def result(self):
        """Per-rank event lists [n_r,4] float32 of the last submit, in rank order."""
        self.wait()
        L = self.last
        st = self.slots[L["slot"]]
        c, nmax, rmax = L["counts"], L["nmax"], L["rmax"]
        if L["wire"] == "pack64":
            out = st["o64"]
            return [unpack_events64(out[r * nmax: r * nmax + c[r]]) for r in range(self.world)]
        outp, outr = st["op32"], st["or32"]
        # every rank received every rank's table, flags included: all ranks raise together (a rank that raised alone would
        # leave its peers with events whose time stamps were searched in a truncated table)
        heads = outr[: self.world * (rmax + 1): rmax + 1].cpu()
        bad = [(r, (int(h) >> 62) & 3) for r, h in enumerate(heads.tolist()) if (int(h) >> 62) & 3]
        if bad:
            raise ValueError("pack32 stream refused on every rank: " + "; ".join(
                "rank %d: %s" % (r, " and ".join(m for b, m in ((1, "a coordinate beyond 2048 x 1024"), (2, "run_bound too small")) if f & b))
                for r, f in bad) + " -- construct the gatherer with sensor=(H, W) / wire='pack64', or pass a correct run_bound")
        return [unpack_events32(outp[r * nmax: r * nmax + max(c[r], 1)], outr[r * (rmax + 1): (r + 1) * (rmax + 1)], c[r])
                for r in range(self.world)]
