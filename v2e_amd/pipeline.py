"""Device-resident video -> events pipeline (BASELINE configs[2]).

The reference's v2e.py runs its three stages through the filesystem: source frames as .npy,
`SuperSloMo.interpolate` writes 8-bit PNGs (slomo.py:436-444), v2e.py reads them back
(v2e.py:832) and calls `EventEmulator.generate_events` per frame.  `VideoToEvents` keeps every
frame in HBM and produces the same uint8 frames for the emulator:

  source uint8 [N,H,W] --Pillow-exact LANCZOS--> [N,h,w] --/255 - mean--> SloMo (HIP, all U time
  points per batch) --(+mean)*255, byte truncation--> uint8 --Pillow-exact BILINEAR--> [P*U,H,W]
  --> emulator (HIP, Philox, one device-resident run)

One clip over several GPUs: `VideoToEvents.run(..., group=)` shards the SuperSloMo stage by source pairs over the ranks of a process
group and funnels the uint8 frames, in order, to the rank that owns the emulator state.

Frame order and times follow slomo.py:391-400, 441 and v2e.py:786-797: output frame b*U + k of a batch has
interpTime pair + k/U (in source frame intervals), and v2e.py stretches the interpTimes so that they SPAN the processed
duration of the source clip: t = interpTime * (N_src - 1) * interval / (max(interpTimes) - min(interpTimes)).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check
from .resample import coeffs_8bpc


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceResampler:
    """Pillow-exact uint8 resize [n,ih,iw] -> [n,oh,ow] on device (tables cached on device)."""

    def __init__(self, in_hw, out_hw, filt, device):
        self.ih, self.iw = in_hw
        self.oh, self.ow = out_hw
        self.device = torch.device(device)
        self.lib = _capi.lib()
        self.hb = self.hk = self.vb = self.vk = None
        self.hks = self.vks = 0
        if self.ow != self.iw:
            b, k = coeffs_8bpc(self.iw, self.ow, filt)
            self.hb, self.hk, self.hks = torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1]
        if self.oh != self.ih:
            b, k = coeffs_8bpc(self.ih, self.oh, filt)
            self.vb, self.vk, self.vks = torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1]

    def __call__(self, x):
        assert x.dtype == torch.uint8 and x.is_contiguous() and tuple(x.shape[-2:]) == (self.ih, self.iw)
        n = x.numel() // (self.ih * self.iw)
        out = torch.empty((n, self.oh, self.ow), dtype=torch.uint8, device=self.device)
        tmp = torch.empty((n, self.ih, self.ow), dtype=torch.uint8, device=self.device) if (self.ow != self.iw and self.oh != self.ih) else None
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.v2e_resample_u8(_ptr(x), _ptr(tmp), _ptr(out), n, self.ih, self.iw, self.oh, self.ow, _ptr(self.hb),
                                       _ptr(self.hk), self.hks, _ptr(self.vb), _ptr(self.vk), self.vks, s), "v2e_resample_u8")
        return out


def interp_frame_times(n_src, U, src_frame_interval_s, batch_size=None):
    """Times of the (n_src - 1) * U interpolated frames as v2e.py computes them, operation for operation: interpTimes are built
    per batch of `batch_size` pairs as inputFrameCounter + k * (1 / U) (slomo.py:391-400; None: one batch), then normalised
    to the processed duration of the source clip (v2e.py:786-797: f = srcVideoRealProcessedDuration / (max(interpTimes) -
    min(interpTimes)); interpTimes = f * interpTimes)."""
    npairs = n_src - 1
    bs = npairs if not batch_size else int(batch_size)
    parts = []
    for b0 in range(0, npairs, bs):
        nb = min(bs, npairs - b0)
        parts.append(b0 + np.array(range(U * nb)) * (1 / U))
    interp = np.concatenate(parts)
    duration = npairs * float(src_frame_interval_s)
    f = duration / (np.max(interp) - np.min(interp))
    return f * interp


class VideoToEvents:
    """SuperSloMo upsampling + DVS emulation of one clip without leaving HBM."""

    def __init__(self, slomo_engine, emulator, upsampling_factor, batch_size=8, mean=0.428):
        self.eng = slomo_engine
        self.emu = emulator
        self.U = int(upsampling_factor)
        self.batch_size = int(batch_size)
        self.mean = float(mean)
        self.lib = _capi.lib()
        self._rs_in = self._rs_out = None

    def upsample(self, frames_u8):
        """frames_u8: device uint8 [N,H,W] -> device uint8 [(N-1)*U, H, W], the frames the reference's
        interpolate() would have written as PNGs (in order)."""
        dev = self.eng.device
        N, H, W = frames_u8.shape
        dim = (int(W / 32) * 32, int(H / 32) * 32)  # (w, h), dataloader.py:122-123
        w, h = dim
        if self._rs_in is None or (self._rs_in.ih, self._rs_in.iw) != (H, W):
            self._rs_in = DeviceResampler((H, W), (h, w), "lanczos", dev)
            self._rs_out = DeviceResampler((h, w), (H, W), "bilinear", dev)
        s = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        small = self._rs_in(frames_u8.contiguous())
        x = torch.empty((N, 1, h, w), dtype=torch.float32, device=dev)
        check(self.lib.v2e_u8_to_f32_norm(_ptr(small), _ptr(x), small.numel(), self.mean, s()), "v2e_u8_to_f32_norm")
        U = self.U
        ts = [(k + 0.5) / U for k in range(U)]
        out = torch.empty(((N - 1) * U, H, W), dtype=torch.uint8, device=dev)
        # the flow UNet sees up to 64 pairs at a time (its small levels fill the GPU only with many samples; a pair's
        # flow does not depend on what else is in the batch), the interpolation UNet U x batch_size samples
        fc = self.batch_size * max(1, 64 // self.batch_size)  # pairs per flow-UNet call: whole interpolation batches
        flow, f0, f1 = None, 0, 0
        for b0 in range(0, N - 1, self.batch_size):
            b1 = min(b0 + self.batch_size, N - 1)
            B = b1 - b0
            if b0 >= f1:
                f0, f1 = b0, min(b0 + fc, N - 1)
                flow = self.eng.flow(x[f0:f1], x[f0 + 1:f1 + 1])
            Ft = self.eng.interpolate(x[b0:b1], x[b0 + 1:b1 + 1], ts, flow=flow[b0 - f0:b1 - f0].contiguous())  # [U,B,1,h,w]
            q = torch.empty((B * U, h, w), dtype=torch.uint8, device=dev)
            check(self.lib.v2e_f32_to_u8_trunc(_ptr(Ft), _ptr(q), U, B, h * w, self.mean, 1, s()), "v2e_f32_to_u8_trunc")
            out[b0 * U:b1 * U] = self._rs_out(q)
        return out

    def upsample_sharded(self, frames_u8, group=None, owner=None):
        """SuperSloMo of ONE clip over the ranks of `group` (BASELINE north_star: "frame batches shard across the
        GPUs"): the P = N - 1 source pairs are independent (slomo.py:330-466: every batch is computed from its own frames), so rank r
        interpolates pairs [r P / G, (r + 1) P / G) -- it needs source frames r P / G .. (r + 1) P / G only -- and the uint8 output
        frames are gathered IN ORDER: to rank `owner` point to point with exact sizes (90 KB per frame at 346x260, 27 MB for a second
        of video at 10x), or, owner = None, to every rank by one all-gather of padded blocks (gather_frames_in_order).
        Returns uint8 [(N - 1) U, H, W] on the assembling rank(s), None elsewhere.  Every rank must hold the same frames_u8 (or at
        least its own slice of it: only frames lo .. hi are read)."""
        import torch.distributed as dist
        if group is None and not dist.is_initialized():
            return self.upsample(frames_u8)
        G, r = dist.get_world_size(group), dist.get_rank(group)
        N = int(frames_u8.shape[0])
        lo, hi = pair_shard(N - 1, G, r)
        mine = self.upsample(frames_u8[lo:hi + 1]) if hi > lo else frames_u8.new_empty((0,) + tuple(frames_u8.shape[1:]))
        return gather_frames_in_order(mine, N - 1, self.U, group, owner)

    def run(self, frames_u8, src_frame_interval_s, return_device=False, group=None, owner=0, chunk_pairs=None):
        """Full pipeline; returns (events, counts_per_interpolated_frame, n_interpolated_frames).
        group: a torch.distributed process group -- the SuperSloMo stage is sharded over its ranks by source pairs
        (`upsample_sharded`), the frames are funnelled in order to rank `owner` (rank within the group), which holds the emulator
        state and runs the DVS model; the other ranks return (None, None, n).
        chunk_pairs (with a group): the clip goes through in CHUNKS of that many source pairs -- every chunk sharded over the ranks
        and sent to the owner (point to point, exact sizes), the owner's DVS stage taking chunk c (enqueued, not waited for) while
        the ranks, the owner included, interpolate chunk c + 1: for a long clip the owner no longer idles through the whole
        interpolation and holds one chunk of frames, not the clip.  Same frames, order, times and events as one rank."""
        if group is not None and chunk_pairs:
            return self._run_chunked(frames_u8, src_frame_interval_s, return_device, group, owner, int(chunk_pairs))
        if group is not None:
            import torch.distributed as dist
            up = self.upsample_sharded(frames_u8, group, owner)
            n = (int(frames_u8.shape[0]) - 1) * self.U
            if dist.get_rank(group) != owner:
                return None, None, n
        else:
            up = self.upsample(frames_u8)
            n = up.shape[0]
        times = interp_frame_times(int(frames_u8.shape[0]), self.U, float(src_frame_interval_s), self.batch_size)
        ev, counts = self.emu.generate_events_batch(up, times, return_device=return_device)
        return ev, counts, n

    def _run_chunked(self, frames_u8, src_frame_interval_s, return_device, group, owner, chunk_pairs):
        import torch.distributed as dist
        G, r = dist.get_world_size(group), dist.get_rank(group)
        N = int(frames_u8.shape[0])
        P, U = N - 1, self.U
        times = interp_frame_times(N, U, float(src_frame_interval_s), self.batch_size)
        shape = tuple(frames_u8.shape[1:])
        evs, counts, pending = [], [], None   # pending: (receive requests, the chunk's frames, its times) -- the chunk the owner takes next
        asynchronous = hasattr(self.emu, "generate_events_batch_async")
        runs = []

        def dvs(chunk, t):
            if asynchronous:  # enqueue; the result is read at the end (the emulator alternates two buffer sets: at most two in flight)
                runs.append(self.emu.generate_events_batch_async(chunk, t, return_device=return_device))
                while len(runs) > 1:
                    e, c = runs.pop(0).result()
                    # (device rows are a view of one of the emulator's two alternating buffers: the chunk after next overwrites it)
                    evs.append(e.clone() if return_device and e is not None else e)
                    counts.append(c)
            else:
                e, c = self.emu.generate_events_batch(chunk, t, return_device=return_device)
                evs.append(e)
                counts.append(c)

        for p0 in range(0, P, chunk_pairs):
            p1 = min(p0 + chunk_pairs, P)
            lo, hi = pair_shard(p1 - p0, G, r)
            # (the owner posts its receives BEFORE its own share of the interpolation: the other ranks' sends then complete as they finish)
            reqs, chunk = send_frames_to_owner_begin(p1 - p0, U, shape, frames_u8, group, owner)
            mine = self.upsample(frames_u8[p0 + lo:p0 + hi + 1]) if hi > lo else frames_u8.new_empty((0,) + shape)
            if r == owner and pending is not None:  # the DVS stage of the chunk BEFORE, while this chunk's frames arrive
                for q in pending[0]:
                    q.wait()
                dvs(pending[1], pending[2])
            reqs = send_frames_to_owner_finish(reqs, chunk, mine, p1 - p0, U, group, owner)
            if r == owner:
                pending = (reqs, chunk, times[p0 * U:p1 * U])
            else:
                for q in reqs:
                    q.wait()
        n = P * U
        if r != owner:
            return None, None, n
        for q in pending[0]:
            q.wait()
        dvs(pending[1], pending[2])
        for run in runs:
            e, c = run.result()
            evs.append(e)
            counts.append(c)
        import numpy as np
        evs = [e for e in evs if e is not None and len(e)]
        if not evs:
            ev = None
        elif torch.is_tensor(evs[0]):
            ev = torch.cat(evs)
        else:
            ev = np.concatenate(evs)
        return ev, np.concatenate([np.asarray(c) for c in counts]), n


def pair_shard(n_pairs, world, rank):
    """Source pairs [lo, hi) of rank `rank`: contiguous blocks r P / G .. (r + 1) P / G; empty when P < G leaves none."""
    return rank * n_pairs // world, (rank + 1) * n_pairs // world


def gather_frames_in_order(mine, n_pairs, U, group=None, owner=None):
    """The ranks' interpolated-frame blocks (uint8 [pairs_r U, H, W], pairs_r from pair_shard) in the clip's frame order.
    owner = a rank of the group: POINT TO POINT to that rank, exact sizes, received straight into the clip's tensor (round 6; an
    all-gather hands every rank the whole clip -- G - 1 times the receive volume anybody needs -- and wants the blocks padded to the
    largest); the other ranks return None.  owner = None: every rank assembles the clip (one all-gather of padded blocks).  Block
    sizes follow from (n_pairs, world) alone, so no size exchange is needed."""
    import torch.distributed as dist
    G, r = dist.get_world_size(group), dist.get_rank(group)
    cnt = [(pair_shard(n_pairs, G, q)[1] - pair_shard(n_pairs, G, q)[0]) * U for q in range(G)]
    assert int(mine.shape[0]) == cnt[r], "this rank interpolated %d frames, its shard has %d" % (int(mine.shape[0]), cnt[r])
    if owner is not None:
        reqs, clip = send_frames_to_owner_begin(n_pairs, U, tuple(mine.shape[1:]), mine, group, owner)
        for q in send_frames_to_owner_finish(reqs, clip, mine, n_pairs, U, group, owner):
            q.wait()
        return clip if r == owner else None
    m = max(cnt)
    pad = mine if cnt[r] == m else torch.cat((mine, mine.new_zeros((m - cnt[r],) + tuple(mine.shape[1:]))))
    parts = [torch.empty_like(pad) for _ in range(G)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    return torch.cat([parts[q][:cnt[q]] for q in range(G)])


def _global_rank(group, q):
    import torch.distributed as dist
    return q if group is None or group is dist.group.WORLD else dist.get_global_rank(group, q)


def send_frames_to_owner_begin(n_pairs, U, frame_shape, like, group, owner):
    """Owner: the tensor of the n_pairs * U frames and one posted receive per other rank's block, straight into its slice (posted
    before the owner's own interpolation, so that a sender never waits for the receiver).  Other ranks: ([], None)."""
    import torch.distributed as dist
    G, r = dist.get_world_size(group), dist.get_rank(group)
    if r != owner:
        return [], None
    clip = like.new_empty((n_pairs * U,) + tuple(frame_shape))
    reqs = []
    for q in range(G):
        lo, hi = pair_shard(n_pairs, G, q)
        if q != owner and hi > lo:
            reqs.append(dist.irecv(clip[lo * U:hi * U], src=_global_rank(group, q), group=group))
    return reqs, clip


def send_frames_to_owner_finish(reqs, clip, mine, n_pairs, U, group, owner):
    """Owner: its own block copied into place, the receive requests handed back.  Other ranks: their block sent (a request to wait for)."""
    import torch.distributed as dist
    G, r = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = pair_shard(n_pairs, G, r)
    if r == owner:
        if hi > lo:
            clip[lo * U:hi * U] = mine
        return reqs
    return [dist.isend(mine.contiguous(), dst=_global_rank(group, owner), group=group)] if hi > lo else []
