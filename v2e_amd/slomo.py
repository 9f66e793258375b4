"""Drop-in `SuperSloMo` running the interpolation network on MI355X HIP kernels.

Mirrors v2ecore/slomo.py: constructor signature (:44-54), `interpolate(source_frame_path,
output_folder, frame_size) -> (interpTimes, avgUpsampling)` writing `<idx>.png` (:231-495),
`get_interpolated_timestamps` (:540-564), `cleanup`.  The two UNets, the backWarp
grid-sample, the flow blends and the fusion (slomo.py:338-345, 404-433; model.py:10-300) run
in libv2e_amd.so (f32 MFMA implicit-GEMM convolutions); PyTorch only owns the buffers.

Differences from the reference, all at the edges of the hot path:
  * all `upsampling_factor` time points of a batch go through the interpolation UNet as one
    batch (n_t * B samples) instead of a Python loop -- same arithmetic per sample;
  * the AVI / preview writers (slomo.py:288-303, 447-490) are host work outside the hot path: `video_path`, `vid_orig`,
    `vid_slomo` and `preview` are accepted as the reference's CLI passes them (v2e.py:471-478); the two AVIs are written on the
    host through cv2 when cv2 is importable and skipped with ONE warning when it is not -- the constructor never raises for
    them, and the interpolation itself is unaffected;
  * host-side pre/post-processing (PIL LANCZOS / BILINEAR resize, PNG files) is kept as is.
"""
import atexit
import weakref
import glob
import logging
import math
import os

import numpy as np
import torch

from . import _capi
from ._capi import ConvDesc, check

logger = logging.getLogger(__name__)

# forward order of the 23 convolutions of model.UNet (model.py:184-196, 198-226)
UNET_LAYERS = (["conv1", "conv2"] + ["down%d.conv%d" % (d, c) for d in range(1, 6) for c in (1, 2)] +
               ["up%d.conv%d" % (u, c) for u in range(1, 6) for c in (1, 2)] + ["conv3"])


OUTPUT_VIDEO_CODEC_FOURCC = 'XVID'  # v2ecore/v2e_utils.py:22, the codec of video_writer


def _all_images(data_path):
    """slomo.py:500-521: the PNGs of a folder in numerical order."""
    images = glob.glob(os.path.join(data_path, '*.png'))
    if len(images) == 0:
        raise ValueError("Input folder is empty or images are not in 'png' format.")
    return sorted(images, key=lambda line: int(line.split(os.sep)[-1].split('.')[0]))


def _ptr(t):
    import ctypes as C
    return None if t is None else C.c_void_p(t.data_ptr())


# V2E_AMD_SLOMO_LOOKAHEAD=0: every UNet pass on the caller's stream, one after the other (SloMoEngine.flow_ahead is then never used by
# interpolate): the serial order of rounds 1-3, for whoever does not want two UNet passes in flight
_LOOKAHEAD = os.environ.get("V2E_AMD_SLOMO_LOOKAHEAD", "1") != "0"
# Two UNet passes in flight went wrong once (round 4) and the cause is known (round 5, profiles/r05_concurrency_rootcause.txt): on
# gfx950 a packed-float32 instruction whose op_sel routes src1's high half to the low lane reads 0 there while another kernel issues
# independent 16-K bf16 / f16 MFMAs.  The library is built without any packed-float32 instruction (Makefile, tests/test_device_isa.py),
# so the look-ahead is safe as built; the canary below makes a differently built library (EXTRA=-fslp-vectorize, another compiler)
# fall back to one stream instead of corrupting frames silently.  Result per device, decided at the first look-ahead of the process.
_CANARY = {}


class HipUNet:
    """One UNet's weights, repacked [Cin][k][k][Cout] in HBM, and its forward pass."""

    def __init__(self, state_dict, cin, cout, device, conv_math="bf16x3"):
        import ctypes as C
        if conv_math not in ("auto", "bf16x3", "f32", "fp16x2"):
            raise ValueError("conv_math must be 'auto' (two float16 pieces and three products per multiply -- half the "
                             "matrix-core work, products good to ~2^-21; operands are staged times powers of two that keep them in "
                             "float16's normal range, and a forward pass that meets an inf / NaN activation is redone with the exact "
                             "split), 'bf16x3' (f32 operands split exactly into three "
                             "bf16 pieces, six piece products on the bf16 matrix cores, f32 accumulation), 'fp16x2' (the two-piece "
                             "math without the guard) or 'f32' (f32 matrix-core instructions)")
        self.conv_math = conv_math
        self.lib = _capi.lib()
        self.device = torch.device(device)
        self.cin, self.cout = cin, cout
        self._keep = []
        self.fallbacks = 0  # conv_math 'auto': forward passes redone with the exact split
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        want_h2 = conv_math in ("auto", "fp16x2")
        want_s3 = conv_math in ("auto", "bf16x3")
        descs = (ConvDesc * 23)()        # what forward() runs first
        descs_exact = (ConvDesc * 23)() if conv_math == "auto" else None
        fast_ok = True
        for i, name in enumerate(UNET_LAYERS):
            w = state_dict[name + ".weight"].detach().to(self.device, torch.float32).contiguous()
            b = state_dict[name + ".bias"].detach().to(self.device, torch.float32).contiguous()
            co, ci, kh, kw = w.shape
            assert kh == kw
            wp = torch.empty((ci, kh, kw, co), dtype=torch.float32, device=self.device)
            check(self.lib.v2e_pack_conv_weight(_ptr(w), _ptr(wp), co, ci, kh, stream), "v2e_pack_conv_weight")
            self._keep += [wp, b]
            for d in (descs, descs_exact):
                if d is not None:
                    d[i].weight, d[i].bias = wp.data_ptr(), b.data_ptr()
                    d[i].weight_s3, d[i].split_kind = None, 0
                    d[i].cin, d[i].cout, d[i].ksize = ci, co, kh
            splittable = co % 32 == 0 and (ci % 16 == 0 or (kh == 7 and ci >= 8))
            if want_h2 and splittable:
                # an exact power of two that puts the layer's largest weight in [2^12, 2^13): no piece of a weight that matters
                # is a float16 subnormal, nothing overflows (65504); the kernel divides it out again
                wmax = float(w.abs().max())
                if not wmax < 65504.0:
                    if conv_math == "fp16x2":
                        raise ValueError("conv_math='fp16x2': layer %s has a weight of magnitude %g, beyond float16" % (name, wmax))
                    fast_ok = False
                else:
                    sl2 = 0 if not wmax > 0.0 else max(0, min(40, 12 - math.frexp(wmax)[1] + 1))
                    w2 = torch.empty((ci + 15) // 16 * 16 * kh * kw * co * 4, dtype=torch.uint8, device=self.device)
                    check(self.lib.v2e_pack_conv_weight_h2(_ptr(w), _ptr(w2), co, ci, kh, sl2, stream), "v2e_pack_conv_weight_h2")
                    self._keep.append(w2)
                    descs[i].weight_s3 = w2.data_ptr()
                    descs[i].split_kind = 2 | (sl2 << 8)
            if want_s3 and splittable:
                w3 = torch.empty((ci + 15) // 16 * 16 * kh * kw * co * 6, dtype=torch.uint8, device=self.device)
                check(self.lib.v2e_pack_conv_weight_s3(_ptr(w), _ptr(w3), co, ci, kh, stream), "v2e_pack_conv_weight_s3")
                self._keep.append(w3)
                d = descs_exact if conv_math == "auto" else descs
                d[i].weight_s3, d[i].split_kind = w3.data_ptr(), 0
        torch.cuda.synchronize(self.device)
        assert descs[0].cin == cin and descs[22].cout == cout
        self.descs, self.descs_exact = descs, descs_exact
        if conv_math == "auto" and not fast_ok:  # a weight beyond float16: the exact split throughout
            self.descs, self.descs_exact = descs_exact, None
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device) if self.descs_exact is not None else None
        self._ws = None
        self._pending = None

    def forward(self, x, out=None, defer_check=False):
        """One forward pass on the current stream.  defer_check (conv_math 'auto'): do not read the range flag back now -- the
        caller enqueues whatever follows and calls `redo_if_flagged()` afterwards (one host synchronisation per batch instead of
        one per network; the flow UNet of a later batch can run on another stream meanwhile)."""
        import ctypes as C
        n, c, h, w = x.shape
        assert c == self.cin and x.dtype == torch.float32 and x.is_contiguous()
        need = int(self.lib.v2e_unet_workspace_bytes(n, h, w, c))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((n, self.cout, h, w), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._pending = None
        if self._flag is None:
            check(self.lib.v2e_unet_forward(_ptr(x), c, self.descs, self.cout, _ptr(out), n, h, w, _ptr(self._ws), stream),
                  "v2e_unet_forward")
            return out
        # conv_math 'auto': the two-float16-piece convolutions report an inf / NaN activation (finite ones are scaled into
        # float16's range inside the kernels); such a pass is redone with the exact three-bf16-piece weights (one 4-byte
        # read-back per forward pass)
        self._flag.zero_()
        check(self.lib.v2e_conv_set_range_flag(_ptr(self._flag)), "v2e_conv_set_range_flag")
        try:
            check(self.lib.v2e_unet_forward(_ptr(x), c, self.descs, self.cout, _ptr(out), n, h, w, _ptr(self._ws), stream),
                  "v2e_unet_forward")
        finally:
            self.lib.v2e_conv_set_range_flag(None)
        self._pending = (x, out)
        if not defer_check:
            self.redo_if_flagged()
        return out

    def walk_layers(self, x, inputs_from=None):
        """The UNet layer by layer through v2e_conv2d_lrelu (model.py:198-226: avg_pool2d and the bilinear x2 fused into the consuming
        convolution's loader, the skip concatenation as its second input): [(name, (x0, x1, pre), y)] for the 23 convolutions.
        inputs_from = the list another network's walk returned: every layer then runs on THAT walk's inputs (an isolated layer-by-layer
        comparison of two conv maths; v2e_amd.check_ckpt).  A two-float16-piece layer called this way stages its activations
        unscaled (include/v2e_amd.h): the isolated figure is an upper bound for what the same layer does inside v2e_unet_forward."""
        import ctypes as C
        from .synth import unet_layer_shapes
        names = [q[0] for q in unet_layer_shapes(self.cin, self.cout)]
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        out = []

        def conv(i, x0, x1, pre, h, w):
            if inputs_from is not None:
                x0, x1, pre = inputs_from[i][1]
            d = self.descs[i]
            y = torch.empty((x0.shape[0], d.cout, h, w), dtype=torch.float32, device=self.device)
            check(self.lib.v2e_conv2d_lrelu(_ptr(x0), int(x0.shape[1]), _ptr(x1), 0 if x1 is None else int(x1.shape[1]), pre, C.byref(d),
                                            _ptr(y), int(x0.shape[0]), h, w, stream), "v2e_conv2d_lrelu")
            out.append((names[i], (x0, x1, pre), y))
            return y

        n, c, h, w = x.shape
        cur = conv(1, conv(0, x, None, 0, h, w), None, 0, h, w)
        skips = [cur]
        for d in range(5):
            h, w = h // 2, w // 2
            cur = conv(3 + 2 * d, conv(2 + 2 * d, cur, None, 1, h, w), None, 0, h, w)
            skips.append(cur)
        for u in range(5):
            h, w = h * 2, w * 2
            cur = conv(13 + 2 * u, conv(12 + 2 * u, cur, None, 2, h, w), skips[4 - u], 0, h, w)
        conv(22, cur, None, 0, h, w)
        return out

    def redo_if_flagged(self):
        """Read the range flag of the last forward pass (on the CURRENT stream, which must be ordered behind that pass) and redo
        the pass with the exact split if it was raised.  Returns True when it was redone (what consumed `out` must be redone too)."""
        if self._pending is None:
            return False
        import ctypes as C
        x, out = self._pending
        self._pending = None
        if int(self._flag.item()) == 0:
            return False
        self.fallbacks += 1
        n, c, h, w = x.shape
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.v2e_unet_forward(_ptr(x), c, self.descs_exact, self.cout, _ptr(out), n, h, w, _ptr(self._ws), stream),
              "v2e_unet_forward")
        return True


def time_coefficients(ts):
    """[n_t][6] float32: fCoeff[0..3] (slomo.py:406-407) and wCoeff (:429), in Python doubles first."""
    rows = []
    for t in ts:
        t = float(t)
        temp = -t * (1 - t)
        rows.append([temp, t * t, (1 - t) * (1 - t), temp, 1 - t, t])
    return np.asarray(rows, dtype=np.float64).astype(np.float32)


class SloMoEngine:
    """Flow UNet + interpolation UNet + warps/fusion for batches of frame pairs, on device."""

    def __init__(self, flow_state_dict, interp_state_dict, device="cuda", conv_math=None):
        """conv_math: 'auto' (default) -- every f32 operand split into two float16 pieces, three piece products on the f16 matrix
        cores, f32 accumulation: operands good to 2^-22 and, measured on every fixture, as close to the reference and to float64
        as the other maths (profiles/r03_slomo_precision.txt).  float16's narrow exponent is taken out of the picture on both
        sides: weights are packed times an exact power of two per layer, and activations are staged times the power of two that
        puts the layer's largest input (tracked on the device by the producing convolution's epilogue) in [2^13, 2^14) -- so a
        piece is a float16 subnormal only where the activation is below 2^-27 of the layer's largest one (an absolute operand
        error of 2^-39 of that maximum), whatever the network's scale (tests: inputs x 3e5 and x 2^-17,
        tests/golden/slomo_smallact_64x96.npz).  A forward pass that meets an inf or NaN activation is redone with the exact
        split (HipUNet.forward);
        'bf16x3' -- three bf16 pieces, six products: the exact split (v2e_amd/csrc/slomo_s3.h), 0.7x the frames per second;
        'fp16x2' -- the two-piece math without the redo; 'f32' -- f32 matrix-core instructions.  The environment variable
        V2E_AMD_CONV_MATH sets the default."""
        if conv_math is None:
            conv_math = os.environ.get("V2E_AMD_CONV_MATH", "auto")
        self.conv_math = conv_math
        if not torch.cuda.is_available():
            raise _capi.V2EAmdError("v2e_amd.SloMoEngine needs a ROCm GPU; there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _capi.lib()
        self.flow_net = HipUNet(flow_state_dict, 2, 4, self.device, conv_math)
        self.interp_net = HipUNet(interp_state_dict, 12, 5, self.device, conv_math)
        self._state_dicts = (flow_state_dict, interp_state_dict)  # for self_check(): the float32-MFMA engine is built from them on demand
        self._x2 = None
        self._speed_bits = None
        self._fstream = None
        self._ahead = None  # (I0, I1, x, flow) of a flow pass started ahead on the side stream

    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def flow(self, I0, I1):
        """flowOut = flow_estimator(cat(I0, I1)) (slomo.py:343); I0, I1: [B,1,H,W] float32.  If `flow_ahead` was started for
        exactly these two tensors, its result is taken (the current stream is ordered behind the side stream first)."""
        la = self._ahead
        if la is not None and la[0] is I0 and la[1] is I1:
            self._ahead = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self._fstream)
            la[3].record_stream(cur)
            la[2].record_stream(cur)
            self.flow_net.redo_if_flagged()  # (the pass ended long ago: no stall)
            return la[3]
        self._drop_ahead()
        x = torch.cat((I0, I1), dim=1).contiguous()  # 2 channels; the large concats are fused in-kernel
        return self.flow_net.forward(x)

    def _drop_ahead(self):
        """An unclaimed look-ahead pass: the flow UNet's workspace is its, so it finishes before the network runs again."""
        if self._ahead is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._fstream)
            self.flow_net._pending = None
            self._ahead = None

    def flow_ahead(self, I0, I1, after=None):
        """Start the flow UNet of a LATER batch on the engine's side stream, behind what the current stream holds NOW -- or behind
        the event `after` recorded on it earlier (interpolate() records one BEFORE it enqueues the batch's interpolation and calls
        this behind it: the side stream still starts where the event is, but the current stream's work no longer waits for the host
        to enqueue these 23 launches first).  At the v2e batch size (8 pairs) the flow UNet is a chain of small
        launches that leaves most of the chip idle -- 3 ms of a 21 ms batch; beside the previous batch's interpolation UNet it is
        nearly free.  `flow(I0, I1)` with the same two tensors collects the result."""
        self._drop_ahead()
        if self._fstream is None:
            # (a high-priority stream: its own hardware queue whatever else the process has created -- streams of one priority
            #  share a few queues round-robin, and on the queue of the current stream the pass would simply run in line)
            self._fstream = torch.cuda.Stream(self.device, priority=-1)
        cur = torch.cuda.current_stream(self.device)
        if after is not None:
            self._fstream.wait_event(after)
        else:
            self._fstream.wait_stream(cur)
        with torch.cuda.stream(self._fstream):
            x = torch.cat((I0, I1), dim=1).contiguous()
            f = self.flow_net.forward(x, defer_check=True)
        self._ahead = (I0, I1, x, f)

    def self_check(self, I0, I1, ts=(0.5,), reference=None):
        """This engine's conv math against the float32-MFMA kernels (the reference's own arithmetic type) on one batch: the largest
        |a - b| of the flow UNet's output, the interpolation UNet's output and the interpolated frames, each in units of its tensor's
        scale max(1, max |b|) ("flow", "intrp", "Ft", "max"; "per_element" = the same with max(1, |b|) per element, which on
        trained-scale weights is ~2e-4 for EVERY conv math incl. the exact split -- float32 summation order on cancelling sums, the
        reference's own float32 noise against float64 is of that size there: tests/test_slomo_gpu.py, noise ratios 1.2 - 1.5).  The default
        conv math carries 22-bit operands; it has met the 1e-5 tolerance on every fixture, but those are seeded random weights (the
        pretrained checkpoint is not obtainable offline) -- SuperSloMo runs this on the FIRST batch it sees and moves to the exact
        three-bf16-piece split if the figure is above 1e-5 (round-4 review, missing item 4).  reference: a float32 SloMoEngine to
        compare with (built from the same weights when None; ~160 MB and two UNet passes, once)."""
        ref = reference if reference is not None else SloMoEngine(self._state_dicts[0], self._state_dicts[1], self.device, conv_math="f32")
        ts = list(ts)
        res = {}
        a = self.interpolate(I0, I1, ts)
        la = dict(self.last)
        b = ref.interpolate(I0, I1, ts)
        lb = ref.last

        def err(x, y):  # in units of the tensor's scale
            return float((x - y).abs().max() / y.abs().max().clamp_min(1.0))

        def err_elem(x, y):  # per element
            return float(((x - y).abs() / y.abs().clamp_min(1.0)).max())

        res["flow"], res["intrp"], res["Ft"] = err(la["flow"], lb["flow"]), err(la["intrp"], lb["intrp"]), err(a, b)
        res["max"] = max(res.values())
        res["per_element"] = max(err_elem(la["flow"], lb["flow"]), err_elem(la["intrp"], lb["intrp"]), err_elem(a, b))
        return res

    def lookahead_ok(self):
        """True iff two UNet passes on two streams reproduce the one-stream result bit for bit on this device with this build of the
        library: one small interpolation-UNet pass alone, then twice beside three flow-UNet passes on the side stream (the
        configuration of profiles/r04_concurrency_finding.txt table 1, which failed 4-6 runs of 6 with the faulty build).
        Run once per process and device (a few ms); a mismatch switches the look-ahead off for the process, with one warning."""
        key = self.device.index
        if key not in _CANARY:
            g = torch.Generator(device=self.device)
            g.manual_seed(1234)
            xi = torch.rand((2, 12, 64, 96), device=self.device, generator=g) - 0.4
            xf = torch.rand((2, 2, 64, 96), device=self.device, generator=g) - 0.4
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device, priority=-1)
            ref = self.interp_net.forward(xi).clone()
            ok = True
            for _ in range(2):
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(3):
                        self.flow_net.forward(xf)
                out = self.interp_net.forward(xi)
                ok = ok and bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
                cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            _CANARY[key] = ok
            if not ok:
                logger.warning("v2e_amd.SloMoEngine: a UNet pass beside another stream's pass differs from its stand-alone result on "
                               "this device with this build of libv2e_amd.so (was it built without -fno-slp-vectorize?): the flow "
                               "look-ahead is off for this process, every pass runs on the caller's stream")
        return _CANARY[key]

    def max_speed(self, flow):
        """slomo.py:352-368: the largest flow magnitude (pixels per source frame) over the batch and both directions, reduced on
        the device (k_max_speed2: max of the squared speed; ONE float32 sqrt on the host -- sqrt is monotone)."""
        b, c, h, w = flow.shape
        assert c == 4 and flow.dtype == torch.float32 and flow.is_contiguous()
        if self._speed_bits is None:
            self._speed_bits = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(self.lib.v2e_slomo_max_speed2(_ptr(flow), b, h, w, _ptr(self._speed_bits), self._stream()), "v2e_slomo_max_speed2")
        bits = int(self._speed_bits.item()) & 0xFFFFFFFF
        if bits == 0xFFFFFFFF:
            return float("nan")
        return float(np.sqrt(np.array([bits], dtype=np.uint32).view(np.float32)[0]))

    def interpolate(self, I0, I1, ts, flow=None, next_pair=None):
        """Ft_p for every t in ts and every pair in the batch: returns [len(ts), B, 1, H, W].
        next_pair = (I0, I1) of the batch the caller will pass NEXT (the same tensor objects): its flow UNet is started on the
        side stream now and runs beside this batch's interpolation UNet (`flow_ahead`)."""
        B, _, H, W = I0.shape
        I0 = I0 if I0.is_contiguous() else I0.contiguous()
        I1 = I1 if I1.is_contiguous() else I1.contiguous()
        if flow is None:
            flow = self.flow(I0, I1)
        # (Round 5 measured the look-ahead enqueued BEHIND this batch's own launches, bound to an event recorded here, so that the
        # current stream does not idle for the ~1.4 ms of host time the 23 look-ahead launches take: 19.9 ms per batch either way --
        # the flow UNet does useful work alone in that gap, and beside the interpolation UNet it stretches it by as much.)
        if next_pair is not None and _LOOKAHEAD and self.lookahead_ok():
            self.flow_ahead(next_pair[0], next_pair[1])
        nt = len(ts)
        key = tuple(float(t) for t in ts)
        if getattr(self, "_coef_key", None) != key:  # (the time points are the same for every batch of a clip: one upload)
            self._coef, self._coef_key = torch.from_numpy(time_coefficients(ts)).to(self.device), key
        coef = self._coef
        x12 = torch.empty((nt * B, 12, H, W), dtype=torch.float32, device=self.device)
        check(self.lib.v2e_slomo_prep(_ptr(I0), _ptr(I1), _ptr(flow), _ptr(coef), nt, B, H, W, _ptr(x12), self._stream()),
              "v2e_slomo_prep")
        out = torch.empty((nt * B, 1, H, W), dtype=torch.float32, device=self.device)
        intrp = self.interp_net.forward(x12, defer_check=True)
        for _ in range(2):  # the range flag is read behind the fusion: one host synchronisation per batch
            check(self.lib.v2e_slomo_fuse(_ptr(I0), _ptr(I1), _ptr(x12), _ptr(intrp), _ptr(coef), nt, B, H, W, _ptr(out),
                                          self._stream()), "v2e_slomo_fuse")
            if not self.interp_net.redo_if_flagged():
                break
        self.last = dict(flow=flow, x12=x12, intrp=intrp)
        return out.view(nt, B, 1, H, W)


def _cleanup_at_exit(ref):
    obj = ref()
    if obj is not None:
        obj.cleanup()


class SuperSloMo(object):
    """Super SloMo class (MI355X implementation of v2ecore.slomo.SuperSloMo)."""

    def __init__(self, model, auto_upsample, upsampling_factor, batch_size=1, video_path=None,
                 vid_orig='original.avi', vid_slomo='slomo.avi', preview=False, avi_frame_rate=30):
        if torch.cuda.is_available():
            self.device = "cuda:0"  # slomo.py:84
        else:
            self.device = "cpu"
            logger.warning('no ROCm GPU visible: v2e_amd.SuperSloMo cannot interpolate (no CPU fallback)')
        self.checkpoint = model
        self.batch_size = batch_size
        if not auto_upsample and (not isinstance(upsampling_factor, int) or upsampling_factor < 2):
            raise ValueError('upsampling_factor={} but must be an int value>1 when auto_upsample=True'.format(
                upsampling_factor))  # slomo.py:91-94
        self.upsampling_factor = upsampling_factor
        self.auto_upsample = auto_upsample
        # slomo.py:110-121: kept as given.  The writers are built on first need in interpolate() (the frame size is known there)
        self.video_path = video_path
        self.preview = preview
        self.preview_resized = False
        self.vid_orig, self.vid_slomo, self.avi_frame_rate = vid_orig, vid_slomo, avi_frame_rate
        self.ori_writer = None
        self.slomo_writer = None
        self.numOrigVideoFramesWritten = 0
        self.numSlomoVideoFramesWritten = 0
        self._cv2 = None
        self._host_video_warned = False
        self.model_loaded = False
        self.engine = None
        self.mean = 0.428  # slomo.py:148: Normalize(mean=[0.428], std=[1]) on the GPU path only
        # (slomo.py:124 registers cleanup with atexit; through a weak reference: a bound method would keep the instance, its engine and
        #  their device memory alive until the process ends)
        atexit.register(_cleanup_at_exit, weakref.ref(self))

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    def cleanup(self):  # slomo.py:126-138
        for attr, name, n in (("ori_writer", "vid_orig", "numOrigVideoFramesWritten"),
                              ("slomo_writer", "vid_slomo", "numSlomoVideoFramesWritten")):
            w = getattr(self, attr, None)
            if w is not None:
                logger.info('closing video AVI {} after writing {} frames'.format(getattr(self, name), getattr(self, n)))
                try:
                    w.release()
                except Exception:
                    pass
                setattr(self, attr, None)
        if getattr(self, "preview", False) and getattr(self, "_cv2", None) is not None:
            try:
                self._cv2.destroyAllWindows()
            except Exception:
                pass

    def _host_video(self):
        """cv2 for the AVI / preview side outputs (host work, not the hot path), or None with one warning."""
        if self._cv2 is not None:
            return self._cv2
        try:
            import cv2
            if not callable(getattr(cv2, "VideoWriter", None)) or not callable(getattr(cv2, "VideoWriter_fourcc", None)):
                raise ImportError("cv2 has no VideoWriter")
            self._cv2 = cv2
        except Exception as e:
            if not self._host_video_warned:
                logger.warning("v2e_amd.SuperSloMo: video_path=%r / preview=%r need OpenCV on the host (%s): the AVI / preview "
                               "side outputs are skipped, the interpolated frames are written as usual",
                               self.video_path, self.preview, e)
                self._host_video_warned = True
        return self._cv2

    def _open_writers(self, ori_dim):
        """slomo.py:287-301: the AVI writers, now that the frame size is known (v2e_utils.video_writer: codec
        OUTPUT_VIDEO_CODEC_FOURCC, frame size (width, height))."""
        if self.video_path is None or (self.vid_orig is None and self.vid_slomo is None):
            return
        cv2 = self._host_video()
        if cv2 is None:
            return
        fourcc = cv2.VideoWriter_fourcc(*OUTPUT_VIDEO_CODEC_FOURCC)
        size = (int(ori_dim[0]), int(ori_dim[1]))  # frame_size is (width, height), as cv2.VideoWriter wants it
        if self.vid_orig is not None and self.ori_writer is None:
            self.ori_writer = cv2.VideoWriter(os.path.join(self.video_path, self.vid_orig), fourcc, self.avi_frame_rate, size)
        if self.vid_slomo is not None and self.slomo_writer is None:
            self.slomo_writer = cv2.VideoWriter(os.path.join(self.video_path, self.vid_slomo), fourcc, self.avi_frame_rate, size)

    def _write_videos(self, source_frame_path, output_folder):
        """slomo.py:463-490: the source frames and the interpolated PNGs, grey -> 3 equal channels, into the two AVIs."""
        from PIL import Image
        if self.ori_writer is not None:
            for f in sorted(glob.glob("{}".format(source_frame_path) + "/*.npy")):
                a = np.load(f)
                if a.ndim == 2:
                    a = np.repeat(a[:, :, None], 3, axis=2)  # cv2.COLOR_GRAY2BGR
                self.ori_writer.write(np.ascontiguousarray(a))
                self.numOrigVideoFramesWritten += 1
        if self.slomo_writer is not None:
            for f in _all_images(output_folder):
                a = np.asarray(Image.open(f).convert("L"))
                self.slomo_writer.write(np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2)))
                self.numSlomoVideoFramesWritten += 1

    def _preview_batch(self, output_folder, first, last):
        """slomo.py:451-468: show the frames just written (needs a display and cv2; off with one warning otherwise)."""
        cv2 = self._host_video()
        if cv2 is None or not callable(getattr(cv2, "imshow", None)):
            return
        try:
            name = "v2e_amd.slomo"
            cv2.namedWindow(name, cv2.WINDOW_NORMAL)
            for idx in range(first, last):
                frame = cv2.imread(os.path.join(output_folder, str(idx) + ".png"))
                cv2.imshow(name, frame)
                if not self.preview_resized:
                    cv2.resizeWindow(name, 800, 600)
                    self.preview_resized = True
                k = cv2.waitKey(1)
                if k == 27 or k == ord('x'):
                    raise SystemExit(0)  # v2e_quit()
        except SystemExit:
            raise
        except Exception as e:  # no display: the side output is dropped
            logger.warning("v2e_amd.SuperSloMo: preview not available (%s)", e)
            self.preview = False

    def _load_model(self):
        if not os.path.isfile(self.checkpoint):
            raise FileNotFoundError('SuperSloMo model checkpoint ' + str(self.checkpoint) +
                                    ' does not exist or is not readable')  # slomo.py:203-205
        logger.info('loading SuperSloMo model from ' + str(self.checkpoint))
        d = torch.load(self.checkpoint, map_location="cpu", weights_only=False)
        self.engine = SloMoEngine(d['state_dictFC'], d['state_dictAT'], self.device)  # slomo.py:225-227
        self.model_loaded = True
        self._selfcheck_pending = os.environ.get("V2E_AMD_SLOMO_SELFCHECK", "1") != "0"

    SELFCHECK_TOLERANCE = 1e-5  # the tolerance the drop-in promises against the reference's float32 arithmetic
    SELFCHECK_WORSE_THAN_EXACT = 2.0  # ... and how much further from the float32 kernels than the exact split 'auto' may be beyond it

    def _first_batch_selfcheck(self, I0, I1):
        """Once per loaded checkpoint, on the first two pairs that come in: the default conv math ('auto': two float16 pieces) against
        the float32-MFMA kernels, in units of each tensor's scale.  Within 1e-5: done.  Above: the exact split ('bf16x3') is measured
        the same way -- on weights with large heads (visibility logits to 100) two float32 summation orders differ by a few 1e-5 in
        the blended frames whatever the operands carry -- and 'auto' is kept if it is no more than 2x further from the float32
        kernels than the exact split; otherwise the engine is rebuilt with the exact split, with one warning.
        V2E_AMD_SLOMO_SELFCHECK=0 skips all this; `python -m v2e_amd.check_ckpt <ckpt>` prints the per-layer picture."""
        self._selfcheck_pending = False
        eng = self.engine
        if eng.conv_math != "auto":
            return
        ref = SloMoEngine(eng._state_dicts[0], eng._state_dicts[1], self.device, conv_math="f32")
        r = eng.self_check(I0[:2], I1[:2], reference=ref)
        self.selfcheck_result = r
        if r["max"] <= self.SELFCHECK_TOLERANCE:
            return
        exact = SloMoEngine(eng._state_dicts[0], eng._state_dicts[1], self.device, conv_math="bf16x3")
        rx = exact.self_check(I0[:2], I1[:2], reference=ref)
        r["exact_split_max"] = rx["max"]
        if r["max"] <= self.SELFCHECK_WORSE_THAN_EXACT * rx["max"]:  # (a NaN fails)
            logger.info("v2e_amd.SuperSloMo: default conv math %.2e from the float32 kernels on the first batch (of the tensors' scale); "
                        "the exact split is %.2e from them: float32 summation order, not operand precision -- keeping the default",
                        r["max"], rx["max"])
            return
        logger.warning("v2e_amd.SuperSloMo: with this checkpoint the default conv math differs from the float32 kernels by %.2e of the "
                       "tensors' scale (flow %.2e, interpolation net %.2e, frames %.2e) on the first batch -- above %.0e and %.1fx what "
                       "the exact split shows (%.2e): switching to the exact three-piece split (conv_math='bf16x3', ~0.7x the frames "
                       "per second)", r["max"], r["flow"], r["intrp"], r["Ft"], self.SELFCHECK_TOLERANCE, r["max"] / max(rx["max"], 1e-30),
                       rx["max"])
        self.engine = exact

    @staticmethod
    def _load_pair_tensor(files, idx, dim):
        """dataloader.py:136-147: np.load -> PIL LANCZOS resize -> ToTensor ([1,H,W] float32 in 0..1)."""
        from PIL import Image
        out = []
        for f in (files[idx], files[idx + 1]):
            img = Image.fromarray(np.load(f)).resize(dim, Image.LANCZOS)
            a = np.asarray(img)
            if a.ndim == 2:
                a = a[None]
            else:
                a = np.transpose(a, (2, 0, 1))
            out.append(torch.from_numpy(a.astype(np.float32) / 255.0))  # ToTensor on uint8
        return out

    def interpolate(self, source_frame_path, output_folder, frame_size):
        """Run interpolation; writes <idx>.png into output_folder (slomo.py:231-495)."""
        from PIL import Image
        if not output_folder:
            raise ValueError('output_folder is None; it must be supplied to store the interpolated frames')
        if self.device == "cpu":
            raise _capi.V2EAmdError("v2e_amd.SuperSloMo needs a ROCm GPU; there is no CPU fallback")
        nframes = len(os.listdir(source_frame_path))
        if nframes / self.batch_size < 2:
            logger.warning(f'only {nframes} input frames with batch_size={self.batch_size}, automatically '
                           f'reducing batch size to provide at least 2 batches')
            while nframes / self.batch_size < 2:
                self.batch_size = int(self.batch_size / 2)
        files = sorted(glob.glob("{}".format(source_frame_path) + "/*.npy"))  # dataloader.py:115
        ori_dim = frame_size
        dim = (int(ori_dim[0] / 32) * 32, int(ori_dim[1] / 32) * 32)  # dataloader.py:122-123
        npairs = len(files) - 1
        nbatches = (npairs + self.batch_size - 1) // self.batch_size
        if nbatches < 2:
            raise Exception('there are only {} batches in {} and we need at least 2; maybe you need to reduce batch '
                            'size or increase number of input frames'.format(nbatches, source_frame_path))
        if not self.model_loaded:
            self._load_model()
        self._open_writers(ori_dim)
        outputFrameCounter = 0
        inputFrameCounter = 0
        upsamplingSum = 0
        nUpsamplingSamples = 0
        interpTimes = None
        dev = self.engine.device
        def load_batch(b0):
            idxs = list(range(b0, min(b0 + self.batch_size, npairs)))
            pairs = [self._load_pair_tensor(files, i, dim) for i in idxs]
            return ((torch.stack([p[0] for p in pairs]) - self.mean).to(dev).contiguous(),
                    (torch.stack([p[1] for p in pairs]) - self.mean).to(dev).contiguous())
        starts = list(range(0, npairs, self.batch_size))
        nxt = load_batch(starts[0])
        for bi, b0 in enumerate(starts):
            # the next batch is loaded first: its flow UNet runs beside this batch's interpolation UNet (SloMoEngine.flow_ahead)
            I0, I1 = nxt
            nxt = load_batch(starts[bi + 1]) if bi + 1 < len(starts) else None
            if getattr(self, "_selfcheck_pending", False):
                self._first_batch_selfcheck(I0, I1)
            num_batch_frames = I0.shape[0]
            flowOut = self.engine.flow(I0, I1)
            if self.auto_upsample:  # slomo.py:352-379
                maxSpeed = self.engine.max_speed(flowOut)
                upsampling_factor = int(np.ceil(maxSpeed))
                if self.upsampling_factor is not None and self.upsampling_factor > upsampling_factor:
                    upsampling_factor = self.upsampling_factor
            else:
                upsampling_factor = self.upsampling_factor
            if upsampling_factor < 2:
                upsampling_factor = 2
            nUpsamplingSamples += 1
            upsamplingSum += upsampling_factor
            numOutputFramesThisBatch = upsampling_factor * num_batch_frames
            interframeTimes = inputFrameCounter + np.array(range(numOutputFramesThisBatch)) * (1 / upsampling_factor)
            interpTimes = interframeTimes if interpTimes is None else np.concatenate((interpTimes, interframeTimes))
            ts = [(k + 0.5) / upsampling_factor for k in range(upsampling_factor)]  # slomo.py:405
            Ft = self.engine.interpolate(I0, I1, ts, flow=flowOut, next_pair=nxt)  # [U,B,1,H,W]
            # ToPILImage after revNormalize (slomo.py:153-161, 437): (x + mean) * 255 -> byte, the CPU conversion the
            # reference performs (truncation, low byte): same kernel as the PNG-free pipeline uses
            import ctypes as C
            q = torch.empty(tuple(Ft.shape), dtype=torch.uint8, device=dev)
            check(self.engine.lib.v2e_f32_to_u8_trunc(_ptr(Ft.contiguous()), _ptr(q), upsampling_factor, num_batch_frames,
                                                      int(Ft.shape[-1] * Ft.shape[-2]), float(self.mean), 0,
                                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "v2e_f32_to_u8_trunc")
            img_u8 = q.cpu().numpy()
            for k in range(upsampling_factor):
                for batchIndex in range(num_batch_frames):
                    img = Image.fromarray(img_u8[k, batchIndex, 0], mode="L")
                    img_resize = img.resize(ori_dim, Image.BILINEAR)
                    outputFrameIdx = outputFrameCounter + upsampling_factor * batchIndex + k
                    img_resize.save(os.path.join(output_folder, str(outputFrameIdx) + ".png"))
            if self.preview:
                self._preview_batch(output_folder, outputFrameCounter, outputFrameCounter + numOutputFramesThisBatch)
            inputFrameCounter += num_batch_frames
            outputFrameCounter += numOutputFramesThisBatch
        self._write_videos(source_frame_path, output_folder)
        avgUpsampling = upsamplingSum / nUpsamplingSamples
        return interpTimes, avgUpsampling

    def get_interpolated_timestamps(self, ts):  # slomo.py:540-564
        new_ts = []
        for i in range(ts.shape[0] - 1):
            start, end = ts[i], ts[i + 1]
            interpolated_ts = np.linspace(start, end, self.upsampling_factor, endpoint=False) + \
                0.5 * (end - start) / self.upsampling_factor
            new_ts.append(interpolated_ts)
        return np.hstack(new_ts)
