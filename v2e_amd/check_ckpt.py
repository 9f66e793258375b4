"""python -m v2e_amd.check_ckpt SuperSloMo39.ckpt [--frames a.npy b.npy] [--size 320x256]

Does the default SuperSloMo arithmetic of this package stay inside the drop-in's tolerance WITH THESE WEIGHTS?

The reference computes the UNets in float32 (slomo.py:188-229 loads the checkpoint, model.py:10-226 runs it).  Here the default conv
math ('auto') splits every float32 operand into two float16 pieces (22 significant bits, three piece products on the f16 matrix
cores); 'bf16x3' is the exact three-piece split, 'f32' the float32 matrix instructions.  All fixtures of this repository are seeded
random weights -- the pretrained checkpoint cannot be downloaded into the build -- so the 1e-5 claim is re-measured here on the
weights you actually run: one pair of frames at 320x256 (a synthetic moving gradient unless --frames gives two uint8 .npy frames),

  * end to end: flow UNet output, interpolation UNet output and the interpolated frame of each conv math against the float32 kernels,
    as max |a - b| in units of the tensor's scale max(1, max |b|) -- the figure the verdict is taken on -- and per element
    (max |a - b| / max(1, |b|)); on trained-scale weights the per-element figure is ~1e-4 for EVERY conv math, the exact split
    included: two float32 summation orders on cancelling sums (the reference's own float32 result is that far from float64 there),
    which is why 'auto' is judged against the tensor's scale and against what the exact split achieves;
  * layer by layer: every one of the 2 x 23 convolutions of each conv math run in isolation on the float32 network's own input to that
    layer (errors do not accumulate: which layer, if any, is the weak one).  A two-piece layer run in isolation stages its
    activations unscaled, so its isolated figure is an upper bound for what it does inside the network.

Exit status 0 when 'auto' is within 1e-5 of the tensors' scale end to end or no further from the float32 kernels than 2x the exact
split, 1 when it is not (SuperSloMo then falls back to 'bf16x3' by itself on the first
batch, with a warning; conv_math can also be forced with V2E_AMD_CONV_MATH)."""
import argparse
import sys

import numpy as np
import torch

TOL = 1e-5


def _err(a, b):  # in units of the tensor's scale
    return float((a - b).abs().max() / b.abs().max().clamp_min(1.0))


def check(flow_sd, interp_sd, I0, I1, device="cuda", out=sys.stdout, maths=("auto", "bf16x3")):
    """Returns {conv_math: {"flow", "intrp", "Ft", "max", "layers": [(net, name, err)]}}; prints the tables to `out`."""
    from .slomo import SloMoEngine
    ref = SloMoEngine(flow_sd, interp_sd, device, conv_math="f32")
    ts = [0.5]
    ref.interpolate(I0, I1, ts)
    x_flow = torch.cat((I0, I1), dim=1).contiguous()
    x12 = ref.last["x12"]
    walks = {"flow": ref.flow_net.walk_layers(x_flow), "interp": ref.interp_net.walk_layers(x12)}
    whole = {"flow": ref.last["flow"], "interp": ref.last["intrp"]}
    for net in ("flow", "interp"):  # the walk is the same network as v2e_unet_forward
        d = _err(walks[net][-1][2], whole[net])
        print("float32 kernels, %-6s UNet: layer walk vs v2e_unet_forward %.2e" % (net, d), file=out)
    res = {}
    for m in maths:
        eng = SloMoEngine(flow_sd, interp_sd, device, conv_math=m)
        r = eng.self_check(I0, I1, ts, reference=ref)
        layers = []
        for net, hn in (("flow", eng.flow_net), ("interp", eng.interp_net)):
            for (name, _, y), (_, _, yr) in zip(hn.walk_layers(walks[net][0][1][0], inputs_from=walks[net]), walks[net]):
                layers.append((net, name, _err(y, yr)))
        r["layers"] = layers
        r["fallbacks"] = eng.flow_net.fallbacks + eng.interp_net.fallbacks
        res[m] = r
        print("\nconv_math %-7s end to end vs float32 kernels (of the tensor's scale): flow %.2e  interpolation net %.2e  frames %.2e   -> %s"
              "   [per element: %.2e]" % (m, r["flow"], r["intrp"], r["Ft"], "within 1e-5" if r["max"] <= TOL else "ABOVE 1e-5", r["per_element"]), file=out)
        if r["fallbacks"]:
            print("   (range guard: %d forward pass(es) were redone with the exact split)" % r["fallbacks"], file=out)
        worst = sorted(layers, key=lambda q: -q[2])[:5]
        print("   isolated layers, worst five: " + ", ".join("%s.%s %.1e" % q for q in worst), file=out)
    print("\n%-8s %-14s" % ("net", "layer") + "".join("%12s" % m for m in maths), file=out)
    for i in range(len(res[maths[0]]["layers"])):
        net, name, _ = res[maths[0]]["layers"][i]
        print("%-8s %-14s" % (net, name) + "".join("%12.2e" % res[m]["layers"][i][2] for m in maths), file=out)
    return res


def synthetic_pair(h, w, device):
    """Two frames of a moving gradient with texture, uint8 -> float32 / 255 - mean as SuperSloMo feeds them (slomo.py:138-161)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    fr = []
    for k in range(2):
        a = 127 + 90 * np.sin((x + 5 * k) / 17.0) * np.cos((y - 3 * k) / 23.0) + 20 * np.sin((x * 0.9 + y * 1.3 + 7 * k) / 3.0)
        fr.append(np.clip(a, 0, 255).astype(np.uint8))
    t = [torch.from_numpy(f.astype(np.float32) / 255.0 - np.float32(0.428))[None, None].to(device).contiguous() for f in fr]
    return t[0], t[1]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m v2e_amd.check_ckpt", description=__doc__.split("\n\n")[1])
    ap.add_argument("checkpoint", help="SuperSloMo39.ckpt (torch.load: keys state_dictFC / state_dictAT)")
    ap.add_argument("--frames", nargs=2, metavar="NPY", help="two uint8 grey frames (.npy) instead of the synthetic pair")
    ap.add_argument("--size", default="320x256", help="WxH the networks run at (multiples of 32; default 320x256 = a 346x260 source)")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("v2e_amd.check_ckpt needs a ROCm GPU (the kernels under test are HIP kernels)")
    w, h = (int(v) for v in args.size.lower().split("x"))
    d = torch.load(args.checkpoint, map_location="cpu", weights_only=False)
    if args.frames:
        from PIL import Image
        t = []
        for f in args.frames:
            a = np.asarray(Image.fromarray(np.load(f)).convert("L").resize((w, h), Image.LANCZOS))
            t.append(torch.from_numpy(a.astype(np.float32) / 255.0 - np.float32(0.428))[None, None].cuda().contiguous())
        I0, I1 = t
    else:
        I0, I1 = synthetic_pair(h, w, "cuda")
    res = check(d["state_dictFC"], d["state_dictAT"], I0, I1)
    a, x = res["auto"]["max"], res["bf16x3"]["max"]
    if a <= TOL:
        print("\n'auto' is within 1e-5 of the float32 kernels (of the tensors' scale) on this checkpoint: nothing to do")
        return 0
    if a <= 2.0 * x:
        print("\n'auto' is %.2e from the float32 kernels, the EXACT split %.2e: float32 summation order on these weights, not operand "
              "precision (no float32 implementation is closer); SuperSloMo keeps 'auto'" % (a, x))
        return 0
    print("\n'auto' is %.2e from the float32 kernels, %.1fx the exact split (%.2e): SuperSloMo will fall back to conv_math='bf16x3' on "
          "its first batch" % (a, a / max(x, 1e-30), x))
    return 1


if __name__ == "__main__":
    sys.exit(main())
