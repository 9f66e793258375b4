"""SloMo beside the emulator chain on two HIP streams (round-4 review, weak items 2 and 10).

Two things ride on this: (1) the silent-corruption mode that -fno-slp-vectorize removed (profiles/r04_concurrency_finding.txt) showed
only when a UNet pass ran beside matrix-core kernels of another stream -- here every conv math runs beside the other stream's
work and must equal its stand-alone output bit for bit; (2) k_chain's redo rendezvous spins inside a normal launch and infers
co-residency from an occupancy query, while a foreign LDS- and wave-hungry kernel (the convolutions) takes slots on the same CUs
-- the refractory fixture's redo passes must still finish (no V2E_FLAG_SYNC_TIMEOUT) and reproduce the reference's digests."""
import numpy as np
import pytest
import torch

from fixtures import PhiloxFixture, sha

pytestmark = pytest.mark.gpu


def _slomo(conv_math):
    from v2e_amd.slomo import SloMoEngine
    from v2e_amd.synth import portable_unet_state_dict
    sd_f, sd_i = portable_unet_state_dict(2, 4, 101), portable_unet_state_dict(12, 5, 102)
    return SloMoEngine({k: torch.from_numpy(v) for k, v in sd_f.items()}, {k: torch.from_numpy(v) for k, v in sd_i.items()},
                       "cuda", conv_math=conv_math)


@pytest.mark.parametrize("conv_math", ["auto", "bf16x3", "f32"])
@pytest.mark.parametrize("fixture", ["philox_refractory_346x260", "philox_defaults_346x260"])
def test_slomo_and_emulator_chain_on_two_streams_equal_their_standalone_runs(conv_math, fixture):
    from v2e_amd import EventEmulator
    dev = torch.device("cuda")
    fx = PhiloxFixture(fixture)
    eng = _slomo(conv_math)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    I0 = torch.rand((4, 1, 128, 160), device=dev, generator=g) - 0.428
    I1 = torch.rand((4, 1, 128, 160), device=dev, generator=g) - 0.428
    ts = [(k + 0.5) / 4 for k in range(4)]
    ref_ft = eng.interpolate(I0, I1, ts).clone()
    torch.cuda.synchronize()

    def emu_run():
        emu = EventEmulator(device="cuda", seed=fx.seed, rng_mode="philox", **fx.kw)
        if fx.preset:
            emu.set_dvs_params(fx.preset)
        pend = emu.generate_events_batch_async(fx.frames, fx.times, return_device=True)
        return emu, pend

    s_emu, s_slomo = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for rep in range(3):
        outs = []
        with torch.cuda.stream(s_emu):
            emu, pend = emu_run()
        with torch.cuda.stream(s_slomo):
            for _ in range(6):  # keeps matrix-core kernels in flight for the whole emulator run
                outs.append(eng.interpolate(I0, I1, ts).clone())
        with torch.cuda.stream(s_emu):
            ev, counts = pend.result()
            ev = ev.cpu().numpy()
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o.view(torch.int32), ref_ft.view(torch.int32)), "SloMo beside the emulator differs from its stand-alone run"
        assert list(counts) == list(fx.n_events)
        row = 0
        for k, n in enumerate(counts):
            if n:
                assert sha(ev[row:row + n]) == fx.ev_sha[k], "frame %d event digest differs beside SloMo" % k
            row += n
        assert sha(emu.base_log_frame.cpu().numpy()) == fx.base_sha
        if fx.ts_mem_sha:
            assert sha(emu.timestamp_mem.cpu().numpy()) == fx.ts_mem_sha


def test_two_unet_passes_on_two_streams_every_math_pair():
    """Table 1 of profiles/r04_concurrency_finding.txt as a standing test: interpolation UNet (victim) beside the flow UNet run
    three times on a side stream, all pairs of conv maths; built as committed every pair is bit-identical."""
    from v2e_amd.slomo import HipUNet
    from v2e_amd.synth import portable_unet_state_dict
    dev = torch.device("cuda")
    sd_f = {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(2, 4, 101).items()}
    sd_i = {k: torch.from_numpy(v) for k, v in portable_unet_state_dict(12, 5, 102).items()}
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    xi = torch.rand((4, 12, 64, 96), device=dev, generator=g) - 0.4
    xf = torch.rand((2, 2, 64, 96), device=dev, generator=g) - 0.4
    side = torch.cuda.Stream(dev)
    maths = ("bf16x3", "fp16x2", "f32")
    victims = {m: HipUNet(sd_i, 12, 5, dev, m) for m in maths}
    noises = {m: HipUNet(sd_f, 2, 4, dev, m) for m in maths}
    for vm, vnet in victims.items():
        ref = vnet.forward(xi).clone()
        torch.cuda.synchronize()
        for nm, nnet in noises.items():
            for rep in range(3):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        nnet.forward(xf)
                out = vnet.forward(xi).clone()
                torch.cuda.synchronize()
                assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (vm, nm, rep)


def test_slomo_pair_sharding_over_an_rccl_group_of_one():
    """VideoToEvents.run(group=): the sharded SuperSloMo stage and the in-order frame gather over RCCL (world size 1: the only size a
    one-GPU box has; the world-2 / world-4 logic runs on gloo in tests/test_dist_cpu.py) equals the unsharded pipeline bit for bit."""
    import os
    import torch.distributed as dist
    from v2e_amd import EventEmulator
    from v2e_amd.pipeline import VideoToEvents
    from v2e_amd.synth import int_gradient_frames
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        kw = dict(pos_thres=.2, neg_thres=.2, sigma_thres=.03, cutoff_hz=300, leak_rate_hz=.01, shot_noise_rate_hz=.001, refractory_period_s=.0005)
        frames = torch.from_numpy(int_gradient_frames(6, 64, 96, seed=4, noise=5, as_array=True)).cuda()
        res = []
        for group in (None, dist.group.WORLD):
            pipe = VideoToEvents(_slomo("auto"), EventEmulator(device="cuda", seed=3, rng_mode="philox", **kw), 4, batch_size=2)
            ev, counts, n = pipe.run(frames, 1 / 30, group=group) if group is not None else pipe.run(frames, 1 / 30)
            res.append((np.asarray(ev), np.asarray(counts), n))
        assert res[0][2] == res[1][2] == 20 and np.array_equal(res[0][1], res[1][1])
        assert res[0][0].shape == res[1][0].shape and np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
        # in chunks (five chunks of one pair: the emulator's two event buffers are reused twice while earlier chunks' rows are held),
        # rows on the host and on the device
        for rd in (False, True):
            pipe = VideoToEvents(_slomo("auto"), EventEmulator(device="cuda", seed=3, rng_mode="philox", **kw), 4, batch_size=2)
            ev, counts, n = pipe.run(frames, 1 / 30, group=dist.group.WORLD, owner=0, chunk_pairs=1, return_device=rd)
            ev = ev.cpu().numpy() if rd else np.asarray(ev)
            assert n == 20 and np.array_equal(np.asarray(counts), res[0][1]), rd
            assert ev.shape == res[0][0].shape and np.array_equal(ev.view(np.uint32), res[0][0].view(np.uint32)), rd
    finally:
        if created:
            dist.destroy_process_group()
