"""ctypes binding of include/v2e_amd.h (libv2e_amd.so, built by __graft_entry__.build()).

The library is the product: there is no fallback.  `lib()` raises if the shared
object is missing or lacks a declared symbol.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V2E_AMD_LIB") or os.path.join(_HERE, "csrc", "libv2e_amd.so")  # (V2E_AMD_LIB: dev builds for A/B)

DT_U8, DT_F32, DT_F64 = 0, 1, 2
RNG_TAPE, RNG_PHILOX = 0, 1
FLAG_EVENTS_DROPPED, FLAG_ITERS_CLAMPED, FLAG_SYNC_TIMEOUT = 1, 2, 4


class EmuParams(C.Structure):
    """struct v2e_emu_params (include/v2e_amd.h)."""
    _fields_ = [
        ("f64_state", C.c_int32),
        ("scalar_thres", C.c_int32),
        ("rng_mode", C.c_int32),
        ("shuffle", C.c_int32),
        ("pos_thres_nominal", C.c_double),
        ("neg_thres_nominal", C.c_double),
        ("pos_thres_scalar", C.c_double),
        ("neg_thres_scalar", C.c_double),
        ("sigma_thres", C.c_double),
        ("cutoff_hz", C.c_double),
        ("leak_rate_hz", C.c_double),
        ("leak_jitter_fraction", C.c_double),
        ("noise_rate_cov_decades", C.c_double),
        ("refractory_period_s", C.c_double),
        ("shot_noise_rate_hz", C.c_double),
        ("shot_noise_inten_factor", C.c_double),
        ("pos_pre_scalar", C.c_float),
        ("neg_pre_scalar", C.c_float),
        ("seed", C.c_uint64),
        ("log_input", C.c_int32),
        ("photoreceptor_noise", C.c_int32),
        ("photoreceptor_noise_vrms", C.c_double),
    ]


class FrameRec(C.Structure):
    """struct v2e_frame_rec."""
    _fields_ = [
        ("max_events", C.c_int32),
        ("flags", C.c_uint32),
        ("n_signal", C.c_uint32),
        ("n_events", C.c_uint32),
        ("n_on", C.c_uint32),
        ("n_off", C.c_uint32),
        ("ev_offset", C.c_uint64),
    ]


class ConvDesc(C.Structure):
    """struct v2e_conv_desc."""
    _fields_ = [
        ("weight", C.c_void_p),
        ("bias", C.c_void_p),
        ("cin", C.c_int32),
        ("cout", C.c_int32),
        ("ksize", C.c_int32),
        ("split_kind", C.c_int32),
        ("weight_s3", C.c_void_p),
    ]


_vp, _i, _u32, _u64, _i64, _d = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64, C.c_double
_PP = C.POINTER(EmuParams)

# name -> (restype, argtypes); must list every symbol include/v2e_amd.h declares
SIGNATURES = {
    "v2e_last_error": (C.c_char_p, []),
    "v2e_version": (_i, []),
    "v2e_emu_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "v2e_emu_destroy": (_i, [_vp]),
    "v2e_emu_npx_pad": (_i64, [_i, _i]),
    "v2e_emu_bind_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v2e_emu_init_state": (_i, [_vp, _PP, _vp, _i, _d, _vp, _vp, _vp, _vp]),
    "v2e_emu_set_pnoise": (_i, [_vp, _vp, _vp]),
    "v2e_emu_set_model_state_planes": (_i, [_vp, _vp, _vp, _vp]),
    "v2e_emu_set_scidvs": (_i, [_vp, _vp, _vp, _vp, _u32]),
    "v2e_emu_frame_host_rows": (_i, [_vp, _vp, _u64]),
    "v2e_emu_set_csdvs": (_i, [_vp, _vp]),
    "v2e_emu_set_csdvs_run": (_i, [_vp, _vp, _vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i), _i, _d, _vp]),
    "v2e_emu_lp_preview": (_i, [_vp, _PP, _vp, _i, C.POINTER(_d), C.POINTER(_d), _u32, _vp, _vp]),
    "v2e_csdvs_update": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _d, _i, _d, C.POINTER(_i), C.POINTER(_d), _vp]),
    "v2e_emu_count": (_i, [_vp, _PP, _vp, _i, C.POINTER(_d), C.POINTER(_d), _u32, _vp, _vp, _vp]),
    "v2e_emu_read_rec": (_i, [_vp, _u32, C.POINTER(FrameRec), _vp]),
    "v2e_emu_shot": (_i, [_vp, _PP, _vp, _i, _u32, _vp, _vp]),
    "v2e_emu_reserve_iters": (_i, [_vp, _i, _vp]),
    "v2e_emu_rank": (_i, [_vp, _PP, _u32, _vp, _i, _vp]),
    "v2e_emu_emit": (_i, [_vp, _PP, _u32, _vp, _i, _vp, _u64, C.POINTER(_u64), _vp]),
    "v2e_emu_read_iter_counts": (_i, [_vp, _u32, _i, C.POINTER(_u32), _vp]),
    "v2e_emu_permute": (_i, [_vp, _vp, _vp, _vp, _u64, _u64, _vp]),
    "v2e_emu_frame": (_i, [_vp, _PP, _vp, _i, _i, _d, _d, _u32, _vp, _u64, C.POINTER(_u32), C.POINTER(C.POINTER(C.c_float)), _vp]),
    "v2e_emu_run": (_i, [_vp, _PP, _vp, _i, _i, C.POINTER(_d), C.POINTER(_d), _u32, _vp, _u64,
                         _vp, _i, _vp]),
    "v2e_emu_run_join": (_i, [_vp, _vp]),
    "v2e_emu_run_ticket": (_i, [_vp]),
    "v2e_emu_event_writer": (_i, [_vp]),
    "v2e_emu_run_wait": (_i, [_vp, _i]),
    "v2e_emu_run_recs": (_vp, [_vp, _i, C.POINTER(_u64)]),
    "v2e_emu_last_profile": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d), C.POINTER(_d),
                                  C.POINTER(_i)]),
    "v2e_emu_chain_plan": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i]),
    "v2e_emu_last_profile_pipe": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "v2e_emu_last_profile_launches": (_i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "v2e_emu_last_pipeline": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "v2e_emu_launch_stamps": (_i, [_vp, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "v2e_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "v2e_pack_conv_weight_s3": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "v2e_pack_conv_weight_h2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "v2e_conv_set_range_flag": (_i, [_vp]),
    "v2e_split3_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "v2e_conv2d_lrelu": (_i, [_vp, _i, _vp, _i, _i, C.POINTER(ConvDesc), _vp, _i, _i, _i, _vp]),
    "v2e_unet_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "v2e_unet_forward": (_i, [_vp, _i, C.POINTER(ConvDesc), _i, _vp, _i, _i, _i, _vp, _vp]),
    "v2e_slomo_prep": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v2e_resample_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "v2e_resize_area_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v2e_bgr2gray_u8": (_i, [_vp, _vp, _i64, _i, _vp]),
    "v2e_u8_to_f32_norm": (_i, [_vp, _vp, _i64, C.c_float, _vp]),
    "v2e_f32_to_u8_trunc": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _i, _vp]),
    "v2e_events_pack_aedat2": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i64, _vp]),
    "v2e_events_pack_h5": (_i, [_vp, _vp, _i64, _vp]),
    "v2e_events_pack64": (_i, [_vp, _vp, _i64, _vp]),
    "v2e_events_unpack64": (_i, [_vp, _vp, _i64, _vp]),
    "v2e_events_pack32": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "v2e_events_pack32_scratch_words": (_i64, [_i64]),
    "v2e_events_unpack32": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "v2e_events_accumulate_frame": (_i, [_vp, _i64, _vp, _vp, _i, _i, _d, _d, _d, _d, _d, _vp]),
    "v2e_frame_normalize": (_i, [_vp, _vp, _i, _d, _vp]),
    "v2e_render_area_segments": (_i, [_vp, _i64, _vp, _i, _i, _d, _i, _vp, _i64, _vp, _vp]),
    "v2e_render_packet": (_i, [_vp, _i64, _i, _vp, _i, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _d, _d, _d, _d, _d, _vp]),
    "v2e_slomo_fuse": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v2e_slomo_max_speed2": (_i, [_vp, _i, _i, _i, _vp, _vp]),
}

_lib = None


class V2EAmdError(RuntimeError):
    pass


def lib():
    """Load libv2e_amd.so (once) and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise V2EAmdError(
            "HIP extension %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise V2EAmdError("libv2e_amd.so lacks symbol %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc, what=""):
    if rc != 0:
        msg = lib().v2e_last_error()
        raise V2EAmdError("%s failed: rc=%d %s" % (what, rc, (msg or b"").decode()))
