"""ctypes binding of librestir_hip.so (include/restir_hip.h) plus thin Python mirrors of the
reference's host objects (Scene / GBuffer / ReSTIR module state / LeveledEAWFilter).

This is plumbing for tests and bench.py; the product is the shared library.  There is NO fallback:
if the library is missing or no MI355X is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from .ctypes_structs import Camera, Material, Reservoir, MATERIAL_DTYPE, RESERVOIR_DTYPE, INDIRECT_RESERVOIR_DTYPE, copy_camera

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RESTIR_HIP_LIB") or os.path.join(_HERE, "librestir_hip.so")   # env override: A/B builds


class RestirHipError(RuntimeError):
    pass


RS_ERR_INVALID_ARGUMENT, RS_ERR_UNSUPPORTED = 10001, 10002

# rs_transport (include/restir_hip.h): the exchange of the strip driver as callbacks
TRANSPORT_GROUP = C.CFUNCTYPE(C.c_int, C.c_void_p)
TRANSPORT_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
TRANSPORT_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("group_begin", TRANSPORT_GROUP), ("send", TRANSPORT_SEND), ("recv", TRANSPORT_RECV),
                ("group_end", TRANSPORT_GROUP), ("stream_ordered", C.c_int)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("numPrims", C.c_int),
        ("vertices", C.c_void_p),
        ("normals", C.c_void_p),
        ("texcoords", C.c_void_p),
        ("materialIds", C.c_void_p),
        ("numMaterials", C.c_int),
        ("materials", C.c_void_p),
        ("bvhSize", C.c_int),
        ("boundingBoxes", C.c_void_p),
        ("bvhNodes", C.c_void_p * 6),
        ("numLights", C.c_int),
        ("lightPrimIds", C.c_void_p),
        ("lightUnitRadiance", C.c_void_p),
        ("lightProb", C.c_void_p),
        ("lightFailId", C.c_void_p),
        ("sumLightPower", C.c_float),
        ("numTextures", C.c_int),
        ("textures", C.c_void_p),
        ("envMapTexId", C.c_int),
        ("envMapProb", C.c_void_p),
        ("envMapFailId", C.c_void_p),
    ]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.c_void_p)]


class SceneFileView(C.Structure):
    _fields_ = [
        ("numPrims", C.c_int),
        ("vertices", C.c_void_p),
        ("normals", C.c_void_p),
        ("texcoords", C.c_void_p),
        ("materialIds", C.c_void_p),
        ("numMaterials", C.c_int),
        ("materials", C.c_void_p),
        ("numTextures", C.c_int),
        ("textures", C.c_void_p),
        ("envMapTexId", C.c_int),
        ("camera", Camera),
        ("iterations", C.c_int),
        ("traceDepth", C.c_int),
        ("imageName", C.c_char_p),
        ("numSkippedObjects", C.c_int),
    ]


class SVGFView(C.Structure):
    _fields_ = [
        ("devAccumColor", C.c_void_p * 2),
        ("devAccumMoment", C.c_void_p * 2),
        ("devVariance", C.c_void_p),
        ("frameIdx", C.c_int),
        ("width", C.c_int),
        ("height", C.c_int),
    ]


class GBufferView(C.Structure):
    _fields_ = [
        ("devAlbedo", C.c_void_p),
        ("devMotion", C.c_void_p),
        ("devNormal", C.c_void_p * 2),
        ("devPrimId", C.c_void_p * 2),
        ("devDepth", C.c_void_p * 2),
        ("frameIdx", C.c_int),
        ("width", C.c_int),
        ("height", C.c_int),
    ]


# every symbol include/restir_hip.h declares; tests check that the library exports all of them
EXPORTS = [
    "rs_last_error", "rs_context_create", "rs_context_destroy", "rs_context_set_current", "rs_init", "rs_set_stream", "rs_set_sync", "rs_set_side_stream", "rs_set_ris_table_pixels", "rs_set_internal_stream_priority", "rs_internal_streams_info", "rs_choose_internal_streams_again", "rs_prepare_streams", "rs_set_stream_plan", "rs_set_denoise_stream", "rs_join_denoise_stream", "rs_set_tile_split", "rs_synchronize",
    "rs_build_bvh", "rs_build_light_table", "rs_build_alias_table", "rs_build_envmap_pdf", "rs_scene_build", "rs_scene_build_textured", "rs_scene_create",
    "rs_scene_host_desc", "rs_scene_set_sample_sequence", "rs_scene_destroy", "rs_camera_update", "rs_trace_closest", "rs_trace_closest_wave", "rs_scene_set_ordered_tree", "rs_ordered_bvh_host_check", "rs_trace_occlusion",
    "rs_gbuffer_create", "rs_gbuffer_destroy", "rs_gbuffer_render", "rs_gbuffer_render_rows", "rs_gbuffer_update",
    "rs_gbuffer_get_view", "rs_gbuffer_rows_bytes", "rs_gbuffer_rows_pack", "rs_gbuffer_rows_unpack", "rs_restir_init", "rs_restir_free", "rs_restir_reset", "rs_restir_direct",
    "rs_restir_phase_a", "rs_restir_phase_b", "rs_restir_end_frame", "rs_restir_launch_choice", "rs_restir_halo_bytes", "rs_restir_halo_pack",
    "rs_restir_halo_unpack", "rs_restir_rows_bytes", "rs_restir_rows_pack", "rs_restir_rows_unpack", "rs_restir_download", "rs_restir_upload", "rs_restir_ray_count", "rs_restir_ray_total", "rs_restir_pass_times",
    "rs_restir_enable_timing", "rs_restir_spatial_times", "rs_restir_set_probe", "rs_restir_last_launch", "rs_pbo_register", "rs_pbo_map", "rs_pbo_unmap", "rs_pbo_unregister", "rs_save_image", "rs_save_image_jpg", "rs_write_png", "rs_write_jpg", "rs_debug_tap_estimate_error", "rs_debug_sqrt_of_uniform_mismatches", "rs_debug_sqrt_of_unit_floats_mismatches", "rs_debug_exact_ops_mismatches", "rs_debug_div_sigma_mismatches", "rs_path_trace_init", "rs_path_trace_free", "rs_path_trace_direct",
    "rs_path_trace", "rs_path_trace_indirect", "rs_restir_indirect", "rs_restir_download_indirect",
    "rs_svgf_create", "rs_svgf_destroy", "rs_svgf_filter", "rs_svgf_next_frame", "rs_svgf_get_view",
    "rs_copy_image_to_pbo", "rs_copy_image2_to_pbo", "rs_copy_imagef_to_pbo", "rs_copy_imagei_to_pbo", "rs_eaw_create", "rs_eaw_destroy", "rs_eaw_set_params", "rs_eaw_get_params", "rs_eaw_set_tiled", "rs_eaw_set_fused", "rs_svgf_set_params", "rs_svgf_get_params", "rs_svgf_set_tiled", "rs_svgf_set_fused", "rs_eaw_filter", "rs_eaw_positions_rows", "rs_eaw_level_rows", "rs_modulate_albedo",
    "rs_add_image", "rs_add_image3", "rs_comm_create_rccl", "rs_comm_create_rccl_lib", "rs_comm_create", "rs_comm_destroy", "rs_comm_self_exchange", "rs_strips_create", "rs_strips_destroy", "rs_strips_rows", "rs_strips_set_comm_stream", "rs_strips_set_gbuffer_halo", "rs_strips_frame", "rs_strips_eaw_filter", "rs_strips_svgf_filter", "rs_strips_exchange_svgf_history", "rs_strips_exchange_history", "rs_strips_gather", "rs_strips_gather_begin", "rs_strips_gather_end", "rs_strips_enable_timing", "rs_strips_halo_wait_ms",
    "rs_scene_file_load", "rs_scene_file_get", "rs_scene_file_free", "rs_build_transformation_matrix", "rs_bake_instance",
]

_lib = None


def lib():
    """Load librestir_hip.so.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch bundles its own HIP/HSA runtime; it has to be the first one loaded in the process,
        # otherwise two runtimes race for the device and hipGetDeviceCount reports none.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RestirHipError(
            f"{LIB_PATH} is missing: build it with `make -C restir_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the MI355X path.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.rs_last_error.restype = C.c_char_p
    L.rs_init.argtypes = [ci]
    L.rs_context_create.argtypes = [ci, C.POINTER(vp)]
    L.rs_context_destroy.argtypes = [vp]
    L.rs_context_set_current.argtypes = [vp]
    L.rs_set_stream.argtypes = [vp]
    L.rs_set_sync.argtypes = [ci]
    L.rs_set_side_stream.argtypes = [ci]
    L.rs_set_ris_table_pixels.argtypes = [ci]
    L.rs_build_bvh.argtypes = [ci, vp, vp, C.POINTER(vp * 6), C.POINTER(ci)]
    L.rs_build_light_table.argtypes = [ci, vp, vp, ci, vp, C.POINTER(ci), vp, vp, vp]
    L.rs_build_alias_table.argtypes = [ci, vp, vp, vp, C.POINTER(cf)]
    L.rs_scene_build.argtypes = [ci, vp, vp, vp, vp, ci, vp, C.POINTER(vp)]
    L.rs_scene_build_textured.argtypes = [ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, C.POINTER(vp)]
    L.rs_build_envmap_pdf.argtypes = [ci, ci, vp, vp]
    L.rs_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    L.rs_scene_file_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rs_scene_file_get.argtypes = [vp, C.POINTER(SceneFileView)]
    L.rs_scene_file_free.argtypes = [vp]
    L.rs_build_transformation_matrix.argtypes = [vp, vp, vp, vp]
    L.rs_bake_instance.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp]
    L.rs_scene_host_desc.argtypes = [vp, C.POINTER(SceneDesc)]
    L.rs_scene_destroy.argtypes = [vp]
    L.rs_camera_update.argtypes = [C.POINTER(Camera)]
    L.rs_trace_closest.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.rs_trace_occlusion.argtypes = [vp, ci, vp, vp]
    L.rs_trace_closest_wave.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.rs_scene_set_ordered_tree.argtypes = [vp, ci, vp]
    L.rs_gbuffer_create.argtypes = [ci, ci, C.POINTER(vp)]
    L.rs_gbuffer_destroy.argtypes = [vp]
    L.rs_gbuffer_render.argtypes = [vp, vp, C.POINTER(Camera)]
    L.rs_gbuffer_render_rows.argtypes = [vp, vp, C.POINTER(Camera), ci, ci]
    L.rs_gbuffer_update.argtypes = [vp, C.POINTER(Camera)]
    L.rs_gbuffer_get_view.argtypes = [vp, C.POINTER(GBufferView)]
    L.rs_restir_init.argtypes = [ci, ci, C.POINTER(vp)]
    L.rs_restir_free.argtypes = [vp]
    L.rs_restir_reset.argtypes = [vp]
    L.rs_restir_direct.argtypes = [vp, vp, C.POINTER(Camera), vp, vp, ci, ci, ci]
    L.rs_restir_phase_a.argtypes = [vp, vp, C.POINTER(Camera), vp, ci, ci, ci, ci]
    L.rs_restir_phase_b.argtypes = [vp, vp, C.POINTER(Camera), vp, vp, ci, ci, ci, ci]
    L.rs_restir_end_frame.argtypes = [vp]
    L.rs_restir_launch_choice.argtypes = [vp, C.POINTER(ci)]
    L.rs_restir_halo_bytes.argtypes = [vp, ci]
    L.rs_restir_halo_bytes.restype = C.c_size_t
    L.rs_restir_halo_pack.argtypes = [vp, ci, ci, vp]
    L.rs_restir_halo_unpack.argtypes = [vp, ci, ci, vp]
    L.rs_restir_rows_bytes.argtypes = [vp, ci, ci]
    L.rs_restir_rows_bytes.restype = C.c_size_t
    L.rs_debug_tap_estimate_error.argtypes = [ci, C.POINTER(cf)]
    L.rs_restir_last_launch.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.rs_pbo_register.argtypes = [C.c_uint, C.POINTER(vp)]
    L.rs_pbo_map.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.rs_pbo_unmap.argtypes = [vp]
    L.rs_pbo_unregister.argtypes = [vp]
    L.rs_save_image.argtypes = [C.c_char_p, vp, ci, ci, ci]
    L.rs_write_png.argtypes = [C.c_char_p, vp, ci, ci]
    L.rs_write_jpg.argtypes = [C.c_char_p, vp, ci, ci]
    L.rs_svgf_set_tiled.argtypes = [vp, ci]
    L.rs_svgf_set_fused.argtypes = [vp, ci]
    L.rs_save_image_jpg.argtypes = [C.c_char_p, vp, ci, ci, ci]
    L.rs_debug_sqrt_of_uniform_mismatches.argtypes = [C.POINTER(C.c_ulonglong)]
    L.rs_debug_sqrt_of_unit_floats_mismatches.argtypes = [C.POINTER(C.c_ulonglong)]
    L.rs_debug_exact_ops_mismatches.argtypes = [ci, C.c_uint, C.c_uint, ci, ci, C.POINTER(C.c_ulonglong)]
    L.rs_debug_div_sigma_mismatches.argtypes = [C.c_float, C.POINTER(C.c_ulonglong)]
    L.rs_scene_set_sample_sequence.argtypes = [vp, vp, ci, ci]
    L.rs_set_stream_plan.argtypes = [ci, ci, ci]
    L.rs_set_denoise_stream.argtypes = [ci]
    L.rs_join_denoise_stream.argtypes = []
    L.rs_restir_rows_pack.argtypes = [vp, ci, ci, ci, vp]
    L.rs_restir_rows_unpack.argtypes = [vp, ci, ci, ci, vp]
    L.rs_gbuffer_rows_bytes.argtypes = [vp, ci]
    L.rs_gbuffer_rows_bytes.restype = C.c_size_t
    L.rs_gbuffer_rows_pack.argtypes = [vp, ci, ci, ci, vp]
    L.rs_gbuffer_rows_unpack.argtypes = [vp, ci, ci, ci, vp]
    L.rs_restir_download.argtypes = [vp, ci, vp]
    L.rs_restir_upload.argtypes = [vp, ci, vp]
    L.rs_restir_ray_count.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.rs_restir_ray_total.argtypes = [vp, ci, C.POINTER(C.c_ulonglong)]
    L.rs_restir_pass_times.argtypes = [vp, C.POINTER(cf * 4)]
    L.rs_restir_enable_timing.argtypes = [vp, ci]
    L.rs_restir_spatial_times.argtypes = [vp, C.POINTER(cf), ci, C.POINTER(ci)]
    L.rs_restir_set_probe.argtypes = [vp, ci]
    L.rs_path_trace_direct.argtypes = [vp, C.POINTER(Camera), vp, ci, ci, C.POINTER(C.c_ulonglong)]
    L.rs_path_trace.argtypes = [vp, C.POINTER(Camera), vp, vp, ci, ci, ci, C.POINTER(C.c_ulonglong)]
    L.rs_path_trace_indirect.argtypes = [vp, C.POINTER(Camera), vp, ci, ci, ci, C.POINTER(C.c_ulonglong)]
    L.rs_restir_indirect.argtypes = [vp, vp, C.POINTER(Camera), vp, vp, ci, ci, ci, ci, C.POINTER(C.c_ulonglong)]
    L.rs_restir_download_indirect.argtypes = [vp, ci, vp]
    L.rs_svgf_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
    L.rs_svgf_destroy.argtypes = [vp]
    L.rs_svgf_filter.argtypes = [vp, C.POINTER(vp), vp, vp, C.POINTER(Camera)]
    L.rs_svgf_next_frame.argtypes = [vp]
    L.rs_svgf_get_view.argtypes = [vp, C.POINTER(SVGFView)]
    L.rs_copy_image_to_pbo.argtypes = [vp, vp, ci, ci, ci, cf]
    for name in ("rs_copy_image2_to_pbo", "rs_copy_imagef_to_pbo", "rs_copy_imagei_to_pbo"):
        getattr(L, name).argtypes = [vp, vp, ci, ci]
    L.rs_comm_create_rccl.argtypes = [vp, ci, ci, C.POINTER(vp)]
    L.rs_comm_create.argtypes = [C.POINTER(Transport), ci, ci, C.POINTER(vp)]
    L.rs_comm_destroy.argtypes = [vp]
    L.rs_strips_create.argtypes = [vp, ci, ci, C.POINTER(ci), C.POINTER(vp)]
    L.rs_strips_destroy.argtypes = [vp]
    L.rs_strips_rows.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.rs_strips_set_comm_stream.argtypes = [vp, ci]
    L.rs_strips_set_gbuffer_halo.argtypes = [vp, ci]
    L.rs_strips_frame.argtypes = [vp, vp, vp, C.POINTER(Camera), vp, vp, ci, ci, ci]
    L.rs_strips_eaw_filter.argtypes = [vp, vp, vp, C.POINTER(Camera), vp, C.POINTER(vp)]
    L.rs_strips_exchange_history.argtypes = [vp, vp, vp]
    L.rs_strips_svgf_filter.argtypes = [vp, vp, vp, C.POINTER(Camera), vp, C.POINTER(vp)]
    L.rs_strips_exchange_svgf_history.argtypes = [vp, vp]
    L.rs_strips_gather.argtypes = [vp, vp, C.c_size_t, ci]
    L.rs_strips_gather_begin.argtypes = [vp, vp, C.c_size_t, ci, ci]
    L.rs_strips_gather_end.argtypes = [vp, ci]
    L.rs_strips_enable_timing.argtypes = [vp, ci]
    L.rs_strips_halo_wait_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rs_comm_create_rccl_lib.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(vp)]
    L.rs_eaw_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
    for name in ("rs_eaw_set_params", "rs_svgf_set_params"):
        getattr(L, name).argtypes = [vp, C.c_float, C.c_float, C.c_float, ci]
    for name in ("rs_eaw_get_params", "rs_svgf_get_params"):
        getattr(L, name).argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(ci)]
    L.rs_eaw_set_tiled.argtypes = [vp, ci]
    L.rs_eaw_set_fused.argtypes = [vp, ci]
    L.rs_eaw_destroy.argtypes = [vp]
    L.rs_eaw_filter.argtypes = [vp, C.POINTER(vp), vp, vp, C.POINTER(Camera)]
    L.rs_eaw_positions_rows.argtypes = [vp, vp, C.POINTER(Camera), ci, ci]
    L.rs_eaw_level_rows.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.rs_modulate_albedo.argtypes = [vp, vp]
    L.rs_add_image.argtypes = [vp, vp, ci, ci]
    L.rs_add_image3.argtypes = [vp, vp, vp, ci, ci]
    _lib = L
    return L


def check(code):
    if code != 0:
        raise RestirHipError(f"librestir_hip error {code}: {lib().rs_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def init(device=0):
    check(lib().rs_init(device))


class Context:
    """rs_context: device, stream, launch mode and internal streams of the objects created while it is current (per thread)."""

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        check(lib().rs_context_create(device, C.byref(self.handle)))

    def make_current(self):
        check(lib().rs_context_set_current(self.handle))

    @staticmethod
    def use_default():
        check(lib().rs_context_set_current(None))

    def destroy(self):
        if self.handle:
            lib().rs_context_destroy(self.handle)
            self.handle = C.c_void_p()


def set_sync(sync):
    check(lib().rs_set_sync(1 if sync else 0))


def set_stream(hip_stream):
    """The stream the library enqueues on (rs_set_stream); 0 / None = the legacy default stream."""
    check(lib().rs_set_stream(C.c_void_p(int(hip_stream or 0))))


def set_side_stream(enable):
    """Overlapped frames when launches are asynchronous (include/restir_hip.h): 0 off, 1 on with GBuffer::render as its own launch,
    2 / 3 on with the render deferred into ReSTIRDirect's primary-ray launch (3: at any size), 4 on with that choice measured."""
    check(lib().rs_set_side_stream(int(enable)))


def set_internal_stream_priority(level):
    """-1 high / 0 normal / 1 low / 2 automatic: the level of the library's own streams (before the first frame)."""
    check(lib().rs_set_internal_stream_priority(int(level)))


def internal_streams_info():
    """(level, calibration us of the chosen streams, fastest candidate us) of the library's own streams; (level, 0, 0): plain streams."""
    p, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
    check(lib().rs_internal_streams_info(C.byref(p), C.byref(a), C.byref(b)))
    return p.value, a.value, b.value


def choose_internal_streams_again():
    check(lib().rs_choose_internal_streams_again())


def set_ris_table_pixels(pixels):
    """Launches of fewer pixels read the RIS light table from global memory instead of LDS (rs_set_ris_table_pixels); 0 = always LDS."""
    check(lib().rs_set_ris_table_pixels(int(pixels)))


def prepare_streams():
    """Choose the library's own streams now rather than inside the first overlapped frame (rs_prepare_streams)."""
    check(lib().rs_prepare_streams())


def set_stream_plan(chain_streams=-1, small_chains=-1, shadow_on_main=-1):
    """How the overlapped mode spreads a frame's kernels over the internal streams (rs_set_stream_plan); -1 keeps a value."""
    check(lib().rs_set_stream_plan(int(chain_streams), int(small_chains), int(shadow_on_main)))


def set_denoise_stream(enable):
    """LeveledEAWFilter and the tone map of its result on a stream of the library, next to the next frame's passes (rs_set_denoise_stream);
    their outputs are then ordered by events: join_denoise_stream() or synchronize() before reading them outside the library."""
    check(lib().rs_set_denoise_stream(int(enable)))


def join_denoise_stream():
    """The library stream waits (on the device) for everything enqueued on the denoise stream so far (rs_join_denoise_stream)."""
    check(lib().rs_join_denoise_stream())


def set_tile_split(threshold):
    """Closest-hit kernels: tiles whose last walk visited at least `threshold` nodes are traced by four waves (rs_set_tile_split); 0 = off."""
    check(lib().rs_set_tile_split(int(threshold)))


def write_jpg(path, rgb):
    """Image::saveJPG's file (stbi_write_jpg, quality 90) from an (H, W, 3) uint8 array."""
    a = np.ascontiguousarray(rgb, np.uint8)
    check(lib().rs_write_jpg(os.fsencode(path), _p(a), a.shape[1], a.shape[0]))


def save_image_jpg(path, dev_image_ptr, width, height, tone_mapping):
    """saveImage(true) (src/main.cpp:105-144): tone map + gamma, mirrored in x, JPEG at quality 90."""
    check(lib().rs_save_image_jpg(os.fsencode(path), dev_image_ptr, width, height, tone_mapping))


def save_image(path, dev_image_ptr, width, height, tone_mapping):
    """saveImage(false) (src/main.cpp:105-144): tone map + gamma, mirrored in x, 8-bit RGB PNG."""
    check(lib().rs_save_image(str(path).encode(), dev_image_ptr, width, height, tone_mapping))


def write_png(path, rgb):
    """rgb: uint8 numpy array (height, width, 3)."""
    import numpy as np
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    check(lib().rs_write_png(str(path).encode(), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))


def synchronize():
    check(lib().rs_synchronize())


def camera_update(cam):
    check(lib().rs_camera_update(C.byref(cam)))
    return cam


# ---- host builders (no GPU needed) ----------------------------------------------------------------
def build_bvh(vertices):
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1)
    n = v.size // 9
    size = 2 * n - 1
    boxes = np.zeros((size, 6), np.float32)
    nodes = np.zeros((6, size, 3), np.int32)
    arr = (C.c_void_p * 6)(*[nodes[i].ctypes.data for i in range(6)])
    got = C.c_int(0)
    check(lib().rs_build_bvh(n, _p(v), _p(boxes), C.byref(arr), C.byref(got)))
    assert got.value == size
    return boxes, nodes


def ordered_bvh_host_check(boxes, nodes):
    """rs_ordered_bvh_host_check on rs_build_bvh's outputs; returns (error code, node counts per axis, depth per axis)."""
    boxes = np.ascontiguousarray(boxes, np.float32); nodes = [np.ascontiguousarray(nodes[k], np.int32) for k in range(6)]
    size = boxes.shape[0]
    arr = (C.c_void_p * 6)(*[nodes[i].ctypes.data for i in range(6)])
    counts = (C.c_int * 3)(); depth = (C.c_int * 3)()
    fn = lib().rs_ordered_bvh_host_check
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p * 6), C.POINTER(C.c_int * 3), C.POINTER(C.c_int * 3)]
    e = fn((size + 1) // 2, size, _p(boxes), C.byref(arr), C.byref(counts), C.byref(depth))
    return e, list(counts), list(depth)


def build_alias_table(values):
    values = np.ascontiguousarray(values, np.float32)
    n = values.size
    prob = np.zeros(n, np.float32); fail = np.zeros(n, np.int32); s = C.c_float(0)
    check(lib().rs_build_alias_table(n, _p(values), _p(prob), _p(fail), C.byref(s)))
    return prob, fail, np.float32(s.value)


def build_light_table(vertices, material_ids, materials):
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1)
    n = v.size // 9
    mats = np.ascontiguousarray(materials, MATERIAL_DTYPE)
    ids = np.zeros(n, np.int32); rad = np.zeros((n, 3), np.float32); power = np.zeros(n, np.float32)
    k = C.c_int(0)
    check(lib().rs_build_light_table(n, _p(v), _p(np.ascontiguousarray(material_ids, np.int32)), len(mats), _p(mats),
                                     C.byref(k), _p(ids), _p(rad), _p(power)))
    return ids[:k.value].copy(), rad[:k.value].copy(), power[:k.value].copy()


def build_envmap_pdf(env):
    """Scene::createLightSampler's environment-map pdf (src/scene.cpp:139-146) for an (H, W, 3) float32 map."""
    env = np.ascontiguousarray(env, np.float32)
    pdf = np.zeros(env.shape[0] * env.shape[1], np.float32)
    check(lib().rs_build_envmap_pdf(env.shape[1], env.shape[0], _p(env), _p(pdf)))
    return pdf


def build_transformation_matrix(translation, rotation, scale):
    """Math::buildTransformationMatrix (src/mathUtil.cpp:13-20): (4, 4) float32, [column][row]."""
    t, r, sc = (np.ascontiguousarray(a, np.float32) for a in (translation, rotation, scale))
    out = np.zeros((4, 4), np.float32)
    check(lib().rs_build_transformation_matrix(_p(t), _p(r), _p(sc), _p(out)))
    return out


def bake_instance(translation, rotation, scale, vertices, normals):
    """Instance baking of Scene::buildDevData (src/scene.cpp:167-168) for (n, 3) vertices / normals."""
    t, r, sc = (np.ascontiguousarray(a, np.float32) for a in (translation, rotation, scale))
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    vo, no = np.zeros_like(v), np.zeros_like(n)
    check(lib().rs_bake_instance(_p(t), _p(r), _p(sc), len(v), _p(v), _p(n), _p(vo), _p(no)))
    return vo, no


class SceneFile:
    """Scene::Scene(filename) (src/scene.cpp:96-131): the parsed scene as flat host arrays (copies)."""

    def __init__(self, path):
        h = C.c_void_p()
        check(lib().rs_scene_file_load(os.fsencode(path), C.byref(h)))
        try:
            v = SceneFileView()
            check(lib().rs_scene_file_get(h, C.byref(v)))
            n = v.numPrims

            def arr(ptr, count, dtype):
                return np.frombuffer(C.string_at(ptr, count * np.dtype(dtype).itemsize), dtype=dtype).copy() if count else np.zeros(0, dtype)
            self.vertices = arr(v.vertices, n * 9, np.float32).reshape(n, 3, 3)
            self.normals = arr(v.normals, n * 9, np.float32).reshape(n, 3, 3)
            self.texcoords = arr(v.texcoords, n * 6, np.float32).reshape(n, 3, 2)
            self.material_ids = arr(v.materialIds, n, np.int32)
            self.materials = arr(v.materials, v.numMaterials, MATERIAL_DTYPE)
            tex = C.cast(v.textures, C.POINTER(Texture))
            self.textures = [arr(tex[i].data, tex[i].width * tex[i].height * 3, np.float32).reshape(tex[i].height, tex[i].width, 3)
                             for i in range(v.numTextures)]
            self.env_map_tex = v.envMapTexId
            self.camera = copy_camera(v.camera)
            self.iterations, self.trace_depth = v.iterations, v.traceDepth
            self.image_name = (v.imageName or b"").decode()
            self.num_skipped_objects = v.numSkippedObjects
        finally:
            lib().rs_scene_file_free(h)

    def build(self):
        """Scene::buildDevData + DevScene::create of the parsed scene."""
        return Scene(self.vertices, self.normals, self.texcoords, self.material_ids, self.materials,
                     textures=self.textures, env_map_tex=self.env_map_tex)


def _texture_table(textures):
    keep = [np.ascontiguousarray(t, np.float32) for t in textures]
    arr = (Texture * max(1, len(keep)))()
    for i, t in enumerate(keep):
        assert t.ndim == 3 and t.shape[2] == 3, "textures are (H, W, 3) float32"
        arr[i].height, arr[i].width, arr[i].data = t.shape[0], t.shape[1], t.ctypes.data
    return keep, arr


# ---- device objects ---------------------------------------------------------------------------------
class Scene:
    """Scene::buildDevData + DevScene (src/scene.cpp:159-215): builds light table, alias table and the
    MTBVH on the host (C++ in the library) and uploads the CDNA4 layout."""

    def __init__(self, vertices, normals, texcoords, material_ids, materials, textures=(), env_map_tex=-1):
        """textures: (H, W, 3) float32 linear-RGB arrays (what Image holds, src/image.h); env_map_tex: index or -1."""
        self._keep = (np.ascontiguousarray(vertices, np.float32), np.ascontiguousarray(normals, np.float32),
                      np.ascontiguousarray(texcoords, np.float32), np.ascontiguousarray(material_ids, np.int32),
                      np.ascontiguousarray(materials, MATERIAL_DTYPE))
        v, n, t, m, mats = self._keep
        self.num_prims = v.size // 9
        self.handle = C.c_void_p()
        self._tex, tex = _texture_table(textures)
        check(lib().rs_scene_build_textured(self.num_prims, _p(v), _p(n), _p(t), _p(m), len(mats), _p(mats),
                                            len(self._tex), C.cast(tex, C.c_void_p), int(env_map_tex), C.byref(self.handle)))

    @classmethod
    def from_tables(cls, vertices, normals, texcoords, material_ids, materials, tables, textures=(), env_map_tex=-1, env_sampler=None):
        """DevScene::create from caller-built tables (rs_scene_create): `tables` as returned by host_desc();
        env_sampler = (prob, failId) of the environment map's alias table when env_map_tex >= 0."""
        self = cls.__new__(cls)
        t = tables
        self._tex, tex = _texture_table(textures)
        self._keep = (np.ascontiguousarray(vertices, np.float32), np.ascontiguousarray(normals, np.float32),
                      np.ascontiguousarray(texcoords, np.float32), np.ascontiguousarray(material_ids, np.int32),
                      np.ascontiguousarray(materials, MATERIAL_DTYPE),
                      np.ascontiguousarray(t["boxes"], np.float32), [np.ascontiguousarray(t["nodes"][k], np.int32) for k in range(6)],
                      np.ascontiguousarray(t["light_prim_ids"], np.int32), np.ascontiguousarray(t["light_radiance"], np.float32),
                      np.ascontiguousarray(t["light_prob"], np.float32), np.ascontiguousarray(t["light_fail"], np.int32))
        v, n, tc, m, mats, boxes, nodes, lp, lr, lpr, lf = self._keep
        self.num_prims = v.size // 9
        d = SceneDesc()
        d.numPrims = self.num_prims
        d.vertices, d.normals, d.texcoords, d.materialIds = _p(v), _p(n), _p(tc), _p(m)
        d.numMaterials, d.materials = len(mats), _p(mats)
        d.bvhSize, d.boundingBoxes = len(boxes), _p(boxes)
        for k in range(6):
            d.bvhNodes[k] = _p(nodes[k])
        d.numLights = len(lpr)                      # sampler length (light primitives + 1 with an environment map)
        d.lightPrimIds, d.lightUnitRadiance, d.lightProb, d.lightFailId = _p(lp), _p(lr), _p(lpr), _p(lf)
        d.sumLightPower = float(t["sum_power"])
        d.numTextures = len(self._tex); d.textures = C.cast(tex, C.c_void_p); d.envMapTexId = int(env_map_tex)
        if env_sampler is not None:
            self._env = (np.ascontiguousarray(env_sampler[0], np.float32), np.ascontiguousarray(env_sampler[1], np.int32))
            d.envMapProb, d.envMapFailId = _p(self._env[0]), _p(self._env[1])
        self.handle = C.c_void_p()
        check(lib().rs_scene_create(C.byref(d), C.byref(self.handle)))
        return self

    def set_sample_sequence(self, table):
        """DevScene::sampleSequence: uint32 [numSamples, 200] selects the Sobol sampler (src/sampler.h:9-36), None the default engine."""
        if table is None:
            check(lib().rs_scene_set_sample_sequence(self.handle, None, 0, 0))
            return
        t = np.ascontiguousarray(table, np.uint32)
        check(lib().rs_scene_set_sample_sequence(self.handle, _p(t), t.shape[0], t.shape[1]))

    def host_desc(self):
        """numpy views of the arrays the scene was built from (for parity checks of the host build)."""
        d = SceneDesc()
        check(lib().rs_scene_host_desc(self.handle, C.byref(d)))
        n, s, l = d.numPrims, d.bvhSize, d.numLights
        lp = l - (1 if d.envMapTexId >= 0 else 0)       # light primitives; the environment map is the last sampler entry
        ne = 0
        if d.envMapTexId >= 0:
            t = C.cast(d.textures, C.POINTER(Texture))[d.envMapTexId]
            ne = t.width * t.height

        def arr(ptr, ctype, count, shape):
            if count == 0:
                return np.zeros(shape, ctype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(ctype))), (count,)).reshape(shape).copy()

        return dict(
            boxes=arr(d.boundingBoxes, np.float32, s * 6, (s, 6)),
            nodes=np.stack([arr(d.bvhNodes[k], np.int32, s * 3, (s, 3)) for k in range(6)]),
            light_prim_ids=arr(d.lightPrimIds, np.int32, lp, (lp,)),
            light_radiance=arr(d.lightUnitRadiance, np.float32, lp * 3, (lp, 3)),
            light_prob=arr(d.lightProb, np.float32, l, (l,)),
            light_fail=arr(d.lightFailId, np.int32, l, (l,)),
            sum_power=np.float32(d.sumLightPower), num_lights=l, bvh_size=s, num_prims=n,
            env_map_tex=d.envMapTexId,
            env_prob=arr(d.envMapProb, np.float32, ne, (ne,)), env_fail=arr(d.envMapFailId, np.int32, ne, (ne,)),
        )

    def destroy(self):
        if self.handle:
            lib().rs_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GBuffer:
    def __init__(self, width, height):
        self.width, self.height = width, height
        self.handle = C.c_void_p()
        check(lib().rs_gbuffer_create(width, height, C.byref(self.handle)))

    def render(self, scene, cam, y0=None, y1=None):
        if y0 is None:
            check(lib().rs_gbuffer_render(self.handle, scene.handle, C.byref(cam)))
        else:
            check(lib().rs_gbuffer_render_rows(self.handle, scene.handle, C.byref(cam), y0, y1))

    def update(self, cam):
        check(lib().rs_gbuffer_update(self.handle, C.byref(cam)))

    def rows_bytes(self, rows):
        return int(lib().rs_gbuffer_rows_bytes(self.handle, rows))

    def rows_pack(self, sel, y0, rows, dev_ptr):
        check(lib().rs_gbuffer_rows_pack(self.handle, sel, y0, rows, dev_ptr))

    def rows_unpack(self, sel, y0, rows, dev_ptr):
        check(lib().rs_gbuffer_rows_unpack(self.handle, sel, y0, rows, dev_ptr))

    def view(self):
        v = GBufferView()
        check(lib().rs_gbuffer_get_view(self.handle, C.byref(v)))
        return v

    def download(self):
        """Copies every plane to host numpy arrays (torch is only used for the D2H copy)."""
        import torch
        v = self.view()
        n = self.width * self.height

        def grab(ptr, count, dtype):
            t = torch.empty(count, dtype=dtype, device="cuda")
            hip_memcpy_d2d(t.data_ptr(), ptr, count * t.element_size())
            return t.cpu().numpy()

        return dict(
            albedo=grab(v.devAlbedo, n * 3, torch.float32).reshape(n, 3),
            motion=grab(v.devMotion, n, torch.int32),
            normal=[grab(v.devNormal[i], n * 3, torch.float32).reshape(n, 3) for i in range(2)],
            prim_id=[grab(v.devPrimId[i], n, torch.int32) for i in range(2)],
            depth=[grab(v.devDepth[i], n, torch.float32) for i in range(2)],
            frame_idx=v.frameIdx,
        )

    def destroy(self):
        if self.handle:
            lib().rs_gbuffer_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
    return _hip


def hip_malloc(nbytes):
    p = C.c_void_p()
    e = _hiprt().hipMalloc(C.byref(p), nbytes)
    if e != 0:
        raise RestirHipError(f"hipMalloc failed: {e}")
    return p.value


def hip_free(ptr):
    _hiprt().hipFree(ptr)


def hip_memcpy_d2d(dst, src, nbytes):
    """Device-to-device copy that has finished when the call returns, ordered after everything the library has enqueued.  (hipMemcpy of
    device memory goes through the null stream and need not block the host; a caller whose own work runs on a NON-BLOCKING stream -- torch's
    torch.cuda.Stream() -- is not ordered with the null stream either, so the copy is waited for here.  bench.py's strips-against-full-frame
    check found the missing wait: the gloo rehearsal transport read a staging buffer before this copy had written it.)"""
    _hiprt()
    synchronize()
    e = _hip.hipMemcpy(dst, src, nbytes, 3)   # hipMemcpyDeviceToDevice
    if e == 0:
        e = _hip.hipStreamSynchronize(None)
    if e != 0:
        raise RestirHipError(f"hipMemcpy failed: {e}")


def hip_memcpy_d2d_async(dst, src, nbytes):
    """Device-to-device copy enqueued on the default stream (the library's stream unless rs_set_stream changed it)."""
    _hiprt()
    e = _hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3, None)
    if e != 0:
        raise RestirHipError(f"hipMemcpyAsync failed: {e}")


class ReSTIR:
    """restir.cu module state + ReSTIRInit/Free/Reset/Direct."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.handle = C.c_void_p()
        check(lib().rs_restir_init(width, height, C.byref(self.handle)))

    def reset(self):
        check(lib().rs_restir_reset(self.handle))

    def direct(self, scene, cam, gbuf, dev_direct_illum_ptr, iter_, looper, reuse):
        check(lib().rs_restir_direct(self.handle, scene.handle, C.byref(cam), gbuf.handle, dev_direct_illum_ptr, iter_, looper, reuse))

    def indirect(self, scene, cam, gbuf, dev_indirect_illum_ptr, iter_, looper, reuse, max_depth):
        """ReSTIRIndirect (src/restir.cu:448-476); returns the number of BVH walks."""
        n = C.c_ulonglong(0)
        check(lib().rs_restir_indirect(self.handle, scene.handle, C.byref(cam), gbuf.handle, dev_indirect_illum_ptr, iter_, looper, reuse, max_depth, C.byref(n)))
        return n.value

    def download_indirect(self, which):
        out = np.zeros(self.width * self.height, INDIRECT_RESERVOIR_DTYPE)
        check(lib().rs_restir_download_indirect(self.handle, which, _p(out)))
        return out

    def phase_a(self, scene, cam, gbuf, looper, reuse, y0, y1):
        check(lib().rs_restir_phase_a(self.handle, scene.handle, C.byref(cam), gbuf.handle, looper, reuse, y0, y1))

    def phase_b(self, scene, cam, gbuf, dev_direct_illum_ptr, iter_, reuse, y0, y1):
        check(lib().rs_restir_phase_b(self.handle, scene.handle, C.byref(cam), gbuf.handle, dev_direct_illum_ptr, iter_, reuse, y0, y1))

    def end_frame(self):
        check(lib().rs_restir_end_frame(self.handle))

    def last_launch(self):
        """(fused, chains) of the last phase-A launch: render in the primary rays' launch? chain streams in turn (0: synchronous)."""
        a, b = C.c_int(), C.c_int()
        check(lib().rs_restir_last_launch(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def launch_choice(self):
        """0 two launches, 1 GBuffer::render walked together with the primary rays, -1 still measuring, -2 nothing to choose."""
        c = C.c_int(-1)
        check(lib().rs_restir_launch_choice(self.handle, C.byref(c)))
        return c.value

    def halo_bytes(self, rows):
        return int(lib().rs_restir_halo_bytes(self.handle, rows))

    def halo_pack(self, y0, rows, dev_ptr):
        check(lib().rs_restir_halo_pack(self.handle, y0, rows, dev_ptr))

    def halo_unpack(self, y0, rows, dev_ptr):
        check(lib().rs_restir_halo_unpack(self.handle, y0, rows, dev_ptr))

    def rows_bytes(self, which, rows):
        return int(lib().rs_restir_rows_bytes(self.handle, which, rows))

    def rows_pack(self, which, y0, rows, dev_ptr):
        check(lib().rs_restir_rows_pack(self.handle, which, y0, rows, dev_ptr))

    def rows_unpack(self, which, y0, rows, dev_ptr):
        check(lib().rs_restir_rows_unpack(self.handle, which, y0, rows, dev_ptr))

    def download(self, which):
        out = np.zeros(self.width * self.height, RESERVOIR_DTYPE)
        check(lib().rs_restir_download(self.handle, which, _p(out)))
        return out

    def upload(self, which, arr):
        arr = np.ascontiguousarray(arr, RESERVOIR_DTYPE)
        check(lib().rs_restir_upload(self.handle, which, _p(arr)))

    def ray_count(self):
        n = C.c_ulonglong(0)
        check(lib().rs_restir_ray_count(self.handle, C.byref(n)))
        return n.value

    def ray_total(self, frames):
        n = C.c_ulonglong(0)
        check(lib().rs_restir_ray_total(self.handle, frames, C.byref(n)))
        return n.value

    def enable_timing(self, on=True):
        """True / 1: every pass bracketed, kernels on the library stream; 2: the spatial pass only, launches as in the overlapped mode."""
        check(lib().rs_restir_enable_timing(self.handle, 2 if on == 2 else (1 if on else 0)))

    def pass_times(self):
        ms = (C.c_float * 4)()
        check(lib().rs_restir_pass_times(self.handle, C.byref(ms)))
        return [float(x) for x in ms]

    def spatial_times(self, capacity=256):
        """enable_timing(2): the spatial pass's duration (ms) in each of the last frames, oldest first (rs_restir_spatial_times)."""
        ms = (C.c_float * capacity)()
        n = C.c_int(0)
        check(lib().rs_restir_spatial_times(self.handle, ms, capacity, C.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    def set_probe(self, on):
        """The spatial pass under the name k_spatial_shade_probe (a measurement's own launches; rs_restir_set_probe)."""
        check(lib().rs_restir_set_probe(self.handle, 1 if on else 0))

    def destroy(self):
        if self.handle:
            lib().rs_restir_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Comm:
    """rs_comm: the exchange of the strip driver.  Comm.rccl(nccl_comm_ptr, rank, world) wraps an ncclComm_t; Comm(callbacks...)
    takes host callbacks send(dev_ptr, nbytes, peer) / recv(dev_ptr, nbytes, peer) / group_begin() / group_end() (tests)."""

    def __init__(self, rank, world, send, recv, group_begin=None, group_end=None, stream_ordered=False):
        def guard(fn):
            def call(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:                      # an exception must not cross the C frame
                    print("transport callback failed:", repr(e), flush=True)
                    return RS_ERR_UNSUPPORTED
            return call
        self._cbs = (TRANSPORT_GROUP(guard(lambda ctx: group_begin() if group_begin else None)),
                     TRANSPORT_SEND(guard(lambda ctx, p, n, peer, st: send(p, n, peer))),
                     TRANSPORT_RECV(guard(lambda ctx, p, n, peer, st: recv(p, n, peer))),
                     TRANSPORT_GROUP(guard(lambda ctx: group_end() if group_end else None)))
        t = Transport(None, self._cbs[0], self._cbs[1], self._cbs[2], self._cbs[3], 1 if stream_ordered else 0)
        self.handle = C.c_void_p()
        check(lib().rs_comm_create(C.byref(t), rank, world, C.byref(self.handle)))

    @classmethod
    def rccl(cls, nccl_comm_ptr, rank, world, librccl_path=None):
        """librccl_path: the copy of RCCL that created the communicator (None: the library searches, rs_comm_create_rccl)."""
        self = cls.__new__(cls)
        self._cbs = ()
        self.handle = C.c_void_p()
        check(lib().rs_comm_create_rccl_lib(C.c_void_p(nccl_comm_ptr), rank, world, os.fsencode(librccl_path) if librccl_path else None, C.byref(self.handle)))
        return self

    def destroy(self):
        if self.handle:
            lib().rs_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class Strips:
    """rs_strips: the row-strip frame of one rank (include/restir_hip.h), the C form of tiling.StripRenderer."""

    def __init__(self, comm, width, height, bounds=None):
        self.comm = comm
        self.handle = C.c_void_p()
        b = (C.c_int * len(bounds))(*bounds) if bounds is not None else None
        check(lib().rs_strips_create(comm.handle, width, height, b, C.byref(self.handle)))
        y0, y1 = C.c_int(), C.c_int()
        check(lib().rs_strips_rows(self.handle, C.byref(y0), C.byref(y1)))
        self.y0, self.y1 = y0.value, y1.value

    def set_comm_stream(self, own_stream):
        """Stream-ordered transports: transfers on the library stream (False, default) or on a stream of the driver (True)."""
        check(lib().rs_strips_set_comm_stream(self.handle, 1 if own_stream else 0))

    def set_gbuffer_halo(self, rows):
        """G-buffer rows that travel with the reservoir rows of a frame: 5 (default), or 32 when a denoiser follows (rs_strips_set_gbuffer_halo)."""
        check(lib().rs_strips_set_gbuffer_halo(self.handle, int(rows)))

    def frame(self, restir, scene, cam, gbuf, dev_direct_illum_ptr, iter_, looper, reuse):
        check(lib().rs_strips_frame(self.handle, restir.handle, scene.handle, C.byref(cam), gbuf.handle, dev_direct_illum_ptr, iter_, looper, reuse))

    def eaw_filter(self, eaw, gbuf, cam, dev_color_ptr):
        """LeveledEAWFilter on the strip (before GBuffer.update); returns the device pointer of the driver's result buffer."""
        p = C.c_void_p()
        check(lib().rs_strips_eaw_filter(self.handle, eaw.handle, gbuf.handle, C.byref(cam), dev_color_ptr, C.byref(p)))
        return p.value

    def svgf_filter(self, svgf, gbuf, cam, dev_color_ptr):
        """SpatioTemporalFilter on the strip (before GBuffer.update); `svgf` is a capi.SVGFFilter, whose out pointer is handed over as
        by its own filter(); returns the device pointer of the filtered image (rows of this strip valid)."""
        p = C.c_void_p(svgf.out_ptr)
        check(lib().rs_strips_svgf_filter(self.handle, svgf.handle, gbuf.handle, C.byref(cam), dev_color_ptr, C.byref(p)))
        svgf.out_ptr = p.value
        return svgf.out_ptr

    def exchange_svgf_history(self, svgf):
        """Moving camera: after svgf_filter and before svgf.next_frame(), the filter's history rows travel to every other rank."""
        check(lib().rs_strips_exchange_svgf_history(self.handle, svgf.handle))

    def exchange_history(self, restir, gbuf):
        """Moving camera: after GBuffer.update, every rank's history rows travel to every other rank."""
        check(lib().rs_strips_exchange_history(self.handle, restir.handle, gbuf.handle))

    def gather(self, dev_image_ptr, bytes_per_pixel, root=-1):
        check(lib().rs_strips_gather(self.handle, dev_image_ptr, bytes_per_pixel, root))

    def gather_begin(self, dev_image_ptr, bytes_per_pixel, root, slot):
        check(lib().rs_strips_gather_begin(self.handle, dev_image_ptr, bytes_per_pixel, root, slot))

    def gather_end(self, slot):
        check(lib().rs_strips_gather_end(self.handle, slot))

    def enable_timing(self, enable=True):
        check(lib().rs_strips_enable_timing(self.handle, 1 if enable else 0))

    def halo_wait_ms(self):
        ms = C.c_float(0)
        check(lib().rs_strips_halo_wait_ms(self.handle, C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.handle:
            lib().rs_strips_destroy(self.handle)
            self.handle = C.c_void_p()


class EAWFilter:
    """LeveledEAWFilter (src/denoiser.h:33-43)."""

    def __init__(self, width, height, level=5):
        self.handle = C.c_void_p()
        check(lib().rs_eaw_create(width, height, level, C.byref(self.handle)))

    def filter(self, out_ptr, in_ptr, gbuf, cam):
        """Returns the device pointer that holds the result (the reference swaps the caller's pointer)."""
        p = C.c_void_p(out_ptr)
        check(lib().rs_eaw_filter(self.handle, C.byref(p), in_ptr, gbuf.handle, C.byref(cam)))
        return p.value

    def set_params(self, sig_lumin, sig_normal, sig_depth, level=5):
        """waveletFilter.sigLumin / sigNormal / sigDepth and level, the members the viewer edits (src/preview.cpp:262-265)."""
        check(lib().rs_eaw_set_params(self.handle, sig_lumin, sig_normal, sig_depth, level))

    def set_tiled(self, tiled):
        """Levels of step 1, 2, 4 from an LDS tile (default) or as plain gathers; same bits."""
        check(lib().rs_eaw_set_tiled(self.handle, int(bool(tiled))))

    def set_fused(self, fused):
        """Taps in fused arithmetic (1, the library's default) or every operation rounded separately in the reference's order (0)."""
        check(lib().rs_eaw_set_fused(self.handle, int(bool(fused))))

    def get_params(self):
        a, b, c, lv = C.c_float(), C.c_float(), C.c_float(), C.c_int()
        check(lib().rs_eaw_get_params(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(lv)))
        return a.value, b.value, c.value, lv.value

    def positions_rows(self, gbuf, cam, y0, y1):
        check(lib().rs_eaw_positions_rows(self.handle, gbuf.handle, C.byref(cam), y0, y1))

    def level_rows(self, out_ptr, in_ptr, gbuf, level, y0, y1):
        check(lib().rs_eaw_level_rows(self.handle, out_ptr, in_ptr, gbuf.handle, level, y0, y1))

    def destroy(self):
        if self.handle:
            lib().rs_eaw_destroy(self.handle)
            self.handle = C.c_void_p()


class SVGFFilter:
    """SpatioTemporalFilter (src/denoiser.h:45-70).  The caller-side `devColorOut` buffer of the reference is kept
    here: it comes from hipMalloc and is swapped with the filter's buffers by every call, as in the reference."""

    def __init__(self, width, height, level=5):
        self.n = width * height
        self.handle = C.c_void_p()
        check(lib().rs_svgf_create(width, height, level, C.byref(self.handle)))
        self.out_ptr = hip_malloc(self.n * 12)

    def filter(self, in_ptr, gbuf, cam):
        """Returns the device pointer that holds the filtered image."""
        p = C.c_void_p(self.out_ptr)
        check(lib().rs_svgf_filter(self.handle, C.byref(p), in_ptr, gbuf.handle, C.byref(cam)))
        self.out_ptr = p.value
        return self.out_ptr

    def set_params(self, sig_lumin, sig_normal, sig_depth, level=5):
        """waveletFilter.sig* and level of SpatioTemporalFilter (src/preview.cpp:278-286)."""
        check(lib().rs_svgf_set_params(self.handle, sig_lumin, sig_normal, sig_depth, level))

    def get_params(self):
        a, b, c, lv = C.c_float(), C.c_float(), C.c_float(), C.c_int()
        check(lib().rs_svgf_get_params(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(lv)))
        return a.value, b.value, c.value, lv.value

    def set_tiled(self, tiled):
        """a-trous levels from the LDS tile (True, default) or as plain gathers (False): same bits."""
        check(lib().rs_svgf_set_tiled(self.handle, 1 if tiled else 0))

    def set_fused(self, fused):
        """Taps in fused arithmetic (True) or every operation rounded separately in the reference's order (False)."""
        check(lib().rs_svgf_set_fused(self.handle, 1 if fused else 0))

    def next_frame(self):
        check(lib().rs_svgf_next_frame(self.handle))

    def view(self):
        v = SVGFView()
        check(lib().rs_svgf_get_view(self.handle, C.byref(v)))
        return v

    def destroy(self):
        if self.handle:
            lib().rs_svgf_destroy(self.handle)
            hip_free(self.out_ptr)
            self.handle = C.c_void_p()


def path_trace(scene, cam, dev_direct_ptr, dev_indirect_ptr, iter_, looper, max_depth):
    n = C.c_ulonglong(0)
    check(lib().rs_path_trace(scene.handle, C.byref(cam), dev_direct_ptr, dev_indirect_ptr, iter_, looper, max_depth, C.byref(n)))
    return n.value


def path_trace_indirect(scene, cam, dev_indirect_ptr, iter_, looper, max_depth):
    n = C.c_ulonglong(0)
    check(lib().rs_path_trace_indirect(scene.handle, C.byref(cam), dev_indirect_ptr, iter_, looper, max_depth, C.byref(n)))
    return n.value


def path_trace_direct(scene, cam, dev_direct_illum_ptr, iter_, looper):
    n = C.c_ulonglong(0)
    check(lib().rs_path_trace_direct(scene.handle, C.byref(cam), dev_direct_illum_ptr, iter_, looper, C.byref(n)))
    return n.value


def copy_image_to_pbo(dev_pbo_ptr, dev_image_ptr, width, height, tone_mapping, scale=1.0):
    check(lib().rs_copy_image_to_pbo(dev_pbo_ptr, dev_image_ptr, width, height, tone_mapping, scale))


def copy_debug_image_to_pbo(dev_pbo_ptr, dev_image_ptr, width, height, kind):
    """The vec2 (kind 0) / float (1) / int (2) overloads of copyImageToPBO."""
    fn = (lib().rs_copy_image2_to_pbo, lib().rs_copy_imagef_to_pbo, lib().rs_copy_imagei_to_pbo)[kind]
    check(fn(dev_pbo_ptr, dev_image_ptr, width, height))


def trace_closest(scene, rays_t):
    """rays_t: torch float32 cuda tensor (n,6). Returns (primId, matId, pos, norm) torch tensors."""
    import torch
    n = rays_t.shape[0]
    prim = torch.empty(n, dtype=torch.int32, device="cuda"); mat = torch.empty(n, dtype=torch.int32, device="cuda")
    pos = torch.empty((n, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    check(lib().rs_trace_closest(scene.handle, n, rays_t.data_ptr(), prim.data_ptr(), mat.data_ptr(), pos.data_ptr(), nrm.data_ptr()))
    return prim, mat, pos, nrm


def trace_closest_wave(scene, rays_t):
    """As trace_closest, through the wave-level service of the multi-bounce kernels (rs_trace_closest_wave)."""
    import torch
    n = rays_t.shape[0]
    prim = torch.empty(n, dtype=torch.int32, device="cuda"); mat = torch.empty(n, dtype=torch.int32, device="cuda")
    pos = torch.empty((n, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    check(lib().rs_trace_closest_wave(scene.handle, n, rays_t.data_ptr(), prim.data_ptr(), mat.data_ptr(), pos.data_ptr(), nrm.data_ptr()))
    return prim, mat, pos, nrm


def set_ordered_tree(scene, on):
    """rs_scene_set_ordered_tree; returns whether the closest-hit trees were in use before the call."""
    import ctypes
    was = ctypes.c_int(0)
    check(lib().rs_scene_set_ordered_tree(scene.handle, 1 if on else 0, ctypes.byref(was)))
    return bool(was.value)


def trace_occlusion(scene, seg_t):
    import torch
    n = seg_t.shape[0]
    occ = torch.empty(n, dtype=torch.int32, device="cuda")
    check(lib().rs_trace_occlusion(scene.handle, n, seg_t.data_ptr(), occ.data_ptr()))
    return occ
