"""ctypes binding of the CPU oracle (oracle/liboracle.so) and, when present, of the
reference-derived checkers under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by restir_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from restir_amd.ctypes_structs import Camera, Material, Reservoir, MATERIAL_DTYPE, RESERVOIR_DTYPE, INDIRECT_RESERVOIR_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_SUBSET_PATH = os.path.join(_HERE, "_ref", "libref_subset.so")
THRUST_PROBE_PATH = os.path.join(_HERE, "_ref", "libthrust_probe.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is mounted)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "restir_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class OrcScene(C.Structure):
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
        ("sumLightPowerInv", C.c_float),
        ("numTextures", C.c_int),
        ("textures", C.c_void_p),
        ("envMapTexId", C.c_int),
        ("envMapSamplerLength", C.c_int),
        ("envMapProb", C.c_void_p),
        ("envMapFailId", C.c_void_p),
        ("sampleSequence", C.c_void_p),
    ]


class OrcTexture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.c_void_p)]


class OrcGBuffer(C.Structure):
    _fields_ = [
        ("albedo", C.c_void_p),
        ("motion", C.c_void_p),
        ("normal", C.c_void_p * 2),
        ("primId", C.c_void_p * 2),
        ("depth", C.c_void_p * 2),
        ("frameIdx", C.c_int),
        ("lastCamera", Camera),
        ("width", C.c_int),
        ("height", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.orc_intersect_triangle.argtypes = [C.c_int, f32p, f32p, i32p, f32p, f32p]
    L.orc_aabb_intersect.argtypes = [C.c_int, f32p, f32p, i32p, f32p]
    L.orc_utilhash.argtypes = [C.c_int, u32p, u32p]
    L.orc_rng_stream.argtypes = [C.c_int, i32p, i32p, i32p, C.c_int, f32p]
    L.orc_rng_stream_raw.argtypes = [C.c_int, i32p, C.c_int, f32p]
    L.orc_sobol_stream.argtypes = [u32p, C.c_int, i32p, i32p, i32p, C.c_int, f32p]
    L.orc_bsdf.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p]
    L.orc_camera_sample.argtypes = [C.POINTER(Camera), C.c_int, i32p, f32p, f32p]
    L.orc_camera_raster_coord.argtypes = [C.POINTER(Camera), C.c_int, f32p, i32p]
    L.orc_camera_position.argtypes = [C.POINTER(Camera), C.c_int, i32p, f32p, f32p]
    L.orc_camera_update.argtypes = [C.POINTER(Camera)]
    L.orc_sample_triangle_uniform.argtypes = [C.c_int, f32p, f32p, f32p]
    L.orc_to_concentric_disk.argtypes = [C.c_int, f32p, f32p]
    L.orc_triangle_misc.argtypes = [C.c_int, f32p, f32p, f32p, f32p, f32p]
    L.orc_tonemap.argtypes = [C.c_int, f32p, C.c_int, f32p]
    L.orc_bvh_build.argtypes = [C.c_int, f32p, f32p, C.POINTER(C.c_void_p * 6)]
    L.orc_bvh_build.restype = C.c_int
    L.orc_alias_build.argtypes = [C.c_int, f32p, f32p, i32p, C.POINTER(C.c_float)]
    L.orc_light_table.argtypes = [C.c_int, f32p, i32p, C.c_void_p, i32p, f32p, f32p]
    L.orc_light_table.restype = C.c_int
    L.orc_envmap_pdf.argtypes = [C.c_int, C.c_int, f32p, f32p]
    L.orc_set_libm_mode.argtypes = [C.c_int]
    L.orc_linear_sample.argtypes = [C.POINTER(OrcTexture), C.c_int, f32p, f32p]
    L.orc_to_sphere.argtypes = [C.c_int, f32p, f32p]
    L.orc_to_plane.argtypes = [C.c_int, f32p, f32p]
    L.orc_procedural_texture.argtypes = [C.c_int, f32p, f32p]
    L.orc_local_to_world.argtypes = [C.c_int, f32p, f32p, f32p]
    L.orc_material_sample.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p, f32p, f32p, u32p]
    L.orc_material_pdf.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p]
    L.orc_intersect.argtypes = [C.POINTER(OrcScene), C.c_int, f32p, i32p, i32p, f32p, f32p, f32p]
    L.orc_test_occlusion.argtypes = [C.POINTER(OrcScene), C.c_int, f32p, i32p]
    L.orc_sample_direct_light_nv.argtypes = [C.POINTER(OrcScene), C.c_int, f32p, f32p, f32p, f32p, f32p, f32p]
    L.orc_gbuffer_render.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(OrcGBuffer), C.c_int, C.c_int]
    L.orc_gbuffer_update.argtypes = [C.POINTER(OrcGBuffer), C.POINTER(Camera)]
    L.orc_pt_direct.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), f32p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.orc_restir_direct.argtypes = [
        C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(OrcGBuffer), f32p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong),
    ]
    L.orc_path_trace.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), f32p, f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.orc_pt_indirect.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.orc_restir_indirect.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(OrcGBuffer), f32p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.orc_restir_state_create.argtypes = [C.c_int, C.c_int]
    L.orc_restir_state_create.restype = C.c_void_p
    L.orc_restir_state_destroy.argtypes = [C.c_void_p]
    L.orc_restir_phase_a.argtypes = [
        C.c_void_p, C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(OrcGBuffer), C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong),
    ]
    L.orc_restir_phase_b.argtypes = [
        C.c_void_p, C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(OrcGBuffer), f32p, C.c_void_p,
        C.c_int, C.c_int, C.c_int, C.c_int,
    ]
    L.orc_send_image_to_pbo.argtypes = [C.c_int, C.c_int, f32p, C.c_int, C.c_float, u8p]
    L.orc_send_debug_to_pbo.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, u8p]
    L.orc_svgf_create.argtypes = [C.c_int, C.c_int]
    L.orc_svgf_destroy.argtypes = [C.c_void_p]
    L.orc_svgf_filter.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OrcGBuffer), C.POINTER(Camera)]
    L.orc_svgf_next_frame.argtypes = [C.c_void_p]
    for _n in ("orc_svgf_variance", "orc_svgf_accum_color", "orc_svgf_accum_moment"):
        getattr(L, _n).argtypes = [C.c_void_p]
    L.orc_eaw_level.argtypes = [C.POINTER(OrcGBuffer), C.POINTER(Camera), f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_int]
    L.orc_eaw_filter.argtypes = [C.POINTER(OrcGBuffer), C.POINTER(Camera), f32p, f32p, f32p]
    L.orc_eaw_filter.restype = C.c_void_p
    L.orc_modulate.argtypes = [C.c_int, C.c_int, f32p, f32p]
    L.orc_add.argtypes = [C.c_int, C.c_int, f32p, f32p]
    L.orc_add3.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p]
    L.orc_build_transformation_matrix.argtypes = [f32p, f32p, f32p, f32p]
    L.orc_bake_instance.argtypes = [f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_tanf.argtypes = [C.c_float]; L.orc_tanf.restype = C.c_float
    L.orc_atanf.argtypes = [C.c_float]; L.orc_atanf.restype = C.c_float
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# host scene build
# ---------------------------------------------------------------------------------------------
def bvh_build(vertices, fn=None):
    """vertices: (numPrims, 3, 3) float32 -> (boxes (S,6) f32, nodes (6,S,3) i32).
    fn: builder entry point (default orc_bvh_build; tests pass ref_subset().ref_bvh_build)."""
    fn = fn or lib().orc_bvh_build
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1)
    n = v.size // 9
    size = 2 * n - 1
    boxes = np.zeros((size, 6), np.float32)
    nodes = np.zeros((6, size, 3), np.int32)
    arr = (C.c_void_p * 6)(*[nodes[i].ctypes.data for i in range(6)])
    got = fn(n, v, boxes.reshape(-1), C.byref(arr))
    assert got == size
    return boxes, nodes


def alias_build(values):
    values = np.ascontiguousarray(values, np.float32)
    n = values.size
    prob = np.zeros(n, np.float32)
    fail = np.zeros(n, np.int32)
    s = C.c_float(0)
    lib().orc_alias_build(n, values, prob, fail, C.byref(s))
    return prob, fail, np.float32(s.value)


def envmap_pdf(env):
    """scene.cpp:139-146 on an (H, W, 3) float32 environment map."""
    env = np.ascontiguousarray(env, np.float32)
    pdf = np.zeros(env.shape[0] * env.shape[1], np.float32)
    lib().orc_envmap_pdf(env.shape[1], env.shape[0], env.reshape(-1), pdf)
    return pdf


def set_libm_mode(correctly_rounded):
    lib().orc_set_libm_mode(int(bool(correctly_rounded)))


def linear_sample(tex, uv):
    tex = np.ascontiguousarray(tex, np.float32); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    t = OrcTexture(); t.height, t.width = tex.shape[0], tex.shape[1]; t.data = tex.ctypes.data
    out = np.zeros((len(uv), 3), np.float32)
    lib().orc_linear_sample(C.byref(t), len(uv), uv.reshape(-1), out.reshape(-1))
    return out


def to_sphere(uv):
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2); out = np.zeros((len(uv), 3), np.float32)
    lib().orc_to_sphere(len(uv), uv.reshape(-1), out.reshape(-1)); return out


def to_plane(d):
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3); out = np.zeros((len(d), 2), np.float32)
    lib().orc_to_plane(len(d), d.reshape(-1), out.reshape(-1)); return out


def local_to_world(n, v):
    n = np.ascontiguousarray(n, np.float32).reshape(-1, 3); v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
    out = np.zeros((len(n), 3), np.float32)
    lib().orc_local_to_world(len(n), n.reshape(-1), v.reshape(-1), out.reshape(-1)); return out


def material_sample(mats, nrm, wo, r3, fn=None):
    """Material::sample on n inputs -> dir (n,3), bsdf (n,3), pdf (n,), type (n,) uint32."""
    mats = np.ascontiguousarray(mats); n = len(mats)
    nrm = np.ascontiguousarray(nrm, np.float32); wo = np.ascontiguousarray(wo, np.float32); r3 = np.ascontiguousarray(r3, np.float32)
    d = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32); p = np.zeros(n, np.float32); t = np.zeros(n, np.uint32)
    (fn or lib().orc_material_sample)(n, mats.ctypes.data_as(C.c_void_p), nrm.reshape(-1), wo.reshape(-1), r3.reshape(-1), d.reshape(-1), b.reshape(-1), p, t)
    return d, b, p, t


def material_pdf(mats, nrm, wo, wi, fn=None):
    mats = np.ascontiguousarray(mats); n = len(mats)
    p = np.zeros(n, np.float32)
    (fn or lib().orc_material_pdf)(n, mats.ctypes.data_as(C.c_void_p), np.ascontiguousarray(nrm, np.float32).reshape(-1),
                                   np.ascontiguousarray(wo, np.float32).reshape(-1), np.ascontiguousarray(wi, np.float32).reshape(-1), p)
    return p


def procedural_texture(uv):
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2); out = np.zeros((len(uv), 3), np.float32)
    lib().orc_procedural_texture(len(uv), uv.reshape(-1), out.reshape(-1)); return out


def light_table(vertices, material_ids, materials):
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1)
    n = v.size // 9
    ids = np.zeros(n, np.int32)
    rad = np.zeros((n, 3), np.float32)
    power = np.zeros(n, np.float32)
    mats = np.ascontiguousarray(materials)
    k = lib().orc_light_table(n, v, np.ascontiguousarray(material_ids, np.int32), _ptr(mats), ids, rad.reshape(-1), power)
    return ids[:k].copy(), rad[:k].copy(), power[:k].copy()


class Scene:
    """Host image of DevScene, built the way Scene::buildDevData does (scene.cpp:159-215)."""

    def __init__(self, vertices, normals, texcoords, material_ids, materials, prebuilt=None, textures=(), env_map_tex=-1):
        # textures: list of (H, W, 3) float32 linear-RGB arrays; env_map_tex: index of the environment map or -1
        self.textures = [np.ascontiguousarray(t, np.float32) for t in textures]
        self.env_map_tex = int(env_map_tex)
        self.env_prob = np.zeros(0, np.float32); self.env_fail = np.zeros(0, np.int32)
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3, 3)
        self.normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3, 3)
        self.texcoords = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 3, 2)
        self.material_ids = np.ascontiguousarray(material_ids, np.int32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        n = self.vertices.shape[0]
        if prebuilt is None:
            self.light_prim_ids, self.light_radiance, self.light_power = light_table(
                self.vertices, self.material_ids, self.materials)
            if self.env_map_tex >= 0:                       # Scene::createLightSampler (scene.cpp:136-152)
                env = self.textures[self.env_map_tex]
                self.env_prob, self.env_fail, env_sum = alias_build(envmap_pdf(env))
                self.light_power = np.concatenate([self.light_power, np.array([env_sum], np.float32)])
            if len(self.light_power):
                self.light_prob, self.light_fail, self.sum_power = alias_build(self.light_power)
            else:
                self.light_prob = np.zeros(0, np.float32)
                self.light_fail = np.zeros(0, np.int32)
                self.sum_power = np.float32(0)
            self.boxes, self.nodes = bvh_build(self.vertices)
        else:
            (self.light_prim_ids, self.light_radiance, self.light_power, self.light_prob,
             self.light_fail, self.sum_power, self.boxes, self.nodes) = prebuilt
        self.bvh_size = 2 * n - 1
        s = OrcScene()
        s.numPrims = n
        s.vertices = self.vertices.ctypes.data
        s.normals = self.normals.ctypes.data
        s.texcoords = self.texcoords.ctypes.data
        s.materialIds = self.material_ids.ctypes.data
        s.numMaterials = len(self.materials)
        s.materials = self.materials.ctypes.data
        s.bvhSize = self.bvh_size
        s.boundingBoxes = self.boxes.ctypes.data
        for i in range(6):
            s.bvhNodes[i] = self.nodes[i].ctypes.data
        s.numLights = len(self.light_prob)              # sampler length: light primitives (+1 with an environment map)
        s.lightPrimIds = self.light_prim_ids.ctypes.data
        s.lightUnitRadiance = self.light_radiance.ctypes.data
        s.lightProb = self.light_prob.ctypes.data
        s.lightFailId = self.light_fail.ctypes.data
        with np.errstate(divide="ignore"):
            s.sumLightPowerInv = np.float32(1.0) / np.float32(self.sum_power)
        self._tex = (OrcTexture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex[i].height, self._tex[i].width = t.shape[0], t.shape[1]
            self._tex[i].data = t.ctypes.data
        s.numTextures = len(self.textures)
        s.textures = C.cast(self._tex, C.c_void_p)
        s.envMapTexId = self.env_map_tex
        s.envMapSamplerLength = len(self.env_prob)
        s.envMapProb = self.env_prob.ctypes.data
        s.envMapFailId = self.env_fail.ctypes.data
        self.c = s
        self.sample_sequence = None

    def set_sample_sequence(self, table):
        """DevScene::sampleSequence (scene.cpp:500-506): the Sobol table as uint32 [SobolSampleNum, SobolSampleDim]; None selects the
        default thrust engine again (SAMPLER_USE_SOBOL false)."""
        if table is None:
            self.sample_sequence = None
            self.c.sampleSequence = None
            return
        t = np.ascontiguousarray(table, np.uint32)
        assert t.ndim == 2 and t.shape[1] == 200, t.shape                 # SobolSampleDim (sampler.h:11)
        # the reference reads data[ptr++] without a bound (sampler.h:20): a guard of zeros behind the table keeps the multi-bounce
        # kernels of the last rows inside the allocation (the product pads its device copy the same way)
        self.sample_sequence = np.concatenate([t.reshape(-1), np.zeros(4096, np.uint32)])
        self.sample_count = t.shape[0]
        self.c.sampleSequence = self.sample_sequence.ctypes.data

    # scene services ---------------------------------------------------------------
    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.zeros(n, np.int32); mat = np.zeros(n, np.int32)
        pos = np.zeros((n, 3), np.float32); nrm = np.zeros((n, 3), np.float32); uv = np.zeros((n, 2), np.float32)
        lib().orc_intersect(C.byref(self.c), n, rays.reshape(-1), prim, mat, pos.reshape(-1), nrm.reshape(-1), uv.reshape(-1))
        return prim, mat, pos, nrm, uv

    def test_occlusion(self, seg):
        seg = np.ascontiguousarray(seg, np.float32).reshape(-1, 6)
        occ = np.zeros(seg.shape[0], np.int32)
        lib().orc_test_occlusion(C.byref(self.c), seg.shape[0], seg.reshape(-1), occ)
        return occ

    def sample_direct_light_nv(self, pos, r):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(r, np.float32).reshape(-1, 4)
        n = pos.shape[0]
        pdf = np.zeros(n, np.float32); Li = np.zeros((n, 3), np.float32)
        wi = np.zeros((n, 3), np.float32); dist = np.zeros(n, np.float32)
        lib().orc_sample_direct_light_nv(C.byref(self.c), n, pos.reshape(-1), r.reshape(-1), pdf, Li.reshape(-1), wi.reshape(-1), dist)
        return pdf, Li, wi, dist


class GBuffer:
    """Host image of GBuffer (gbuffer.h:15-59; allocation as denoiser.cu:373-388)."""

    def __init__(self, width, height):
        n = width * height
        self.width, self.height = width, height
        self.albedo = np.zeros((n, 3), np.float32)
        self.motion = np.zeros(n, np.int32)
        self.normal = [np.zeros((n, 3), np.float32) for _ in range(2)]
        self.prim_id = [np.zeros(n, np.int32) for _ in range(2)]
        self.depth = [np.zeros(n, np.float32) for _ in range(2)]
        g = OrcGBuffer()
        g.albedo = self.albedo.ctypes.data
        g.motion = self.motion.ctypes.data
        for i in range(2):
            g.normal[i] = self.normal[i].ctypes.data
            g.primId[i] = self.prim_id[i].ctypes.data
            g.depth[i] = self.depth[i].ctypes.data
        g.frameIdx = 0
        g.width, g.height = width, height
        self.c = g

    @property
    def frame_idx(self):
        return self.c.frameIdx

    def render(self, scene, cam, y0=0, y1=None):
        lib().orc_gbuffer_render(C.byref(scene.c), C.byref(cam), C.byref(self.c), y0, self.height if y1 is None else y1)

    def update(self, cam):
        lib().orc_gbuffer_update(C.byref(self.c), C.byref(cam))


class ReSTIR:
    """Module state of restir.cu:8-18,478-518 + launcher :418-446 on host memory."""

    def __init__(self, width, height):
        n = width * height
        self.n = n
        self.reservoir = np.zeros(n, RESERVOIR_DTYPE)       # devDirectReservoir
        self.last = np.zeros(n, RESERVOIR_DTYPE)            # devLastDirectReservoir
        self.temp = np.zeros(n, RESERVOIR_DTYPE)            # devDirectTemp
        self.ind_reservoir = np.zeros(n, INDIRECT_RESERVOIR_DTYPE)   # devIndTemporalReservoir (restir.cu:13-14)
        self.ind_last = np.zeros(n, INDIRECT_RESERVOIR_DTYPE)        # devIndLastTemporalReservoir
        self.first = True
        self.rays = 0

    def indirect(self, scene, cam, gbuf, indirect_illum, iter_, looper, reuse, max_depth):
        """ReSTIRIndirect (restir.cu:448-476); shares ReSTIRFirstFrame with direct()."""
        rays = C.c_ulonglong(0)
        lib().orc_restir_indirect(C.byref(scene.c), C.byref(cam), C.byref(gbuf.c), indirect_illum.reshape(-1),
                                  _ptr(self.ind_reservoir), _ptr(self.ind_last), looper, iter_, max_depth, int(self.first), reuse, C.byref(rays))
        self.ind_reservoir, self.ind_last = self.ind_last, self.ind_reservoir
        self.first = False
        self.rays = rays.value
        return rays.value

    def reset(self):
        self.first = True

    # row-range phases (the decomposition the strip tiling uses)
    def phase_a(self, scene, cam, gbuf, looper, reuse, y0, y1):
        if getattr(self, "_state", None) is None:
            self._state = lib().orc_restir_state_create(gbuf.width, gbuf.height)
        rays = C.c_ulonglong(0)
        lib().orc_restir_phase_a(self._state, C.byref(scene.c), C.byref(cam), C.byref(gbuf.c), _ptr(self.reservoir),
                                 _ptr(self.last), _ptr(self.temp), looper, int(self.first), reuse, y0, y1, C.byref(rays))
        self.rays = rays.value

    def phase_b(self, scene, cam, gbuf, direct_illum, iter_, reuse, y0, y1):
        lib().orc_restir_phase_b(self._state, C.byref(scene.c), C.byref(cam), C.byref(gbuf.c), direct_illum.reshape(-1),
                                 _ptr(self.temp), iter_, reuse, y0, y1)

    def end_frame(self):
        self.reservoir, self.last = self.last, self.reservoir
        self.first = False

    def direct(self, scene, cam, gbuf, direct_illum, iter_, looper, reuse):
        rays = C.c_ulonglong(0)
        lib().orc_restir_direct(C.byref(scene.c), C.byref(cam), C.byref(gbuf.c), direct_illum.reshape(-1),
                                _ptr(self.reservoir), _ptr(self.last), _ptr(self.temp),
                                looper, iter_, int(self.first), reuse, C.byref(rays))
        self.reservoir, self.last = self.last, self.reservoir
        self.first = False
        self.rays = rays.value
        return rays.value


def pt_direct(scene, cam, direct_illum, iter_, looper):
    rays = C.c_ulonglong(0)
    lib().orc_pt_direct(C.byref(scene.c), C.byref(cam), direct_illum.reshape(-1), looper, iter_, C.byref(rays))
    return rays.value


def path_trace(scene, cam, direct_illum, indirect_illum, iter_, looper, max_depth):
    """pathTrace = singleKernelPT (pathtrace.cu:156-277,434-455)."""
    rays = C.c_ulonglong(0)
    lib().orc_path_trace(C.byref(scene.c), C.byref(cam), direct_illum.reshape(-1), indirect_illum.reshape(-1), looper, iter_, max_depth, C.byref(rays))
    return rays.value


def pt_indirect(scene, cam, indirect_illum, iter_, looper, max_depth):
    """pathTraceIndirect = PTIndirectKernel (pathtrace.cu:330-432,478-497)."""
    rays = C.c_ulonglong(0)
    lib().orc_pt_indirect(C.byref(scene.c), C.byref(cam), indirect_illum.reshape(-1), looper, iter_, max_depth, C.byref(rays))
    return rays.value


def send_image_to_pbo(image, w, h, tone_mapping, scale=1.0):
    out = np.zeros((h * w, 4), np.uint8)
    lib().orc_send_image_to_pbo(w, h, np.ascontiguousarray(image, np.float32).reshape(-1), tone_mapping, scale, out.reshape(-1))
    return out


def send_debug_to_pbo(image, w, h, kind):
    """kind 0: (N,2) float32, 1: (N,) float32, 2: (N,) int32 pixel indices (pathtrace.cu:58-106)."""
    image = np.ascontiguousarray(image, np.int32 if kind == 2 else np.float32)
    out = np.zeros((h * w, 4), np.uint8)
    lib().orc_send_debug_to_pbo(w, h, image.ctypes.data_as(C.c_void_p), kind, out.reshape(-1))
    return out


class SVGF:
    """SpatioTemporalFilter (denoiser.h:45-70) on the CPU oracle."""

    def __init__(self, width, height):
        self.n = width * height
        L = lib()
        L.orc_svgf_create.restype = C.c_void_p
        for name in ("orc_svgf_filter", "orc_svgf_variance", "orc_svgf_accum_color", "orc_svgf_accum_moment"):
            getattr(L, name).restype = C.POINTER(C.c_float)
        self.h = C.c_void_p(L.orc_svgf_create(width, height))

    def _grab(self, ptr, count):
        return np.ctypeslib.as_array(ptr, (count,)).copy()

    def filter(self, color_in, gbuf, cam):
        p = lib().orc_svgf_filter(self.h, np.ascontiguousarray(color_in, np.float32).reshape(-1).ctypes.data_as(C.c_void_p), C.byref(gbuf.c), C.byref(cam))
        return self._grab(p, self.n * 3).reshape(self.n, 3)

    def next_frame(self):
        lib().orc_svgf_next_frame(self.h)

    def state(self):
        L = lib()
        return dict(variance=self._grab(L.orc_svgf_variance(self.h), self.n),
                    accum_color=self._grab(L.orc_svgf_accum_color(self.h), self.n * 3).reshape(self.n, 3),
                    accum_moment=self._grab(L.orc_svgf_accum_moment(self.h), self.n * 3).reshape(self.n, 3))

    def __del__(self):
        try:
            lib().orc_svgf_destroy(self.h)
        except Exception:
            pass


def eaw_filter(gbuf, cam, color_in):
    n = gbuf.width * gbuf.height
    out = np.zeros((n, 3), np.float32); tmp = np.zeros((n, 3), np.float32)
    p = lib().orc_eaw_filter(C.byref(gbuf.c), C.byref(cam), np.ascontiguousarray(color_in, np.float32).reshape(-1), out.reshape(-1), tmp.reshape(-1))
    return out if p == out.ctypes.data else tmp


def eaw_filter_with(gbuf, cam, color_in, sig_lumin, sig_normal, sig_depth):
    """LeveledEAWFilter::filter (src/denoiser.cu:463-477) with the sigmas the viewer may have set (src/preview.cpp:263-265)."""
    n = gbuf.width * gbuf.height
    src = np.ascontiguousarray(color_in, np.float32).reshape(-1).copy()
    bufs = [np.zeros(n * 3, np.float32), np.zeros(n * 3, np.float32)]
    for level in range(5):
        dst = bufs[level % 2]
        lib().orc_eaw_level(C.byref(gbuf.c), C.byref(cam), src, dst, sig_depth, sig_normal, sig_lumin, level)
        src = dst
    return src.reshape(n, 3)


def camera_update(cam):
    lib().orc_camera_update(C.byref(cam))
    return cam


def build_transformation_matrix(t, r, s):
    """Math::buildTransformationMatrix (src/mathUtil.cpp:13-20) -> (4, 4) float32, [column][row]."""
    out = np.zeros(16, np.float32)
    lib().orc_build_transformation_matrix(*(np.ascontiguousarray(a, np.float32) for a in (t, r, s)), out)
    return out.reshape(4, 4)


def bake_instance(t, r, s, vertices, normals):
    """scene.cpp:167-168 for (n, 3) vertices / normals of one instance."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1)
    vo, no = np.zeros_like(v), np.zeros_like(n)
    lib().orc_bake_instance(*(np.ascontiguousarray(a, np.float32) for a in (t, r, s)), len(v) // 3, v, n, vo, no)
    return vo.reshape(-1, 3), no.reshape(-1, 3)


def tanf(x):
    return np.float32(lib().orc_tanf(float(x)))


def atanf(x):
    return np.float32(lib().orc_atanf(float(x)))


# ---------------------------------------------------------------------------------------------
# reference-derived checkers (this container only)
# ---------------------------------------------------------------------------------------------
def ref_subset():
    if not os.path.exists(REF_SUBSET_PATH):
        return None
    R = C.CDLL(REF_SUBSET_PATH)
    R.ref_intersect_triangle.argtypes = [C.c_int, f32p, f32p, i32p, f32p, f32p]
    R.ref_aabb_intersect.argtypes = [C.c_int, f32p, f32p, i32p, f32p]
    R.ref_utilhash.argtypes = [C.c_int, u32p, u32p]
    R.ref_bsdf.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p]
    R.ref_camera_sample.argtypes = [C.POINTER(Camera), C.c_int, i32p, f32p, f32p]
    R.ref_camera_raster_coord.argtypes = [C.POINTER(Camera), C.c_int, f32p, i32p]
    R.ref_camera_position.argtypes = [C.POINTER(Camera), C.c_int, i32p, f32p, f32p]
    R.ref_camera_update.argtypes = [C.POINTER(Camera)]
    R.ref_sample_triangle_uniform.argtypes = [C.c_int, f32p, f32p, f32p]
    R.ref_to_concentric_disk.argtypes = [C.c_int, f32p, f32p]
    R.ref_triangle_misc.argtypes = [C.c_int, f32p, f32p, f32p, f32p, f32p]
    R.ref_tonemap.argtypes = [C.c_int, f32p, C.c_int, f32p]
    R.ref_bvh_build.argtypes = [C.c_int, f32p, f32p, C.POINTER(C.c_void_p * 6)]
    R.ref_material_sample.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p, f32p, f32p, u32p]
    R.ref_material_pdf.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p]
    R.ref_linear_sample.argtypes = [C.c_int, C.c_int, f32p, C.c_int, f32p, f32p]
    R.ref_to_sphere.argtypes = [C.c_int, f32p, f32p]
    R.ref_to_plane.argtypes = [C.c_int, f32p, f32p]
    R.ref_local_to_world.argtypes = [C.c_int, f32p, f32p, f32p]
    R.ref_bvh_build.restype = C.c_int
    R.ref_closest_hit_loop.argtypes = [C.c_int, f32p, C.POINTER(C.c_void_p * 6), f32p, C.c_int, f32p, i32p, f32p]
    R.ref_build_transformation_matrix.argtypes = [f32p, f32p, f32p, f32p]
    R.ref_bake_instance.argtypes = [f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, f32p]
    return R


def ref_closest_hit_loop(R, vertices, rays, min_seconds=0.0):
    """Closest hits of `rays` (n, 6) in the triangle soup `vertices` through the reference's own builder and intersection code
    (oracle/ref_subset.cpp ref_bvh_build + ref_closest_hit_loop).  The traced loop is repeated until `min_seconds` have passed.
    Returns (prim ids, distances, seconds per pass over the rays)."""
    import time
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1)
    n_prims = v.size // 9
    size = 2 * n_prims - 1
    boxes = np.zeros(size * 6, np.float32)
    nodes = [np.zeros(size * 3, np.int32) for _ in range(6)]
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in nodes])
    bvh_size = R.ref_bvh_build(n_prims, v, boxes, C.byref(ptrs))
    r = np.ascontiguousarray(rays, np.float32).reshape(-1)
    n = r.size // 6
    prim = np.zeros(n, np.int32); dist = np.zeros(n, np.float32)
    R.ref_closest_hit_loop(bvh_size, boxes, C.byref(ptrs), v, min(n, 4096), r, prim, dist)      # untimed: thread start-up
    passes, t0 = 0, time.perf_counter()
    while True:
        R.ref_closest_hit_loop(bvh_size, boxes, C.byref(ptrs), v, n, r, prim, dist)
        passes += 1
        if time.perf_counter() - t0 >= min_seconds:
            break
    return prim, dist, (time.perf_counter() - t0) / passes


def ref_loaders():
    """The reference's own file loaders (oracle/ref_loaders.cpp); None when not built."""
    path = os.path.join(os.path.dirname(REF_SUBSET_PATH), "libref_loaders.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_image_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int]
    R.ref_obj_load.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_read_lines.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    R.ref_save_jpg.argtypes = [C.c_char_p, C.c_int, C.c_int, f32p]
    return R


def thrust_probe():
    if not os.path.exists(THRUST_PROBE_PATH):
        return None
    T = C.CDLL(THRUST_PROBE_PATH)
    T.thr_rng_stream_raw.argtypes = [C.c_int, i32p, C.c_int, f32p]
    T.thr_version.restype = C.c_int
    return T
