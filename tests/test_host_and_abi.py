"""CPU: the C-ABI library loads and exports every symbol include/restir_hip.h declares, and its
host-side builders (BVH, alias table, light table, camera) agree bit-for-bit with the oracle.
No compute call needs a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import binding as ob
from restir_amd import capi, scenes
from restir_amd.ctypes_structs import LIGHT, copy_camera, make_camera, make_materials
from tests.common import bits_equal, get_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "restir_hip.h")).read()
    declared = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 45
    lib = capi.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)


def test_struct_layouts_match_reference_sizes():
    from restir_amd.ctypes_structs import Camera, Material, Reservoir
    assert C.sizeof(Material) == 44 and C.sizeof(Camera) == 196 and C.sizeof(Reservoir) == 36


@pytest.mark.parametrize("name", ["cornell", "sponza:0.02", "sponza:0.1"])
def test_scene_build_matches_oracle(name):
    sd = get_scene(name)
    ba, na = capi.build_bvh(sd.vertices)
    bb, nb = ob.bvh_build(sd.vertices)
    assert bits_equal(ba, bb) and np.array_equal(na, nb)
    la = capi.build_light_table(sd.vertices, sd.material_ids, sd.materials)
    lb = ob.light_table(sd.vertices, sd.material_ids, sd.materials)
    assert np.array_equal(la[0], lb[0]) and bits_equal(la[1], lb[1]) and bits_equal(la[2], lb[2])
    aa = capi.build_alias_table(la[2]); ab = ob.alias_build(lb[2])
    assert bits_equal(aa[0], ab[0]) and np.array_equal(aa[1], ab[1]) and aa[2] == ab[2]


def test_environment_map_sampler_build_matches_oracle():
    """Scene::createLightSampler (src/scene.cpp:136-157): pdf = lum * sin(theta) per texel, alias table over it, and
    the map's total as one more entry of the light sampler."""
    sd = get_scene("cornell_textured")
    env = sd.textures[sd.env_map_tex]
    pa, pb = capi.build_envmap_pdf(env), ob.envmap_pdf(env)
    assert bits_equal(pa, pb) and pa.shape == (env.shape[0] * env.shape[1],)
    aa, ab = capi.build_alias_table(pa), ob.alias_build(pb)
    assert bits_equal(aa[0], ab[0]) and np.array_equal(aa[1], ab[1]) and aa[2] == ab[2]
    osc = ob.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, textures=sd.textures, env_map_tex=sd.env_map_tex)
    la = capi.build_light_table(sd.vertices, sd.material_ids, sd.materials)
    power = np.concatenate([la[2], [aa[2]]]).astype(np.float32)
    al = capi.build_alias_table(power)
    assert bits_equal(al[0], osc.light_prob) and np.array_equal(al[1], osc.light_fail) and al[2] == osc.sum_power
    assert len(osc.light_prob) == len(osc.light_prim_ids) + 1


def test_bvh_degenerate_inputs():
    rng = np.random.default_rng(3)
    one = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32)
    b, n = capi.build_bvh(one)
    assert n.shape == (6, 1, 3) and (n[:, 0, 0] == 0).all() and (n[:, 0, 2] == 1).all()
    # identical triangles: every split has dimMax == dimMin (bvh.cpp:83 int(NaN) -> bucket 0)
    same = np.repeat(one, 9, axis=0)
    ba, na = capi.build_bvh(same); bb, nb = ob.bvh_build(same)
    assert bits_equal(ba, bb) and np.array_equal(na, nb)
    # threaded links always point forward and end at BVHSize
    size = na.shape[1]
    assert (na[:, :, 2] > np.arange(size)[None, :]).all() and (na[:, :, 2] <= size).all()
    for k in range(6):
        assert sorted(na[k, :, 1].tolist()) == list(range(size))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1024])
def test_alias_table_properties(n):
    rng = np.random.default_rng(n)
    v = np.exp(rng.uniform(-4, 4, n)).astype(np.float32)
    prob, fail, total = capi.build_alias_table(v)
    pr, fr, tr = ob.alias_build(v)
    assert bits_equal(prob, pr) and np.array_equal(fail, fr) and total == tr
    # the table reproduces the distribution: p_i = (prob_i + sum_{j: fail_j = i} (1 - prob_j)) / n
    p = np.minimum(prob, 1.0).astype(np.float64)
    mass = p.copy()
    for j in range(n):
        if fail[j] != j:
            mass[fail[j]] += 1.0 - p[j]
    assert np.allclose(mass / n, v.astype(np.float64) / v.astype(np.float64).sum(), atol=2e-5)


def test_light_table_empty_and_errors():
    sd = scenes.cornell_box()
    mats = sd.materials.copy(); mats["type"] = 0
    ids, rad, power = capi.build_light_table(sd.vertices, sd.material_ids, mats)
    assert len(ids) == 0
    with pytest.raises(capi.RestirHipError):
        capi.build_light_table(sd.vertices, sd.material_ids + 100, sd.materials)
    assert b"out of range" in capi.lib().rs_last_error()


def test_camera_update_matches_oracle():
    for args in [(256, 256, (0, 1, 3.5), (-90, 0, 0), 19.5), (1920, 1080, (3, 2, -7), (37, -12, 0), 30.0), (640, 360, (0, 0, 0), (0, 89.5, 0), 45.0)]:
        a = make_camera(*args); b = copy_camera(a)
        capi.camera_update(a); ob.camera_update(b)
        assert bytes(a) == bytes(b)


def test_no_gpu_means_loud_failure():
    """On a machine without an MI355X the device entry points raise; nothing silently falls back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RestirHipError):
        capi.init(0)


def test_texture_table_is_validated_before_any_device_work():
    """Map ids a material may carry (src/scene.h:78-99) are checked on the host; the errors need no GPU."""
    sd = get_scene("cornell_maps")
    ok = dict(vertices=sd.vertices, normals=sd.normals, texcoords=sd.texcoords, material_ids=sd.material_ids)

    def build(mats, textures, env=-1):
        return capi.Scene(ok["vertices"], ok["normals"], ok["texcoords"], ok["material_ids"], mats, textures=textures, env_map_tex=env)

    bad = sd.materials.copy(); bad[0]["baseColorMapId"] = len(sd.textures)          # beyond the table
    with pytest.raises(capi.RestirHipError, match="texture table"):
        build(bad, sd.textures)
    bad = sd.materials.copy(); bad[5]["normalMapId"] = -2                            # ProceduralTexId is not a normal map
    with pytest.raises(capi.RestirHipError, match="map id"):
        build(bad, sd.textures)
    with pytest.raises(capi.RestirHipError, match="texture table"):
        build(sd.materials, sd.textures, env=len(sd.textures))                       # environment map id out of range


def test_argument_errors_are_reported_not_crashed():
    """Entry points validate their arguments before any device work: 0 sizes, null pointers, inconsistent tables."""
    import ctypes as C
    L = capi.lib()
    h = C.c_void_p()
    for fn, args in ((L.rs_gbuffer_create, (0, 16, C.byref(h))), (L.rs_gbuffer_create, (16, -1, C.byref(h))),
                     (L.rs_restir_init, (0, 0, C.byref(h))), (L.rs_eaw_create, (0, 8, 5, C.byref(h))), (L.rs_svgf_create, (8, 0, 5, C.byref(h)))):
        assert fn(*args) != 0 and h.value is None and len(L.rs_last_error()) > 0
    assert L.rs_scene_create(None, C.byref(h)) != 0
    assert L.rs_build_envmap_pdf(0, 4, None, None) != 0
    v = np.zeros(9, np.float32)
    n = C.c_int(0)
    assert L.rs_build_bvh(0, v.ctypes.data_as(C.c_void_p), None, None, C.byref(n)) != 0
    with pytest.raises(capi.RestirHipError):
        capi.check(L.rs_build_alias_table(-1, None, None, None, None))
    # a scene description whose BVH size does not match its triangle count
    sd = get_scene("cornell")
    with pytest.raises(capi.RestirHipError):
        t = dict(boxes=np.zeros((3, 6), np.float32), nodes=np.zeros((6, 3, 3), np.int32), light_prim_ids=np.zeros(0, np.int32),
                 light_radiance=np.zeros((0, 3), np.float32), light_prob=np.zeros(0, np.float32), light_fail=np.zeros(0, np.int32), sum_power=0.0)
        capi.Scene.from_tables(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, t)


def test_malformed_mtbvh_tables_are_refused_on_the_host():
    """rs_scene_create takes caller-built tables (DevScene::create from the viewer's own BVHBuilder).  Links must point forward
    and the threaded order must be a tree over distinct boxes: anything else is refused before any device work -- an inner
    node at the last index used to read one record past the table."""
    from oracle import binding as ob
    sd = get_scene("cornell")
    boxes, nodes = ob.bvh_build(sd.vertices)
    lp, lr, power = ob.light_table(sd.vertices, sd.material_ids, sd.materials)
    prob, fail, total = ob.alias_build(power)
    def tables(n):
        return dict(boxes=boxes, nodes=n, light_prim_ids=lp, light_radiance=lr, light_prob=prob, light_fail=fail, sum_power=total)
    def create(n):
        return capi.Scene.from_tables(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, tables(n))
    size = nodes.shape[1]
    bad = nodes.copy(); bad[0, size - 1, 0] = -1                       # the last record claims to be an inner node
    with pytest.raises(capi.RestirHipError, match="inner node without children|malformed"):
        create(bad)
    bad = nodes.copy(); bad[0, 1, 1] = bad[0, 2, 1]                    # two nodes share a bounding box id
    with pytest.raises(capi.RestirHipError, match="boundingBoxId"):
        create(bad)
    bad = nodes.copy(); bad[0, 1, 2] = 2                               # a subtree that ends before its first child does
    with pytest.raises(capi.RestirHipError):
        create(bad)
    bad = nodes.copy(); bad[3, 5, 2] = 3                               # a backward link in another order
    with pytest.raises(capi.RestirHipError, match="forward"):
        create(bad)


def test_png_writer_round_trip(tmp_path):
    """rs_write_png (the file Image::savePNG writes, src/image.cpp:41-58): sizes around the 65 535-byte stored-block limit, every chunk
    CRC and the Adler-32 checked by zlib, pixels back bit for bit; bad arguments are refused."""
    from restir_amd import capi
    from tests.common import read_png_rgb
    rng = np.random.default_rng(3)
    for h, w in ((1, 1), (7, 5), (5, 4369), (300, 421)):          # 5 x (1 + 3 * 4369) = 65 540 bytes: just over one stored block
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        capi.write_png(tmp_path / "t.png", a)
        assert np.array_equal(read_png_rgb(tmp_path / "t.png"), a)
    with pytest.raises(capi.RestirHipError):
        capi.write_png(tmp_path / "no_such_dir" / "t.png", np.zeros((2, 2, 3), np.uint8))
    assert capi.lib().rs_write_png(b"x.png", None, 2, 2) != 0


def test_jpg_writer_writes_the_reference_s_file(tmp_path):
    """rs_write_jpg (restir_amd/csrc/jpeg_writer.cpp) against the bytes the REFERENCE's own Image::saveJPG wrote for the same pictures
    (src/image.cpp:60-74 -> stbi_write_jpg at quality 90, compiled in place; tests/golden/jpeg_ref.npz by make_jpeg_golden.py):
    40 pictures -- sizes that are and are not multiples of 8, a single pixel, flat, gradients, noise, saturated -- byte for byte."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_ref.npz"))
    n = len(g.files) // 2
    assert n == 40
    for i in range(n):
        path = tmp_path / ("p%d.jpg" % i)
        capi.write_jpg(path, g["in_%d" % i])
        got = np.frombuffer(open(path, "rb").read(), np.uint8)
        assert np.array_equal(got, g["out_%d" % i]), (i, g["in_%d" % i].shape)
    with pytest.raises(capi.RestirHipError):
        capi.write_jpg(tmp_path / "no" / "such" / "dir.jpg", g["in_0"])


def test_procedural_scene_budgets():
    sd = scenes.sponza_class(1, 1.0)
    assert sd.num_prims == 262144
    assert int((sd.materials["type"][sd.material_ids] == LIGHT).sum()) == 1024
    b = scenes.bistro_class(2, 0.01)
    assert b.num_prims > 10000 and int((b.materials["type"][b.material_ids] == LIGHT).sum()) == 2 * max(8, round(5120 * 0.01))


@pytest.mark.parametrize("name", ["cornell", "sponza:0.05", "bistro:0.02"])
def test_closest_hit_trees_keep_the_reference_order(name):
    """Host side of the bounce rays' closest-hit trees (occlusion_bvh.cpp rs_build_ordered_bvh): read in walking order their
    leaves list the triangles exactly as the threaded orders of src/bvh.cpp:156-193 meet them (forward tree = order 2a, mirrored
    tree = order 2a + 1), boxes contain what is below them, miss links nest -- checked by the library's own host routine on
    rs_build_bvh's tables; and a table whose odd order is not the mirror image of the even one is refused."""
    from restir_amd import capi
    from tests.common import get_scene
    sd = get_scene(name)
    boxes, nodes = capi.build_bvh(sd.vertices)
    e, counts, depth = capi.ordered_bvh_host_check(boxes, nodes)
    assert e == 0, capi.lib().rs_last_error().decode()
    n = sd.vertices.size // 9
    for c, d in zip(counts, depth):
        assert n / 4 <= c < 2 * n            # leaves of 1..4 triangles
        assert d <= 64 + int(np.log2(n)) + 2
    bad = nodes.copy(); bad[1] = bad[0]
    e, _, _ = capi.ordered_bvh_host_check(boxes, bad)
    assert e == 10002                        # RS_ERR_UNSUPPORTED
