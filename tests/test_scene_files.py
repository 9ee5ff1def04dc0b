"""The scene-file front end: Scene::Scene(filename), loadMaterial / loadModel / loadCamera, Resource::loadOBJMesh, Image(filename)
and the instance baking of buildDevData (/root/reference/src/scene.cpp:27-61,96-131,161-176,222-433).

Three layers, each bit-exact:
  * the oracle restatement (oracle/scene_format.py + orc_bake_instance) against the reference's own loaders -- tinyobj::LoadObj,
    stbi_loadf, safeGetline / tokenizeString, Math::buildTransformationMatrix + GLM -- through the committed fixture
    tests/golden/scene_files.npz (made by tests/golden/make_scene_golden.py) and, in the build container, live on fuzzed inputs;
  * the product (rs_scene_file_load & co. of librestir_hip.so, host code) against the oracle and the same fixture;
  * on the GPU: a scene loaded from disk by the product renders the same frames as the oracle rendering the oracle-parsed scene.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as ob
from oracle import scene_format as sf
from restir_amd import capi
from tests import scene_file_cases as cases
from tests.common import bits_equal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scene_files.npz")
RL = ob.ref_loaders()
RS = ob.ref_subset()
needs_ref = pytest.mark.skipif(RL is None or RS is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def case_dir(g, tmp_path_factory):
    """The fixture's files written back to disk (the bytes the reference's loaders were run on)."""
    d = tmp_path_factory.mktemp("scene_case")
    for name in cases.CASE_FILES:
        (d / name).write_bytes(g["file_" + name].tobytes())
    return str(d)


def _joined(lines):
    return "".join("\t".join(sf.tokenize(l)) + "\n" for l in lines).encode("latin-1")


# ---- oracle vs the reference's loaders (fixture) ----------------------------------------------------------------------
def test_case_generator_is_deterministic(g, tmp_path):
    cases.write_case(str(tmp_path))
    for name in cases.CASE_FILES:
        assert (tmp_path / name).read_bytes() == g["file_" + name].tobytes(), name


def test_oracle_obj_reader_matches_tinyobj(g, case_dir):
    for name in cases.CASE_FILES:
        if not name.endswith(".obj"):
            continue
        v, n, t = sf.load_obj(os.path.join(case_dir, name))
        assert len(v) == len(g["obj_v_" + name]) and len(v) % 3 == 0, name
        assert bits_equal(v, g["obj_v_" + name]) and bits_equal(n, g["obj_n_" + name]) and bits_equal(t, g["obj_t_" + name]), name
    assert len(g["obj_v_cube.obj"]) == 36 and len(g["obj_v_numbers.obj"]) == 18      # quads became two triangles each
    assert len(g["obj_v_poly.obj"]) == 3 * (3 + 6 + 4 + 3)                            # ear-clipped 5-, 8-, 6- and 5-gons


def test_oracle_image_decoder_matches_stb_image(g, case_dir):
    for name in cases.CASE_FILES:
        if name.endswith((".ppm", ".hdr", ".png")):
            assert bits_equal(sf.load_image(os.path.join(case_dir, name), True), g["img_flip_" + name]), name
            assert bits_equal(sf.load_image(os.path.join(case_dir, name), False), g["img_noflip_" + name]), name
            assert not np.array_equal(g["img_flip_" + name], g["img_noflip_" + name])
    assert g["img_noflip_rgba.png"].shape == (9, 13, 3) and g["img_noflip_rgb16.png"].shape == (5, 7, 3)
    assert g["img_noflip_env.hdr"].max() > 50 and g["img_noflip_flat.hdr"].max() < 0.01      # HDR range, not byte / 255


def test_oracle_line_reader_matches_safe_getline(g, case_dir, tmp_path):
    for name in cases.CASE_FILES:
        assert _joined(sf.read_lines(os.path.join(case_dir, name))) == g["lines_" + name].tobytes(), name
    for key in ("crlf", "cr", "nofinal", "blanks"):
        p = tmp_path / "variant.txt"
        p.write_bytes(g["variant_" + key].tobytes())
        assert _joined(sf.read_lines(str(p))) == g["variant_lines_" + key].tobytes(), key


def test_oracle_baking_matches_glm(g):
    for i in range(len(g["bake_t"])):
        t, r, s = g["bake_t"][i], g["bake_r"][i], g["bake_s"][i]
        assert bits_equal(ob.build_transformation_matrix(t, r, s).reshape(-1), g["bake_matrix"][i]), i
        v, n = ob.bake_instance(t, r, s, g["bake_verts"][i], g["bake_normals"][i])
        assert bits_equal(v, g["bake_verts_out"][i]) and bits_equal(n, g["bake_normals_out"][i]), i


# ---- oracle vs the reference's loaders, live (build container) --------------------------------------------------------
@needs_ref
def test_obj_number_syntax_fuzz_against_tinyobj(tmp_path):
    rng = np.random.default_rng(21)
    toks = []
    for _ in range(6000):
        kind = rng.integers(0, 6)
        x = float(np.float32(rng.normal() * 10.0 ** rng.integers(-12, 13)))
        if kind == 0:
            toks.append(repr(x))
        elif kind == 1:
            toks.append("%.*g" % (int(rng.integers(1, 25)), x))
        elif kind == 2:
            toks.append("%.*f" % (int(rng.integers(0, 30)), x))
        elif kind == 3:
            toks.append("%.*e" % (int(rng.integers(0, 22)), x))
        elif kind == 4:
            toks.append(repr(np.float32(x).item()).replace("0.", ".", 1) if abs(x) < 1 else "+" + repr(x))
        else:
            toks.append(str(int(rng.integers(-10 ** 9, 10 ** 9))) + rng.choice(["", ".", ".0", "e0", "E+1", "e-2"]))
    toks += ["abc", "-", ".", "1e", "1e+", "--1", "1..2", "0x10", "1e400", "1e-400", "nan", "inf", "3,5"]
    while len(toks) % 3:
        toks.append("0")
    p = tmp_path / "fuzz.obj"
    n = len(toks) // 3
    with open(p, "w") as f:
        for i in range(n):
            f.write("v " + " ".join(toks[3 * i:3 * i + 3]) + "\n")
        f.write("vn 0 1 0\n")
        for i in range(n // 3):
            f.write(f"f {3 * i + 1}//1 {3 * i + 2}//1 {3 * i + 3}//1\n")
    cap = n + 8
    v = np.zeros((cap, 3), np.float32); nn = np.zeros((cap, 3), np.float32); t = np.zeros((cap, 2), np.float32)
    cnt = RL.ref_obj_load(str(p).encode(), cap, v.ctypes.data, nn.ctypes.data, t.ctypes.data)
    mv, mn, mt = sf.load_obj(str(p))
    assert cnt == len(mv) == (n // 3) * 3
    assert bits_equal(v[:cnt], mv) and bits_equal(nn[:cnt], mn) and bits_equal(t[:cnt], mt)
    # the product reads the same numbers: a one-object scene with the identity instance, against the oracle's baking of the
    # reference's numbers (vertices pass through the baking arithmetic unchanged unless they are not finite)
    scene = tmp_path / "fuzz.txt"
    scene.write_text("Object o\n" + str(p) + "\nMaterial Null\nScale 1 1 1\n\nCamera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\n"
                     "ApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n")
    a = capi.SceneFile(str(scene))
    bv, bn = ob.bake_instance((0, 0, 0), (0, 0, 0), (1, 1, 1), v[:cnt], nn[:cnt])
    assert bits_equal(a.vertices.reshape(-1, 3), bv) and bits_equal(a.normals.reshape(-1, 3), bn)


@needs_ref
def test_hdr_decoder_fuzz_against_stb_image(tmp_path):
    """Random Radiance pictures -- widths on both sides of the run-length limits, long runs, zero exponents, flat and coded --
    decoded by the oracle restatement and by the product equal stbi_loadf's floats under both flip settings."""
    from restir_amd import scene_io
    rng = np.random.default_rng(31)
    cam = "Camera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"
    obj = tmp_path / "t.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    for i, (h, w, rle) in enumerate([(3, 7, True), (4, 8, True), (5, 33, True), (2, 300, True), (6, 40, False), (1, 129, True)]):
        img = rng.uniform(0, 1, (h, w, 3)) * 10.0 ** rng.integers(-6, 6, (h, w, 1))
        img[:, w // 3: w // 3 + w // 2] = img[:, w // 3: w // 3 + 1]            # long runs
        img[0, :2] = 0.0                                                       # zero exponent
        p = str(tmp_path / f"r{i}.hdr")
        scene_io.write_hdr(p, img, rle=rle)
        for flip in (0, 1):
            ww, hh = C.c_int(), C.c_int()
            buf = np.zeros(h * w * 3, np.float32)
            assert RL.ref_image_load(p.encode(), flip, C.byref(ww), C.byref(hh), buf.ctypes.data, buf.size) == 0
            assert (hh.value, ww.value) == (h, w)
            ref = buf.reshape(h, w, 3)
            assert bits_equal(sf.load_hdr(p, bool(flip)), ref), (i, flip)
            # the product: as a texture (flipped) or as the environment map (not flipped) of a one-triangle scene
            scene = tmp_path / "s.txt"
            if flip:
                scene.write_text(f"Material m\nType Lambertian\nBaseColor {p}\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n"
                                 f"Object o\n{obj}\nMaterial m\nScale 1 1 1\n\n" + cam)
            else:
                scene.write_text(f"Object o\n{obj}\nMaterial Null\nScale 1 1 1\n\n" + cam + f"EnvMap {p}\n")
            a = capi.SceneFile(str(scene))
            assert len(a.textures) == 1 and bits_equal(a.textures[0], ref), (i, flip)
        dec = sf.load_hdr(p, False)                                            # RGBE keeps 8 bits of a pixel's largest component
        assert (np.abs(dec - img).max(axis=2) <= img.max(axis=2) / 128 + 1e-30).all()


@needs_ref
def test_png_decoder_against_stb_image(tmp_path):
    """Every legal PNG colour type / bit depth, plain and Adam7-interlaced, all five scan-line filters, stored / fixed / dynamic
    deflate blocks and split IDAT chunks: the oracle restatement and the product's own inflate + unfilter equal stbi_loadf's
    floats under both flip settings."""
    from restir_amd import scene_io
    rng = np.random.default_rng(4)
    obj = tmp_path / "t.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    cam = "Camera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"
    combos = [(ct, dp, il) for ct, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16)))
              for dp in depths for il in (False, True)]
    combos += [(2, 8, False)] * 3                                             # larger pictures: long matches, many blocks
    for i, (ct, dp, il) in enumerate(combos):
        big = i >= len(combos) - 3
        h, w = (int(rng.integers(60, 90)), int(rng.integers(100, 140))) if big else (int(rng.integers(1, 20)), int(rng.integers(1, 40)))
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
        pal = rng.integers(0, 256, (min(256, 1 << dp), 3), dtype=np.uint8) if ct == 3 else None
        smp = rng.integers(0, 1 << dp, (h, w, ch))
        if i % 3 == 0 or big:
            smp[:, w // 2:] = smp[:, w // 2: w // 2 + 1]                      # runs: back-references in the deflate stream
        if big:
            smp[h // 2:] = smp[: h - h // 2]
        p = str(tmp_path / f"c{i}.png")
        scene_io.write_png(p, smp, ct, dp, palette=pal, interlace=il, level=(0, 1, 9)[i % 3] if not big else 9,
                           idat_split=37 if i % 2 else 0, filters=None if i % 4 else int(rng.integers(0, 5)))
        for flip in (0, 1):
            ww, hh = C.c_int(), C.c_int()
            buf = np.zeros(h * w * 3, np.float32)
            assert RL.ref_image_load(p.encode(), flip, C.byref(ww), C.byref(hh), buf.ctypes.data, buf.size) == 0
            ref = buf.reshape(h, w, 3)
            assert bits_equal(sf.load_png(p, bool(flip)), ref), (ct, dp, il, flip)
            scene = tmp_path / "s.txt"
            if flip:
                scene.write_text(f"Material m\nType Lambertian\nBaseColor {p}\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n"
                                 f"Object o\n{obj}\nMaterial m\nScale 1 1 1\n\n" + cam)
            else:
                scene.write_text(f"Object o\n{obj}\nMaterial Null\nScale 1 1 1\n\n" + cam + f"EnvMap {p}\n")
            a = capi.SceneFile(str(scene))
            assert len(a.textures) == 1 and bits_equal(a.textures[0], ref), (ct, dp, il, flip)


@needs_ref
def test_polygon_triangulation_fuzz_against_tinyobj(tmp_path):
    """Polygons with 5..11 corners -- convex, star-shaped, self-intersecting, tilted out of every coordinate plane, with repeated
    corners -- are ear-clipped into the triangles tinyobjloader makes of them, by the oracle restatement and by the product."""
    rng = np.random.default_rng(3)
    for it in range(120):
        lines, faces, nv = ["vn 0 0 1"], [], 0
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.integers(5, 12))
            kind = it % 4
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            rad = rng.uniform(0.3, 1.0, n) if kind in (1, 3) else np.ones(n)
            pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(n)], 1)
            if kind == 2:
                pts = rng.uniform(-1, 1, (n, 3)) * [1, 1, 0.05]
            if kind == 3:
                pts[:, 2] = 0.1 * pts[:, 0]
            rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]
            pts = (pts @ rot.T + rng.uniform(-2, 2, 3)).astype(np.float32)
            if it % 7 == 0:
                pts[1] = pts[0]
            lines += ["v %r %r %r" % tuple(float(x) for x in q) for q in pts]
            faces.append("f " + " ".join(f"{nv + k + 1}//1" for k in range(n)))
            nv += n
        p = tmp_path / "poly.obj"
        p.write_text("\n".join(lines + faces) + "\n")
        cap = 4096
        v = np.zeros((cap, 3), np.float32); nn = np.zeros((cap, 3), np.float32); t = np.zeros((cap, 2), np.float32)
        cnt = RL.ref_obj_load(str(p).encode(), cap, v.ctypes.data, nn.ctypes.data, t.ctypes.data)
        mv, mn, mt = sf.load_obj(str(p))
        assert cnt == len(mv) and bits_equal(v[:cnt], mv), it
        if cnt:
            scene = tmp_path / "poly.txt"
            scene.write_text(f"Object o\n{p}\nMaterial Null\nScale 1 1 1\n\n" + _CAMERA)
            a = capi.SceneFile(str(scene))
            bv, _ = ob.bake_instance((0, 0, 0), (0, 0, 0), (1, 1, 1), v[:cnt], nn[:cnt])
            assert bits_equal(a.vertices.reshape(-1, 3), bv), it


@needs_ref
def test_baking_fuzz_against_glm():
    rng = np.random.default_rng(8)
    for i in range(400):
        t = rng.uniform(-50, 50, 3).astype(np.float32)
        r = rng.uniform(-720, 720, 3).astype(np.float32)
        s = (np.exp(rng.uniform(-4, 4, 3)) * rng.choice([-1.0, 1.0], 3)).astype(np.float32)
        verts = rng.uniform(-10, 10, (16, 3)).astype(np.float32); nrm = rng.normal(size=(16, 3)).astype(np.float32)
        m = np.zeros(16, np.float32); vo = np.zeros(48, np.float32); no = np.zeros(48, np.float32)
        RS.ref_build_transformation_matrix(t, r, s, m)
        RS.ref_bake_instance(t, r, s, 16, verts.reshape(-1), nrm.reshape(-1), vo, no)
        assert bits_equal(ob.build_transformation_matrix(t, r, s).reshape(-1), m)
        v, n = ob.bake_instance(t, r, s, verts, nrm)
        assert bits_equal(v.reshape(-1), vo) and bits_equal(n.reshape(-1), no)


# ---- product (host code of librestir_hip.so) vs oracle and fixture ----------------------------------------------------
def test_product_scene_file_matches_oracle(case_dir):
    path = os.path.join(case_dir, "scene.txt")
    a = capi.SceneFile(path)
    b = sf.load_scene(path)
    assert cases.parsed_equal(a, b) == []
    assert a.num_skipped_objects == 1 and a.iterations == 7 and a.trace_depth == 3 and a.image_name == "case"
    assert a.vertices.shape[0] == 2 + 8 + 5 * 12 + 2 and a.env_map_tex == 4 and len(a.textures) == 5
    assert a.textures[4].max() > 50                          # the Radiance HDR environment map keeps its range
    # instances of one mesh file share the pool entry; "Material Null" appended a default material after the seven named ones
    assert len(a.materials) == 8 and a.material_ids[10 + 24] == 7
    assert a.materials[5]["type"] == 0                      # the unknown type token
    assert a.camera.resolution[0] == 96 and a.camera.fov[1] == 27.0


@pytest.mark.parametrize("key", ["crlf", "cr", "nofinal"])       # ("blanks": whitespace-only lines index tokens[0] of nothing in the reference)
def test_product_line_ending_variants(g, case_dir, key):
    ref = capi.SceneFile(os.path.join(case_dir, "scene.txt"))
    p = os.path.join(case_dir, f"variant_{key}.txt")
    with open(p, "wb") as f:
        f.write(g["variant_" + key].tobytes())
    a, b = capi.SceneFile(p), sf.load_scene(p)
    assert cases.parsed_equal(a, b) == []
    assert cases.parsed_equal(a, ref) == []


@pytest.mark.parametrize("name", ["cube.obj", "poly.obj"])
def test_product_obj_reader_matches_tinyobj_fixture(g, case_dir, tmp_path, name):
    """Quads (cube.obj) and ear-clipped polygons (poly.obj) through a one-object scene with the identity instance."""
    scene = tmp_path / "one.txt"
    scene.write_text("Object o\n" + os.path.join(case_dir, name) + "\nMaterial Null\nScale 1 1 1\n\n" + _CAMERA)
    a = capi.SceneFile(str(scene))
    v, n = ob.bake_instance((0, 0, 0), (0, 0, 0), (1, 1, 1), g["obj_v_" + name], g["obj_n_" + name])
    assert bits_equal(a.vertices.reshape(-1, 3), v) and bits_equal(a.normals.reshape(-1, 3), n)
    assert bits_equal(a.texcoords.reshape(-1, 2), g["obj_t_" + name])
    assert a.env_map_tex == -1 and len(a.textures) == 0 and len(a.materials) == 1


_ONE_TRIANGLE = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n"
_CAMERA = "Camera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"


def _decode_through_scene(tmp_path, image_path, flipped):
    """The library's picture decoder, reached the way a scene reaches it: as a texture (rows flipped) or as the environment map."""
    obj = tmp_path / "one.obj"
    obj.write_text(_ONE_TRIANGLE)
    scene = tmp_path / "one.txt"
    if flipped:
        scene.write_text(f"Material m\nType Lambertian\nBaseColor {image_path}\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n"
                         f"Object o\n{obj}\nMaterial m\nScale 1 1 1\n\n" + _CAMERA)
    else:
        scene.write_text(f"Object o\n{obj}\nMaterial Null\nScale 1 1 1\n\n" + _CAMERA + f"EnvMap {image_path}\n")
    a = capi.SceneFile(str(scene))
    assert len(a.textures) == 1
    return a.textures[0]


def test_product_jpeg_decoder_matches_stb_image_fixture(g, tmp_path):
    """Baseline and progressive JPEG files (4:4:4, 4:2:2, 4:2:0 with optimised tables, 4:1:1, restart markers, grey, RGB-tagged):
    the library's decoder -- Huffman decoding, spectral selection / successive approximation, stb_image's integer IDCT, chroma
    up-sampling and fixed-point YCbCr conversion -- returns stbi_loadf's floats bit for bit.  Arithmetic coding is refused."""
    keys = [k[len("jpeg_file_"):] for k in g.files if k.startswith("jpeg_file_")]
    assert len(keys) >= 11
    for key in keys:
        p = tmp_path / (key + ".jpg")
        p.write_bytes(g["jpeg_file_" + key].tobytes())
        assert bits_equal(_decode_through_scene(tmp_path, p, True), g["jpeg_flip_" + key]), key
        assert bits_equal(_decode_through_scene(tmp_path, p, False), g["jpeg_noflip_" + key]), key
    raw = bytearray(g["jpeg_file_444"].tobytes())
    i = raw.find(b"\xff\xc0")
    raw[i + 1] = 0xc9                                       # claim to be arithmetic-coded
    p = tmp_path / "arith.jpg"
    p.write_bytes(bytes(raw))
    with pytest.raises(capi.RestirHipError, match="unsupported JPEG coding process"):
        _decode_through_scene(tmp_path, p, True)


def test_product_tga_decoder_matches_stb_image_fixture(g, tmp_path):
    """TGA files -- true colour 24 / 32 / 16 bit (5-5-5), grey, grey + alpha, colour-mapped (24-bit and 15-bit maps, indices past
    the map), plain and run-length coded, bottom-up and top-down -- decode to stbi_loadf's floats under both flip settings."""
    keys = [k[len("tga_file_"):] for k in g.files if k.startswith("tga_file_")]
    assert len(keys) >= 7
    for key in keys:
        p = tmp_path / (key + ".TGA")                        # recognised by name, any case
        p.write_bytes(g["tga_file_" + key].tobytes())
        assert bits_equal(_decode_through_scene(tmp_path, p, True), g["tga_flip_" + key]), key
        assert bits_equal(_decode_through_scene(tmp_path, p, False), g["tga_noflip_" + key]), key


def test_product_bmp_decoder_matches_stb_image_fixture(g, tmp_path):
    """BMP files -- 1 / 4 / 8-bit palette, 16-bit 5-5-5 and bit fields, 24-bit, 32-bit with and without masks, header sizes 12 / 40 /
    56 / 108 / 124, bottom-up and top-down -- decode to the floats of the reference's stb_image 2.21, its handling of a 40-byte
    header followed by bit-field masks included."""
    keys = [k[len("bmp_file_"):] for k in g.files if k.startswith("bmp_file_")]
    assert len(keys) >= 13
    for key in keys:
        p = tmp_path / (key + ".bmp")
        p.write_bytes(g["bmp_file_" + key].tobytes())
        assert bits_equal(_decode_through_scene(tmp_path, p, True), g["bmp_flip_" + key]), key
        assert bits_equal(_decode_through_scene(tmp_path, p, False), g["bmp_noflip_" + key]), key
    for key in ("pgm", "ppm_max100"):                        # binary PGM, and a PPM whose maxval is not 255
        p = tmp_path / (key + ".pnm")
        p.write_bytes(g["pnm_file_" + key].tobytes())
        assert bits_equal(_decode_through_scene(tmp_path, p, True), g["pnm_flip_" + key]), key
        assert bits_equal(_decode_through_scene(tmp_path, p, False), g["pnm_noflip_" + key]), key


@needs_ref
def test_jpeg_decoder_fuzz_against_stb_image(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(9)
    n = 0
    for i in range(90):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 100))
        yy, xx = np.mgrid[0:h, 0:w]
        pic = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 11.0), 128 + 100 * np.cos(xx / 5.0), 128 + 90 * np.sin(yy / 3.0)], axis=2)
        pic = np.clip(pic + rng.normal(0, float(rng.choice([0, 10, 60])), pic.shape), 0, 255).astype(np.uint8)
        opt = dict(quality=int(rng.integers(1, 101)))
        grey = i % 7 == 3
        if not grey:
            opt["subsampling"] = [0, 1, 2, "4:1:1"][int(rng.integers(0, 4))]
            if i % 9 == 4:
                opt["keep_rgb"] = True; opt["subsampling"] = 0
        if i % 3 == 0:
            opt["optimize"] = True
        if i % 5 in (1, 2):
            opt["progressive"] = True
        if i % 4 == 1:
            opt["restart_marker_blocks"] = int(rng.integers(1, 9))
        if i % 11 == 5:
            opt["restart_marker_rows"] = 1
        p = str(tmp_path / f"f{i}.jpg")
        (Image.fromarray(pic[..., 0]) if grey else Image.fromarray(pic)).save(p, "JPEG", **opt)
        for flip in (0, 1):
            ww, hh = C.c_int(), C.c_int()
            buf = np.zeros(h * w * 3, np.float32)
            assert RL.ref_image_load(p.encode(), flip, C.byref(ww), C.byref(hh), buf.ctypes.data, buf.size) == 0
            assert bits_equal(_decode_through_scene(tmp_path, p, bool(flip)), buf.reshape(h, w, 3)), (i, opt, flip)
            n += 1
    assert n == 180


def test_product_baking_matches_glm_fixture(g):
    for i in range(len(g["bake_t"])):
        t, r, s = g["bake_t"][i], g["bake_r"][i], g["bake_s"][i]
        assert bits_equal(capi.build_transformation_matrix(t, r, s).reshape(-1), g["bake_matrix"][i]), i
        v, n = capi.bake_instance(t, r, s, g["bake_verts"][i], g["bake_normals"][i])
        assert bits_equal(v, g["bake_verts_out"][i]) and bits_equal(n, g["bake_normals_out"][i]), i


def test_product_scene_file_errors(case_dir, tmp_path):
    cam = "Camera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"

    def load(text):
        p = tmp_path / "s.txt"
        p.write_text(text)
        return capi.SceneFile(str(p))
    with pytest.raises(capi.RestirHipError, match="Error reading from file"):
        capi.SceneFile(str(tmp_path / "nope.txt"))
    with pytest.raises(capi.RestirHipError, match="No mesh data loaded"):            # scene.cpp:192-195
        load(cam)
    with pytest.raises(capi.RestirHipError, match="doesn't exist"):                 # scene.cpp:250-253
        load("Object o\n" + os.path.join(case_dir, "cube.obj") + "\nMaterial nobody\n\n" + cam)
    (tmp_path / "nonormal.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\n")
    with pytest.raises(capi.RestirHipError, match="without a valid normal"):
        load("Object o\n" + str(tmp_path / "nonormal.obj") + "\nMaterial Null\nScale 1 1 1\n\n" + cam)
    (tmp_path / "tex.png").write_bytes(b"\x89PNG\r\n\x1a\n....")
    with pytest.raises(capi.RestirHipError, match="PNG"):
        load("Material m\nType Lambertian\nBaseColor " + str(tmp_path / "tex.png") + "\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n" + cam)
    (tmp_path / "tex.jpg").write_bytes(b"\xff\xd8\xff\xe0....")
    with pytest.raises(capi.RestirHipError, match="JPEG"):
        load("Material m\nType Lambertian\nBaseColor " + str(tmp_path / "tex.jpg") + "\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n" + cam)
    (tmp_path / "tex.bmp").write_bytes(b"BM" + bytes(60))
    with pytest.raises(capi.RestirHipError, match="BMP"):
        load("Material m\nType Lambertian\nBaseColor " + str(tmp_path / "tex.bmp") + "\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n" + cam)
    (tmp_path / "tex.gif").write_bytes(b"GIF89a" + bytes(60))
    with pytest.raises(capi.RestirHipError, match="are decoded here"):
        load("Material m\nType Lambertian\nBaseColor " + str(tmp_path / "tex.gif") + "\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n" + cam)
    with pytest.raises(capi.RestirHipError, match="only OBJ"):
        load("Object o\nmesh.gltf\nMaterial Null\n\n" + cam)
    with pytest.raises(capi.RestirHipError, match="malformed number"):
        load("Object o\n" + os.path.join(case_dir, "cube.obj") + "\nMaterial Null\nScale one 1 1\n\n" + cam)


def test_export_scene_data_round_trip(tmp_path):
    """scene_io.export_scene_data writes a generated scene in the reference's format; reading it back gives the same triangles
    (identity instance: vertices unchanged, normals re-normalised exactly as the reference would)."""
    from restir_amd import scene_io, scenes
    sd = scenes.sponza_class(1, 0.01)
    path = scene_io.export_scene_data(sd, str(tmp_path), 160, 90, name="sp")
    a = capi.SceneFile(path)
    assert a.vertices.shape == sd.vertices.shape and np.array_equal(a.vertices, sd.vertices)        # -0 -> +0 allowed: value equality
    assert np.array_equal(a.material_ids, sd.material_ids) and bits_equal(a.texcoords, sd.texcoords)
    assert np.abs(a.normals - sd.normals).max() < 1e-6
    assert a.materials[:len(sd.materials)].tobytes() == np.asarray(sd.materials).tobytes()
    b = sf.load_scene(path)
    assert cases.parsed_equal(a, b) == []


# ---- GPU: a scene from disk renders the oracle's frames ----------------------------------------------------------------
@pytest.mark.gpu
def test_scene_from_disk_renders_oracle_frames(case_dir):
    from tests.common import HipRenderer, OracleRenderer
    from restir_amd.scenes import SceneData
    path = os.path.join(case_dir, "scene.txt")
    a, b = capi.SceneFile(path), sf.load_scene(path)
    assert cases.parsed_equal(a, b) == []

    def scene_data(p):
        sd = SceneData.__new__(SceneData)
        sd.name = "from_disk"
        sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials = p.vertices, p.normals, p.texcoords, p.material_ids, p.materials
        sd.textures, sd.env_map_tex = list(p.textures), p.env_map_tex
        cam = p.camera
        sd.camera = lambda w, h: cam                       # the parsed camera, not one made from camera_args
        return sd
    W, H = a.camera.resolution[0], a.camera.resolution[1]
    ob.set_libm_mode(1)                                    # textured scene: the correctly rounded libm mode the device is pinned to
    try:
        orc = OracleRenderer(scene_data(b), W, H)
        hip = HipRenderer(capi, scene_data(a), W, H)
        for frame in range(4):
            if frame == 2:
                orc.set_camera_position((0.05, 1.0, 3.45)); hip.set_camera_position((0.05, 1.0, 3.45))
            fo = orc.frame(True).copy()
            fh = hip.frame(True)
            assert bits_equal(fo, fh), frame
            assert orc.rays == hip.rays, frame
        assert np.isfinite(fo).all() and (fo.sum(axis=1) > 0).mean() > 0.8
    finally:
        ob.set_libm_mode(0)


@pytest.mark.gpu
def test_headless_viewer_renders_scene_file(case_dir, tmp_path):
    """main()'s own start-up sequence -- `scene = new Scene(argv[1]); ... scene->buildDevData();` (src/main.cpp:60-83) -- through
    restir_compat.h: the viewer loads the scene text itself and its RGBA8 frame equals the oracle's."""
    import subprocess
    from tests.common import OracleRenderer
    from restir_amd.scenes import SceneData
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "restir_amd", "host", "headless_viewer")
    assert os.path.exists(exe), "build it with make -C restir_amd/csrc"
    path = os.path.join(case_dir, "scene.txt")
    frames, reuse = 3, 1
    subprocess.check_call([exe, path, str(frames), str(reuse), str(tmp_path / "out.ppm")], cwd=str(tmp_path))
    b = sf.load_scene(path)
    W, H = b.camera.resolution[0], b.camera.resolution[1]
    data = open(tmp_path / "out.ppm", "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H * W, 3)
    sd = SceneData.__new__(SceneData)
    sd.name = "from_disk"
    sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials = b.vertices, b.normals, b.texcoords, b.material_ids, b.materials
    sd.textures, sd.env_map_tex = list(b.textures), b.env_map_tex
    sd.camera = lambda w, h: b.camera
    ob.set_libm_mode(1)
    try:
        o = OracleRenderer(sd, W, H)
        for _ in range(frames):
            img = o.frame(reuse)
        ref = ob.send_image_to_pbo(img, W, H, 2, 1.0)[:, :3]
    finally:
        ob.set_libm_mode(0)
    diff = np.abs(ref.astype(np.int32) - got.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff > 0) <= 1e-3
