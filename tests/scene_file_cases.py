"""A small scene in the reference's on-disk format (scene text + OBJ meshes + PPM / PNG / HDR images), used by the scene-file tests.

Everything is generated here (restir_amd.scene_io writers + hand-written OBJ text that exercises quads, negative and
mixed index styles, groups and the number syntax); nothing comes from the reference."""
import os

import numpy as np

from restir_amd import scene_io, scenes

CUBE_OBJ = """# unit cube: quads split along the shorter diagonal, mixed index styles
mtllib nothing.mtl
o cube
v -0.5 -0.5 0.5
v 0.5 -0.5 0.5
v 0.5 0.5 0.5
v -0.5 0.5 0.5
v -0.5 -0.5 -0.5
v 0.5 -0.5 -0.5
v 0.5 0.5 -0.5
v -0.5 0.5 -0.5
vn 0 0 1
vn 0 0 -1
vn 1 0 0
vn -1 0 0
vn 0 1 0
vn 0 -1 0
vt 0.0 0.0
vt 1.0 0.0
vt 1.0 1.0 0.0
vt 0.0 1.0
s off
g front
f 1/1/1 2/2/1 3/3/1 4/4/1
f 6/1/2 5/2/2 8/3/2 7/4/2
usemtl foo
g side
f 2/1/3 6/2/3 7/3/3 3/4/3
f -4/1/4 -8/2/4 -5/3/4 -1/4/4
f 4/1/5 3/2/5 7/3/5
f 4/1/5 7/3/5 8/4/5
f 5/1/6 6/2/6 2/3/6 1/4/6
"""

# number syntax (tinyobj's own decimal parser, not strtof) and a skewed quad whose diagonals differ
NUMBERS_OBJ = """v 1e-3 .5 -.25
v +3 1. 12345.678901234
v 0.1 0.2 0.3
v 1.17549435e-38 3.4028234e38 -7.00649232e-46
v 16777217 0.30000001192092896 2.5E+2
v 0.333333333333333333333 6.02214076e23 -1.5e-7
v 0.7 -0.1 0.05
v 0.123456789 9.87654321 5e0
vn 0.57735026 0.57735026 0.57735026
vn 0 1 0
f 1//1 2//1 3//2
f 4//2 5//2 6//1 7//1
f 8//2 7//1 6//2 5//1
f 3//1 7//2 8//2
"""


# polygons with more than four corners: tinyobjloader's ear clipping (convex, concave, tilted, with a repeated corner)
POLY_OBJ = """vn 0 0 1
vn 0.6 0 0.8
v 0 0 0
v 2 0 0
v 2.5 1 0
v 1 2 0
v -0.5 1 0
f 1//1 2//1 3//1 4//1 5//1
v 0 0 1
v 1 0.2 1
v 2 0 1
v 1.6 1 1
v 2 2 1
v 1 1.7 1
v 0 2 1
v 0.4 1 1
f 6//1 7//1 8//1 9//1 10//1 11//1 12//1 13//1
v 0 0 2
v 1 0 2.75
v 1 0 2.75
v 2 1 3.5
v 1 2 2.75
v 0 1 2
f 14//2 15//2 16//2 17//2 18//2 19//2
f -6//2 -4//2 -3//2 -2//2 -1//2
"""


def _noise_image(rng, h, w):
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


def write_case(directory, newline="\n", seed=11):
    """Writes the case into `directory`; returns the scene file path."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed)
    for name, (h, w) in (("base.ppm", (16, 32)), ("metal.ppm", (8, 8)), ("rough.ppm", (4, 8)), ("nrm.ppm", (16, 16))):
        img = _noise_image(rng, h, w)
        if name == "nrm.ppm":                                      # a plausible tangent-space normal map
            img[..., 2] = 255
            img[..., :2] = 128 + rng.integers(-40, 41, (h, w, 2))
        scene_io.write_ppm(os.path.join(directory, name), img)
    # PNG textures: the metallic map as an interlaced 8-bit grey picture, the roughness map as a 4-bit palette picture, and an
    # RGBA / 16-bit pair for the decoder only
    scene_io.write_png(os.path.join(directory, "metal.png"), rng.integers(0, 256, (8, 8)), 0, 8, interlace=True)
    pal = rng.integers(30, 230, (16, 3), dtype=np.uint8)
    scene_io.write_png(os.path.join(directory, "rough.png"), rng.integers(0, 16, (4, 8)), 3, 4, palette=pal)
    scene_io.write_png(os.path.join(directory, "rgba.png"), rng.integers(0, 256, (9, 13, 4)), 6, 8, idat_split=64)
    scene_io.write_png(os.path.join(directory, "rgb16.png"), rng.integers(0, 65536, (5, 7, 3)), 2, 16, interlace=True)
    env = (40 + 60 * np.linspace(1.0, 0.2, 16)[:, None, None] * np.array([0.6, 0.8, 1.0])).astype(np.uint8) + np.zeros((16, 32, 3), np.uint8)
    env[3:5, 8:10] = 255                                            # a bright patch above the open front
    env = (env.astype(np.int32) + rng.integers(0, 4, env.shape)).clip(0, 255).astype(np.uint8)
    scene_io.write_ppm(os.path.join(directory, "env.ppm"), env)
    # the environment map the scene uses: HDR, with a sun far above 1.0 (run-length coded scan lines); a second picture in the
    # flat layout and a narrow one (flat by width) for the decoder
    sky = (0.3 + 0.5 * np.linspace(1.0, 0.1, 16))[:, None, None] * np.array([0.6, 0.8, 1.0]) + np.zeros((16, 32, 3))
    sky[3:5, 8:10] = (90.0, 80.0, 60.0)
    sky[10:, :] *= 0.25
    sky += 0.01 * rng.uniform(size=sky.shape) * (np.arange(32)[None, :, None] % 4 == 0)
    scene_io.write_hdr(os.path.join(directory, "env.hdr"), sky)
    scene_io.write_hdr(os.path.join(directory, "flat.hdr"), sky * 3.7e-5, rle=False)
    scene_io.write_hdr(os.path.join(directory, "narrow.hdr"), rng.uniform(0, 5e4, (9, 5, 3)))

    sd = scenes.cornell_box()
    # triangles 0..9 = the five walls, 10..33 = the two boxes (replaced by instances of cube.obj), 34.. = the light
    v = sd.vertices
    tc = np.zeros((v.shape[0], 3, 2), np.float32)
    tc[:, :, 0] = v[:, :, 0] * 0.7 + v[:, :, 2] * 0.3 + 0.13
    tc[:, :, 1] = v[:, :, 1] * 0.6 - v[:, :, 2] * 0.2 - 0.21
    scene_io.write_obj(os.path.join(directory, "floor.obj"), sd.vertices[:2], sd.normals[:2], tc[:2])
    scene_io.write_obj(os.path.join(directory, "walls.obj"), sd.vertices[2:10], sd.normals[2:10], tc[2:10])
    scene_io.write_obj(os.path.join(directory, "light.obj"), sd.vertices[34:], sd.normals[34:], None)
    with open(os.path.join(directory, "cube.obj"), "w") as f:
        f.write(CUBE_OBJ)
    with open(os.path.join(directory, "numbers.obj"), "w") as f:
        f.write(NUMBERS_OBJ)
    with open(os.path.join(directory, "poly.obj"), "w") as f:
        f.write(POLY_OBJ)

    mats = [("white", dict(type=0, baseColor="base.ppm")),
            ("red", dict(type=0, baseColor="Procedural")),
            ("metal", dict(type=1, baseColor=(0.9, 0.8, 0.5), metallic="metal.png", roughness="rough.png", normalMap="nrm.ppm")),
            ("satin", dict(type="MetallicWorkflow", baseColor=(0.3, 0.5, 0.8), metallic=0.25, roughness=0.35)),
            ("glass", dict(type=2, baseColor=(0.95, 0.95, 0.95), ior=1.45)),
            ("odd", dict(type="NoSuchType", baseColor=(0.2, 0.7, 0.3))),           # unknown type token -> Lambertian
            ("lamp", dict(type=4, baseColor=(10, 10, 10)))]
    objs = [dict(name="floor", file="floor.obj", material="red"),
            dict(name="walls", file="walls.obj", material="white"),
            dict(name="b1", file="cube.obj", material="metal", translate=(-0.3, 0.6, -0.3), rotate=(0, 17.5, 0), scale=(0.55, 1.2, 0.55)),
            dict(name="b2", file="cube.obj", material="glass", translate=(0.35, 0.3, 0.3), rotate=(10, -20, 5), scale=(0.5, 0.6, 0.5)),
            dict(name="b3", file="cube.obj", material=None, translate=(0.0, 0.1, 0.7), rotate=(0, 45, 0), scale=(0.2, 0.2, 0.2)),
            dict(name="b4", file="cube.obj", material="satin", translate=(0.6, 0.15, -0.5), rotate=(33, 0, 12), scale=(0.3, 0.3, 0.3)),
            dict(name="b5", file="cube.obj", material="odd", translate=(-0.65, 0.12, 0.55), rotate=(0, 0, 0), scale=(0.24, 0.24, 0.24)),
            dict(name="gone", file="missing.obj", material="red"),                    # "[Fail to load, skipped]"
            dict(name="lamp", file="light.obj", material="lamp")]
    cam = dict(width=96, height=64, fov_y=27.0, position=(0, 1, 3.5), rotation=(-90, 0, 0), sample=7, depth=3, file="case")
    path = os.path.join(directory, "scene.txt")
    scene_io.write_scene(path, mats, objs, cam, env_map="env.hdr", newline=newline)
    return path


CASE_FILES = ("scene.txt", "floor.obj", "walls.obj", "light.obj", "cube.obj", "numbers.obj", "poly.obj",
              "base.ppm", "metal.ppm", "rough.ppm", "nrm.ppm", "env.ppm", "env.hdr", "flat.hdr", "narrow.hdr",
              "metal.png", "rough.png", "rgba.png", "rgb16.png")


def parsed_equal(a, b):
    """Bit equality of two parsed scenes (capi.SceneFile / oracle.scene_format.ParsedScene); returns the list of differing fields."""
    bad = []
    for k in ("vertices", "normals", "texcoords"):
        x, y = np.ascontiguousarray(getattr(a, k), np.float32), np.ascontiguousarray(getattr(b, k), np.float32)
        if x.shape != y.shape or not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
            bad.append(k)
    if not np.array_equal(a.material_ids, b.material_ids):
        bad.append("material_ids")
    if a.materials.tobytes() != b.materials.tobytes():
        bad.append("materials")
    if len(a.textures) != len(b.textures) or not all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a.textures, b.textures)):
        bad.append("textures")
    if a.env_map_tex != b.env_map_tex:
        bad.append("env_map_tex")
    if bytes(a.camera) != bytes(b.camera):
        bad.append("camera")
    for k in ("iterations", "trace_depth", "image_name", "num_skipped_objects"):
        if getattr(a, k) != getattr(b, k):
            bad.append(k)
    return bad
