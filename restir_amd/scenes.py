"""Procedural, seeded stand-ins for the scenes BASELINE.json names (the reference ships no scenes,
meshes or scene files -- SURVEY.md section 6).  Every generator returns a baked triangle soup in the
layout Scene::buildDevData produces (src/scene.cpp:159-190): de-indexed vertices / normals /
texcoords (3 per triangle), one material id per triangle, and the material table.

    cornell_box()            config 1/2: 34 triangles + a 2-triangle ceiling light, Le = 10
    sponza_class(seed=1)     config 3/4: 262 144 triangles, 1 024 emissive (512 lantern quads)
    bistro_class(seed=2)     config 5:   ~2.8 M triangles, 10 240 emissive

Cameras use the reference's conventions (src/sceneStructs.h:88-102): rotation = (yaw, pitch, roll)
in degrees, fov.y = vertical HALF angle in degrees.
"""
import numpy as np

from .ctypes_structs import (LAMBERTIAN, METALLIC_WORKFLOW, LIGHT, make_camera, make_materials)


class TriangleSoup:
    def __init__(self):
        self.v, self.n, self.m = [], [], []

    def add(self, verts, normals, mat):
        """verts, normals: (k,3,3) float arrays; mat: int or (k,) int array."""
        verts = np.asarray(verts, np.float32).reshape(-1, 3, 3)
        normals = np.asarray(normals, np.float32).reshape(-1, 3, 3)
        self.v.append(verts)
        self.n.append(normals)
        self.m.append(np.broadcast_to(np.asarray(mat, np.int32), (verts.shape[0],)).copy())

    def add_flat(self, verts, mat):
        verts = np.asarray(verts, np.float32).reshape(-1, 3, 3)
        fn = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = fn / np.where(ln > 0, ln, 1)
        self.add(verts, np.repeat(fn[:, None, :], 3, axis=1), mat)

    def add_quad(self, a, b, c, d, mat):
        """Two triangles (a,b,c), (a,c,d); geometric normal = cross(b-a, c-a)."""
        self.add_flat(np.array([[a, b, c], [a, c, d]], np.float32), mat)

    def add_grid(self, fn_pos, nu, nv, mat, flip=False, smooth=True):
        """Tessellate a parametric surface p(u,v), u,v in [0,1], into nu x nv x 2 triangles."""
        u = np.linspace(0.0, 1.0, nu + 1)
        v = np.linspace(0.0, 1.0, nv + 1)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        p = fn_pos(uu, vv).astype(np.float64)                     # (nu+1, nv+1, 3)
        eps = 1e-4
        du = (fn_pos(np.clip(uu + eps, 0, 1), vv) - fn_pos(np.clip(uu - eps, 0, 1), vv))
        dv = (fn_pos(uu, np.clip(vv + eps, 0, 1)) - fn_pos(uu, np.clip(vv - eps, 0, 1)))
        nrm = np.cross(du, dv)
        ln = np.linalg.norm(nrm, axis=2, keepdims=True)
        nrm = nrm / np.where(ln > 0, ln, 1)
        if flip:
            nrm = -nrm
        p00, p10, p11, p01 = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
        n00, n10, n11, n01 = nrm[:-1, :-1], nrm[1:, :-1], nrm[1:, 1:], nrm[:-1, 1:]
        if flip:
            t1 = np.stack([p00, p11, p10], axis=2); m1 = np.stack([n00, n11, n10], axis=2)
            t2 = np.stack([p00, p01, p11], axis=2); m2 = np.stack([n00, n01, n11], axis=2)
        else:
            t1 = np.stack([p00, p10, p11], axis=2); m1 = np.stack([n00, n10, n11], axis=2)
            t2 = np.stack([p00, p11, p01], axis=2); m2 = np.stack([n00, n11, n01], axis=2)
        verts = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0)
        norms = np.concatenate([m1.reshape(-1, 3, 3), m2.reshape(-1, 3, 3)], axis=0)
        if smooth:
            self.add(verts, norms, mat)
        else:
            self.add_flat(verts, mat)

    def finish(self):
        v = np.ascontiguousarray(np.concatenate(self.v, axis=0), np.float32)
        n = np.ascontiguousarray(np.concatenate(self.n, axis=0), np.float32)
        m = np.ascontiguousarray(np.concatenate(self.m, axis=0), np.int32)
        t = np.zeros((v.shape[0], 3, 2), np.float32)
        return v, n, t, m

    def count(self):
        return int(sum(x.shape[0] for x in self.v))


class SceneData:
    def __init__(self, name, soup, materials, camera_args, textures=(), env_map_tex=-1):
        self.name = name
        self.vertices, self.normals, self.texcoords, self.material_ids = soup.finish()
        self.materials = materials
        self.camera_args = camera_args       # dict(position=, rotation=, fov_y=, focal_dist=)
        self.textures = list(textures)       # (H, W, 3) float32 linear-RGB images (what Image holds, src/image.h)
        self.env_map_tex = env_map_tex       # Scene::envMapTexId

    @property
    def num_prims(self):
        return self.vertices.shape[0]

    def camera(self, width, height):
        return make_camera(width, height, **self.camera_args)


def _box(soup, lo, hi, mat, yaw_deg=0.0):
    lo = np.asarray(lo, np.float64); hi = np.asarray(hi, np.float64)
    c = (lo + hi) / 2; h = (hi - lo) / 2
    a = np.radians(yaw_deg)
    rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    def P(sx, sy, sz):
        return c + rot @ (h * np.array([sx, sy, sz]))
    faces = [  # outward-facing windings
        (P(-1, -1, 1), P(1, -1, 1), P(1, 1, 1), P(-1, 1, 1)),     # +z
        (P(1, -1, -1), P(-1, -1, -1), P(-1, 1, -1), P(1, 1, -1)), # -z
        (P(1, -1, 1), P(1, -1, -1), P(1, 1, -1), P(1, 1, 1)),     # +x
        (P(-1, -1, -1), P(-1, -1, 1), P(-1, 1, 1), P(-1, 1, -1)), # -x
        (P(-1, 1, 1), P(1, 1, 1), P(1, 1, -1), P(-1, 1, -1)),     # +y
        (P(-1, -1, -1), P(1, -1, -1), P(1, -1, 1), P(-1, -1, 1)), # -y
    ]
    for f in faces:
        soup.add_quad(*f, mat)


def cornell_box():
    """Config 1/2: 5 walls + 2 boxes = 34 triangles, white/red/green Lambertian,
    one 2-triangle ceiling light Le=(10,10,10); camera Eye 0 1 3.5, Rotation -90 0 0, FovY 19.5."""
    mats = make_materials([
        dict(type=LAMBERTIAN, baseColor=(0.73, 0.73, 0.73)),   # 0 white
        dict(type=LAMBERTIAN, baseColor=(0.65, 0.05, 0.05)),   # 1 red
        dict(type=LAMBERTIAN, baseColor=(0.12, 0.45, 0.15)),   # 2 green
        dict(type=LIGHT, baseColor=(10.0, 10.0, 10.0)),        # 3 light
    ])
    s = TriangleSoup()
    # room x in [-1,1], y in [0,2], z in [-1,1]; inward-facing windings
    s.add_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), 0)      # floor, normal +y
    s.add_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1), 0)      # ceiling, normal -y
    s.add_quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1), 0)    # back, normal +z
    s.add_quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1), 1)    # left (red), normal +x
    s.add_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1), 2)        # right (green), normal -x
    _box(s, (-0.7, 0.0, -0.65), (-0.1, 1.2, -0.05), 0, yaw_deg=18.0)   # tall box
    _box(s, (0.1, 0.0, 0.0), (0.7, 0.6, 0.6), 0, yaw_deg=-17.0)        # short box
    # ceiling light, facing down (geometric normal -y), just below the ceiling
    s.add_quad((-0.25, 1.98, -0.25), (0.25, 1.98, -0.25), (0.25, 1.98, 0.25), (-0.25, 1.98, 0.25), 3)
    assert s.count() == 36
    return SceneData("cornell", s, mats, dict(position=(0.0, 1.0, 3.5), rotation=(-90.0, 0.0, 0.0), fov_y=19.5, focal_dist=1.0))


def cornell_textured(seed=3, env=True):
    """The Cornell box of config 1/2 with everything getTexturedMaterialAndSurface and the environment-map light
    can do (src/scene.h:78-99,358-403): a base-colour map, the procedural texture, metallic / roughness / normal
    maps on a MetallicWorkflow box, a normal map on a Lambertian box, and (env=True) an HDR environment map with a
    small bright sun that is visible through the open front and is the last entry of the light sampler.
    Texture coordinates are a planar projection that leaves [0,1] (exercises the wrap of linearSample)."""
    base = cornell_box()
    rng = np.random.Generator(np.random.PCG64(seed))
    checker = np.indices((16, 32)).sum(0) % 2
    tex_base = (0.25 + 0.6 * checker[..., None] * np.array([1.0, 0.9, 0.7]) + 0.1 * rng.uniform(size=(16, 32, 3))).astype(np.float32)
    tex_metal = np.repeat(rng.uniform(0.0, 1.0, (16, 16, 1)), 3, axis=2).astype(np.float32)
    tex_rough = np.repeat(rng.uniform(0.15, 0.9, (8, 8, 1)), 3, axis=2).astype(np.float32)
    bump = rng.uniform(-0.2, 0.2, (16, 16, 2))
    tex_normal = np.concatenate([0.5 + bump, np.full((16, 16, 1), 1.0)], axis=2).astype(np.float32)
    textures = [tex_base, tex_metal, tex_rough, tex_normal]
    env_id = -1
    if env:
        h, w = 32, 64
        yy, xx = np.mgrid[0:h, 0:w]
        sky = 0.3 + 0.4 * (1.0 - yy / h)
        envmap = np.stack([sky * 0.6, sky * 0.8, sky * 1.0], axis=2)
        envmap[6:9, 14:18] = (60.0, 55.0, 40.0)                   # the sun
        envmap += 0.02 * rng.uniform(size=envmap.shape)
        textures.append(envmap.astype(np.float32))
        env_id = len(textures) - 1
    mats = make_materials([
        dict(type=LAMBERTIAN, baseColor=(0.73, 0.73, 0.73)),                       # 0 walls: base-colour map
        dict(type=LAMBERTIAN, baseColor=(0.65, 0.05, 0.05)),                       # 1 red wall: procedural texture
        dict(type=LAMBERTIAN, baseColor=(0.12, 0.45, 0.15)),                       # 2 green wall: plain
        dict(type=LIGHT, baseColor=(10.0, 10.0, 10.0)),                            # 3 light
        dict(type=METALLIC_WORKFLOW, baseColor=(0.9, 0.8, 0.5), metallic=0.5, roughness=0.4),   # 4 tall box: all maps
        dict(type=LAMBERTIAN, baseColor=(0.7, 0.7, 0.9)),                          # 5 short box: normal map only
    ])
    mats[0]["baseColorMapId"] = 0
    mats[1]["baseColorMapId"] = -2
    mats[4]["baseColorMapId"] = 0; mats[4]["metallicMapId"] = 1; mats[4]["roughnessMapId"] = 2; mats[4]["normalMapId"] = 3
    mats[5]["normalMapId"] = 3
    sd = SceneData.__new__(SceneData)
    sd.name = "cornell_textured"
    sd.vertices, sd.normals, sd.material_ids = base.vertices, base.normals, base.material_ids.copy()
    # triangles 10..21 = tall box, 22..33 = short box (cornell_box order: 5 quads, 2 boxes, light)
    sd.material_ids[10:22] = 4
    sd.material_ids[22:34] = 5
    v = sd.vertices
    t = np.zeros((v.shape[0], 3, 2), np.float32)
    t[:, :, 0] = v[:, :, 0] * 0.7 + v[:, :, 2] * 0.3 + 0.13
    t[:, :, 1] = v[:, :, 1] * 0.6 - v[:, :, 2] * 0.2 - 0.21
    sd.texcoords = np.ascontiguousarray(t)
    sd.materials = mats
    sd.camera_args = dict(base.camera_args, fov_y=27.0)      # wide enough to see past the box: environment-map pixels
    sd.textures = textures
    sd.env_map_tex = env_id
    return sd


def _column(soup, cx, cz, radius, height, nseg, nring, mat, flute=0.03):
    def pos(u, v):
        ang = u * 2 * np.pi
        r = radius * (1.0 + flute * np.cos(ang * 12)) * (1.0 - 0.12 * v)
        return np.stack([cx + r * np.cos(ang), v * height, cz + r * np.sin(ang)], axis=-1)
    soup.add_grid(pos, nseg, nring, mat, flip=True)


def _lantern_quads(soup, rng, centers, mat0, area_lo=0.01, area_hi=0.05):
    """One small emissive quad (2 triangles) per centre, facing down-and-inward."""
    k = centers.shape[0]
    area = rng.uniform(area_lo, area_hi, k)        # area of one triangle (m^2)
    aspect = rng.uniform(0.6, 1.6, k)
    w = np.sqrt(2 * area * aspect); h = np.sqrt(2 * area / aspect)
    for i in range(k):
        c = centers[i]
        tilt = rng.uniform(-0.5, 0.5)
        yaw = rng.uniform(0, 2 * np.pi)
        ux = np.array([np.cos(yaw), 0.0, np.sin(yaw)])
        uz = np.array([-np.sin(yaw) * np.cos(tilt), np.sin(tilt), np.cos(yaw) * np.cos(tilt)])
        a = c - ux * w[i] / 2 - uz * h[i] / 2
        b = c + ux * w[i] / 2 - uz * h[i] / 2
        cc = c + ux * w[i] / 2 + uz * h[i] / 2
        d = c - ux * w[i] / 2 + uz * h[i] / 2
        nrm = np.cross(b - a, cc - a)
        if nrm[1] > 0:     # make the emitting side face downward
            soup.add_quad(a, d, cc, b, mat0 + i)
        else:
            soup.add_quad(a, b, cc, d, mat0 + i)


def sponza_class(seed=1, scale=1.0):
    """Config 3/4: colonnaded hall (fluted columns, barrel vault, floor, walls, drapes) with 512
    emissive lantern quads = 1 024 emissive triangles, radiance log-uniform in [1,50], triangle area
    0.01-0.05 m^2.  scale=1.0 gives exactly 262 144 triangles (BVHSize 524 287); smaller scales
    shrink the tessellation for CPU-sized tests (light count scales too)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_quads = max(4, int(round(512 * scale)))
    base = [
        dict(type=LAMBERTIAN, baseColor=(0.70, 0.66, 0.58)),                       # 0 stone
        dict(type=LAMBERTIAN, baseColor=(0.45, 0.40, 0.35)),                       # 1 floor
        dict(type=LAMBERTIAN, baseColor=(0.55, 0.10, 0.10)),                       # 2 red drape
        dict(type=LAMBERTIAN, baseColor=(0.10, 0.25, 0.50)),                       # 3 blue drape
        dict(type=METALLIC_WORKFLOW, baseColor=(0.9, 0.7, 0.3), metallic=0.8, roughness=0.35),  # 4 brass
    ]
    rad = np.exp(rng.uniform(np.log(1.0), np.log(50.0), n_quads))
    tint = rng.uniform(0.75, 1.0, (n_quads, 3))
    lights = [dict(type=LIGHT, baseColor=tuple((rad[i] * tint[i]).tolist())) for i in range(n_quads)]
    mats = make_materials(base + lights)

    q = lambda n: max(2, int(round(n * np.sqrt(scale))))
    s = TriangleSoup()
    half_w, half_l, wall_h, vault_r = 8.0, 20.0, 7.0, 8.0
    # floor (slightly wavy flagstones), normal up
    s.add_grid(lambda u, v: np.stack([-half_w + 2 * half_w * u,
                                      0.02 * np.sin(u * 60) * np.sin(v * 150),
                                      half_l - 2 * half_l * v], -1), q(128), q(128), 1)
    # side walls, normals inward
    s.add_grid(lambda u, v: np.stack([np.full_like(u, -half_w) + 0.05 * np.sin(v * 40) * np.sin(u * 90),
                                      wall_h * v, -half_l + 2 * half_l * u], -1), q(64), q(64), 0)
    s.add_grid(lambda u, v: np.stack([np.full_like(u, half_w) - 0.05 * np.sin(v * 40) * np.sin(u * 90),
                                      wall_h * v, half_l - 2 * half_l * u], -1), q(64), q(64), 0)
    # end walls
    s.add_grid(lambda u, v: np.stack([-half_w + 2 * half_w * u, (wall_h + vault_r) * v,
                                      np.full_like(u, -half_l)], -1), q(32), q(32), 0)
    s.add_grid(lambda u, v: np.stack([half_w - 2 * half_w * u, (wall_h + vault_r) * v,
                                      np.full_like(u, half_l)], -1), q(32), q(32), 0)
    # barrel vault with coffers, normal inward (down)
    def vault(u, v):
        ang = np.pi * u
        r = vault_r * (1.0 - 0.015 * (np.sin(u * 50) * np.sin(v * 120) > 0.3))
        return np.stack([-r * np.cos(ang), wall_h + r * np.sin(ang), -half_l + 2 * half_l * v], -1)
    s.add_grid(vault, q(128), q(256), 0)
    # two rows of fluted columns
    ncol = 10
    for row in (-1, 1):
        for k in range(ncol):
            z = -half_l + (k + 0.5) * (2 * half_l / ncol)
            _column(s, row * 4.0, z, 0.5, 6.0, q(64), q(48), 4 if (k % 5 == 2) else 0)
    # drapes hanging between columns
    for i, (x, z0, m) in enumerate([(-4.0, -14.0, 2), (4.0, -6.0, 3), (-4.0, 2.0, 3), (4.0, 10.0, 2)]):
        def drape(u, v, x=x, z0=z0, i=i):
            return np.stack([x + 0.25 * np.sin(u * 25 + i) * (0.3 + v), 5.8 - 3.5 * v,
                             z0 + 3.0 * u], -1)
        s.add_grid(drape, q(48), q(48), m, flip=(x > 0))
    # lanterns: near columns and along the nave at varying heights
    centers = np.stack([rng.uniform(-7.0, 7.0, n_quads), rng.uniform(2.5, 6.5, n_quads),
                        rng.uniform(-half_l + 1, half_l - 1, n_quads)], axis=1)
    _lantern_quads(s, rng, centers, len(base))
    # filler "debris" triangles on the floor to land on the exact triangle budget
    target = int(round(262144 * scale))
    fill = target - s.count()
    if fill > 0:
        c = np.stack([rng.uniform(-7.5, 7.5, fill), np.full(fill, 0.03), rng.uniform(-19.5, 19.5, fill)], 1)
        d = rng.uniform(-0.08, 0.08, (fill, 3, 3)); d[:, :, 1] = np.abs(d[:, :, 1]) * 0.5
        s.add_flat(c[:, None, :] + d, 1)
    return SceneData(f"sponza_class_seed{seed}", s, mats,
                     dict(position=(0.5, 2.2, 17.0), rotation=(-92.0, -2.0, 0.0), fov_y=30.0, focal_dist=1.0))


def bistro_class(seed=2, scale=1.0):
    """Config 5: street canyon with relief facades, awnings, furniture and 5 120 emissive quads
    (10 240 emissive triangles).  scale=1.0 gives 2 830 336 triangles ("~2.8 M")."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_quads = max(8, int(round(5120 * scale)))
    base = [
        dict(type=LAMBERTIAN, baseColor=(0.62, 0.58, 0.52)),
        dict(type=LAMBERTIAN, baseColor=(0.30, 0.30, 0.32)),
        dict(type=LAMBERTIAN, baseColor=(0.55, 0.22, 0.15)),
        dict(type=LAMBERTIAN, baseColor=(0.20, 0.35, 0.22)),
        dict(type=METALLIC_WORKFLOW, baseColor=(0.8, 0.8, 0.85), metallic=0.9, roughness=0.25),
    ]
    rad = np.exp(rng.uniform(np.log(1.0), np.log(50.0), n_quads))
    tint = rng.uniform(0.7, 1.0, (n_quads, 3))
    lights = [dict(type=LIGHT, baseColor=tuple((rad[i] * tint[i]).tolist())) for i in range(n_quads)]
    mats = make_materials(base + lights)
    q = lambda n: max(2, int(round(n * np.sqrt(scale))))
    s = TriangleSoup()
    half_w, half_l, h = 7.0, 60.0, 18.0
    # cobbled street
    s.add_grid(lambda u, v: np.stack([-half_w + 2 * half_w * u,
                                      0.03 * np.sin(u * 180) * np.sin(v * 1500),
                                      half_l - 2 * half_l * v], -1), q(352), q(1408), 1)
    # facades with window relief
    def facade(side):
        def f(u, v):
            relief = 0.25 * ((np.sin(u * 2 * np.pi * 40) > 0.2) & (np.sin(v * 2 * np.pi * 6) > 0.0))
            x = side * (half_w - relief)
            z = (-half_l + 2 * half_l * u) if side < 0 else (half_l - 2 * half_l * u)
            return np.stack([x + 0 * u, h * v, z], -1)
        return f
    s.add_grid(facade(-1), q(1280), q(320), 0)
    s.add_grid(facade(1), q(1280), q(320), 0)
    # awnings
    for k in range(24):
        side = -1 if k % 2 == 0 else 1
        z0 = -half_l + 4 + (k // 2) * 9.5
        def awn(u, v, side=side, z0=z0, k=k):
            return np.stack([side * (half_w - 0.3 - 1.8 * v), 3.4 - 0.6 * v + 0.05 * np.sin(u * 40),
                             z0 + 4.0 * u], -1)
        s.add_grid(awn, q(64), q(32), 2 + (k % 2), flip=(side < 0))
    # street furniture: poles (metal) and planters
    for k in range(40):
        side = -1 if k % 2 == 0 else 1
        z = -half_l + 2 + k * 2.9
        _column(s, side * 5.2, z, 0.08, 4.5, q(24), q(48), 4, flute=0.0)
    # lights: strings across the street and facade lamps
    centers = np.stack([rng.uniform(-6.3, 6.3, n_quads), rng.uniform(2.8, 9.0, n_quads),
                        rng.uniform(-half_l + 1, half_l - 1, n_quads)], axis=1)
    _lantern_quads(s, rng, centers, len(base))
    return SceneData(f"bistro_class_seed{seed}", s, mats,
                     dict(position=(0.3, 1.8, 55.0), rotation=(-91.0, 1.0, 0.0), fov_y=30.0, focal_dist=1.0))


def orbit_position(base_position, frame, radius=1.0, dt=1.0 / 60.0, speed=2.7):
    """Camera orbit of runCuda (src/main.cpp:149-153) with a fixed time step instead of glfwGetTime:
    position = base + (cos t, 0, sin t) * radius, t = frame * dt * animateSpeed."""
    t = np.float32(frame * dt) * np.float32(speed)
    return (np.float32(base_position[0]) + np.float32(np.cos(t)) * np.float32(radius),
            np.float32(base_position[1]),
            np.float32(base_position[2]) + np.float32(np.sin(t)) * np.float32(radius))


def dump_scene(sd, path):
    """Binary triangle soup read by restir_amd/host/headless_viewer.cpp: int32 numPrims, numMaterials;
    float32 camera position[3], rotation[3], fovY, focalDist; vertices; normals; texcoords; int32
    materialIds; 44-byte Material records."""
    with open(path, "wb") as f:
        np.array([sd.num_prims, len(sd.materials)], np.int32).tofile(f)
        a = sd.camera_args
        np.array([*a["position"], *a["rotation"], a["fov_y"], a.get("focal_dist", 1.0)], np.float32).tofile(f)
        sd.vertices.tofile(f); sd.normals.tofile(f); sd.texcoords.tofile(f); sd.material_ids.tofile(f)
        sd.materials.tofile(f)
