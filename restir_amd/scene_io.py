"""Writers for the reference's on-disk scene format (the reader is in the library: capi.SceneFile /
rs_scene_file_load, restir_amd/csrc/scene_file.cpp).

The text format is the one Scene::Scene(filename) parses (/root/reference/src/scene.cpp:96-131 with loadMaterial
:371-433, loadModel :222-283, loadCamera :285-354):

    Material <name>                Object <name>                     Camera
    Type <Lambertian|...>          <mesh file>.obj                   Resolution <w> <h>
    BaseColor <r g b|file|Procedural>   Material <name|Null>         FovY <deg>
    Metallic <x|file>              Translate <x y z>                 LensRadius <x>
    Roughness <x|file>             Rotate <x y z>      (degrees)     FocalDist <x>
    Ior <x>                        Scale <x y z>                     ApertureMask Null
    NormalMap <file|Null>          <empty line>                      Sample <n>
                                                                     Depth <n>
    EnvMap <file|Null>                                               File <name>
                                                                     Eye / Rotation / Up <x y z>, then an empty line

so a scene written here can be rendered by the reference itself on its own hardware and compared with this port.
Images are written as binary PPM (P6, decoded to byte / 255), PNG or Radiance HDR (.hdr, RGBE), which both stb_image (the
reference) and the library's reader decode to the same floats.
"""
import os

import numpy as np

_TYPE_NAMES = {0: "Lambertian", 1: "MetallicWorkflow", 2: "Dielectric", 4: "Light"}


def _num(x):
    return repr(float(np.float32(x)))       # exact decimal expansion of the float, ends in a digit


def _vec(v):
    return " ".join(_num(x) for x in v)


def write_ppm(path, rgb):
    """rgb: (h, w, 3) uint8."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def write_png(path, samples, color_type, depth=8, palette=None, interlace=False, filters=None, level=6, idat_split=0):
    """A PNG from raw samples: samples (h, w) or (h, w, channels) integers below 2**depth (palette indices for colour type
    3, with palette = (n, 3) uint8).  color_type 0 grey, 2 RGB, 3 palette, 4 grey + alpha, 6 RGBA.  filters: a scan-line filter
    type 0..4, or a callable(row_index) -> type (default: cycles through all five); interlace: Adam7; idat_split: bytes per
    IDAT chunk (0 = one chunk).  For tests of the decoders: every legal colour type / depth combination can be written."""
    import struct
    import zlib
    a = np.asarray(samples)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, ch = a.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bits = ch * depth
    bpp = max(1, bits // 8)

    def pack_rows(sub):                              # (ph, pw, ch) -> list of byte rows
        rows = []
        for r in sub:
            flat = r.reshape(-1).astype(np.uint32)
            if depth == 8:
                rows.append(flat.astype(np.uint8).tobytes())
            elif depth == 16:
                rows.append(flat.astype(">u2").tobytes())
            else:
                nb = (len(flat) * depth + 7) // 8
                out = np.zeros(nb, np.uint8)
                for i, v in enumerate(flat):
                    bit = i * depth
                    out[bit >> 3] |= (int(v) & ((1 << depth) - 1)) << (8 - depth - (bit & 7))
                rows.append(out.tobytes())
        return rows

    def filt(rows, row0):
        out = bytearray()
        prev = bytes(len(rows[0])) if rows else b""
        for y, cur in enumerate(rows):
            ft = filters(row0 + y) if callable(filters) else (row0 + y) % 5 if filters is None else int(filters)
            line = bytearray(len(cur))
            for i in range(len(cur)):
                aa = cur[i - bpp] if i >= bpp else 0
                bb = prev[i]
                cc = prev[i - bpp] if i >= bpp else 0
                if ft == 0:
                    pred = 0
                elif ft == 1:
                    pred = aa
                elif ft == 2:
                    pred = bb
                elif ft == 3:
                    pred = (aa + bb) >> 1
                else:
                    pp = aa + bb - cc
                    pa, pb, pc = abs(pp - aa), abs(pp - bb), abs(pp - cc)
                    pred = aa if (pa <= pb and pa <= pc) else (bb if pb <= pc else cc)
                line[i] = (cur[i] - pred) & 255
            out.append(ft)
            out += line
            prev = cur
        return bytes(out)

    stream = b""
    if interlace:
        for n, (x0, y0, dx, dy) in enumerate(((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))):
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                stream += filt(pack_rows(sub), n)
    else:
        stream = filt(pack_rows(a), 0)
    z = zlib.compress(stream, level)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0)))
        f.write(chunk(b"gAMA", struct.pack(">I", 45455)))            # ancillary chunks are skipped by the readers
        if palette is not None:
            f.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        step = idat_split or len(z)
        for i in range(0, len(z), step):
            f.write(chunk(b"IDAT", z[i:i + step]))
        f.write(chunk(b"IEND", b""))


def float_to_rgbe(rgb):
    """(h, w, 3) float -> (h, w, 4) uint8 Radiance RGBE (shared exponent of the largest component)."""
    rgb = np.maximum(np.asarray(rgb, np.float64), 0.0)
    m = rgb.max(axis=2)
    e = np.zeros_like(m, dtype=np.int64)
    nz = m > 1e-32
    e[nz] = np.floor(np.log2(m[nz])).astype(np.int64) + 1             # m = f * 2^e, f in [0.5, 1)
    scale = np.where(nz, 256.0 / np.exp2(e.astype(np.float64)), 0.0)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(nz, e + 128, 0).astype(np.uint8)
    return out


def write_hdr(path, rgb, rle=True):
    """Radiance .hdr (32-bit_rle_rgbe, -Y h +X w) from a float (h, w, 3) image; run-length coded scan lines when `rle`
    (needs 8 <= w < 32768), flat RGBE quadruples otherwise."""
    px = float_to_rgbe(rgb)
    h, w = px.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# written by restir_amd.scene_io\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle or w < 8 or w >= 32768:
            f.write(px.tobytes())
            return
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                row = px[y, :, k]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        lit = x
                        while lit < w and lit - x < 128 and not (lit + 2 < w and row[lit] == row[lit + 1] == row[lit + 2]):
                            lit += 1
                        f.write(bytes([lit - x]) + row[x:lit].tobytes())
                        x = lit


def write_obj(path, vertices, normals, texcoords=None):
    """De-indexed triangle soup: vertices / normals (n, 3, 3), texcoords (n, 3, 2) or None."""
    v = np.asarray(vertices, np.float32).reshape(-1, 3)
    n = np.asarray(normals, np.float32).reshape(-1, 3)
    t = None if texcoords is None else np.asarray(texcoords, np.float32).reshape(-1, 2)
    with open(path, "w") as f:
        for p in v:
            f.write("v " + _vec(p) + "\n")
        for p in n:
            f.write("vn " + _vec(p) + "\n")
        if t is not None:
            for p in t:
                f.write("vt " + _vec(p) + "\n")
        for i in range(0, len(v), 3):
            c = [(f"{k + 1}/{k + 1}/{k + 1}" if t is not None else f"{k + 1}//{k + 1}") for k in (i, i + 1, i + 2)]
            f.write("f " + " ".join(c) + "\n")


def material_lines(name, m):
    """m: dict(type=int|str, baseColor=(r, g, b)|"Procedural"|file, metallic=x|file, roughness=x|file, ior=x, normalMap=file|None)."""
    ty = m.get("type", 0)
    base = m.get("baseColor", (0.9, 0.9, 0.9))

    def scalar(x):
        return x if isinstance(x, str) else _num(x)
    return [
        f"Material {name}",
        f"Type {ty if isinstance(ty, str) else _TYPE_NAMES[int(ty)]}",
        "BaseColor " + (base if isinstance(base, str) else _vec(base)),
        "Metallic " + scalar(m.get("metallic", 0.0)),
        "Roughness " + scalar(m.get("roughness", 1.0)),
        "Ior " + _num(m.get("ior", 1.5)),
        "NormalMap " + (m.get("normalMap") or "Null"),
        "",
    ]


def object_lines(name, mesh_file, material, translate=(0, 0, 0), rotate=(0, 0, 0), scale=(1, 1, 1)):
    return [f"Object {name}", mesh_file, f"Material {material or 'Null'}",
            "Translate " + _vec(translate), "Rotate " + _vec(rotate), "Scale " + _vec(scale), ""]


def camera_lines(width, height, fov_y, position, rotation, up=(0, 1, 0), lens_radius=0.0, focal_dist=1.0,
                 sample=1, depth=4, file="out"):
    return ["Camera", f"Resolution {int(width)} {int(height)}", "FovY " + _num(fov_y), "LensRadius " + _num(lens_radius),
            "FocalDist " + _num(focal_dist), "ApertureMask Null", f"Sample {int(sample)}", f"Depth {int(depth)}", f"File {file}",
            "Eye " + _vec(position), "Rotation " + _vec(rotation), "Up " + _vec(up), ""]


def write_scene(path, materials, objects, camera, env_map=None, newline="\n"):
    """materials: [(name, dict)], objects: [dict(name=, file=, material=, translate=, rotate=, scale=)], camera: kwargs of
    camera_lines, env_map: file name or None."""
    lines = []
    for name, m in materials:
        lines += material_lines(name, m)
    for o in objects:
        lines += object_lines(o["name"], o["file"], o.get("material"), o.get("translate", (0, 0, 0)), o.get("rotate", (0, 0, 0)),
                              o.get("scale", (1, 1, 1)))
    lines += camera_lines(**camera)
    lines += [f"EnvMap {env_map or 'Null'}"]
    with open(path, "w", newline="") as f:
        f.write(newline.join(lines) + newline)


def export_scene_data(sd, directory, width, height, name="scene", sample=1, depth=4):
    """Write a restir_amd.scenes.SceneData (untextured materials) as <directory>/<name>.txt + one OBJ per material, with
    identity transforms, e.g. to render the benchmark scenes in the reference itself.  Returns the scene file path."""
    os.makedirs(directory, exist_ok=True)
    mats, objs = [], []
    for i, m in enumerate(sd.materials):
        mats.append((f"m{i}", dict(type=int(m["type"]), baseColor=tuple(m["baseColor"]), metallic=float(m["metallic"]),
                                   roughness=float(m["roughness"]), ior=float(m["ior"]))))
    ids = np.asarray(sd.material_ids)
    # the reader appends instances in file order: keep triangle order by cutting the soup into runs of equal material
    cuts = np.flatnonzero(np.diff(ids)) + 1
    starts = np.concatenate([[0], cuts]); ends = np.concatenate([cuts, [len(ids)]])
    for k, (a, b) in enumerate(zip(starts, ends)):
        fn = f"{name}_{k}.obj"
        write_obj(os.path.join(directory, fn), sd.vertices[a:b], sd.normals[a:b], sd.texcoords[a:b])
        objs.append(dict(name=f"o{k}", file=fn, material=f"m{int(ids[a])}"))
    ca = sd.camera_args
    cam = dict(width=width, height=height, fov_y=ca["fov_y"], position=ca["position"], rotation=ca["rotation"],
               focal_dist=ca.get("focal_dist", 1.0), lens_radius=ca.get("lens_radius", 0.0), sample=sample, depth=depth, file=name)
    path = os.path.join(directory, name + ".txt")
    write_scene(path, mats, objs, cam)
    return path
