"""oracle/scene_format.py -- CPU restatement of the reference's scene-file front end (TEST INFRASTRUCTURE ONLY).

Follows, function by function:
    Scene::Scene(filename)          /root/reference/src/scene.cpp:96-131
    Scene::loadMaterial             src/scene.cpp:371-433
    Scene::loadModel                src/scene.cpp:222-283
    Scene::loadCamera               src/scene.cpp:285-354
    Resource::loadOBJMesh           src/scene.cpp:27-61 over tinyobjloader 2.0 (external/include/tiny_obj_loader.h:
                                    tryParseDouble 866-996, quad split 1429-1524)
    Image::Image(filename)          src/image.cpp:14-31 (stbi_loadf: an 8-bit image with ldr_to_hdr gamma 1 = value / 255; Radiance
                                    HDR = mantissa * 2^(e - 136)); PPM, PNG and HDR are restated here -- the product's JPEG and TGA
                                    decoders are pinned against stb_image directly, without an oracle restatement
    instance baking of buildDevData src/scene.cpp:161-176 (through liboracle's orc_bake_instance)
    safeGetline / tokenizeString    src/utilities.cpp:57-95

Pinning: the OBJ reader, the image decoder and the line reader / tokenizer are checked against the reference's own code
(oracle/_ref/libref_loaders.so: tinyobj::LoadObj, Image(filename), safeGetline + tokenizeString) in
tests/test_oracle_vs_reference.py and through tests/golden/scene_files.npz; the baking math against Math::buildTransformationMatrix
and GLM.  The grammar of the scene file itself (the order and meaning of the Material / Object / Camera lines) has no
reference fixture or buildable reference code behind it (scene.cpp needs Thrust and nvcc) -- that part is a reading of the
source text: "parity unpinned" for the grammar.

Pure Python loops: meant for the small scenes of the tests, not for production meshes.
"""
import math
import os

import numpy as np

from . import binding as ob
from restir_amd.ctypes_structs import Camera, MATERIAL_DTYPE

_BLANK = " \t\n\v\f\r"


def read_lines(path):
    """What the `while (fp.good()) safeGetline(fp, line)` loop sees: a list of lines."""
    with open(path, "rb") as f:
        text = f.read().decode("latin-1")
    out, at, n = [], 0, len(text)
    while True:
        end = at
        while end < n and text[end] not in "\r\n":
            end += 1
        line = text[at:end]
        out.append(line)
        if end == n:
            if line == "":
                return out          # this read set eofbit: the loop ends after it
            at = end
        else:
            at = end + (2 if text[end] == "\r" and text[end + 1:end + 2] == "\n" else 1)


def tokenize(line):
    out, cur = [], ""
    for ch in line:
        if ch in _BLANK:
            if cur:
                out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


class _Reader:
    def __init__(self, path):
        self.lines = read_lines(path)
        self.i = 0

    def good(self):
        return self.i < len(self.lines)

    def next(self):
        if self.i >= len(self.lines):
            return ""
        s = self.lines[self.i]
        self.i += 1
        return s


def obj_real(tok):
    """tinyobj's tryParseDouble + the narrowing to float (real_t)."""
    n = len(tok)
    if n == 0:
        return np.float32(0)
    i, mant, expo, neg, exp_neg, leading_dot = 0, 0.0, 0, False, False, False

    def digit(k):
        return k < n and "0" <= tok[k] <= "9"
    if tok[0] in "+-":
        neg = tok[0] == "-"
        i = 1
        if i < n and tok[i] == ".":
            leading_dot = True
    elif digit(0):
        pass
    elif tok[0] == ".":
        leading_dot = True
    else:
        return np.float32(0)
    if not leading_dot:
        got = 0
        while digit(i):
            mant = mant * 10 + (ord(tok[i]) - 48)
            i += 1
            got += 1
        if got == 0:
            return np.float32(0)
    if i < n and tok[i] == ".":
        lut = [1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001]
        i += 1
        k = 1
        while digit(i):
            mant += (ord(tok[i]) - 48) * (lut[k] if k < 8 else math.pow(10.0, -k))
            k += 1
            i += 1
    elif i < n and tok[i] not in "eE":
        i = n
    if i < n and tok[i] in "eE":
        i += 1
        if i < n and tok[i] in "+-":
            exp_neg = tok[i] == "-"
            i += 1
        elif not digit(i):
            return np.float32(0)
        got = 0
        while digit(i):
            if expo > 2147483647 // 10:
                return np.float32(0)
            expo = expo * 10 + (ord(tok[i]) - 48)
            i += 1
            got += 1
        if exp_neg:
            expo = -expo
        if got == 0:
            return np.float32(0)
    with np.errstate(all="ignore"):              # C semantics: overflow gives inf, underflow 0, no exception
        val = (-1 if neg else 1) * (np.ldexp(np.float64(mant) * np.power(np.float64(5.0), np.float64(expo)), expo) if expo else np.float64(mant))
        return np.float32(val)


def _ear_clip(cs, pv):
    """tinyobjloader's triangulation of a polygon with five or more corners (tiny_obj_loader.h:1536-1800), float32 throughout."""
    f = np.float32
    n0 = len(cs)
    P = [np.asarray(pv[c[0]], np.float32) for c in cs]
    ax0, ax1 = 1, 2
    with np.errstate(all="ignore"):
        for k in range(n0):
            a, b, c = P[k], P[(k + 1) % n0], P[(k + 2) % n0]
            e0, e1 = b - a, c - b
            cx = abs(f(f(e0[1] * e1[2]) - f(e0[2] * e1[1]))); cy = abs(f(f(e0[2] * e1[0]) - f(e0[0] * e1[2]))); cz = abs(f(f(e0[0] * e1[1]) - f(e0[1] * e1[0])))
            eps = f(1.1920928955078125e-7)
            if cx > eps or cy > eps or cz > eps:
                if not (cx > cy and cx > cz):
                    ax0 = 0
                    if cz > cx and cz > cy:
                        ax1 = 1
                break
        rest = list(range(n0))
        out = []
        guess, budget, previous = 0, n0, n0
        while len(rest) > 3 and budget > 0:
            n = len(rest)
            if guess >= n:
                guess -= n
            if previous != n:
                previous, budget = n, n
            else:
                budget -= 1
            ear = [rest[(guess + k) % n] for k in range(3)]
            vx = [P[e][ax0] for e in ear]; vy = [P[e][ax1] for e in ear]
            e0x, e0y, e1x, e1y = f(vx[1] - vx[0]), f(vy[1] - vy[0]), f(vx[2] - vx[1]), f(vy[2] - vy[1])
            cross = f(f(e0x * e1y) - f(e0y * e1x))
            area = f(f(f(vx[0] * vy[1]) - f(vy[0] * vx[1])) * f(0.5))
            if f(cross * area) < 0:
                guess += 1
                continue
            overlap = False
            for other in range(3, n):
                o = rest[(guess + other) % n]
                tx, ty = P[o][ax0], P[o][ax1]
                inside = False
                j = 2
                for i in range(3):
                    if (vy[i] > ty) != (vy[j] > ty) and tx < f(f(f(f(vx[j] - vx[i]) * f(ty - vy[i])) / f(vy[j] - vy[i])) + vx[i]):
                        inside = not inside
                    j = i
                if inside:
                    overlap = True
                    break
            if overlap:
                guess += 1
                continue
            out += ear
            del rest[(guess + 1) % n]
        if len(rest) == 3:
            out += rest
    return [cs[i] for i in out]


def load_obj(path):
    """(vertices, normals, texcoords) de-indexed as Resource::loadOBJMesh leaves them: (n, 3), (n, 3), (n, 2) float32; None when the
    file cannot be opened."""
    if not os.path.isfile(path):
        return None
    pv, pn, pt, flat = [], [], [], []
    for line in read_lines(path):
        tok = tokenize(line)
        if not tok:
            continue

        def real(k):
            return obj_real(tok[k]) if k < len(tok) else np.float32(0)
        if tok[0] == "v":
            pv.append([real(1), real(2), real(3)])
        elif tok[0] == "vn":
            pn.append([real(1), real(2), real(3)])
        elif tok[0] == "vt":
            pt.append([real(1), real(2)])
        elif tok[0] == "f":
            cs = []
            for t in tok[1:]:
                f = t.split("/")
                raw = [int(x) if x not in ("", "+", "-") else None for x in (f + ["", ""])[:3]]
                counts = (len(pv), len(pt), len(pn))
                idx = [(-1 if r is None else r - 1 if r > 0 else c + r if r < 0 else -2) for r, c in zip(raw, counts)]
                if not (0 <= idx[0] < len(pv)) or not (0 <= idx[2] < len(pn)) or idx[1] < -1 or idx[1] >= len(pt):
                    raise ValueError(f"{path}: bad face corner {t}")
                cs.append(tuple(idx))
            if len(cs) < 3:
                continue
            if len(cs) == 3:
                flat += cs
            elif len(cs) == 4:
                p = [np.asarray(pv[c[0]], np.float32) for c in cs]
                e02, e13 = p[2] - p[0], p[3] - p[1]
                with np.errstate(all="ignore"):
                    sqr02 = np.float32(np.float32(e02[0] * e02[0] + e02[1] * e02[1]) + e02[2] * e02[2])
                    sqr13 = np.float32(np.float32(e13[0] * e13[0] + e13[1] * e13[1]) + e13[2] * e13[2])
                order = (0, 1, 2, 0, 2, 3) if sqr02 < sqr13 else (0, 1, 3, 1, 2, 3)
                flat += [cs[q] for q in order]
            else:
                flat += _ear_clip(cs, pv)
    has_tc = len(pt) > 0
    v = np.array([pv[c[0]] for c in flat], np.float32).reshape(-1, 3)
    n = np.array([pn[c[2]] for c in flat], np.float32).reshape(-1, 3)
    t = np.array([pt[c[1]] if has_tc else (0.0, 0.0) for c in flat], np.float32).reshape(-1, 2)
    return v, n, t


def load_ppm(path, flip):
    """Binary PPM (P6, maxval 255) -> (h, w, 3) float32 = byte / 255, rows reversed when `flip`."""
    with open(path, "rb") as f:
        raw = f.read()
    pos, fields = 0, []
    while len(fields) < 4:
        while raw[pos:pos + 1] in b" \t\r\n":
            pos += 1
        if raw[pos:pos + 1] == b"#":
            while raw[pos:pos + 1] not in (b"\n", b""):
                pos += 1
            continue
        end = pos
        while raw[end:end + 1] not in (b" ", b"\t", b"\r", b"\n", b""):
            end += 1
        fields.append(raw[pos:end])
        pos = end
    assert fields[0] == b"P6" and int(fields[3]) == 255, "only binary PPM / 8 bit"
    w, h = int(fields[1]), int(fields[2])
    pos += 1                                    # the single whitespace byte after maxval
    px = np.frombuffer(raw, np.uint8, w * h * 3, pos).reshape(h, w, 3)
    if flip:
        px = px[::-1]
    return (px.astype(np.float32) / np.float32(255)).astype(np.float32)


def load_hdr(path, flip):
    """Radiance .hdr (32-bit_rle_rgbe, -Y h +X w) -> (h, w, 3) float32 = mantissa * 2^(e - 136), 0 where e == 0 (stb_image's
    stbi__hdr_load + stbi__hdr_convert, external/include/stb_image.h:6736-6862); rows reversed when `flip`."""
    raw = open(path, "rb").read()
    at = 0

    def line():
        nonlocal at
        end = raw.find(b"\n", at)
        end = len(raw) if end < 0 else end
        l = raw[at:end]
        at = min(end + 1, len(raw))
        return l
    assert line() in (b"#?RADIANCE", b"#?RGBE"), "not HDR"
    ok = False
    while True:
        l = line()
        if l == b"":
            break
        ok |= l == b"FORMAT=32-bit_rle_rgbe"
    assert ok, "unsupported HDR format"
    f = line().split()
    assert f[0] == b"-Y" and f[2] == b"+X"
    h, w = int(f[1]), int(f[3])
    px = np.zeros((h, w, 4), np.uint8)
    if w < 8 or w >= 32768 or not (raw[at] == 2 and raw[at + 1] == 2 and not raw[at + 2] & 0x80):
        px[:] = np.frombuffer(raw, np.uint8, w * h * 4, at).reshape(h, w, 4)
    else:
        for y in range(h):
            assert raw[at] == 2 and raw[at + 1] == 2 and ((raw[at + 2] << 8) | raw[at + 3]) == w
            at += 4
            for k in range(4):
                x = 0
                while x < w:
                    count = raw[at]; at += 1
                    if count > 128:
                        px[y, x:x + count - 128, k] = raw[at]; at += 1
                        x += count - 128
                    else:
                        px[y, x:x + count, k] = np.frombuffer(raw, np.uint8, count, at); at += count
                        x += count
    if flip:
        px = px[::-1]
    e = px[..., 3].astype(np.int32)
    scale = np.where(e != 0, np.ldexp(np.float32(1), e - 136), np.float32(0)).astype(np.float32)
    return (px[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


def load_png(path, flip):
    """PNG -> (h, w, 3) float32 as stbi_loadf(.., 3) returns it: 8-bit samples (high byte of 16-bit ones, 1 / 2 / 4-bit grey scaled
    to 0..255), palette expanded, grey replicated, alpha dropped, byte / 255 (stb_image.h: stbi__do_png, stbi__convert_format,
    stbi__convert_16_to_8, stbi__ldr_to_hdr with gamma 1).  zlib does the inflate here."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, palette = 8, b"", None
    while at + 12 <= len(raw):
        n, tag = struct.unpack(">I4s", raw[at:at + 8])
        body = raw[at + 8:at + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat += body
        elif tag == b"IEND":
            break
        at += 12 + n
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = ch * depth
    bpp = max(1, bits // 8)
    px = zlib.decompress(idat)
    rgb = np.zeros((h, w, 3), np.uint8)
    passes = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)) if interlace else ((0, 0, 1, 1),)
    pos = 0
    for x0, y0, dx, dy in passes:
        pw, ph = (w - x0 + dx - 1) // dx, (h - y0 + dy - 1) // dy
        if pw <= 0 or ph <= 0:
            continue
        nb = (pw * bits + 7) // 8
        prev = bytearray(nb)
        for y in range(ph):
            ft = px[pos]
            cur = bytearray(px[pos + 1:pos + 1 + nb])
            pos += 1 + nb
            for i in range(nb):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 2:
                    pred = b
                elif ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    pred = 0
                cur[i] = (cur[i] + pred) & 255
            prev = cur
            row = np.frombuffer(bytes(cur), np.uint8)
            if depth == 8:
                s = row[:pw * ch].reshape(pw, ch)
            elif depth == 16:
                s = row[:pw * ch * 2].reshape(pw, ch, 2)[:, :, 0]
            else:
                bitsarr = np.unpackbits(row)[:pw * depth].reshape(pw, depth)
                s = (bitsarr * (1 << np.arange(depth - 1, -1, -1))).sum(axis=1).astype(np.uint8).reshape(pw, 1)
            if ctype == 3:
                out = palette[s[:, 0]]
            elif ch <= 2:
                g = (s[:, 0].astype(np.uint16) * ({1: 255, 2: 85, 4: 17}.get(depth, 1) if depth < 8 else 1)).astype(np.uint8)
                out = np.stack([g, g, g], axis=1)
            else:
                out = s[:, :3]
            rgb[y0 + y * dy, x0::dx][:pw] = out
    if flip:
        rgb = rgb[::-1]
    return (rgb.astype(np.float32) / np.float32(255)).astype(np.float32)


def load_image(path, flip):
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic == b"#?":
        return load_hdr(path, flip)
    if magic == b"\x89P":
        return load_png(path, flip)
    if magic == b"P6":
        return load_ppm(path, flip)
    # JPEG and TGA: the product's decoders are pinned directly against the reference's stb_image (tests/golden/scene_files.npz,
    # tests/test_scene_files.py); there is no restatement of them here
    raise NotImplementedError(f"{path}: the oracle restates PPM, PNG and Radiance HDR decoding only")


_TYPES = {"Lambertian": 0, "MetallicWorkflow": 1, "Dielectric": 2, "Light": 4}


def _default_material():
    m = np.zeros(1, MATERIAL_DTYPE)[0]
    m["type"] = 0
    m["baseColor"] = (0.9, 0.9, 0.9)
    m["metallic"], m["roughness"], m["ior"] = 0.0, 1.0, 1.5
    for k in ("baseColorMapId", "metallicMapId", "roughnessMapId", "normalMapId"):
        m[k] = -1
    return m


def _f(tok):
    return np.float32(float(tok))               # std::stof: the correctly rounded float of the decimal


class ParsedScene:
    pass


def load_scene(path):
    """Scene::Scene(filename) + the vertex loop of buildDevData -> ParsedScene with the flat arrays."""
    fp = _Reader(path)
    base = os.path.dirname(path)

    def resolve(name):
        return name if os.path.isabs(name) or os.path.exists(name) else os.path.join(base, name)

    s = ParsedScene()
    s.materials, material_map = [], {}
    s.textures, texture_ids = [], {}
    s.env_map_tex = -1
    s.camera, s.iterations, s.trace_depth, s.image_name = None, 0, 0, ""
    s.num_skipped_objects = 0
    verts, norms, tcs, mat_ids = [], [], [], []
    mesh_pool = {}

    def add_texture(name, flip):
        if name not in texture_ids:
            texture_ids[name] = len(s.textures)
            s.textures.append(load_image(resolve(name), flip))
        return texture_ids[name]

    while fp.good():
        line = fp.next()
        if line == "":
            continue
        tokens = tokenize(line)
        if not tokens:
            continue
        if tokens[0] == "Material":
            m = _default_material()
            for _ in range(6):
                t = tokenize(fp.next())
                if len(t) < 2:
                    continue
                if t[0] == "Type":
                    m["type"] = _TYPES.get(t[1], 0)
                elif t[0] == "BaseColor":
                    if len(t) > 2:
                        m["baseColor"] = (_f(t[1]), _f(t[2]), _f(t[3]))
                    elif t[1] == "Procedural":
                        m["baseColorMapId"] = -2
                    else:
                        m["baseColorMapId"] = add_texture(t[1], True)
                elif t[0] in ("Metallic", "Roughness"):
                    key = t[0].lower()
                    if t[1][-1].isdigit():
                        m[key] = _f(t[1])
                    else:
                        m[key + "MapId"] = add_texture(t[1], True)
                elif t[0] == "Ior":
                    m["ior"] = _f(t[1])
                elif t[0] == "NormalMap" and t[1] != "Null":
                    m["normalMapId"] = add_texture(t[1], True)
            material_map[tokens[1]] = len(s.materials)
            s.materials.append(m)
        elif tokens[0] == "Object":
            filename = fp.next()
            if filename not in mesh_pool:
                mesh = load_obj(resolve(filename))
                if mesh is None:
                    line = filename
                    while line != "" and fp.good():
                        line = fp.next()
                    s.num_skipped_objects += 1
                    continue
                mesh_pool[filename] = mesh
            mesh = mesh_pool[filename]
            material_id = 0
            line = fp.next()
            if line != "" and fp.good():
                t = tokenize(line)
                if t[1] == "Null":
                    material_id = len(s.materials)
                    s.materials.append(_default_material())
                else:
                    material_id = material_map[t[1]]
            tr = {"Translate": np.zeros(3, np.float32), "Rotate": np.zeros(3, np.float32), "Scale": np.zeros(3, np.float32)}
            line = fp.next()
            while line != "" and fp.good():
                t = tokenize(line)
                if len(t) >= 4 and t[0] in tr:
                    tr[t[0]] = np.array([_f(t[1]), _f(t[2]), _f(t[3])], np.float32)
                line = fp.next()
            v, n = ob.bake_instance(tr["Translate"], tr["Rotate"], tr["Scale"], mesh[0], mesh[1])
            verts.append(v); norms.append(n); tcs.append(mesh[2])
            mat_ids += [material_id] * (len(v) // 3)
        elif tokens[0] == "Camera":
            cam = Camera()
            fovy = np.float32(0)
            for _ in range(8):
                t = tokenize(fp.next())
                if len(t) < 2:
                    continue
                if t[0] == "Resolution":
                    cam.resolution[0], cam.resolution[1] = int(t[1]), int(t[2])
                elif t[0] == "FovY":
                    fovy = _f(t[1])
                elif t[0] == "LensRadius":
                    cam.lensRadius = _f(t[1])
                elif t[0] == "FocalDist":
                    cam.focalDist = _f(t[1])
                elif t[0] == "Sample":
                    s.iterations = int(t[1])
                elif t[0] == "Depth":
                    s.trace_depth = int(t[1])
                elif t[0] == "File":
                    s.image_name = t[1]
            line = fp.next()
            while line != "" and fp.good():
                t = tokenize(line)
                dst = {"Eye": cam.position, "Rotation": cam.rotation, "Up": cam.up}.get(t[0]) if len(t) >= 4 else None
                if dst is not None:
                    for k in range(3):
                        dst[k] = _f(t[1 + k])
                line = fp.next()
            pi = np.float32(3.1415926535897932384626422832795028841971)
            yscaled = ob.tanf(np.float32(fovy * np.float32(pi / np.float32(180))))
            xscaled = np.float32(np.float32(yscaled * np.float32(cam.resolution[0])) / np.float32(cam.resolution[1]))
            cam.fov[0] = np.float32(np.float32(ob.atanf(xscaled) * np.float32(180)) / pi)
            cam.fov[1] = fovy
            cam.tanFovY = ob.tanf(np.float32(np.float32(fovy * np.float32(0.5)) * np.float32(0.01745329251994329576923690768489)))
            ob.camera_update(cam)
            s.camera = cam
        elif tokens[0] == "EnvMap":
            if tokens[1] != "Null":
                s.env_map_tex = add_texture(tokens[1], False)

    s.vertices = np.concatenate(verts).reshape(-1, 3, 3)
    s.normals = np.concatenate(norms).reshape(-1, 3, 3)
    s.texcoords = np.concatenate(tcs).reshape(-1, 3, 2)
    s.material_ids = np.array(mat_ids, np.int32)
    s.materials = np.array(s.materials, MATERIAL_DTYPE)
    return s
