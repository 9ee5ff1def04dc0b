"""Generates tests/golden/scene_files.npz: the scene-file test case (tests/scene_file_cases.py) as file bytes, together with
what the REFERENCE'S OWN loaders make of those files -- tinyobj::LoadObj + the flattening loop of Resource::loadOBJMesh,
Image(filename) = stbi_loadf under both flip settings, the safeGetline / tokenizeString read loop, and
Math::buildTransformationMatrix + the GLM baking of Scene::buildDevData -- through oracle/_ref/libref_loaders.so and
libref_subset.so (oracle/ref_loaders.cpp, oracle/ref_subset.cpp; compiled from /root/reference in place).

Run in the build container (needs /root/reference):   python tests/golden/make_scene_golden.py
The fixture holds data only: generated inputs and the reference's outputs for them."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from tests import scene_file_cases as cases  # noqa: E402


def ref_obj(R, path):
    cap = 1 << 16
    v = np.zeros((cap, 3), np.float32); n = np.zeros((cap, 3), np.float32); t = np.zeros((cap, 2), np.float32)
    cnt = R.ref_obj_load(path.encode(), cap, v.ctypes.data, n.ctypes.data, t.ctypes.data)
    assert 0 <= cnt <= cap, cnt
    return v[:cnt].copy(), n[:cnt].copy(), t[:cnt].copy()


def ref_image(R, path, flip):
    w, h = C.c_int(), C.c_int()
    assert R.ref_image_load(path.encode(), flip, C.byref(w), C.byref(h), None, 0) == 0
    buf = np.zeros(w.value * h.value * 3, np.float32)
    assert R.ref_image_load(path.encode(), flip, C.byref(w), C.byref(h), buf.ctypes.data, buf.size) == 0
    return buf.reshape(h.value, w.value, 3)


def ref_lines(R, path):
    out = C.create_string_buffer(1 << 22)
    ln = R.ref_read_lines(path.encode(), out, 1 << 22)
    assert 0 <= ln <= (1 << 22)
    return out.raw[:ln]


def main():
    R, S = ob.ref_loaders(), ob.ref_subset()
    assert R is not None and S is not None, "build oracle/_ref first (make -C oracle)"
    g = {}
    with tempfile.TemporaryDirectory() as d:
        cases.write_case(d)
        for name in cases.CASE_FILES:
            g["file_" + name] = np.frombuffer(open(os.path.join(d, name), "rb").read(), np.uint8)
        for name in cases.CASE_FILES:
            p = os.path.join(d, name)
            if name.endswith(".obj"):
                g["obj_v_" + name], g["obj_n_" + name], g["obj_t_" + name] = ref_obj(R, p)
            elif name.endswith((".ppm", ".hdr", ".png")):
                g["img_flip_" + name] = ref_image(R, p, 1)
                g["img_noflip_" + name] = ref_image(R, p, 0)
            g["lines_" + name] = np.frombuffer(ref_lines(R, p), np.uint8)
        # line-ending variants of the scene text
        text = open(os.path.join(d, "scene.txt"), "rb").read()
        variants = {"crlf": text.replace(b"\n", b"\r\n"), "cr": text.replace(b"\n", b"\r"), "nofinal": text.rstrip(b"\n"),
                    "blanks": text.replace(b"\n\n", b"\n \t\n") + b"\n\n"}
        for key, data in variants.items():
            p = os.path.join(d, "variant.txt")
            open(p, "wb").write(data)
            g["variant_" + key] = np.frombuffer(data, np.uint8)
            g["variant_lines_" + key] = np.frombuffer(ref_lines(R, p), np.uint8)
    # JPEG pictures (written with Pillow: a generation-time dependency only) and what the reference's stb_image makes of them.
    # The product's decoder is pinned against these directly; the oracle has no JPEG restatement.
    from PIL import Image
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:37, 0:53]
    pic = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 11.0), 128 + 100 * np.cos(xx / 5.0), 128 + 90 * np.sin(yy / 3.0)], axis=2) + rng.normal(0, 25, (37, 53, 3))
    pic = np.clip(pic, 0, 255).astype(np.uint8)
    variants = {"444": dict(subsampling=0, quality=92), "422": dict(subsampling=1, quality=75), "420": dict(subsampling=2, quality=60, optimize=True),
                "411": dict(subsampling="4:1:1", quality=70), "restart": dict(subsampling=2, quality=85, restart_marker_blocks=3),
                "grey": dict(quality=80), "rgb": dict(subsampling=0, quality=90, keep_rgb=True),
                "prog420": dict(subsampling=2, quality=65, progressive=True), "prog444": dict(subsampling=0, quality=90, progressive=True, optimize=True),
                "proggrey": dict(quality=70, progressive=True), "progrestart": dict(subsampling=1, quality=80, progressive=True, restart_marker_blocks=2)}
    with tempfile.TemporaryDirectory() as d:
        for key, opt in variants.items():
            p = os.path.join(d, key + ".jpg")
            (Image.fromarray(pic[..., 1]) if key.endswith("grey") else Image.fromarray(pic)).save(p, "JPEG", **opt)
            g["jpeg_file_" + key] = np.frombuffer(open(p, "rb").read(), np.uint8)
            g["jpeg_flip_" + key] = ref_image(R, p, 1)
            g["jpeg_noflip_" + key] = ref_image(R, p, 0)
    # TGA pictures: Pillow writes the 8 / 24 / 32-bit kinds (plain and run-length coded, both row orders, colour-mapped); the
    # 16-bit 5-5-5 kinds (true colour and a 15-bit colour map) are assembled by hand
    import struct
    rng = np.random.default_rng(13)
    with tempfile.TemporaryDirectory() as d:
        def add(key, p):
            g["tga_file_" + key] = np.frombuffer(open(p, "rb").read(), np.uint8)
            g["tga_flip_" + key] = ref_image(R, p, 1)
            g["tga_noflip_" + key] = ref_image(R, p, 0)
        a = rng.integers(0, 256, (11, 19, 4), dtype=np.uint8)
        a[:, 9:] = a[:, 9:10]
        for key, im, opt in (("rgb", Image.fromarray(a[..., :3]), dict(orientation=-1)), ("rgba_rle", Image.fromarray(a), dict(compression="tga_rle", orientation=1)),
                             ("grey_rle", Image.fromarray(a[..., 0]), dict(compression="tga_rle", orientation=-1)), ("la", Image.fromarray(a[..., :2], "LA"), dict(orientation=1))):
            p = os.path.join(d, key + ".tga")
            im.save(p, "TGA", **opt)
            add(key, p)
        im = Image.fromarray(a[..., 0], "P"); im.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tobytes())
        p = os.path.join(d, "pal.tga"); im.save(p, "TGA", compression="tga_rle", orientation=-1); add("pal_rle", p)
        px16 = rng.integers(0, 65536, (7, 13), dtype=np.uint16)
        p = os.path.join(d, "rgb16.tga")
        open(p, "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 13, 7, 16, 0x20) + px16.astype("<u2").tobytes())
        add("rgb16", p)
        idx = rng.integers(0, 40, (5, 9), dtype=np.uint8)
        pal = rng.integers(0, 32768, 32, dtype=np.uint16)              # indices 32..39 fall outside the map: entry 0, as stb_image does
        p = os.path.join(d, "pal15.tga")
        open(p, "wb").write(struct.pack("<BBBHHBHHHHBB", 3, 1, 1, 0, 32, 15, 0, 0, 9, 5, 8, 0) + b"id!" + pal.astype("<u2").tobytes() + idx.tobytes())
        add("pal15", p)
    # BMP pictures: Pillow writes the palette / 24-bit / 32-bit kinds; 16-bit and bit-field pictures, the V4 / V5 / OS-2 headers and a
    # top-down file are assembled by hand
    rng = np.random.default_rng(14)
    with tempfile.TemporaryDirectory() as d:
        def add_bmp(key, p):
            g["bmp_file_" + key] = np.frombuffer(open(p, "rb").read(), np.uint8)
            g["bmp_flip_" + key] = ref_image(R, p, 1)
            g["bmp_noflip_" + key] = ref_image(R, p, 0)
        a = rng.integers(0, 256, (9, 14, 4), dtype=np.uint8)
        pal = rng.integers(0, 256, 768, dtype=np.uint8).tobytes()
        for key, im, opt in (("mono", Image.fromarray(a[..., 0] > 127), {}), ("grey", Image.fromarray(a[..., 0]), {}), ("rgb", Image.fromarray(a[..., :3]), {}),
                             ("rgba", Image.fromarray(a), {}), ("pal8", Image.fromarray(a[..., 0], "P"), {}), ("pal4", Image.fromarray(a[..., 0] % 16, "P"), dict(bits=4))):
            if key.startswith("pal"):
                im.putpalette(pal)
            p = os.path.join(d, key + ".bmp")
            im.save(p, "BMP", **opt)
            add_bmp(key, p)

        def bmp(hsz, w, h, bpp, compress, pixels, masks=b"", v4masks=None, palette=b""):
            if hsz == 12:
                hdr = struct.pack("<IHHHH", 12, w, h & 0xffff, 1, bpp)
            else:
                hdr = struct.pack("<IiiHHIIiiII", hsz, w, h, 1, bpp, compress, 0, 2835, 2835, 0, 0)
                if hsz == 56:
                    hdr += struct.pack("<IIII", 0, 0, 0, 0)
                if hsz in (108, 124):
                    hdr += struct.pack("<IIII", *v4masks) + struct.pack("<I", 0x73524742) + bytes(48)
                    if hsz == 124:
                        hdr += bytes(16)
            off = 14 + len(hdr) + len(masks) + len(palette)
            return b"BM" + struct.pack("<IHHI", off + len(pixels), 0, 0, off) + hdr + masks + palette + pixels

        def rows(a, nbytes, w):
            pad = (-(w * nbytes)) & 3
            return b"".join(a[j].tobytes() + bytes(pad) for j in range(a.shape[0]))
        h, w = 6, 7
        px16 = rng.integers(0, 65536, (h, w), dtype=np.uint16).astype("<u2")
        px32 = rng.integers(0, 2 ** 32, (h, w), dtype=np.uint32).astype("<u4")
        hand = {"rgb555": bmp(40, w, h, 16, 0, rows(px16, 2, w)),
                "rgb565_bitfields40": bmp(40, w, h, 16, 3, rows(px16, 2, w), masks=struct.pack("<III", 0xf800, 0x07e0, 0x001f)),     # stb_image 2.21 skips 12 pixel bytes here
                "v4_masks_topdown": bmp(108, w, -h, 32, 3, rows(px32, 4, w), v4masks=(0x00fc0000, 0x0000f800, 0x0000007e, 0x07000000)),
                "v5_8888": bmp(124, w, h, 32, 3, rows(px32, 4, w), v4masks=(0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000)),
                "x8888": bmp(40, w, h, 32, 0, rows(px32, 4, w)),
                "os2_pal": bmp(12, w, h, 8, 0, rows(rng.integers(0, 12, (h, w), dtype=np.uint8), 1, w), palette=rng.integers(0, 256, 48, dtype=np.uint8).tobytes()),
                "v3_56": bmp(56, w, h, 16, 3, rows(px16, 2, w), masks=struct.pack("<III", 0x7c00, 0x03e0, 0x001f))}
        for key, data in hand.items():
            p = os.path.join(d, key + ".bmp")
            open(p, "wb").write(data)
            add_bmp(key, p)
    # binary PGM (P5) and a PPM with a maxval below 255 (stb_image takes the bytes as they are)
    rng = np.random.default_rng(15)
    with tempfile.TemporaryDirectory() as d:
        for key, magic, comp, maxv in (("pgm", b"P5", 1, 255), ("ppm_max100", b"P6", 3, 100)):
            p = os.path.join(d, key + ".pnm")
            open(p, "wb").write(magic + b"\n# comment\n%d %d\n%d\n" % (9, 5, maxv) + rng.integers(0, maxv + 1, (5, 9, comp), dtype=np.uint8).tobytes())
            g["pnm_file_" + key] = np.frombuffer(open(p, "rb").read(), np.uint8)
            g["pnm_flip_" + key] = ref_image(R, p, 1)
            g["pnm_noflip_" + key] = ref_image(R, p, 0)
    # baking math
    rng = np.random.default_rng(5)
    k = 200
    t = rng.uniform(-5, 5, (k, 3)).astype(np.float32)
    r = rng.uniform(-360, 360, (k, 3)).astype(np.float32)
    s = np.exp(rng.uniform(-3, 3, (k, 3))).astype(np.float32) * rng.choice([-1.0, 1.0], (k, 3)).astype(np.float32)
    r[:20] = np.round(r[:20] / 90) * 90
    t[0] = 0; r[0] = 0; s[0] = 1
    verts = rng.uniform(-3, 3, (k, 32, 3)).astype(np.float32)
    nrm = rng.normal(size=(k, 32, 3)).astype(np.float32)
    mats = np.zeros((k, 16), np.float32); vo = np.zeros_like(verts); no = np.zeros_like(nrm)
    for i in range(k):
        S.ref_build_transformation_matrix(t[i], r[i], s[i], mats[i])
        S.ref_bake_instance(t[i], r[i], s[i], 32, verts[i].reshape(-1), nrm[i].reshape(-1), vo[i].reshape(-1), no[i].reshape(-1))
    g.update(bake_t=t, bake_r=r, bake_s=s, bake_verts=verts, bake_normals=nrm, bake_matrix=mats, bake_verts_out=vo, bake_normals_out=no)
    out = os.path.join(ROOT, "tests", "golden", "scene_files.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
