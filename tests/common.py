"""Shared helpers for the parity tests: build the same scene for the oracle and for the HIP
library, drive both through the reference's per-frame call sequence (runCuda, src/main.cpp:146-185)
and compare."""
import ctypes as C

import numpy as np

from oracle import binding as ob
from restir_amd import scenes
from restir_amd.ctypes_structs import RESERVOIR_DTYPE, copy_camera


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def mismatch_fraction(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.float32:
        ne = a.view(np.uint32) != b.view(np.uint32)
    else:
        ne = a != b
    return float(np.count_nonzero(ne)) / max(1, ne.size)


def oracle_scene(sd):
    return ob.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials,
                    textures=sd.textures, env_map_tex=sd.env_map_tex)


def hip_scene(capi, sd):
    return capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials,
                      textures=sd.textures, env_map_tex=sd.env_map_tex)


def get_scene(name):
    if name == "cornell":
        return scenes.cornell_box()
    if name == "cornell_textured":
        return scenes.cornell_textured(env=True)
    if name == "cornell_maps":
        return scenes.cornell_textured(env=False)
    if name.startswith("sponza"):
        return scenes.sponza_class(1, float(name.split(":")[1]))
    if name.startswith("bistro"):
        return scenes.bistro_class(2, float(name.split(":")[1]))
    raise KeyError(name)


def next_looper(looper, sobol_num):
    """The tail of every launcher (src/restir.cu:441-445): looper++ or, with the Sobol sampler, (looper + 1) % SobolSampleNum."""
    return looper + 1 if sobol_num is None else (looper + 1) % sobol_num


class OracleRenderer:
    """runCuda on the CPU oracle."""

    def __init__(self, sd, width, height, scene=None, sobol=None):
        """sobol: the Sobol table (uint32 [SobolSampleNum, 200]) = a build with SAMPLER_USE_SOBOL true; None = the default engine."""
        self.sd = sd
        self.scene = scene or oracle_scene(sd)
        if sobol is not None or scene is None:
            self.scene.set_sample_sequence(sobol)
        self.sobol_num = None if sobol is None else len(sobol)
        self.W, self.H = width, height
        self.cam = ob.camera_update(sd.camera(width, height))
        self.gbuf = ob.GBuffer(width, height)
        self.restir = ob.ReSTIR(width, height)
        self.image = np.zeros((width * height, 3), np.float32)
        self.looper = 0
        self.rays = 0

    def set_camera_position(self, pos):
        for i in range(3):
            self.cam.position[i] = float(pos[i])
        ob.camera_update(self.cam)

    def frame(self, reuse, use_reservoir=True, iteration=0):
        self.gbuf.render(self.scene, self.cam)
        if use_reservoir:
            self.rays = self.restir.direct(self.scene, self.cam, self.gbuf, self.image, iteration, self.looper, reuse)
        else:
            self.rays = ob.pt_direct(self.scene, self.cam, self.image, iteration, self.looper)
        self.looper = next_looper(self.looper, self.sobol_num)
        self.gbuf.update(self.cam)
        return self.image


class HipRenderer:
    """runCuda on librestir_hip through the C ABI."""

    def __init__(self, capi, sd, width, height, scene=None, sobol=None):
        import torch
        self.torch = torch
        self.capi = capi
        self.sd = sd
        self.scene = scene or hip_scene(capi, sd)
        if sobol is not None or scene is None:
            self.scene.set_sample_sequence(sobol)
        self.sobol_num = None if sobol is None else len(sobol)
        self.W, self.H = width, height
        self.cam = capi.camera_update(sd.camera(width, height))
        self.gbuf = capi.GBuffer(width, height)
        self.restir = capi.ReSTIR(width, height)
        self.image = torch.zeros((width * height, 3), dtype=torch.float32, device="cuda")
        self.looper = 0

    def set_camera_position(self, pos):
        for i in range(3):
            self.cam.position[i] = float(pos[i])
        self.capi.camera_update(self.cam)

    def frame(self, reuse, use_reservoir=True, iteration=0):
        self.gbuf.render(self.scene, self.cam)
        if use_reservoir:
            self.restir.direct(self.scene, self.cam, self.gbuf, self.image.data_ptr(), iteration, self.looper, reuse)
            self.rays = self.restir.ray_count()
        else:
            self.rays = self.capi.path_trace_direct(self.scene, self.cam, self.image.data_ptr(), iteration, self.looper)
        self.looper = next_looper(self.looper, self.sobol_num)
        self.gbuf.update(self.cam)
        return self.image.cpu().numpy()


def radiance_stats(a, b):
    """a, b: (N,3) float32.  mean per-pixel L1, fraction of pixels with L1 > 1e-3, bit mismatch fraction."""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).sum(axis=1)
    return dict(mean_l1=float(d.mean()), flip_frac=float(np.mean(d > 1e-3)), bit_mismatch=mismatch_fraction(a, b), max_l1=float(d.max()))


class OracleBackend:
    """Test double with the HipBackend interface (restir_amd/tiling.py), computing with the CPU oracle
    and exchanging torch CPU tensors, so the strip decomposition runs under gloo without a GPU."""

    def __init__(self, scene, cam, width, height):
        import torch
        self.torch = torch
        self.scene, self.cam = scene, cam
        self.W, self.H = width, height
        self.gbuf = ob.GBuffer(width, height)
        self.restir = ob.ReSTIR(width, height)
        self.image = np.zeros((width * height, 3), np.float32)

    def empty(self, nbytes):
        return self.torch.zeros(nbytes, dtype=self.torch.uint8)

    def gbuffer_render(self, y0, y1):
        self.gbuf.render(self.scene, self.cam, y0, y1)

    def phase_a(self, looper, reuse, y0, y1):
        self.restir.phase_a(self.scene, self.cam, self.gbuf, looper, reuse, y0, y1)

    def phase_b(self, iteration, reuse, y0, y1):
        self.restir.phase_b(self.scene, self.cam, self.gbuf, self.image, iteration, reuse, y0, y1)

    def end_frame(self):
        self.restir.end_frame()
        self.gbuf.update(self.cam)

    def _rows(self, arr, y0, rows):
        return arr[y0 * self.W:(y0 + rows) * self.W]

    # halo = published reservoirs + the G-buffer id / normal / depth rows the spatial taps compare against (as HipBackend)
    def halo_pack(self, y0, rows):
        res = self.torch.from_numpy(self._rows(self.restir.temp, y0, rows).copy().view(np.uint8).reshape(-1))
        return self.torch.cat([res, self.gbuffer_rows_get(y0, rows)])

    def halo_unpack(self, y0, rows, buf):
        n = rows * self.W * RESERVOIR_DTYPE.itemsize
        b = buf.contiguous()
        self._rows(self.restir.temp, y0, rows)[:] = b[:n].numpy().view(RESERVOIR_DTYPE)
        self.gbuffer_rows_put(y0, rows, b[n:])

    # LeveledEAWFilter on strips (the interface of HipBackend): two full-frame buffers, the oracle's level on the whole frame
    # (rows outside the strip and its exchanged border hold stale values and are never looked at)
    def _eaw_input(self, level):
        if not hasattr(self, "eaw_buf"):
            self.eaw_buf = [np.zeros_like(self.image), np.zeros_like(self.image)]
        return self.image if level == 0 else self.eaw_buf[(level - 1) % 2]

    def gbuffer_rows_get(self, y, rows):
        cur = self.gbuf.frame_idx
        parts = [self._rows(self.gbuf.prim_id[cur], y, rows).view(np.uint8).reshape(-1),
                 self._rows(self.gbuf.normal[cur], y, rows).reshape(-1).view(np.uint8),
                 self._rows(self.gbuf.depth[cur], y, rows).view(np.uint8).reshape(-1)]
        return self.torch.from_numpy(np.concatenate(parts))

    def gbuffer_rows_put(self, y, rows, buf):
        cur = self.gbuf.frame_idx
        n = rows * self.W
        b = buf.numpy()
        self._rows(self.gbuf.prim_id[cur], y, rows)[:] = b[:n * 4].view(np.int32)
        self._rows(self.gbuf.normal[cur], y, rows)[:] = b[n * 4:n * 16].view(np.float32).reshape(n, 3)
        self._rows(self.gbuf.depth[cur], y, rows)[:] = b[n * 16:n * 20].view(np.float32)

    def eaw_rows_get(self, level, y, rows):
        return self.torch.from_numpy(self._rows(self._eaw_input(level), y, rows).copy())

    def eaw_rows_put(self, level, y, rows, buf):
        self._rows(self._eaw_input(level), y, rows)[:] = buf.numpy()

    def eaw_positions(self, y0, y1):
        pass                                             # the oracle reconstructs positions per tap from the depth plane

    def eaw_level(self, level, y0, y1):
        src = self._eaw_input(level)
        ob.lib().orc_eaw_level(C.byref(self.gbuf.c), C.byref(self.cam), src.reshape(-1), self.eaw_buf[level % 2].reshape(-1), 1.0, 0.2, 64.0, level)

    def eaw_result(self):
        return self.eaw_buf[0]

    def history_bytes(self, rows):
        return rows * self.W * (36 + 20)

    def history_pack(self, y0, rows):
        last = self.gbuf.frame_idx ^ 1
        parts = [self._rows(self.restir.last, y0, rows).view(np.uint8).reshape(-1),
                 self._rows(self.gbuf.prim_id[last], y0, rows).view(np.uint8).reshape(-1),
                 self._rows(self.gbuf.normal[last], y0, rows).reshape(-1).view(np.uint8),
                 self._rows(self.gbuf.depth[last], y0, rows).view(np.uint8).reshape(-1)]
        return self.torch.from_numpy(np.concatenate(parts))

    def history_unpack(self, y0, rows, buf):
        last = self.gbuf.frame_idx ^ 1
        n = rows * self.W
        b = buf.numpy()
        o = 0
        self._rows(self.restir.last, y0, rows)[:] = b[o:o + n * 36].view(RESERVOIR_DTYPE); o += n * 36
        self._rows(self.gbuf.prim_id[last], y0, rows)[:] = b[o:o + n * 4].view(np.int32); o += n * 4
        self._rows(self.gbuf.normal[last], y0, rows)[:] = b[o:o + n * 12].view(np.float32).reshape(n, 3); o += n * 12
        self._rows(self.gbuf.depth[last], y0, rows)[:] = b[o:o + n * 4].view(np.float32)


def read_png_rgb(path):
    """Decodes an 8-bit RGB PNG whose rows use filter type 0 (what rs_write_png produces) with zlib alone; checks every chunk CRC."""
    import struct, zlib
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, size = 8, b"", None
    while off < len(d):
        n, = struct.unpack(">I", d[off:off + 4])
        typ, body = d[off + 4:off + 8], d[off + 8:off + 8 + n]
        crc, = struct.unpack(">I", d[off + 8 + n:off + 12 + n])
        assert zlib.crc32(typ + body) & 0xffffffff == crc, typ
        if typ == b"IHDR":
            w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
            size = (h, w)
        elif typ == b"IDAT":
            idat += body
        off += 12 + n
    h, w = size
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3).copy()
