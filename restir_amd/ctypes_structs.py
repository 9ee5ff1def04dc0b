"""ctypes mirrors of the plain-C structs at the drop-in boundary (include/restir_hip.h).

Layouts follow the reference's own structs so a `Camera` / `Material` / `Reservoir<DirectLiSample>`
can be handed over by reinterpretation:
  rs_material  <- src/material.h:258-267   (44 B)
  rs_camera    <- src/sceneStructs.h:104-117 (196 B, mat3/mat4 column-major)
  rs_reservoir <- src/restir.h:7-11,114-116 (36 B)
"""
import ctypes as C

import numpy as np


class Material(C.Structure):
    _fields_ = [
        ("type", C.c_int),
        ("baseColor", C.c_float * 3),
        ("metallic", C.c_float),
        ("roughness", C.c_float),
        ("ior", C.c_float),
        ("baseColorMapId", C.c_int),
        ("metallicMapId", C.c_int),
        ("roughnessMapId", C.c_int),
        ("normalMapId", C.c_int),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("resolution", C.c_int * 2),
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 3),
        ("view", C.c_float * 3),
        ("up", C.c_float * 3),
        ("right", C.c_float * 3),
        ("fov", C.c_float * 2),
        ("pixelLength", C.c_float * 2),
        ("rotationMatInv", C.c_float * 9),
        ("viewProjection", C.c_float * 16),
        ("lensRadius", C.c_float),
        ("focalDist", C.c_float),
        ("tanFovY", C.c_float),
    ]


class Reservoir(C.Structure):
    _fields_ = [
        ("Li", C.c_float * 3),
        ("wi", C.c_float * 3),
        ("dist", C.c_float),
        ("numSamples", C.c_int),
        ("weight", C.c_float),
    ]


assert C.sizeof(Material) == 44
assert C.sizeof(Camera) == 196
assert C.sizeof(Reservoir) == 36

MATERIAL_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("baseColor", "<f4", (3,)),
        ("metallic", "<f4"),
        ("roughness", "<f4"),
        ("ior", "<f4"),
        ("baseColorMapId", "<i4"),
        ("metallicMapId", "<i4"),
        ("roughnessMapId", "<i4"),
        ("normalMapId", "<i4"),
    ]
)
RESERVOIR_DTYPE = np.dtype(
    [("Li", "<f4", (3,)), ("wi", "<f4", (3,)), ("dist", "<f4"), ("numSamples", "<i4"), ("weight", "<f4")]
)
# Reservoir<IndirectLiSample> (src/restir.h:13-27,114-116), 68 bytes
INDIRECT_RESERVOIR_DTYPE = np.dtype(
    [("Lo", "<f4", (3,)), ("xv", "<f4", (3,)), ("nv", "<f4", (3,)), ("xs", "<f4", (3,)), ("ns", "<f4", (3,)),
     ("numSamples", "<i4"), ("weight", "<f4")]
)
assert MATERIAL_DTYPE.itemsize == 44 and RESERVOIR_DTYPE.itemsize == 36 and INDIRECT_RESERVOIR_DTYPE.itemsize == 68

# Material::Type (src/material.h:114-120)
LAMBERTIAN, METALLIC_WORKFLOW, DIELECTRIC, DISNEY, LIGHT = range(5)
# ReservoirReuse (src/common.h:38-45)
REUSE_NONE, REUSE_TEMPORAL, REUSE_SPATIAL, REUSE_SPATIOTEMPORAL = 0, 1, 2, 3
# ToneMapping (src/common.h:20-24)
TONEMAP_NONE, TONEMAP_FILMIC, TONEMAP_ACES = 0, 1, 2


def make_materials(specs):
    """specs: list of dicts(type=, baseColor=, metallic=, roughness=, ior=) -> structured array."""
    out = np.zeros(len(specs), dtype=MATERIAL_DTYPE)
    for i, s in enumerate(specs):
        out[i]["type"] = s.get("type", LAMBERTIAN)
        out[i]["baseColor"] = s.get("baseColor", (0.9, 0.9, 0.9))
        out[i]["metallic"] = s.get("metallic", 0.0)
        out[i]["roughness"] = s.get("roughness", 1.0)
        out[i]["ior"] = s.get("ior", 1.5)
        for k in ("baseColorMapId", "metallicMapId", "roughnessMapId", "normalMapId"):
            out[i][k] = -1
    return out


def make_camera(width, height, position, rotation, fov_y, focal_dist=1.0, lens_radius=0.0):
    """Fill the caller-set fields of Camera (src/scene.cpp:288-355 loadCamera); the derived
    fields (view/up/right/rotationMatInv) are set by the library's camera_update."""
    cam = Camera()
    cam.resolution[0], cam.resolution[1] = int(width), int(height)
    for i in range(3):
        cam.position[i] = float(position[i])
        cam.rotation[i] = float(rotation[i])
    cam.fov[0] = float(fov_y) * float(width) / float(height)
    cam.fov[1] = float(fov_y)
    cam.lensRadius = float(lens_radius)
    cam.focalDist = float(focal_dist)
    cam.tanFovY = float(np.tan(np.float32(np.radians(fov_y))))
    return cam


def copy_camera(cam):
    out = Camera()
    C.memmove(C.byref(out), C.byref(cam), C.sizeof(Camera))
    return out
