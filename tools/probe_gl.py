#!/usr/bin/env python3
"""tools/probe_gl.py -- is there a GL stack on this box that HIP's graphics interop could register a buffer object from?
(the viewer's pixel-buffer object, src/preview.cpp:111-135, src/main.cpp:176-181.)
A surfaceless context needs libEGL (eglGetPlatformDisplay with the device or surfaceless platform) or an X display for GLX;
the image ships Mesa's libGL / libGLX only.  Prints what it finds; builds nothing."""
import ctypes
import ctypes.util
import glob
import os

print("DISPLAY =", os.environ.get("DISPLAY"), " WAYLAND_DISPLAY =", os.environ.get("WAYLAND_DISPLAY"))
for name in ("EGL", "GL", "OpenGL", "GLESv2", "gbm", "GLX", "OSMesa", "glfw"):
    print("find_library(%s) = %s" % (name, ctypes.util.find_library(name)))
print("/dev/dri:", sorted(glob.glob("/dev/dri/*")))
print("Mesa DRI drivers:", [os.path.basename(p) for p in glob.glob("/usr/lib/x86_64-linux-gnu/dri/*radeonsi*") + glob.glob("/usr/lib/x86_64-linux-gnu/dri/*swrast*")])
print("other libEGL copies:", [p for p in glob.glob("/usr/local/lib/python3*/dist-packages/*/executable/bin/swiftshader/libEGL.so")], "(SwiftShader: a CPU rasteriser, no device memory to share)")
egl = ctypes.util.find_library("EGL")
if egl:
    lib = ctypes.CDLL(egl)
    lib.eglGetProcAddress.restype = ctypes.c_void_p
    lib.eglGetProcAddress.argtypes = [ctypes.c_char_p]
    for fn in (b"eglGetPlatformDisplayEXT", b"eglQueryDevicesEXT", b"eglGLInteropExportObjectMESA"):
        print(fn.decode(), "->", hex(lib.eglGetProcAddress(fn) or 0))
else:
    print("no libEGL: no surfaceless GL context can be created; hipGraphicsGLRegisterBuffer has no context to register from")
try:
    gl = ctypes.CDLL(ctypes.util.find_library("GL"))
    gl.glXGetProcAddressARB.restype = ctypes.c_void_p
    gl.glXGetProcAddressARB.argtypes = [ctypes.c_char_p]
    print("glXGetProcAddressARB(MesaGLInteropGLXExportObject) ->", hex(gl.glXGetProcAddressARB(b"MesaGLInteropGLXExportObject") or 0), "(GLX needs an X display)")
except Exception as e:
    print("libGL:", e)
