// viewer_io.hip -- what the interactive viewer does with the frame besides rendering it:
//
//   the OpenGL pixel-buffer object the frame is tone-mapped into          src/preview.cpp:112-133 cudaGLSetGLDevice / cudaGLRegisterBufferObject
//                                                                         src/main.cpp:176-181    cudaGLMapBufferObject / cudaGLUnmapBufferObject
//                                                                         src/preview.cpp:88      cudaGLUnregisterBufferObject
//   saveImage: tone map + gamma, mirrored in x, written as a PNG           src/main.cpp:105-144, src/image.cpp:36-58
//
// The PBO path is HIP's graphics interop (hipGraphicsGLRegisterBuffer and friends): the buffer object of the viewer's GL context
// is mapped for one frame, rs_copy_image_to_pbo writes into the mapped pointer, the buffer is unmapped before GL draws from it.
// Without a GL context in the process (a headless node) registration fails with the runtime's error and nothing else happens.
#include <hip/hip_runtime.h>
#include <hip/hip_gl_interop.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "rs_internal.h"

struct rs_pbo {
    rs_context* ctx = nullptr;
    hipGraphicsResource* res = nullptr;
    bool mapped = false;
};

namespace {

// ---- PNG, 8-bit RGB, filter 0, zlib stream of stored blocks (no compression: a screenshot is written once) ----------------------
uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x); }
void chunk(std::vector<unsigned char>& out, const char type[4], const std::vector<unsigned char>& data) {
    put32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32_update(0xffffffffu, out.data() + at, out.size() - at) ^ 0xffffffffu);
}

}  // namespace

extern "C" {

// Image::savePNG's file (src/image.cpp:41-58) from bytes that are already clamped and scaled: rgb = height rows of width * 3 bytes.
int rs_write_png(const char* path, const unsigned char* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_write_png: bad argument");
    const size_t row = (size_t)width * 3, rawSize = (row + 1) * (size_t)height;
    std::vector<unsigned char> raw(rawSize);
    for (int y = 0; y < height; y++) {
        raw[(size_t)y * (row + 1)] = 0;                                       // filter type 0
        std::memcpy(&raw[(size_t)y * (row + 1) + 1], rgb + (size_t)y * row, row);
    }
    std::vector<unsigned char> z;
    z.reserve(rawSize + rawSize / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);                                      // zlib header: deflate, 32 K window, no preset dictionary
    uint32_t a = 1, b = 0;                                                     // Adler-32 of the raw data
    for (size_t off = 0; off < rawSize; off += 65535) {
        const size_t n = rawSize - off < 65535 ? rawSize - off : 65535;
        z.push_back(off + n == rawSize ? 1 : 0);                               // BFINAL, BTYPE = 00 (stored)
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        for (size_t i = 0; i < n; i++) { a += raw[off + i]; if (a >= 65521u) a -= 65521u; b += a; if (b >= 65521u) b -= 65521u; }
    }
    put32(z, (b << 16) | a);
    std::vector<unsigned char> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);      // 8 bits, colour type 2 (RGB), deflate, filter 0, no interlace
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, (std::string("rs_write_png: cannot open ") + path).c_str());
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok) return rs_fail(RS_ERR_INVALID_ARGUMENT, (std::string("rs_write_png: cannot write ") + path).c_str());
    return 0;
}

// saveImage(false) (src/main.cpp:105-144): every pixel through the tone map and the gamma curve, stored at (width - 1 - x, y), clamped
// to [0, 1], scaled by 255 and truncated (src/image.cpp:44-50) -- the bytes sendImageToPBO produces with scale 1 (src/pathtrace.cu:41-55),
// mirrored in x.  `path` is the complete file name (the viewer appends ".<time>.<samples>samp.png" to its image name).
static int save_image(const char* path, const float* devImage, int width, int height, int toneMapping, bool jpg);
int rs_save_image(const char* path, const float* devImage, int width, int height, int toneMapping) { return save_image(path, devImage, width, height, toneMapping, false); }
// saveImage(true): the same bytes through Image::saveJPG (src/image.cpp:60-74, stbi_write_jpg at quality 90), jpeg_writer.cpp
int rs_save_image_jpg(const char* path, const float* devImage, int width, int height, int toneMapping) { return save_image(path, devImage, width, height, toneMapping, true); }
static int save_image(const char* path, const float* devImage, int width, int height, int toneMapping, bool jpg) {
    rs_ctx_scope scope(nullptr);
    if (!path || !devImage || width <= 0 || height <= 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_save_image: bad argument");
    const size_t n = (size_t)width * height;
    unsigned char* dev = nullptr;
    RS_TRY(rs_dev_alloc(&dev, n * 4));
    int e = rs_copy_image_to_pbo(dev, devImage, width, height, toneMapping, 1.f);
    if (!e) e = rs_denoise_join();                      // (the conversion of an image the denoise stream wrote ran there)
    std::vector<unsigned char> rgba(n * 4);
    if (!e) e = rs_check_hip(hipStreamSynchronize(rs_stream()), "rs_save_image");
    if (!e) e = rs_check_hip(hipMemcpy(rgba.data(), dev, n * 4, hipMemcpyDeviceToHost), "rs_save_image");
    rs_dev_free(dev);
    if (e) return e;
    std::vector<unsigned char> rgb(n * 3);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const unsigned char* s = &rgba[((size_t)y * width + x) * 4];
            unsigned char* d = &rgb[((size_t)y * width + (width - 1 - x)) * 3];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        }
    return jpg ? rs_write_jpg(path, rgb.data(), width, height) : rs_write_png(path, rgb.data(), width, height);
}

// cudaGLRegisterBufferObject(pbo) (src/preview.cpp:133): glBuffer is the GLuint of the viewer's GL_PIXEL_UNPACK_BUFFER, created in the GL
// context that is current on the calling thread.  The contents are overwritten every frame (write-discard).
int rs_pbo_register(unsigned glBuffer, rs_pbo** out) {
    if (!out) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_pbo_register: null");
    *out = nullptr;
    rs_pbo* p = new rs_pbo();
    p->ctx = rs_ctx();
    rs_ctx_scope scope(p->ctx);
    const hipError_t e = hipGraphicsGLRegisterBuffer(&p->res, (GLuint)glBuffer, hipGraphicsRegisterFlagsWriteDiscard);
    if (e != hipSuccess || !p->res) {
        (void)hipGetLastError();
        delete p;
        return rs_fail(e != hipSuccess ? (int)e : RS_ERR_UNSUPPORTED, (std::string("rs_pbo_register: hipGraphicsGLRegisterBuffer failed (is an OpenGL context current on this thread?): ") + hipGetErrorString(e)).c_str());
    }
    *out = p;
    return 0;
}

// cudaGLMapBufferObject((void**)&devPBO, pbo) (src/main.cpp:177): the device pointer is valid until rs_pbo_unmap; the map is ordered
// on the library stream, so rs_copy_image_to_pbo may follow at once.
int rs_pbo_map(rs_pbo* p, void** devPBO, size_t* bytes) {
    RS_SCOPE(p);
    if (!p || !devPBO) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_pbo_map: null");
    if (p->mapped) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_pbo_map: already mapped");
    RS_HIP(hipGraphicsMapResources(1, &p->res, rs_stream()));
    size_t n = 0;
    const hipError_t e = hipGraphicsResourceGetMappedPointer(devPBO, &n, p->res);
    if (e != hipSuccess) { (void)hipGraphicsUnmapResources(1, &p->res, rs_stream()); return rs_check_hip(e, "rs_pbo_map"); }
    if (bytes) *bytes = n;
    p->mapped = true;
    return 0;
}

// cudaGLUnmapBufferObject(pbo) (src/main.cpp:181): after the work enqueued on the library stream so far; GL may use the buffer afterwards
int rs_pbo_unmap(rs_pbo* p) {
    RS_SCOPE(p);
    if (!p) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_pbo_unmap: null");
    if (!p->mapped) return 0;
    p->mapped = false;
    RS_TRY(rs_denoise_join());                          // (rs_set_denoise_stream(1): the tone map of a filtered image ran on that stream)
    RS_HIP(hipGraphicsUnmapResources(1, &p->res, rs_stream()));
    return rs_after_launch("rs_pbo_unmap");
}

// cudaGLUnregisterBufferObject(pbo) (src/preview.cpp:88)
int rs_pbo_unregister(rs_pbo* p) {
    RS_SCOPE(p);
    if (!p) return 0;
    int e = 0;
    if (p->mapped) e = rs_pbo_unmap(p);
    if (p->res) { const int r = rs_check_hip(hipGraphicsUnregisterResource(p->res), "rs_pbo_unregister"); if (!e) e = r; }
    delete p;
    return e;
}

}  // extern "C"
