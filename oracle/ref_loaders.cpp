/*
 * ref_loaders.cpp -- extern "C" wrappers over the REFERENCE'S OWN file loaders, compiled from the sources where
 * they lie under /root/reference (nothing is copied into this repo):
 *   src/image.cpp             Image::Image(filename)          (stbi_loadf, src/image.cpp:14-31)
 *   src/stb.cpp               stb_image / stb_image_write implementation units
 *   src/tiny_obj_loader.cpp   tinyobjloader implementation unit
 *   src/utilities.cpp         utilityCore::safeGetline / tokenizeString (src/utilities.cpp:57-95)
 *
 * TEST INFRASTRUCTURE ONLY (builds oracle/_ref/libref_loaders.so, see oracle/Makefile).  It pins the scene-file
 * front end of the product (restir_amd/csrc/scene_file.cpp): the OBJ reader against tinyobj::LoadObj followed by the
 * flattening loop of Resource::loadOBJMesh (src/scene.cpp:41-50, restated below because scene.cpp needs Thrust / nvcc
 * and cannot be compiled here), the PPM decoder against Image(filename) under the two flip settings of
 * Scene::Scene (src/scene.cpp:97-98,124-126), and the line reader / tokenizer against the reference's own.
 */
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "image.h"           // /root/reference/src (via -I)
#include "utilities.h"
#include <stb_image.h>
#include <tiny_obj_loader.h>

extern "C" {

/* Image(filename) with stbi_ldr_to_hdr_gamma(1) and the given flip flag.  Returns 0 and fills w / h (and data when cap is
 * large enough); -1 when the file cannot be opened (Image's own failure path is `throw;` without an exception = terminate). */
int ref_image_load(const char* path, int flip, int* w, int* h, float* data, int cap) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return -1;
    std::fclose(f);
    int x = 0, y = 0, comp = 0;
    if (!stbi_info(path, &x, &y, &comp)) return -2;
    stbi_ldr_to_hdr_gamma(1.f);
    stbi_set_flip_vertically_on_load(flip);
    Image img{ std::string(path) };
    *w = img.width(); *h = img.height();
    if (cap >= img.width() * img.height() * 3) std::memcpy(data, img.data(), img.byteSize());
    return 0;
}

/* tinyobj::LoadObj + the loop of Resource::loadOBJMesh (scene.cpp:35-50).  Returns the corner count (3 per triangle) or
 * -1 when LoadObj fails; fills the arrays when cap (corners) is large enough. */
int ref_obj_load(const char* path, int cap, float* vertices, float* normals, float* texcoords) {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, nullptr, &warn, &err, path)) return -1;
    const bool hasTexcoord = !attrib.texcoords.empty();
    int n = 0;
    for (const auto& shape : shapes) n += (int)shape.mesh.indices.size();
    if (cap < n) return n;
    int i = 0;
    for (const auto& shape : shapes) {
        for (auto idx : shape.mesh.indices) {
            const glm::vec3 v = *((glm::vec3*)attrib.vertices.data() + idx.vertex_index);
            const glm::vec3 nn = *((glm::vec3*)attrib.normals.data() + idx.normal_index);
            const glm::vec2 t = hasTexcoord ? *((glm::vec2*)attrib.texcoords.data() + idx.texcoord_index) : glm::vec2(0.f);
            std::memcpy(vertices + 3 * i, &v, 12); std::memcpy(normals + 3 * i, &nn, 12); std::memcpy(texcoords + 2 * i, &t, 8);
            i++;
        }
    }
    return n;
}

/* Image(width, height) + setPixel for every pixel + saveJPG (src/image.cpp:8-12,36-39,60-74): writes baseFilename + ".jpg".
 * pixels: width * height * 3 floats, row-major. */
void ref_save_jpg(const char* baseFilename, int width, int height, const float* pixels) {
    Image img(width, height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float* p = pixels + ((size_t)y * width + x) * 3;
            img.setPixel(x, y, glm::vec3(p[0], p[1], p[2]));
        }
    img.saveJPG(std::string(baseFilename));
}

/* The read loop of Scene::Scene (scene.cpp:108-111): every line safeGetline returns while the stream is good, tokens
 * joined by '\t', lines by '\n'.  Returns the length written (or needed when cap is too small). */
int ref_read_lines(const char* path, char* out, int cap) {
    std::ifstream fp(path);
    if (!fp.is_open()) return -1;
    std::string all;
    while (fp.good()) {
        std::string line;
        utilityCore::safeGetline(fp, line);
        const std::vector<std::string> tokens = utilityCore::tokenizeString(line);
        for (size_t k = 0; k < tokens.size(); k++) { if (k) all += '\t'; all += tokens[k]; }
        all += '\n';
    }
    if ((int)all.size() <= cap) std::memcpy(out, all.data(), all.size());
    return (int)all.size();
}

}  // extern "C"
