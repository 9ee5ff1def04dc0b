// headless_viewer.cpp -- the reference's main() / runCuda() call sequence (src/main.cpp:55-103,146-185)
// written against restir_compat.h, with the GLFW/ImGui/PBO display replaced by a PPM dump.
// It exists to show (and compile-check) that the reference's host code drives librestir_hip through
// the same names.  Scene input: a scene file in the reference's text format (Scene(filename), as main() does), or a
// binary triangle soup written by restir_amd/scenes.py (dump_scene).
//
//   headless_viewer scene.txt frames reuse out.ppm | out.png        (resolution and camera from the file; .png = saveImage)
//   headless_viewer scene.bin width height frames reuse out.ppm
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "restir_compat.h"

static Scene* scene = nullptr;
static GBuffer gBuffer;
static rsc::vec3* devDirectIllum = nullptr;
static int iteration = 0, width = 0, height = 0;

static void runCuda(void* devPBO) {                       // src/main.cpp:146-185
    if (!Settings::accumulate) State::camChanged = true;
    if (State::camChanged) { iteration = 0; scene->camera.update(); State::camChanged = false; }
    gBuffer.render(scene->devScene, scene->camera);
    if (Settings::useReservoir) ReSTIRDirect(devDirectIllum, iteration, gBuffer);
    else pathTraceDirect(devDirectIllum, iteration);
    copyImageToPBO(devPBO, devDirectIllum, width, height, Settings::toneMapping);
    iteration++;
    gBuffer.update(scene->camera);
}

static int load_soup(int argc, char** argv, int& frames) {
    if (argc < 7) return 1;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int32_t hdr[2];                                       // numPrims, numMaterials
    float camv[8];                                        // position[3], rotation[3], fovY, focalDist
    if (std::fread(hdr, 4, 2, f) != 2 || std::fread(camv, 4, 8, f) != 8) return 1;
    const size_t np = (size_t)hdr[0];
    std::vector<float> v(np * 9), n(np * 9), t(np * 6);
    std::vector<int> m(np);
    std::vector<Material> mats((size_t)hdr[1]);
    if (std::fread(v.data(), 4, np * 9, f) != np * 9 || std::fread(n.data(), 4, np * 9, f) != np * 9 ||
        std::fread(t.data(), 4, np * 6, f) != np * 6 || std::fread(m.data(), 4, np, f) != np ||
        std::fread(mats.data(), sizeof(Material), mats.size(), f) != mats.size()) return 1;
    std::fclose(f);

    width = std::atoi(argv[2]); height = std::atoi(argv[3]);
    frames = std::atoi(argv[4]);
    Settings::reservoirReuse = std::atoi(argv[5]);

    scene = new Scene();
    Camera& cam = scene->camera;
    std::memset(static_cast<rs_camera*>(&cam), 0, sizeof(rs_camera));
    cam.resolution[0] = width; cam.resolution[1] = height;
    for (int i = 0; i < 3; i++) { cam.position[i] = camv[i]; cam.rotation[i] = camv[3 + i]; }
    cam.fov[1] = camv[6]; cam.focalDist = camv[7];
    scene->buildDevData((int)np, v.data(), n.data(), t.data(), m.data(), hdr[1], mats.data());
    return 0;
}

int main(int argc, char** argv) {
    const size_t len = argc > 1 ? std::strlen(argv[1]) : 0;
    const bool text = len > 4 && std::strcmp(argv[1] + len - 4, ".txt") == 0;
    if (argc < (text ? 5 : 7)) {
        std::printf("Usage: %s scene.txt frames reuse out.ppm\n       %s scene.bin width height frames reuse out.ppm\n", argv[0], argv[0]);
        return 1;
    }
    rsc::check(rs_init(0), "init");
    int frames = 0;
    const char* outName = argv[text ? 4 : 6];
    if (text) {                                           // src/main.cpp:60-83
        scene = new Scene(std::string(argv[1]));
        width = scene->camera.resolution[0]; height = scene->camera.resolution[1];
        frames = std::atoi(argv[2]);
        Settings::reservoirReuse = std::atoi(argv[3]);
        scene->buildDevData();
    }
    else if (load_soup(argc, argv, frames)) return 1;
    State::scene = scene;
    if (hipMalloc((void**)&devDirectIllum, sizeof(rsc::vec3) * (size_t)width * height) != hipSuccess) return 1;
    (void)hipMemset(devDirectIllum, 0, sizeof(rsc::vec3) * (size_t)width * height);
    gBuffer.create(width, height);
    pathTraceInit();
    ReSTIRInit();
    void* devPBO = nullptr;
    if (hipMalloc(&devPBO, 4 * (size_t)width * height) != hipSuccess) return 1;

    for (int i = 0; i < frames; i++) runCuda(devPBO);       // mainLoop (src/preview.cpp:337-369)

    const size_t outLen = std::strlen(outName);
    if (outLen > 4 && std::strcmp(outName + outLen - 4, ".png") == 0) {
        // saveImage(false) (src/main.cpp:105-144): tone map + gamma of the radiance image, mirrored in x, as a PNG
        rsc::check(rs_save_image(outName, reinterpret_cast<const float*>(devDirectIllum), width, height, Settings::toneMapping), "saveImage");
    }
    else {                                                  // the bytes the PBO holds, as they are
        std::vector<unsigned char> rgba(4 * (size_t)width * height);
        (void)hipMemcpy(rgba.data(), devPBO, rgba.size(), hipMemcpyDeviceToHost);
        FILE* o = std::fopen(outName, "wb");
        std::fprintf(o, "P6\n%d %d\n255\n", width, height);
        for (size_t i = 0; i < (size_t)width * height; i++) std::fwrite(&rgba[4 * i], 1, 3, o);
        std::fclose(o);
    }

    scene->clear();
    gBuffer.destroy();
    pathTraceFree();
    ReSTIRFree();
    (void)hipFree(devPBO); (void)hipFree(devDirectIllum);
    delete scene;
    return 0;
}
