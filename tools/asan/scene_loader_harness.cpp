#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "restir_hip.h"
// minimal stand-ins for the two library services scene_file.cpp uses besides rs_camera_update
static std::string g_err;
int rs_fail(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        rs_scene_file* f = nullptr;
        int e = rs_scene_file_load(argv[i], &f);
        if (e == 0) { rs_scene_file_view v; rs_scene_file_get(f, &v); ok++; rs_scene_file_free(f); } else bad++;
    }
    std::printf("loaded %d, refused %d\n", ok, bad);
    return 0;
}
