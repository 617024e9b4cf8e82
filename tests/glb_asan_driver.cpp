// Test driver (CPU only): the GLB/PNG/JPEG reader of libart compiled by g++ with AddressSanitizer + UBSan, run on one file.
// Prints "rc=<code> <message>" and exits 0 whatever the reader answers; a memory error is the sanitizer's report and a non-zero exit.
// art_scene_add_primitive / art_last_error live in art_api.hip (HIP runtime), which this build leaves out: the two symbols
// art_glb.hip references from it are defined here and never called (the driver does not add anything to a scene).
#include "../include/art.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" int32_t art_scene_add_primitive(ArtContext *, const ArtVertex *, uint32_t, const void *, uint32_t, uint32_t, const uint8_t *, uint32_t, uint32_t, const float *, uint32_t *) { std::abort(); }
extern "C" const char *art_last_error(void) { return ""; }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    for (int normalize = 0; normalize <= 1; normalize++) {
        ArtGlb *g = nullptr;
        int32_t rc = art_glb_open(argv[1], normalize, 2, &g);
        std::printf("rc=%d %s\n", rc, rc ? art_glb_last_error() : "ok");
        if (rc) continue;
        uint32_t n = 0; art_glb_primitive_count(g, &n);
        std::vector<ArtGlbCopyInfo> infos(n ? n : 1);
        size_t total = 0;
        rc = art_glb_copy_model_data(g, 1u | 2u | 4u | 8u | 16u, 1u | 2u | 4u, nullptr, 0, infos.data(), n, &total);
        if (rc == 0) { std::vector<uint8_t> blob(total); rc = art_glb_copy_model_data(g, 31u, 7u, blob.data(), blob.size(), infos.data(), n, &total); }
        std::printf("copy rc=%d total=%zu\n", rc, total);
        rc = art_glb_copy_model_data(g, 1u | 16u, 0, nullptr, 0, infos.data(), n, &total);   // positions + indices only: files without the other attributes
        if (rc == 0) { std::vector<uint8_t> blob(total); rc = art_glb_copy_model_data(g, 17u, 0, blob.data(), blob.size(), infos.data(), n, &total); }
        std::printf("copy(pos+idx) rc=%d total=%zu\n", rc, total);
        float c[3], r = 0;
        rc = art_glb_bounding_sphere(g, c, &r);
        std::printf("sphere rc=%d r=%g\n", rc, r);
        art_glb_close(g);
    }
    return 0;
}
