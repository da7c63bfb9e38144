/* c_abi_client.c -- the drop-in boundary from plain C: no Python, no torch, only include/clive2_amd.h and
 * libclive2_amd.so.  Reads a scene as the raw arrays the reference's create_scene builds (src/scene.py:71-89, layouts of
 * src/struct_types.py), renders n samples, writes the accumulators.
 *
 *   gcc -O2 -I include examples/c_abi_client.c -o c_abi_client -L clive2_amd -lclive2_amd -Wl,-rpath,$PWD/clive2_amd
 *   ./c_abi_client <dir with boxes.bin triangles.bin materials.bin camera.bin light_triangles.bin light_areas.bin
 *                   light_indices.bin seeds.bin> <width> <height> <samples>
 * writes <dir>/summed_image.bin (H*W*3 f32), summed_weights.bin (H*W f32), counts.bin (H*W i32), unidirectional.bin. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "clive2_amd.h"

static void* slurp(const char* dir, const char* name, size_t* bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n : 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read of %s\n", path); exit(2); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

static void dump(const char* dir, const char* name, const void* p, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

#define CHECK(call)                                                                             \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != CL2_OK) {                                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cl2_last_error(r));             \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s <scene dir> <width> <height> <samples>\n", argv[0]); return 2; }
    const char* dir = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), samples = atoi(argv[4]);
    size_t nb, nt, nm, nc, nl, na, ni, ns;
    void* boxes = slurp(dir, "boxes.bin", &nb);            /* Box 48 B      */
    void* tris = slurp(dir, "triangles.bin", &nt);         /* Triangle 128 B */
    void* mats = slurp(dir, "materials.bin", &nm);         /* Material 48 B */
    void* cam = slurp(dir, "camera.bin", &nc);             /* Camera 112 B  */
    void* ltris = slurp(dir, "light_triangles.bin", &nl);
    float* areas = slurp(dir, "light_areas.bin", &na);
    int32_t* lidx = slurp(dir, "light_indices.bin", &ni);
    uint32_t* seeds = slurp(dir, "seeds.bin", &ns);
    if (nc != 112 || nl / 128 != na / 4 || na / 4 != ni / 4 || ns != (size_t)W * H * 8) { fprintf(stderr, "inconsistent scene files\n"); return 2; }

    cl2_renderer* r = NULL;
    if (cl2_create(0, W, H, &r) != CL2_OK) { fprintf(stderr, "cl2_create: %s\n", cl2_last_error(NULL)); return 1; }
    CHECK(cl2_upload_scene(r, boxes, (int)(nb / 48), tris, (int)(nt / 128), mats, (int)(nm / 48), cam, ltris, areas, lidx, (int)(na / 4)));
    CHECK(cl2_set_seeds(r, seeds, (size_t)W * H * 2));
    CHECK(cl2_run_samples(r, samples));

    const size_t B = (size_t)W * H;
    float* img = malloc(B * 3 * sizeof(float)); float* wts = malloc(B * sizeof(float));
    int32_t* cnt = malloc(B * sizeof(int32_t)); float* uni = malloc(B * 3 * sizeof(float));
    CHECK(cl2_read_accumulators(r, img, wts, cnt, uni, B));
    cl2_counters c;
    CHECK(cl2_read_counters(r, &c));
    dump(dir, "summed_image.bin", img, B * 3 * sizeof(float));
    dump(dir, "summed_weights.bin", wts, B * sizeof(float));
    dump(dir, "counts.bin", cnt, B * sizeof(int32_t));
    dump(dir, "unidirectional.bin", uni, B * 3 * sizeof(float));
    printf("abi %d, %d x %d, %d samples, %llu rays\n", cl2_abi_version(), W, H, samples, (unsigned long long)c.rays);
    cl2_destroy(r);
    return 0;
}
