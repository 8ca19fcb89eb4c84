/* Compiled as C99 by tests/test_c_abi.py: include/mrt.h must be a plain C header,
 * the AoS mirrors must have the byte layout of the reference's packed structs
 * (CLStructs.hpp:8-128) and a C program must be able to drive the loader and the
 * band helpers without a GPU. */
#include "mrt.h"

#include <stddef.h>
#include <stdio.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { printf("FAILED: %s\n", #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    CHECK(sizeof(mrt_cl_float3) == 16);
    CHECK(sizeof(mrt_cl_float2) == 8);
    CHECK(sizeof(mrt_cl_rgb) == 16);
    CHECK(sizeof(mrt_cl_light) == 48);
    CHECK(sizeof(mrt_cl_cam) == 64);
    CHECK(sizeof(mrt_cl_vp) == 32);
    CHECK(sizeof(mrt_cl_ray) == 32);
    CHECK(sizeof(mrt_cl_phong) == 32);
    CHECK(sizeof(mrt_cl_triangle) == 80);
    CHECK(sizeof(mrt_cl_circle) == 48);
    CHECK(sizeof(mrt_cl_object) == 144);
    CHECK(sizeof(mrt_cl_intersect) == 80);
    CHECK(offsetof(mrt_cl_object, circleObj) == 80);
    CHECK(offsetof(mrt_cl_object, is_triangle) == 128);
    CHECK(offsetof(mrt_cl_object, is_circle) == 129);
    CHECK(offsetof(mrt_cl_triangle, shininess) == 16);
    CHECK(offsetof(mrt_cl_triangle, a) == 20);
    CHECK(offsetof(mrt_cl_triangle, b) == 36);
    CHECK(offsetof(mrt_cl_triangle, c) == 52);
    CHECK(offsetof(mrt_cl_circle, center) == 20);
    CHECK(offsetof(mrt_cl_circle, radius) == 36);
    CHECK(offsetof(mrt_cl_intersect, normal) == 16);
    CHECK(offsetof(mrt_cl_intersect, intersects) == 32);
    CHECK(offsetof(mrt_cl_intersect, t_) == 33);
    CHECK(offsetof(mrt_cl_intersect, shininess) == 60);
    CHECK(offsetof(mrt_cl_intersect, color) == 64);
    CHECK(offsetof(mrt_cl_phong, ambient_color) == 12);
    CHECK(mrt_abi_version() == MRT_ABI_VERSION);
    /* configuration structs of the handles: plain 32-bit fields in this order */
    CHECK(sizeof(mrt_config) == 8);
    CHECK(sizeof(mrt_sequence_config) == 12 && offsetof(mrt_sequence_config, in_flight) == 8);
    CHECK(offsetof(mrt_camera, xres) == 48 && offsetof(mrt_camera, projection) == 72 && sizeof(mrt_camera) == 80);
    /* no GPU needed to be told so: NULL handles are refused, not dereferenced */
    CHECK(mrt_sequence_wait(NULL) == MRT_ERR_INVALID);
    CHECK(mrt_sequence_submit_host(NULL, NULL, NULL) == MRT_ERR_INVALID);
    CHECK(mrt_sequence_in_flight(NULL) == 0);

    if (argc > 1) {
        mrt_loaded_scene *ls = NULL;
        char err[256];
        int rc = mrt_scene_load_file(argv[1], &ls, err, sizeof err);
        if (rc != MRT_OK) { printf("load failed: %s\n", err); return 1; }
        const mrt_scene *s = mrt_scene_view(ls);
        printf("%u %u %u %u\n", s->xres, s->yres, s->n_spheres, s->n_tris);
        mrt_scene_free(ls);
    }
    CHECK(mrt_band_rows(2160, 64, 0, 8) + mrt_band_rows(2160, 64, 7, 8) == 320 + 256);
    CHECK(strlen(mrt_last_error(NULL)) == 0 || 1);
    printf("ok\n");
    return 0;
}
