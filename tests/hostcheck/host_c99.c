/*
 * host_c99.c -- a host written against include/wr_abi.h alone, in C99, linked to libwr_hip.so.
 *
 * It stands where the reference's Rust host stands (src/core/state.rs:69-542 State::new,
 * :598-628 State::update, :630-809 State::render): load a GLB, hand meshes, textures,
 * materials and instances to the library, fill the three uniforms, render, read the frame
 * back.  Nothing here is Python or C++: what this program needs of the header and of the
 * library is what a Rust `extern "C"` block needs of them.
 *
 *   host_c99 --layout
 *       prints sizeof / offsetof of every struct of the header as JSON; touches no device.
 *   host_c99 <cube.glb> <out-prefix> [width height]
 *       BASELINE.json configs[0] (Cube.glb as geometry, one point light, PBS only, light
 *       marker on; instance scale 100, camera (-500, 150, 0) pitch -15, t = 2.0) at
 *       width x height (default 160 x 90); writes <out-prefix>.color (RGBA8),
 *       <out-prefix>.vis (u32) and <out-prefix>.depth (f32), row-major, top row first.
 *
 * Exit status: 0, or 1 with the library's error text on stderr.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wr_abi.h"

#define FIELD(S, f) printf("%s\"%s\": [%lu, %lu]", first ? "" : ", ", #f, (unsigned long)offsetof(S, f), (unsigned long)sizeof(((S*)0)->f)), first = 0
#define BEGIN(S) printf("%s\"%s\": {\"size\": %lu, \"fields\": {", any ? ", " : "", #S, (unsigned long)sizeof(S)), any = 1, first = 1
#define END() printf("}}")

static void print_layout(void) {
    int any = 0, first = 1;
    printf("{");
    BEGIN(WrModelVertex); FIELD(WrModelVertex, position); FIELD(WrModelVertex, tex_coords); FIELD(WrModelVertex, normal);
    FIELD(WrModelVertex, tangent); FIELD(WrModelVertex, bitangent); END();
    BEGIN(WrInstanceRaw); FIELD(WrInstanceRaw, model); FIELD(WrInstanceRaw, normal); END();
    BEGIN(WrCameraUniform); FIELD(WrCameraUniform, view); FIELD(WrCameraUniform, proj); FIELD(WrCameraUniform, inv_view_proj);
    FIELD(WrCameraUniform, position); FIELD(WrCameraUniform, planes); END();
    BEGIN(WrLightUniform); FIELD(WrLightUniform, position); FIELD(WrLightUniform, _padding); FIELD(WrLightUniform, color);
    FIELD(WrLightUniform, matrices); END();
    BEGIN(WrGlobalUniforms); FIELD(WrGlobalUniforms, time); FIELD(WrGlobalUniforms, light_matrix_index);
    FIELD(WrGlobalUniforms, use_shadowmaps); FIELD(WrGlobalUniforms, _padding); END();
    BEGIN(WrMaterialUniform); FIELD(WrMaterialUniform, factors); END();
    BEGIN(WrConfig); FIELD(WrConfig, width); FIELD(WrConfig, height); FIELD(WrConfig, shadow_map_size); FIELD(WrConfig, color_format);
    FIELD(WrConfig, device); FIELD(WrConfig, tile_rank); FIELD(WrConfig, tile_world); FIELD(WrConfig, stream);
    FIELD(WrConfig, frames_in_flight); FIELD(WrConfig, reserved); END();
    BEGIN(WrFrameStats); FIELD(WrFrameStats, ms_shadow); FIELD(WrFrameStats, ms_geometry); FIELD(WrFrameStats, ms_shade);
    FIELD(WrFrameStats, ms_fog); FIELD(WrFrameStats, ms_total); FIELD(WrFrameStats, profiled_frames);
    FIELD(WrFrameStats, geometry_bin_items); FIELD(WrFrameStats, shadow_bin_items); FIELD(WrFrameStats, fog_fragments);
    FIELD(WrFrameStats, fog_noise_evals); FIELD(WrFrameStats, shaded_fragments); FIELD(WrFrameStats, replay_pixels);
    FIELD(WrFrameStats, replay_slow_pixels); FIELD(WrFrameStats, ms_fog_march); FIELD(WrFrameStats, ms_fog_light_march);
    FIELD(WrFrameStats, fog_noise_evals_view); FIELD(WrFrameStats, fog_noise_evals_light); END();
    BEGIN(WrTransport); FIELD(WrTransport, user); FIELD(WrTransport, group_start); FIELD(WrTransport, send); FIELD(WrTransport, recv);
    FIELD(WrTransport, group_end); END();
    printf("}\n");
}

static int write_file(const char* prefix, const char* ext, const void* data, size_t bytes) {
    char path[1024];
    FILE* f;
    snprintf(path, sizeof(path), "%s.%s", prefix, ext);
    f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "host_c99: cannot write %s\n", path); return 1; }
    if (fwrite(data, 1, bytes, f) != bytes) { fclose(f); fprintf(stderr, "host_c99: short write to %s\n", path); return 1; }
    return fclose(f) != 0;
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != WR_OK) {                                                                          \
            fprintf(stderr, "host_c99: %s -> %d: %s\n", #call, rc_, ctx ? wr_last_error(ctx) : ""); \
            goto fail;                                                                               \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    WrContext* ctx = NULL;
    WrGlb* glb = NULL;
    WrConfig cfg;
    WrInstanceRaw inst;
    WrCameraUniform cam;
    WrLightUniform light;
    WrGlobalUniforms glob;
    uint32_t width = 160, height = 90, meshes = 0, materials = 0, textures = 0, i;
    unsigned char* color = NULL;
    uint32_t* vis = NULL;
    float* depth = NULL;
    size_t npx;
    char err[512];
    int status = 1;
    /* state.rs:407-412 instance of the geometry model, here the configs[0] framing */
    const float origin[3] = {0.0f, 0.0f, 0.0f}, identity[4] = {1.0f, 0.0f, 0.0f, 0.0f}, scale100[3] = {100.0f, 100.0f, 100.0f};
    /* state.rs:420-425 fog volume instance (not drawn in this configuration, set as State::new does) */
    const float fog_pos[3] = {0.0f, 30.0f, 0.0f}, fog_scale[3] = {1360.0f, 30.0f, 600.0f};
    const float eye[3] = {-500.0f, 150.0f, 0.0f};
    const float t = 2.0f;

    if (argc >= 2 && strcmp(argv[1], "--layout") == 0) { print_layout(); return 0; }
    if (argc < 3) { fprintf(stderr, "usage: host_c99 --layout | host_c99 <file.glb> <out-prefix> [width height]\n"); return 2; }
    if (argc >= 5) { width = (uint32_t)strtoul(argv[3], NULL, 10); height = (uint32_t)strtoul(argv[4], NULL, 10); }

    err[0] = 0;
    if (wr_glb_load_file(argv[1], &glb, err, sizeof(err)) != WR_OK) { fprintf(stderr, "host_c99: %s: %s\n", argv[1], err); return 1; }
    wr_glb_counts(glb, &meshes, &materials, &textures);

    memset(&cfg, 0, sizeof(cfg));
    cfg.width = width; cfg.height = height; cfg.shadow_map_size = 256;
    cfg.color_format = WR_FORMAT_RGBA8_UNORM_SRGB;
    cfg.frames_in_flight = 1;
    CHECK(wr_create(&cfg, &ctx));

    /* state.rs:379-405: geom_model through load_model_gltf; fog_model and light_model are the
     * same file's meshes, drawn without materials (light.rs:97-118, state.rs:794) */
    CHECK(wr_upload_glb(ctx, WR_MODEL_GEOMETRY, glb));
    for (i = 0; i < meshes; ++i) {
        const WrModelVertex* v = NULL;
        const uint32_t* idx = NULL;
        uint32_t nv = 0, ni = 0, mat = 0;
        CHECK(wr_glb_mesh(glb, i, &v, &nv, &idx, &ni, &mat));
        CHECK(wr_upload_mesh(ctx, WR_MODEL_FOG_VOLUME, v, nv, idx, ni, 0));
        CHECK(wr_upload_mesh(ctx, WR_MODEL_LIGHT_MARKER, v, nv, idx, ni, 0));
    }
    wr_instance_to_raw(&inst, origin, identity, scale100);
    CHECK(wr_set_instances(ctx, WR_MODEL_GEOMETRY, &inst, 1));
    wr_instance_to_raw(&inst, fog_pos, identity, fog_scale);
    CHECK(wr_set_instances(ctx, WR_MODEL_FOG_VOLUME, &inst, 1));

    /* State::update, state.rs:598-628 */
    wr_camera_uniform_update(&cam, eye, -15.0f, 0.0f, 55.0f, width, height);
    memset(&light, 0, sizeof(light));
    light.color[0] = light.color[1] = light.color[2] = 1.0f;
    light.color[3] = 250000.0f; /* state.rs:134 */
    wr_light_update_matrices(&light);
    wr_light_animate(&light, t);
    glob.time = t; glob.light_matrix_index = 0; glob.use_shadowmaps = 0; glob._padding = 0;

    /* State::render, state.rs:630-809, as the wasm32 configuration runs it: no shadow passes, no fog */
    CHECK(wr_render_frame(ctx, &cam, &light, &glob, WR_PASS_LIGHT_DEBUG));
    CHECK(wr_sync(ctx));

    npx = (size_t)width * height;
    color = (unsigned char*)malloc(npx * 4);
    vis = (uint32_t*)malloc(npx * 4);
    depth = (float*)malloc(npx * 4);
    if (!color || !vis || !depth) { fprintf(stderr, "host_c99: out of host memory\n"); goto fail; }
    CHECK(wr_read_color(ctx, color, npx * 4));
    CHECK(wr_read_visibility(ctx, vis, npx * 4));
    CHECK(wr_read_depth(ctx, depth, npx * 4));
    if (write_file(argv[2], "color", color, npx * 4) || write_file(argv[2], "vis", vis, npx * 4) || write_file(argv[2], "depth", depth, npx * 4))
        goto fail;
    printf("host_c99: %s: %u mesh(es), %u material(s), %u texture(s); %ux%u frame written to %s.{color,vis,depth}\n", argv[1],
           (unsigned)meshes, (unsigned)materials, (unsigned)textures, (unsigned)width, (unsigned)height, argv[2]);
    status = 0;
fail:
    free(color); free(vis); free(depth);
    if (ctx) wr_destroy(ctx);
    if (glb) wr_glb_free(glb);
    return status;
}
