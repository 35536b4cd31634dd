/* Plain-C client of libneddf_hip.so (mode B of INTEGRATION.md): no Python, no torch.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/capi/c_smoke.c \
 *       -L neddf_amd/csrc -lneddf_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/neddf_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_smoke
 *   ./c_smoke out.bin
 *
 * Builds the shipped NeDDF architecture with weights from a fixed linear congruential generator (the pytest side regenerates
 * the same numbers), renders 96 rays through neddf_render_rays (65 coarse + 129 importance samples, all outputs incl.
 * fields_penalty), a second time through the eval-minimal path (pixels only), gathers the pixels over a one-rank communicator
 * and writes everything to `out.bin` for tests/test_gpu_multi.py::test_c_client_of_the_abi to compare with the Python binding.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neddf_hip.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        int rc_ = (call);                                                                                  \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, neddf_last_error(ctx)); return 2; } \
    } while (0)
#define HIP(call)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)

static uint32_t lcg_state = 12345u;
static float lcg(void)          /* uniform in [-1, 1), the same recurrence in numpy on the pytest side */
{
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((double)(lcg_state >> 8) / 8388608.0 - 1.0);
}

enum { RAYS = 96, SC = 64, SF = 128, TRUNK = 7, COL = 3 };

int main(int argc, char **argv)
{
    neddf_ctx *ctx = NULL;
    if (argc < 2) return 1;
    if (neddf_abi_version() != NEDDF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    CHECK(neddf_create(0, &ctx));

    /* config/network/neddf.yaml: 8-layer distance trunk with a skip at 4, 4-layer colour trunk, width 256 */
    neddf_field_desc d;
    memset(&d, 0, sizeof d);
    d.kind = NEDDF_FIELD_NEDDF; d.embed_pos_rank = 10; d.embed_dir_rank = 4;
    d.layer_count = 8; d.layer_width = 256; d.col_layer_count = 4; d.col_layer_width = 256;
    d.n_skips = 1; d.skips[0] = 4; d.activation = NEDDF_ACT_TANHEXP; d.density_activation = NEDDF_ACT_LEAKY; d.d_near = 0.001f;
    const float pw[6] = { 0.05f, 0.5f, 1.0f, 1.0f, 0.1f, 0.0001f };
    for (int k = 0; k < 6; ++k) { d.penalty_weight[k] = pw[k]; d.penalty_has[k] = 1; }
    d.weight_dtype = NEDDF_DTYPE_F32;
    const int n_t = TRUNK + COL + 3;
    int in_dim[TRUNK + COL + 3], out_dim[TRUNK + COL + 3];
    for (int l = 0; l < TRUNK; ++l) { in_dim[l] = l == 0 ? 60 : (l == 5 ? 316 : 256); out_dim[l] = 256; }
    for (int l = 0; l < COL; ++l) { in_dim[TRUNK + l] = l == 0 ? 343 : 256; out_dim[TRUNK + l] = 256; }
    in_dim[TRUNK + COL] = 256; out_dim[TRUNK + COL] = 1;            /* layer_ddf_out */
    in_dim[TRUNK + COL + 1] = 256; out_dim[TRUNK + COL + 1] = 1;    /* layer_aux_out */
    in_dim[TRUNK + COL + 2] = 256; out_dim[TRUNK + COL + 2] = 3;    /* layer_col_out */
    float *W[TRUNK + COL + 3], *B[TRUNK + COL + 3];
    for (int t = 0; t < n_t; ++t) {
        const float s = sqrtf(2.0f / (float)(in_dim[t] + out_dim[t]));
        W[t] = (float *)malloc(sizeof(float) * in_dim[t] * out_dim[t]);
        B[t] = (float *)malloc(sizeof(float) * out_dim[t]);
        for (int i = 0; i < in_dim[t] * out_dim[t]; ++i) W[t][i] = 1.7f * s * lcg();
        for (int i = 0; i < out_dim[t]; ++i) B[t][i] = 0.05f * lcg();
    }
    for (int slot = 0; slot < 2; ++slot) {
        CHECK(neddf_set_field(ctx, slot, &d, (const float *const *)W, (const float *const *)B, n_t));
        float lowpass[10];
        for (int e = 0; e < 10; ++e) lowpass[e] = 1.0f;
        CHECK(neddf_set_iter(ctx, slot, 1.1f, 2.0f, lowpass));
    }

    /* rays of a 400 x 400 view looking at the origin from z = +4 */
    neddf_camera cam;
    const float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(cam.R, R, sizeof R);
    cam.T[0] = 0.1f; cam.T[1] = -0.05f; cam.T[2] = 4.0f;
    cam.calib[0] = cam.calib[1] = 555.6f; cam.calib[2] = cam.calib[3] = 200.0f;
    neddf_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.sample_coarse = SC; rp.sample_fine = SF; rp.dist_near = 2.0f; rp.dist_far = 6.0f; rp.max_dist = 6.0f;
    rp.cone_sampling = 1; rp.ray_radius = 1.0 / 1111.0 / sqrt(12.0);
    int64_t h_uv[RAYS * 2];
    static float h_uc[RAYS * (SC + 1)], h_uf[RAYS * (SF + 1)];
    for (int r = 0; r < RAYS; ++r) { h_uv[2 * r] = 140 + (r * 7) % 120; h_uv[2 * r + 1] = 150 + (r * 11) % 100; }
    for (int i = 0; i < RAYS * (SC + 1); ++i) h_uc[i] = 0.5f * (lcg() + 1.0f);
    for (int i = 0; i < RAYS * (SF + 1); ++i) h_uf[i] = 0.5f * (lcg() + 1.0f);

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    void *d_uv, *d_uc, *d_uf;
    float *d_out;           /* [full: color 3 | depth | trans | penalty][minimal: color 3 | depth | trans][packed 5][gathered 5] per ray */
    int *d_flag;
    HIP(hipMalloc(&d_uv, sizeof h_uv)); HIP(hipMalloc(&d_uc, sizeof h_uc)); HIP(hipMalloc(&d_uf, sizeof h_uf));
    HIP(hipMalloc((void **)&d_out, sizeof(float) * RAYS * 21)); HIP(hipMalloc((void **)&d_flag, sizeof(int)));
    HIP(hipMemcpy(d_uv, h_uv, sizeof h_uv, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_uc, h_uc, sizeof h_uc, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_uf, h_uf, sizeof h_uf, hipMemcpyHostToDevice));
    HIP(hipMemset(d_flag, 0, sizeof(int)));
    float *full = d_out, *mini = d_out + RAYS * 6, *packed = d_out + RAYS * 11, *gathered = d_out + RAYS * 16;

    neddf_render_outputs o;
    memset(&o, 0, sizeof o);
    o.color = full; o.depth = full + RAYS * 3; o.transmittance = full + RAYS * 4; o.fields_penalty = full + RAYS * 5; o.nan_flag = d_flag;
    CHECK(neddf_render_rays(ctx, d_uv, NEDDF_UV_I64, RAYS, &cam, &rp, (const float *)d_uc, (const float *)d_uf, &o, stream));
    memset(&o, 0, sizeof o);
    o.color = mini; o.depth = mini + RAYS * 3; o.transmittance = mini + RAYS * 4; o.nan_flag = d_flag;
    CHECK(neddf_render_rays(ctx, d_uv, NEDDF_UV_I64, RAYS, &cam, &rp, (const float *)d_uc, (const float *)d_uf, &o, stream));

    /* pixels [ray][5] and the library's own all-gather on a one-rank communicator */
    HIP(hipStreamSynchronize(stream));
    static float h_all[RAYS * 21];
    HIP(hipMemcpy(h_all, d_out, sizeof(float) * RAYS * 11, hipMemcpyDeviceToHost));
    static float h_packed[RAYS * 5];
    for (int r = 0; r < RAYS; ++r) {
        for (int k = 0; k < 3; ++k) h_packed[5 * r + k] = h_all[RAYS * 6 + 3 * r + k];
        h_packed[5 * r + 3] = h_all[RAYS * 9 + r];
        h_packed[5 * r + 4] = h_all[RAYS * 10 + r];
    }
    HIP(hipMemcpy(packed, h_packed, sizeof h_packed, hipMemcpyHostToDevice));
    char id[NEDDF_COMM_ID_BYTES];
    CHECK(neddf_comm_unique_id(ctx, id));
    CHECK(neddf_comm_init(ctx, 0, 1, id));
    int rank = -1, nranks = -1, version = 0;
    CHECK(neddf_comm_info(ctx, &rank, &nranks, &version));
    CHECK(neddf_gather_pixels(ctx, packed, RAYS, 5, gathered, stream));
    CHECK(neddf_comm_wait(ctx, stream));
    CHECK(neddf_comm_wait_host(ctx, 20000));
    HIP(hipStreamSynchronize(stream));
    CHECK(neddf_comm_destroy(ctx));

    int h_flag = -1;
    HIP(hipMemcpy(h_all, d_out, sizeof h_all, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost));
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 4;
    fwrite(h_all, sizeof(float), RAYS * 21, f);
    fclose(f);
    printf("c_smoke ok: %d rays, nan_flag %d, communicator %d/%d, rccl %d, pixel[0] = %.6f %.6f %.6f depth %.5f\n", RAYS, h_flag, rank, nranks,
           version, h_all[RAYS * 6], h_all[RAYS * 6 + 1], h_all[RAYS * 6 + 2], h_all[RAYS * 9]);
    neddf_destroy(ctx);
    return h_flag == 0 && rank == 0 && nranks == 1 ? 0 : 5;
}
