// kernels.h -- argument blocks shared by the HIP kernels and the C-ABI host code.
// gfx950 only (wave64, f32 MFMA 32x32x2); no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace neddf {

constexpr int kWaves = 4;            // waves per workgroup (one per SIMD)
constexpr int kThreads = 64 * kWaves;
constexpr int kActLd = 260;          // LDS row stride (floats): 256 + 4 -> conflict-free ds_read_b128 of MFMA A fragments
constexpr int kWidth = 256;          // engine width of the shipped configurations (and of the training kernels); other hidden widths run on 128 / 384 / 512
constexpr int kMaxWidth = 512;
constexpr int kMaxLayers = 12;
constexpr int kMaxStash = 4;         // early partials per field: skip connections (+ the NeRF colour head's direction segment)
constexpr int kSchedInts = 16;          // tile-queue head (+ padding)
constexpr int kPtAux = 16;           // floats per point handed from the distance kernel to the colour kernel
// floats of one stash slot of one workgroup: 4 waves x (MT=4 x NT=2 x 4 float4) x 64 lanes x 4
constexpr size_t kStashFloatsPerWg = (size_t)kWaves * (4 * 2 * 4) * 64 * 4;

// Sample points taken straight from the rays (round 5: SURVEY section 7 step 6, the cone moments in the field prologue): the
// reverse-mode distance kernel derives position / variance / direction of point i from ray i / S and its distances
// (device_math.h sample_moments: the stand-alone sampling kernel's own arithmetic, bit for bit) and hands them to the colour
// kernel INSIDE the per-point record it writes anyway -- the [N, 3] x 3 sampling tensors (36 B per point written and 60 B read)
// and the sampling launch disappear from neddf_render_rays' eval-minimal route.
struct RaySrc {
    const float *rd = nullptr, *ro = nullptr, *view = nullptr, *dists = nullptr;   // [B, 3], [B, 3], [B, 3] or NULL, [B, S]
    int S = 0;              // samples per ray
    float r2 = 0.f;         // ray_radius^2 (cone sampling)
    int cone = 0;
    int64_t base = 0;       // index of this launch's first point in the [B, S] grid
    double radius = -1.0;   // host side: the caller's ray radius as given (launch_sampling squares it itself), < 0 for point samples
};

// per-point record written by the distance-trunk kernel (float index); with RaySrc in eval-minimal mode the slots the colour kernel
// does not read there carry the sample point: PA_R_DIR (0..2), PA_R_POS (8..10), PA_R_VAR (11..13)
enum { PA_R_DIR = 0, PA_R_POS = 8, PA_R_VAR = 11 };
enum { PA_D = 0, PA_RHO = 1, PA_AUX = 2, PA_N0 = 3, PA_N1 = 4, PA_N2 = 5, PA_DDF_RAW = 6, PA_AUX_RAW = 7,
       PA_DG0 = 8, PA_DG1 = 9, PA_DG2 = 10, PA_AGG0 = 11, PA_AGG1 = 12, PA_AGG2 = 13, PA_DGN = 14, PA_DDDT = 15 };

// One dense layer in "fragment-major" packing (see pack_layer() in neddf_capi.hip):
//   wp[(((wave*NT + t)*ksteps + S)*64 + lane)*4 + r] = W[k = 8S + 4*(lane>>5) + r][n = (wave*NT + t)*32 + (lane&31)]
// so that one global_load_dwordx4 per lane yields the B operands of four
// v_mfma_f32_32x32x2_f32 k-steps, and a wave's load is 1 KiB contiguous.
struct LayerW {
    const float *wp;       // packed weights of the act-segment (K = 8*ksteps)
    const float *bias;     // [nout]
    int ksteps;            // super-steps of 8 k's
    int stash;             // -1, or index of the early partial (input-feature segment of a skip layer) to add
};

struct StashW {
    const float *wp;       // packed weights of the input-feature segment
    int col0;              // first act column of the segment (multiple of 8)
    int ksteps;
};

struct EncodeDesc {
    int E, Ed;             // embed_pos_rank, embed_dir_rank
    int KH, KD;            // padded half-widths: roundup(3E,4), roundup(3Ed,4)
    float lowpass[10];     // get_lowpass_scale per frequency
};

// Distance trunk of NeDDF (neddf.py:193-241): PE -> n_layers x (LinearGrad + activation) with
// (value, d/dx, d/dy, d/dz) rows -> distance / aux-gradient heads -> density, normal.
struct DdfArgs {
    const float *pos, *dir, *var;
    int64_t n_points;
    EncodeDesc enc;
    int n_layers;
    int activation, density_activation;
    LayerW layer[kMaxLayers];
    int n_stash;
    StashW stash[kMaxStash];
    int width;                            // engine width: hidden width padded to 128 / 256 / 384 / 512 (padding = zero weights: exact)
    const float *w_ddf_out, *w_aux_out;   // [width] each
    float b_ddf_out, b_aux_out;
    float d_near, aux_grad_scale;
    int operands;                         // 0 fp32 (32x32x2 f32 MFMA), 1 bf16, 2 split fp16 (three fp16 products per multiply-add), see tile_engine.h
    int neus;                             // 1: NeuS sdf trunk (neus.py:118-145): plain PE, no heads, sdf = feature 0
    float neus_v10;                       // variance * 10
    float *scratch;                       // per-workgroup stash area
    // reverse-mode distance gradient (ddf_rev_kernel, eval-minimal): transposed weights and a per-workgroup scratch
    const float *wT[kMaxLayers];          // [l >= 1] packed (hidden rows of W_l)^T, width x width
    const float *wT_pe0;                  // packed [width x 64]: W_0^T (engine column order of the encoding)
    const float *wT_pe_skip[kMaxStash];   // ... and the encoding rows^T of the skip layer that owns stash[s]
    int skip_layer;                       // a trunk layer whose input is cat([encoding, h]) (the last one), or -1
    int ks_hidden;                        // super-steps of a width-wide product under this operand policy
    float *rev_scratch;                   // per workgroup: y' of every layer [n_layers][P][width] + encoding Jacobian, copy and parked gradient [P][192]
    int *sched;                           // [0] tile queue head (zeroed before each launch)
    int sched_flags;                      // bit 1: dynamic tile queue (always set by the library)
    float *features;                      // [n_points][feat_rows][width]
    int feat_rows;                        // 1 (value row) or 4 (value + Jacobian rows)
    float *ptaux;                         // [n_points][kPtAux]
    float *distance, *density, *aux_grad; // optional outputs [n_points]
    RaySrc rays;                          // rays.rd != NULL: pos / dir / var are NULL, the points come from the rays (ddf_rev_kernel)
    unsigned long long *stamps = nullptr; // -DNEDDF_STAMP builds only (`make stamp`, tools/stamp_timeline.py): phase time stamps of a few workgroups
};
#ifndef NEDDF_STAMP_TILE_INDEX
#define NEDDF_STAMP_TILE_INDEX 6
#endif
constexpr int kStampBlocks = 8, kStampSlots = 160, kStampTile = NEDDF_STAMP_TILE_INDEX;      // workgroups stamped, stamps per wave, which tile of the workgroup
// (-DNEDDF_STAMP_TILE_INDEX=<k>, `make stamp STAMP_TILE=<k>`: a tile in the middle or at the end of a launch instead of its 7th)
constexpr int kStampWgTail = 4096;     // behind the stamps, four arrays of one word per workgroup: tiles it took | XCC_ID << 20 | HW_ID << 24; its first / last moment on the 100 MHz clock; its shader cycles between the two (tools/stamp_tiles.py)
constexpr int kStampPairTiles = 4;     // NEDDF_STAMP_PAIRS builds: tiles kStampTile .. + 3 of workgroups {0..3, 256..259} (a CU's two workgroups), slot 0 = HW_ID

// Colour trunk of NeDDF (neddf.py:243-300).
struct ColArgs {
    const float *pos, *dir, *var;
    int64_t n_points;
    EncodeDesc enc;
    int n_layers;                         // hidden layers (NeDDF: col_layer_count - 1, NeuS: col_layer_count)
    int activation;
    int mode;                             // 0 NeDDF inputs [embed_pos | embed_dir | normal], 1 NeuS inputs [pos | gradient | embed_dir]
    int final_act;                        // activation id applied to the 3 outputs (NeuS, neus.py:150-152) or -1
    int operands;                         // as DdfArgs::operands
    int width;                            // as DdfArgs::width
    int ksteps_a;                         // super-steps of layer 0's small-input segment [pe_pos | pe_dir | normal]
    const float *wp_a;                    // its packed weights
    LayerW layer[kMaxLayers];             // layer[0] = feature segment of layer 0
    const float *w_out;                   // [width][3] row-major (layer_col_out.weight)
    float b_out[3];
    const float *features;                // from DdfArgs
    int feat_rows;
    const float *ptaux;
    int *sched;                           // as DdfArgs::sched
    int sched_flags;
    float *color;                         // [n_points][3]
    int rays;                             // 1: pos / var / dir are read from the per-point record (PA_R_*), the pointers above are NULL
    float *penalty;                       // [n_points] (full mode) or NULL
    float distance_range_max;
    float penalty_weight[6];
    int penalty_has[6];
    unsigned long long *stamps = nullptr; // -DNEDDF_STAMP builds only: phase time stamps of the colour kernel (NEDDF_STAMP_FILE_COL, tools/stamp_timeline_col.py)
};

// Plain NeRF field (nerf.py:107-165), value rows only.
struct NerfArgs {
    const float *pos, *dir, *var;
    int64_t n_points;
    EncodeDesc enc;
    int n_layers;
    int activation, density_activation;
    LayerW layer[kMaxLayers];
    int n_stash;
    StashW stash[kMaxStash];              // [0..] skip partials, last = colour-head dir partial
    int col_stash;                        // stash index of the colour head's dir segment
    const float *w_density;               // [width]
    float b_density;
    LayerW col0;                          // outL_color.0: layer_width (+dir) -> layer_width / 2, padded to HC = a multiple of 128 columns
    const float *w_col1;                  // [3][HC] (nn.Linear layout, zero-padded)
    float b_col1[3];
    int operands;                         // as DdfArgs::operands
    int width;                            // as DdfArgs::width
    float *scratch;
    float *density, *color;
};

struct CameraArg {
    float R[9], T[3], calib[4];
};

size_t field_lds_bytes(int mt);
void launch_ddf(const DdfArgs &a, int grid, hipStream_t s);
void launch_ddf_rev(const DdfArgs &a, int grid, hipStream_t s);
size_t ddf_rev_scratch_floats_per_wg(int n_layers, int points, int width);
int ddf_rev_points(int operands, int width);        // sample points per tile of ddf_rev_kernel under an operand policy / engine width
int ddf_rev_wgs_per_cu(int operands, int width);
void launch_col(const ColArgs &a, int grid, bool rows4, hipStream_t s);
void launch_nerf(const NerfArgs &a, int grid, hipStream_t s);
int ddf_points_per_tile(int operands, int width);
int col_points_per_tile(bool rows4, int operands, int width);
int nerf_points_per_tile(int width);
int nerf_wgs_per_cu(int width);
int field_wgs_per_cu(int operands, int width);
int col_wgs_per_cu(int operands, int width);

void launch_raygen(const void *uv, int uv_type, int64_t n, const CameraArg &cam, float *dir, float *orig, hipStream_t s);
void launch_sample_coarse(const float *U, int64_t n, int S1, float near_, float far_, float *dists, hipStream_t s);
void launch_sampling(const float *rd, const float *ro, const float *view, const float *dists, int64_t n, int S, double radius,
                     float *pos, float *dir, float *var, hipStream_t s);
void launch_ndc(const float *rd, const float *ro, int64_t n, float width, float height, float fx, float fy, float near_, float *nd,
                float *no, hipStream_t s);
void launch_composite(const float *dists, const float *dens, const float *col, int64_t n, int S, float max_dist,
                      float *w, float *depth, float *color, float *trans, int *nan_flag, hipStream_t s);
void launch_integrate_penalty(const float *dists, const float *pen, int64_t n, int S, float *out, hipStream_t s);
void launch_resample(const float *dists, float *weights, const float *U, int64_t n_rays, int n, int nf, int cat,
                     float *out, int64_t *ids, int *flag, int64_t group, int64_t offset, hipStream_t s);

void launch_linear_grad(const float *x, const float *J, int64_t n, int cin, int ldx, int cout_block, int ksteps, const float *wp,
                        const float *bias, float *y, float *G, int ldo, int nvalid, int accumulate, int grid, hipStream_t s);
void launch_op_activation(int kind, const float *x, const float *J, int64_t N, int C, float *y, float *G, hipStream_t s);
void launch_op_pe(const float *x, const float *J, const float *scale, int64_t N, int E, float *y, float *G, hipStream_t s);
void launch_op_pe_weights(const float *var, int64_t N, int E, float *w, hipStream_t s);

}  // namespace neddf
