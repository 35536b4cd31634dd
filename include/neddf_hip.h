/*
 * neddf_hip.h -- C ABI of libneddf_hip.so, the MI355X (gfx950) volumetric
 * renderer behind the reference project's Python plugin surface.
 *
 * The reference (ueda0319/neddf) has no FFI: its "plugin API" is the set of
 * Python classes Hydra instantiates.  This header is
 * the boundary those classes' replacements (package neddf_amd, aliased as
 * neddf) call through ctypes.  Each entry point names the reference function
 * it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - extern "C", POD only, no torch types.  Every `d_` pointer is a DEVICE
 *     pointer owned by the caller (a torch allocation); `h_` pointers are HOST.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     stage calls are asynchronous on that stream; the caller synchronises.
 *   - Return 0 on success, a negative NEDDF_E* code otherwise; never throws.
 *     neddf_last_error(ctx) returns a static/ctx-owned message.
 *   - One ctx per device; a ctx is not thread-safe, distinct ctxs are
 *     independent.  The ctx owns packed weights and scratch workspaces.
 */
#ifndef NEDDF_HIP_H
#define NEDDF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libneddf_hip.so is built with -fvisibility=hidden: the entry points declared between this push and the pop at the end of the
 * file are its whole export list (tests/test_host.py holds `nm -D` to exactly these names). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define NEDDF_ABI_VERSION 5

enum { NEDDF_OK = 0, NEDDF_EINVAL = -1, NEDDF_EHIP = -2, NEDDF_EUNSUPPORTED = -3, NEDDF_ENOFIELD = -4,
       NEDDF_ECOMM = -5,      /* RCCL reported an error (message in neddf_last_error) or is not loadable */
       NEDDF_ETIMEOUT = -6 }; /* neddf_comm_wait_host: the collective did not finish in time; the communicator was aborted */
enum { NEDDF_FIELD_NEDDF = 0, NEDDF_FIELD_NERF = 1, NEDDF_FIELD_NEUS = 2 };
enum { NEDDF_ACT_RELU = 0, NEDDF_ACT_LEAKY = 1, NEDDF_ACT_TANHEXP = 2 };
/* operand type of the 256-wide dense layers: fp32 (exact, the parity path) or bf16 weights + bf16 activations with
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16 (BASELINE.json configs[4]); heads, biases, encodings stay fp32 */
enum { NEDDF_DTYPE_F32 = 0, NEDDF_DTYPE_BF16 = 1,
       /* fp32 weights and activations contracted on the fp16 matrix instructions: every operand split into two fp16 terms
        * (21-22 bits), the three products above 2^-22 accumulated in fp32; weights pre-scaled by 2^10 to stay in fp16's normal
        * range.  Errors at the level of the fp32 MFMA path's own, 2.5x its throughput; operands beyond +-65504 saturate */
       NEDDF_DTYPE_F16_SPLIT = 2 };
enum { NEDDF_SLOT_COARSE = 0, NEDDF_SLOT_FINE = 1, NEDDF_NUM_SLOTS = 4 };
/* uv element types accepted by neddf_raygen (the reference takes int64 in
 * render_image, int16 in training, float in its tests) */
enum { NEDDF_UV_F32 = 0, NEDDF_UV_I64 = 1, NEDDF_UV_I32 = 2, NEDDF_UV_I16 = 3 };
/* field output selection */
enum { NEDDF_OUT_MINIMAL = 0,   /* density + color (+distance, aux_grad): what compositing consumes */
       NEDDF_OUT_FULL = 1 };    /* + fields_penalty: Jacobian through the colour trunk (neddf.py:243-300) */

typedef struct neddf_ctx neddf_ctx;

/* Architecture of one field network.  Mirrors the constructor keywords of
 * NeDDF (neddf/network/neddf.py:52-66) and NeRF (neddf/network/nerf.py:34-44). */
typedef struct {
    int kind;                 /* NEDDF_FIELD_* */
    int embed_pos_rank;       /* 1..10 */
    int embed_dir_rank;       /* >= 1; both encodings + 32 columns must fit a 512-column tile row (training: <= 10) */
    int layer_count;          /* NeDDF: ddf_layer_count, NeRF: layer_count */
    int layer_width;          /* rendering: 1..512; training: 256 or 512 (other widths zero-padded by the caller) */
    int col_layer_count;      /* NeDDF only */
    int col_layer_width;      /* NeDDF (== layer_width) and NeuS */
    int n_skips;
    int skips[8];
    int activation;           /* NEDDF_ACT_* */
    int density_activation;   /* NEDDF_ACT_* */
    float d_near;             /* NeDDF only */
    /* NeDDF penalty weights in the dict order of neddf.py:260-291:
     * constraints_aux_grad, constraints_dDdt, range_distance, range_aux_grad,
     * range_color, constraints_color; has[i]==0 leaves the term unweighted. */
    float penalty_weight[6];
    int penalty_has[6];
    int weight_dtype;         /* NEDDF_DTYPE_* */
} neddf_field_desc;

/* Pinhole camera: Camera.R / Camera.T (camera.py:117-118) and
 * PinholeCalib [fx, fy, cx, cy] (pinhole_calib.py:8). */
typedef struct {
    float R[9];
    float T[3];
    float calib[4];
} neddf_camera;

/* NeRFRender constructor values (neddf/render/nerf_render.py:40-81). */
typedef struct {
    int sample_coarse;
    int sample_fine;
    float dist_near, dist_far, max_dist;
    int cone_sampling;        /* sampling_type == "cone" */
    double ray_radius;        /* 1/1111/sqrt(12), nerf_render.py:144-145 */
    /* forward-facing scenes (not in the reference, see neddf_rays_to_ndc): when ndc_rays != 0 the samples are taken
     * along normalised-device-coordinate rays (dist_near / dist_far are then NDC depths, usually 0 and 1) while the
     * field still receives the world-space unit viewing direction */
    int ndc_rays;
    int ndc_width, ndc_height;
    float ndc_near;
    /* neddf_render_rays on a batch that stands for several render_rays calls of the reference (render_image hands the
     * library many `chunk`s at once): rays [k*nan_group, (k+1)*nan_group) share sample_pdf's NaN fallback decision
     * (base_neural_render.py:105-114 decides per call) and fall back to the linspace of group k's first ray.
     * 0 = the whole batch is one call.  nan_group_offset = index, inside its group, of the batch's first ray (a ray-sharded
     * slab may begin in the middle of a chunk; the leading partial group then falls back to ITS first ray's linspace). */
    int nan_group;
    int nan_group_offset;
} neddf_render_params;

int neddf_abi_version(void);
int neddf_create(int device, neddf_ctx **out);
void neddf_destroy(neddf_ctx *ctx);
const char *neddf_last_error(neddf_ctx *ctx);
/* number of compute units of the ctx's device (for roofline reporting) */
int neddf_device_cus(neddf_ctx *ctx);
/* Bounds probe (ABI v5; no reference counterpart -- the stand-in for a GPU-side sanitizer run).  With NEDDF_GUARD=1 in the environment
 * every workspace of the context is allocated at its exact size between two poisoned 4 KiB bands and every carve of the render arena
 * is followed by a 256 B one; this call synchronises the device and reports how many bands exist and how many of their bytes a kernel
 * has overwritten (0 = no out-of-bounds store reached a band).  Without NEDDF_GUARD it reports 0 bands. */
int neddf_debug_check_guards(neddf_ctx *ctx, int64_t *n_bands, int64_t *n_bad_bytes);

/* Replaces nn.Module.load_state_dict for one network (base_trainer.py:121).
 * h_weights / h_biases: HOST fp32 arrays in state-dict order
 *   NeDDF: layers_ddf.0..n, layers_col.0..m, layer_ddf_out, layer_aux_out, layer_col_out
 *          (LinearGradLayer weights are [in,out], linear.py:113)
 *   NeRF : layers.0..n, outL_density, outL_color.0, outL_color.2
 *          (nn.Linear weights are [out,in], nerf.py:88-103)
 *   NeuS : layers_sdf.0..n-1, layers_col.0..m (m = col_layer_count; last is 256 -> 3), then `variance`
 *          as a 1-element weight with a dummy bias (nn.Linear layout, neus.py:80-99); desc.layer_count =
 *          sdf_layer_count, desc.col_layer_count = col_layer_count, activation ReLU or tanhExp
 * The library packs them into MFMA fragment order and uploads; the caller keeps
 * ownership of the sources. */
int neddf_set_field(neddf_ctx *ctx, int slot, const neddf_field_desc *desc,
                    const float *const *h_weights, const float *const *h_biases, int n_tensors);
/* Replaces NeDDF.set_iter / NeRF.set_iter (neddf.py:311-326, nerf.py:167-178):
 * h_lowpass[embed_pos_rank] = get_lowpass_scale(lowpass_alpha) per frequency. */
int neddf_set_iter(neddf_ctx *ctx, int slot, float aux_grad_scale, float distance_range_max,
                   const float *h_lowpass);

/* Camera.create_rays (camera.py:155-171) + get_center_of_pixels (:173-187) +
 * PinholeCalib.unproject_local (pinhole_calib.py:51-74). */
int neddf_raygen(neddf_ctx *ctx, const void *d_uv, int uv_type, int64_t n_rays, const neddf_camera *h_cam,
                 float *d_ray_dir, float *d_ray_orig, void *stream);
/* Stratified coarse distances (nerf_render.py:131-140): dists[b,j] =
 * linspace(near,far,S1)[j] + U[b,j]*(far-near)/(S1-1). */
int neddf_sample_coarse(neddf_ctx *ctx, const float *d_U, int64_t n_rays, int S1, float dist_near,
                        float dist_far, float *d_dists, void *stream);
/* Ray.get_sampling_cones (ray.py:128-194) when ray_radius >= 0, else
 * Ray.get_sampling_points (ray.py:88-126).  Outputs [n_rays,S,3] each. */
int neddf_sampling(neddf_ctx *ctx, const float *d_ray_dir, const float *d_ray_orig, const float *d_dists,
                   int64_t n_rays, int S, double ray_radius, float *d_pos, float *d_dir, float *d_var,
                   void *stream);
/* The same with a separate viewing direction [n_rays,3] copied to d_dir (NDC rays: positions follow the NDC ray,
 * the field sees the world-space direction). */
int neddf_sampling_view(neddf_ctx *ctx, const float *d_ray_dir, const float *d_ray_orig, const float *d_view_dir,
                        const float *d_dists, int64_t n_rays, int S, double ray_radius, float *d_pos, float *d_dir,
                        float *d_var, void *stream);
/* World-space rays -> normalised-device-coordinate rays of a width x height pinhole view with focal lengths fx, fy and
 * near plane z = -near (Mildenhall et al. 2020, appendix C; the reference has no NDC code -- parity is pinned on the
 * projective identity NDC(o + t d) = o' + t' d', t' = 1 - oz_near / (oz_near + t dz), checked in tests). */
int neddf_rays_to_ndc(neddf_ctx *ctx, const float *d_ray_dir, const float *d_ray_orig, int64_t n_rays, int width, int height,
                      float fx, float fy, float near_plane, float *d_ndc_dir, float *d_ndc_orig, void *stream);
/* NeDDF.forward (neddf.py:162-309) / NeRF.forward (nerf.py:107-165) on N
 * sample points (pos/dir/var [N,3]).  Any output pointer may be NULL.
 * NeRF fields produce density and color only; NeuS fields (neus.py:101-162) return the sdf in d_distance.
 * NeDDF without a penalty output (NEDDF_OUT_MINIMAL, or d_penalty == NULL) and NeuS: the position gradient of the distance / sdf is taken
 * in reverse mode (one gradient row per point instead of the reference's three forward-mode Jacobian rows, neddf.py:206-230);
 * with a penalty output the Jacobian rows are carried forward as in the reference.  The two agree within rounding (the
 * parity gates of tests/test_gpu_parity.py hold for both).
 * Hidden tanhExp activations (nn_module/with_grad/tanh_exp.py:15-54) are evaluated as x (1 - 2 / (e^(2 e^x) + 1)) for every x: absolute
 * error ~1e-7 |x| where the reference's tanh keeps relative accuracy; network outputs stay as close to an fp64 evaluation as the
 * reference's own fp32 ones.  The stand-alone neddf_op_activation below keeps the reference's form. */
int neddf_field_forward(neddf_ctx *ctx, int slot, const float *d_pos, const float *d_dir, const float *d_var,
                        int64_t n_points, int out_mode, float *d_distance, float *d_density, float *d_color,
                        float *d_fields_penalty, float *d_aux_grad, void *stream);
/* BaseNeuralRender.integrate_volume_render (base_neural_render.py:117-172).
 * d_weight [n_rays,S-1] may be NULL.  *d_nan_flag (int, may be NULL) is set to 1
 * if any weight is NaN (the reference asserts, :155). */
int neddf_composite(neddf_ctx *ctx, const float *d_dists, const float *d_density, const float *d_color,
                    int64_t n_rays, int S, float max_dist, float *d_weight, float *d_depth, float *d_out_color,
                    float *d_transmittance, int *d_nan_flag, void *stream);
/* penalty line integral (nerf_render.py:153-159): out[b] = sum_j (t[j+1]-t[j]) * pen[b,j] */
int neddf_integrate_penalty(neddf_ctx *ctx, const float *d_dists, const float *d_penalty, int64_t n_rays, int S,
                            float *d_out, void *stream);
/* BaseNeuralRender.sample_pdf (base_neural_render.py:27-115) with explicit
 * uniforms d_U [n_rays,n_fine].  d_dists [n_rays,n], d_weights [n_rays,n-1] is
 * sanitised IN PLACE like the reference (:52-55).  d_out [n_rays, n_fine+n]
 * (cat_coarse) or [n_rays,n_fine]; d_ids (int64, may be NULL) receives the
 * searchsorted indices.  The NaN fallback (:105-114) is applied on device, batch-wide
 * (one call of the reference = one call here). */
int neddf_importance_resample(neddf_ctx *ctx, const float *d_dists, float *d_weights, const float *d_U,
                              int64_t n_rays, int n, int n_fine, int cat_coarse, float *d_out, int64_t *d_ids,
                              void *stream);

/* ---- stand-alone layer ops (neddf/nn_module): unit-level counterparts of what the
 * fused field kernels do internally; same device code as their epilogues. ---- */
enum { NEDDF_OP_RELU = 0, NEDDF_OP_LEAKY = 1, NEDDF_OP_TANHEXP = 2, NEDDF_OP_SOFTPLUS = 3, NEDDF_OP_SIGMOID = 4 };
/* {ReLU,LeakyReLU,TanhExp,Softplus,Sigmoid}GradFunction.forward (with_grad/{relu,leaky_relu,tanh_exp,softplus,sigmoid}.py):
 * d_x [N,C], d_J [N,3,C] -> d_y [N,C], d_G [N,3,C].  With d_J == NULL: the plain
 * activation (F.relu / F.leaky_relu / tanhExp, nn_module/tanh_exp.py:15-33). */
int neddf_op_activation(neddf_ctx *ctx, int op, const float *d_x, const float *d_J, int64_t N, int C, float *d_y,
                        float *d_G, void *stream);
/* PositionalEncodingGradLayer.forward (with_grad/positional_encoding.py:34-87) when
 * d_J != NULL, PositionalEncoding.forward (positional_encoding.py:37-65) otherwise.
 * d_x [N,3], d_J [N,3,3], d_scale [N,3E] or NULL -> d_y [N,6E], d_G [N,3,6E]. */
int neddf_op_positional_encoding(neddf_ctx *ctx, const float *d_x, const float *d_J, const float *d_scale, int64_t N,
                                 int embed_dim, float *d_y, float *d_G, void *stream);
/* Sampling.get_pe_weights (sampling.py:44-71): d_var [N,3] -> d_w [N,3E] */
int neddf_op_pe_weights(neddf_ctx *ctx, const float *d_var, int64_t N, int embed_dim, float *d_w, void *stream);
/* LinearGradFunction.forward (with_grad/linear.py:15-46) on the MFMA tile engine:
 * y = xW + b, G = JW.  h_W [Cin,Cout] / h_b [Cout] are HOST arrays (packed + uploaded
 * per call: this op is a test/compat entry point, the renderer keeps weights resident).
 * Any Cin, Cout (round 4): K blocks of 256 input columns accumulate in the outputs, N blocks of 256 / 128 output columns. */
int neddf_op_linear_grad(neddf_ctx *ctx, const float *d_x, const float *d_J, const float *h_W, const float *h_b,
                         int64_t N, int Cin, int Cout, float *d_y, float *d_G, void *stream);

/* Outputs of the fused renderer; every pointer may be NULL. Shapes per ray. */
typedef struct {
    float *color;            /* [3] */
    float *depth;            /* [1] */
    float *transmittance;    /* [1] */
    float *weight;           /* [S_fine-1]   (S_fine = sample_fine+1+sample_coarse+1) */
    float *fields_penalty;   /* [1]  (forces NEDDF_OUT_FULL on NeDDF fields) */
    float *color_coarse, *depth_coarse, *transmittance_coarse;
    float *weight_coarse;    /* [sample_coarse] (sanitised, as the reference returns it) */
    float *fields_penalty_coarse;
    float *dists_coarse;     /* [sample_coarse+1] */
    float *dists_fine;       /* [S_fine] */
    int *nan_flag;           /* single int: NaN weight seen (reference asserts) */
} neddf_render_outputs;

/* NeRFRender.render_rays (nerf_render.py:109-188): raygen -> stratified ->
 * sampling -> field(coarse) -> composite -> sample_pdf -> sampling ->
 * field(fine) -> composite, all on `stream`, no host round trip.
 * d_U_coarse [n_rays, sample_coarse+1], d_U_fine [n_rays, sample_fine+1]: the
 * uniforms the reference draws with torch.rand (nerf_render.py:137,
 * base_neural_render.py:75). */
int neddf_render_rays(neddf_ctx *ctx, const void *d_uv, int uv_type, int64_t n_rays, const neddf_camera *h_cam,
                      const neddf_render_params *params, const float *d_U_coarse, const float *d_U_fine,
                      const neddf_render_outputs *out, void *stream);
/* Single-pass variant (BASELINE.json configs[1]: "128 samples/ray"): raygen ->
 * stratified S1 samples -> sampling -> field(slot) -> composite. */
int neddf_render_rays_single(neddf_ctx *ctx, int slot, const void *d_uv, int uv_type, int64_t n_rays,
                             const neddf_camera *h_cam, const neddf_render_params *params, int S1,
                             const float *d_U, const neddf_render_outputs *out, void *stream);

/* Stage timing: hipEvent pairs recorded on the launch stream around every stage kernel launched through this ctx while
 * timing is enabled (neddf_set_timing), drained by either getter (both synchronise on the recorded events).
 * neddf_get_timings: ms[0..2] = summed duration of the distance-trunk / colour-trunk / NeRF kernel launches,
 * ms[3..5] = the corresponding launch counts; n must be >= 6.
 * neddf_get_stage_timings: ms[k], launches[k] for every NEDDF_STAGE_* (n_stages >= NEDDF_STAGE_COUNT). */
enum { NEDDF_STAGE_DDF = 0, NEDDF_STAGE_COL = 1, NEDDF_STAGE_NERF = 2, NEDDF_STAGE_RAYGEN = 3, NEDDF_STAGE_NDC = 4,
       NEDDF_STAGE_SAMPLE_COARSE = 5, NEDDF_STAGE_SAMPLING = 6, NEDDF_STAGE_COMPOSITE = 7, NEDDF_STAGE_PENALTY = 8,
       NEDDF_STAGE_RESAMPLE = 9, NEDDF_STAGE_GATHER = 10, NEDDF_STAGE_COUNT = 11 };
int neddf_set_timing(neddf_ctx *ctx, int enable);
int neddf_get_timings(neddf_ctx *ctx, float *ms, int n);
int neddf_get_stage_timings(neddf_ctx *ctx, float *ms, int *launches, int n_stages);

/* ---- multi-GPU: rays shard, pixels are gathered ----------------------------
 * The reference renders on one device; its rays are independent (nerf_render.py:128-188 has no cross-ray term), so
 * the flat pixel index of a frame (or of several frames) is cut into one contiguous slab per rank and the only
 * exchange is an all-gather of the rendered pixels over RCCL/xGMI.  One process per GPU, one communicator per ctx.
 * RCCL is loaded on first use (dlopen "librccl.so.1": in a torch process that is the copy torch already mapped);
 * a library user that never calls these needs no RCCL.
 *
 * Bootstrap: rank 0 calls neddf_comm_unique_id and hands the NEDDF_COMM_ID_BYTES to every rank by any channel the
 * integrator has (file, socket, MPI, a torch.distributed store); every rank then calls neddf_comm_init -- a blocking
 * collective over the ranks. */
#define NEDDF_COMM_ID_BYTES 128
int neddf_comm_unique_id(neddf_ctx *ctx, void *h_id);
int neddf_comm_init(neddf_ctx *ctx, int rank, int nranks, const void *h_id);
/* rank / nranks of the ctx's communicator (0 / 0 without one) and the RCCL version code (e.g. 22606) */
int neddf_comm_info(neddf_ctx *ctx, int *rank, int *nranks, int *rccl_version);
int neddf_comm_destroy(neddf_ctx *ctx);
/* Slab [lo, hi) of range(n_total) that `rank` of `nranks` owns: contiguous, sizes differ by at most one. */
void neddf_shard_range(int64_t n_total, int rank, int nranks, int64_t *lo, int64_t *hi);
/* The same with slab boundaries on multiples of `granule` rows (the last granule of the range may be short): granule counts per
 * rank differ by at most one.  render_image's chunk is the granule of a sharded frame, so that no chunk -- the unit the reference
 * decides sample_pdf's NaN fallback on, base_neural_render.py:105-114 -- is split between two ranks.  granule = 1 is
 * neddf_shard_range. */
void neddf_shard_range_granular(int64_t n_total, int64_t granule, int rank, int nranks, int64_t *lo, int64_t *hi);
/* All-gather of the per-rank slabs d_local [hi-lo, channels] (fp32, neddf_shard_range order) into d_all
 * [n_total, channels] on every rank.  The collective runs on the ctx's own communication stream, ordered after the
 * work already enqueued on `stream` (the renderer's stream), and returns immediately: the caller keeps rendering the
 * next view on `stream` while the pixels travel.  d_local and d_all must stay untouched until neddf_comm_wait.
 * With equal slabs the gather lands directly in d_all; ragged slabs go through a ctx-owned padded staging buffer. */
int neddf_gather_pixels(neddf_ctx *ctx, const float *d_local, int64_t n_total, int channels, float *d_all, void *stream);
/* ... of slabs cut by neddf_shard_range_granular(n_total, granule, ...). */
int neddf_gather_pixels_granular(neddf_ctx *ctx, const float *d_local, int64_t n_total, int64_t granule, int channels, float *d_all,
                                 void *stream);
/* Make `stream` wait (device-side, no host block) for the last neddf_gather_pixels of this ctx.  NEDDF_ECOMM (once) if the
 * communicator was aborted while that gather was in flight: d_all is incomplete. */
int neddf_comm_wait(neddf_ctx *ctx, void *stream);
/* Host-side wait with a deadline: 0 when the last gather has completed; on an asynchronous RCCL error NEDDF_ECOMM; after
 * timeout_ms without completion the communicator is aborted (ncclCommAbort), the ctx returns to "no communicator"
 * (neddf_comm_info: 0 ranks; neddf_comm_init may be called again) and NEDDF_ETIMEOUT is returned -- a peer died. */
int neddf_comm_wait_host(neddf_ctx *ctx, int timeout_ms);

/* ---- training step -------------------------------------------
 * The reference trains through torch autograd over its with_grad modules
 * (the modules under neddf/nn_module/with_grad: each has a hand-written backward for the
 * (value, Jacobian) pair).  These entry points are that forward + backward for one NeDDF
 * network on N sample points.  NeRF fields (nerf.py:107-165; plain torch autograd in the reference) are
 * supported through the same calls: only density and color are produced / consumed, nn.Linear layout.
 * NeuS fields (neus.py:101-162) likewise: d_distance / d_g_distance carry the sdf, penalty and aux_grad are
 * not touched, d_var is ignored; the last tensor is `variance` (1 element, its bias slot is not read or
 * written).  The reference differentiates NeuS twice (the normal is torch.autograd.grad(create_graph=True));
 * here that is the reverse pass over the (value, Jacobian) rows of the sdf trunk, with the second derivative
 * of tanhExp that torch derives from the plain Function's backward (nn_module/tanh_exp.py:36-60).
 * The parameters are DEVICE fp32 arrays in the reference's
 * state-dict layout and order (see neddf_set_field); gradients are ACCUMULATED into d_gW / d_gB
 * (same shapes).  The slot supplies the architecture and the set_iter state; the weights it was
 * loaded with are not used here.  `d_workspace` (neddf_train_workspace_floats floats, caller-owned)
 * carries the saved activations from the forward call to the matching backward call. */
int64_t neddf_train_workspace_floats(neddf_ctx *ctx, int slot, int64_t n_points);
/* NeDDF.forward (neddf.py:162-309) in training mode: all five outputs ([N] each, color [N,3]); any may be NULL. */
int neddf_train_field_forward(neddf_ctx *ctx, int slot, const float *const *d_W, const float *const *d_B, int n_tensors,
                              const float *d_pos, const float *d_dir, const float *d_var, int64_t n_points,
                              float *d_workspace, float *d_distance, float *d_density, float *d_color,
                              float *d_fields_penalty, float *d_aux_grad, void *stream);
/* Reverse pass: upstream gradients of the five outputs (any may be NULL = zero) -> parameter gradients. */
int neddf_train_field_backward(neddf_ctx *ctx, int slot, const float *const *d_W, const float *const *d_B, int n_tensors,
                               int64_t n_points, const float *d_workspace, const float *d_g_distance,
                               const float *d_g_density, const float *d_g_color, const float *d_g_fields_penalty,
                               const float *d_g_aux_grad, float *const *d_gW, float *const *d_gB, void *stream);
/* Backward of integrate_volume_render (base_neural_render.py:148-171): gradients of weight [n_rays,S-1],
 * depth [n_rays], color [n_rays,3], transmittance [n_rays] (any may be NULL) -> d_g_density [n_rays,S],
 * d_g_point_color [n_rays,S,3]. */
int neddf_composite_backward(neddf_ctx *ctx, const float *d_dists, const float *d_density, const float *d_color,
                             int64_t n_rays, int S, float max_dist, const float *d_g_weight, const float *d_g_depth,
                             const float *d_g_color, const float *d_g_transmittance, float *d_g_density,
                             float *d_g_point_color, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NEDDF_HIP_H */
