/* dpc_hip.h -- C ABI of the MI355X (gfx950) differentiable point-cloud projector.
 *
 * The reference (eldar/differentiable-point-clouds) has NO native / FFI
 * boundary: its hot path sits behind plain Python functions built from stock
 * TensorFlow ops.  Each entry point below cites the reference function
 * (file:line under /root/reference) whose arithmetic it replaces; the Python
 * functions with the reference's own names and signatures that sit on top of
 * this ABI live in differentiable-point-clouds_amd/util/ (see INTEGRATION.md).
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous fp32 (unless noted);
 *     inputs are never written; outputs are fully overwritten
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); no sync, no
 *     allocation, no global state; safe from several threads on different
 *     streams / devices
 *   - return value: 0 = ok, <0 = invalid argument (DPC_E_*), >0 = hipError_t
 *   - grids are [B, Dz, D, D] (z = depth axis = component 0 of the transformed
 *     cloud, then y, then x; x fastest); images are [B, D, D] (the trailing
 *     channel dim of the reference layouts has size 1 and is implicit);
 *     `flip_h` applies the reference's tf.reverse along H
 *     (dpc/util/point_cloud.py:270,273): image row h <-> grid row D-1-h
 *   - nullable pointers are marked; a null `taps_*` with K*=0 means "no blur"
 */
#ifndef DPC_HIP_H
#define DPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dpc_stream_t; /* hipStream_t */

#define DPC_OK 0
#define DPC_E_NULL (-1)      /* required pointer is null          */
#define DPC_E_SHAPE (-2)     /* non-positive / unsupported sizes  */
#define DPC_E_TAPS (-3)      /* even or too large kernel size     */
#define DPC_E_WORKSPACE (-4) /* workspace too small / misaligned  */
#define DPC_E_MODE (-5)      /* unsupported parameter combination */

#define DPC_MAX_TAPS 63

#define DPC_COLLAPSE_DRC 0 /* dpc/util/drc.py:47-123            */
#define DPC_COLLAPSE_MAX 1 /* dpc/util/point_cloud.py:264-267    */

typedef struct DpcShape {
  int32_t B;          /* instances (model x view x pose candidate) */
  int32_t N;          /* points per instance                       */
  int32_t Dz;         /* grid depth  (cfg.vox_size_z or vox_size)  */
  int32_t D;          /* grid height = width (cfg.vox_size)        */
  int32_t Kx, Ky, Kz; /* odd tap counts, 0 = axis not blurred      */
} DpcShape;

typedef struct DpcParams {
  float camera_distance;      /* cfg.camera_distance      (2.0)    */
  float focal_length;         /* cfg.focal_length         (1.875)  */
  float eps;                  /* cfg.drc_logsum_clip_val  (1e-5)   */
  float max_depth;            /* cfg.max_depth            (10.0)   */
  int32_t pose_is_quaternion; /* 1: pose [B,4] (w,x,y,z); 0: [B,4,4] */
  int32_t collapse_mode;      /* DPC_COLLAPSE_*                    */
  int32_t flags;              /* reserved, 0                       */
  /* fused point dropout (dpc/util/point_cloud.py:293-319): keep exactly `dropout_keep` of the N
   * points of every instance, drawn without replacement, independently per instance, from the keyed
   * permutation of (dropout_seed, instance); 0 or >= N = keep everything.  Only the fused path
   * (dpc_saved_layout bit 1) honours it -- other shapes return DPC_E_MODE; dropped points get a zero
   * gradient. */
  int32_t dropout_keep;
  uint32_t dropout_seed;
  /* nullable DEVICE pointer to {int32 keep, int32 seed}: when set, the sort kernels read the pair from
   * there at run time instead of from the two fields above, so a step recorded into a hipGraph draws a
   * fresh subset (and follows a keep-probability schedule) on every replay -- the caller advances the
   * pair with its own enqueued work.  Same meaning of the values; same DPC_E_MODE rule. */
  const int32_t* dropout_state;
  /* fused L2 silhouette-loss epilogue (dpc/models/model_pc.py:414-415, the branch without pose candidates):
   * when l2_target (device, [B,D,D], same layout as proj) is set, the collapse kernel also writes
   * l2_grad[b,y,x] = l2_weight * (proj[b,y,x] - l2_target[b,y,x]) -- the loss gradient the backward call takes as
   * dproj -- from the registers that hold proj: no extra pass over the images.  DRC collapse only
   * (DPC_E_MODE otherwise); l2_grad must then be non-null (DPC_E_NULL). */
  const float* l2_target;
  float* l2_grad;
  float l2_weight;
  /* instance replication inside the kernels (tf_repeat_0 of dpc/models/model_pc.py:23-32 as used at :270-279: every
   * predicted cloud is projected under step_size views x num_candidates poses): when > 1, `pc` holds B / views_per_cloud
   * clouds [B / R, N, 3], instance b reads cloud b / R, and dpc_project_backward returns dpc as [B / R, N, 3] -- the sum
   * over a cloud's R instances (the gradient of the replication) -- so that the [B,N,3] copies never exist.  0 or 1:
   * one cloud per instance.  B must be a multiple of it; fused path only (DPC_E_MODE otherwise). */
  int32_t views_per_cloud;
  /* silhouette-loss epilogue with pose candidates inside the z kernels (dpc/models/model_pc.py:308-337 and :383-423):
   * forward: when sil_gt (device, [B / sil_C, sil_S, sil_S] masks, sil_S >= D; resized bilinearly the TF1 way on the
   * fly) and sil_err_parts (device, [B, dpc_sil_parts_per_view(shape)]) are set, the collapse kernel also leaves the
   * per-work-group partial sums of (gt - proj)^2 of every instance -- dpc_silhouette_select turns them into the
   * per-instance errors, the winning candidate of every (model, view) group, the weights and the loss, without reading
   * proj again.  backward: when sil_weight ([B], from dpc_silhouette_select), sil_dloss ([1], d L / d loss) and sil_proj
   * ([B,D,D], the forward's proj) are set, the collapse VJP adds sil_dloss * w_b^2 / (B / sil_C) * (proj - gt) to
   * dproj (which may then be null) itself: no dproj image is formed.  DRC collapse, fused path only. */
  const float* sil_gt;
  float* sil_err_parts;
  const float* sil_weight;
  const float* sil_dloss;
  const float* sil_proj;
  int32_t sil_C, sil_S;
} DpcParams;

const char* dpc_version(void);

/* sizeof(DpcShape) (which = 0) / sizeof(DpcParams) (which = 1) as THIS build lays them out, 0 for anything
 * else: a binding (ctypes / cgo / JNI struct mirror) compares it with its own layout at load time instead
 * of discovering a stale mirror through wrong results. */
size_t dpc_abi_struct_bytes(int which);

/* Smallest Gaussian tap count >= K for which the library holds kernels unrolled at compile time (odd counts 3 .. 21:
 * the reference's configurations use 11 and 21, default_config.yaml:55 / experiments/<name>/config.yaml; the counts in
 * between serve filters whose outer taps have decayed to nothing under the sigma schedule of
 * dpc/models/model_pc.py:33-38), or 0 when there is none -- such a filter runs through the run-time-K kernels of the
 * generic path.  A caller that knows that only the central K' taps of its K-tap filter matter may pass the slice
 * taps + (K - K') / 2 with tap count K' = dpc_compiled_taps(K'_needed): the taps are a plain array. */
int dpc_compiled_taps(int K);

/* Optional per-kernel timing for benchmarking: while enabled, every kernel /
 * memset the library enqueues is bracketed by HIP events recorded on the launch
 * stream.  dpc_profile_get synchronises on record i and returns its label
 * (static string, e.g. "zfwd") and elapsed milliseconds.  enable(…) clears
 * the records.  Process-global; not for concurrent use. */
int dpc_profile_enable(int on);
int dpc_profile_count(void);
int dpc_profile_get(int i, const char** label, float* ms);

/* Diagnostic: grid-stride copy of n floats (n % 4 == 0) with `width` (1|2|4)
 * floats per lane -- a kernel with exactly known HBM traffic (4n read, 4n
 * written) used to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters.
 * width 44 / 48: float4 with 4 / 8 loads in flight per lane, 144 / 148: the same with the
 * nontemporal policy (n a multiple of 4096 * that count) -- the variants bench.py takes its
 * on-box copy ceiling from. */
int dpc_debug_copy(dpc_stream_t stream, const float* src, float* dst, size_t n, int width);
/* Diagnostics, the one-directional companions of dpc_debug_copy (bench.py's read / write ceilings): a pure reader
 * (float4 loads, `variant` 4 | 8 in flight per lane, +100 = nontemporal; the per-work-group sums land in `partials`,
 * which must hold 8192 floats; n a multiple of 4096 * that count) and a pure writer (float4 stores of `value`;
 * variant 0, or 100 = nontemporal; n % 4 == 0). */
int dpc_debug_read(dpc_stream_t stream, const float* src, size_t n, float* partials, int variant);
int dpc_debug_fill(dpc_stream_t stream, float* dst, size_t n, float value, int variant);

/* Which clip-gradient record dpc_project_forward leaves for the backward, for
 * this shape: bit 0 (value 1) = grid_raw, the dense pre-clip scatter [B,Dz,D,D]
 * (generic path: zero-fill + global float atomics); bit 1 (value 2) = clip_mask
 * 4*B*N bytes ([B,2,N,2]: view, corner plane, sorted slot, corner row), one bit per touched trilinear corner (fused path: points are
 * bucketed by depth cell and splatted into per-plane LDS tiles, the raw grid
 * never reaches HBM); bit 2 (value 4) = point_index, int32
 * [dpc_point_index_ints(shape)] = 5*B*N + B*(Dz+2) + B*8 (+ the chunk maps below; 16-byte aligned): the points of
 * each view sorted by depth cell as 16-byte records (w, v, u, original index), the inverse
 * map (slot of point n), the bucket starts, and 8 words of plane-occupancy bits per view, which the backward
 * re-uses (set together with bit 1; clip_mask is indexed by the sorted slot); for grids whose rows are whole
 * 32-ray words (D % 32 == 0) also the chunk maps of the chunk-sparse grid layout: two copies of B*Dz*D bytes (bit c
 * of the byte of (view, plane, row): the 128-byte chunk c of that row holds anything), ordered by plane and by row;
 * bit 3 (value 8, informational) = grid_blur holds the xy-blurred grid rather
 * than G2; bit 4 (value 16, informational) = the saved and gradient grids are chunk-sparse for this shape at the
 * moment (dpc_set_chunk_sparse): only chunks within the blur's reach of a point are written / read, the rest of
 * grid_blur is left untouched.  Buffers that are not used may be null.  <0 on error. */
int dpc_saved_layout(const DpcShape* shape, const DpcParams* params);
/* Partial sums per instance that the fused silhouette-loss epilogue leaves in DpcParams.sil_err_parts
 * (= the collapse kernel's work-groups per view), 0 when the shape cannot use it. */
size_t dpc_sil_parts_per_view(const DpcShape* shape);
/* The loss side of that epilogue: inst_err[b] = fixed-order sum of the partials; then, as dpc_silhouette_loss_fwd:
 * winners [B/C] (first minimum), weight [B] = [c == winner] * valid (valid nullable; ignored for C == 1),
 * loss = sum_g valid_g^2 err[g, win] / (2 B/C). */
int dpc_silhouette_select(dpc_stream_t stream, int B, int C, int nparts, const float* err_parts, const float* valid,
                          float* inst_err, int32_t* winners, float* weight, float* loss);
/* Number of int32 elements of the point_index buffer (0 when the shape does not
 * use it). */
size_t dpc_point_index_ints(const DpcShape* shape);
/* Chunk-sparse grids (new; no reference counterpart -- the reference's grids are dense TF tensors,
 * dpc/util/point_cloud.py:60-145): the fused path stores, and later reads, only the 128-byte chunks of the saved /
 * gradient grids that lie within the blur's reach of a point on their plane; whether a shape uses it is the
 * library's choice (short filters against the grid width).  mode 0 / 1 forces it off / on for the calls that follow
 * (forward and backward of one step must see the same mode), -1 restores the rule; returns the previous mode.
 * For A/B measurements and for tests that compare the two forms bit for bit. */
int dpc_set_chunk_sparse(int mode);
/* The sparse z walk (new, round 6; no reference counterpart -- the reference's tf.cumsum / conv3d over the depth axis touch every
 * plane, dpc/util/drc.py:47-123, dpc/util/point_cloud.py:139-145): a wavefront of the fused collapse kernels skips, behind one
 * scalar branch per group of plane steps, the groups in which none of its rays has anything within the blur's reach, advancing
 * only the ray state -- in the dense walk's order with the dense walk's operations, so both walks agree bit for bit.
 * on = 1 (the default) enables it where the library's per-shape rule launches the walking instantiations (chunk-sparse grids from
 * 128-wide rows up, filters up to 11 taps), 2 wherever they are compiled (the tests of the walk on small grids), 0 makes every
 * wavefront walk every plane for the calls that follow; returns the previous setting.  For A/B measurements and for the tests
 * that compare the two walks. */
int dpc_set_sparse_walk(int on);

/* Bytes of scratch `dpc_project_forward` (direction 0) / `dpc_project_backward`
 * (direction 1) need.  256-byte aligned device memory. */
size_t dpc_workspace_bytes(const DpcShape* shape, int direction);

/* ---- fused hot path ------------------------------------------------------
 * pointcloud_project_fast, dpc/util/point_cloud.py:229-290, i.e.
 *   pc_perspective_transform (:157-216, quaternion.py:96-117)
 *   -> pointcloud2voxels3d_fast (:60-136) -> clip (:240)
 *   -> smoothen_voxels3d (:139-145) -> scale*clip (:249-253)
 *   -> drc_projection / reduce_max (drc.py:110-123 / :264-267)
 *   -> drc_depth_projection (drc.py:146-153) -> flips (:270,273).
 * Saved for backward (caller-owned): tr_pc [B,N,3], grid_raw [B,Dz,D,D]
 * (pre-clip scatter) OR clip_mask [B,N,4] bytes + point_index (see dpc_saved_layout),
 * grid_blur [B,Dz,D,D] (opaque to the caller: the post-blur, pre-scale grid G2 --
 * or, on the fused path with at most 11 z taps, the xy-blurred grid with its
 * empty planes left unwritten; the backward re-applies the z blur to it),
 * ray_sums [B,D,D,2] float64 (per ray, grid-row order: sum_{j<Dz} p_j and
 * sum_{j<=Dz} p_j psi_j of the event probabilities, needed by the backward).
 * trans/scale/focal/taps/proj_depth nullable.  proj, proj_depth are H-flipped. */
int dpc_project_forward(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                        const float* pc, const float* pose, const float* trans /*[B,3]|null*/,
                        const float* scale /*[B]|null*/, const float* focal /*[B]|null*/,
                        const float* taps_x, const float* taps_y, const float* taps_z,
                        float* tr_pc, float* grid_raw, unsigned char* clip_mask, int32_t* point_index,
                        float* grid_blur, double* ray_sums, float* proj, float* proj_depth /*null for MAX*/,
                        void* workspace, size_t workspace_bytes);

/* Backward of the above = what TF autodiff builds for
 * dpc/run/train.py:92 over the ops of point_cloud.py:229-290.
 * Upstream: dproj [B,D,D] (H-flipped layout), dproj_depth [B,D,D]|null,
 * dtr_pc_in [B,N,3]|null (gradient arriving through the `tr_pc` output).
 * Outputs: dpc [B,N,3], dpose [B,4]|[B,4,4], dtrans [B,3]|null, dscale [B]|null,
 * dfocal [B]|null (non-null only if the matching input was given). */
int dpc_project_backward(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                         const float* pc, const float* pose, const float* trans,
                         const float* scale, const float* focal,
                         const float* taps_x, const float* taps_y, const float* taps_z,
                         const float* tr_pc, const float* grid_raw, const unsigned char* clip_mask,
                         const int32_t* point_index, const float* grid_blur, const double* ray_sums,
                         const float* dproj, const float* dproj_depth, const float* dtr_pc_in,
                         float* dpc, float* dpose, float* dtrans, float* dscale, float* dfocal,
                         void* workspace, size_t workspace_bytes);

/* ---- stage-level entry points (finer-grained reference API) --------------- */

/* pc_perspective_transform, dpc/util/point_cloud.py:157-216 */
int dpc_transform_fwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                      const float* pc, const float* pose, const float* trans, const float* focal,
                      float* tr_pc);
/* scratch: >= B*16 floats (zeroed by the call) */
int dpc_transform_bwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                      const float* pc, const float* pose, const float* trans, const float* focal,
                      const float* dtr_pc, float* dpc, float* dpose, float* dtrans, float* dfocal,
                      float* scratch);

/* pointcloud2voxels3d_fast, dpc/util/point_cloud.py:60-136 (zero-fills, then
 * trilinear scatter-add; out-of-cube / NaN points dropped) and its gather VJP */
int dpc_voxelize_fwd(dpc_stream_t stream, const DpcShape* shape, const float* tr_pc, float* grid);
int dpc_voxelize_bwd(dpc_stream_t stream, const DpcShape* shape, const float* tr_pc,
                     const float* dgrid, float* dtr_pc);

/* RGB branch of pointcloud2voxels3d_fast, dpc/util/point_cloud.py:111-118: values [B,N,C]
 * (C <= 16) are spread with the trilinear weights into a CHANNEL-MAJOR grid [B,C,Dz,D,D]
 * (zero-filled first), so the scalar blur applies to it as B*C views.  bwd: dvalues [B,N,C]
 * and (nullable; omit for cfg.pc_rgb_stop_points_gradient) the gradient w.r.t. the points. */
int dpc_voxelize_values_fwd(dpc_stream_t stream, const DpcShape* shape, int channels, const float* tr_pc,
                            const float* values, float* grid);
int dpc_voxelize_values_bwd(dpc_stream_t stream, const DpcShape* shape, int channels, const float* tr_pc,
                            const float* values, const float* dgrid, float* dvalues, float* dtr_pc);

/* smoothen_voxels3d (separable), dpc/util/point_cloud.py:139-145.  order 0 =
 * x,y,z (gauss_kernel.py:27-32); order 1 = adjoint (z first).  The blur is
 * self-adjoint for odd symmetric taps.  tmp: one grid [B,Dz,D,D]; in != out. */
int dpc_blur3d(dpc_stream_t stream, const DpcShape* shape, const float* in, float* out,
               const float* taps_x, const float* taps_y, const float* taps_z, float* tmp, int order);

/* drc_projection, dpc/util/drc.py:47-123: voxels [B,Dz,D,D] -> proj [B,D,D],
 * probs [Dz+1,B,D,D] (nullable).  flip_h applies to both outputs. */
int dpc_drc_fwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                const float* voxels, float* proj, float* probs, int flip_h);
int dpc_drc_bwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params,
                const float* voxels, const float* dproj /*nullable*/, const float* dprobs /*nullable*/,
                float* dvoxels, int flip_h);

/* tf.reduce_max(voxels, [1]), dpc/util/point_cloud.py:264-267; ties share the
 * gradient equally (TF _MinOrMaxGrad). */
int dpc_max_collapse_fwd(dpc_stream_t stream, const DpcShape* shape, const float* voxels,
                         float* proj, int flip_h);
int dpc_max_collapse_bwd(dpc_stream_t stream, const DpcShape* shape, const float* voxels,
                         const float* dproj, float* dvoxels, int flip_h);

/* Silhouette loss epilogue: replaces ModelPointCloud.add_proj_loss /
 * proj_loss_pose_candidates (dpc/models/model_pc.py:383-423, :308-337) for the
 * default switches (bilinear GT downsampling, no GT blur).  Instances are
 * ordered group-major, b = g*C + c with G = B/C (model, view) groups and C pose
 * candidates.  proj [B,D,D]; gt [G,S,S] with S >= D (resized on the fly with
 * TF1's legacy bilinear sampling); valid [G] per-group weights
 * (inputs["valid_samples"], cfg.variable_num_views) or NULL, ignored when
 * C == 1 as in the reference.  Outputs: inst_err [B] = sum (gt-proj)^2,
 * winners [G] = argmin_c (first minimum; may be NULL), weight [B] =
 * [c == winner]*valid_g (saved for backward), loss [1] =
 * sum_g valid_g^2 * inst_err[g, winner] / (2 G)  (tf.nn.l2_loss / num_samples).
 * Backward: dproj [B,D,D] = dloss * weight_b^2 / G * (proj - gt). */
int dpc_silhouette_loss_fwd(dpc_stream_t stream, int B, int C, int D, int S, const float* proj,
                            const float* gt, const float* valid, float* inst_err, int32_t* winners,
                            float* weight, float* loss);
int dpc_silhouette_loss_bwd(dpc_stream_t stream, int B, int C, int D, int S, const float* proj,
                            const float* gt, const float* weight, const float* dloss, float* dproj);

/* Student pose loss: replaces the default branch of add_student_loss (dpc/models/model_pc.py:338-381;
 * quaternion_multiply / quaternion_conjugate / quaternion_normalise of dpc/util/quaternion.py:32-117 composed
 * there).  poses [n*C,4] candidate quaternions (w,x,y,z), sample-major; winners [n] int64 = the arg-min candidate
 * of every sample (dpc_silhouette_loss_fwd's output, widened); student [n,4]; weights [n] or NULL
 * (inputs["valid_samples"]).  With p = teacher (x) conj(student), a = p_w / |p|:
 * loss [1] = scale / n * sum_i w_i (1 - a_i^2), dstudent [n,4] = d loss / d student (the teacher carries no
 * gradient: tf.stop_gradient).  scale = cfg.pose_predictor_student_loss_weight. */
int dpc_student_loss(dpc_stream_t stream, int n, int C, const float* poses, const int64_t* winners,
                     const float* student, const float* weights, float scale, float* loss, float* dstudent);

/* Nearest-neighbour distance: replaces point_cloud_distance
 * (dpc/util/point_cloud_distance.py:26-39), the kernel of the Chamfer evaluation
 * (dpc/run/eval_chamfer.py:18-34, fp64 there).  vs [ns,3], vt [nt,3] in the
 * given precision (dtype_bytes 4 = float, 8 = double).  For every source point:
 * idx [ns] = argmin_j sqrt(sum (vt_j - vs)^2) (first minimum), min_dist [ns] that
 * distance, proj [ns,3] = vt[idx].  No chunking needed (the reference splits the
 * sources 10 ways only to bound its [ns,nt,3] intermediate). */
int dpc_nn_distance(dpc_stream_t stream, int dtype_bytes, int ns, int nt, const void* vs,
                    const void* vt, void* proj, void* min_dist, int32_t* idx);

/* Exact Gaussian voxeliser: replaces pointcloud2voxels
 * (dpc/util/point_cloud.py:17-57), the cfg.pc_fast:false splat of
 * pointcloud_project (:219-226).  pc [B,N,3]; lattice of G nodes per axis
 * spanning [-1,1]; output axis a takes point component perm[a]
 * (pointcloud2voxels' own meshgrid layout is perm = {1,0,2}; the layout after
 * pointcloud_project's transpose, [b,iz,iy,ix] for tr_pc = (w,v,u), is
 * {0,1,2}).  normalise: DPC_GAUSS_NORM_NONE, _SUM (cfg.pc_normalise_gauss:
 * every point's Gaussian divided by its sum over the lattice; needs inv_norm
 * [B,N,3] scratch) or _ANALYTICAL (cfg.pc_normalise_gauss_analytical, the
 * default).  raw [B,G,G,G] = summed Gaussians (saved for backward), vox =
 * clip(raw,0,1).  Backward: dpc [B,N,3] from dvox; workspace of
 * dpc_gauss_voxelize_workspace_bytes(B,G) bytes. */
#define DPC_GAUSS_NORM_NONE 0
#define DPC_GAUSS_NORM_SUM 1
#define DPC_GAUSS_NORM_ANALYTICAL 2
size_t dpc_gauss_voxelize_workspace_bytes(int B, int G);
int dpc_gauss_voxelize_fwd(dpc_stream_t stream, int B, int N, int G, const int* perm, float sigma,
                           int normalise, const float* pc, float* inv_norm, float* raw, float* vox);
int dpc_gauss_voxelize_bwd(dpc_stream_t stream, int B, int N, int G, const int* perm, float sigma,
                           int normalise, const float* pc, const float* raw, const float* dvox,
                           float* dpc, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DPC_HIP_H */
