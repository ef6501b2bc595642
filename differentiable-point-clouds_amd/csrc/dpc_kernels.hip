// dpc_kernels.hip -- main translation unit of the MI355X (gfx950 / CDNA4, wave64) differentiable point-cloud
// projector: the C ABI (include/dpc_hip.h), the launch logic (host_launch.inc) and every kernel WITHOUT a Gaussian
// tap-count parameter.  The kernels unrolled for a compile-time tap count K (3 .. 21 odd) are compiled by tu_taps.hip,
// one translation unit per K, and reached through dpck::TapKernels<K> (host_launch_k.inc).
//
// Fused hot path (power-of-two D in [32,256], Kx == Ky in {3..21 odd}, Dz <= 256); grids are [B,Dz,D,D], x fastest;
// V = bytes of one grid of one view.  Planes that hold no trilinear mass (one bit per plane, from the depth-cell
// histogram) are neither written nor read anywhere on this path:
//
//   forward   k_zsort       WG/view (N <= 8192; k_zhist + k_zscatter, several WGs per view, beyond): camera transform
//                           (quaternion or matrix) -> tr_pc, LDS counting sort of the points by depth cell into
//                           16-byte records (w, v, u, n), inverse map, plane-occupancy bits; the fused dropout's keyed
//                           permutation decides which points survive; in-kernel replication over views
//             k_splat_xy    WG/(view, plane, y-strip): zero an LDS tile, FIXED-POINT ds_add_u32 splat of the plane's
//                           points (order-independent: bitwise reproducible), one clip-gradient bit per touched
//                           corner, clip, x-blur (a grid row = one 16-lane DPP row, halo through row_shr / row_shl with
//                           bound_ctrl), y-blur from a register window -> xy-blurred plane.          writes 1 V
//             k_zfwd        thread = 2 adjacent rays, streams z once: packed register z-FIR, scale / clip, DRC collapse
//                           as a running transmittance product (no log / exp), silhouette + depth + fp64 ray sums,
//                           optionally the L2 loss gradient or the per-work-group partials of the candidate loss.  Up
//                           to 19 z taps it only READS: the xy-blurred grid itself is what is saved for backward;
//                           at 21 it also writes G2.                                            reads 1 V (+ 1 V)
//   backward  k_zbwd        same walk: (re-applies the z-FIR to the saved xy grid, bit for bit,) DRC VJP (suffix sum =
//                           saved total - fp64 prefix), scale / clip masks, per-WG dscale partials, z-FIR adjoint
//                           (taps reversed); forms the candidate loss's gradient itself.        reads 1 V, writes 1 V
//             k_gather_yx   WG/(view, plane, y-strip): rows -> LDS, y-blur adjoint in place, x-blur adjoint (dense in
//                           registers while staging for one-strip planes with many points, per-point windows otherwise)
//                           + clip bits + trilinear gather -> per-(corner plane[, row]) partial d(tr_pc).  reads ~1 V
//             k_points_bwd_sorted / k_points_bwd_slots   thread = point (caller's order; partials fetched through
//                           slot_of) or thread = slot (multi-strip grids: records and partials stream, the gradient
//                           store scatters): camera-transform VJP, block reduction of dq / dt / df into a [B,16]
//                           accumulator (cleared by k_zbwd's first work-group per view) with returning atomics; the
//                           LAST work-group of a view finishes it (quaternion Jacobian, fixed-order dscale sum)
//             k_sum_views   (views_per_cloud > 1) point gradient of a cloud = fixed-order sum over its instances
//
// Outside the headline path, same ABI: k_sil_* (silhouette loss epilogue), k_student_loss, k_gv_* (exact Gaussian
// voxeliser, the reference's pc_fast:false splat), k_nn_distance (nearest neighbour / Chamfer), k_scatter_vals /
// k_gather_vals (RGB channels), k_copy* / k_read_sum / k_fill (bench.py's HBM ceilings, PMC calibration).
//
// Generic path (any D, odd K <= 63, max-collapse, no blur, stage-level API):
//   k_points_fwd (transform + 8 global_atomic_add_f32 into zero-filled G0), k_blur_xy_stream / k_blur_plane (plane
//   blur, LDS-free / LDS-tiled), k_blur_z / k_blur_z_generic, k_scatter, k_gather, k_points_bwd + k_pose_finalize,
//   k_max_fwd / k_max_bwd.
//
// No MFMA: this is scatter / stencil / scan work bounded by HBM (and, before the rewrites recorded in profiles/, by
// VALU issue and LDS crossbar time).
//
// The same sources compile for the CPU-only test tier with -DDPC_EMU (tests/hipemu/hip_emu.h); that build is never
// loaded by the product.

#include "k_prelude.inc"

#include <atomic>
#include <mutex>
#include <vector>

// per-kernel timing (declared in k_prelude.inc; the one definition of the whole library)
namespace dpcprof {
struct Rec {
  const char* label;
  hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;
bool begin(const char* label, hipStream_t st) {
  if (!g_on) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.label = label;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return true;
}
void end(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs.back().b, st);
}
static inline void clear() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
}
}  // namespace dpcprof

// ---------------------------------------------------------------------------
// the kernels and their launchers (order matters: later files use earlier ones).  The kernels that are unrolled for a
// compile-time tap count are only DECLARED to this unit (host_launch_k.inc); tu_taps.hip compiles them, one unit per K.
// ---------------------------------------------------------------------------
#include "k_device_common.inc"
#include "k_points.inc"
#include "k_blur_lds.inc"
#include "k_fir.inc"
#include "k_blur_stream.inc"
#include "k_fused.inc"
#include "k_zpass.inc"
#include "host_launch_k.inc"
#include "host_launch.inc"
#include "k_extras.inc"

extern "C" {

const char* dpc_version(void) { return "dpc_hip 0.2.0 (gfx950)"; }

size_t dpc_abi_struct_bytes(int which) { return which == 0 ? sizeof(DpcShape) : which == 1 ? sizeof(DpcParams) : 0; }

int dpc_compiled_taps(int K) {
  for (int k = K < 3 ? 3 : (K | 1); k <= DPC_MAX_TAPS; k += 2)
    if (tap_compiled(k)) return k;
  return 0;
}

int dpc_set_chunk_sparse(int mode) { return chunk_sparse_mode().exchange(mode < 0 ? -1 : (mode ? 1 : 0)); }

int dpc_set_sparse_walk(int on) { return sparse_walk_mode().exchange(on == 2 ? 2 : (on ? 1 : 0)); }
#ifdef DPC_EMU
// (CPU test tier only) dead groups the emulated wavefronts of the z kernels took since the last call
long long dpc_emu_dead_groups_take(void) { return dpc_emu_dead_groups().exchange(0); }
// ... wavefronts of the 1024-thread z kernels that were dealt another tile of their work-group (zdeal_tiles) since the last call
long long dpc_emu_deals_take(void) { return dpc_emu_deals().exchange(0); }
#endif

int dpc_profile_enable(int on) {
  dpcprof::clear();
  std::lock_guard<std::mutex> lk(dpcprof::g_mu);
  dpcprof::g_on = on != 0;
  return DPC_OK;
}

int dpc_saved_layout(const DpcShape* shape, const DpcParams* params) {
  if (check_shape(shape, true) != DPC_OK || !params) return DPC_E_SHAPE;
  if (!splat_plan(*shape).ok) return 1;   // bit 0: grid_raw
  // bits 1+2: clip_mask + point_index; bit 3 (informational): grid_blur holds the xy-blurred grid, not G2;
  // bit 4 (informational): the grids are chunk-sparse for this shape right now (the rule, or dpc_set_chunk_sparse)
  return 6 | (save_xy_mode(*shape, params->collapse_mode == DPC_COLLAPSE_DRC) ? 8 : 0) | (chunk_sparse_marks(*shape, params->collapse_mode == DPC_COLLAPSE_DRC) ? 16 : 0);
}

size_t dpc_sil_parts_per_view(const DpcShape* shape) {
  // (the epilogue lives in the fused collapse kernels: fused front end, a compile-time z tap count, whole work-groups)
  if (check_shape(shape, true) != DPC_OK || !splat_plan(*shape).ok || !z_fixed(shape->Kz)) return 0;
  return ((size_t)shape->D * shape->D / pick_cx(shape->D)) % DPC_BLOCK == 0 ? (size_t)zbwd_blocks(*shape) : 0;
}

int dpc_silhouette_select(dpc_stream_t stream, int B, int C, int nparts, const float* err_parts, const float* valid,
                          float* inst_err, int32_t* winners, float* weight, float* loss) {
  if (B <= 0 || C <= 0 || B % C != 0 || nparts <= 0) return DPC_E_SHAPE;
  if (!err_parts || !inst_err || !weight || !loss) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  DPC_LAUNCH("sil_select", (k_sil_select_parts), dim3(1, 1, 1), dim3(DPC_BLOCK, 1, 1), 0, st, err_parts, nparts, valid,
             inst_err, (int*)winners, weight, loss, B / C, C);
  return last_error();
}

size_t dpc_point_index_ints(const DpcShape* shape) {
  if (check_shape(shape, true) != DPC_OK || !splat_plan(*shape).ok) return 0;
  return point_index_ints(*shape);
}

#define DPC_DEBUG_READ_BLOCKS (256 * 32)
int dpc_debug_copy(dpc_stream_t stream, const float* src, float* dst, size_t n, int width) {
  if (!src || !dst) return DPC_E_NULL;
  if (n == 0 || (n % 4) != 0) return DPC_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(256 * 8, 1, 1), block(DPC_BLOCK, 1, 1);
  if (width == 4)
    DPC_LAUNCH("copy4", (k_copy<4>), grid, block, 0, st, src, dst, n);
  else if (width == 2)
    DPC_LAUNCH("copy2", (k_copy<2>), grid, block, 0, st, src, dst, n);
  else if (width == 1)
    DPC_LAUNCH("copy1", (k_copy<1>), grid, block, 0, st, src, dst, n);
  else if (width == 44 || width == 48 || width == 144 || width == 148) {
    // ceiling variants: float4, 4 or 8 loads in flight per lane; 1xx = nontemporal loads and stores.  n must be a
    // multiple of the grid's footprint (4 * U * threads floats) -- bench.py's buffers are.
    const int U = (width % 100) - 40;
    const size_t per = (size_t)4 * U * DPC_BLOCK;
    if (n % per != 0) return DPC_E_SHAPE;
    size_t blocks = n / per;
    if (blocks > 256 * 32) blocks = 256 * 32;
    while ((n / per) % blocks != 0) --blocks;
    const dim3 g((unsigned)blocks, 1, 1);
    if (width == 44) DPC_LAUNCH("copy4x4", (k_copy_unrolled<4, false>), g, block, 0, st, src, dst, n);
    else if (width == 48) DPC_LAUNCH("copy4x8", (k_copy_unrolled<8, false>), g, block, 0, st, src, dst, n);
    else if (width == 144) DPC_LAUNCH("copy4x4nt", (k_copy_unrolled<4, true>), g, block, 0, st, src, dst, n);
    else DPC_LAUNCH("copy4x8nt", (k_copy_unrolled<8, true>), g, block, 0, st, src, dst, n);
  }
  else
    return DPC_E_MODE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DPC_OK : (int)e;
}

int dpc_debug_read(dpc_stream_t stream, const float* src, size_t n, float* partials, int variant) {
  if (!src || !partials) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int U = (variant % 100) >= 8 ? 8 : 4;
  const size_t per = (size_t)4 * U * DPC_BLOCK;
  if (n == 0 || n % per != 0) return DPC_E_SHAPE;
  size_t blocks = n / per;
  if (blocks > DPC_DEBUG_READ_BLOCKS) blocks = DPC_DEBUG_READ_BLOCKS;
  while ((n / per) % blocks != 0) --blocks;
  const dim3 g((unsigned)blocks, 1, 1), block(DPC_BLOCK, 1, 1);
  if (variant == 4) DPC_LAUNCH("read4x4", (k_read_sum<4, false>), g, block, 0, st, src, partials, n);
  else if (variant == 8) DPC_LAUNCH("read4x8", (k_read_sum<8, false>), g, block, 0, st, src, partials, n);
  else if (variant == 104) DPC_LAUNCH("read4x4nt", (k_read_sum<4, true>), g, block, 0, st, src, partials, n);
  else if (variant == 108) DPC_LAUNCH("read4x8nt", (k_read_sum<8, true>), g, block, 0, st, src, partials, n);
  else return DPC_E_MODE;
  return last_error();
}

int dpc_debug_fill(dpc_stream_t stream, float* dst, size_t n, float value, int variant) {
  if (!dst) return DPC_E_NULL;
  if (n == 0 || (n % 4) != 0) return DPC_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(256 * 16, 1, 1), block(DPC_BLOCK, 1, 1);
  if (variant == 0) DPC_LAUNCH("fill4", (k_fill<false>), g, block, 0, st, dst, n, value);
  else if (variant == 100) DPC_LAUNCH("fill4nt", (k_fill<true>), g, block, 0, st, dst, n, value);
  else return DPC_E_MODE;
  return last_error();
}

int dpc_profile_count(void) {
  std::lock_guard<std::mutex> lk(dpcprof::g_mu);
  return (int)dpcprof::g_recs.size();
}

int dpc_profile_get(int i, const char** label, float* ms) {
  std::lock_guard<std::mutex> lk(dpcprof::g_mu);
  if (i < 0 || i >= (int)dpcprof::g_recs.size() || !label || !ms) return DPC_E_NULL;
  *label = dpcprof::g_recs[i].label;
  *ms = 0.f;
  hipError_t e = hipEventSynchronize(dpcprof::g_recs[i].b);
  if (e != hipSuccess) return (int)e;
  e = hipEventElapsedTime(ms, dpcprof::g_recs[i].a, dpcprof::g_recs[i].b);
  if (e != hipSuccess) return (int)e;
  return DPC_OK;
}

size_t dpc_workspace_bytes(const DpcShape* shape, int direction) {
  if (check_shape(shape, false) != DPC_OK) return 0;
  const size_t g = align256(grid_elems(*shape) * sizeof(float));
  const size_t acc = align256(sizeof(float) * 16 * (size_t)shape->B);
  return direction == 0 ? g : 2 * g + acc + parts_bytes(*shape) + dsparts_bytes(*shape) + dpc_views_bytes(*shape);
}

int dpc_transform_fwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* pc,
                      const float* pose, const float* trans, const float* focal, float* tr_pc) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!params || !pc || !pose || !tr_pc) return DPC_E_NULL;
  if (!params->pose_is_quaternion && trans) return DPC_E_MODE;  // point_cloud.py:211-213
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = point_grid(*shape), block(DPC_BLOCK, 1, 1);
  if (params->pose_is_quaternion)
    DPC_LAUNCH("points_fwd", (k_points_fwd<true>), grid, block, 0, st, *shape, *params, pc, pose, trans, focal, tr_pc,
               (float*)nullptr);
  else
    DPC_LAUNCH("points_fwd", (k_points_fwd<false>), grid, block, 0, st, *shape, *params, pc, pose, trans, focal, tr_pc,
               (float*)nullptr);
  return last_error();
}

int dpc_transform_bwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* pc,
                      const float* pose, const float* trans, const float* focal, const float* dtr_pc,
                      float* dpc, float* dpose, float* dtrans, float* dfocal, float* scratch) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!params || !pc || !pose || !dtr_pc || !dpc || !dpose || !scratch) return DPC_E_NULL;
  if (!params->pose_is_quaternion && trans) return DPC_E_MODE;
  return launch_points_bwd((hipStream_t)stream, *shape, *params, pc, pose, trans, focal, nullptr, nullptr,
                           nullptr, nullptr, dtr_pc, false, dpc, dpose, dtrans, dfocal, nullptr,
                           scratch, true);
}

int dpc_voxelize_fwd(dpc_stream_t stream, const DpcShape* shape, const float* tr_pc, float* grid) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!tr_pc || !grid) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = dpc_memset("memset_grid", grid, grid_elems(*shape) * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  DPC_LAUNCH("scatter", (k_scatter), point_grid(*shape), dim3(DPC_BLOCK, 1, 1), 0, st, *shape, tr_pc, grid);
  return last_error();
}

int dpc_voxelize_bwd(dpc_stream_t stream, const DpcShape* shape, const float* tr_pc, const float* dgrid,
                     float* dtr_pc) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!tr_pc || !dgrid || !dtr_pc) return DPC_E_NULL;
  DPC_LAUNCH("gather", (k_gather), point_grid(*shape), dim3(DPC_BLOCK, 1, 1), 0, (hipStream_t)stream, *shape, tr_pc,
             dgrid, dtr_pc);
  return last_error();
}

int dpc_voxelize_values_fwd(dpc_stream_t stream, const DpcShape* shape, int channels, const float* tr_pc,
                            const float* values, float* grid) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!tr_pc || !values || !grid) return DPC_E_NULL;
  if (channels <= 0 || channels > 16) return DPC_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = dpc_memset("memset_grid", grid, grid_elems(*shape) * sizeof(float) * channels, st);
  if (e != hipSuccess) return (int)e;
  DPC_LAUNCH("scatter_vals", (k_scatter_vals), point_grid(*shape), dim3(DPC_BLOCK, 1, 1), 0, st, *shape, channels,
             tr_pc, values, grid);
  return last_error();
}

int dpc_voxelize_values_bwd(dpc_stream_t stream, const DpcShape* shape, int channels, const float* tr_pc,
                            const float* values, const float* dgrid, float* dvalues, float* dtr_pc) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!tr_pc || !values || !dgrid || !dvalues) return DPC_E_NULL;
  if (channels <= 0 || channels > 16) return DPC_E_SHAPE;
  DPC_LAUNCH("gather_vals", (k_gather_vals), point_grid(*shape), dim3(DPC_BLOCK, 1, 1), 0, (hipStream_t)stream,
             *shape, channels, tr_pc, values, dgrid, dvalues, dtr_pc);
  return last_error();
}

int dpc_blur3d(dpc_stream_t stream, const DpcShape* shape, const float* in, float* out, const float* taps_x,
               const float* taps_y, const float* taps_z, float* tmp, int order) {
  int rc = check_shape(shape, false);
  if (rc) return rc;
  if (!in || !out || in == out) return DPC_E_NULL;
  const DpcShape& S = *shape;
  if ((S.Kx > 0 && !taps_x) || (S.Ky > 0 && !taps_y) || (S.Kz > 0 && !taps_z)) return DPC_E_NULL;
  const bool plane = S.Kx > 0 || S.Ky > 0, zed = S.Kz > 0;
  if (!plane && !zed) return DPC_E_TAPS;
  if (plane && zed && !tmp) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  // order 1 = the adjoint of order 0: passes in reverse order, every tap vector reversed
  const int rev = order != 0;
  if (plane && zed) {
    if (order == 0) {
      rc = launch_blur_plane(st, S, in, tmp, taps_x, taps_y, S.Kx, S.Ky, 0);
      if (rc) return rc;
      return launch_blur_z(st, S, tmp, out, taps_z, S.Kz);
    }
    rc = launch_blur_z(st, S, in, tmp, taps_z, S.Kz, 1);
    if (rc) return rc;
    return launch_blur_plane(st, S, tmp, out, taps_x, taps_y, S.Kx, S.Ky, 0, 1);
  }
  if (plane) return launch_blur_plane(st, S, in, out, taps_x, taps_y, S.Kx, S.Ky, 0, rev);
  return launch_blur_z(st, S, in, out, taps_z, S.Kz, rev);
}

int dpc_drc_fwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* voxels,
                float* proj, float* probs, int flip_h) {
  int rc = check_shape(shape, false);
  if (rc) return rc;
  if (!params || !voxels || !proj) return DPC_E_NULL;
  return launch_zfwd((hipStream_t)stream, *shape, *params, voxels, nullptr, 0, nullptr, nullptr, probs, proj,
                     nullptr, nullptr, 0, flip_h);
}

int dpc_drc_bwd(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* voxels,
                const float* dproj, const float* dprobs, float* dvoxels, int flip_h) {
  int rc = check_shape(shape, false);
  if (rc) return rc;
  if (!params || !voxels || !dvoxels || (!dproj && !dprobs)) return DPC_E_NULL;
  return launch_zbwd((hipStream_t)stream, *shape, *params, voxels, nullptr, 0, nullptr, nullptr, dproj,
                     nullptr, dprobs, dvoxels, nullptr, flip_h);
}

int dpc_max_collapse_fwd(dpc_stream_t stream, const DpcShape* shape, const float* voxels, float* proj,
                         int flip_h) {
  int rc = check_shape(shape, false);
  if (rc) return rc;
  if (!voxels || !proj) return DPC_E_NULL;
  DPC_LAUNCH("max_fwd", (k_max_fwd), col_grid(*shape, 1), dim3(DPC_BLOCK, 1, 1), 0, (hipStream_t)stream, voxels,
             (const float*)nullptr, proj, shape->Dz, shape->D, flip_h);
  return last_error();
}

int dpc_max_collapse_bwd(dpc_stream_t stream, const DpcShape* shape, const float* voxels, const float* dproj,
                         float* dvoxels, int flip_h) {
  int rc = check_shape(shape, false);
  if (rc) return rc;
  if (!voxels || !dproj || !dvoxels) return DPC_E_NULL;
  DPC_LAUNCH("max_bwd", (k_max_bwd), col_grid(*shape, 1), dim3(DPC_BLOCK, 1, 1), 0, (hipStream_t)stream, voxels,
             (const float*)nullptr, dproj, dvoxels, (float*)nullptr, shape->Dz, shape->D, flip_h);
  return last_error();
}

int dpc_project_forward(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* pc,
                        const float* pose, const float* trans, const float* scale, const float* focal,
                        const float* taps_x, const float* taps_y, const float* taps_z, float* tr_pc,
                        float* grid_raw, unsigned char* clip_mask, int32_t* point_index, float* grid_blur,
                        double* ray_sums, float* proj, float* proj_depth, void* workspace,
                        size_t workspace_bytes) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!params || !pc || !pose || !tr_pc || !grid_blur || !proj) return DPC_E_NULL;
  const DpcShape& S = *shape;
  const DpcParams& P = *params;
  if ((S.Kx > 0 && !taps_x) || (S.Ky > 0 && !taps_y) || (S.Kz > 0 && !taps_z)) return DPC_E_NULL;
  if (!P.pose_is_quaternion && trans) return DPC_E_MODE;
  const bool drc = P.collapse_mode == DPC_COLLAPSE_DRC;
  if (!drc && P.collapse_mode != DPC_COLLAPSE_MAX) return DPC_E_MODE;
  if (drc && !ray_sums) return DPC_E_NULL;
  if (P.l2_target && !drc) return DPC_E_MODE;   // the L2 epilogue lives in the DRC collapse kernel
  if (P.l2_target && !P.l2_grad) return DPC_E_NULL;
  if (P.sil_gt && P.sil_err_parts) {        // fused candidate-loss epilogue: DRC collapse of the fused path, whole work-groups
    if (!drc || dpc_sil_parts_per_view(shape) == 0 || !z_fixed(S.Kz)) return DPC_E_MODE;
    if (P.sil_C <= 0 || S.B % P.sil_C != 0 || P.sil_S < S.D) return DPC_E_SHAPE;
  }
  const SplatPlan plan = splat_plan(S);
  if (P.views_per_cloud > 1 && (!plan.ok || S.B % P.views_per_cloud != 0)) return DPC_E_MODE;   // replication lives in the fused path's kernels
  if (plan.ok ? (!clip_mask || !point_index) : !grid_raw) return DPC_E_NULL;
  if (plan.ok && ((uintptr_t)point_index & 15) != 0) return DPC_E_WORKSPACE;   // 16-byte point records
  // the fused dropout lives in the depth sort of the fused path: refuse rather than silently keep every point
  if (!plan.ok && (P.dropout_state || (P.dropout_keep > 0 && P.dropout_keep < S.N))) return DPC_E_MODE;
  const bool plane = S.Kx > 0 || S.Ky > 0;
  if (plane && (!workspace || workspace_bytes < dpc_workspace_bytes(shape, 0) ||
                ((uintptr_t)workspace & 255) != 0))
    return DPC_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* tmp = (float*)workspace;
  const dim3 pgrid = point_grid(S), pblock(DPC_BLOCK, 1, 1);

  const float* zin;
  int clip_in;
  unsigned* live = nullptr;
  bool save_xy = false;
  if (plan.ok) {
    // 1+2 fused front end: transform + z-bucket (one WG per view) -> per-plane LDS splat + clip + x,y blur
    const PointIndex pi = point_index_views(S, point_index);
    live = pi.live;
    // With the fused z pass as consumer, planes without mass are skipped end to end, and the xy-blurred
    // grid is what gets SAVED (in the grid_blur buffer): k_zfwd then only reads -- storing G2 was its
    // bottleneck -- and k_zbwd re-applies the z-FIR to the saved planes (DPC_SAVE_XY).
    save_xy = save_xy_mode(S, drc);
    float* xy_out = save_xy ? grid_blur : tmp;
    // (the depth sort's per-work-group histograms live at the head of the workspace until the splat runs)
    rc = launch_splat_xy(st, S, P, plan, pc, pose, trans, focal, tr_pc, pi, taps_x, taps_y, xy_out, clip_mask,
                         drc && z_fixed(S.Kz), workspace, workspace_bytes);
    if (rc) return rc;
    zin = xy_out;
    clip_in = 0;
  } else {
    // 1. zero G0, transform + scatter (global float atomics)
    hipError_t e = dpc_memset("memset_grid", grid_raw, grid_elems(S) * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (P.pose_is_quaternion)
      DPC_LAUNCH("points_fwd", (k_points_fwd<true>), pgrid, pblock, 0, st, S, P, pc, pose, trans, focal, tr_pc, grid_raw);
    else
      DPC_LAUNCH("points_fwd", (k_points_fwd<false>), pgrid, pblock, 0, st, S, P, pc, pose, trans, focal, tr_pc, grid_raw);
    rc = last_error();
    if (rc) return rc;
    // 2. clip + x,y blur
    zin = grid_raw;
    clip_in = 1;
    if (plane) {
      rc = launch_blur_plane(st, S, grid_raw, tmp, taps_x, taps_y, S.Kx, S.Ky, 1);
      if (rc) return rc;
      zin = tmp;
      clip_in = 0;
    }
  }
  // 3. z blur fused with the ray collapse
  if (drc && z_fixed(S.Kz))
    return launch_zfwd(st, S, P, zin, taps_z, S.Kz, scale, save_xy ? nullptr : grid_blur, nullptr, proj, proj_depth,
                       ray_sums, clip_in, 1, live);
  // generic tap count or max-collapse: materialise G2, then collapse separately
  if (S.Kz > 0) {
    if (clip_in) return DPC_E_MODE;  // z-only blur of the raw grid is not a reference configuration
    rc = launch_blur_z(st, S, zin, grid_blur, taps_z, S.Kz);
    if (rc) return rc;
    zin = grid_blur;
  }
  if (drc) {
    // G2 already in grid_blur (or still the unblurred grid): collapse with a K=1 pass
    return launch_zfwd(st, S, P, zin, nullptr, 0, scale, (zin == grid_blur ? nullptr : grid_blur), nullptr,
                       proj, proj_depth, ray_sums, clip_in, 1);
  }
  if (zin != grid_blur) {  // no z blur: G2 = (clipped) input; copy through the K=1 FIR kernel
    rc = launch_zfwd(st, S, P, zin, nullptr, 0, nullptr, grid_blur, nullptr, nullptr, nullptr, nullptr,
                     clip_in, 1);
    if (rc) return rc;
  }
  DPC_LAUNCH("max_fwd", (k_max_fwd), col_grid(S, 1), dim3(DPC_BLOCK, 1, 1), 0, st, (const float*)grid_blur, scale, proj,
             S.Dz, S.D, 1);
  return last_error();
}

int dpc_project_backward(dpc_stream_t stream, const DpcShape* shape, const DpcParams* params, const float* pc,
                         const float* pose, const float* trans, const float* scale, const float* focal,
                         const float* taps_x, const float* taps_y, const float* taps_z, const float* tr_pc,
                         const float* grid_raw, const unsigned char* clip_mask, const int32_t* point_index,
                         const float* grid_blur, const double* ray_sums, const float* dproj,
                         const float* dproj_depth,
                         const float* dtr_pc_in, float* dpc, float* dpose, float* dtrans, float* dscale,
                         float* dfocal, void* workspace, size_t workspace_bytes) {
  int rc = check_shape(shape, true);
  if (rc) return rc;
  if (!params || !pc || !pose || !tr_pc || !grid_blur || !dpc || !dpose) return DPC_E_NULL;
  const DpcShape& S = *shape;
  const DpcParams& P = *params;
  if ((S.Kx > 0 && !taps_x) || (S.Ky > 0 && !taps_y) || (S.Kz > 0 && !taps_z)) return DPC_E_NULL;
  if (!P.pose_is_quaternion && trans) return DPC_E_MODE;
  const bool drc = P.collapse_mode == DPC_COLLAPSE_DRC;
  if (!drc && P.collapse_mode != DPC_COLLAPSE_MAX) return DPC_E_MODE;
  if (P.sil_weight) {
    if (!drc || !splat_plan(S).ok || !z_fixed(S.Kz)) return DPC_E_MODE;
    if (!P.sil_gt || !P.sil_dloss || !P.sil_proj) return DPC_E_NULL;
    if (P.sil_C <= 0 || S.B % P.sil_C != 0 || P.sil_S < S.D) return DPC_E_SHAPE;
  }
  if (!dproj && !(drc && dproj_depth) && !P.sil_weight) return DPC_E_NULL;
  if (scale && !dscale) return DPC_E_NULL;
  const SplatPlan plan = splat_plan(S);
  if (P.views_per_cloud > 1 && (!plan.ok || S.B % P.views_per_cloud != 0)) return DPC_E_MODE;
  const bool use_cmask = plan.ok;
  if (use_cmask ? (!clip_mask || !point_index) : !grid_raw) return DPC_E_NULL;
  if (use_cmask && ((uintptr_t)point_index & 15) != 0) return DPC_E_WORKSPACE;
  if (!workspace || workspace_bytes < dpc_workspace_bytes(shape, 1) || ((uintptr_t)workspace & 255) != 0)
    return DPC_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t gbytes = align256(grid_elems(S) * sizeof(float));
  float* tA = (float*)workspace;
  float* tB = (float*)((char*)workspace + gbytes);
  float* accum = (float*)((char*)workspace + 2 * gbytes);  // [B,16]: pose/trans/focal sums, slot 15 = dscale
  float* parts = (float*)((char*)accum + align256(sizeof(float) * 16 * (size_t)S.B));  // [B,N,4,3]

  float* dsparts = (float*)((char*)parts + parts_bytes(S));   // [B, zbwd work-groups]: dscale partials
  const bool zfused = drc && z_fixed(S.Kz);
  if (!zfused) {  // (the fused z kernel clears the accumulator itself and returns dscale as partials)
    hipError_t e = dpc_memset("memset_small", accum, sizeof(float) * 16 * (size_t)S.B, st);
    if (e != hipSuccess) return (int)e;
  }
  float* ds_acc = scale ? accum : nullptr;
  int nzb = zbwd_blocks(S);      // (the fused z kernel reports the work-groups per view it really used)
  // 1. collapse VJP (+ z-FIR adjoint) -> tA
  const bool yx = use_cmask;   // consumer of tA is k_gather_yx (reads occupied planes only)
  PointIndex pi = {nullptr, nullptr, nullptr, nullptr};
  if (use_cmask) pi = point_index_views(S, point_index);
  if (zfused) {
    const unsigned* live = yx ? pi.live : nullptr;
    // fused forward (plan.ok): grid_blur holds the xy-blurred grid, see dpc_project_forward
    rc = launch_zbwd(st, S, P, grid_blur, taps_z, S.Kz, scale, ray_sums, dproj, dproj_depth, nullptr, tA,
                     ds_acc, 1, live, scale ? dsparts : nullptr, accum, pi.live, use_cmask && save_xy_mode(S, drc), &nzb);
    if (rc) return rc;
  } else {
    float* first = (S.Kz > 0) ? tB : tA;
    if (drc) {
      rc = launch_zbwd(st, S, P, grid_blur, nullptr, 0, scale, ray_sums, dproj, dproj_depth, nullptr, first,
                       ds_acc, 1);
    } else {
      DPC_LAUNCH("max_bwd", (k_max_bwd), col_grid(S, 1), dim3(DPC_BLOCK, 1, 1), 0, st, grid_blur, scale, dproj, first,
                 ds_acc, S.Dz, S.D, 1);
      rc = last_error();
    }
    if (rc) return rc;
    if (S.Kz > 0) {
      rc = launch_blur_z(st, S, tB, tA, taps_z, S.Kz, 1);
      if (rc) return rc;
    }
  }
  if (yx) {
    // 2+3 fused: per-plane LDS pass (y-blur + sparse x-blur + clip bits + trilinear gather),
    // then the camera-transform VJP over the per-slot partials
    rc = launch_gather_yx(st, S, plan, tA, pi, clip_mask, taps_x, taps_y, parts, chunk_sparse_marks(S, drc));
    if (rc) return rc;
    // clouds replicated inside the kernels: per-instance point gradients go to the workspace, their sum over a
    // cloud's instances to the caller's dpc [B / R, N, 3]
    const int R = P.views_per_cloud > 1 ? P.views_per_cloud : 1;
    float* dpc_views = R > 1 ? (float*)((char*)dsparts + dsparts_bytes(S)) : dpc;
    rc = launch_points_bwd_sorted(st, S, P, pc, pose, trans, focal, tr_pc, pi, dtr_pc_in, parts,
                                  plan.nstrips == 1 ? 2 : 4, dpc_views, dpose, dtrans,
                                  dfocal, scale ? dscale : nullptr, accum, (zfused && scale) ? dsparts : nullptr,
                                  nzb);
    if (rc || R == 1) return rc;
    const int L = 3 * S.N;
    if (L % 4 == 0)
      DPC_LAUNCH("sum_views", (k_sum_views<4>), dim3((L / 4 + DPC_BLOCK - 1) / DPC_BLOCK, S.B / R, 1), dim3(DPC_BLOCK, 1, 1), 0,
                 st, (const float*)dpc_views, dpc, R, L);
    else
      DPC_LAUNCH("sum_views", (k_sum_views<1>), dim3((L + DPC_BLOCK - 1) / DPC_BLOCK, S.B / R, 1), dim3(DPC_BLOCK, 1, 1), 0, st,
                 (const float*)dpc_views, dpc, R, L);
    return last_error();
  }
  // 2. y-blur adjoint (dense) -> tB ; the x-blur is evaluated sparsely in step 3
  const float* dg = tA;
  if (S.Ky > 0) {
    rc = launch_blur_plane(st, S, tA, tB, nullptr, taps_y, 0, S.Ky, 0, 1);
    if (rc) return rc;
    dg = tB;
  }
  // 3. sparse x-blur + clip mask + gather + transform VJP + reductions
  return launch_points_bwd(st, S, P, pc, pose, trans, focal, tr_pc, dg, grid_raw, taps_x, dtr_pc_in,
                           true, dpc, dpose, dtrans, dfocal, scale ? dscale : nullptr, accum, false,
                           (zfused && scale) ? dsparts : nullptr, nzb);
}

int dpc_silhouette_loss_fwd(dpc_stream_t stream, int B, int C, int D, int S, const float* proj, const float* gt,
                            const float* valid, float* inst_err, int32_t* winners, float* weight, float* loss) {
  if (B <= 0 || C <= 0 || B % C != 0 || D <= 0 || S < D || D > 4096) return DPC_E_SHAPE;
  if (!proj || !gt || !inst_err || !weight || !loss) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  DPC_LAUNCH("sil_err", (k_sil_err), dim3(B, 1, 1), dim3(DPC_BLOCK, 1, 1), 0, st, proj, gt, inst_err, C, D, S);
  DPC_LAUNCH("sil_select", (k_sil_select), dim3(1, 1, 1), dim3(DPC_BLOCK, 1, 1), 0, st, (const float*)inst_err, valid,
             (int*)winners, weight, loss, B / C, C);
  return last_error();
}

int dpc_silhouette_loss_bwd(dpc_stream_t stream, int B, int C, int D, int S, const float* proj, const float* gt,
                            const float* weight, const float* dloss, float* dproj) {
  if (B <= 0 || C <= 0 || B % C != 0 || D <= 0 || S < D || D > 4096) return DPC_E_SHAPE;
  if (!proj || !gt || !weight || !dloss || !dproj) return DPC_E_NULL;
  DPC_LAUNCH("sil_grad", (k_sil_grad), dim3((D * D + DPC_BLOCK - 1) / DPC_BLOCK, B, 1), dim3(DPC_BLOCK, 1, 1), 0,
             (hipStream_t)stream, proj, gt, weight, dloss, dproj, C, D, S, 1.f / (float)(B / C));
  return last_error();
}

int dpc_student_loss(dpc_stream_t stream, int n, int C, const float* poses, const int64_t* winners,
                     const float* student, const float* weights, float scale, float* loss, float* dstudent) {
  if (n <= 0 || C <= 0) return DPC_E_SHAPE;
  if (!poses || !winners || !student || !loss || !dstudent) return DPC_E_NULL;
  DPC_LAUNCH("student_loss", (k_student_loss), dim3(1, 1, 1), dim3(DPC_BLOCK, 1, 1), 0, (hipStream_t)stream, poses,
             (const long long*)winners, student, weights, n, C, scale, loss, dstudent);
  return last_error();
}

int dpc_nn_distance(dpc_stream_t stream, int dtype_bytes, int ns, int nt, const void* vs, const void* vt, void* proj,
                    void* min_dist, int32_t* idx) {
  if (ns <= 0 || nt <= 0) return DPC_E_SHAPE;
  if (dtype_bytes != 4 && dtype_bytes != 8) return DPC_E_MODE;
  if (!vs || !vt || !proj || !min_dist || !idx) return DPC_E_NULL;
  const dim3 block(256, 1, 1);
  const bool wide = ns > 32768;   // enough sources to fill the chip with 64 per work-group
  const dim3 grid(wide ? (ns + 63) / 64 : (ns + 15) / 16, 1, 1);
#define DPC_NN(T, PH, label)                                                                                   \
  DPC_LAUNCH(label, (k_nn_distance<T, PH>), grid, block, 0, (hipStream_t)stream, (const T*)vs, (const T*)vt, ns, nt, \
             (T*)proj, (T*)min_dist, (int*)idx)
  if (dtype_bytes == 8) {
    if (wide) DPC_NN(double, 4, "nn_distance_f64");
    else DPC_NN(double, 16, "nn_distance_f64");
  } else {
    if (wide) DPC_NN(float, 4, "nn_distance_f32");
    else DPC_NN(float, 16, "nn_distance_f32");
  }
#undef DPC_NN
  return last_error();
}

static int gv_args(int B, int N, int G, const int* perm, float sigma, int normalise, float* k, float* scale,
                   float* step) {
  if (B <= 0 || N <= 0 || G <= 0 || G > 512 || !(sigma > 0.f)) return DPC_E_SHAPE;
  if (!perm) return DPC_E_NULL;
  if (perm[0] < 0 || perm[0] > 2 || perm[1] < 0 || perm[1] > 2 || perm[2] < 0 || perm[2] > 2 ||
      perm[0] == perm[1] || perm[0] == perm[2] || perm[1] == perm[2])
    return DPC_E_MODE;
  if (normalise < 0 || normalise > 2) return DPC_E_MODE;
  *k = 1.f / (2.f * sigma * sigma);
  *step = G > 1 ? 2.f / (float)(G - 1) : 0.f;
  *scale = 1.f;
  if (normalise == DPC_GAUSS_NORM_ANALYTICAL) {
    const float sn = sigma * (float)G;                       // point_cloud.py:46-51
    *scale = 1.f / (1.78984352254f * sn * sn * sn);
  }
  return 0;
}

size_t dpc_gauss_voxelize_workspace_bytes(int B, int G) {
  if (B <= 0 || G <= 0) return 0;
  return ((size_t)B * G * G * G + DPC_GV_KC) * sizeof(float);
}

int dpc_gauss_voxelize_fwd(dpc_stream_t stream, int B, int N, int G, const int* perm, float sigma, int normalise,
                           const float* pc, float* inv_norm, float* raw, float* vox) {
  float k, scale, step;
  int rc = gv_args(B, N, G, perm, sigma, normalise, &k, &scale, &step);
  if (rc) return rc;
  if (!pc || !raw || !vox || (normalise == DPC_GAUSS_NORM_SUM && !inv_norm)) return DPC_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const float* inv = nullptr;
  if (normalise == DPC_GAUSS_NORM_SUM) {
    const int total = B * N * 3;
    DPC_LAUNCH("gv_norm", (k_gv_norm), dim3((total + DPC_BLOCK - 1) / DPC_BLOCK, 1, 1), dim3(DPC_BLOCK, 1, 1), 0, st,
               pc, inv_norm, total, G, k, step);
    inv = inv_norm;
  }
  const int nt = (G + DPC_GV_TILE - 1) / DPC_GV_TILE;
  DPC_LAUNCH("gv_fwd", (k_gv_fwd), dim3(B, G, nt * nt), dim3(256, 1, 1), 0, st, pc, inv, raw, vox, N, G, perm[0],
             perm[1], perm[2], k, scale, step);
  return last_error();
}

int dpc_gauss_voxelize_bwd(dpc_stream_t stream, int B, int N, int G, const int* perm, float sigma, int normalise,
                           const float* pc, const float* raw, const float* dvox, float* dpc, void* workspace,
                           size_t workspace_bytes) {
  float k, scale, step;
  int rc = gv_args(B, N, G, perm, sigma, normalise, &k, &scale, &step);
  if (rc) return rc;
  if (!pc || !raw || !dvox || !dpc) return DPC_E_NULL;
  if (!workspace || workspace_bytes < dpc_gauss_voxelize_workspace_bytes(B, G)) return DPC_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* gm = (float*)workspace;
  const size_t total = (size_t)B * G * G * G;
  DPC_LAUNCH("gv_mask", (k_gv_mask), dim3((unsigned)((total + DPC_GV_KC + DPC_BLOCK - 1) / DPC_BLOCK), 1, 1),
             dim3(DPC_BLOCK, 1, 1), 0, st, raw, dvox, gm, total, DPC_GV_KC);
  DPC_LAUNCH("gv_bwd", (k_gv_bwd), dim3((N + 63) / 64, B, 1), dim3(64, 1, 1), 0, st, pc, (const float*)gm, dpc, N, G,
             perm[0], perm[1], perm[2], k, scale, step, normalise == DPC_GAUSS_NORM_SUM ? 1 : 0);
  return last_error();
}

}  // extern "C"
