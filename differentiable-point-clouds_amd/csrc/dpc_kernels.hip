// dpc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) and the
// C ABI (include/dpc_hip.h) of the differentiable point-cloud projector.
//
// Fused hot path (power-of-two D in [32,256], K in {5,11,21}); grids are
// [B,Dz,D,D], x fastest; V = bytes of one grid of one view.  Planes that hold no
// trilinear mass (one bit per plane, from the depth-cell histogram) are neither
// written nor read anywhere on this path:
//
//   forward   k_zsort      WG/view: camera transform (quaternion or matrix) ->
//                          tr_pc, LDS counting sort of the points by depth cell,
//                          plane-occupancy bits
//             k_splat_xy   WG/(view, plane, y-strip): zero an LDS tile, ds_add_f32
//                          the plane's points, record one clip-gradient bit per
//                          touched corner, clip, x-blur (halo from neighbour
//                          lanes, ds_bpermute), y-blur (rotating register FIR)
//                          -> xy-blurred plane.                      writes 1 V
//             k_zfwd       thread = CX adjacent rays, streams z once: register
//                          z-FIR, writes G2 (saved), scale/clip, DRC collapse as a
//                          running transmittance product (no log/exp), silhouette
//                          + depth + two fp64 sums per ray.     reads 1 V, writes 1 V
//   backward  k_zbwd       same walk: DRC VJP (suffix sum = saved total - fp64
//                          prefix), scale/clip masks, dscale, z-FIR adjoint.
//                                                               reads 1 V, writes 1 V
//             k_gather_yx  WG/(view, plane, y-strip): rows -> LDS, y-blur in place, sparse
//                          x-blur + clip bits + trilinear gather at the plane's
//                          points -> per-corner partial d(tr_pc).   reads ~1.2 V
//             k_points_bwd thread/point: sum partials, camera-transform VJP,
//                          block reduction of dq/dt/df into a [B,16] accumulator
//                          (cleared by k_zbwd's first work-group per view)
//             k_pose_finalize  quaternion normalisation Jacobian; dscale = fixed-order
//                          sum of k_zbwd's per-work-group partials
//
// Outside the headline path, same ABI: k_sil_* (silhouette loss epilogue), k_gv_*
// (exact Gaussian voxeliser, the reference's pc_fast:false splat), k_nn_distance
// (nearest neighbour / Chamfer), k_scatter_vals / k_gather_vals (RGB channels).
//
// Generic path (any D, odd K <= 63, max-collapse, no blur, stage-level API):
//   k_points_fwd (transform + 8 global_atomic_add_f32 into zero-filled G0),
//   k_blur_xy_stream / k_blur_plane (plane blur, LDS-free / LDS-tiled),
//   k_blur_z / k_blur_z_generic, k_scatter, k_gather, k_max_fwd/bwd.
//
// No MFMA: this is scatter / stencil / scan work bounded by HBM (and, before
// the rewrites recorded in profiles/, by VALU issue and LDS crossbar time).
//
// The same source compiles for the CPU-only test tier with -DDPC_EMU (see
// tests/hipemu/hip_emu.h); that build is never loaded by the product.

#include "dpc_hip.h"

#ifdef DPC_EMU
#include "hip_emu.h"  // tests/hipemu: a thread-per-lane model of the HIP API used below (CPU test tier only)
#else
#include <hip/hip_runtime.h>
#endif

// dynamic LDS of the kernel as a typed pointer (16-byte aligned base)
#define DPC_DYN_SMEM(type, name)                     \
  HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, name##_raw) \
  type* name = reinterpret_cast<type*>(name##_raw)

#include <math.h>

#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------
// optional per-kernel timing: HIP events recorded on the launch stream around
// every launch while enabled (bench.py's roofline leg).  Off by default; not
// meant for concurrent use from several threads.
// ---------------------------------------------------------------------------
namespace dpcprof {
struct Rec {
  const char* label;
  hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;
static inline bool begin(const char* label, hipStream_t st) {
  if (!g_on) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.label = label;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return true;
}
static inline void end(hipStream_t st) {
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

#define DPC_LAUNCH(label, kernel, grid, block, smem, stream, ...)          \
  do {                                                                     \
    const bool prof_ = dpcprof::begin(label, stream);                      \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);    \
    if (prof_) dpcprof::end(stream);                                       \
  } while (0)

static inline hipError_t dpc_memset(const char* label, void* p, size_t n, hipStream_t st) {
  const bool prof_ = dpcprof::begin(label, st);
  hipError_t e = hipMemsetAsync(p, 0, n, st);
  if (prof_) dpcprof::end(st);
  return e;
}

#define DPC_BLOCK 256
#define DPC_XC 8  // x-blur outputs per thread (register window)
#define DPC_YC 8  // y-blur outputs per thread

namespace {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// clip_by_value = max(min(v, hi), lo); one v_med3_f32 on the GPU (inputs are never NaN here:
// NaN points are dropped before the grid)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

// 1/x, 1 ulp (v_rcp_f32); x is in [eps, 1] here
__device__ __forceinline__ float dpc_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct Quat {
  float w, x, y, z;
};
__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {  // quaternion.py:62-78
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ Quat qconj(const Quat& a) {
  Quat r = {a.w, -a.x, -a.y, -a.z};
  return r;
}

// Sum NV per-thread values over the block.  Result valid in thread 0 only.
// Must be called by every thread of the block (block-uniform control flow).
template <int NV>
__device__ __forceinline__ void block_reduce_sum(float (&v)[NV]) {
  __shared__ float red[NV * (DPC_BLOCK / 64)];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave * NV + i] = s;
  }
  __syncthreads();
  if (tid == 0) {
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[w * NV + i];
      v[i] = s;
    }
  }
  __syncthreads();
}

// exact floor(item / d) for 0 <= item < 2^20, 1 <= d <= 2^10
__device__ __forceinline__ int fast_div(int item, int d, float inv_d) {
  (void)d;
  return (int)(((float)item + 0.5f) * inv_d);
}

// ---------------------------------------------------------------------------
// camera pose, loaded per thread (uniform per block => scalar loads)
// ---------------------------------------------------------------------------
struct Pose {
  Quat q;       // normalised quaternion            (quaternion branch)
  float qnorm;  // |q| before normalisation
  float M[12];  // rows 0..2 of diag(1,f,f,1) * E   (matrix branch)
  float t[3];
  float f;
  float cd;
  bool has_t;
};

template <bool QUAT>
__device__ __forceinline__ void load_pose(const DpcParams& P, const float* __restrict__ pose,
                                          const float* __restrict__ trans,
                                          const float* __restrict__ focal, int b, Pose& o) {
  o.cd = P.camera_distance;
  o.has_t = false;
  o.t[0] = o.t[1] = o.t[2] = 0.f;
  if (QUAT) {
    const float* q = pose + 4 * b;
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);  // tf.norm
    o.qnorm = n;
    o.q.w = q[0] / n;
    o.q.x = q[1] / n;
    o.q.y = q[2] / n;
    o.q.z = q[3] / n;
    o.f = focal ? focal[b] : P.focal_length;
    if (trans) {
      o.has_t = true;
      o.t[0] = trans[3 * b + 0];
      o.t[1] = trans[3 * b + 1];
      o.t[2] = trans[3 * b + 2];
    }
  } else {
    const float* E = pose + 16 * b;
    o.f = P.focal_length;  // camera.py:5-13: the matrix branch always uses cfg.focal_length
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o.M[0 * 4 + j] = E[0 * 4 + j];
      o.M[1 * 4 + j] = o.f * E[1 * 4 + j];
      o.M[2 * 4 + j] = o.f * E[2 * 4 + j];
    }
  }
}

// pc_perspective_transform (point_cloud.py:157-216): p -> (w=depth, v=y, u=x)
template <bool QUAT>
__device__ __forceinline__ void transform_point(const Pose& ps, float p0, float p1, float p2,
                                                float& w, float& v, float& u) {
  if (QUAT) {
    Quat P = {0.f, p0, p1, p2};
    Quat r = qmul(qmul(ps.q, P), qconj(ps.q));
    float d = r.x, y = r.y, x = r.z;
    if (ps.has_t) {
      d += ps.t[0];
      y += ps.t[1];
      x += ps.t[2];
    }
    float zs = d + ps.cd;
    float xs = x * ps.f;
    float ys = y * ps.f;
    xs = xs / zs;
    ys = ys / zs;
    zs = zs - ps.cd;
    if (ps.has_t) zs = zs - ps.t[0];
    w = zs;
    v = ys;
    u = xs;
  } else {
    const float* M = ps.M;
    float zs = M[0] * p0 + M[1] * p1 + M[2] * p2 + M[3];
    float ys = M[4] * p0 + M[5] * p1 + M[6] * p2 + M[7];
    float xs = M[8] * p0 + M[9] * p1 + M[10] * p2 + M[11];
    u = xs / zs;
    v = ys / zs;
    w = zs - ps.cd;
  }
}

// VJP of transform_point.  acc[16]: quaternion: [0..3]=dq_hat, [4..6]=dtrans,
// [7]=dfocal; matrix: [0..11]=dM (rows 0..2 of d(intr*E)).
template <bool QUAT>
__device__ __forceinline__ void transform_point_bwd(const Pose& ps, float p0, float p1, float p2,
                                                    float dw, float dv, float du, float& g0,
                                                    float& g1, float& g2, float (&acc)[16]) {
  if (QUAT) {
    Quat P = {0.f, p0, p1, p2};
    Quat t = qmul(ps.q, P);
    Quat r = qmul(t, qconj(ps.q));
    float d = r.x, y = r.y, x = r.z;
    if (ps.has_t) {
      d += ps.t[0];
      y += ps.t[1];
      x += ps.t[2];
    }
    const float Z = d + ps.cd;
    const float u = ps.f * x / Z, v = ps.f * y / Z;
    const float dx = du * ps.f / Z;
    const float dy = dv * ps.f / Z;
    const float dZ = -(du * u + dv * v) / Z;
    acc[7] += (du * x + dv * y) / Z;
    acc[4] += dZ;  // d w / d t0 cancels (w = Z - cd - t0)
    acc[5] += dy;
    acc[6] += dx;
    Quat dr = {0.f, dw + dZ, dy, dx};
    // reverse of r = t (x) q*, t = q (x) P:  <dc, a(x)b>  =>  da = dc (x) b*, db = a* (x) dc
    Quat dt = qmul(dr, ps.q);
    Quat dqc = qmul(qconj(t), dr);
    Quat dq2 = qmul(dt, qconj(P));
    acc[0] += dqc.w + dq2.w;
    acc[1] += -dqc.x + dq2.x;
    acc[2] += -dqc.y + dq2.y;
    acc[3] += -dqc.z + dq2.z;
    Quat dP = qmul(qconj(ps.q), dt);
    g0 = dP.x;
    g1 = dP.y;
    g2 = dP.z;
  } else {
    const float* M = ps.M;
    const float Z = M[0] * p0 + M[1] * p1 + M[2] * p2 + M[3];
    const float ys = M[4] * p0 + M[5] * p1 + M[6] * p2 + M[7];
    const float xs = M[8] * p0 + M[9] * p1 + M[10] * p2 + M[11];
    const float u = xs / Z, v = ys / Z;
    const float d2 = du / Z, d1 = dv / Z, d0 = dw - (du * u + dv * v) / Z;
    const float h[4] = {p0, p1, p2, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0 + j] += d0 * h[j];
      acc[4 + j] += d1 * h[j];
      acc[8 + j] += d2 * h[j];
    }
    g0 = d0 * M[0] + d1 * M[4] + d2 * M[8];
    g1 = d0 * M[1] + d1 * M[5] + d2 * M[9];
    g2 = d0 * M[2] + d1 * M[6] + d2 * M[10];
  }
}

// pointcloud2voxels3d_fast cell lookup (point_cloud.py:76-92)
struct Cell {
  int iz, iy, ix;
  float rz, ry, rx;
  bool valid;
};
__device__ __forceinline__ Cell locate(float w, float v, float u, int Dz, int D) {
  Cell c;
  c.valid = (w >= -0.5f) && (w <= 0.5f) && (v >= -0.5f) && (v <= 0.5f) && (u >= -0.5f) && (u <= 0.5f);
  const float gz = (w + 0.5f) * (float)(Dz - 1);
  const float gy = (v + 0.5f) * (float)(D - 1);
  const float gx = (u + 0.5f) * (float)(D - 1);
  const float fz = floorf(gz), fy = floorf(gy), fx = floorf(gx);
  c.iz = c.valid ? (int)fz : 0;
  c.iy = c.valid ? (int)fy : 0;
  c.ix = c.valid ? (int)fx : 0;
  c.rz = gz - fz;
  c.ry = gy - fy;
  c.rx = gx - fx;
  return c;
}

__device__ __forceinline__ void scatter_point(float* __restrict__ grid, int b, int Dz, int D, float w,
                                              float v, float u) {
  const Cell c = locate(w, v, u, Dz, D);
  if (!c.valid) return;
  const float wz[2] = {1.0f - c.rz, c.rz};
  const float wy[2] = {1.0f - c.ry, c.ry};
  const float wx[2] = {1.0f - c.rx, c.rx};
  float* g = grid + (size_t)b * Dz * D * D;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int zz = c.iz + k, yy = c.iy + j, xx = c.ix + l;
        if (zz < Dz && yy < D && xx < D)  // == size only with weight 0 (coordinate exactly +0.5)
          atomicAdd(g + ((size_t)zz * D + yy) * D + xx, wz[k] * wy[j] * wx[l]);
      }
}

// Trilinear gather VJP.  dgrid is either d(G0) itself (Kx == 0, mask == null)
// or the (z,y)-blurred gradient, in which case the x-blur (taps_x, Kx) is
// evaluated here, only at the <= 8 touched cells, and `mask` (= G0, dense) or
// `cmask` (per-point corner bits) applies the clip_by_value(.,0,1) gradient
// mask of point_cloud.py:240.
template <int KC>
__device__ __forceinline__ void gather_point(const float* __restrict__ dgrid,
                                             const float* __restrict__ mask,
                                             const unsigned char* __restrict__ cmask /*4 bytes of this point*/,
                                             const float* __restrict__ taps_x, int Kx, int b, int Dz,
                                             int D, float w, float v, float u, float& dw, float& dv,
                                             float& du) {
  dw = dv = du = 0.f;
  const Cell c = locate(w, v, u, Dz, D);
  if (!c.valid) return;  // boolean_mask gradient: zeros at dropped rows
  const float wz[2] = {1.0f - c.rz, c.rz};
  const float wy[2] = {1.0f - c.ry, c.ry};
  const float wx[2] = {1.0f - c.rx, c.rx};
  const size_t base_b = (size_t)b * Dz * D * D;
  const int h = Kx >> 1;
  float drz = 0.f, dry = 0.f, drx = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int zz = c.iz + k, yy = c.iy + j;
      if (zz >= Dz || yy >= D) continue;
      const size_t row = base_b + ((size_t)zz * D + yy) * D;
      float g[2] = {0.f, 0.f};
      if (KC > 0) {
        // compile-time K: all K+1 window loads are unconditional (clamped address,
        // value zeroed when outside the row) so they are issued back to back
        float win[(KC > 0 ? KC : 1) + 1];
#pragma unroll
        for (int m = 0; m <= KC; ++m) {
          const int x = c.ix - KC / 2 + m;
          const int xc = x < 0 ? 0 : (x >= D ? D - 1 : x);
          const float val = dgrid[row + xc];
          win[m] = (x == xc) ? val : 0.f;
        }
#pragma unroll
        for (int m = 0; m < KC; ++m) {
          g[0] += taps_x[m] * win[m];
          g[1] += taps_x[m] * win[m + 1];
        }
      } else if (Kx > 0) {
        // window x in [ix-h, ix+1+h]; tap m of output l sits at x = ix + l + m - h
        for (int m = 0; m <= Kx; ++m) {
          const int x = c.ix - h + m;
          if (x < 0 || x >= D) continue;
          const float val = dgrid[row + x];
          if (m < Kx) g[0] += taps_x[m] * val;
          if (m >= 1) g[1] += taps_x[m - 1] * val;
        }
      } else {
        g[0] = dgrid[row + c.ix];
        if (c.ix + 1 < D) g[1] = dgrid[row + c.ix + 1];
      }
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int xx = c.ix + l;
        if (xx >= D) continue;
        float gg = g[l];
        if (cmask) {  // bit l of byte (k,j): 0 <= G0 <= 1 at that corner (written by k_splat_xy)
          gg = ((cmask[k * 2 + j] >> l) & 1) ? gg : 0.f;
        } else if (mask) {
          const float m0 = mask[row + xx];
          gg = (m0 >= 0.f && m0 <= 1.f) ? gg : 0.f;
        }
        drz += gg * (k ? 1.f : -1.f) * wy[j] * wx[l];
        dry += gg * wz[k] * (j ? 1.f : -1.f) * wx[l];
        drx += gg * wz[k] * wy[j] * (l ? 1.f : -1.f);
      }
    }
  dw = drz * (float)(Dz - 1);
  dv = dry * (float)(D - 1);
  du = drx * (float)(D - 1);
}

}  // namespace

// ===========================================================================
// point kernels
// ===========================================================================
template <bool QUAT>
__global__ void __launch_bounds__(DPC_BLOCK)
k_points_fwd(DpcShape S, DpcParams P, const float* __restrict__ pc, const float* __restrict__ pose,
             const float* __restrict__ trans, const float* __restrict__ focal,
             float* __restrict__ tr_pc, float* __restrict__ grid /*nullable*/) {
  const int b = blockIdx.x;  // view-major block ids: b + B*chunk => one XCD (L2) per view when B % 8 == 0
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= S.N) return;
  Pose ps;
  load_pose<QUAT>(P, pose, trans, focal, b, ps);
  const size_t o = ((size_t)b * S.N + n) * 3;
  float w, v, u;
  transform_point<QUAT>(ps, pc[o], pc[o + 1], pc[o + 2], w, v, u);
  tr_pc[o] = w;
  tr_pc[o + 1] = v;
  tr_pc[o + 2] = u;
  if (grid) scatter_point(grid, b, S.Dz, S.D, w, v, u);
}

__global__ void __launch_bounds__(DPC_BLOCK)
k_scatter(DpcShape S, const float* __restrict__ tr_pc, float* __restrict__ grid) {
  const int b = blockIdx.x;  // view-major block ids: b + B*chunk => one XCD (L2) per view when B % 8 == 0
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= S.N) return;
  const size_t o = ((size_t)b * S.N + n) * 3;
  scatter_point(grid, b, S.Dz, S.D, tr_pc[o], tr_pc[o + 1], tr_pc[o + 2]);
}

__global__ void __launch_bounds__(DPC_BLOCK)
k_gather(DpcShape S, const float* __restrict__ tr_pc, const float* __restrict__ dgrid,
         float* __restrict__ dtr_pc) {
  const int b = blockIdx.x;  // view-major block ids: b + B*chunk => one XCD (L2) per view when B % 8 == 0
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= S.N) return;
  const size_t o = ((size_t)b * S.N + n) * 3;
  float dw, dv, du;
  gather_point<0>(dgrid, nullptr, nullptr, nullptr, 0, b, S.Dz, S.D, tr_pc[o], tr_pc[o + 1], tr_pc[o + 2], dw, dv, du);
  dtr_pc[o] = dw;
  dtr_pc[o + 1] = dv;
  dtr_pc[o + 2] = du;
}

// RGB channels (point_cloud.py:111-118): per-point values vals[b,n,c] spread with the same
// trilinear weights into a channel-major grid [B,C,Dz,D,D] (so that the scalar blur kernels
// apply to it as B*C views), and the VJP w.r.t. the values and the point positions.
__global__ void __launch_bounds__(DPC_BLOCK)
k_scatter_vals(DpcShape S, int C, const float* __restrict__ tr_pc, const float* __restrict__ vals,
               float* __restrict__ grid) {
  const int b = blockIdx.x;
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= S.N) return;
  const size_t o = ((size_t)b * S.N + n) * 3;
  const int Dz = S.Dz, D = S.D;
  const Cell c = locate(tr_pc[o], tr_pc[o + 1], tr_pc[o + 2], Dz, D);
  if (!c.valid) return;
  const float wz[2] = {1.0f - c.rz, c.rz};
  const float wy[2] = {1.0f - c.ry, c.ry};
  const float wx[2] = {1.0f - c.rx, c.rx};
  const size_t V = (size_t)Dz * D * D;
  for (int ch = 0; ch < C; ++ch) {
    const float val = vals[((size_t)b * S.N + n) * C + ch];
    float* g = grid + ((size_t)b * C + ch) * V;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const int zz = c.iz + k, yy = c.iy + j, xx = c.ix + l;
          if (zz < Dz && yy < D && xx < D) atomicAdd(g + ((size_t)zz * D + yy) * D + xx, wz[k] * wy[j] * wx[l] * val);
        }
  }
}

__global__ void __launch_bounds__(DPC_BLOCK)
k_gather_vals(DpcShape S, int C, const float* __restrict__ tr_pc, const float* __restrict__ vals,
              const float* __restrict__ dgrid, float* __restrict__ dvals, float* __restrict__ dtr_pc /*nullable*/) {
  const int b = blockIdx.x;
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= S.N) return;
  const size_t o = ((size_t)b * S.N + n) * 3;
  const int Dz = S.Dz, D = S.D;
  const Cell c = locate(tr_pc[o], tr_pc[o + 1], tr_pc[o + 2], Dz, D);
  float drz = 0.f, dry = 0.f, drx = 0.f;
  const float wz[2] = {1.0f - c.rz, c.rz};
  const float wy[2] = {1.0f - c.ry, c.ry};
  const float wx[2] = {1.0f - c.rx, c.rx};
  const size_t V = (size_t)Dz * D * D;
  for (int ch = 0; ch < C; ++ch) {
    float dv = 0.f;
    if (c.valid) {
      const float val = vals[((size_t)b * S.N + n) * C + ch];
      const float* g = dgrid + ((size_t)b * C + ch) * V;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int l = 0; l < 2; ++l) {
            const int zz = c.iz + k, yy = c.iy + j, xx = c.ix + l;
            if (zz < Dz && yy < D && xx < D) {
              const float gg = g[((size_t)zz * D + yy) * D + xx];
              dv += gg * wz[k] * wy[j] * wx[l];
              const float gv = gg * val;
              drz += gv * (k ? 1.f : -1.f) * wy[j] * wx[l];
              dry += gv * wz[k] * (j ? 1.f : -1.f) * wx[l];
              drx += gv * wz[k] * wy[j] * (l ? 1.f : -1.f);
            }
          }
    }
    dvals[((size_t)b * S.N + n) * C + ch] = dv;
  }
  if (dtr_pc) {
    dtr_pc[o] = drz * (float)(Dz - 1);
    dtr_pc[o + 1] = dry * (float)(D - 1);
    dtr_pc[o + 2] = drx * (float)(D - 1);
  }
}

// Gather (+ sparse x-blur + clip mask) + camera-transform VJP + per-instance
// reductions.  GATHER=false: d(tr_pc) is read from dtr_in instead.
template <bool QUAT, bool GATHER, int KC>
__global__ void __launch_bounds__(DPC_BLOCK)
k_points_bwd(DpcShape S, DpcParams P, const float* __restrict__ pc, const float* __restrict__ pose,
             const float* __restrict__ trans, const float* __restrict__ focal,
             const float* __restrict__ tr_pc, const float* __restrict__ dgrid,
             const float* __restrict__ mask, const unsigned char* __restrict__ cmask,
             const float* __restrict__ taps_x, const float* __restrict__ dtr_in /*nullable when GATHER*/,
             const float* __restrict__ parts /*nullable: [B,N,4,3] from k_gather_yx*/,
             float* __restrict__ dpc, float* __restrict__ accum /*[B,16], zeroed*/) {
  const int b = blockIdx.x;  // view-major block ids: b + B*chunk => one XCD (L2) per view when B % 8 == 0
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (n < S.N) {
    Pose ps;
    load_pose<QUAT>(P, pose, trans, focal, b, ps);
    const size_t o = ((size_t)b * S.N + n) * 3;
    float dw = 0.f, dv = 0.f, du = 0.f;
    if (GATHER)
      gather_point<KC>(dgrid, mask, cmask ? cmask + ((size_t)b * S.N + n) * 4 : nullptr, taps_x, S.Kx, b, S.Dz, S.D,
                   tr_pc[o], tr_pc[o + 1], tr_pc[o + 2], dw, dv, du);
    if (parts) {  // one [3] slot per (corner plane k, corner row j), written by its owning WG
      const Cell c = locate(tr_pc[o], tr_pc[o + 1], tr_pc[o + 2], S.Dz, S.D);
      const float* pp = parts + ((size_t)b * S.N + n) * 12;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // slots of corners outside the grid (and of dropped points) were never written
        if (c.valid && c.iz + (q >> 1) < S.Dz && c.iy + (q & 1) < S.D) {
          dw += pp[3 * q];
          dv += pp[3 * q + 1];
          du += pp[3 * q + 2];
        }
      }
    }
    if (dtr_in) {
      dw += dtr_in[o];
      dv += dtr_in[o + 1];
      du += dtr_in[o + 2];
    }
    float g0, g1, g2;
    transform_point_bwd<QUAT>(ps, pc[o], pc[o + 1], pc[o + 2], dw, dv, du, g0, g1, g2, acc);
    dpc[o] = g0;
    dpc[o + 1] = g1;
    dpc[o + 2] = g2;
  }
  block_reduce_sum<16>(acc);
  if (threadIdx.x == 0) {
    const int nacc = QUAT ? 8 : 12;
    for (int i = 0; i < nacc; ++i) atomicAdd(accum + 16 * b + i, acc[i]);
  }
}

template <bool QUAT>
__global__ void k_pose_finalize(DpcShape S, DpcParams P, const float* __restrict__ pose,
                                const float* __restrict__ accum, float* __restrict__ dpose,
                                float* __restrict__ dtrans, float* __restrict__ dfocal,
                                float* __restrict__ dscale /*nullable: from accumulator slot 15 or the partials*/,
                                const float* __restrict__ dsparts /*nullable: [B,nzb] from k_zbwd*/, int nzb) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  const float* a = accum + 16 * b;
  if (dscale) {
    float ds = 0.f;
    if (dsparts) {
      for (int i = 0; i < nzb; ++i) ds += dsparts[(size_t)b * nzb + i];
    } else {
      ds = a[15];
    }
    dscale[b] = ds;
  }
  if (QUAT) {
    const float* q = pose + 4 * b;
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float qh[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
    const float dot = qh[0] * a[0] + qh[1] * a[1] + qh[2] * a[2] + qh[3] * a[3];
    for (int i = 0; i < 4; ++i) dpose[4 * b + i] = (a[i] - qh[i] * dot) / n;  // (I - q q^T)/|q|
    if (dtrans)
      for (int i = 0; i < 3; ++i) dtrans[3 * b + i] = a[4 + i];
    if (dfocal) dfocal[b] = a[7];
  } else {
    const float f = P.focal_length;
    for (int j = 0; j < 4; ++j) {
      dpose[16 * b + 0 + j] = a[0 + j];
      dpose[16 * b + 4 + j] = f * a[4 + j];
      dpose[16 * b + 8 + j] = f * a[8 + j];
      dpose[16 * b + 12 + j] = 0.f;
    }
  }
}

// ===========================================================================
// plane blur: x then y on a (TY + 2 hy) x D tile staged in LDS
// ===========================================================================
// KC > 0: compile-time tap count (both Kx and Ky equal KC when enabled);
// KC == 0: run-time tap counts (gather form, K LDS reads per output).
template <int KC>
__global__ void __launch_bounds__(DPC_BLOCK)
k_blur_plane(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps_x,
             const float* __restrict__ taps_y, int Kx, int Ky, int Dz, int D, int TY, int nyt, int PA,
             int PB, int clip_in, int nblocks) {
  DPC_DYN_SMEM(float, smem);
  // XCD-aware remap (bijective): consecutive logical tiles (which share halo
  // rows) stay on one XCD's L2.  Placement only affects speed.
  int bid = blockIdx.x;
  {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int yt = bid % nyt;
  const int pz = bid / nyt;  // = b * Dz + z
  const int hx = Kx >> 1, hy = Ky >> 1;
  const int R = TY + 2 * hy;
  const int y0 = yt * TY;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwave = nth >> 6;
  float* A = smem;            // [R][PA]: input tile, x halo of hx zeros each side (+ pad)
  float* Bm = smem + R * PA;  // [R][PB]: x-blurred tile
  const float* plane = in + (size_t)pz * D * D;

  // ---- stage 1: global -> LDS (clip fused), zero halos / out-of-range rows
  for (int r = wave; r < R; r += nwave) {
    const int gy = y0 - hy + r;
    const bool rowok = (gy >= 0) && (gy < D);
    const float* src = plane + (size_t)(rowok ? gy : 0) * D;
    for (int c = lane; c < PA; c += 64) {
      const int x = c - hx;
      float val = 0.f;
      if (rowok && x >= 0 && x < D) {
        val = src[x];
        if (clip_in) val = clampf(val, 0.f, 1.f);
      }
      A[r * PA + c] = val;
    }
  }
  __syncthreads();

  // ---- stage 2: x-blur A -> Bm (lanes walk rows: odd pitches => no conflicts)
  const float* S3 = A;  // source of stage 3
  int PS = PA;
  if (Kx > 0) {
    const int nxc = (D + DPC_XC - 1) / DPC_XC;
    const float invR = 1.0f / (float)R;
    for (int item = tid; item < R * nxc; item += nth) {
      const int xc = fast_div(item, R, invR);
      const int r = item - xc * R;
      const float* src = A + r * PA + xc * DPC_XC;
      float* dst = Bm + r * PB + xc * DPC_XC;
      if (KC > 0) {
        float win[DPC_XC + (KC > 0 ? KC : 1) - 1];
#pragma unroll
        for (int i = 0; i < DPC_XC + KC - 1; ++i) win[i] = src[i];
#pragma unroll
        for (int o = 0; o < DPC_XC; ++o) {
          float a = 0.f;
#pragma unroll
          for (int m = 0; m < KC; ++m) a += taps_x[m] * win[o + m];
          if (xc * DPC_XC + o < D) dst[o] = a;
        }
      } else {
        for (int o = 0; o < DPC_XC; ++o) {
          if (xc * DPC_XC + o >= D) break;
          float a = 0.f;
          for (int m = 0; m < Kx; ++m) a += taps_x[m] * src[o + m];
          dst[o] = a;
        }
      }
    }
    __syncthreads();
    S3 = Bm;
    PS = PB;
  }

  // ---- stage 3: y-blur -> global (lanes walk x: coalesced stores)
  float* oplane = out + (size_t)pz * D * D;
  const int nyc = TY / DPC_YC;
  const float invD = 1.0f / (float)D;
  for (int item = tid; item < D * nyc; item += nth) {
    const int yc = fast_div(item, D, invD);
    const int x = item - yc * D;
    const float* src = S3 + (yc * DPC_YC) * PS + x;
    const int gy0 = y0 + yc * DPC_YC;
    if (Ky > 0) {
      if (KC > 0) {
        float win[DPC_YC + (KC > 0 ? KC : 1) - 1];
#pragma unroll
        for (int i = 0; i < DPC_YC + KC - 1; ++i) win[i] = src[i * PS];
#pragma unroll
        for (int o = 0; o < DPC_YC; ++o) {
          float a = 0.f;
#pragma unroll
          for (int m = 0; m < KC; ++m) a += taps_y[m] * win[o + m];
          if (gy0 + o < D) oplane[(size_t)(gy0 + o) * D + x] = a;
        }
      } else {
        for (int o = 0; o < DPC_YC; ++o) {
          if (gy0 + o >= D) break;
          float a = 0.f;
          for (int m = 0; m < Ky; ++m) a += taps_y[m] * src[(o + m) * PS];
          oplane[(size_t)(gy0 + o) * D + x] = a;
        }
      }
    } else {
      for (int o = 0; o < DPC_YC; ++o)
        if (gy0 + o < D) oplane[(size_t)(gy0 + o) * D + x] = src[o * PS];
    }
  }
}

// ===========================================================================
// z streaming kernels: one thread owns CX adjacent rays (y,x .. x+CX-1)
// ===========================================================================
template <int CX>
__device__ __forceinline__ void load_cx(const float* __restrict__ p, float (&v)[CX]) {
  if (CX == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x;
    v[1 % CX] = t.y;
    v[2 % CX] = t.z;
    v[3 % CX] = t.w;
  } else if (CX == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x;
    v[1 % CX] = t.y;
  } else {
    v[0] = p[0];
  }
}
template <int CX>
__device__ __forceinline__ void store_cx(float* __restrict__ p, const float (&v)[CX]) {
  if (CX == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % CX], v[2 % CX], v[3 % CX]);
  } else if (CX == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1 % CX]);
  } else {
    p[0] = v[0];
  }
}

// Register FIR in "accumulate" form: pushing input plane t completes output
// plane t - h (zero padding falls out because only real planes are pushed).
// The K partial sums live in a ROTATING set of registers: inside a loop body
// unrolled over a group of G steps (G a multiple of K) the slot of logical
// accumulator k at step u is (k + u) % K, a compile-time constant, so no
// register-to-register shifting is ever executed.
typedef float dpc_v2f __attribute__((vector_size(8)));  // one v_pk_fma_f32 operand pair

template <int KC, int CX>
struct ZFir {
  static constexpr int NP = CX / 2;        // packed pairs (explicit 2-vectors => v_pk_fma_f32)
  static constexpr int NS = CX - 2 * NP;   // odd leftover lane value
  dpc_v2f accp[KC][NP > 0 ? NP : 1];
  float accs[KC][NS > 0 ? NS : 1];
  float tp[KC];
  __device__ __forceinline__ void init(const float* __restrict__ taps) {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      tp[j] = taps ? taps[j] : 1.0f;
#pragma unroll
      for (int c = 0; c < NP; ++c) accp[j][c] = dpc_v2f{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NS; ++c) accs[j][c] = 0.f;
    }
  }
  // u = step index inside the unrolled group (compile-time after unrolling)
  __device__ __forceinline__ void push(const float (&v)[CX], float (&out)[CX], int u) {
    dpc_v2f vp[NP > 0 ? NP : 1];
#pragma unroll
    for (int c = 0; c < NP; ++c) vp[c] = dpc_v2f{v[2 * c], v[2 * c + 1]};
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const float t = tp[KC - 1 - k];
      const dpc_v2f tt = dpc_v2f{t, t};
#pragma unroll
      for (int c = 0; c < NP; ++c) accp[(k + u) % KC][c] += tt * vp[c];
#pragma unroll
      for (int c = 0; c < NS; ++c) accs[(k + u) % KC][c] += t * v[2 * NP + c];
    }
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      out[2 * c] = accp[u % KC][c][0];
      out[2 * c + 1] = accp[u % KC][c][1];
      accp[u % KC][c] = dpc_v2f{0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      out[2 * NP + c] = accs[u % KC][c];
      accs[u % KC][c] = 0.f;
    }
  }
};
// group length: a multiple of KC, at least 4 planes/rows (= loads kept in flight per lane)
constexpr int zgroup(int KC) { return KC >= 4 ? KC : KC * ((4 + KC - 1) / KC); }

// ===========================================================================
// plane blur, streaming form (power-of-two D <= 256): no LDS, no barriers.
// Lanes lie along x with 4 floats each; a row is LR = D/4 lanes, so one wave
// covers 64/LR planes side by side and marches down y.  The x-blur pulls its
// halo from neighbour lanes with ds_bpermute (__shfl); the y-blur is a register
// FIR (ZFir).  A whole group of rows is loaded (unconditionally, from clamped
// addresses) one group ahead, so every lane keeps G 16-byte loads in flight.
// ===========================================================================
template <int KC, bool DO_X, bool DO_Y>
__global__ void __launch_bounds__(DPC_BLOCK)
k_blur_xy_stream(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps_x,
                 const float* __restrict__ taps_y, int nplanes, int D, int lr_shift, int clip_in) {
  constexpr int h = KC / 2;
  constexpr int G = zgroup(KC);
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int LR = 1 << lr_shift;
  const int sub = lane >> lr_shift;
  const int lx = lane & (LR - 1);
  const int plane = wave * (64 >> lr_shift) + sub;
  const bool ok = plane < nplanes;
  const size_t pbase = (size_t)(ok ? plane : 0) * D * D + (size_t)lx * 4;
  float tpx[KC];
#pragma unroll
  for (int m = 0; m < KC; ++m) tpx[m] = DO_X ? taps_x[m] : 0.f;
  ZFir<KC, 4> fir;
  fir.init(DO_Y ? taps_y : nullptr);
  const int T = D + (DO_Y ? h : 0);

  float cur[G][4], nxt[G][4];
#pragma unroll
  for (int u = 0; u < G; ++u) load_cx<4>(in + pbase + (size_t)(u < D ? u : D - 1) * D, cur[u]);
  for (int y0 = 0; y0 < T; y0 += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int yy = y0 + G + u;
      load_cx<4>(in + pbase + (size_t)(yy < D ? yy : D - 1) * D, nxt[u]);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int y = y0 + u;
      if (y < T) {  // uniform
        float v[4], xb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float r = (y < D) ? cur[u][c] : 0.f;
          v[c] = clip_in ? clampf(r, 0.f, 1.f) : r;
        }
        if (DO_X && y < D) {
          float w[4 + 2 * h];
#pragma unroll
          for (int c = 0; c < 4; ++c) w[h + c] = v[c];
#pragma unroll
          for (int e = 1; e <= h; ++e) {
            // left halo element x = 4*lx - e ; right halo element x = 4*lx + 3 + e
            const int dl = (e + 3) / 4;       // lanes to the left / right
            const int jl = (4 * dl - e) & 3;  // element index inside that lane
            const float vl = __shfl(v[jl], (lane - dl) & 63, 64);
            w[h - e] = (lx - dl >= 0) ? vl : 0.f;
            const int jr = (e - 1) & 3;
            const float vr = __shfl(v[jr], (lane + dl) & 63, 64);
            w[h + 3 + e] = (lx + dl < LR) ? vr : 0.f;
          }
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < KC; ++m) a += tpx[m] * w[o + m];
            xb[o] = a;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) xb[c] = v[c];
        }
        if (DO_Y) {
          float o[4];
          fir.push(xb, o, u);
          if (y >= h && ok) store_cx<4>(out + pbase + (size_t)(y - h) * D, o);
        } else {
          if (ok) store_cx<4>(out + pbase + (size_t)y * D, xb);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) cur[u][c] = nxt[u][c];
  }
}

// ===========================================================================
// fused forward front end: z-bucketed points -> per-plane LDS splat -> clip ->
// x,y blur -> xy-blurred plane.  Replaces {zero-fill G0, global float atomics,
// read G0 back} of the generic path: the raw grid G0 never exists in HBM.  The
// clip_by_value(G0,0,1) gradient mask that backward needs is kept as one bit
// per touched corner (cmask [B,N,4] bytes: byte k*2+j, bit l).
// ===========================================================================

// Plane occupancy: plane z of a view holds trilinear mass iff a valid point sits in depth cell z-1
// or z, i.e. zstart[z+1] > zstart[max(z-1,0)].  k_zsort packs that into DPC_LIVE_WORDS x 32 bits per
// view; planes without mass are never written by k_splat_xy nor read by k_zfwd, and the planes
// k_gather_yx skips (same test) are never written by k_zbwd.  Objects rarely span the whole depth
// range of the lattice, so a sizeable share of the planes is free.
#define DPC_LIVE_WORDS 8   // Dz <= 256
// every thread of the work-group calls; zs = the view's bucket starts in LDS
__device__ __forceinline__ void write_live_words(const int* zs, int Dz, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
  for (int z0 = wave * 64; z0 < 32 * DPC_LIVE_WORDS; z0 += nwave * 64) {
    const int z = z0 + lane;
    const unsigned long long m = __ballot(z < Dz && zs[(z < Dz ? z : 0) + 1] > zs[(z > 0 && z < Dz) ? z - 1 : 0]);
    if (lane == 0) {
      out[z0 >> 5] = (unsigned)m;
      out[(z0 >> 5) + 1] = (unsigned)(m >> 32);
    }
  }
}
struct LiveMask {
  unsigned w[DPC_LIVE_WORDS];
  // live == nullptr: every plane counts as occupied (dense producers / consumers)
  __device__ __forceinline__ void load(const unsigned* __restrict__ live, int b) {
#pragma unroll
    for (int k = 0; k < DPC_LIVE_WORDS; ++k) w[k] = live ? live[(size_t)b * DPC_LIVE_WORDS + k] : 0xffffffffu;
  }
  // occupancy bits of planes s .. s+31 (bit u = plane s+u; planes outside [0, 32*DPC_LIVE_WORDS) read 0).
  // Uniform over the block (b is a block index): pinned to a scalar register, so that the per-plane
  // tests `(win >> u) & 1` with compile-time u cost one scalar op each.
  __device__ __forceinline__ unsigned window(int s) const {
    const int i = s >> 5, sft = s & 31;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < DPC_LIVE_WORDS; ++k) {
      lo = (i == k) ? w[k] : lo;
      hi = (i + 1 == k) ? w[k] : hi;
    }
    const unsigned r = sft ? ((lo >> sft) | (hi << (32 - sft))) : lo;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
  }
};

// In-place exclusive prefix sum of h[0..M) over the work-group, total into h[M] (M <= a few hundred:
// the per-view depth-cell histogram).  Every thread of the block must call it; ends with a barrier.
__device__ __forceinline__ void block_exclusive_scan(int* h, int M) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, nth = blockDim.x;
  const int per = (M + nth - 1) / nth;
  const int lo = tid * per < M ? tid * per : M, hi = lo + per < M ? lo + per : M;
  int own = 0;
  for (int i = lo; i < hi; ++i) own += h[i];
  part[tid] = own;
  __syncthreads();
  for (int off = 1; off * per < M && off < nth; off <<= 1) {   // threads beyond the last owner hold 0
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - own;
  for (int i = lo; i < hi; ++i) {
    const int c = h[i];
    h[i] = run;
    run += c;
  }
  if (hi == M && lo < M) h[M] = run;   // the owner of the last entry also knows the total
  __syncthreads();
}

// camera transform of one view's points (-> tr_pc) followed by an LDS counting
// sort by depth cell iz (bin Dz = dropped points)
template <bool QUAT>
__global__ void __launch_bounds__(1024)
k_zsort(DpcShape S, DpcParams P, const float* __restrict__ pc, const float* __restrict__ pose,
        const float* __restrict__ trans, const float* __restrict__ focal, float* __restrict__ tr_pc,
        int* __restrict__ order, int* __restrict__ zstart, unsigned* __restrict__ live) {
  DPC_DYN_SMEM(int, hist);  // [Dz + 2]
  const int b = blockIdx.x;
  const int N = S.N, Dz = S.Dz, D = S.D;
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int i = tid; i < Dz + 2; i += nth) hist[i] = 0;
  __syncthreads();
  Pose ps;
  load_pose<QUAT>(P, pose, trans, focal, b, ps);
  float* tp = tr_pc + (size_t)b * N * 3;
  const float* pp = pc + (size_t)b * N * 3;
  // points are handled in batches of PB per thread: all PB loads are issued before the first
  // use, and the bin of every point stays in a register for the scatter pass (one WG per view
  // leaves the latency of each dependent trip to memory fully exposed otherwise)
  constexpr int PB = 8;
  if (N <= PB * nth) {
    float p0[PB], p1[PB], p2[PB];
    int bin[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int n = tid + u * nth;
      const int nc = n < N ? n : N - 1;
      p0[u] = pp[3 * nc];
      p1[u] = pp[3 * nc + 1];
      p2[u] = pp[3 * nc + 2];
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int n = tid + u * nth;
      float w, v, uu;
      transform_point<QUAT>(ps, p0[u], p1[u], p2[u], w, v, uu);
      const Cell c = locate(w, v, uu, Dz, D);
      bin[u] = c.valid ? c.iz : Dz;
      if (n < N) {
        tp[3 * n] = w;
        tp[3 * n + 1] = v;
        tp[3 * n + 2] = uu;
        atomicAdd(&hist[bin[u]], 1);
      }
    }
    __syncthreads();
    block_exclusive_scan(hist, Dz + 1);
    for (int i = tid; i < Dz + 2; i += nth) zstart[(size_t)b * (Dz + 2) + i] = hist[i];
    write_live_words(hist, Dz, live + (size_t)b * DPC_LIVE_WORDS);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int n = tid + u * nth;
      if (n < N) {
        const int slot = atomicAdd(&hist[bin[u]], 1);
        order[(size_t)b * N + slot] = n;
      }
    }
    return;
  }
  for (int n = tid; n < N; n += nth) {
    float w, v, u;
    transform_point<QUAT>(ps, pp[3 * n], pp[3 * n + 1], pp[3 * n + 2], w, v, u);
    tp[3 * n] = w;
    tp[3 * n + 1] = v;
    tp[3 * n + 2] = u;
    const Cell c = locate(w, v, u, Dz, D);
    atomicAdd(&hist[c.valid ? c.iz : Dz], 1);
  }
  __syncthreads();
  block_exclusive_scan(hist, Dz + 1);
  for (int i = tid; i < Dz + 2; i += nth) zstart[(size_t)b * (Dz + 2) + i] = hist[i];
  write_live_words(hist, Dz, live + (size_t)b * DPC_LIVE_WORDS);
  __syncthreads();
  for (int n = tid; n < N; n += nth) {  // tr_pc rows written above by this same work-group
    const Cell c = locate(tp[3 * n], tp[3 * n + 1], tp[3 * n + 2], Dz, D);
    const int slot = atomicAdd(&hist[c.valid ? c.iz : Dz], 1);
    order[(size_t)b * N + slot] = n;
  }
}

// WG = (view b, plane z, y-strip).  LDS tile = rows [y0-h, y0+SH+h) x D.
template <int KC, int VY>
__global__ void __launch_bounds__(DPC_BLOCK)
k_splat_xy(DpcShape S, const float* __restrict__ tr_pc, const int* __restrict__ order,
           const int* __restrict__ zstart, const float* __restrict__ taps_x,
           const float* __restrict__ taps_y, float* __restrict__ out, unsigned char* __restrict__ cmask,
           int SH, int nstrips, int lr_shift, int skip_empty) {
  DPC_DYN_SMEM(float, tile);
  constexpr int h = KC / 2;
  constexpr int G = zgroup(KC);
  const int D = S.D, Dz = S.Dz, N = S.N;
  const int bid = blockIdx.x;
  const int strip = bid % nstrips;
  const int pz = bid / nstrips;
  const int z = pz % Dz, b = pz / Dz;
  const int y0 = strip * SH;
  const int RT = SH + 2 * h;
  const int tid = threadIdx.x, nth = blockDim.x;

  const int* zs = zstart + (size_t)b * (Dz + 2);
  const int lo = zs[z > 0 ? z - 1 : 0], mid = zs[z], hi = zs[z + 1];
  if (skip_empty && hi == lo) return;  // no mass in this plane: the live-mask consumer never reads it
  const float* tp = tr_pc + (size_t)b * N * 3;

  // 0. sparsity: point clouds are surfaces, most (plane, strip) tiles see no point at all.
  //    Such a tile blurs to exactly zero: store zeros and leave.
  {
    int touched = 0;
    for (int i = lo + tid; i < hi; i += nth) {
      const int n = order[(size_t)b * N + i];
      const Cell c = locate(tp[3 * n], tp[3 * n + 1], tp[3 * n + 2], Dz, D);
      touched |= (c.iy + 1 >= y0 - h) && (c.iy < y0 + SH + h);
    }
    if (!__syncthreads_or(touched)) {
      float* oplane = out + (size_t)pz * D * D + (size_t)y0 * D;
      const int n4 = (SH < D - y0 ? SH : D - y0) * D;
      for (int i = tid * 4; i < n4; i += nth * 4)
        *reinterpret_cast<float4*>(oplane + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
  }

  // 1. zero the tile and the per-row "some point landed here" flags
  int* rowflag = reinterpret_cast<int*>(tile + RT * D);  // [RT]
  for (int i = tid * 4; i < RT * D; i += nth * 4)
    *reinterpret_cast<float4*>(tile + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < RT; i += nth) rowflag[i] = 0;
  __syncthreads();

  // 2. splat the points of depth cells z-1 (upper corner, k=1) and z (k=0)
  for (int i = lo + tid; i < hi; i += nth) {
    const int n = order[(size_t)b * N + i];
    const int k = (i < mid) ? 1 : 0;
    const Cell c = locate(tp[3 * n], tp[3 * n + 1], tp[3 * n + 2], Dz, D);
    const float wz = k ? c.rz : (1.0f - c.rz);
    const float wy[2] = {1.0f - c.ry, c.ry};
    const float wx[2] = {1.0f - c.rx, c.rx};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yy = c.iy + j;
      const int t = yy - (y0 - h);
      if (yy >= D || t < 0 || t >= RT) continue;
      rowflag[t] = 1;  // benign race: every writer stores 1
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int xx = c.ix + l;
        if (xx < D) atomicAdd(&tile[t * D + xx], wz * wy[j] * wx[l]);
      }
    }
  }
  __syncthreads();

  // 3. clip-gradient bits of the corners this strip owns
  for (int i = lo + tid; i < hi; i += nth) {
    const int n = order[(size_t)b * N + i];
    const int k = (i < mid) ? 1 : 0;
    const Cell c = locate(tp[3 * n], tp[3 * n + 1], tp[3 * n + 2], Dz, D);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yy = c.iy + j;
      if (yy >= D || yy < y0 || yy >= y0 + SH) continue;
      const int t = yy - (y0 - h);
      unsigned bits = 0;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int xx = c.ix + l;
        if (xx < D) {
          const float g0 = tile[t * D + xx];
          bits |= (g0 >= 0.f && g0 <= 1.f) ? (1u << l) : 0u;
        }
      }
      cmask[((size_t)b * N + n) * 4 + k * 2 + j] = (unsigned char)bits;
    }
  }
  __syncthreads();

  // 4. clip + x-blur, rows in place (halo from neighbour lanes, as in k_blur_xy_stream)
  const int lane = tid & 63, wave = tid >> 6;
  const int LR = 1 << lr_shift;
  const int PL = 64 >> lr_shift;
  const int lx = lane & (LR - 1);
  const int stream = wave * PL + (lane >> lr_shift);
  const int nstream = (nth >> 6) * PL;
  float tpx[KC];
#pragma unroll
  for (int m = 0; m < KC; ++m) tpx[m] = taps_x[m];
  for (int t0 = 0; t0 < RT; t0 += nstream) {
    const int t = t0 + stream;
    const bool rowok = t < RT;
    if (!__any(rowok && rowflag[rowok ? t : 0])) continue;  // wave-uniform: untouched rows stay zero
    float v[4], xb[4];
    load_cx<4>(tile + (rowok ? t : 0) * D + lx * 4, v);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = rowok ? clampf(v[c], 0.f, 1.f) : 0.f;
    float w[4 + 2 * h];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[h + c] = v[c];
#pragma unroll
    for (int e = 1; e <= h; ++e) {
      const int dl = (e + 3) / 4;
      const int jl = (4 * dl - e) & 3;
      const float vl = __shfl(v[jl], (lane - dl) & 63, 64);
      w[h - e] = (lx - dl >= 0) ? vl : 0.f;
      const int jr = (e - 1) & 3;
      const float vr = __shfl(v[jr], (lane + dl) & 63, 64);
      w[h + 3 + e] = (lx + dl < LR) ? vr : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float a = 0.f;
#pragma unroll
      for (int m = 0; m < KC; ++m) a += tpx[m] * w[o + m];
      xb[o] = a;
    }
    if (rowok) store_cx<4>(tile + t * D + lx * 4, xb);
  }
  __syncthreads();

  // 5. y-blur: each stream produces RS output rows with a register FIR over RS + 2h
  //    tile rows.  VY floats per lane: narrower lanes = fewer, longer streams = less
  //    halo redundancy (RS + 2h pushes per RS outputs).
  const int LRy = D / VY;                 // lanes per row (power of two <= 64)
  const int sy = (wave * 64 + lane) / LRy;  // stream id
  const int ly = lane & (LRy - 1);
  const int nsy = nth / LRy;
  const int RS = SH / nsy;
  const int steps = RS + 2 * h;
  float* oplane = out + (size_t)pz * D * D;
  int live = 0;
  for (int q = 0; q < steps; ++q) live |= rowflag[sy * RS + q];
  if (!live) {  // every input row of this stream is zero => so are its RS output rows
    float zero[VY];
#pragma unroll
    for (int c = 0; c < VY; ++c) zero[c] = 0.f;
    for (int r = 0; r < RS; ++r) {
      const int gy = y0 + sy * RS + r;
      if (gy < D) store_cx<VY>(oplane + (size_t)gy * D + ly * VY, zero);
    }
    return;
  }
  ZFir<KC, VY> fir;
  fir.init(taps_y);
  for (int q0 = 0; q0 < steps; q0 += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int q = q0 + u;
      if (q < steps) {
        float v[VY], o[VY];
        load_cx<VY>(tile + (sy * RS + q) * D + ly * VY, v);
        fir.push(v, o, u);
        const int gy = y0 + sy * RS + q - 2 * h;
        if (q >= 2 * h && gy < D) store_cx<VY>(oplane + (size_t)gy * D + ly * VY, o);
      }
    }
  }
}

// Backward counterpart of k_splat_xy.  WG = (view b, PZ consecutive planes,
// y-strip).  Per plane: stage rows [y0-h, y0+SH+h) of dGz (the z-blurred ray
// gradients) in LDS, y-blur them in place (register FIR, outputs parked in
// registers across a barrier), then, for the points of depth cells z-1 and z
// (same z-bucketed lists as forward), evaluate the x-blur only at the touched
// cells, apply the clip-gradient bits and the trilinear weights, and write one
// [3] partial d(tr_pc) per (corner plane k, corner row j) slot.  The next
// plane's rows are already in flight in registers while the current plane is
// processed.  Replaces the dense y-blur pass (read V + write V) and the
// scattered global gather by one pass that reads dGz once (+ halo).
#ifndef DPC_GATHER_PZ
#define DPC_GATHER_PZ 1   // A/B on MI355X at cfg2 (ms): 1: 0.067, 2: 0.100, 3: 0.095, 4: 0.110, 8: 0.111
#endif
template <int KC, int VY, int RS>
__global__ void __launch_bounds__(DPC_BLOCK)
k_gather_yx(DpcShape S, const float* __restrict__ dgz, const float* __restrict__ tr_pc,
            const int* __restrict__ order, const int* __restrict__ zstart,
            const unsigned char* __restrict__ cmask, const float* __restrict__ taps_x,
            const float* __restrict__ taps_y, float* __restrict__ parts, int SH, int nstrips) {
  DPC_DYN_SMEM(float, tin);  // [RT][D]
  constexpr int h = KC / 2;
  constexpr int NLD = 11;    // 16-byte loads per thread and plane (host guarantees RT*D/4 <= NLD*256)
  const int D = S.D, Dz = S.Dz, N = S.N;
  const int nzg = (Dz + DPC_GATHER_PZ - 1) / DPC_GATHER_PZ;
  const int bid = blockIdx.x;
  const int strip = bid % nstrips;
  const int zg = (bid / nstrips) % nzg;
  const int b = bid / (nstrips * nzg);
  const int y0 = strip * SH;
  const int RT = SH + 2 * h;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int q4 = D >> 2;  // float4 per row
  const int total4 = RT * q4;
  const int lane = tid & 63, wave = tid >> 6;
  const int LRy = D / VY;
  const int sy = (wave * 64 + lane) / LRy;
  const int ly = lane & (LRy - 1);
  const int* zs = zstart + (size_t)b * (Dz + 2);
  const float* tp = tr_pc + (size_t)b * N * 3;
  float tpx[KC + 1];
#pragma unroll
  for (int m = 0; m < KC; ++m) tpx[m] = taps_x[m];

  float pre[NLD][4];
  auto prefetch = [&](int z) {  // rows of plane z -> registers (unconditional, clamped; zeroed at store time)
    const float* plane = dgz + ((size_t)b * Dz + (z < Dz ? z : Dz - 1)) * D * D;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * nth;
      const int ic = i < total4 ? i : total4 - 1;
      const int t = ic / q4, c4 = ic - t * q4;
      const int gy = y0 - h + t;
      const int gyc = gy < 0 ? 0 : (gy >= D ? D - 1 : gy);
      load_cx<4>(plane + (size_t)gyc * D + c4 * 4, pre[u]);
    }
  };
  const int zbeg = zg * DPC_GATHER_PZ;
  // sparsity: a plane whose two depth-cell buckets are empty contributes nothing -- it is
  // neither loaded nor blurred.  (A finer per-strip test needs a scan + barrier per plane,
  // which cost more than it saved: 0.106 -> 0.122 ms at cfg2.)
  unsigned need = 0;
  for (int zi = 0; zi < DPC_GATHER_PZ; ++zi) {
    const int z = zbeg + zi;
    if (z < Dz && zs[z + 1] > zs[z > 0 ? z - 1 : 0]) need |= 1u << zi;
  }
  bool have_pre = false;
  for (int zi = 0; zi < DPC_GATHER_PZ; ++zi) {
    const int z = zbeg + zi;
    if (!((need >> zi) & 1u)) continue;  // uniform
    if (!have_pre) prefetch(z);
    // 1. registers -> LDS (rows outside the grid are zero)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * nth;
      if (i < total4) {
        const int t = i / q4;
        const int gy = y0 - h + t;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (gy >= 0 && gy < D) ? pre[u][c] : 0.f;
        store_cx<4>(tin + i * 4, v);
      }
    }
    __syncthreads();
    have_pre = (zi + 1 < DPC_GATHER_PZ) && ((need >> (zi + 1)) & 1u);
    if (have_pre) prefetch(z + 1);  // in flight during steps 2-3

    // 2. y-blur (adjoint of the forward y-blur: same symmetric taps), outputs in registers
    float outr[RS][VY];
    {
      ZFir<KC, VY> fir;
      fir.init(taps_y);
#pragma unroll
      for (int q = 0; q < RS + 2 * h; ++q) {
        float v[VY], o[VY];
        load_cx<VY>(tin + (sy * RS + q) * D + ly * VY, v);
        fir.push(v, o, q);
        if (q >= 2 * h) {
#pragma unroll
          for (int c = 0; c < VY; ++c) outr[q - 2 * h][c] = o[c];
        }
      }
    }
    __syncthreads();  // every stream has read its rows: overwrite rows [h, h+SH) with the blurred ones
#pragma unroll
    for (int r = 0; r < RS; ++r) store_cx<VY>(tin + (h + sy * RS + r) * D + ly * VY, outr[r]);
    __syncthreads();

    // 3. sparse x-blur + clip bits + trilinear gather for this plane's points
    const int lo = zs[z > 0 ? z - 1 : 0], mid = zs[z], hi = zs[z + 1];
    for (int i = lo + tid; i < hi; i += nth) {
      const int n = order[(size_t)b * N + i];
      const int k = (i < mid) ? 1 : 0;
      const Cell c = locate(tp[3 * n], tp[3 * n + 1], tp[3 * n + 2], Dz, D);
      const float wzk = k ? c.rz : (1.0f - c.rz);
      const float wy[2] = {1.0f - c.ry, c.ry};
      const float wx[2] = {1.0f - c.rx, c.rx};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int yy = c.iy + j;
        if (yy >= D || yy < y0 || yy >= y0 + SH) continue;
        const float* row = tin + (h + yy - y0) * D;
        float g[2] = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m <= KC; ++m) {
          const int x = c.ix - h + m;
          const int xc = x < 0 ? 0 : (x >= D ? D - 1 : x);
          const float val = (x == xc) ? row[xc] : 0.f;
          if (m < KC) g[0] += tpx[m] * val;
          if (m >= 1) g[1] += tpx[m - 1] * val;
        }
        const unsigned bits = cmask[((size_t)b * N + n) * 4 + k * 2 + j];
        float drz = 0.f, dry = 0.f, drx = 0.f;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          if (c.ix + l >= D) continue;
          const float gg = ((bits >> l) & 1u) ? g[l] : 0.f;
          drz += gg * (k ? 1.f : -1.f) * wy[j] * wx[l];
          dry += gg * wzk * (j ? 1.f : -1.f) * wx[l];
          drx += gg * wzk * wy[j] * (l ? 1.f : -1.f);
        }
        float* pp = parts + (((size_t)b * N + n) * 4 + k * 2 + j) * 3;
        pp[0] = drz * (float)(Dz - 1);
        pp[1] = dry * (float)(D - 1);
        pp[2] = drx * (float)(D - 1);
      }
    }
    __syncthreads();  // the tile is refilled by the next plane
  }
}

// plane `t` of a ray bundle, loaded unconditionally from a clamped address
// (so the compiler can keep a whole group of loads in flight); the caller
// zeroes it when t is outside [0,Dz)
template <int CX>
__device__ __forceinline__ void zload(const float* __restrict__ base, int ncol, int t, int Dz, float (&v)[CX]) {
  const int tc = t < 0 ? 0 : (t < Dz ? t : Dz - 1);
  load_cx<CX>(base + (size_t)tc * ncol, v);
}

// plain z blur, compile-time K
template <int KC, int CX>
__global__ void __launch_bounds__(DPC_BLOCK)
k_blur_z(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps, int Dz,
         int D) {
  const int b = blockIdx.y;
  const int ncol = D * D;
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * CX;
  if (col >= ncol) return;
  const size_t base = (size_t)b * Dz * ncol + col;
  constexpr int h = KC / 2;
  constexpr int G = zgroup(KC);
  ZFir<KC, CX> fir;
  fir.init(taps);
  const int T = Dz + h;
  float cur[G][CX], nxt[G][CX];
#pragma unroll
  for (int u = 0; u < G; ++u) zload<CX>(in + base, ncol, u, Dz, cur[u]);
  for (int t0 = 0; t0 < T; t0 += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) zload<CX>(in + base, ncol, t0 + G + u, Dz, nxt[u]);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int t = t0 + u;
      if (t < T) {
        float v[CX], o[CX];
#pragma unroll
        for (int c = 0; c < CX; ++c) v[c] = (t < Dz) ? cur[u][c] : 0.f;
        fir.push(v, o, u);
        if (t >= h) store_cx<CX>(out + base + (size_t)(t - h) * ncol, o);
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int c = 0; c < CX; ++c) cur[u][c] = nxt[u][c];
  }
}

// plain z blur, run-time K (gather form; inputs re-read through L1/L2)
__global__ void __launch_bounds__(DPC_BLOCK)
k_blur_z_generic(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps,
                 int K, int Dz, int D) {
  const int b = blockIdx.y;
  const int ncol = D * D;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  const size_t base = (size_t)b * Dz * ncol + col;
  const int h = K >> 1;
  for (int o = 0; o < Dz; ++o) {
    float a = 0.f;
    for (int m = 0; m < K; ++m) {
      const int z = o + m - h;
      if (z >= 0 && z < Dz) a += taps[m] * in[base + (size_t)z * ncol];
    }
    out[base + (size_t)o * ncol] = a;
  }
}

// ---------------------------------------------------------------------------
// Ray collapse (drc.py:47-123), transcendental-free: with c_j = clip(G3_j, eps,
// 1-eps) the reference's p_j = exp(sum_{i<j} log(1-c_i) + log c_j) is the
// product T_j c_j with T_{j+1} = T_j (1 - c_j) (identical up to fp32 rounding);
// its quirks are kept: the "unity" of the log-space prefix is eps, so
// p_0 = e^eps c_0 and p_Dz = e^eps T_Dz (drc.py:58-59,92-96).
// Forward also leaves two float64 sums per ray for the backward pass:
//   P = sum_{j<Dz} p_j,   Q = sum_{j<=Dz} p_j psi_j.
// ---------------------------------------------------------------------------

// Forward: z-FIR + (scale, clip) + DRC collapse + depth (drc.py:139-153),
// streaming each ray once.
//   in      xy-blurred grid (or raw grid with clip_in when there is no blur)
//   g2_out  post-blur grid G2, saved for backward            (nullable)
//   probs   event probabilities [Dz+1,B,D,D]                 (nullable)
//   proj/depth [B,D,D] (flip_h: image row D-1-y); sums [B,D,D,2] double, row y
template <int KC, int CX>
__global__ void __launch_bounds__(DPC_BLOCK)
k_zfwd(DpcParams P, const float* __restrict__ in, const float* __restrict__ taps,
       const float* __restrict__ scale, float* __restrict__ g2_out, float* __restrict__ probs,
       float* __restrict__ proj, float* __restrict__ depth, double* __restrict__ sums, int B, int Dz,
       int D, int clip_in, int flip_h, const unsigned* __restrict__ live) {
  const int b = blockIdx.y;
  LiveMask lm;
  lm.load(live, b);
  const int ncol = D * D;
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * CX;
  if (col >= ncol) return;
  const size_t base = (size_t)b * Dz * ncol + col;
  const int y = col / D, x0 = col - y * D;
  const int ocol = (flip_h ? (D - 1 - y) : y) * D + x0;
  constexpr int h = KC / 2;
  constexpr int G = zgroup(KC);
  const float eps = P.eps, one_m = 1.0f - P.eps;
  const float e_eps = expf(eps);
  const bool has_s = scale != nullptr;
  const float s = has_s ? scale[b] : 1.0f;
  const float rDz = 1.0f / (float)Dz;  // psi_i = i/Dz - 0.5 + cd (drc.py:139-143); exact for power-of-two Dz
  ZFir<KC, CX> fir;
  fir.init(taps);
  float Tr[CX];
  double Ps[CX], Qs[CX];
#pragma unroll
  for (int c = 0; c < CX; ++c) {
    Tr[c] = 1.0f;
    Ps[c] = 0.0;
    Qs[c] = 0.0;
  }
  const int T = Dz + h;
  // One rolling group of G planes per lane: slot u holds plane t0+u; as soon as it has been consumed
  // the load of plane t0+G+u is issued into the same registers, so G loads stay in flight per lane at
  // all times with a single buffer (the register budget decides how many waves hide the latency).
  static_assert(G <= 32, "one occupancy window per group of planes");
  auto fetch = [&](unsigned win, int u, int t, float (&v)[CX]) {
    // planes without mass were never written by the producer: substitute zeros (uniform branch)
    if (!live || ((win >> u) & 1u)) {
      zload<CX>(in + base, ncol, t, Dz, v);
    } else {
#pragma unroll
      for (int c = 0; c < CX; ++c) v[c] = 0.f;
    }
  };
  auto process = [&](float (&buf)[G][CX], int t0, unsigned win_next) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int t = t0 + u;
      float v[CX];
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        const float r = (t < Dz) ? buf[u][c] : 0.f;
        v[c] = clip_in ? clampf(r, 0.f, 1.f) : r;
      }
      fetch(win_next, u, t + G, buf[u]);
      if (t < T) {
        float g2[CX];
        fir.push(v, g2, u);
        if (t >= h) {
          const int o = t - h;
          if (g2_out) store_cx<CX>(g2_out + base + (size_t)o * ncol, g2);
          const float psi = (float)o * rDz - 0.5f + P.camera_distance;
          float pv[CX];
#pragma unroll
          for (int c = 0; c < CX; ++c) {
            const float g3 = has_s ? clampf(g2[c] * s, 0.f, 1.f) : g2[c];
            const float cc = clampf(g3, eps, one_m);
            const float p = (o == 0 ? e_eps : Tr[c]) * cc;
            pv[c] = p;
            Ps[c] += (double)p;
            Qs[c] += (double)(p * psi);
            Tr[c] *= 1.0f - cc;
          }
          if (probs) store_cx<CX>(probs + ((size_t)o * B + b) * ncol + ocol, pv);
        }
      }
    }
  };
  float buf[G][CX];
  {
    const unsigned win = lm.window(0);
#pragma unroll
    for (int u = 0; u < G; ++u) fetch(win, u, u, buf[u]);
  }
  for (int t0 = 0; t0 < T; t0 += G) process(buf, t0, lm.window(t0 + G));
  float pl[CX], pj[CX], dp[CX];
#pragma unroll
  for (int c = 0; c < CX; ++c) {
    pl[c] = Tr[c] * e_eps;
    Qs[c] += (double)(pl[c] * P.max_depth);
    pj[c] = (float)Ps[c];
    dp[c] = (float)Qs[c];
  }
  if (probs) store_cx<CX>(probs + ((size_t)Dz * B + b) * ncol + ocol, pl);
  if (proj) store_cx<CX>(proj + (size_t)b * ncol + ocol, pj);
  if (depth) store_cx<CX>(depth + (size_t)b * ncol + ocol, dp);
  if (sums) {
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      sums[((size_t)b * ncol + col + c) * 2 + 0] = Ps[c];
      sums[((size_t)b * ncol + col + c) * 2 + 1] = Qs[c];
    }
  }
}

#ifndef DPC_ZBWD_PACKED
#define DPC_ZBWD_PACKED 1   // 0: scalar per-ray arithmetic (A/B baseline)
#endif
// Backward, same walking direction as forward (so T_j and p_j are reproduced
// bit for bit).  With gamma_i = dL/dp_i and a_i = gamma_i p_i:
//   dL/dc_j = gamma_j Tq_j - (sum_{i>j} a_i) / (1 - c_j),   Tq_0 = e^eps, Tq_j = T_j,
// masked by eps <= G3_j <= 1-eps; then the scale/clip mask, dscale, and the z-FIR
// adjoint.  sum_{i>j} a_i = total - sum_{i<=j} a_i with both sums in float64, the
// total coming from forward's saved (P,Q) when gamma_i = g + gd psi_i; a general
// gamma (dprobs != null) or sums == null adds an ascending pre-pass for the total.
template <int KC, int CX>
__global__ void __launch_bounds__(DPC_BLOCK)
k_zbwd(DpcParams P, const float* __restrict__ g2_in, const float* __restrict__ taps,
       const float* __restrict__ scale, const double* __restrict__ sums,
       const float* __restrict__ dproj, const float* __restrict__ ddepth,
       const float* __restrict__ dprobs, float* __restrict__ dgz, float* __restrict__ dscale, int B,
       int Dz, int D, int flip_h, const unsigned* __restrict__ live, float* __restrict__ dsparts,
       float* __restrict__ accum_zero) {
  const int b = blockIdx.y;
  // first kernel of the fused backward: its first work-group per view clears the view's [16]
  // pose accumulator for k_points_bwd (saves a memset launch)
  if (accum_zero && blockIdx.x == 0 && threadIdx.x < 16) accum_zero[16 * (size_t)b + threadIdx.x] = 0.f;
  LiveMask lm;
  lm.load(live, b);
  const int ncol = D * D;
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * CX;
  const bool active = col < ncol;
  float dsacc[1] = {0.f};
  dpc_v2f dsacc2 = dpc_v2f{0.f, 0.f};
  if (active) {
    const size_t base = (size_t)b * Dz * ncol + col;
    const int y = col / D, x0 = col - y * D;
    const int ocol = (flip_h ? (D - 1 - y) : y) * D + x0;
    constexpr int h = KC / 2;
    constexpr int G = zgroup(KC);
    const float eps = P.eps, one_m = 1.0f - P.eps;
    const float e_eps = expf(eps);
    const bool has_s = scale != nullptr;
    const float s = has_s ? scale[b] : 1.0f;
    const float rDz = 1.0f / (float)Dz;  // psi_i = i/Dz - 0.5 + cd (drc.py:139-143); exact for power-of-two Dz
    float g[CX], gd[CX], Tr[CX];
    double tot[CX];
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      g[c] = dproj ? dproj[(size_t)b * ncol + ocol + c] : 0.f;
      gd[c] = ddepth ? ddepth[(size_t)b * ncol + ocol + c] : 0.f;
      Tr[c] = 1.0f;
    }
    if (sums && !dprobs) {
#pragma unroll
      for (int c = 0; c < CX; ++c)
        tot[c] = (double)g[c] * sums[((size_t)b * ncol + col + c) * 2 + 0] +
                 (double)gd[c] * sums[((size_t)b * ncol + col + c) * 2 + 1];
    } else {
      float Tp[CX];
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        tot[c] = 0.0;
        Tp[c] = 1.0f;
      }
      for (int j = 0; j < Dz; ++j) {
        float v[CX];
        load_cx<CX>(g2_in + base + (size_t)j * ncol, v);
        const float psi = (float)j * rDz - 0.5f + P.camera_distance;
#pragma unroll
        for (int c = 0; c < CX; ++c) {
          const float g3 = has_s ? clampf(v[c] * s, 0.f, 1.f) : v[c];
          const float cc = clampf(g3, eps, one_m);
          const float p = (j == 0 ? e_eps : Tp[c]) * cc;
          float gam = g[c] + gd[c] * psi;
          if (dprobs) gam += dprobs[((size_t)j * B + b) * ncol + ocol + c];
          tot[c] += (double)(gam * p);
          Tp[c] *= 1.0f - cc;
        }
      }
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        float gl = gd[c] * P.max_depth;
        if (dprobs) gl += dprobs[((size_t)Dz * B + b) * ncol + ocol + c];
        tot[c] += (double)(gl * (Tp[c] * e_eps));
      }
    }
    ZFir<KC, CX> fir;
    fir.init(taps);
    const int T = Dz + h;
    // rem = total - sum_{i<=j} a_i, kept in float64 (no cancellation error)
    // rolling group of G planes, refilled slot by slot (see k_zfwd)
    auto process = [&](float (&buf)[G][CX], int t0) {
      const unsigned swin = lm.window(t0 - h);   // bit u: output plane t0 + u - h is read by k_gather_yx
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = t0 + u;
        float bv[CX];
#pragma unroll
        for (int c = 0; c < CX; ++c) bv[c] = buf[u][c];
        zload<CX>(g2_in + base, ncol, j + G, Dz, buf[u]);
        if (j < T) {
          float dg2[CX], o[CX];
          if (DPC_ZBWD_PACKED && CX == 2 && j < Dz) {
            // two rays as one 2-vector: every add / mul / fma below is a v_pk_* instruction
            const float psi = (float)j * rDz - 0.5f + P.camera_distance;
            const dpc_v2f vv = dpc_v2f{bv[0], bv[1 % CX]};
            const dpc_v2f sg = vv * dpc_v2f{s, s};
            dpc_v2f g3 = vv;
            if (has_s) g3 = dpc_v2f{clampf(sg[0], 0.f, 1.f), clampf(sg[1], 0.f, 1.f)};
            const dpc_v2f cc = dpc_v2f{clampf(g3[0], eps, one_m), clampf(g3[1], eps, one_m)};
            const dpc_v2f omc = dpc_v2f{1.f, 1.f} - cc;
            const dpc_v2f Tq = (j == 0) ? dpc_v2f{e_eps, e_eps} : dpc_v2f{Tr[0], Tr[1]};
            dpc_v2f gam = dpc_v2f{g[0], g[1]} + dpc_v2f{gd[0], gd[1]} * dpc_v2f{psi, psi};
            if (dprobs) {
              const float* dp = dprobs + ((size_t)j * B + b) * ncol + ocol;
              gam += dpc_v2f{dp[0], dp[1]};
            }
            const dpc_v2f gT = gam * Tq;           // = a_j / c_j
            const dpc_v2f aj = gT * cc;            // a_j = gamma_j p_j
            tot[0] -= (double)aj[0];
            tot[1] -= (double)aj[1];
            const dpc_v2f rem = dpc_v2f{(float)tot[0], (float)tot[1]};
            const dpc_v2f dc = gT - rem * dpc_v2f{dpc_rcp(omc[0]), dpc_rcp(omc[1])};
            // eps <= G3 <= 1-eps  <=>  the clip was inactive
            const dpc_v2f dg3 = dpc_v2f{(cc[0] == g3[0]) ? dc[0] : 0.f, (cc[1] == g3[1]) ? dc[1] : 0.f};
            const dpc_v2f tn = dpc_v2f{Tr[0], Tr[1]} * omc;
            Tr[0] = tn[0];
            Tr[1] = tn[1];
            if (has_s) {
              const dpc_v2f sd = dpc_v2f{s, s} * dg3, vd = vv * dg3;
              const bool m0 = (g3[0] == sg[0]), m1 = (g3[1] == sg[1]);   // 0 <= s G2 <= 1  <=>  clip inactive
              dg2[0] = m0 ? sd[0] : 0.f;
              dg2[1] = m1 ? sd[1] : 0.f;
              dsacc2 += dpc_v2f{m0 ? vd[0] : 0.f, m1 ? vd[1] : 0.f};
            } else {
              dg2[0] = dg3[0];
              dg2[1] = dg3[1];
            }
          } else if (j < Dz) {
            const float psi = (float)j * rDz - 0.5f + P.camera_distance;
#pragma unroll
            for (int c = 0; c < CX; ++c) {
              const float vv = bv[c];
              const float sg = vv * s;
              const float g3 = has_s ? clampf(sg, 0.f, 1.f) : vv;
              const float cc = clampf(g3, eps, one_m);
              const float omc = 1.0f - cc;
              const float Tq = (j == 0) ? e_eps : Tr[c];
              float gam = g[c] + gd[c] * psi;
              if (dprobs) gam += dprobs[((size_t)j * B + b) * ncol + ocol + c];
              const float gT = gam * Tq;           // = a_j / c_j
              tot[c] -= (double)(gT * cc);          // a_j = gamma_j p_j
              const float dc = gT - (float)tot[c] * dpc_rcp(omc);
              const float dg3 = (cc == g3) ? dc : 0.f;   // eps <= G3 <= 1-eps  <=>  the clip was inactive
              Tr[c] *= omc;
              if (has_s) {
                const bool m2 = (g3 == sg);          // 0 <= s G2 <= 1        <=>  the clip was inactive
                dg2[c] = m2 ? s * dg3 : 0.f;
                dsacc[0] += m2 ? vv * dg3 : 0.f;
              } else {
                dg2[c] = dg3;
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < CX; ++c) dg2[c] = 0.f;
          }
          fir.push(dg2, o, u);
          // planes without points are not read by k_gather_yx (same test): leave them unwritten
          if (j >= h && (!live || ((swin >> u) & 1u))) store_cx<CX>(dgz + base + (size_t)(j - h) * ncol, o);
        }
      }
    };
    float buf[G][CX];
#pragma unroll
    for (int u = 0; u < G; ++u) zload<CX>(g2_in + base, ncol, u, Dz, buf[u]);
    for (int t0 = 0; t0 < T; t0 += G) process(buf, t0);
  }
  dsacc[0] += dsacc2[0] + dsacc2[1];
  if (dsparts) {  // uniform across the grid: one partial per work-group, summed in fixed order by k_pose_finalize
    block_reduce_sum<1>(dsacc);
    if (threadIdx.x == 0) dsparts[(size_t)b * gridDim.x + blockIdx.x] = dsacc[0];
  } else if (dscale) {
    block_reduce_sum<1>(dsacc);
    if (threadIdx.x == 0) atomicAdd(dscale + 16 * (size_t)b + 15, dsacc[0]);  // [B,16] accumulator, slot 15
  }
}

// tf.reduce_max over z (point_cloud.py:264-267) and its tie-sharing gradient
__global__ void __launch_bounds__(DPC_BLOCK)
k_max_fwd(const float* __restrict__ vox, const float* __restrict__ scale, float* __restrict__ proj,
          int Dz, int D, int flip_h) {
  const int b = blockIdx.y;
  const int ncol = D * D;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  const size_t base = (size_t)b * Dz * ncol + col;
  const bool has_s = scale != nullptr;
  const float s = has_s ? scale[b] : 1.0f;
  float m = -INFINITY;
  for (int j = 0; j < Dz; ++j) {
    const float v = vox[base + (size_t)j * ncol];
    m = fmaxf(m, has_s ? clampf(v * s, 0.f, 1.f) : v);
  }
  const int y = col / D, x = col - y * D;
  proj[(size_t)b * ncol + (flip_h ? (D - 1 - y) : y) * D + x] = m;
}

__global__ void __launch_bounds__(DPC_BLOCK)
k_max_bwd(const float* __restrict__ vox, const float* __restrict__ scale,
          const float* __restrict__ dproj, float* __restrict__ dvox, float* __restrict__ dscale, int Dz,
          int D, int flip_h) {
  const int b = blockIdx.y;
  const int ncol = D * D;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  float dsacc[1] = {0.f};
  if (col < ncol) {
    const size_t base = (size_t)b * Dz * ncol + col;
    const bool has_s = scale != nullptr;
    const float s = has_s ? scale[b] : 1.0f;
    float m = -INFINITY;
    for (int j = 0; j < Dz; ++j) {
      const float v = vox[base + (size_t)j * ncol];
      m = fmaxf(m, has_s ? clampf(v * s, 0.f, 1.f) : v);
    }
    int cnt = 0;
    for (int j = 0; j < Dz; ++j) {
      const float v = vox[base + (size_t)j * ncol];
      cnt += ((has_s ? clampf(v * s, 0.f, 1.f) : v) == m) ? 1 : 0;
    }
    const int y = col / D, x = col - y * D;
    const float g = dproj[(size_t)b * ncol + (flip_h ? (D - 1 - y) : y) * D + x] / (float)cnt;
    for (int j = 0; j < Dz; ++j) {
      const float v = vox[base + (size_t)j * ncol];
      const float sg = v * s;
      const float g3 = has_s ? clampf(sg, 0.f, 1.f) : v;
      float d = (g3 == m) ? g : 0.f;
      if (has_s) {
        const bool m2 = (sg >= 0.f) && (sg <= 1.f);
        dsacc[0] += m2 ? v * d : 0.f;
        d = m2 ? s * d : 0.f;
      }
      dvox[base + (size_t)j * ncol] = d;
    }
  }
  if (dscale) {
    block_reduce_sum<1>(dsacc);
    if (threadIdx.x == 0) atomicAdd(dscale + 16 * (size_t)b + 15, dsacc[0]);  // [B,16] accumulator, slot 15
  }
}

// streaming copy with W floats per lane: known byte counts for calibrating the
// rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 (MI355X_MICROARCH.md HBM)
template <int W>
__global__ void __launch_bounds__(DPC_BLOCK)
k_copy(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * W;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i + W <= n; i += stride) {
    float v[W];
    load_cx<W>(src + i, v);
    store_cx<W>(dst + i, v);
  }
}

// ===========================================================================
// host side: validation, launch geometry, C ABI
// ===========================================================================
namespace {

int check_shape(const DpcShape* S, bool need_points) {
  if (!S) return DPC_E_NULL;
  if (S->B <= 0 || S->Dz <= 0 || S->D <= 0 || S->B > 65535) return DPC_E_SHAPE;
  if (need_points && (S->N <= 0 || S->N > 65535 * DPC_BLOCK)) return DPC_E_SHAPE;
  if ((long long)S->D * S->D > (1 << 20) || S->Dz > 4096) return DPC_E_SHAPE;
  const int ks[3] = {S->Kx, S->Ky, S->Kz};
  for (int i = 0; i < 3; ++i) {
    if (ks[i] < 0 || ks[i] > DPC_MAX_TAPS) return DPC_E_TAPS;
    if (ks[i] > 0 && (ks[i] % 2) == 0) return DPC_E_TAPS;
  }
  return DPC_OK;
}

inline size_t grid_elems(const DpcShape& S) { return (size_t)S.B * S.Dz * S.D * S.D; }
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

inline int last_error() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DPC_OK : (int)e;
}

inline dim3 point_grid(const DpcShape& S) { return dim3(S.B, (S.N + DPC_BLOCK - 1) / DPC_BLOCK, 1); }

// ---- plane blur launch -------------------------------------------------------
// streaming (LDS-free) plane blur: D a power of two in [16,256], K in {5,11,21}
bool launch_blur_xy_stream(hipStream_t st, const DpcShape& S, const float* in, float* out, const float* tx,
                           const float* ty, int Kx, int Ky, int clip_in, int* rc) {
  const int D = S.D;
  if (D < 16 || D > 256 || (D & (D - 1)) != 0) return false;
  if (!(Kx == 0 || Ky == 0 || Kx == Ky)) return false;
  const int K = Kx > 0 ? Kx : Ky;
  if (K != 5 && K != 11 && K != 21) return false;
  int lr_shift = 0;
  while ((4 << lr_shift) < D) ++lr_shift;
  const int PL = 64 >> lr_shift;
  const long long nplanes = (long long)S.B * S.Dz;
  const long long nwaves = (nplanes + PL - 1) / PL;
  const long long nblk = (nwaves + (DPC_BLOCK / 64) - 1) / (DPC_BLOCK / 64);
  if (nblk > 0x7fffffffLL) return false;
  const dim3 grid((unsigned)nblk, 1, 1), block(DPC_BLOCK, 1, 1);
#define DPC_XY(KC, X, Y)                                                                             \
  DPC_LAUNCH("blur_xy", (k_blur_xy_stream<KC, X, Y>), grid, block, 0, st, in, out, tx, ty, (int)nplanes, D, \
             lr_shift, clip_in)
#define DPC_XY_K(KC)                          \
  do {                                        \
    if (Kx > 0 && Ky > 0) DPC_XY(KC, true, true);   \
    else if (Kx > 0) DPC_XY(KC, true, false); \
    else DPC_XY(KC, false, true);             \
  } while (0)
  if (K == 5) DPC_XY_K(5);
  else if (K == 11) DPC_XY_K(11);
  else DPC_XY_K(21);
#undef DPC_XY_K
#undef DPC_XY
  *rc = last_error();
  return true;
}

int launch_blur_plane(hipStream_t st, const DpcShape& S, const float* in, float* out, const float* tx,
                      const float* ty, int Kx, int Ky, int clip_in) {
  {
    int rc = DPC_OK;
    if (launch_blur_xy_stream(st, S, in, out, tx, ty, Kx, Ky, clip_in, &rc)) return rc;
  }
  const int D = S.D;
  const int hx = Kx / 2, hy = Ky / 2;
  const int nxc = (D + DPC_XC - 1) / DPC_XC;
  int PA = nxc * DPC_XC + 2 * hx;
  if ((PA & 1) == 0) PA += 1;
  const int PB = D | 1;
  int TY = 32;
  size_t bytes = 0;
  for (;;) {
    const int R = TY + 2 * hy;
    bytes = sizeof(float) * ((size_t)R * PA + (Kx > 0 ? (size_t)R * PB : 0));
    if (bytes <= 60 * 1024 || TY == DPC_YC) break;
    TY >>= 1;
  }
  if (bytes > 64 * 1024) return DPC_E_SHAPE;
  const int nyt = (D + TY - 1) / TY;
  const long long nblocks = (long long)S.B * S.Dz * nyt;
  if (nblocks > 0x7fffffffLL) return DPC_E_SHAPE;
  const dim3 grid((unsigned)nblocks, 1, 1), block(DPC_BLOCK, 1, 1);
  const bool fixed_ok = (Kx == 0 || Ky == 0 || Kx == Ky);
  const int K = Kx > 0 ? Kx : Ky;
#define DPC_PLANE_CASE(KC)                                                                          \
  DPC_LAUNCH("blur_plane", (k_blur_plane<KC>), grid, block, bytes, st, in, out, tx, ty, Kx, Ky, S.Dz, D, TY, nyt, \
             PA, PB, clip_in, (int)nblocks)
  if (fixed_ok && K == 5) {
    DPC_PLANE_CASE(5);
  } else if (fixed_ok && K == 11) {
    DPC_PLANE_CASE(11);
  } else if (fixed_ok && K == 21) {
    DPC_PLANE_CASE(21);
  } else {
    DPC_PLANE_CASE(0);
  }
#undef DPC_PLANE_CASE
  return last_error();
}

#ifndef DPC_CX_PREF
#define DPC_CX_PREF 2
#endif
inline int pick_cx(int D) {
  if (DPC_CX_PREF >= 4 && D % 4 == 0) return 4;
  if (DPC_CX_PREF >= 2 && D % 2 == 0) return 2;
  return 1;
}
inline dim3 col_grid(const DpcShape& S, int cx) {
  const int nthr = (S.D * S.D + cx - 1) / cx;
  return dim3((nthr + DPC_BLOCK - 1) / DPC_BLOCK, S.B, 1);
}
inline bool z_fixed(int K) { return K == 0 || K == 3 || K == 5 || K == 7 || K == 9 || K == 11 || K == 21; }

#define DPC_Z_CASES(CXV, MACRO)   \
  switch (k_) {                    \
    case 1: MACRO(1, CXV); break;  \
    case 3: MACRO(3, CXV); break;  \
    case 5: MACRO(5, CXV); break;  \
    case 7: MACRO(7, CXV); break;  \
    case 9: MACRO(9, CXV); break;  \
    case 11: MACRO(11, CXV); break; \
    case 21: MACRO(21, CXV); break; \
  }
#if DPC_CX_PREF >= 4
#define DPC_Z_CASES4(MACRO) DPC_Z_CASES(4, MACRO)
#else
#define DPC_Z_CASES4(MACRO)
#endif
#define DPC_Z_DISPATCH(K, CX, MACRO)      \
  do {                                    \
    const int k_ = (K) == 0 ? 1 : (K);    \
    if (CX == 4) {                        \
      DPC_Z_CASES4(MACRO)                 \
    } else if (CX == 2) {                 \
      DPC_Z_CASES(2, MACRO)               \
    } else {                              \
      DPC_Z_CASES(1, MACRO)               \
    }                                     \
  } while (0)

int launch_blur_z(hipStream_t st, const DpcShape& S, const float* in, float* out, const float* tz, int Kz) {
  const dim3 block(DPC_BLOCK, 1, 1);
  if (z_fixed(Kz) && Kz > 0) {
    const int cx = pick_cx(S.D);
    const dim3 grid = col_grid(S, cx);
#define DPC_M(KC, CXV) DPC_LAUNCH("blur_z", (k_blur_z<KC, CXV>), grid, block, 0, st, in, out, tz, S.Dz, S.D)
    DPC_Z_DISPATCH(Kz, cx, DPC_M);
#undef DPC_M
  } else {
    DPC_LAUNCH("blur_z_generic", (k_blur_z_generic), col_grid(S, 1), block, 0, st, in, out, tz, Kz, S.Dz, S.D);
  }
  return last_error();
}

// in -> (z-FIR Kz) -> collapse.  Kz must be z_fixed().
int launch_zfwd(hipStream_t st, const DpcShape& S, const DpcParams& P, const float* in, const float* tz,
                int Kz, const float* scale, float* g2_out, float* probs, float* proj, float* depth,
                double* sums, int clip_in, int flip_h, const unsigned* live = nullptr) {
  const dim3 block(DPC_BLOCK, 1, 1);
  const int cx = pick_cx(S.D);
  const dim3 grid = col_grid(S, cx);
#define DPC_M(KC, CXV)                                                                                \
  DPC_LAUNCH("zfwd", (k_zfwd<KC, CXV>), grid, block, 0, st, P, in, (Kz > 0 ? tz : (const float*)nullptr), scale, \
             g2_out, probs, proj, depth, sums, S.B, S.Dz, S.D, clip_in, flip_h, live)
  DPC_Z_DISPATCH(Kz, cx, DPC_M);
#undef DPC_M
  return last_error();
}

int launch_zbwd(hipStream_t st, const DpcShape& S, const DpcParams& P, const float* g2, const float* tz,
                int Kz, const float* scale, const double* sums, const float* dproj, const float* ddepth,
                const float* dprobs, float* dgz, float* dscale, int flip_h, const unsigned* live = nullptr,
                float* dsparts = nullptr, float* accum_zero = nullptr) {
  const dim3 block(DPC_BLOCK, 1, 1);
  const int cx = pick_cx(S.D);
  const dim3 grid = col_grid(S, cx);
#define DPC_M(KC, CXV)                                                                                \
  DPC_LAUNCH("zbwd", (k_zbwd<KC, CXV>), grid, block, 0, st, P, g2, (Kz > 0 ? tz : (const float*)nullptr), scale, \
             sums, dproj, ddepth, dprobs, dgz, dscale, S.B, S.Dz, S.D, flip_h, live, dsparts, accum_zero)
  DPC_Z_DISPATCH(Kz, cx, DPC_M);
#undef DPC_M
  return last_error();
}

int launch_points_bwd(hipStream_t st, const DpcShape& S, const DpcParams& P, const float* pc,
                      const float* pose, const float* trans, const float* focal, const float* tr_pc,
                      const float* dgrid, const float* mask, const unsigned char* cmask, const float* taps_x,
                      const float* dtr_in, const float* parts, bool gather, float* dpc, float* dpose,
                      float* dtrans, float* dfocal, float* dscale, float* accum, bool zero_accum,
                      const float* dsparts = nullptr, int nzb = 0) {
  if (zero_accum) {
    hipError_t e = dpc_memset("memset_small", accum, sizeof(float) * 16 * (size_t)S.B, st);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid = point_grid(S), block(DPC_BLOCK, 1, 1);
  const bool quat = P.pose_is_quaternion != 0;
#define DPC_PB(Q, G, KC)                                                                              \
  DPC_LAUNCH("points_bwd", (k_points_bwd<Q, G, KC>), grid, block, 0, st, S, P, pc, pose, trans, focal, tr_pc, dgrid, \
             mask, cmask, taps_x, dtr_in, parts, dpc, accum)
  const int kc = (gather && (S.Kx == 5 || S.Kx == 11 || S.Kx == 21)) ? S.Kx : 0;
  if (quat && gather) {
    if (kc == 11) DPC_PB(true, true, 11);
    else if (kc == 21) DPC_PB(true, true, 21);
    else if (kc == 5) DPC_PB(true, true, 5);
    else DPC_PB(true, true, 0);
  } else if (quat) {
    DPC_PB(true, false, 0);
  } else if (gather) {
    DPC_PB(false, true, 0);
  } else {
    DPC_PB(false, false, 0);
  }
#undef DPC_PB
  const dim3 fg((S.B + 63) / 64, 1, 1), fb(64, 1, 1);
  if (quat)
    DPC_LAUNCH("pose_finalize", (k_pose_finalize<true>), fg, fb, 0, st, S, P, pose, accum, dpose, dtrans, dfocal,
               dscale, dsparts, nzb);
  else
    DPC_LAUNCH("pose_finalize", (k_pose_finalize<false>), fg, fb, 0, st, S, P, pose, accum, dpose, dtrans, dfocal,
               dscale, dsparts, nzb);
  return last_error();
}

// ---- fused front end (k_zsort + k_splat_xy) -------------------------------------
struct SplatPlan {
  bool ok;
  int SH, nstrips, lr_shift, vy;
  size_t lds_bytes;
  int gSH, gRS, gstrips;  // k_gather_yx strips (gSH == 0: not applicable)
  size_t glds_bytes;
};
SplatPlan splat_plan(const DpcShape& S) {
  SplatPlan p = {false, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int D = S.D, K = S.Kx;
  if (S.Kx != S.Ky || (K != 5 && K != 11 && K != 21)) return p;
  if (D < 32 || D > 256 || (D & (D - 1)) != 0 || S.N <= 0) return p;
  int lr_shift = 0;
  while ((4 << lr_shift) < D) ++lr_shift;
  const int nstream = (DPC_BLOCK / 64) * (64 >> lr_shift);
#ifndef DPC_SPLAT_LDS_KB
#define DPC_SPLAT_LDS_KB 48
#endif
  int SH = D;
  while (SH >= nstream && sizeof(float) * (size_t)(SH + 2 * (K / 2)) * D > DPC_SPLAT_LDS_KB * 1024) SH >>= 1;
  if (SH < nstream || SH % nstream != 0) return p;
  p.vy = (D <= 128) ? 2 : 4;              // y-phase floats per lane (k_splat_xy step 5)
  const int nsy = DPC_BLOCK / (D / p.vy);
  if (nsy < 1 || SH % nsy != 0) return p;
  // k_gather_yx: one LDS tile of gSH + 2h rows, gRS = gSH / nsy rows per y-stream in {16, 8},
  // at most 11 16-byte loads per thread and plane; otherwise backward uses the generic kernels
  p.gSH = 0;
#ifndef DPC_GATHER_RS_MAX
#define DPC_GATHER_RS_MAX 16
#endif
  for (int rs = DPC_GATHER_RS_MAX; rs >= 8; rs >>= 1) {
    const int g = rs * nsy;
    if (g > D) continue;
    const size_t rows = (size_t)g + 2 * (K / 2);
    if (sizeof(float) * rows * D <= 48 * 1024 && rows * (D / 4) <= 11 * DPC_BLOCK) {
      p.gSH = g;
      p.gRS = rs;
      break;
    }
  }
  if (p.gSH > 0) {
    p.gstrips = D / p.gSH;
    p.glds_bytes = sizeof(float) * (size_t)(p.gSH + 2 * (K / 2)) * D;
  }
  p.ok = true;
  p.SH = SH;
  p.nstrips = D / SH;
  p.lr_shift = lr_shift;
  p.lds_bytes = sizeof(float) * (size_t)(SH + 2 * (K / 2)) * (D + 1);  // tile + per-row flags
  return p;
}
inline size_t point_index_ints(const DpcShape& S) {
  return (size_t)S.B * S.N + (size_t)S.B * (S.Dz + 2) + (size_t)S.B * DPC_LIVE_WORDS;
}
inline size_t parts_bytes(const DpcShape& S) { return align256(sizeof(float) * 12 * (size_t)S.B * S.N); }
// per-work-group dscale partials of k_zbwd: [B, work-groups per view]
inline int zbwd_blocks(const DpcShape& S) { return (int)col_grid(S, pick_cx(S.D)).x; }
inline size_t dsparts_bytes(const DpcShape& S) { return align256(sizeof(float) * (size_t)S.B * zbwd_blocks(S)); }

int launch_gather_yx(hipStream_t st, const DpcShape& S, const SplatPlan& pl, const float* dgz, const float* tr_pc,
                     const int* order, const int* zstart, const unsigned char* cmask, const float* tx,
                     const float* ty, float* parts) {
  const int nzg = (S.Dz + DPC_GATHER_PZ - 1) / DPC_GATHER_PZ;
  const long long nblk = (long long)S.B * nzg * pl.gstrips;
  if (nblk > 0x7fffffffLL) return DPC_E_SHAPE;
  const dim3 grid((unsigned)nblk, 1, 1), block(DPC_BLOCK, 1, 1);
#define DPC_GY(KC, VY, RS)                                                                                   \
  DPC_LAUNCH("gather_yx", (k_gather_yx<KC, VY, RS>), grid, block, pl.glds_bytes, st, S, dgz, tr_pc, order,  \
             zstart, cmask, tx, ty, parts, pl.gSH, pl.gstrips)
#define DPC_GYV(KC)                                      \
  do {                                                   \
    if (pl.vy == 2 && pl.gRS == 16) DPC_GY(KC, 2, 16);   \
    else if (pl.vy == 2) DPC_GY(KC, 2, 8);               \
    else if (pl.gRS == 16) DPC_GY(KC, 4, 16);            \
    else DPC_GY(KC, 4, 8);                               \
  } while (0)
  if (S.Kx == 5) DPC_GYV(5);
  else if (S.Kx == 11) DPC_GYV(11);
  else DPC_GYV(21);
#undef DPC_GYV
#undef DPC_GY
  return last_error();
}

int launch_splat_xy(hipStream_t st, const DpcShape& S, const DpcParams& P, const SplatPlan& pl, const float* pc,
                    const float* pose, const float* trans, const float* focal, float* tr_pc, int* order,
                    int* zstart, const float* tx, const float* ty, float* out, unsigned char* cmask,
                    unsigned* live, bool skip_empty) {
  int zt = 64;
  while (zt < 1024 && zt < S.N) zt <<= 1;
  if (P.pose_is_quaternion)
    DPC_LAUNCH("zsort", (k_zsort<true>), dim3(S.B, 1, 1), dim3(zt, 1, 1), sizeof(int) * (size_t)(S.Dz + 2), st, S, P,
               pc, pose, trans, focal, tr_pc, order, zstart, live);
  else
    DPC_LAUNCH("zsort", (k_zsort<false>), dim3(S.B, 1, 1), dim3(zt, 1, 1), sizeof(int) * (size_t)(S.Dz + 2), st, S, P,
               pc, pose, trans, focal, tr_pc, order, zstart, live);
  const long long nblk = (long long)S.B * S.Dz * pl.nstrips;
  if (nblk > 0x7fffffffLL) return DPC_E_SHAPE;
  const dim3 grid((unsigned)nblk, 1, 1), block(DPC_BLOCK, 1, 1);
#define DPC_SP(KC, VY)                                                                                     \
  DPC_LAUNCH("splat_xy", (k_splat_xy<KC, VY>), grid, block, pl.lds_bytes, st, S, tr_pc, (const int*)order, \
             (const int*)zstart, tx, ty, out, cmask, pl.SH, pl.nstrips, pl.lr_shift, skip_empty ? 1 : 0)
#define DPC_SPV(KC)                \
  do {                             \
    if (pl.vy == 2) DPC_SP(KC, 2); \
    else DPC_SP(KC, 4);            \
  } while (0)
  if (S.Kx == 5) DPC_SPV(5);
  else if (S.Kx == 11) DPC_SPV(11);
  else DPC_SPV(21);
#undef DPC_SPV
#undef DPC_SP
  return last_error();
}

}  // namespace

// ===========================================================================
// Silhouette loss epilogue (reference model_pc.py:308-337 proj_loss_pose_candidates,
// :383-423 add_proj_loss): bilinear GT-mask resize on the fly, per-instance squared
// error, arg-min over the C pose candidates of each (model, view) group, masked L2.
// Three tiny launches over [B,D,D] images; proj is 0.1 % of the path's HBM traffic.
// ===========================================================================
// tf.image.resize_images(BILINEAR) of TF1: src = dst * (in/out), no half-pixel
// centres, hi index clamped; S == D reads the pixel itself.
__device__ __forceinline__ float gt_sample(const float* __restrict__ gt, int S, int D, float ratio, int y, int x) {
  if (S == D) return gt[y * S + x];
  const float sy = (float)y * ratio, sx = (float)x * ratio;
  const int y0 = (int)sy, x0 = (int)sx;  // sy, sx >= 0: truncation == floor
  const int y1 = y0 + 1 < S ? y0 + 1 : S - 1, x1 = x0 + 1 < S ? x0 + 1 : S - 1;
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float tl = gt[y0 * S + x0], tr = gt[y0 * S + x1], bl = gt[y1 * S + x0], br = gt[y1 * S + x1];
  const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
  return top + (bot - top) * ly;
}

// one work-group per instance b = g*C + c: inst_err[b] = sum (gt_g - proj_b)^2
__global__ void __launch_bounds__(DPC_BLOCK) k_sil_err(const float* __restrict__ proj, const float* __restrict__ gt,
                                                       float* __restrict__ inst_err, int C, int D, int S) {
  const int b = blockIdx.x, g = b / C;
  const float* pb = proj + (size_t)b * D * D;
  const float* gg = gt + (size_t)g * S * S;
  const float ratio = (float)S / (float)D;
  float acc[1] = {0.f};
  for (int i = threadIdx.x; i < D * D; i += DPC_BLOCK) {
    const int y = i / D, x = i - y * D;
    const float d = gt_sample(gg, S, D, ratio, y, x) - pb[i];
    acc[0] = fmaf(d, d, acc[0]);
  }
  block_reduce_sum<1>(acc);
  if (threadIdx.x == 0) inst_err[b] = acc[0];
}

// single work-group: winners[g] = argmin_c inst_err[g,c] (first minimum, as tf.argmin),
// weight[b] = [c == winner] * valid_g, loss = sum_g valid_g^2 * err[g,win] / (2 G)
__global__ void __launch_bounds__(DPC_BLOCK) k_sil_select(const float* __restrict__ inst_err,
                                                          const float* __restrict__ valid, int* __restrict__ winners,
                                                          float* __restrict__ weight, float* __restrict__ loss, int G,
                                                          int C) {
  float acc[1] = {0.f};
  for (int g = threadIdx.x; g < G; g += DPC_BLOCK) {
    int win = 0;
    float best = inst_err[(size_t)g * C];
    for (int c = 1; c < C; ++c) {
      const float e = inst_err[(size_t)g * C + c];
      if (e < best) {
        best = e;
        win = c;
      }
    }
    const float w = (valid && C > 1) ? valid[g] : 1.f;  // the C == 1 branch of add_proj_loss ignores valid_samples
    for (int c = 0; c < C; ++c) weight[(size_t)g * C + c] = (c == win) ? w : 0.f;
    if (winners) winners[g] = win;
    acc[0] += w * w * best;
  }
  block_reduce_sum<1>(acc);
  if (threadIdx.x == 0) loss[0] = acc[0] * 0.5f / (float)G;
}

// dproj[b] = dloss * weight_b^2 / G * (proj_b - gt_g)
__global__ void __launch_bounds__(DPC_BLOCK) k_sil_grad(const float* __restrict__ proj, const float* __restrict__ gt,
                                                        const float* __restrict__ weight,
                                                        const float* __restrict__ dloss, float* __restrict__ dproj,
                                                        int C, int D, int S, float inv_G) {
  const int b = blockIdx.y, g = b / C;
  const int i = blockIdx.x * DPC_BLOCK + threadIdx.x;
  if (i >= D * D) return;
  const float w = weight[b];
  float out = 0.f;
  if (w != 0.f) {
    const int y = i / D, x = i - y * D;
    const float d = proj[(size_t)b * D * D + i] - gt_sample(gt + (size_t)g * S * S, S, D, (float)S / (float)D, y, x);
    out = dloss[0] * w * w * inv_G * d;
  }
  dproj[(size_t)b * D * D + i] = out;
}

// ===========================================================================
// Nearest-neighbour distance (reference util/point_cloud_distance.py:26-39, the Chamfer
// evaluation of run/eval_chamfer.py:18-34): for every source point the closest target
// point, in the tensor's own precision (fp64 in the evaluation).  Brute force, targets
// streamed through LDS; a work-group is 256/PH sources x PH target phases (phase p takes every
// PH-th target), merged lexicographically on (distance, index) so the result is tf.argmin's
// first minimum of sqrt(sum diff^2) exactly.  PH = 16 for small source sets (an evaluation has
// ~8000 predicted points: 64 sources per work-group would leave half the chip idle).
// ===========================================================================
#define DPC_NN_CHUNK 1024
template <typename T, int PH>
__global__ void __launch_bounds__(256) k_nn_distance(const T* __restrict__ vs, const T* __restrict__ vt, int ns, int nt,
                                                     T* __restrict__ proj, T* __restrict__ min_dist,
                                                     int* __restrict__ idx) {
  __shared__ T tile[DPC_NN_CHUNK * 3];
  __shared__ T m_s[256];
  __shared__ int m_i[256];
  constexpr int SRC = 256 / PH;
  const int tid = threadIdx.x, lane = tid % SRC, phase = tid / SRC;
  const int s = blockIdx.x * SRC + lane;
  const int sc = s < ns ? s : ns - 1;
  const T sx = vs[(size_t)sc * 3], sy = vs[(size_t)sc * 3 + 1], sz = vs[(size_t)sc * 3 + 2];
  T best_d2 = (T)INFINITY, best_s = (T)INFINITY;
  int best_i = 0;
  for (int base = 0; base < nt; base += DPC_NN_CHUNK) {
    const int cnt = nt - base < DPC_NN_CHUNK ? nt - base : DPC_NN_CHUNK;
    __syncthreads();
    for (int i = tid; i < cnt * 3; i += 256) tile[i] = vt[(size_t)base * 3 + i];
    __syncthreads();
    for (int j = phase; j < cnt; j += PH) {
#pragma clang fp contract(off)
      // (vt - vs)^2 summed x, y, z in this order without FMA contraction, as tf.reduce_sum(diff**2, axis=2) does
      const T dx = tile[j * 3] - sx, dy = tile[j * 3 + 1] - sy, dz = tile[j * 3 + 2] - sz;
      const T d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 < best_d2) {          // sqrt is monotone: only a smaller d2 can give a smaller distance
        best_d2 = d2;
        const T sd = sqrt(d2);
        if (sd < best_s) {         // equal after rounding: the earlier index stays (first minimum)
          best_s = sd;
          best_i = base + j;
        }
      }
    }
  }
  m_s[tid] = best_s;
  m_i[tid] = best_i;
  __syncthreads();
  if (phase == 0 && s < ns) {
    for (int w = 1; w < PH; ++w) {
      const T os = m_s[w * SRC + lane];
      const int oi = m_i[w * SRC + lane];
      if (os < best_s || (os == best_s && oi < best_i)) {
        best_s = os;
        best_i = oi;
      }
    }
    min_dist[s] = best_s;
    idx[s] = best_i;
    proj[(size_t)s * 3] = vt[(size_t)best_i * 3];
    proj[(size_t)s * 3 + 1] = vt[(size_t)best_i * 3 + 1];
    proj[(size_t)s * 3 + 2] = vt[(size_t)best_i * 3 + 2];
  }
}

// ===========================================================================
// Exact Gaussian voxeliser (reference pointcloud2voxels, point_cloud.py:17-57, the
// pc_fast:false path of pointcloud_project :219-226): every point adds
// exp(-|p - g|^2 / 2 sigma^2) to EVERY node g of a G^3 lattice spanning [-1,1]^3.
// O(N G^3) per view, so it is a cross-check / debugging path, not the training path.
// The Gaussian factorises, out[a0,a1,a2] = sum_n f0[n,a0] f1[n,a1] f2[n,a2], which is
// what both kernels use: 3 exp per (point, node line) instead of one per (point, node).
//   f_a[n,i] = exp(-(c_a[n] - r_i)^2 / 2 sigma^2) * inv_norm_a[n],   r_i = -1 + 2i/(G-1)
// inv_norm: 1 (no normalisation / analytical constant folded into `scale`) or
// 1 / sum_i exp(..) per axis (cfg.pc_normalise_gauss: the reference's sum over the
// whole lattice is the product of the three per-axis sums).
// Output axis a takes point component perm[a].
// ===========================================================================
#define DPC_GV_TILE 64   // (a1, a2) tile edge of the forward kernel
#define DPC_GV_NC 64     // points per LDS chunk
__device__ __forceinline__ float gv_node(int i, float step) { return -1.f + (float)i * step; }

// per-point inverse normalisers (cfg.pc_normalise_gauss): inv_norm[b,n,a] = 1 / sum_i exp(-(c_a - r_i)^2 k)
__global__ void __launch_bounds__(DPC_BLOCK) k_gv_norm(const float* __restrict__ pc, float* __restrict__ inv_norm,
                                                       int total, int G, float k, float step) {
  const int i = blockIdx.x * DPC_BLOCK + threadIdx.x;
  if (i >= total) return;
  const float c = pc[i];
  float s = 0.f;
  for (int j = 0; j < G; ++j) {
    const float d = c - gv_node(j, step);
    s += expf(-d * d * k);
  }
  inv_norm[i] = 1.f / s;
}

// grid (B, G, tiles): one work-group = one a0 slice x a 64x64 (a1,a2) tile, each thread a 4x4 register block.
__global__ void __launch_bounds__(256) k_gv_fwd(const float* __restrict__ pc, const float* __restrict__ inv_norm,
                                                float* __restrict__ raw, float* __restrict__ vox, int N, int G,
                                                int p0, int p1, int p2, float k, float scale, float step) {
  __shared__ float F0[DPC_GV_NC];
  __shared__ __attribute__((aligned(16))) float F1[DPC_GV_NC][DPC_GV_TILE];
  __shared__ __attribute__((aligned(16))) float F2[DPC_GV_NC][DPC_GV_TILE];
  const int b = blockIdx.x, a0 = blockIdx.y;
  const int nt = (G + DPC_GV_TILE - 1) / DPC_GV_TILE;
  const int t1 = (blockIdx.z / nt) * DPC_GV_TILE, t2 = (blockIdx.z % nt) * DPC_GV_TILE;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const float r0 = gv_node(a0, step);
  const float* pb = pc + (size_t)b * N * 3;
  const float* nb = inv_norm ? inv_norm + (size_t)b * N * 3 : nullptr;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int base = 0; base < N; base += DPC_GV_NC) {
    __syncthreads();
    if (tid < DPC_GV_NC) {
      const int n = base + tid;
      float f = 0.f;
      if (n < N) {
        const float d = pb[n * 3 + p0] - r0;
        f = expf(-d * d * k) * scale * (nb ? nb[n * 3 + p0] : 1.f);
      }
      F0[tid] = f;
    }
    __syncthreads();
    for (int q = tid; q < DPC_GV_NC * 2 * DPC_GV_TILE; q += 256) {
      const int n = q / (2 * DPC_GV_TILE), r = q % (2 * DPC_GV_TILE);
      const int second = r >= DPC_GV_TILE, col = r & (DPC_GV_TILE - 1);
      const int node = (second ? t2 : t1) + col, comp = second ? p2 : p1;
      float f = 0.f;
      if (base + n < N && node < G) {
        const float d = pb[(base + n) * 3 + comp] - gv_node(node, step);
        f = expf(-d * d * k) * (nb ? nb[(base + n) * 3 + comp] : 1.f);
      }
      if (second) F2[n][col] = f;
      else F1[n][col] = f * F0[n];
    }
    __syncthreads();
#pragma unroll 4
    for (int n = 0; n < DPC_GV_NC; ++n) {
      const float4 u = *(const float4*)&F1[n][ty * 4];
      const float4 v = *(const float4*)&F2[n][tx * 4];
      const float uu[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(uu[i], vv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int a1 = t1 + ty * 4 + i;
    if (a1 >= G) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a2 = t2 + tx * 4 + j;
      if (a2 >= G) continue;
      const size_t o = (((size_t)b * G + a0) * G + a1) * G + a2;
      raw[o] = acc[i][j];
      vox[o] = clampf(acc[i][j], 0.f, 1.f);
    }
  }
}

// gm = dvox * [0 <= raw <= 1] (closed interval, as everywhere); zero the read-ahead pad behind the tensor
__global__ void __launch_bounds__(DPC_BLOCK) k_gv_mask(const float* __restrict__ raw, const float* __restrict__ dvox,
                                                       float* __restrict__ gm, size_t total, int pad) {
  const size_t i = (size_t)blockIdx.x * DPC_BLOCK + threadIdx.x;
  if (i < total) {
    const float r = raw[i];
    gm[i] = (r >= 0.f && r <= 1.f) ? dvox[i] : 0.f;
  } else if (i < total + pad) {
    gm[i] = 0.f;
  }
}

// one thread per point, one wave per work-group: the masked gradient rows are wave-uniform reads.
//   d c_a = scale * sum g * df_a * f_b * f_c,   df_a[i] = f_a[i] * ((r_i - c_a)/sigma^2 - q_a)
// q_a = S'_a / S_a under per-point normalisation (quotient rule), else 0.
#define DPC_GV_KC 32   // a2 nodes held in registers at a time
__global__ void __launch_bounds__(64) k_gv_bwd(const float* __restrict__ pc, const float* __restrict__ gm,
                                               float* __restrict__ dpc, int N, int G, int p0, int p1, int p2, float k,
                                               float scale, float step, int normalise) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 64 + threadIdx.x;
  const int nc = n < N ? n : N - 1;
  const float* pp = pc + ((size_t)b * N + nc) * 3;
  const float c0 = pp[p0], c1 = pp[p1], c2 = pp[p2];
  const float is2 = 2.f * k;  // 1 / sigma^2
  float inv0 = 1.f, inv1 = 1.f, inv2 = 1.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
  if (normalise) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f;
    for (int i = 0; i < G; ++i) {
      const float r = gv_node(i, step);
      const float x0 = expf(-(c0 - r) * (c0 - r) * k), x1 = expf(-(c1 - r) * (c1 - r) * k),
                  x2 = expf(-(c2 - r) * (c2 - r) * k);
      s0 += x0; s1 += x1; s2 += x2;
      e0 = fmaf(x0, (r - c0) * is2, e0); e1 = fmaf(x1, (r - c1) * is2, e1); e2 = fmaf(x2, (r - c2) * is2, e2);
    }
    inv0 = 1.f / s0; inv1 = 1.f / s1; inv2 = 1.f / s2;
    q0 = e0 * inv0; q1 = e1 * inv1; q2 = e2 * inv2;
  }
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  const float* gb = gm + (size_t)b * G * G * G;
  for (int kb = 0; kb < G; kb += DPC_GV_KC) {
    float f2[DPC_GV_KC], g2[DPC_GV_KC];
#pragma unroll
    for (int j = 0; j < DPC_GV_KC; ++j) {
      const float r = gv_node(kb + j, step);
      const float f = (kb + j < G) ? expf(-(c2 - r) * (c2 - r) * k) * inv2 : 0.f;
      f2[j] = f;
      g2[j] = f * ((r - c2) * is2 - q2);
    }
    for (int a0 = 0; a0 < G; ++a0) {
      const float r0 = gv_node(a0, step);
      const float f0 = expf(-(c0 - r0) * (c0 - r0) * k) * inv0;
      const float g0 = f0 * ((r0 - c0) * is2 - q0);
      float s = 0.f, s1 = 0.f, s2 = 0.f;
      for (int a1 = 0; a1 < G; ++a1) {
        const float* row = gb + ((size_t)a0 * G + a1) * G + kb;   // wave-uniform address
        float u = 0.f, u2 = 0.f;
#pragma unroll
        for (int j = 0; j < DPC_GV_KC; ++j) {
          const float g = row[j];
          u = fmaf(f2[j], g, u);
          u2 = fmaf(g2[j], g, u2);
        }
        const float r1 = gv_node(a1, step);
        const float f1 = expf(-(c1 - r1) * (c1 - r1) * k) * inv1;
        const float g1 = f1 * ((r1 - c1) * is2 - q1);
        s = fmaf(f1, u, s);
        s1 = fmaf(g1, u, s1);
        s2 = fmaf(f1, u2, s2);
      }
      d0 = fmaf(g0, s, d0);
      d1 = fmaf(f0, s1, d1);
      d2 = fmaf(f0, s2, d2);
    }
  }
  if (n < N) {
    float* o = dpc + ((size_t)b * N + n) * 3;
    o[p0] = d0 * scale;
    o[p1] = d1 * scale;
    o[p2] = d2 * scale;
  }
}

extern "C" {

const char* dpc_version(void) { return "dpc_hip 0.1.0 (gfx950)"; }

int dpc_profile_enable(int on) {
  dpcprof::clear();
  std::lock_guard<std::mutex> lk(dpcprof::g_mu);
  dpcprof::g_on = on != 0;
  return DPC_OK;
}

int dpc_saved_layout(const DpcShape* shape, const DpcParams* params) {
  if (check_shape(shape, true) != DPC_OK || !params) return DPC_E_SHAPE;
  return splat_plan(*shape).ok ? 6 : 1;  // bit 0: grid_raw; bits 1+2: clip_mask + point_index
}

size_t dpc_point_index_ints(const DpcShape* shape) {
  if (check_shape(shape, true) != DPC_OK || !splat_plan(*shape).ok) return 0;
  return point_index_ints(*shape);
}

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
  else
    return DPC_E_MODE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DPC_OK : (int)e;
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
  return direction == 0 ? g : 2 * g + acc + parts_bytes(*shape) + dsparts_bytes(*shape);
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
                           nullptr, nullptr, nullptr, dtr_pc, nullptr, false, dpc, dpose, dtrans, dfocal, nullptr,
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
  if (plane && zed) {
    if (order == 0) {
      rc = launch_blur_plane(st, S, in, tmp, taps_x, taps_y, S.Kx, S.Ky, 0);
      if (rc) return rc;
      return launch_blur_z(st, S, tmp, out, taps_z, S.Kz);
    }
    rc = launch_blur_z(st, S, in, tmp, taps_z, S.Kz);
    if (rc) return rc;
    return launch_blur_plane(st, S, tmp, out, taps_x, taps_y, S.Kx, S.Ky, 0);
  }
  if (plane) return launch_blur_plane(st, S, in, out, taps_x, taps_y, S.Kx, S.Ky, 0);
  return launch_blur_z(st, S, in, out, taps_z, S.Kz);
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
  const SplatPlan plan = splat_plan(S);
  if (plan.ok ? (!clip_mask || !point_index) : !grid_raw) return DPC_E_NULL;
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
  if (plan.ok) {
    // 1+2 fused front end: transform + z-bucket (one WG per view) -> per-plane LDS splat + clip + x,y blur
    int* order = (int*)point_index;                  // [B,N]   points sorted by depth cell
    int* zstart = order + (size_t)S.B * S.N;         // [B,Dz+2] bucket starts
    live = (unsigned*)(zstart + (size_t)S.B * (S.Dz + 2));   // [B,8] plane-occupancy bits
    // planes without mass are skipped end to end when the consumer is the mask-aware k_zfwd
    rc = launch_splat_xy(st, S, P, plan, pc, pose, trans, focal, tr_pc, order, zstart, taps_x, taps_y, tmp,
                         clip_mask, live, drc && z_fixed(S.Kz));
    if (rc) return rc;
    zin = tmp;
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
    return launch_zfwd(st, S, P, zin, taps_z, S.Kz, scale, grid_blur, nullptr, proj, proj_depth, ray_sums,
                       clip_in, 1, live);
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
  if (!dproj && !(drc && dproj_depth)) return DPC_E_NULL;
  if (scale && !dscale) return DPC_E_NULL;
  const SplatPlan plan = splat_plan(S);
  const bool use_cmask = plan.ok;
  if (use_cmask ? (!clip_mask || !point_index) : !grid_raw) return DPC_E_NULL;
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
  const int nzb = zbwd_blocks(S);
  // 1. collapse VJP (+ z-FIR adjoint) -> tA
  const bool yx = use_cmask && plan.gSH > 0;   // consumer of tA is k_gather_yx (reads occupied planes only)
  if (zfused) {
    const unsigned* live =
        yx ? (const unsigned*)(point_index + (size_t)S.B * S.N + (size_t)S.B * (S.Dz + 2)) : nullptr;
    rc = launch_zbwd(st, S, P, grid_blur, taps_z, S.Kz, scale, ray_sums, dproj, dproj_depth, nullptr, tA,
                     ds_acc, 1, live, scale ? dsparts : nullptr, accum);
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
      rc = launch_blur_z(st, S, tB, tA, taps_z, S.Kz);
      if (rc) return rc;
    }
  }
  if (yx) {
    // 2+3 fused: per-plane LDS pass (y-blur + sparse x-blur + clip bits + trilinear gather),
    // then the camera-transform VJP over the per-corner partials
    const int* order = (const int*)point_index;
    const int* zstart = order + (size_t)S.B * S.N;
    rc = launch_gather_yx(st, S, plan, tA, tr_pc, order, zstart, clip_mask, taps_x, taps_y, parts);
    if (rc) return rc;
    return launch_points_bwd(st, S, P, pc, pose, trans, focal, tr_pc, nullptr, nullptr, nullptr, nullptr,
                             dtr_pc_in, parts, false, dpc, dpose, dtrans, dfocal, scale ? dscale : nullptr,
                             accum, false, (zfused && scale) ? dsparts : nullptr, nzb);
  }
  // 2. y-blur adjoint (dense) -> tB ; the x-blur is evaluated sparsely in step 3
  const float* dg = tA;
  if (S.Ky > 0) {
    rc = launch_blur_plane(st, S, tA, tB, nullptr, taps_y, 0, S.Ky, 0);
    if (rc) return rc;
    dg = tB;
  }
  // 3. sparse x-blur + clip mask + gather + transform VJP + reductions
  return launch_points_bwd(st, S, P, pc, pose, trans, focal, tr_pc, dg, use_cmask ? nullptr : grid_raw,
                           use_cmask ? clip_mask : nullptr, taps_x, dtr_pc_in, nullptr, true, dpc, dpose, dtrans,
                           dfocal, scale ? dscale : nullptr, accum, false, (zfused && scale) ? dsparts : nullptr, nzb);
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
