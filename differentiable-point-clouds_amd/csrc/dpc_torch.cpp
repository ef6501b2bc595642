// dpc_torch.cpp -- compiled PyTorch binding of the fused projector: the thin layer SURVEY.md 8(b) plans between
// torch tensors and the C ABI (include/dpc_hip.h).  It does exactly what ops.ProjectFused does through ctypes --
// argument checks, the per-shape plan (which buffers the library wants saved, their offsets in ONE arena), output
// allocation, the current HIP stream, the two C-ABI calls, autograd bookkeeping -- as one C++ autograd node, because a
// launch-bound caller (BASELINE configs[0]: 4 views of 1000 points, 65 us of device work per step) was paying ~180 us of
// Python per step for it.  No arithmetic happens here: every number comes out of libdpc_hip.so, which this file reaches
// through dlopen()/dlsym() on the path Python hands it (so it always drives the library the ctypes loader holds --
// the product library, an A/B build, or the CPU emulation build of the test tier with host pointers).
//
// Built by __graft_entry__.build() with torch.utils.cpp_extension (host compiler only) into csrc/build_torch/.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "dpc_hip.h"

namespace {

using at::Tensor;
using c10::optional;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Api {
  std::string path;
  bool host_memory = false;
  decltype(&dpc_project_forward) forward = nullptr;
  decltype(&dpc_project_backward) backward = nullptr;
  decltype(&dpc_saved_layout) saved_layout = nullptr;
  decltype(&dpc_workspace_bytes) workspace_bytes = nullptr;
  decltype(&dpc_point_index_ints) point_index_ints = nullptr;
  decltype(&dpc_sil_parts_per_view) sil_parts = nullptr;
  decltype(&dpc_silhouette_select) sil_select = nullptr;
  decltype(&dpc_abi_struct_bytes) abi_struct_bytes = nullptr;
};
std::mutex g_mu;
std::vector<Api> g_apis;
bool g_dry_run = false;     // dev switch (scripts/host_path_cpu.py): skip the compute calls, time the host path alone

template <class F>
void sym(void* h, const char* name, F& f, const std::string& path) {
  f = reinterpret_cast<F>(dlsym(h, name));
  TORCH_CHECK(f != nullptr, path, ": symbol ", name, " not found");
}

// -> id of the library (an index into g_apis); the same path opened twice gives the same id
int64_t open_library(const std::string& path, bool host_memory) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < g_apis.size(); ++i)
    if (g_apis[i].path == path) return (int64_t)i;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  TORCH_CHECK(h != nullptr, "dlopen(", path, "): ", dlerror());
  Api a;
  a.path = path;
  a.host_memory = host_memory;
  sym(h, "dpc_project_forward", a.forward, path);
  sym(h, "dpc_project_backward", a.backward, path);
  sym(h, "dpc_saved_layout", a.saved_layout, path);
  sym(h, "dpc_workspace_bytes", a.workspace_bytes, path);
  sym(h, "dpc_point_index_ints", a.point_index_ints, path);
  sym(h, "dpc_sil_parts_per_view", a.sil_parts, path);
  sym(h, "dpc_silhouette_select", a.sil_select, path);
  sym(h, "dpc_abi_struct_bytes", a.abi_struct_bytes, path);
  TORCH_CHECK(a.abi_struct_bytes(0) == sizeof(DpcShape) && a.abi_struct_bytes(1) == sizeof(DpcParams), path,
              ": DpcShape / DpcParams are laid out differently in the library and in this binding (rebuild both)");
  g_apis.push_back(a);
  return (int64_t)g_apis.size() - 1;
}

// a copy of the library's entry points (taken under the lock: open_library may grow the table meanwhile)
Api api_of(int64_t lib) {
  std::lock_guard<std::mutex> lk(g_mu);
  TORCH_CHECK(lib >= 0 && (size_t)lib < g_apis.size(), "unknown library id ", lib);
  return g_apis[(size_t)lib];
}

inline size_t a256(size_t n) { return (n + 255) & ~(size_t)255; }

// Everything about one (library, B, N, grid, taps, collapse) combination that does not change from call to call
struct Plan {
  DpcShape shape;
  int layout = 0;
  size_t ws_fwd = 0, ws_bwd = 0;
  bool drc = false;
  int64_t off_raw = -1, off_cmask = -1, off_pindex = -1, off_blur = 0, off_sums = -1, arena_bytes = 0;
  int64_t sil_parts = 0;
};
std::unordered_map<std::string, Plan> g_plans;

// (returned by value: the cache may be cleared by another thread)
Plan plan_for(int64_t lib, const Api& api, int B, int N, int Dz, int D, int collapse, int Kx, int Ky, int Kz) {
  char key[160];
  snprintf(key, sizeof key, "%ld/%d/%d/%d/%d/%d/%d/%d/%d", (long)lib, B, N, Dz, D, collapse, Kx, Ky, Kz);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return it->second;
  if (g_plans.size() > 256) g_plans.clear();
  Plan p;
  p.shape = DpcShape{B, N, Dz, D, Kx, Ky, Kz};
  DpcParams q;
  memset(&q, 0, sizeof q);
  q.camera_distance = 2.0f;
  q.focal_length = 1.875f;
  q.eps = 1e-5f;
  q.max_depth = 10.0f;
  q.pose_is_quaternion = 1;
  q.collapse_mode = collapse;
  p.layout = api.saved_layout(&p.shape, &q);
  TORCH_CHECK_VALUE(p.layout >= 0, "dpc_saved_layout: unsupported sizes (B=", B, ", N=", N, ", grid ", Dz, "x", D, "x", D, ", taps ",
                    Kx, "/", Ky, "/", Kz, ")");
  p.ws_fwd = api.workspace_bytes(&p.shape, 0);
  p.ws_bwd = api.workspace_bytes(&p.shape, 1);
  p.drc = collapse == DPC_COLLAPSE_DRC;
  const size_t grid = (size_t)4 * B * Dz * D * D;
  size_t off = 0;
  if (p.layout & 1) { p.off_raw = (int64_t)off; off += a256(grid); }
  if (p.layout & 2) { p.off_cmask = (int64_t)off; off += a256((size_t)4 * B * N); }
  if (p.layout & 4) { p.off_pindex = (int64_t)off; off += a256(4 * api.point_index_ints(&p.shape)); }
  p.off_blur = (int64_t)off;
  off += a256(grid);
  if (p.drc) { p.off_sums = (int64_t)off; off += a256((size_t)16 * B * D * D); }
  p.arena_bytes = (int64_t)off + 256;
  p.sil_parts = p.drc ? (int64_t)api.sil_parts(&p.shape) : 0;
  return g_plans.emplace(key, p).first->second;
}

inline void* ptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void* ptr(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }
inline Tensor opt(const optional<Tensor>& t) { return t.has_value() ? *t : Tensor(); }
inline char* at_off(char* base, int64_t off) { return off < 0 ? nullptr : base + off; }

#define DPC_CALL(expr) (g_dry_run ? 0 : (expr))
void check_rc(int rc, const char* what) {
  // (the Python side turns the prefix back into dpc_amd.DpcError with the library's own message)
  TORCH_CHECK(rc == 0, "DPC_RC:", rc, ":", what);
}

dpc_stream_t stream_of(const Api& api, const Tensor& ref) {
  if (api.host_memory) return nullptr;
  return (dpc_stream_t)c10::hip::getCurrentHIPStream(ref.device().index()).stream();
}

void check_tensor(const Api& api, const Tensor& t, const char* name, const Tensor& ref, at::ScalarType dt = at::kFloat) {
  if (!t.defined()) return;
  TORCH_CHECK_TYPE(t.scalar_type() == dt, "the projector computes in float32; got ", t.scalar_type(), " for ", name);
  if (api.host_memory) {
    TORCH_CHECK_VALUE(!t.is_cuda(), "emulation library needs host tensors");
  } else {
    TORCH_CHECK_VALUE(t.is_cuda(), "the HIP projector needs tensors on a ROCm device (no CPU fallback)");
  }
  TORCH_CHECK_VALUE(t.device() == ref.device(), "tensors on different devices: ", t.device(), " and ", ref.device());
}

struct Meta {
  int64_t lib, Dz, D;
  double cd, f, eps, max_depth;
  bool quat;
  int64_t collapse, dropout_keep, dropout_seed;
  double l2_weight;
  int64_t views_per_cloud, sil_C;
  int64_t poison;    // 0 off, 1: 0xff-fill what the kernels must define, 2: ... and 0x00 over the point index / chunk marks
};

DpcParams params_of(const Meta& m, const Tensor& dropout_state) {
  DpcParams p;
  memset(&p, 0, sizeof p);
  p.camera_distance = (float)m.cd;
  p.focal_length = (float)m.f;
  p.eps = (float)m.eps;
  p.max_depth = (float)m.max_depth;
  p.pose_is_quaternion = m.quat ? 1 : 0;
  p.collapse_mode = (int32_t)m.collapse;
  p.dropout_keep = (int32_t)m.dropout_keep;
  p.dropout_seed = (uint32_t)m.dropout_seed;
  p.dropout_state = (const int32_t*)ptr(dropout_state);
  p.l2_weight = (float)m.l2_weight;
  p.views_per_cloud = (int32_t)m.views_per_cloud;
  return p;
}

Tensor contig(const Tensor& t) { return (!t.defined() || t.is_contiguous()) ? t : t.contiguous(); }
Tensor flat_taps(const Tensor& t) { return (!t.defined() || (t.dim() == 1 && t.is_contiguous())) ? t : t.contiguous().reshape({-1}); }

class ProjectFusedFn : public torch::autograd::Function<ProjectFusedFn> {
 public:
  // -> the DEFINED ones of (proj, tr_pc, proj_depth, l2_grad, sil_loss, sil_winners, sil_inst_err), in that order (a custom
  // function cannot return undefined tensors): proj_depth iff the DRC collapse, l2_grad iff an L2 target, sil_* iff masks
  static variable_list forward(AutogradContext* ctx, const Tensor& pc_in, const Tensor& pose_in, const optional<Tensor>& trans_in,
                               const optional<Tensor>& scale_in, const optional<Tensor>& focal_in, const optional<Tensor>& tx_in,
                               const optional<Tensor>& ty_in, const optional<Tensor>& tz_in, const optional<Tensor>& dropout_state_in,
                               const optional<Tensor>& l2_target_in, const optional<Tensor>& sil_gt_in,
                               const optional<Tensor>& sil_valid_in, int64_t lib, int64_t Dz, int64_t D, double cd, double f, double eps,
                               double max_depth, bool quat, int64_t collapse, int64_t dropout_keep, int64_t dropout_seed,
                               double l2_weight, int64_t views_per_cloud, int64_t sil_C, int64_t poison) {
    const Api api = api_of(lib);
    const Meta m{lib, Dz, D, cd, f, eps, max_depth, quat, collapse, dropout_keep, dropout_seed, l2_weight, views_per_cloud, sil_C, poison};
    TORCH_CHECK_VALUE(pc_in.dim() == 3 && pc_in.size(2) == 3, "point_cloud must be [B,N,3], got ", pc_in.sizes());
    const int64_t R = views_per_cloud > 1 ? views_per_cloud : 1;
    const int64_t B = pc_in.size(0) * R, N = pc_in.size(1);
    if (quat) {
      TORCH_CHECK_VALUE(pose_in.dim() == 2 && pose_in.size(0) == B && pose_in.size(1) == 4,
                        "Can't create a quaternion from a tensor with shape ", pose_in.sizes(), ". The last dimension must be 4.");
    } else {
      TORCH_CHECK_VALUE(pose_in.dim() == 3 && pose_in.size(0) == B && pose_in.size(1) == 4 && pose_in.size(2) == 4,
                        "camera matrix must be [B,4,4], got ", pose_in.sizes());
      TORCH_CHECK_VALUE(!trans_in.has_value(), "predicted_translation requires a quaternion pose (the reference's matrix branch "
                                               "cannot slice it)");
    }
    Tensor pc = contig(pc_in), pose = contig(pose_in), trans = contig(opt(trans_in)), scale = contig(opt(scale_in)),
           focal = contig(opt(focal_in));
    Tensor tx = flat_taps(opt(tx_in)), ty = flat_taps(opt(ty_in)), tz = flat_taps(opt(tz_in));
    Tensor dstate = opt(dropout_state_in), tgt = opt(l2_target_in), sgt = opt(sil_gt_in), sval = opt(sil_valid_in);
    if (trans.defined()) TORCH_CHECK_VALUE(trans.dim() == 2 && trans.size(0) == B && trans.size(1) == 3, "predicted_translation must be [B,3]");
    if (scale.defined()) TORCH_CHECK_VALUE(scale.numel() == B, "scaling_factor must have B=", B, " elements, got ", scale.sizes());
    if (focal.defined()) TORCH_CHECK_VALUE(focal.numel() == B, "focal_length must have B=", B, " elements, got ", focal.sizes());
    check_tensor(api, pc, "point_cloud", pc);
    check_tensor(api, pose, "transform", pc);
    check_tensor(api, trans, "predicted_translation", pc);
    check_tensor(api, scale, "scaling_factor", pc);
    check_tensor(api, focal, "focal_length", pc);
    int K[3] = {0, 0, 0};
    const Tensor* taps[3] = {&tx, &ty, &tz};
    for (int i = 0; i < 3; ++i) {
      if (!taps[i]->defined()) continue;
      check_tensor(api, *taps[i], "kernel", pc);
      K[i] = (int)taps[i]->numel();
      TORCH_CHECK_VALUE(K[i] % 2 == 1, "even Gaussian kernel sizes are not supported (TF pads them asymmetrically)");
      TORCH_CHECK_VALUE(K[i] <= DPC_MAX_TAPS, "kernel size ", K[i], " > ", DPC_MAX_TAPS);
    }
    const Plan plan = plan_for(lib, api, (int)B, (int)N, (int)Dz, (int)D, (int)collapse, K[0], K[1], K[2]);
    const bool fused = (plan.layout & 2) != 0;
    TORCH_CHECK_VALUE(R == 1 || fused, "views_per_cloud needs the fused path (vox_size a multiple of 4 in (16, 256], odd kernel size 3..31)");
    if (tgt.defined()) {
      TORCH_CHECK_VALUE(plan.drc, "the fused L2 epilogue lives in the DRC collapse kernel (ptn_max_projection is off it)");
      check_tensor(api, tgt, "l2 target", pc);
      TORCH_CHECK_VALUE(tgt.numel() == B * D * D && tgt.dim() >= 3 && tgt.size(0) == B && tgt.size(1) == D && tgt.size(2) == D &&
                            tgt.is_contiguous(),
                        "l2 target must be a contiguous [B,D,D] or [B,D,D,1] image, got ", tgt.sizes(), " for B=", B, ", D=", D);
    }
    if (dstate.defined())
      TORCH_CHECK_VALUE(dstate.scalar_type() == at::kInt && dstate.numel() == 2 && dstate.is_contiguous() && dstate.device() == pc.device(),
                        "dropout state must be a contiguous int32 tensor {keep, seed} on the points' device");
    TORCH_CHECK_VALUE(fused || !((dropout_keep > 0 && dropout_keep < N) || dstate.defined()),
                      "fused point dropout needs the fused path (vox_size a multiple of 4 in (16, 256], odd kernel size 3..31, vox_size_z <= 256); use pc_point_dropout for this shape");
    if (sgt.defined()) {
      TORCH_CHECK_VALUE(plan.sil_parts > 0, "the fused candidate-loss epilogue needs the fused path with the DRC collapse");
      TORCH_CHECK_VALUE(sil_C > 0 && B % sil_C == 0, "B=", B, " instances do not split into groups of ", sil_C, " pose candidates");
      TORCH_CHECK_VALUE((sgt.dim() == 3 || sgt.dim() == 4) && sgt.size(0) == B / sil_C && sgt.size(1) == sgt.size(2) && sgt.size(1) >= D &&
                            sgt.is_contiguous() && sgt.scalar_type() == at::kFloat && sgt.device() == pc.device(),
                        "silhouette masks must be contiguous float32 [", B / sil_C, ",S,S(,1)] with S >= ", D, " on the points' device, got ",
                        sgt.sizes());
      if (sval.defined())
        TORCH_CHECK_VALUE(sval.numel() == B / sil_C && sval.scalar_type() == at::kFloat && sval.is_contiguous() && sval.device() == pc.device(),
                          "valid_samples must be ", B / sil_C, " contiguous float32 values on the points' device");
    }
    const auto fopt = pc.options().dtype(at::kFloat);
    const auto bopt = pc.options().dtype(at::kByte);
    const float nan = std::numeric_limits<float>::quiet_NaN();
    auto new_img = [&]() {
      Tensor t = at::empty({B, D, D, 1}, fopt);
      if (poison) t.fill_(nan);
      return t;
    };
    // independent image tensors (in-place edits of an output are legal, holding one does not pin the others)
    Tensor proj = new_img(), depth = plan.drc ? new_img() : Tensor(), l2_grad = tgt.defined() ? new_img() : Tensor();
    Tensor tr_pc = at::empty({B, N, 3}, fopt);
    Tensor arena = at::empty({plan.arena_bytes}, bopt);          // everything saved for backward: one allocation, raw offsets
    Tensor work = at::empty({(int64_t)plan.ws_fwd + 256}, bopt);
    if (poison) {
      tr_pc.fill_(nan);
      arena.fill_(255);
      work.fill_(255);
    }
    char* base = (char*)a256((size_t)arena.data_ptr());
    if (poison == 2 && plan.off_pindex >= 0)     // marks nobody wrote must read "empty", not "everything marked"
      arena.narrow(0, (int64_t)(base - (char*)arena.data_ptr()) + plan.off_pindex, plan.off_blur - plan.off_pindex).zero_();
    DpcParams params = params_of(m, dstate);
    if (l2_grad.defined()) {
      params.l2_target = (const float*)tgt.data_ptr();
      params.l2_grad = (float*)l2_grad.data_ptr();
    }
    Tensor err_parts;
    if (sgt.defined()) {
      err_parts = at::empty({B, plan.sil_parts}, fopt);
      params.sil_gt = (const float*)sgt.data_ptr();
      params.sil_err_parts = (float*)err_parts.data_ptr();
      params.sil_C = (int32_t)sil_C;
      params.sil_S = (int32_t)sgt.size(1);
    }
    dpc_stream_t st = stream_of(api, pc);
    check_rc(DPC_CALL(api.forward(st, &plan.shape, &params, (const float*)pc.data_ptr(), (const float*)pose.data_ptr(), (const float*)ptr(trans),
                         (const float*)ptr(scale), (const float*)ptr(focal), (const float*)ptr(tx), (const float*)ptr(ty),
                         (const float*)ptr(tz), (float*)tr_pc.data_ptr(), (float*)at_off(base, plan.off_raw),
                         (unsigned char*)at_off(base, plan.off_cmask), (int32_t*)at_off(base, plan.off_pindex),
                         (float*)(base + plan.off_blur), (double*)at_off(base, plan.off_sums), (float*)proj.data_ptr(),
                         (float*)ptr(depth), (void*)a256((size_t)work.data_ptr()), plan.ws_fwd)),
             "dpc_project_forward");
    Tensor sil_loss, sil_win, sil_err, sil_w;
    if (sgt.defined()) {
      sil_err = at::empty({B}, fopt);
      sil_w = at::empty({B}, fopt);
      Tensor win32 = at::empty({B / sil_C}, pc.options().dtype(at::kInt));
      sil_loss = at::empty({}, fopt);
      check_rc(DPC_CALL(api.sil_select(st, (int)B, (int)sil_C, (int)plan.sil_parts, (const float*)err_parts.data_ptr(), (const float*)ptr(sval),
                              (float*)sil_err.data_ptr(), (int32_t*)win32.data_ptr(), (float*)sil_w.data_ptr(),
                              (float*)sil_loss.data_ptr())),
               "dpc_silhouette_select");
      sil_win = win32.to(at::kLong);
    }
    ctx->set_materialize_grads(false);       // unused outputs (depth, tr_pc) arrive undefined, not as zero fills
    ctx->save_for_backward({pc, pose, trans, scale, focal, tx, ty, tz, tr_pc, arena, sgt, sil_w, sgt.defined() ? proj : Tensor(), dstate});
    // (two packed entries instead of twenty string-keyed ones: this runs every step)
    ctx->saved_data["i"] = std::vector<int64_t>{lib, Dz, D, quat ? 1 : 0, collapse, dropout_keep, dropout_seed, views_per_cloud, sil_C,
                                                poison, K[0], K[1], K[2], depth.defined() ? 1 : 0, l2_grad.defined() ? 1 : 0};
    ctx->saved_data["d"] = std::vector<double>{cd, f, eps, max_depth, l2_weight};
    if (scale.defined()) ctx->saved_data["scale_shape"] = scale_in->sizes().vec();
    if (focal.defined()) ctx->saved_data["focal_shape"] = focal_in->sizes().vec();
    variable_list nd;
    if (l2_grad.defined()) nd.push_back(l2_grad);
    if (sgt.defined()) {
      nd.push_back(sil_win);
      nd.push_back(sil_err);
    }
    if (!nd.empty()) ctx->mark_non_differentiable(nd);
    variable_list out = {proj, tr_pc};
    if (depth.defined()) out.push_back(depth);
    if (l2_grad.defined()) out.push_back(l2_grad);
    if (sgt.defined()) {
      out.push_back(sil_loss);
      out.push_back(sil_win);
      out.push_back(sil_err);
    }
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &pc = saved[0], &pose = saved[1], &trans = saved[2], &scale = saved[3], &focal = saved[4], &tx = saved[5],
                 &ty = saved[6], &tz = saved[7], &tr_pc = saved[8], &arena = saved[9], &sgt = saved[10], &sil_w = saved[11],
                 &sil_proj = saved[12], &dstate = saved[13];
    auto& sd = ctx->saved_data;
    const std::vector<int64_t> iv = sd["i"].toIntVector();
    const std::vector<double> dv = sd["d"].toDoubleVector();
    const Meta m{iv[0], iv[1], iv[2], dv[0], dv[1], dv[2], dv[3], iv[3] != 0, iv[4], iv[5], iv[6], dv[4], iv[7], iv[8], iv[9]};
    const Api api = api_of(m.lib);
    const int64_t R = m.views_per_cloud > 1 ? m.views_per_cloud : 1;
    const int64_t B = pc.size(0) * R, N = pc.size(1);
    const Plan plan = plan_for(m.lib, api, (int)B, (int)N, (int)m.Dz, (int)m.D, (int)m.collapse, (int)iv[10], (int)iv[11], (int)iv[12]);
    size_t gi = 2;
    Tensor dproj = contig(g[0]), dtr = contig(g[1]);
    Tensor ddepth = iv[13] ? contig(g[gi++]) : Tensor();
    if (iv[14]) ++gi;
    Tensor dsil = sgt.defined() ? g[gi] : Tensor();
    DpcParams params = params_of(m, dstate);
    const bool use_sil = sgt.defined() && dsil.defined();
    if (use_sil) {
      // the candidate loss's gradient w.r.t. proj is formed inside the collapse VJP from (proj, masks, weights)
      dsil = contig(dsil.to(at::kFloat));
      params.sil_gt = (const float*)sgt.data_ptr();
      params.sil_weight = (const float*)sil_w.data_ptr();
      params.sil_dloss = (const float*)dsil.data_ptr();
      params.sil_proj = (const float*)sil_proj.data_ptr();
      params.sil_C = (int32_t)m.sil_C;
      params.sil_S = (int32_t)sgt.size(1);
    }
    const auto fopt = pc.options().dtype(at::kFloat);
    if (!dproj.defined() && !ddepth.defined() && !use_sil) dproj = at::zeros({B, m.D, m.D, 1}, fopt);
    const float nan = std::numeric_limits<float>::quiet_NaN();
    auto fresh = [&](at::IntArrayRef s) {
      Tensor t = at::empty(s, fopt);
      if (m.poison) t.fill_(nan);
      return t;
    };
    Tensor dpc = fresh({B / R, N, 3});           // per cloud: the kernels sum a cloud's R instances
    Tensor dpose = at::empty_like(pose);
    if (m.poison) dpose.fill_(nan);
    Tensor dtrans = trans.defined() ? fresh({B, 3}) : Tensor();
    Tensor dscale = scale.defined() ? fresh({B}) : Tensor();
    // the matrix branch never reads the per-instance focal length (point_cloud.py:191-205): no gradient
    Tensor dfocal = (focal.defined() && m.quat) ? fresh({B}) : Tensor();
    Tensor work = at::empty({(int64_t)plan.ws_bwd + 256}, pc.options().dtype(at::kByte));
    if (m.poison) work.fill_(255);
    char* base = (char*)a256((size_t)arena.data_ptr());
    check_rc(DPC_CALL(api.backward(stream_of(api, pc), &plan.shape, &params, (const float*)pc.data_ptr(), (const float*)pose.data_ptr(),
                          (const float*)ptr(trans), (const float*)ptr(scale), (const float*)ptr(focal), (const float*)ptr(tx),
                          (const float*)ptr(ty), (const float*)ptr(tz), (const float*)tr_pc.data_ptr(),
                          (const float*)at_off(base, plan.off_raw), (const unsigned char*)at_off(base, plan.off_cmask),
                          (const int32_t*)at_off(base, plan.off_pindex), (const float*)(base + plan.off_blur),
                          (const double*)at_off(base, plan.off_sums), (const float*)ptr(dproj), (const float*)ptr(ddepth),
                          (const float*)ptr(dtr), (float*)dpc.data_ptr(), (float*)dpose.data_ptr(), (float*)ptr(dtrans),
                          (float*)ptr(dscale), (float*)ptr(dfocal), (void*)a256((size_t)work.data_ptr()), plan.ws_bwd)),
             "dpc_project_backward");
    if (dscale.defined()) dscale = dscale.reshape(sd["scale_shape"].toIntVector());
    if (dfocal.defined()) dfocal = dfocal.reshape(sd["focal_shape"].toIntVector());
    variable_list out(27);      // one slot per forward argument (non-tensor arguments stay undefined)
    out[0] = dpc;
    out[1] = dpose;
    out[2] = dtrans;
    out[3] = dscale;
    out[4] = dfocal;
    return out;
  }
};

// -> (proj, proj_depth | None, tr_pc, l2_grad | None, sil_loss | None, sil_winners | None, sil_inst_err | None)
std::vector<optional<Tensor>> project_fused(const Tensor& pc, const Tensor& pose, const optional<Tensor>& trans, const optional<Tensor>& scale,
                                  const optional<Tensor>& focal, const optional<Tensor>& tx, const optional<Tensor>& ty,
                                  const optional<Tensor>& tz, const optional<Tensor>& dropout_state, const optional<Tensor>& l2_target,
                                  const optional<Tensor>& sil_gt, const optional<Tensor>& sil_valid, int64_t lib, int64_t Dz, int64_t D,
                                  double cd, double f, double eps, double max_depth, bool quat, int64_t collapse, int64_t dropout_keep,
                                  int64_t dropout_seed, double l2_weight, int64_t views_per_cloud, int64_t sil_C, int64_t poison) {
  variable_list o = ProjectFusedFn::apply(pc, pose, trans, scale, focal, tx, ty, tz, dropout_state, l2_target, sil_gt, sil_valid, lib, Dz, D,
                                          cd, f, eps, max_depth, quat, collapse, dropout_keep, dropout_seed, l2_weight, views_per_cloud,
                                          sil_C, poison);
  std::vector<optional<Tensor>> out(7);
  size_t i = 2;
  out[0] = o[0];
  out[2] = o[1];
  if (collapse == DPC_COLLAPSE_DRC) out[1] = o[i++];
  if (l2_target.has_value() && l2_target->defined()) out[3] = o[i++];
  if (sil_gt.has_value() && sil_gt->defined()) {
    out[4] = o[i];
    out[5] = o[i + 1];
    out[6] = o[i + 2];
  }
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled PyTorch binding of the MI355X point-cloud projector (C ABI: include/dpc_hip.h)";
  m.def("set_dry_run", [](bool on) { g_dry_run = on; }, "dev switch: skip the compute calls (host-path timing only)");
  m.def("open_library", &open_library, "dlopen a build of the C-ABI library; returns its id");
  m.def("project_fused", &project_fused, "pointcloud_project_fast as one C++ autograd node (see ops.ProjectFused)");
}
