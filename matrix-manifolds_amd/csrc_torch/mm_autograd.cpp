// C++ autograd nodes for the all-pairs distances (Manifold.pdist, manifolds/base.py:59-63; spd.py:175-181): the forward and
// the backward of `man.pdist(x)` are each ONE call of the C ABI (include/mm_manifolds.h), issued from C++ — the autograd
// engine runs the backward node on its device thread without re-entering Python (graphembed/manifolds/spd.py and vector.py
// hold the same two calls as torch.autograd.Function classes: ~130 us of host time per forward + backward there, most of
// it the engine's hand-off to a Python callback).  Host plumbing only: no kernels here, the library is bound
// at run time (dlopen of the libmm_manifolds.so the ctypes layer already loaded), tensors come from torch's allocator.
#include <dlfcn.h>

#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>

namespace {

using stream_t = void*;
struct Abi {
  size_t (*spd_ws_bytes)(int, int64_t, int) = nullptr;
  int (*spd_fwd)(int, const void*, int64_t, int, int64_t, int64_t, int, double, double, void*, void*, int, stream_t) = nullptr;
  int (*spd_bwd)(int, const void*, const void*, int64_t, int, int64_t, int64_t, int, double, double, void*, void*, int,
                 stream_t) = nullptr;
  int (*spd_status)(const void*, int64_t, int*, stream_t) = nullptr;
  size_t (*vec_ws_bytes)(int, int64_t, int) = nullptr;
  int (*vec_fwd)(int, int, const void*, int64_t, int, int64_t, int64_t, int, void*, stream_t) = nullptr;
  int (*vec_fwd_gram)(int, int, const void*, int64_t, int, int64_t, int64_t, int, void*, stream_t) = nullptr;
  int (*vec_bwd)(int, int, const void*, const void*, int64_t, int, int64_t, int64_t, int, void*, void*, stream_t) = nullptr;
  int (*vec_bwd_gram)(int, int, const void*, const void*, int64_t, int, int64_t, int64_t, int, void*, stream_t) = nullptr;
  int64_t (*pair_offset)(int64_t, int64_t) = nullptr;
  bool ready = false;
} abi;

constexpr int kAbiBuiltFor = 2;   // include/mm_manifolds.h, mm_abi_version()

template <typename F> void bind(void* h, F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(h, name));
  if (!f) throw std::runtime_error(std::string("libmm_manifolds.so lacks ") + name);
}

void init(const std::string& path) {
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen(path.c_str(), RTLD_NOW);
  if (!h) throw std::runtime_error("cannot load " + path + ": " + dlerror());
  // the entry points below are bound by NAME: a library with an older C ABI would be called through mismatched signatures
  int (*abi_version)() = nullptr;
  bind(h, abi_version, "mm_abi_version");
  if (abi_version() < kAbiBuiltFor)
    throw std::runtime_error(path + " has C-ABI version " + std::to_string(abi_version()) + ", this module was written for " +
                             std::to_string(kAbiBuiltFor) + ": rebuild both with `python __graft_entry__.py`");
  bind(h, abi.spd_ws_bytes, "mm_spd_pdist_ws_bytes");
  bind(h, abi.spd_fwd, "mm_spd_pdist_fwd");
  bind(h, abi.spd_bwd, "mm_spd_pdist_bwd");
  bind(h, abi.spd_status, "mm_spd_status");
  bind(h, abi.vec_ws_bytes, "mm_vec_pdist_ws_bytes");
  bind(h, abi.vec_fwd, "mm_vec_pdist_fwd");
  bind(h, abi.vec_fwd_gram, "mm_vec_pdist_fwd_gram");
  bind(h, abi.vec_bwd, "mm_vec_pdist_bwd");
  bind(h, abi.vec_bwd_gram, "mm_vec_pdist_bwd_gram");
  bind(h, abi.pair_offset, "mm_pair_offset");
  abi.ready = true;
}

int dtype_code(const at::Tensor& t) {
  if (t.scalar_type() == at::kFloat) return 0;   // MM_F32
  if (t.scalar_type() == at::kDouble) return 1;  // MM_F64
  throw std::invalid_argument("matrix-manifolds_amd kernels exist for float32/float64");
}
void check(int rc, const char* name) {
  if (rc == 0) return;
  const std::string kind = rc == -1 ? "invalid argument" : rc == -2 ? "unsupported size/dtype" : "hipError_t " + std::to_string(rc);
  throw std::runtime_error(std::string(name) + " failed: " + kind);
}
stream_t stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
void require_gpu(const at::Tensor& t) {
  if (!abi.ready) throw std::runtime_error("mm_autograd: init(path) has not been called");
  if (!t.is_cuda())
    throw std::runtime_error("matrix-manifolds_amd runs on MI355X only: got a CPU tensor. Move the embedding to 'cuda' "
                             "(torch-ROCm device); CPU execution is the reference's job.");
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// SymmetricPositiveDefinite.pdist (spd.py:175-181) over the rows [row_begin, row_end) of the pair list
struct SpdPdist : torch::autograd::Function<SpdPdist> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, int64_t d, bool squared, double wmin, double wmax,
                            int64_t row_begin, int64_t row_end, bool check_pd) {
    require_gpu(x);
    const at::Tensor xc = x.detach().contiguous();
    const int64_t n = xc.size(0);
    const int dt = dtype_code(xc);
    const int64_t npairs = abi.pair_offset(n, row_end) - abi.pair_offset(n, row_begin);
    ctx->saved_data["empty"] = npairs == 0;
    ctx->save_for_backward({xc});
    if (npairs == 0) return at::empty({0}, xc.options());
    c10::DeviceGuard guard(xc.device());
    at::Tensor ws = at::empty({int64_t(abi.spd_ws_bytes(dt, n, int(d)))}, xc.options().dtype(at::kByte));
    at::Tensor out = at::empty({npairs}, xc.options());
    check(abi.spd_fwd(dt, xc.data_ptr(), n, int(d), row_begin, row_end, squared ? 1 : 0, wmin, wmax, out.data_ptr(), ws.data_ptr(),
                      0, stream_of(xc)), "mm_spd_pdist_fwd");
    if (check_pd) {
      int st = 0;
      check(abi.spd_status(ws.data_ptr(), n, &st, stream_of(xc)), "mm_spd_status");
      if (st) throw std::runtime_error("linalg: pdist: " + std::to_string(st) + " input matrices are not positive-definite");
    }
    ctx->saved_data["ws"] = ws;
    ctx->saved_data["d"] = d;
    ctx->saved_data["squared"] = squared;
    ctx->saved_data["wmin"] = wmin;
    ctx->saved_data["wmax"] = wmax;
    ctx->saved_data["rb"] = row_begin;
    ctx->saved_data["re"] = row_end;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const at::Tensor xc = ctx->get_saved_variables()[0];
    variable_list out(8);
    if (ctx->saved_data["empty"].toBool()) {
      out[0] = at::zeros_like(xc);
      return out;
    }
    const at::Tensor g = grads[0].contiguous();
    c10::DeviceGuard guard(xc.device());
    at::Tensor grad = at::empty_like(xc);
    at::Tensor ws = ctx->saved_data["ws"].toTensor();
    check(abi.spd_bwd(dtype_code(xc), xc.data_ptr(), g.data_ptr(), xc.size(0), int(ctx->saved_data["d"].toInt()),
                      ctx->saved_data["rb"].toInt(), ctx->saved_data["re"].toInt(), ctx->saved_data["squared"].toBool() ? 1 : 0,
                      ctx->saved_data["wmin"].toDouble(), ctx->saved_data["wmax"].toDouble(), grad.data_ptr(), ws.data_ptr(),
                      1 /* MM_WS_PREPARED */, stream_of(xc)), "mm_spd_pdist_bwd");
    out[0] = grad;
    return out;
  }
};

// Manifold.pdist of a vector manifold (base.py:59-63 with euclidean.py / lorentz.py / sphere.py); fwd_gram / bwd_gram: the
// matrix-core forms (chosen by graphembed/manifolds/vector.py, as for its own autograd class)
struct VecPdist : torch::autograd::Function<VecPdist> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, int64_t kind, int64_t m, bool squared, int64_t row_begin,
                            int64_t row_end, bool fwd_gram, bool bwd_gram) {
    require_gpu(x);
    const int64_t n = x.size(0);
    const at::Tensor xc = x.detach().reshape({n, m}).contiguous();
    const int dt = dtype_code(xc);
    const int64_t npairs = abi.pair_offset(n, row_end) - abi.pair_offset(n, row_begin);
    ctx->saved_data["empty"] = npairs == 0;
    ctx->saved_data["shape"] = x.sizes().vec();
    ctx->save_for_backward({xc});
    if (npairs == 0) return at::empty({0}, xc.options());
    c10::DeviceGuard guard(xc.device());
    at::Tensor out = at::empty({npairs}, xc.options());
    check((fwd_gram ? abi.vec_fwd_gram : abi.vec_fwd)(dt, int(kind), xc.data_ptr(), n, int(m), row_begin, row_end, squared ? 1 : 0,
                                                      out.data_ptr(), stream_of(xc)),
          fwd_gram ? "mm_vec_pdist_fwd_gram" : "mm_vec_pdist_fwd");
    ctx->saved_data["kind"] = kind;
    ctx->saved_data["m"] = m;
    ctx->saved_data["squared"] = squared;
    ctx->saved_data["rb"] = row_begin;
    ctx->saved_data["re"] = row_end;
    ctx->saved_data["bwd_gram"] = bwd_gram;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const at::Tensor xc = ctx->get_saved_variables()[0];
    const auto shape = ctx->saved_data["shape"].toIntVector();
    variable_list out(8);
    if (ctx->saved_data["empty"].toBool()) {
      out[0] = at::zeros(shape, xc.options());
      return out;
    }
    const at::Tensor g = grads[0].contiguous();
    const int64_t n = xc.size(0), m = ctx->saved_data["m"].toInt();
    const int dt = dtype_code(xc), kind = int(ctx->saved_data["kind"].toInt()), sq = ctx->saved_data["squared"].toBool() ? 1 : 0;
    c10::DeviceGuard guard(xc.device());
    at::Tensor grad = at::empty_like(xc);
    if (ctx->saved_data["bwd_gram"].toBool()) {
      check(abi.vec_bwd_gram(dt, kind, xc.data_ptr(), g.data_ptr(), n, int(m), ctx->saved_data["rb"].toInt(),
                             ctx->saved_data["re"].toInt(), sq, grad.data_ptr(), stream_of(xc)), "mm_vec_pdist_bwd_gram");
    } else {
      at::Tensor ws = at::empty({int64_t(abi.vec_ws_bytes(dt, n, int(m)))}, xc.options().dtype(at::kByte));
      check(abi.vec_bwd(dt, kind, xc.data_ptr(), g.data_ptr(), n, int(m), ctx->saved_data["rb"].toInt(),
                        ctx->saved_data["re"].toInt(), sq, grad.data_ptr(), ws.data_ptr(), stream_of(xc)), "mm_vec_pdist_bwd");
    }
    out[0] = grad.reshape(shape);
    return out;
  }
};

at::Tensor spd_pdist(const at::Tensor& x, int64_t d, bool squared, double wmin, double wmax, int64_t row_begin, int64_t row_end,
                     bool check_pd) {
  return SpdPdist::apply(x, d, squared, wmin, wmax, row_begin, row_end, check_pd);
}
at::Tensor vec_pdist(const at::Tensor& x, int64_t kind, int64_t m, bool squared, int64_t row_begin, int64_t row_end, bool fwd_gram,
                     bool bwd_gram) {
  return VecPdist::apply(x, kind, m, squared, row_begin, row_end, fwd_gram, bwd_gram);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
  mod.def("init", &init, "bind the C ABI of libmm_manifolds.so (path)");
  mod.def("spd_pdist", &spd_pdist);
  mod.def("vec_pdist", &vec_pdist);
}
