// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Own driver around the reference's UNPATCHED sparse-conv CPU sources, compiled where they
// lie under /root/reference by oracle/build_ref.py into oracle/_ref/ (never copied):
//
//   mmdet3d/ops/spconv/include/spconv/geometry.h   getIndicePairsConv :145, getIndicePairsSubM :248
//   mmdet3d/ops/spconv/src/reordering.cc           SparseGatherFunctor<tv::CPU> :21, SparseScatterAddFunctor<tv::CPU> :34
//
// The reference's `tensorview.h` includes <cuda_runtime_api.h>; the image ships the real
// header inside the triton wheel (triton/backends/nvidia/include) — no stand-in is written.
// The driver allocates exactly what `getIndicePair<3>` (spconv_ops.h:28-141) allocates on the
// CPU branch and calls the same geometry functions the CPU functors of src/indice.cc:22-62
// forward to; the conv forward / backward follow the call sequence of `indiceConv` /
// `indiceConvBackward` (spconv_ops.h:260-456): centre-offset mm for SubM, then per offset
// {reference gather, torch::mm_out, reference scatter-add}.
#include <spconv/geometry.h>
#include <spconv/reordering.h>
#include <torch/extension.h>

#include <algorithm>
#include <limits>
#include <vector>

namespace {

template <typename T>
tv::TensorView<T> view_of(const torch::Tensor &t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back(s);
  return tv::TensorView<T>(t.data_ptr<std::remove_const_t<T>>(), shape);
}

std::vector<torch::Tensor> ref_get_indice_pairs(torch::Tensor indices, int64_t batch,
                                                std::vector<int64_t> out_shape,
                                                std::vector<int64_t> ksize,
                                                std::vector<int64_t> stride,
                                                std::vector<int64_t> padding,
                                                std::vector<int64_t> dilation, bool subm) {
  TORCH_CHECK(indices.dtype() == torch::kInt32 && indices.dim() == 2 && indices.size(1) == 4);
  indices = indices.contiguous();
  const int64_t n = indices.size(0);
  int64_t kvol = 1, ovol = 1;
  for (int i = 0; i < 3; ++i) { kvol *= ksize[i]; ovol *= out_shape[i]; }
  auto i32 = torch::dtype(torch::kInt32);
  torch::Tensor pairs = torch::full({kvol, 2, n}, -1, i32);
  torch::Tensor num = torch::zeros({kvol}, i32);
  torch::Tensor grid = torch::full({batch * ovol}, -1, i32);
  int os[3], ks[3], st[3], pd[3], dl[3];
  for (int i = 0; i < 3; ++i) {
    os[i] = out_shape[i]; ks[i] = ksize[i]; dl[i] = dilation[i];
    st[i] = subm ? 1 : stride[i];
    pd[i] = subm ? ksize[i] / 2 : padding[i];
  }
  if (subm) {
    spconv::getIndicePairsSubM<int, int, 3>(view_of<const int>(indices), view_of<int>(grid),
                                            view_of<int>(pairs), view_of<int>(num), ks, st, pd,
                                            dl, os);
    return {indices, pairs, num};
  }
  torch::Tensor outids = torch::zeros({n * kvol, 4}, i32);
  int nout = spconv::getIndicePairsConv<int, int, 3>(
      view_of<const int>(indices), view_of<int>(outids), view_of<int>(grid), view_of<int>(pairs),
      view_of<int>(num), ks, st, pd, dl, os);
  return {outids.slice(0, 0, nout).contiguous(), pairs, num};
}

torch::Tensor ref_indice_conv(torch::Tensor features, torch::Tensor filters, torch::Tensor pairs,
                              torch::Tensor num, int64_t n_out, bool inverse, bool subm) {
  features = features.contiguous();
  const int64_t kvol = pairs.size(0), cin = features.size(1);
  const int64_t cout = filters.size(filters.dim() - 1);
  const int *cnt = num.data_ptr<int>();
  const int *mx = std::max_element(cnt, cnt + kvol);
  const int max_off = mx - cnt, max_n = *mx;
  auto opt = features.options();
  torch::Tensor out = torch::zeros({n_out, cout}, opt);
  torch::Tensor ibuf = torch::zeros({max_n, cin}, opt);
  torch::Tensor obuf = torch::zeros({max_n, cout}, opt);
  filters = filters.view({-1, cin, cout});
  if (subm) torch::mm_out(out, features, filters[max_off]);
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> gather;
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> scatter;
  for (int k = 0; k < kvol; ++k) {
    const int hot = cnt[k];
    if (hot <= 0 || (subm && k == max_off)) continue;
    auto ob = torch::from_blob(obuf.data_ptr<float>(), {hot, cout}, opt);
    auto ib = torch::from_blob(ibuf.data_ptr<float>(), {hot, cin}, opt);
    gather(tv::CPU(), view_of<float>(ibuf), view_of<const float>(features),
           view_of<const int>(pairs).subview(k, int(inverse)), hot);
    torch::mm_out(ob, ib, filters[k]);
    scatter(tv::CPU(), view_of<float>(out), view_of<const float>(obuf),
            view_of<const int>(pairs).subview(k, int(!inverse)), hot, true);
  }
  return out;
}

std::vector<torch::Tensor> ref_indice_conv_backward(torch::Tensor features, torch::Tensor filters,
                                                    torch::Tensor out_grad, torch::Tensor pairs,
                                                    torch::Tensor num, bool inverse, bool subm) {
  features = features.contiguous();
  out_grad = out_grad.contiguous();
  const int64_t kvol = pairs.size(0), cin = features.size(1);
  const int64_t cout = filters.size(filters.dim() - 1);
  const int *cnt = num.data_ptr<int>();
  const int *mx = std::max_element(cnt, cnt + kvol);
  const int max_off = mx - cnt, max_n = *mx;
  auto opt = features.options();
  auto fshape = filters.sizes().vec();
  torch::Tensor in_grad = torch::zeros(features.sizes(), opt);
  torch::Tensor f_grad = torch::zeros({kvol, cin, cout}, opt);
  torch::Tensor ibuf = torch::zeros({max_n, cin}, opt);
  torch::Tensor obuf = torch::zeros({max_n, cout}, opt);
  filters = filters.view({-1, cin, cout});
  if (subm) {
    auto sub = f_grad[max_off];
    torch::mm_out(sub, features.t(), out_grad);
    torch::mm_out(in_grad, out_grad, filters[max_off].t());
  }
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> gather;
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> scatter;
  for (int k = 0; k < kvol; ++k) {
    const int hot = cnt[k];
    if (hot <= 0 || (subm && k == max_off)) continue;
    gather(tv::CPU(), view_of<float>(ibuf), view_of<const float>(features),
           view_of<const int>(pairs).subview(k, int(inverse)), hot);
    gather(tv::CPU(), view_of<float>(obuf), view_of<const float>(out_grad),
           view_of<const int>(pairs).subview(k, int(!inverse)), hot);
    auto sub = f_grad[k];
    auto ob = torch::from_blob(obuf.data_ptr<float>(), {hot, cout}, opt);
    auto ib = torch::from_blob(ibuf.data_ptr<float>(), {hot, cin}, opt);
    torch::mm_out(sub, ib.t(), ob);
    torch::mm_out(ib, ob, filters[k].t());
    scatter(tv::CPU(), view_of<float>(in_grad), view_of<const float>(ibuf),
            view_of<const int>(pairs).subview(k, int(inverse)), hot, false);
  }
  return {in_grad, f_grad.view(fshape)};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("get_indice_pairs", &ref_get_indice_pairs);
  m.def("indice_conv", &ref_indice_conv);
  m.def("indice_conv_backward", &ref_indice_conv_backward);
}
