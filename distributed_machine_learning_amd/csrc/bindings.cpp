// Python bindings of the native runtime (pybind11, no torch headers).
//
// Tensors cross the boundary as raw device addresses (tensor.data_ptr()) and streams as
// hipStream_t handles (torch.cuda.current_stream().cuda_stream); the Python wrappers in
// ops/native.py validate dtype / shape / contiguity before calling in.  Memory is owned
// by the torch caching allocator, the native side never frees or allocates tensors.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "dml_comm.h"
#include "dml_engine.h"
#include "dml_kernels.h"

namespace py = pybind11;
using namespace dml;

namespace {

template <typename T>
T* ptr(uintptr_t p) { return reinterpret_cast<T*>(p); }
hipStream_t st(uintptr_t s) { return reinterpret_cast<hipStream_t>(s); }

ConvShape shape(int batch, int hw, int cin, int cout) {
    ConvShape s;
    s.batch = batch; s.hw = hw; s.cin = cin; s.cout = cout;
    return s;
}

}  // namespace

PYBIND11_MODULE(_C, m) {
    m.doc() = "MI355X-native kernels, training engine and RCCL communicator";

    // ------------------------------------------------------------------ kernels
    m.def("conv3x3_igemm", [](uintptr_t x, uintptr_t w, uintptr_t y, uintptr_t bias, uintptr_t mask,
                              uintptr_t ws, size_t ws_elems, int batch, int hw, int cin, int cout,
                              int epi, uintptr_t stream, int bm, int bn, int splits, int pm,
                              int waves, uintptr_t stats) {
        const int force[5] = {bm, bn, splits, pm, waves};
        conv3x3_igemm(ptr<const bf16_raw>(x), ptr<const bf16_raw>(w), ptr<bf16_raw>(y),
                      ptr<const float>(bias), ptr<const bf16_raw>(mask), ptr<float>(ws), ws_elems,
                      shape(batch, hw, cin, cout), (ConvEpilogue)epi, st(stream),
                      bm > 0 ? force : nullptr, ptr<float>(stats));
    }, py::arg("x"), py::arg("w"), py::arg("y"), py::arg("bias"), py::arg("mask"), py::arg("ws"),
       py::arg("ws_elems"), py::arg("batch"), py::arg("hw"), py::arg("cin"), py::arg("cout"),
       py::arg("epi"), py::arg("stream"), py::arg("bm") = 0, py::arg("bn") = 0,
       py::arg("splits") = 0, py::arg("pm") = -1, py::arg("waves") = 4, py::arg("stats") = 0);
    // backward-data pass that also leaves the BatchNorm + pool backward sums of the layer whose
    // pooled gradient it writes; returns the number of partial rows (0: the plan cannot)
    m.def("conv3x3_igemm_bn_pool", [](uintptr_t x, uintptr_t w, uintptr_t y, uintptr_t ws,
                                      size_t ws_elems, int batch, int hw, int cin, int cout,
                                      uintptr_t stream, uintptr_t pre, uintptr_t gamma,
                                      uintptr_t beta, uintptr_t mean, uintptr_t rstd,
                                      uintptr_t partials) {
        BnPoolReduce red;
        red.pre = ptr<const bf16_raw>(pre);
        red.gamma = ptr<const float>(gamma); red.beta = ptr<const float>(beta);
        red.mean = ptr<const float>(mean); red.rstd = ptr<const float>(rstd);
        red.partials = ptr<float>(partials);
        BnPartials done;
        conv3x3_igemm(ptr<const bf16_raw>(x), ptr<const bf16_raw>(w), ptr<bf16_raw>(y), nullptr,
                      nullptr, ptr<float>(ws), ws_elems, shape(batch, hw, cin, cout), EPI_NONE,
                      st(stream), nullptr, nullptr, nullptr, &red, &done);
        return done.data ? done.rows : 0;
    });
    m.def("conv3x3_igemm_stats_rows", [](int batch, int hw, int cin, int cout) {
        return conv3x3_igemm_stats_rows(shape(batch, hw, cin, cout));
    });
    m.def("conv1_fprop_stats_rows", &conv1_fprop_stats_rows);
    m.def("conv3x3_igemm_workspace_elems", [](int batch, int hw, int cin, int cout) {
        return conv3x3_igemm_workspace_elems(shape(batch, hw, cin, cout));
    });
    m.def("conv3x3_wgrad", [](uintptr_t x, uintptr_t dy, uintptr_t dw, int batch, int hw, int cin,
                              int cout, uintptr_t stream, int splits, uintptr_t slab) {
        conv3x3_wgrad(ptr<const bf16_raw>(x), ptr<const bf16_raw>(dy), ptr<float>(dw),
                      shape(batch, hw, cin, cout), st(stream), splits, ptr<float>(slab));
    }, py::arg("x"), py::arg("dy"), py::arg("dw"), py::arg("batch"), py::arg("hw"),
       py::arg("cin"), py::arg("cout"), py::arg("stream"), py::arg("splits") = 0,
       py::arg("slab") = 0);
    m.def("conv3x3_wgrad_splits", [](int batch, int hw, int cin, int cout) {
        return conv3x3_wgrad_splits(shape(batch, hw, cin, cout));
    });
    m.def("sum_slabs_batch", [](uintptr_t src, uintptr_t dst, std::vector<size_t> src_off,
                                std::vector<size_t> dst_off, std::vector<size_t> count,
                                std::vector<int> splits, uintptr_t stream) {
        SlabSumJobs jobs;
        if (src_off.size() > (size_t)SlabSumJobs::kMax || src_off.size() != dst_off.size() ||
            src_off.size() != count.size() || src_off.size() != splits.size())
            throw std::invalid_argument("sum_slabs_batch: bad job lists");
        jobs.n = (int)src_off.size();
        for (int j = 0; j < jobs.n; ++j) {
            jobs.src_off[j] = src_off[j]; jobs.dst_off[j] = dst_off[j];
            jobs.count[j] = count[j]; jobs.splits[j] = splits[j];
        }
        sum_slabs_batch(ptr<const float>(src), ptr<float>(dst), jobs, st(stream));
    });
    m.def("conv1_fprop", [](uintptr_t x, uintptr_t w1, uintptr_t bias, uintptr_t y, int batch,
                            bool relu, uintptr_t stream, uintptr_t stats) {
        conv1_fprop(ptr<const bf16_raw>(x), ptr<const bf16_raw>(w1), ptr<const float>(bias),
                    ptr<bf16_raw>(y), batch, relu, st(stream), ptr<float>(stats));
    }, py::arg("x"), py::arg("w1"), py::arg("bias"), py::arg("y"), py::arg("batch"),
       py::arg("relu"), py::arg("stream"), py::arg("stats") = 0);
    m.def("conv1_wgrad", [](uintptr_t x, uintptr_t dy, uintptr_t dw, uintptr_t db, int batch,
                            uintptr_t ws, uintptr_t stream) {
        conv1_wgrad(ptr<const bf16_raw>(x), ptr<const bf16_raw>(dy), ptr<float>(dw), ptr<float>(db),
                    batch, ptr<float>(ws), st(stream));
    });
    m.def("conv1_wgrad_workspace_elems", &conv1_wgrad_workspace_elems);
    m.def("conv1_wgrad_relu_pool", [](uintptr_t x, uintptr_t act, uintptr_t dyp, uintptr_t dw,
                                      uintptr_t db, int batch, uintptr_t ws, uintptr_t s) {
        conv1_wgrad_relu_pool(ptr<const bf16_raw>(x), ptr<const bf16_raw>(act),
                              ptr<const bf16_raw>(dyp), ptr<float>(dw), ptr<float>(db), batch,
                              ptr<float>(ws), st(s));
    });
    m.def("conv1_wgrad_bn_pool", [](uintptr_t x, uintptr_t pre, uintptr_t dyp, uintptr_t gamma,
                                    uintptr_t beta, uintptr_t mean, uintptr_t rstd, uintptr_t dgamma,
                                    uintptr_t dbeta, uintptr_t dw, int batch, uintptr_t ws,
                                    uintptr_t s) {
        conv1_wgrad_bn_pool(ptr<const bf16_raw>(x), ptr<const bf16_raw>(pre),
                            ptr<const bf16_raw>(dyp), ptr<const float>(gamma), ptr<const float>(beta),
                            ptr<const float>(mean), ptr<const float>(rstd),
                            ptr<const float>(dgamma), ptr<const float>(dbeta), ptr<float>(dw), batch,
                            ptr<float>(ws), st(s));
    });
    m.def("column_reduce_workspace_elems", &bn_workspace_elems);
    m.def("repack_dgrad_weights", [](uintptr_t wf, uintptr_t wd, int cout, int cin, uintptr_t s) {
        repack_dgrad_weights(ptr<const bf16_raw>(wf), ptr<bf16_raw>(wd), cout, cin, st(s));
    });
    m.def("repack_conv1_weights", [](uintptr_t w, uintptr_t w1, uintptr_t s) {
        repack_conv1_weights(ptr<const float>(w), ptr<bf16_raw>(w1), st(s));
    });
    m.def("column_sum", [](uintptr_t x, uintptr_t out, int rows, int cols, uintptr_t ws,
                           uintptr_t s) {
        column_sum(ptr<const bf16_raw>(x), ptr<float>(out), rows, cols, ptr<float>(ws), st(s));
    });
    m.def("maxpool2x2_fwd", [](uintptr_t x, uintptr_t y, int batch, int hw, int c, uintptr_t s) {
        maxpool2x2_fwd(ptr<const bf16_raw>(x), ptr<bf16_raw>(y), batch, hw, c, st(s));
    });
    m.def("maxpool2x2_bwd", [](uintptr_t x, uintptr_t dy, uintptr_t dx, int batch, int hw, int c,
                               bool relu_mask, uintptr_t s) {
        maxpool2x2_bwd(ptr<const bf16_raw>(x), ptr<const bf16_raw>(dy), ptr<bf16_raw>(dx), batch,
                       hw, c, relu_mask, st(s));
    });
    m.def("bn_relu_fwd_train", [](uintptr_t x, uintptr_t y, uintptr_t gamma, uintptr_t beta,
                                  uintptr_t rm, uintptr_t rv, uintptr_t smean, uintptr_t srstd,
                                  uintptr_t ws, int rows, int c, float momentum, float eps,
                                  uintptr_t s) {
        bn_relu_fwd_train(ptr<const bf16_raw>(x), ptr<bf16_raw>(y), ptr<const float>(gamma),
                          ptr<const float>(beta), ptr<float>(rm), ptr<float>(rv), ptr<float>(smean),
                          ptr<float>(srstd), ptr<float>(ws), rows, c, momentum, eps, st(s));
    });
    m.def("bn_relu_fwd_eval", [](uintptr_t x, uintptr_t y, uintptr_t gamma, uintptr_t beta,
                                 uintptr_t rm, uintptr_t rv, int rows, int c, float eps,
                                 uintptr_t s) {
        bn_relu_fwd_eval(ptr<const bf16_raw>(x), ptr<bf16_raw>(y), ptr<const float>(gamma),
                         ptr<const float>(beta), ptr<const float>(rm), ptr<const float>(rv), rows,
                         c, eps, st(s));
    });
    m.def("bn_relu_bwd", [](uintptr_t x, uintptr_t dy, uintptr_t dx, uintptr_t gamma,
                            uintptr_t beta, uintptr_t smean, uintptr_t srstd, uintptr_t dgamma,
                            uintptr_t dbeta, uintptr_t ws, int rows, int c, uintptr_t s,
                            uintptr_t slabs, int splits, size_t stride) {
        bn_relu_bwd(ptr<const bf16_raw>(x), ptr<bf16_raw>(dy), ptr<bf16_raw>(dx),
                    ptr<const float>(gamma), ptr<const float>(beta), ptr<const float>(smean),
                    ptr<const float>(srstd), ptr<float>(dgamma), ptr<float>(dbeta), ptr<float>(ws),
                    rows, c, st(s), SplitSlabs{ptr<const float>(slabs), splits, stride});
    }, py::arg("x"), py::arg("dy"), py::arg("dx"), py::arg("gamma"), py::arg("beta"),
       py::arg("save_mean"), py::arg("save_rstd"), py::arg("dgamma"), py::arg("dbeta"),
       py::arg("ws"), py::arg("rows"), py::arg("c"), py::arg("stream"), py::arg("slabs") = 0,
       py::arg("splits") = 0, py::arg("stride") = 0);
    m.def("bn_relu_pool_fwd_train", [](uintptr_t x, uintptr_t y, uintptr_t gamma, uintptr_t beta,
                                       uintptr_t rmean, uintptr_t rvar, uintptr_t smean,
                                       uintptr_t srstd, uintptr_t ws, int batch, int hw, int c,
                                       float momentum, float eps, uintptr_t s) {
        bn_relu_pool_fwd_train(ptr<const bf16_raw>(x), ptr<bf16_raw>(y), ptr<const float>(gamma),
                               ptr<const float>(beta), ptr<float>(rmean), ptr<float>(rvar),
                               ptr<float>(smean), ptr<float>(srstd), ptr<float>(ws), batch, hw, c,
                               momentum, eps, st(s));
    });
    m.def("bn_relu_pool_fwd_eval", [](uintptr_t x, uintptr_t y, uintptr_t gamma, uintptr_t beta,
                                      uintptr_t rmean, uintptr_t rvar, int batch, int hw, int c,
                                      float eps, uintptr_t s) {
        bn_relu_pool_fwd_eval(ptr<const bf16_raw>(x), ptr<bf16_raw>(y), ptr<const float>(gamma),
                              ptr<const float>(beta), ptr<const float>(rmean),
                              ptr<const float>(rvar), batch, hw, c, eps, st(s));
    });
    m.def("bn_relu_pool_bwd", [](uintptr_t x, uintptr_t dy, uintptr_t dx, uintptr_t gamma,
                                 uintptr_t beta, uintptr_t smean, uintptr_t srstd, uintptr_t dgamma,
                                 uintptr_t dbeta, uintptr_t ws, int batch, int hw, int c,
                                 uintptr_t s, uintptr_t slabs, int splits, size_t stride) {
        bn_relu_pool_bwd(ptr<const bf16_raw>(x), ptr<bf16_raw>(dy), ptr<bf16_raw>(dx),
                         ptr<const float>(gamma), ptr<const float>(beta), ptr<const float>(smean),
                         ptr<const float>(srstd), ptr<float>(dgamma), ptr<float>(dbeta),
                         ptr<float>(ws), batch, hw, c, st(s),
                         SplitSlabs{ptr<const float>(slabs), splits, stride});
    }, py::arg("x"), py::arg("dy"), py::arg("dx"), py::arg("gamma"), py::arg("beta"),
       py::arg("save_mean"), py::arg("save_rstd"), py::arg("dgamma"), py::arg("dbeta"),
       py::arg("ws"), py::arg("batch"), py::arg("hw"), py::arg("c"), py::arg("stream"),
       py::arg("slabs") = 0, py::arg("splits") = 0, py::arg("stride") = 0);
    m.def("streams_run_concurrently", [](uintptr_t busy, uintptr_t candidate) {
        return streams_run_concurrently((hipStream_t)busy, (hipStream_t)candidate);
    });
    m.def("fc_train", [](uintptr_t x, uintptr_t w, uintptr_t b, uintptr_t target, uintptr_t logits,
                         uintptr_t loss, uintptr_t correct, uintptr_t dx, uintptr_t dw, uintptr_t db,
                         uintptr_t ws, int batch, int in_f, int out_f, uintptr_t s) {
        fc_train(ptr<const bf16_raw>(x), ptr<const float>(w), ptr<const float>(b),
                 ptr<const int64_t>(target), ptr<float>(logits), ptr<float>(loss),
                 ptr<int32_t>(correct), ptr<bf16_raw>(dx), ptr<float>(dw), ptr<float>(db),
                 ptr<float>(ws), batch, in_f, out_f, st(s));
    });
    m.def("fc_train_workspace_elems", &fc_train_workspace_elems);
    m.def("linear_fwd", [](uintptr_t x, uintptr_t w, uintptr_t b, uintptr_t logits, int batch,
                           int in_f, int out_f, uintptr_t s) {
        linear_fwd(ptr<const bf16_raw>(x), ptr<const float>(w), ptr<const float>(b),
                   ptr<float>(logits), batch, in_f, out_f, st(s));
    });
    m.def("linear_bwd", [](uintptr_t x, uintptr_t w, uintptr_t dl, uintptr_t dw, uintptr_t db,
                           uintptr_t dx, int batch, int in_f, int out_f, uintptr_t s) {
        linear_bwd(ptr<const bf16_raw>(x), ptr<const float>(w), ptr<const float>(dl),
                   ptr<float>(dw), ptr<float>(db), ptr<bf16_raw>(dx), batch, in_f, out_f, st(s));
    });
    m.def("softmax_xent", [](uintptr_t logits, uintptr_t target, uintptr_t loss, uintptr_t dl,
                             uintptr_t correct, int batch, int classes, float gscale, uintptr_t s) {
        softmax_xent(ptr<const float>(logits), ptr<const int64_t>(target), ptr<float>(loss),
                     ptr<float>(dl), ptr<int32_t>(correct), batch, classes, gscale, st(s));
    });
    m.def("sgd_momentum", [](uintptr_t p, uintptr_t g, uintptr_t mom, uintptr_t shadow, size_t n,
                             float lr, float mu, float wd, float gscale, bool first, uintptr_t s) {
        sgd_momentum(ptr<float>(p), ptr<const float>(g), ptr<float>(mom), ptr<bf16_raw>(shadow), n,
                     lr, mu, wd, gscale, first, st(s));
    });
    m.def("cast_f32_to_bf16", [](uintptr_t x, uintptr_t out, size_t n, uintptr_t s) {
        cast_f32_to_bf16(ptr<const float>(x), ptr<bf16_raw>(out), n, st(s));
    });
    m.def("augment_normalize", [](uintptr_t img, uintptr_t aug, uintptr_t out, int batch,
                                  std::vector<float> mean, std::vector<float> stdv, uintptr_t s) {
        if (mean.size() != 3 || stdv.size() != 3) throw std::invalid_argument("mean/std need 3");
        augment_normalize(ptr<const uint8_t>(img), ptr<const int32_t>(aug), ptr<bf16_raw>(out),
                          batch, mean.data(), stdv.data(), st(s));
    });
    m.def("nchw_to_nhwc4", [](uintptr_t x, uintptr_t out, int batch, uintptr_t s) {
        nchw_to_nhwc4(ptr<const float>(x), ptr<bf16_raw>(out), batch, st(s));
    });
    m.def("nchw_f32_to_nhwc_bf16", [](uintptr_t x, uintptr_t out, int b, int c, int hw, uintptr_t s) {
        nchw_f32_to_nhwc_bf16(ptr<const float>(x), ptr<bf16_raw>(out), b, c, hw, st(s));
    });
    m.def("nhwc_bf16_to_nchw_f32", [](uintptr_t x, uintptr_t out, int b, int c, int hw, uintptr_t s) {
        nhwc_bf16_to_nchw_f32(ptr<const bf16_raw>(x), ptr<float>(out), b, c, hw, st(s));
    });
    m.def("sum_slabs", [](uintptr_t slabs, int world, size_t n, uintptr_t s) {
        sum_slabs(ptr<float>(slabs), world, n, st(s));
    });
    m.def("fill_uniform_bf16", [](uintptr_t out, size_t n, uint32_t seed, float scale, uintptr_t s) {
        fill_uniform_bf16(ptr<bf16_raw>(out), n, seed, scale, st(s));
    });

    // ------------------------------------------------------------- communication
    bind_comm(m);
    // -------------------------------------------------------------------- engine
    bind_engine(m);
}
