// Call latency of cudf::groupby::aggregate through the C++ API (no Python): where do 200 us go at 10K rows?
#include <cudf/aggregation.hpp>
#include <cudf/groupby.hpp>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
  for (int n : {10000, 1000000}) {
    std::vector<int64_t> hk(n);
    std::vector<double> hv(n, 1.0);
    for (int i = 0; i < n; ++i) hk[i] = (i * 7919ll) % 1000;
    int64_t* dk; double* dv;
    (void)hipMalloc(&dk, n * 8); (void)hipMalloc(&dv, n * 8);
    (void)hipMemcpy(dk, hk.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dv, hv.data(), n * 8, hipMemcpyHostToDevice);
    cudf::column_view kc{cudf::data_type{cudf::type_id::INT64}, n, dk, nullptr, 0}, vc{cudf::data_type{cudf::type_id::FLOAT64}, n, dv, nullptr, 0};
    auto call = [&] {
      cudf::groupby::groupby g(cudf::table_view({kc}));
      std::vector<cudf::groupby::aggregation_request> r(1);
      r[0].values = vc;
      r[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_aggregation>());
      r[0].aggregations.push_back(cudf::make_count_aggregation<cudf::groupby_aggregation>());
      return g.aggregate(r);
    };
    for (int i = 0; i < 10; ++i) call();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) call();
    (void)hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    std::printf("C++ groupby n=%d: %.1f us per call\n", n, us);
    // reference points: an empty kernel launch + sync, and a 4-byte D2H copy + sync
    t0 = std::chrono::steady_clock::now();
    int h;
    for (int i = 0; i < 200; ++i) { (void)hipMemcpyAsync(&h, dk, 4, hipMemcpyDeviceToHost, 0); (void)hipStreamSynchronize(0); }
    std::printf("   4-byte D2H + sync: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
  }
}
