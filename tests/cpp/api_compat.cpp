// Source-compatibility check of the C++ boundary: this file is written the way a libcudf C++ user (or a Cython
// .pxd) uses the reference API — cudf::groupby::groupby / aggregation_request / make_*_aggregation
// (cpp/include/cudf/groupby.hpp:54-184), cudf::inner_join (join/join.hpp:160-166), cudf::hash_join
// (join/hash_join.hpp:71), cudf::gather — and only swaps the stream type to hipStream_t.
// Data and expectations are the reference's own KATs (sum_tests.cpp:68-80, mean_tests.cpp:37-55,
// join_tests.cpp:2091-2118, :2396-2417).
#include <cudf/aggregation.hpp>
#include <cudf/copying.hpp>
#include <cudf/groupby.hpp>
#include <cudf/interop.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/partitioning.hpp>
#include <cudf/utilities/span.hpp>
#include <cudf/table/table.hpp>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <span>
#include <stdexcept>
#include <string>
#include <vector>

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "FAILED %s at %s:%d\n", #cond, __FILE__, __LINE__);       \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

template <typename T>
struct dev_vec {
  T* p{};
  size_t n;
  explicit dev_vec(std::vector<T> const& h) : n{h.size()}
  {
    (void)hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
    (void)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  }
  ~dev_vec() { (void)hipFree(p); }
};
template <typename T>
std::vector<T> to_host(cudf::column_view const& c)
{
  std::vector<T> h(c.size());
  (void)hipMemcpy(h.data(), c.data<T>(), h.size() * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}

int main()
{
  // ---- groupby: keys {1,2,3,1,2,2,1,3,3,2}, vals 0..9 -> SUM {9,19,17}, MEAN {3, 19/4, 17/3}, COUNT {3,4,3}
  dev_vec<int32_t> keys{{1, 2, 3, 1, 2, 2, 1, 3, 3, 2}};
  dev_vec<int32_t> vals{{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}};
  cudf::column_view kcol{cudf::data_type{cudf::type_id::INT32}, 10, keys.p, nullptr, 0};
  cudf::column_view vcol{cudf::data_type{cudf::type_id::INT32}, 10, vals.p, nullptr, 0};
  cudf::groupby::groupby gb_obj(cudf::table_view({kcol}));
  std::vector<cudf::groupby::aggregation_request> requests;
  requests.emplace_back();
  requests[0].values = vcol;
  requests[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_aggregation>());
  requests[0].aggregations.push_back(cudf::make_mean_aggregation<cudf::groupby_aggregation>());
  requests[0].aggregations.push_back(cudf::make_count_aggregation<cudf::groupby_aggregation>());
  auto result = gb_obj.aggregate(requests);
  CHECK(result.first->num_rows() == 3);
  CHECK(result.second.size() == 1 && result.second[0].results.size() == 3);
  CHECK(result.second[0].results[0]->type().id() == cudf::type_id::INT64);
  CHECK(result.second[0].results[1]->type().id() == cudf::type_id::FLOAT64);
  CHECK(result.second[0].results[2]->type().id() == cudf::type_id::INT32);
  auto k = to_host<int32_t>(result.first->view().column(0));
  auto s = to_host<int64_t>(result.second[0].results[0]->view());
  auto m = to_host<double>(result.second[0].results[1]->view());
  auto c = to_host<int32_t>(result.second[0].results[2]->view());
  std::map<int32_t, int> at;
  for (int i = 0; i < 3; ++i) at[k[i]] = i;
  CHECK(at.size() == 3);
  CHECK(s[at[1]] == 9 && s[at[2]] == 19 && s[at[3]] == 17);
  CHECK(c[at[1]] == 3 && c[at[2]] == 4 && c[at[3]] == 3);
  CHECK(std::abs(m[at[1]] - 3.0) < 1e-12 && std::abs(m[at[2]] - 19.0 / 4) < 1e-12 && std::abs(m[at[3]] - 17.0 / 3) < 1e-12);

  // the reference signature takes std::span<aggregation_request const> (groupby.hpp:181-184): a vector converts, and a
  // span over part of it works the same
  {
    auto again = gb_obj.aggregate(std::span<cudf::groupby::aggregation_request const>{requests.data(), 1});
    CHECK(again.first->num_rows() == 3 && again.second[0].results.size() == 3);
  }
  // SUM_OVERFLOW: struct {sum: source type, overflow: bool} (aggregation.hpp:214-217; sum_overflow_tests.cpp:41-90)
  {
    std::vector<cudf::groupby::aggregation_request> ro(1);
    ro[0].values = vcol;
    ro[0].aggregations.push_back(cudf::make_sum_overflow_aggregation<cudf::groupby_aggregation>());
    auto out = gb_obj.aggregate(ro);
    auto const& st = *out.second[0].results[0];
    CHECK(st.type().id() == cudf::type_id::STRUCT && st.num_children() == 2 && st.size() == 3);
    CHECK(st.view().child(0).type().id() == cudf::type_id::INT32 && st.view().child(1).type().id() == cudf::type_id::BOOL8);
    auto ko = to_host<int32_t>(out.first->view().column(0));
    auto so = to_host<int32_t>(st.view().child(0));
    auto fo = to_host<uint8_t>(st.view().child(1));
    std::map<int32_t, int32_t> mo;
    for (int i = 0; i < 3; ++i) { mo[ko[i]] = so[i]; CHECK(fo[i] == 0); }
    CHECK(mo[1] == 9 && mo[2] == 19 && mo[3] == 17);
  }

  // a kind without a hash implementation takes the call down the sort-based path (groupby.cu:64-69): keys come back ascending.
  // median_tests.cpp:33-53, quantile_tests.cpp:132-161 (two quantiles -> groups x 2 values), nth_element_tests.cpp:93-118
  {
    std::vector<cudf::groupby::aggregation_request> rs;
    rs.emplace_back();
    rs[0].values = vcol;
    rs[0].aggregations.push_back(cudf::make_median_aggregation<cudf::groupby_aggregation>());
    rs[0].aggregations.push_back(cudf::make_quantile_aggregation<cudf::groupby_aggregation>({0.25, 0.75}, cudf::interpolation::LINEAR));
    rs[0].aggregations.push_back(cudf::make_nth_element_aggregation<cudf::groupby_aggregation>(-1));
    rs[0].aggregations.push_back(cudf::make_nunique_aggregation<cudf::groupby_aggregation>());
    rs[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_aggregation>());
    auto out = gb_obj.aggregate(rs);
    CHECK(gb_obj.last_path() == cudf::groupby::hash_path::SORT);
    CHECK((to_host<int32_t>(out.first->view().column(0)) == std::vector<int32_t>{1, 2, 3}));
    CHECK((to_host<double>(out.second[0].results[0]->view()) == std::vector<double>{3., 4.5, 7.}));
    CHECK(out.second[0].results[1]->size() == 6);
    CHECK((to_host<double>(out.second[0].results[1]->view()) == std::vector<double>{1.5, 4.5, 3.25, 6., 4.5, 7.5}));
    CHECK((to_host<int32_t>(out.second[0].results[2]->view()) == std::vector<int32_t>{6, 9, 8}));
    CHECK((to_host<int32_t>(out.second[0].results[3]->view()) == std::vector<int32_t>{3, 4, 3}));
    CHECK((to_host<int64_t>(out.second[0].results[4]->view()) == std::vector<int64_t>{9, 19, 17}));
  }

  // size mismatch -> cudf::logic_error (groupby.cu:225-229)
  {
    dev_vec<int32_t> bad{{1, 2, 3}};
    std::vector<cudf::groupby::aggregation_request> r2;
    r2.emplace_back();
    r2[0].values = cudf::column_view{cudf::data_type{cudf::type_id::INT32}, 3, bad.p, nullptr, 0};
    r2[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_aggregation>());
    bool threw = false;
    try {
      gb_obj.aggregate(r2);
    } catch (cudf::logic_error const&) {
      threw = true;
    }
    CHECK(threw);
  }

  // ---- inner_join corner case: {4,1,3,2,2,2,2} x {2} -> 4 pairs (join_tests.cpp:2091-2118)
  dev_vec<int64_t> l{{4, 1, 3, 2, 2, 2, 2}};
  dev_vec<int64_t> r{{2}};
  cudf::table_view lt({cudf::column_view{cudf::data_type{cudf::type_id::INT64}, 7, l.p, nullptr, 0}});
  cudf::table_view rt({cudf::column_view{cudf::data_type{cudf::type_id::INT64}, 1, r.p, nullptr, 0}});
  auto [li, ri] = cudf::inner_join(lt, rt, cudf::null_equality::EQUAL);
  CHECK(li->size() == 4 && ri->size() == 4);
  std::vector<cudf::size_type> hl(4), hr(4);
  (void)hipMemcpy(hl.data(), li->data(), 16, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hr.data(), ri->data(), 16, hipMemcpyDeviceToHost);
  std::sort(hl.begin(), hl.end());
  CHECK((hl == std::vector<cudf::size_type>{3, 4, 5, 6}));
  CHECK(std::all_of(hr.begin(), hr.end(), [](auto x) { return x == 0; }));

  // ---- hash_join object + gather of the joined rows (join_tests.cpp:2396-2417)
  dev_vec<int32_t> t0{{3, 1, 2, 0, 2}};
  dev_vec<int32_t> t1{{2, 2, 0, 4, 3}};
  cudf::table_view t0v({cudf::column_view{cudf::data_type{cudf::type_id::INT32}, 5, t0.p, nullptr, 0}});
  cudf::table_view t1v({cudf::column_view{cudf::data_type{cudf::type_id::INT32}, 5, t1.p, nullptr, 0}});
  cudf::hash_join hj(t1v, cudf::null_equality::EQUAL);
  CHECK(hj.inner_join_size(t0v) == 6);
  auto [pl, pr] = hj.inner_join(t0v);
  CHECK(pl->size() == 6);
  // pylibcudf moves the index vector into a column (join.pyx:51-64); then gather
  cudf::column lcol{std::move(*pl), rmm::device_buffer{}, 0};
  cudf::column rcol{std::move(*pr), rmm::device_buffer{}, 0};
  auto gl = cudf::gather(t0v, lcol.view());
  auto gr = cudf::gather(t1v, rcol.view());
  auto a  = to_host<int32_t>(gl->view().column(0));
  auto b  = to_host<int32_t>(gr->view().column(0));
  CHECK(a == b);  // joined keys agree row by row
  std::sort(a.begin(), a.end());
  CHECK((a == std::vector<int32_t>{0, 2, 2, 2, 2, 3}));
  // ---- chunked probing (hash_join.hpp:276-412): match counts {1,0,2,1,2} (join_tests.cpp:2445-2449), then the join of
  // left rows [2, 5) with indices that refer to the complete left table
  {
    auto ctx = hj.inner_join_match_context(t0v);
    std::vector<cudf::size_type> counts(5);
    (void)hipMemcpy(counts.data(), ctx._match_counts->data(), 20, hipMemcpyDeviceToHost);
    CHECK((counts == std::vector<cudf::size_type>{1, 0, 2, 1, 2}));
    cudf::join_partition_context part{std::make_unique<cudf::join_match_context>(std::move(ctx)), 2, 5};
    auto [cl, cr] = hj.partitioned_inner_join(part);
    CHECK(cl->size() == 5);
    std::vector<cudf::size_type> hcl(5);
    (void)hipMemcpy(hcl.data(), cl->data(), 20, hipMemcpyDeviceToHost);
    CHECK(std::all_of(hcl.begin(), hcl.end(), [](auto x) { return x >= 2 && x < 5; }));
  }

  // ---- finalize_partitioned_full_join takes host_span<device_span<size_type const> const> (hash_join.hpp:433-441)
  {
    auto ctx = hj.full_join_match_context(t0v);
    cudf::join_partition_context p0{std::make_unique<cudf::join_match_context>(std::move(ctx)), 0, 5};
    auto [fl, fr] = hj.partitioned_full_join(p0);
    std::vector<cudf::device_span<cudf::size_type const>> lparts{cudf::device_span<cudf::size_type const>{*fl}};
    std::vector<cudf::device_span<cudf::size_type const>> rparts{cudf::device_span<cudf::size_type const>{*fr}};
    auto [tl, tr] = cudf::hash_join::finalize_partitioned_full_join(lparts, rparts, 5, 5);
    // left rows {3,1,2,0,2} vs right {2,2,0,4,3}: 6 matched pairs + left row 1 (key 1) alone + right row 3 (key 4) alone
    CHECK(tl->size() == 8 && tr->size() == 8);
  }
  // ---- hash_partition: num_partitions + 1 offsets, the last one the row count (partitioning.hpp:84-101); a keys table
  {
    auto [pt, offs] = cudf::hash_partition(t0v, std::vector<cudf::size_type>{0}, 3);
    CHECK(offs.size() == 4 && offs[0] == 0 && offs[3] == 5 && pt->num_rows() == 5);
    auto [pt2, offs2] = cudf::hash_partition(t0v, t1v, 3);
    CHECK(offs2.size() == 4 && offs2[3] == 5);
    bool threw = false;
    try {
      (void)cudf::hash_partition(t0v, std::vector<cudf::size_type>{-1}, 3);
    } catch (std::out_of_range const&) {
      threw = true;
    }
    CHECK(threw);
  }

  // ---- Arrow C Data Interface: export the groupby result to host Arrow memory and import it again (interop.hpp)
  {
    std::vector<cudf::column_view> out_cols{result.first->view().column(0), result.second[0].results[0]->view()};
    cudf::table_view out_tv{out_cols};
    auto schema = cudf::to_arrow_schema(out_tv, {cudf::column_metadata{"k"}, cudf::column_metadata{"sum"}});
    auto arr    = cudf::to_arrow_host(out_tv);
    CHECK(std::string{schema->format} == "+s" && schema->n_children == 2);
    CHECK(std::string{schema->children[0]->format} == "i" && std::string{schema->children[1]->format} == "l");
    CHECK(arr->device_type == ARROW_DEVICE_CPU && arr->array.length == 3 && arr->array.n_children == 2);
    auto back = cudf::from_arrow(schema.get(), &arr->array);
    CHECK(back->num_rows() == 3 && back->num_columns() == 2);
    auto k2 = to_host<int32_t>(back->view().column(0));
    auto s2 = to_host<int64_t>(back->view().column(1));
    CHECK(k2 == k && s2 == s);
  }
  // ---- device export and re-import without a host bounce (to_arrow_device, interop.hpp:500-610): the array's buffers are the
  // columns' device buffers; from_arrow_device views them again
  {
    std::vector<cudf::column_view> out_cols{result.first->view().column(0), result.second[0].results[0]->view()};
    cudf::table_view out_tv{out_cols};
    auto schema = cudf::to_arrow_schema(out_tv, {cudf::column_metadata{"k"}, cudf::column_metadata{"sum"}});
    auto darr   = cudf::to_arrow_device(out_tv);
    CHECK(darr->device_type == ARROW_DEVICE_ROCM && darr->sync_event != nullptr && darr->array.n_children == 2);
    CHECK(darr->array.children[0]->buffers[1] == out_tv.column(0).head());  // zero copy
    auto tv = cudf::from_arrow_device(schema.get(), darr.get());
    CHECK(tv->view.num_rows() == 3 && tv->view.column(1).data<int64_t>() == out_tv.column(1).data<int64_t>());
    // the owning form: the table's buffers move into the array and stay valid until its release callback runs
    auto copy  = std::make_unique<cudf::table>(out_tv);
    auto owned = cudf::to_arrow_device(std::move(*copy));
    copy.reset();
    auto tv2 = cudf::from_arrow_device(schema.get(), owned.get());
    CHECK(to_host<int32_t>(tv2->view.column(0)) == k && to_host<int64_t>(tv2->view.column(1)) == s);
  }
  // ---- zero-copy import of DEVICE Arrow data (ArrowDeviceArray, device_type ARROW_DEVICE_ROCM): the views alias the
  // producer's memory; a validity bitmap and an array offset are honoured
  {
    dev_vec<uint32_t> vmask{{0x000003dfu}};  // row 5 of the values is null
    ArrowSchema sk{"i", "k", nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    ArrowSchema sv{"i", "v", nullptr, ARROW_FLAG_NULLABLE, 0, nullptr, nullptr, nullptr, nullptr};
    ArrowSchema* sch[2] = {&sk, &sv};
    ArrowSchema st{"+s", "", nullptr, 0, 2, sch, nullptr, nullptr, nullptr};
    const void* kb[2] = {nullptr, keys.p};
    const void* vb[2] = {vmask.p, vals.p};
    ArrowArray ak{10, 0, 0, 2, 0, kb, nullptr, nullptr, nullptr, nullptr};
    ArrowArray av{10, 1, 0, 2, 0, vb, nullptr, nullptr, nullptr, nullptr};
    ArrowArray* ach[2] = {&ak, &av};
    const void* sb[1] = {nullptr};
    ArrowDeviceArray da{};
    da.array       = ArrowArray{10, 0, 0, 1, 2, sb, ach, nullptr, nullptr, nullptr};
    da.device_id   = 0;
    da.device_type = ARROW_DEVICE_ROCM;
    auto tv = cudf::from_arrow_device(&st, &da);
    CHECK(tv->view.num_columns() == 2 && tv->view.num_rows() == 10);
    CHECK(tv->view.column(0).data<int32_t>() == keys.p);  // zero copy
    CHECK(tv->view.column(1).null_count() == 1);
    cudf::groupby::groupby g2(cudf::table_view({tv->view.column(0)}));
    std::vector<cudf::groupby::aggregation_request> r2(1);
    r2[0].values = tv->view.column(1);
    r2[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_aggregation>());
    auto out = g2.aggregate(r2);
    auto k3  = to_host<int32_t>(out.first->view().column(0));
    auto s3  = to_host<int64_t>(out.second[0].results[0]->view());
    std::map<int32_t, int64_t> m3;
    for (int i = 0; i < 3; ++i) m3[k3[i]] = s3[i];
    CHECK(m3[1] == 9 && m3[2] == 14 && m3[3] == 17);  // key 2 loses row 5 (value 5): 19 - 5
    da.device_type = ARROW_DEVICE_CPU;
    bool threw = false;
    try {
      (void)cudf::from_arrow_device(&st, &da);
    } catch (std::invalid_argument const&) {
      threw = true;
    }
    CHECK(threw);
  }
  std::puts("api_compat OK");
  return 0;
}
