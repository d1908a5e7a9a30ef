// SPDX-License-Identifier: Apache-2.0
// Host side of cudf::hash_join and cudf::inner_join / left_join / full_join.
// Reference counterparts: cpp/src/join/join.cu:30-118 (build on the smaller side, swap results),
// cpp/src/join/hash_join/hash_join.cu:32-59 (is_trivial_join, validate_hash_join_probe), :113-149 (ctor/build),
// retrieve_impl.cuh:169-222 (probe entry), size_impl.cuh:26-61 (size), join_utils.cu:45-221 (trivial left join,
// full-join complement), join_common_utils.hpp:25-32 (load-factor check).
#include "engine.hpp"
#include "../common/profiler.hpp"

#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>
#include <optional>
#include <cmath>
#include <cstdlib>
#include <stdexcept>

namespace cudf {
namespace detail {

using join::join_args;

namespace {
double checked_load_factor(double load_factor)
{
  CUDF_EXPECTS(load_factor > 0, "Invalid load factor: must be greater than 0.", std::invalid_argument);
  CUDF_EXPECTS(load_factor <= 1, "Invalid load factor: must be less than or equal to 1.", std::invalid_argument);
  return load_factor;
}

bool is_trivial_join(table_view const& left, table_view const& right, join_kind kind)
{
  if (left.is_empty() || right.is_empty()) return true;
  if (kind == join_kind::LEFT_JOIN && left.num_rows() == 0) return true;
  if (kind == join_kind::INNER_JOIN && (left.num_rows() == 0 || right.num_rows() == 0)) return true;
  return false;
}

bool same_types(table_view const& a, table_view const& b)
{
  if (a.num_columns() != b.num_columns()) return false;
  for (size_type i = 0; i < a.num_columns(); ++i)
    if (a.column(i).type() != b.column(i).type()) return false;
  return true;
}

bool is_single64(table_view const& t)
{
  if (t.num_columns() != 1) return false;
  auto const id  = t.column(0).type().id();
  auto const cls = class_of(id);
  return size_of_id(id) == 8 && (cls == CLS_SINT || cls == CLS_UINT);
}

// 8 or 4: the table is ONE key column of that width that the partitioned joins take - an integer (timestamps, durations and
// decimals included; 4-byte keys are widened by the first scatter level) or, for the radix join only, a float whose NORMALISED bits
// (-0 -> +0, one NaN: the row equality of common/device_table.hpp) serve as the 8-byte key; 0 otherwise
int single_int_width(table_view const& t)
{
  if (t.num_columns() != 1) return 0;
  auto const id  = t.column(0).type().id();
  auto const cls = class_of(id);
  auto const w   = size_of_id(id);
  if (cls == CLS_F64) return 8;
  if (cls == CLS_F32) return 4;
  return (cls == CLS_SINT || cls == CLS_UINT) && (w == 8 || w == 4) ? static_cast<int>(w) : 0;
}
bool is_float_key(table_view const& t)
{
  auto const cls = t.num_columns() == 1 ? class_of(t.column(0).type().id()) : CLS_NONE;
  return cls == CLS_F32 || cls == CLS_F64;
}
// the table is TWO 4-byte integer key columns: the radix join packs them into its 8-byte key
bool is_two_int32(table_view const& t)
{
  if (t.num_columns() != 2) return false;
  for (int c = 0; c < 2; ++c) {
    auto const id  = t.column(c).type().id();
    auto const cls = class_of(id);
    if (!(cls == CLS_SINT || cls == CLS_UINT) || size_of_id(id) != 4) return false;
  }
  return true;
}
// TWO key columns of 4 or 8 bytes each (integers of any kind, floats), other than the two 4-byte integers is_two_int32 packs into one
// word: the radix join carries them as two 8-byte key words (round 4)
bool is_two_word_key(table_view const& t)
{
  if (t.num_columns() != 2 || is_two_int32(t)) return false;
  for (int c = 0; c < 2; ++c) {
    auto const id  = t.column(c).type().id();
    auto const cls = class_of(id);
    auto const w   = size_of_id(id);
    if (!((cls == CLS_SINT || cls == CLS_UINT || cls == CLS_F32 || cls == CLS_F64) && (w == 4 || w == 8))) return false;
  }
  return true;
}
int32_t float_class_of(column_view const& c)
{
  auto const cls = class_of(c.type().id());
  return (cls == CLS_F32 || cls == CLS_F64) ? static_cast<int32_t>(cls) : 0;
}
uint64_t const* key_bytes(column_view const& c, int width)
{
  return reinterpret_cast<uint64_t const*>(c.head<uint8_t>() + static_cast<std::size_t>(c.offset()) * static_cast<std::size_t>(width));
}

int64_t env_flag(char const* name, int64_t dflt)
{
  char const* v = std::getenv(name);
  return v ? std::atoll(v) : dflt;
}

template <typename T>
struct dev_scalar {  // one T in device memory, stream ordered
  rmm::device_buffer buf;
  hipStream_t s;
  dev_scalar(T init, hipStream_t stream) : buf{sizeof(T), stream, cudf::get_current_device_resource_ref()}, s{stream}
  {
    static_assert(sizeof(T) <= 8);
    T host = init;
    CUDF_HIP_TRY(hipMemcpyAsync(buf.data(), &host, sizeof(T), hipMemcpyHostToDevice, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
  }
  T* ptr() { return static_cast<T*>(buf.data()); }
  T value()
  {
    T host{};
    CUDF_HIP_TRY(hipMemcpyAsync(&host, buf.data(), sizeof(T), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    return host;
  }
};
}  // namespace

struct second_key {  // the second key column: of two packed 4-byte columns (radix_scatter_args::keys2) or of a two-word key (key2)
  int32_t width{0}, cls{0};  // two-word keys
  int32_t is_signed{0};
  uint32_t const* keys{nullptr};
  bitmask_type const* mask{nullptr};
  int64_t mask_offset{0};
};

class hash_join_impl {
 public:
  hash_join_impl(table_view const& right, bool has_nulls, null_equality compare_nulls, double load_factor,
                 stream_ref stream, rmm::device_async_resource_ref mr)
    : _right{right}, _has_nulls{has_nulls}, _nulls_equal{compare_nulls}, _is_empty{right.num_rows() == 0}, _mr{mr}
  {
    CUDF_EXPECTS(0 != right.num_columns(), "Hash join right table is empty", std::invalid_argument);
    load_factor = checked_load_factor(load_factor);
    if (_is_empty) return;
    _build_dev        = make_device_table(right);
    auto const rows   = static_cast<uint64_t>(right.num_rows());
    // `load_factor` is the caller's upper bound (reference default 0.5). The probe is bound by random memory requests
    // and a multiset walk ends at the first empty slot: 2.5 slots at load 0.5, 1.4 at 0.25 - so the table is kept
    // at <= 0.25 unless that would take more than 1/8 of the device memory (build_classic).
    // 8-byte slots {hash tag | row}, direct windowed probe. (The sliced table with inline keys and a radix-partitioned probe side
    // of rounds 1-3 - CUDF_AMD_JOIN_PARTITIONED - was superseded by the LDS radix join and removed in round 4.)
    bool const build_check_nulls = _has_nulls && cudf::has_nulls(right);
    bool const key64 = is_single64(right) && (!build_check_nulls || _nulls_equal != null_equality::EQUAL);
    // (one 4-byte integer key: only the partitioned joins take it - they widen it in their first scatter level; everything else
    // about such a table goes through the generic row comparator of the open-addressing table)
    _keyw         = (!build_check_nulls || _nulls_equal != null_equality::EQUAL) ? single_int_width(right) : 0;
    // (two 4-byte integer key columns: the radix join packs them into its 8-byte key; no dense table for those)
    _pack2        = (!build_check_nulls || _nulls_equal != null_equality::EQUAL) && is_two_int32(right);
    if (_pack2) _keyw = 4;
    _kw2          = (!build_check_nulls || _nulls_equal != null_equality::EQUAL) && is_two_word_key(right);
    if (_kw2) _keyw = static_cast<int>(size_of_id(right.column(0).type().id()));
    _key_signed   = _keyw != 0 && class_of(right.column(0).type().id()) == CLS_SINT;
    if (_kw2) decide_packed_words(right, stream);
    _key_class    = (_keyw != 0 && !_pack2 && (is_float_key(right) || _kw2)) ? float_class_of(right.column(0)) : 0;
    _key64        = key64;
    _classic_load = load_factor;
    // Dense build keys (one 8-byte integer key column, NULLs never match, valid values within a small range): a direct-address
    // table over [min, max] replaces the hash table - see engine.hpp. Decided from the exact minimum / maximum of the build keys.
    if (_keyw != 0 && !_pack2 && !_kw2 && _key_class == 0 && rows >= static_cast<uint64_t>(env_flag("CUDF_AMD_JOIN_DENSE_MIN_ROWS", 1 << 16)) && env_flag("CUDF_AMD_JOIN_DENSE", 1) != 0) {
      hipStream_t const s = stream.value();
      auto tmp            = cudf::get_current_device_resource_ref();
      bool const is_signed = class_of(right.column(0).type().id()) == CLS_SINT;
      rmm::device_buffer mm{2 * sizeof(uint64_t), s, tmp}, d_args{sizeof(join_args), s, tmp};
      join_args a = base_args(right, 0);
      join::launch_key_minmax(a, static_cast<join_args*>(d_args.data()), is_signed ? 1 : 0, static_cast<uint64_t*>(mm.data()), s);
      uint64_t h_mm[2] = {0, 0};
      CUDF_HIP_TRY(hipMemcpyAsync(h_mm, mm.data(), sizeof(h_mm), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));
      bool const any_valid = is_signed ? static_cast<int64_t>(h_mm[0]) <= static_cast<int64_t>(h_mm[1]) : h_mm[0] <= h_mm[1];
      uint64_t const width = h_mm[1] - h_mm[0];  // exact in two's complement for either ordering
      // at most 4 table entries (16 bytes) per build row, and at most 2^29 entries (2 GiB)
      if (any_valid && width < std::min<uint64_t>(std::max<uint64_t>(4 * rows, uint64_t{1} << 20), uint64_t{1} << 29)) {
        _dense_lo    = h_mm[0];
        _dense_range = width + 1;
        _dense_head  = rmm::device_buffer{_dense_range * sizeof(int32_t), s, mr};
        _dense_next  = rmm::device_buffer{rows * sizeof(int32_t), s, mr};
        CUDF_HIP_TRY(hipMemsetAsync(_dense_head.data(), 0xff, _dense_head.size(), s));
        // big tables: rows partitioned by key range, plain stores that stay in L2, uniqueness from a count (dense_part_kernels.hip)
        _dense = key64;  // (the direct passes over a dense table read 8-byte keys)
        bool part_built = false;
        try {
          part_built = try_dense_part_build(right, stream);
        } catch (std::bad_alloc const&) {  // (no room for the partition scratch: the atomic-exchange build needs none)
          CUDF_HIP_TRY(hipMemsetAsync(_dense_head.data(), 0xff, _dense_head.size(), s));
        }
        if (part_built) return;
        if (!key64) {  // a 4-byte key whose rows did not take the partitioned build: the hash table
          _dense_head = rmm::device_buffer{};
          _dense_next = rmm::device_buffer{};
        }
      }
      if (any_valid && _dense) {
        rmm::device_buffer dups{sizeof(int32_t), s, tmp};
        CUDF_HIP_TRY(hipMemsetAsync(dups.data(), 0, sizeof(int32_t), s));
        join_args b  = base_args(right, 0);
        b.dense_dups = static_cast<int32_t*>(dups.data());
        join::launch_dense_build(b, static_cast<join_args*>(d_args.data()), s);
        int32_t h_dups = 0;
        CUDF_HIP_TRY(hipMemcpyAsync(&h_dups, dups.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CUDF_HIP_TRY(hipStreamSynchronize(s));
        _dense_has_dups = h_dups != 0;
        if (_dense_has_dups) {  // row lists instead of chains (engine.hpp launch_dense_csr)
          std::size_t const entries = join::dense_csr_entries(_dense_range);
          _dense_head = rmm::device_buffer{entries * sizeof(int32_t), s, mr};
          CUDF_HIP_TRY(hipMemsetAsync(_dense_head.data(), 0, _dense_head.size(), s));
          rmm::device_buffer cursor{_dense_range * sizeof(int32_t), s, tmp}, sums{entries / 8192 * sizeof(uint32_t) + 16, s, tmp};
          CUDF_HIP_TRY(hipMemsetAsync(cursor.data(), 0, cursor.size(), s));
          join_args c = base_args(right, 0);
          join::launch_dense_csr(c, static_cast<join_args*>(d_args.data()), static_cast<int32_t*>(cursor.data()), static_cast<uint32_t*>(sums.data()), s);
          CUDF_HIP_TRY(hipStreamSynchronize(s));
        }
        return;
      }
    }
    // LDS radix join (engine.hpp): the build side partitioned into LDS-sized partitions; the open-addressing table in HBM is then
    // only built if a call needs it (left / full joins, match contexts, small probe sides, a probe side that overflows a region)
    if (_keyw != 0) {
      bool radix_built = false;
      try {
        radix_built = try_radix_build(right, stream);
      } catch (std::bad_alloc const&) {
        _rx_build = radix_side{};
      }
      if (radix_built) return;
    }
    build_classic(stream);
  }

 private:
  // The open-addressing multiset in HBM (slots of {hash tag | build row}, linear window + key-dependent stride): what every probe
  // used before the radix join and what everything but big inner joins on one 8-byte key still uses.
  void build_classic(stream_ref stream) const
  {
    auto const rows     = static_cast<uint64_t>(_right.num_rows());
    double const target_load = std::min(_classic_load, 0.01 * static_cast<double>(env_flag("CUDF_AMD_JOIN_MAX_LOAD_PCT", 25)));
    uint64_t capacity = static_cast<uint64_t>(std::ceil(static_cast<double>(rows) / target_load));
    if (capacity * 8 > (uint64_t{36} << 30)) capacity = static_cast<uint64_t>(std::ceil(static_cast<double>(rows) / _classic_load));
    capacity          = std::clamp<uint64_t>(capacity, rows + 1, (uint64_t{1} << 32) - 2);  // always one empty slot
    auto mr           = _mr;
    capacity          = std::min<uint64_t>((capacity + 1) / 2 * 2, ((uint64_t{1} << 32) - 1) / 2 * 2);  // even: 16-byte aligned slot pairs
    _capacity         = capacity;
    _table            = rmm::device_buffer{capacity * sizeof(uint64_t), stream.value(), mr};
    CUDF_HIP_TRY(hipMemsetAsync(_table.data(), 0xff, capacity * sizeof(uint64_t), stream.value()));
    join_args a = base_args(_right, 0);
    rmm::device_buffer d_args{sizeof(join_args), stream.value(), cudf::get_current_device_resource_ref()};
    rmm::device_buffer skip{sizeof(uint64_t) * join::BUILD_SKIP_ENTRIES, stream.value(), cudf::get_current_device_resource_ref()};
    CUDF_HIP_TRY(hipMemsetAsync(skip.data(), 0xff, skip.size(), stream.value()));  // no home slot is 2^40 - 1
    a.build_skip = static_cast<uint64_t*>(skip.data());
    join::launch_build(a, static_cast<join_args*>(d_args.data()), stream.value());
    CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));  // d_args goes out of scope; build is done for probes on any stream
    _classic_built = true;
  }
  // Two integer key columns whose build-side ranges need at most 63 bits together travel through the radix join as ONE word (engine.hpp
  // radix_scatter_args::pack): the exact minimum / maximum of both build columns decide (two passes over the build keys, one read-back).
  void decide_packed_words(table_view const& right, stream_ref stream)
  {
    if (env_flag("CUDF_AMD_JOIN_PACK_RANGE", 1) == 0 || right.num_rows() < 2) return;
    for (int c = 0; c < 2; ++c) {
      auto const cls = class_of(right.column(c).type().id());
      if (cls != CLS_SINT && cls != CLS_UINT) return;
    }
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    rmm::device_buffer mm{4 * sizeof(uint64_t), s, tmp}, d_args{2 * sizeof(join_args), s, tmp};
    bool sgn[2];
    for (int c = 0; c < 2; ++c) {
      sgn[c]      = class_of(right.column(c).type().id()) == CLS_SINT;
      join_args a = base_args(right, 0);
      a.build.col[0] = _build_dev.col[c];
      join::launch_key_minmax(a, static_cast<join_args*>(d_args.data()) + c, sgn[c] ? 1 : 0, static_cast<uint64_t*>(mm.data()) + 2 * c, s);
    }
    uint64_t h[4] = {0, 0, 0, 0};
    CUDF_HIP_TRY(hipMemcpyAsync(h, mm.data(), sizeof(h), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    int bits[2];
    for (int c = 0; c < 2; ++c) {
      bool const any_valid = sgn[c] ? static_cast<int64_t>(h[2 * c]) <= static_cast<int64_t>(h[2 * c + 1]) : h[2 * c] <= h[2 * c + 1];
      if (!any_valid) return;
      uint64_t const range = h[2 * c + 1] - h[2 * c];
      bits[c]              = 0;
      while (bits[c] < 64 && (range >> bits[c]) != 0) ++bits[c];
    }
    if (bits[0] + bits[1] > 63) return;
    _kw2_packed = true;
    _pk_lo[0] = h[0], _pk_lo[1] = h[2];
    _pk_range[0] = h[1] - h[0], _pk_range[1] = h[3] - h[2];
    _pk_bits1 = bits[1];
  }
  [[nodiscard]] bool two_words() const { return _kw2 && !_kw2_packed; }
  // ---- LDS radix join, build side (engine.hpp): two ring-scatter levels of {key, row id}; false if the build side does not take it
  // (too small or too big for 2048 ... 32768 partitions, a region or a partition that overflows: heavily duplicated keys)
  struct radix_side {
    rmm::device_buffer key1, row1, cnt1, key2, row2, cnt2;
    rmm::device_buffer key1w, key2w;  // two-word keys: the second word of level 1 / level 2
    int64_t cap2{0};
    int32_t slices2{0};
  };
  second_key second_of(table_view const& t) const
  {
    second_key k2{};
    if (_pack2 || _kw2) {
      auto const& c = t.column(1);
      k2.width      = _kw2 ? static_cast<int32_t>(size_of_id(c.type().id())) : 0;
      k2.cls        = _kw2 ? float_class_of(c) : 0;
      k2.is_signed  = class_of(c.type().id()) == CLS_SINT ? 1 : 0;
      k2.keys       = reinterpret_cast<uint32_t const*>(key_bytes(c, _kw2 ? k2.width : 4));
      k2.mask       = (_has_nulls && c.has_nulls()) ? c.null_mask() : nullptr;
      k2.mask_offset = c.offset();
    }
    return k2;
  }
  bool radix_partition(uint64_t const* keys, bitmask_type const* mask, int64_t mask_offset, int64_t nrows, int64_t valid_rows, radix_side& out,
                       stream_ref stream, rmm::device_async_resource_ref mr2, second_key k2, bool keep_outside = false) const
  {
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    int64_t const P1 = 128, P2 = _rx_nparts / P1, S1 = std::clamp<int64_t>((nrows + 4095) / 4096, 1, 256), slices2 = 4;
    auto cap_for = [](double mean) {  // six sigmas of a Poisson cell on top of its mean, whole 32-record granules
      return (static_cast<int64_t>(mean + 6.0 * std::sqrt(std::max(mean, 1.0)) + 64.0) + 31) / 32 * 32;
    };
    int64_t const tiles = (nrows + 4095) / 4096, wg_rows = std::min<int64_t>(nrows, (tiles + S1 - 1) / S1 * 4096);
    int64_t const cap1  = cap_for(static_cast<double>(wg_rows) / static_cast<double>(P1));
    int64_t const cap2  = cap_for(static_cast<double>(valid_rows) / static_cast<double>(_rx_nparts * slices2) * 1.02);
    rmm::device_buffer ovf{sizeof(int32_t), s, tmp}, d_args{sizeof(join::radix_scatter_args), s, tmp};
    CUDF_HIP_TRY(hipMemsetAsync(ovf.data(), 0, sizeof(int32_t), s));
    out.key1 = rmm::device_buffer{static_cast<std::size_t>(P1 * S1 * cap1) * 8, s, tmp};
    out.row1 = rmm::device_buffer{static_cast<std::size_t>(P1 * S1 * cap1) * 4, s, tmp};
    out.cnt1 = rmm::device_buffer{static_cast<std::size_t>(P1 * S1) * 4, s, tmp};
    if (two_words()) {
      out.key1w = rmm::device_buffer{static_cast<std::size_t>(P1 * S1 * cap1) * 8, s, tmp};
      out.key2w = rmm::device_buffer{static_cast<std::size_t>(_rx_nparts * slices2 * cap2) * 8, s, mr2};
    }
    out.key2 = rmm::device_buffer{static_cast<std::size_t>(_rx_nparts * slices2 * cap2) * 8, s, mr2};
    out.row2 = rmm::device_buffer{static_cast<std::size_t>(_rx_nparts * slices2 * cap2) * 4, s, mr2};
    out.cnt2 = rmm::device_buffer{static_cast<std::size_t>(_rx_nparts * slices2) * 4, s, mr2};
    out.cap2    = cap2;
    out.slices2 = static_cast<int32_t>(slices2);
    CUDF_HIP_TRY(hipMemsetAsync(out.cnt1.data(), 0, out.cnt1.size(), s));
    CUDF_HIP_TRY(hipMemsetAsync(out.cnt2.data(), 0, out.cnt2.size(), s));
    auto log2i = [](int64_t v) { int l = 0; while ((int64_t{1} << l) < v) ++l; return l; };
    join::radix_scatter_args a1{};
    a1.level        = 1;
    a1.keys         = keys;
    a1.key_width    = _keyw;
    a1.key_signed   = _key_signed ? 1 : 0;
    a1.key_class    = _key_class;
    a1.keys2        = _kw2 ? nullptr : k2.keys;
    a1.kw           = two_words() ? 2 : 1;
    a1.key2         = _kw2 ? static_cast<void const*>(k2.keys) : nullptr;
    a1.key2_width   = k2.width;
    a1.key2_class   = k2.cls;
    a1.out_key1     = two_words() ? static_cast<uint64_t*>(out.key1w.data()) : nullptr;
    if (_kw2_packed) {
      a1.pack              = 1;
      a1.pack_bits1        = _pk_bits1;
      a1.pack_keep_outside = keep_outside ? 1 : 0;
      a1.key2_signed       = k2.is_signed;
      a1.pack_lo0 = _pk_lo[0], a1.pack_lo1 = _pk_lo[1];
      a1.pack_range0 = _pk_range[0], a1.pack_range1 = _pk_range[1];
    }
    a1.mask2        = k2.mask;
    a1.mask2_offset = k2.mask_offset;
    a1.mask         = mask;
    a1.mask_offset  = mask_offset;
    a1.nrows        = nrows;
    a1.P            = static_cast<int32_t>(P1);
    int const ring_log2 = two_words() ? 12 : 13;  // (two-word keys: half as many ring slots, 2048-row tiles)
    a1.capl         = ring_log2 - log2i(P1);
    a1.shift        = 64 - log2i(P1);
    a1.slices       = static_cast<int32_t>(S1);
    a1.out_key      = static_cast<uint64_t*>(out.key1.data());
    a1.out_row      = static_cast<uint32_t*>(out.row1.data());
    a1.region_cap   = cap1;
    a1.region_count = static_cast<int32_t*>(out.cnt1.data());
    a1.overflow     = static_cast<int32_t*>(ovf.data());
    join::launch_radix_scatter(a1, static_cast<join::radix_scatter_args*>(d_args.data()), s);
    join::radix_scatter_args a2{};
    a2.level           = 2;
    a2.in_key          = a1.out_key;
    a2.kw              = a1.kw;
    a2.in_key1         = a1.out_key1;
    a2.out_key1        = two_words() ? static_cast<uint64_t*>(out.key2w.data()) : nullptr;
    a2.in_row          = a1.out_row;
    a2.in_region_count = a1.region_count;
    a2.in_region_cap   = cap1;
    a2.in_slices       = static_cast<int32_t>(S1);
    a2.nseg            = static_cast<int32_t>(P1);
    a2.P               = static_cast<int32_t>(P2);
    a2.capl            = ring_log2 - log2i(P2);
    a2.shift           = 64 - log2i(P1) - log2i(P2);
    a2.slices          = static_cast<int32_t>(slices2);
    a2.out_key         = static_cast<uint64_t*>(out.key2.data());
    a2.out_row         = static_cast<uint32_t*>(out.row2.data());
    a2.region_cap      = cap2;
    a2.region_count    = static_cast<int32_t*>(out.cnt2.data());
    a2.overflow        = a1.overflow;
    rmm::device_buffer d_args2{sizeof(join::radix_scatter_args), s, tmp};
    join::launch_radix_scatter(a2, static_cast<join::radix_scatter_args*>(d_args2.data()), s);
    int32_t h_ovf = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&h_ovf, ovf.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    out.key1  = rmm::device_buffer{};  // (the first level's regions go back to the pool)
    out.key1w = rmm::device_buffer{};
    out.row1 = rmm::device_buffer{};
    out.cnt1 = rmm::device_buffer{};
    return h_ovf == 0;
  }
  bool try_radix_build(table_view const& right, stream_ref stream)
  {
    if (env_flag("CUDF_AMD_JOIN_RADIX", 1) == 0) return false;
    auto const& col     = right.column(0);
    int64_t const rows  = right.num_rows();
    // (two packed columns: the rows with a NULL in either are dropped; their number is at least the larger of the two null counts)
    int64_t const valid = rows - std::max<int64_t>(col.nullable() ? col.null_count() : 0, ((_pack2 || _kw2) && right.column(1).nullable()) ? right.column(1).null_count() : 0);
    // partitions: a power of two with at most ~3500 build rows each (LDS tables of 8192 slots: load <= 0.43), 128 x (16 ... 256)
    int64_t const part_rows = std::clamp<int64_t>(env_flag("CUDF_AMD_JOIN_RADIX_PART_ROWS", 3500), 256, 3500);
    int64_t nparts = 2048;
    while (nparts < (two_words() ? 16384 : 32768) && valid > nparts * part_rows) nparts <<= 1;  // (two-word keys: second-level rings of at least 32 slots)
    if (valid < env_flag("CUDF_AMD_JOIN_RADIX_MIN_BUILD", 3 << 20) || valid > nparts * 3500) return false;
    _rx_nparts = static_cast<int32_t>(nparts);
    hipStream_t const s = stream.value();
    if (!radix_partition(key_bytes(col, _keyw), (_has_nulls && col.has_nulls()) ? col.null_mask() : nullptr, col.offset(), rows, valid, _rx_build, stream, _mr,
                         second_of(right)))
      return false;
    // every partition must fit its LDS table
    auto tmp = cudf::get_current_device_resource_ref();
    rmm::device_buffer mx{sizeof(int32_t), s, tmp};
    CUDF_HIP_TRY(hipMemsetAsync(mx.data(), 0, sizeof(int32_t), s));
    join::launch_radix_partition_max(static_cast<int32_t const*>(_rx_build.cnt2.data()), _rx_nparts, _rx_build.slices2, static_cast<int32_t*>(mx.data()), s);
    int32_t h_max = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&h_max, mx.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    if (h_max > RADIX_FILL_LIMIT) {
      _rx_build = radix_side{};
      return false;
    }
    _radix = true;
    return true;
  }
  static constexpr int32_t RADIX_TABLE_SLOTS = 8192, RADIX_FILL_LIMIT = 5200;

  // inner join of a big probe side against the radix-partitioned build side; nullopt: this probe side does not take it.
  // size_only != nullptr: the count pass only - *size_only = the number of pairs, the pair holds no vectors (inner / left_join_size
  // against a radix build: no second copy of the build side in an open-addressing table just to count)
  std::optional<join_index_pair> radix_probe(table_view const& left, stream_ref stream, rmm::device_async_resource_ref mr, int64_t row_base,
                                             bool left_join = false, std::size_t* size_only = nullptr) const
  {
    auto const& col = left.column(0);
    bool const probe_nulls = _has_nulls && col.has_nulls();
    bool const second_nulls = (_pack2 || _kw2) && _has_nulls && left.num_columns() == 2 && left.column(1).has_nulls();
    if ((_kw2 ? !is_two_word_key(left) : _pack2 ? !is_two_int32(left) : single_int_width(left) != _keyw) || ((probe_nulls || second_nulls) && _nulls_equal == null_equality::EQUAL))
      return std::nullopt;
    // (left join of two packed columns with NULLs: the rows dropped for a NULL in either column are not one column's null count)
    if (left_join && (_pack2 || _kw2) && (probe_nulls || second_nulls)) return std::nullopt;
    int64_t const rows  = left.num_rows();
    int64_t const valid = rows - std::max<int64_t>(col.nullable() ? col.null_count() : 0, second_nulls ? left.column(1).null_count() : 0);
    if (rows < env_flag("CUDF_AMD_JOIN_RADIX_MIN_PROBE", 8 << 20) || rows > (int64_t{1} << 31) - 1) return std::nullopt;
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    radix_side probe;
    if (!radix_partition(key_bytes(col, _keyw), probe_nulls ? col.null_mask() : nullptr, col.offset(), rows, valid, probe, stream, tmp, second_of(left), left_join))
      return std::nullopt;
    rmm::device_buffer counts{(static_cast<std::size_t>(_rx_nparts) + 1) * sizeof(unsigned long long), s, tmp}, ovf{sizeof(int32_t), s, tmp},
      d_args{sizeof(join::radix_join_args), s, tmp};
    CUDF_HIP_TRY(hipMemsetAsync(ovf.data(), 0, sizeof(int32_t), s));
    join::radix_join_args a{};
    a.kw     = two_words() ? 2 : 1;
    a.b_key1 = two_words() ? static_cast<uint64_t const*>(_rx_build.key2w.data()) : nullptr;
    a.p_key1 = two_words() ? static_cast<uint64_t const*>(probe.key2w.data()) : nullptr;
    a.b_key = static_cast<uint64_t const*>(_rx_build.key2.data());
    a.b_row = static_cast<uint32_t const*>(_rx_build.row2.data());
    a.b_count = static_cast<int32_t const*>(_rx_build.cnt2.data());
    a.b_cap = _rx_build.cap2;
    a.b_slices = _rx_build.slices2;
    a.p_key = static_cast<uint64_t const*>(probe.key2.data());
    a.p_row = static_cast<uint32_t const*>(probe.row2.data());
    a.p_count = static_cast<int32_t const*>(probe.cnt2.data());
    a.p_cap = probe.cap2;
    a.p_slices = probe.slices2;
    a.nparts = _rx_nparts;
    a.cap = RADIX_TABLE_SLOTS;
    a.fill_limit = RADIX_FILL_LIMIT;
    a.pair_counts = static_cast<unsigned long long*>(counts.data());
    a.probe_row_base = row_base;
    a.overflow = static_cast<int32_t*>(ovf.data());
    a.left = left_join ? 1 : 0;
    a.stage_cap = probe.cap2 * probe.slices2;  // (as many pairs as the partition has room for probe rows: enough unless keys repeat a lot)
    rmm::device_buffer stage{static_cast<std::size_t>(_rx_nparts) * static_cast<std::size_t>(a.stage_cap) * sizeof(uint64_t), s, tmp};
    a.stage = static_cast<uint64_t*>(stage.data());
    join::launch_radix_join(a, static_cast<join::radix_join_args*>(d_args.data()), false, s);
    join_args sc{};
    sc.block_counts = a.pair_counts;
    sc.nblocks      = _rx_nparts;
    join::launch_scan(sc, s);
    unsigned long long total = 0;
    int32_t h_ovf            = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&total, a.pair_counts + _rx_nparts, sizeof(total), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipMemcpyAsync(&h_ovf, ovf.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    if ((h_ovf & 3) != 0) return std::nullopt;
    // (left join: the probe rows with a NULL key never entered a partition; their {row, JoinNoMatch} pairs follow the partitions')
    unsigned long long const null_rows = (left_join && probe_nulls) ? static_cast<unsigned long long>(col.null_count()) : 0ull;
    unsigned long long const pairs_all = total + null_rows;
    if (size_only != nullptr) {
      *size_only = static_cast<std::size_t>(pairs_all);
      return join_index_pair{nullptr, nullptr};
    }
    CUDF_EXPECTS(pairs_all <= static_cast<unsigned long long>(std::numeric_limits<size_type>::max()),
                 "Join result exceeds the maximum column size; use the *_join_size API and chunk the probe side.", std::overflow_error);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(pairs_all), s, mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(pairs_all), s, mr);
    a.out_probe    = out_l->data();
    a.out_build    = out_r->data();
    a.out_capacity = pairs_all;
    rmm::device_buffer d_args2{sizeof(join::radix_join_args), s, tmp};
    join::launch_radix_emit_staged(a, static_cast<join::radix_join_args*>(d_args2.data()), s);
    if ((h_ovf & 4) != 0) join::launch_radix_join(a, static_cast<join::radix_join_args*>(d_args2.data()), true, s);
    if (null_rows != 0)  // (pair_counts[nparts] = the number of pairs of the partitions: the cursor the NULL rows append at)
      join::launch_radix_null_rows(col.null_mask(), col.offset(), rows, row_base, a.out_probe, a.out_build, pairs_all, a.pair_counts + _rx_nparts, s);
    // (no synchronisation: the scratch goes back to the pool in stream order, the caller reads the result on this stream)
    return join_index_pair{std::move(out_l), std::move(out_r)};
  }

  // ---- partitioned dense join (engine.hpp dense_part_args)
  struct dense_side {
    rmm::device_buffer recs, counts, ovf;
    int64_t cap{0};
    int32_t P{0}, S{0}, P_used{0};
  };
  // rows -> records {key - lo | row << 32} in regions (partition, slice); `expected` = the rows a slice may have to take
  void dense_partition(uint64_t const* keys, bitmask_type const* mask, int64_t mask_offset, int64_t nrows, dense_side& out, stream_ref stream) const
  {
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    auto log2c = [](uint64_t v) { int l = 0; while ((uint64_t{1} << l) < v) ++l; return l; };
    // slices of >= 1 MB (CUDF_AMD_JOIN_DENSE_PART_SLICE_LOG2: keys per slice, the tests shrink it), at most 256 of them
    int const shift     = std::max(static_cast<int>(env_flag("CUDF_AMD_JOIN_DENSE_PART_SLICE_LOG2", 18)), log2c(_dense_range) - 8);
    int64_t const P     = static_cast<int64_t>((_dense_range - 1) >> shift) + 1;
    int64_t Pring = 16;
    while (Pring < P) Pring <<= 1;
    // CUDF_AMD_JOIN_DENSE_PART_BLOCK=512: two workgroups of 512 threads per CU (8192 ring slots each) instead of one of 1024 - measured
    // 8 % slower on C3 (1.73 against 1.60 ms for both scatters), kept for the record (profiles/r3_c3_dense_part.txt)
    int64_t const rpt   = env_flag("CUDF_AMD_JOIN_DENSE_PART_RPT", 4) == 8 ? 8 : 4;
    int64_t const block = env_flag("CUDF_AMD_JOIN_DENSE_PART_BLOCK", 1024) == 512 ? 512 : 1024, tile_rows = block * (block == 1024 ? rpt : 4);
    int64_t const tiles = (nrows + tile_rows - 1) / tile_rows, S = std::clamp<int64_t>(tiles, 1, block == 512 ? 512 : 256);
    double const mean   = static_cast<double>(std::min<int64_t>(nrows, (tiles + S - 1) / S * tile_rows)) / static_cast<double>(P);
    int64_t const cap   = (static_cast<int64_t>(mean * 1.02 + 6.0 * std::sqrt(std::max(mean, 1.0)) + 64.0) + 31) / 32 * 32;
    out.recs   = rmm::device_buffer{static_cast<std::size_t>(Pring * S * cap) * sizeof(uint64_t), s, tmp};
    out.counts = rmm::device_buffer{static_cast<std::size_t>(Pring * S) * sizeof(int32_t), s, tmp};
    out.ovf    = rmm::device_buffer{sizeof(int32_t), s, tmp};
    out.cap    = cap;
    out.P      = static_cast<int32_t>(Pring);
    out.P_used = static_cast<int32_t>(P);
    out.S      = static_cast<int32_t>(S);
    CUDF_HIP_TRY(hipMemsetAsync(out.counts.data(), 0, out.counts.size(), s));
    CUDF_HIP_TRY(hipMemsetAsync(out.ovf.data(), 0, sizeof(int32_t), s));
    join::radix_scatter_args a{};
    a.level          = 1;
    a.keys           = keys;
    a.key_width      = _keyw;
    a.key_signed     = _key_signed ? 1 : 0;
    a.mask           = mask;
    a.mask_offset    = mask_offset;
    a.nrows          = nrows;
    a.P              = static_cast<int32_t>(Pring);
    a.capl           = (block == 512 ? 13 : 14) - log2c(static_cast<uint64_t>(Pring));  // 16384 (8192) ring slots of 8 bytes
    a.block          = static_cast<int32_t>(block);
    a.rpt            = static_cast<int32_t>(rpt);
    a.shift          = shift;
    a.slices         = static_cast<int32_t>(S);
    a.out_key        = static_cast<uint64_t*>(out.recs.data());
    a.region_cap     = cap;
    a.region_count   = static_cast<int32_t*>(out.counts.data());
    a.overflow       = static_cast<int32_t*>(out.ovf.data());
    a.dense          = 1;
    a.dense_lo       = _dense_lo;
    a.dense_range    = _dense_range;
    a.pending_budget = 64;
    rmm::device_buffer d_args{sizeof(join::radix_scatter_args), s, tmp};
    join::launch_radix_scatter(a, static_cast<join::radix_scatter_args*>(d_args.data()), s);
  }
  join::dense_part_args dense_part_args_of(dense_side const& side) const
  {
    join::dense_part_args a{};
    a.recs         = static_cast<uint64_t const*>(side.recs.data());
    a.region_count = static_cast<int32_t const*>(side.counts.data());
    a.region_cap   = side.cap;
    a.P            = side.P;
    a.P_used       = side.P_used;
    a.S            = side.S;
    a.head         = const_cast<int32_t*>(static_cast<int32_t const*>(_dense_head.data()));
    a.overflow     = static_cast<int32_t const*>(side.ovf.data());
    return a;
  }
  // the dense table of a big build side without atomics; false: a key repeats (or the rows cluster): the caller builds it the
  // other way (head is all -1 again)
  bool try_dense_part_build(table_view const& right, stream_ref stream)
  {
    auto const& col    = right.column(0);
    int64_t const rows = right.num_rows();
    if (env_flag("CUDF_AMD_JOIN_DENSE_PART", 1) == 0 || rows < env_flag("CUDF_AMD_JOIN_DENSE_PART_MIN_BUILD", 4 << 20) ||
        static_cast<int64_t>(_dense_range) < env_flag("CUDF_AMD_JOIN_DENSE_PART_MIN_RANGE", 8 << 20) || rows > (int64_t{1} << 31) - 1)
      return false;
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    dense_side side;
    dense_partition(key_bytes(col, _keyw), (_has_nulls && col.has_nulls()) ? col.null_mask() : nullptr, col.offset(), rows, side, stream);
    auto a = dense_part_args_of(side);
    // {filled table entries, records the scatter produced}: equal <=> no build key repeats
    rmm::device_buffer d_args{sizeof(join::dense_part_args), s, tmp}, tally{2 * sizeof(unsigned long long), s, tmp};
    CUDF_HIP_TRY(hipMemsetAsync(tally.data(), 0, 2 * sizeof(unsigned long long), s));
    auto* d_tally = static_cast<unsigned long long*>(tally.data());
    join::launch_dense_part_store(a, static_cast<join::dense_part_args*>(d_args.data()), s);
    join::launch_dense_count_filled(a.head, _dense_range, d_tally, s);
    join::launch_sum_region_counts(a.region_count, static_cast<int64_t>(side.P) * side.S, d_tally + 1, s);
    unsigned long long h_tally[2] = {0, 0};
    int32_t h_ovf                 = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(h_tally, d_tally, sizeof(h_tally), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipMemcpyAsync(&h_ovf, side.ovf.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    if (h_ovf != 0 || h_tally[0] != h_tally[1]) {
      CUDF_HIP_TRY(hipMemsetAsync(_dense_head.data(), 0xff, _dense_head.size(), s));
      return false;
    }
    _dense_has_dups = false;
    _dense_part     = true;
    return true;
  }
  // inner join against a dense unique table with the pairs in probe-row order (engine.hpp dense_stage_args); nullopt: not this call
  std::optional<join_index_pair> dense_ordered_probe(table_view const& left, stream_ref stream, rmm::device_async_resource_ref mr, int64_t row_base) const
  {
    auto const& col        = left.column(0);
    bool const probe_nulls = _has_nulls && col.has_nulls();
    int64_t const rows     = left.num_rows();
    if (single_int_width(left) != _keyw || _keyw == 0 || (probe_nulls && _nulls_equal == null_equality::EQUAL) ||
        rows < env_flag("CUDF_AMD_JOIN_DENSE_ORDERED_MIN_PROBE", 1 << 20) || rows + row_base > (int64_t{1} << 31) - 1)
      return std::nullopt;
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    join::dense_stage_args a{};
    // ~16K waves (64 per CU) of at least 4096 rows each
    a.wave_rows = std::max<int64_t>(4096, ((rows + 16383) / 16384 + 63) / 64 * 64);
    a.nwaves    = static_cast<int32_t>((rows + a.wave_rows - 1) / a.wave_rows);
    rmm::device_buffer stage{static_cast<std::size_t>(a.nwaves) * static_cast<std::size_t>(a.wave_rows) * sizeof(uint64_t), s, tmp},
      counts{(static_cast<std::size_t>(a.nwaves) + 1) * sizeof(unsigned long long), s, tmp}, d_args{sizeof(join::dense_stage_args), s, tmp};
    a.keys           = key_bytes(col, _keyw);
    a.key_width      = _keyw;
    a.key_signed     = _key_signed ? 1 : 0;
    a.mask           = probe_nulls ? col.null_mask() : nullptr;
    a.mask_offset    = col.offset();
    a.nrows          = rows;
    a.dense_lo       = _dense_lo;
    a.dense_range    = _dense_range;
    a.head           = static_cast<int32_t const*>(_dense_head.data());
    a.stage          = static_cast<uint64_t*>(stage.data());
    a.pair_counts    = static_cast<unsigned long long*>(counts.data());
    a.probe_row_base = row_base;
    join::launch_dense_probe_staged(a, static_cast<join::dense_stage_args*>(d_args.data()), s);
    join_args sc{};
    sc.block_counts = a.pair_counts;
    sc.nblocks      = a.nwaves;
    join::launch_scan(sc, s);
    unsigned long long total = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&total, a.pair_counts + a.nwaves, sizeof(total), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    CUDF_EXPECTS(total <= static_cast<unsigned long long>(std::numeric_limits<size_type>::max()),
                 "Join result exceeds the maximum column size; use the *_join_size API and chunk the probe side.", std::overflow_error);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(total), s, mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(total), s, mr);
    join::radix_join_args e{};  // (the copy of the staged pairs is the radix join's: one "partition" per wave)
    e.nparts       = a.nwaves;
    e.pair_counts  = a.pair_counts;
    e.stage        = a.stage;
    e.stage_cap    = a.wave_rows;
    e.out_probe    = out_l->data();
    e.out_build    = out_r->data();
    e.out_capacity = total;
    rmm::device_buffer d_args2{sizeof(join::radix_join_args), s, tmp};
    join::launch_radix_emit_staged(e, static_cast<join::radix_join_args*>(d_args2.data()), s);
    // (no synchronisation: the scratch goes back to the pool in stream order, the caller reads the result on this stream)
    return join_index_pair{std::move(out_l), std::move(out_r)};
  }

  // inner join of a big probe side against the dense unique table; nullopt: this probe side does not take it
  std::optional<join_index_pair> dense_part_probe(table_view const& left, stream_ref stream, rmm::device_async_resource_ref mr, int64_t row_base) const
  {
    auto const& col        = left.column(0);
    bool const probe_nulls = _has_nulls && col.has_nulls();
    int64_t const rows     = left.num_rows();
    if (single_int_width(left) != _keyw || (probe_nulls && _nulls_equal == null_equality::EQUAL) ||
        rows < env_flag("CUDF_AMD_JOIN_DENSE_PART_MIN_PROBE", 8 << 20) || rows > (int64_t{1} << 31) - 1)
      return std::nullopt;
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    dense_side side;
    dense_partition(key_bytes(col, _keyw), probe_nulls ? col.null_mask() : nullptr, col.offset(), rows, side, stream);
    auto a                     = dense_part_args_of(side);
    std::size_t const nregions = static_cast<std::size_t>(join::dense_part_grid());  // (one stage and one pair count per workgroup)
    a.stage_cap                = join::dense_part_regions_per_workgroup(side.P_used, side.S) * side.cap;
    rmm::device_buffer counts{(nregions + 1) * sizeof(unsigned long long), s, tmp},
      stage{nregions * static_cast<std::size_t>(a.stage_cap) * sizeof(uint64_t), s, tmp}, d_args{sizeof(join::dense_part_args), s, tmp};
    a.pair_counts    = static_cast<unsigned long long*>(counts.data());
    a.stage          = static_cast<uint64_t*>(stage.data());
    a.probe_row_base = row_base;
    join::launch_dense_part_lookup(a, static_cast<join::dense_part_args*>(d_args.data()), s);
    join_args sc{};
    sc.block_counts = a.pair_counts;
    sc.nblocks      = static_cast<int32_t>(nregions);
    join::launch_scan(sc, s);
    unsigned long long total = 0;
    int32_t h_ovf            = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&total, a.pair_counts + nregions, sizeof(total), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipMemcpyAsync(&h_ovf, side.ovf.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    if (h_ovf != 0) return std::nullopt;
    CUDF_EXPECTS(total <= static_cast<unsigned long long>(std::numeric_limits<size_type>::max()),
                 "Join result exceeds the maximum column size; use the *_join_size API and chunk the probe side.", std::overflow_error);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(total), s, mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(static_cast<std::size_t>(total), s, mr);
    join::radix_join_args e{};  // (the copy of the staged pairs is the radix join's)
    e.nparts       = static_cast<int32_t>(nregions);
    e.pair_counts  = a.pair_counts;
    e.stage        = a.stage;
    e.stage_cap    = a.stage_cap;
    e.out_probe    = out_l->data();
    e.out_build    = out_r->data();
    e.out_capacity = total;
    rmm::device_buffer d_args2{sizeof(join::radix_join_args), s, tmp};
    join::launch_radix_emit_staged(e, static_cast<join::radix_join_args*>(d_args2.data()), s);
    // (no synchronisation: the scratch goes back to the pool in stream order, the caller reads the result on this stream)
    return join_index_pair{std::move(out_l), std::move(out_r)};
  }

  void ensure_classic(stream_ref stream) const
  {
    std::lock_guard<std::mutex> g{_classic_mu};
    if (!_classic_built && !_is_empty && !_dense) build_classic(stream);
  }

 public:
  [[nodiscard]] std::size_t join_size(table_view const& left, join_kind kind, stream_ref stream,
                                      std::vector<uint8_t>* /*unused*/ = nullptr) const
  {
    validate_probe(left);
    if (kind == join_kind::INNER_JOIN) {
      if (is_trivial_join(left, _right, kind)) return 0;
    } else {
      if (_is_empty) return kind == join_kind::FULL_JOIN ? left.num_rows() : left.num_rows();
      if (is_trivial_join(left, _right, kind)) return kind == join_kind::FULL_JOIN ? _right.num_rows() : 0;
    }
    if (kind == join_kind::FULL_JOIN) {
      // needs the matched set of the build side: run the real join and take its size
      auto r = probe(left, kind, std::nullopt, stream, cudf::get_current_device_resource_ref());
      return r.first->size();
    }
    if (_radix) {  // the radix join's count pass (partition the probe side, count per partition): nothing else is built
      std::size_t pairs = 0;
      try {
        if (radix_probe(left, stream, cudf::get_current_device_resource_ref(), 0, kind == join_kind::LEFT_JOIN, &pairs).has_value()) return pairs;
      } catch (std::bad_alloc const&) {  // (its scratch did not fit: the tables below)
      }
    }
    ensure_classic(stream);
    join_args a = base_args(left, kind == join_kind::INNER_JOIN ? 0 : 1);
    rmm::device_buffer counts{(static_cast<std::size_t>(a.nblocks) + 1) * sizeof(unsigned long long), stream.value(),
                              cudf::get_current_device_resource_ref()};
    a.block_counts = static_cast<unsigned long long*>(counts.data());
    rmm::device_buffer cache{static_cast<std::size_t>(left.num_rows()) * sizeof(uint32_t), stream.value(),
                             cudf::get_current_device_resource_ref()};
    a.match_cache = static_cast<uint32_t*>(cache.data());
    rmm::device_buffer d_args{sizeof(join_args), stream.value(), cudf::get_current_device_resource_ref()};
    join::launch_count(a, static_cast<join_args*>(d_args.data()), stream.value());
    join::launch_scan(a, stream.value());
    unsigned long long total = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&total, a.block_counts + a.nblocks, sizeof(total), hipMemcpyDeviceToHost, stream.value()));
    CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
    return static_cast<std::size_t>(total);
  }

  // Per-left-row match counts (reference inner/left/full_join_match_context, hash_join.hpp:276-329).
  [[nodiscard]] std::unique_ptr<rmm::device_uvector<size_type>> match_counts(table_view const& left, join_kind kind,
                                                                             stream_ref stream,
                                                                             rmm::device_async_resource_ref mr) const
  {
    validate_probe(left);
    hipStream_t const s = stream.value();
    auto const n        = static_cast<std::size_t>(left.num_rows());
    auto out            = std::make_unique<rmm::device_uvector<size_type>>(n, s, mr);
    if (n == 0) return out;
    if (_is_empty) {  // nothing to match: 0 per row, or the row's own JoinNoMatch pair
      std::vector<size_type> h(n, kind == join_kind::INNER_JOIN ? 0 : 1);
      CUDF_HIP_TRY(hipMemcpyAsync(out->data(), h.data(), n * sizeof(size_type), hipMemcpyHostToDevice, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));
      return out;
    }
    ensure_classic(stream);
    join_args a = base_args(left, kind == join_kind::INNER_JOIN ? 0 : 1);
    rmm::device_buffer counts{(static_cast<std::size_t>(a.nblocks) + 1) * sizeof(unsigned long long), s,
                              cudf::get_current_device_resource_ref()};
    a.block_counts = static_cast<unsigned long long*>(counts.data());
    rmm::device_buffer cache{n * sizeof(uint32_t), s, cudf::get_current_device_resource_ref()};
    a.match_cache = static_cast<uint32_t*>(cache.data());
    a.row_counts  = out->data();
    rmm::device_buffer d_args{sizeof(join_args), s, cudf::get_current_device_resource_ref()};
    join::launch_count(a, static_cast<join_args*>(d_args.data()), s);
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    return out;
  }

  // Join of rows [start, end) of the context's left table; left indices refer to the complete table (reference
  // partitioned_inner/left/full_join, hash_join.hpp:353-412). The full-join form emits the probe side only.
  [[nodiscard]] join_index_pair partitioned(join_partition_context const& ctx, join_kind kind, stream_ref stream,
                                            rmm::device_async_resource_ref mr) const
  {
    CUDF_EXPECTS(ctx.left_table_context != nullptr, "join_partition_context has no match context", std::invalid_argument);
    CUDF_EXPECTS(ctx.left_table_context->_match_counts != nullptr, "join match context has no match counts",
                 std::invalid_argument);
    auto const& left = ctx.left_table_context->_left_table;
    CUDF_EXPECTS(ctx.left_start_idx >= 0 && ctx.left_start_idx <= ctx.left_end_idx && ctx.left_end_idx <= left.num_rows(),
                 "join partition is outside the bounds of the left table", std::invalid_argument);
    std::vector<column_view> cols;
    for (auto const& c : left) {
      size_type nulls = 0;
      if (c.nullable() && c.null_count() > 0)
        nulls = cudf::null_count(c.null_mask(), c.offset() + ctx.left_start_idx, c.offset() + ctx.left_end_idx, stream);
      cols.emplace_back(c.type(), ctx.left_end_idx - ctx.left_start_idx, c.head(), c.null_mask(), nulls,
                        c.offset() + ctx.left_start_idx);
    }
    auto const k = kind == join_kind::FULL_JOIN ? join_kind::LEFT_JOIN : kind;
    return probe(table_view{cols}, k, std::nullopt, stream, mr, ctx.left_start_idx);
  }

  [[nodiscard]] join_index_pair probe(table_view const& left, join_kind kind, std::optional<std::size_t> output_size,
                                      stream_ref stream, rmm::device_async_resource_ref mr, int64_t row_base = 0) const
  {
    validate_probe(left);
    auto empty_pair = [&] {
      return join_index_pair{std::make_unique<rmm::device_uvector<size_type>>(0, stream.value(), mr),
                             std::make_unique<rmm::device_uvector<size_type>>(0, stream.value(), mr)};
    };
    if (kind == join_kind::INNER_JOIN) {
      if (is_trivial_join(left, _right, kind)) return empty_pair();
    } else {
      if (_is_empty) return trivial_left(left, stream, mr, row_base);
      if (is_trivial_join(left, _right, kind)) {
        // left side has no rows: a full join still returns every right row unmatched
        if (kind == join_kind::FULL_JOIN && !left.is_empty() && _right.num_rows() > 0) return trivial_right(stream, mr);
        return empty_pair();
      }
    }
    int const k = kind == join_kind::INNER_JOIN ? 0 : kind == join_kind::LEFT_JOIN ? 1 : 2;
    // FULL join (round 4): the LEFT join of the partitioned paths (radix partitions, dense unique table) + the build rows nobody
    // matched - marked from the left join's build indices, appended by the complement kernel of the table path (reference
    // hash_join.cu full-join complement, join_utils.cu:45-221)
    if (k == 2 && env_flag("CUDF_AMD_JOIN_FULL_PARTITIONED", 1) != 0) {
      if (auto r = probe_partitioned_paths(left, 1, stream, cudf::get_current_device_resource_ref(), row_base); r.has_value()) {
        auto full = full_from_left(std::move(*r), stream, mr);
        if (output_size.has_value())
          CUDF_EXPECTS(*output_size == full.first->size(), "hash join: output_size does not match the number of matches", std::invalid_argument);
        return full;
      }
    }
    if (auto r = probe_partitioned_paths(left, k, stream, mr, row_base); r.has_value()) {
      if (output_size.has_value())
        CUDF_EXPECTS(*output_size == r->first->size(), "hash join: output_size does not match the number of matches", std::invalid_argument);
      return std::move(*r);
    }
    return probe_tables(left, kind, k, output_size, stream, mr, row_base);
  }

 private:
  // the pairs of a LEFT join + {JoinNoMatch, build row} for every build row that appears in none of them
  [[nodiscard]] join_index_pair full_from_left(join_index_pair lp, stream_ref stream, rmm::device_async_resource_ref mr) const
  {
    hipStream_t const s = stream.value();
    auto tmp            = cudf::get_current_device_resource_ref();
    std::size_t const P = lp.first->size(), nr = static_cast<std::size_t>(_right.num_rows());
    rmm::device_buffer matched{nr, s, tmp};
    CUDF_HIP_TRY(hipMemsetAsync(matched.data(), 0, nr, s));
    if (P > 0) join::launch_mark_matched(lp.second->data(), P, static_cast<uint8_t*>(matched.data()), s);
    rmm::device_uvector<size_type> comp_l{nr, s, tmp}, comp_r{nr, s, tmp};
    dev_scalar<unsigned long long> cursor{0ull, s};
    join_args a{};
    a.build         = _build_dev;
    a.build_matched = static_cast<uint8_t*>(matched.data());
    a.total         = cursor.ptr();
    a.out_probe     = comp_l.data();
    a.out_build     = comp_r.data();
    a.out_capacity  = nr;
    rmm::device_buffer d_args{sizeof(join_args), s, tmp};
    join::launch_complement(a, static_cast<join_args*>(d_args.data()), s);
    std::size_t const U = static_cast<std::size_t>(cursor.value());  // (synchronises)
    CUDF_EXPECTS(P + U <= static_cast<std::size_t>(std::numeric_limits<size_type>::max()),
                 "Join result exceeds the maximum column size; use the *_join_size API and chunk the probe side.", std::overflow_error);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(P + U, s, mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(P + U, s, mr);
    if (P > 0) {
      CUDF_HIP_TRY(hipMemcpyAsync(out_l->data(), lp.first->data(), P * sizeof(size_type), hipMemcpyDeviceToDevice, s));
      CUDF_HIP_TRY(hipMemcpyAsync(out_r->data(), lp.second->data(), P * sizeof(size_type), hipMemcpyDeviceToDevice, s));
    }
    if (U > 0) {
      CUDF_HIP_TRY(hipMemcpyAsync(out_l->data() + P, comp_l.data(), U * sizeof(size_type), hipMemcpyDeviceToDevice, s));
      CUDF_HIP_TRY(hipMemcpyAsync(out_r->data() + P, comp_r.data(), U * sizeof(size_type), hipMemcpyDeviceToDevice, s));
    }
    // (the left join's vectors and the scratch go back to the pool in stream order behind these copies)
    return join_index_pair{std::move(out_l), std::move(out_r)};
  }

  // ---- round 3: the paths in front of the tables' count / retrieve passes (engine.hpp), in the order they are tried. nullopt: none
  // of them takes this call (kind, key shape, size, a region that overflowed, no room for the scratch).
  std::optional<join_index_pair> probe_partitioned_paths(table_view const& left, int k, stream_ref stream, rmm::device_async_resource_ref mr,
                                                         int64_t row_base) const
  {
    // (the partitioned joins take scratch of ~3x the probe column - regions sized for the worst case, the pair stage: when the pool
    // cannot give it, the direct passes still can run)
    auto or_nothing = [](auto&& f) -> std::optional<join_index_pair> {
      try {
        return f();
      } catch (std::bad_alloc const&) {
        return std::nullopt;
      }
    };
    // big inner and left joins on one sparse integer key: both sides in LDS-sized partitions
    if (_radix && (k == 0 || k == 1)) {
      if (auto r = or_nothing([&] { return radix_probe(left, stream, mr, row_base, k == 1); }); r.has_value()) return r;
    }
    // a dense unique table. CUDF_AMD_JOIN_DENSE_PROBE: 1 (default) inner joins by the ordered direct probe (pairs in probe-row
    // order), 2 by probe rows partitioned by key range (~0.6 ms less at C3, pairs partition-major), 0 neither; left joins in one pass
    int64_t dense_probe     = env_flag("CUDF_AMD_JOIN_DENSE_PROBE", -1);
    bool const unique_table = (_dense && !_dense_has_dups) || _dense_part;
    if (dense_probe < 0) {
      // The ordered direct probe pays one random access into the table per probe key INSIDE the table's range (a key outside is
      // rejected by the range test for free): with 30 % of C3's probe keys in range it takes 3.3 ms, with 95 % in range 9.3 ms - the
      // probe rows partitioned by key range (L2-local lookups) take 6.0 ms + the pairs partition-major (profiles/r4_c3_inrange.txt).
      // A strided sample of the probe keys decides; the table must be large enough for the partitions to matter.
      dense_probe = 1;
      if (_dense_part && k == 0 && left.num_rows() >= env_flag("CUDF_AMD_JOIN_DENSE_PART_MIN_PROBE", 8 << 20) && single_int_width(left) == _keyw) {
        double const share = dense_inrange_share(left, stream);
        if (share > 0.01 * static_cast<double>(env_flag("CUDF_AMD_JOIN_DENSE_PART_INRANGE_PCT", 55))) dense_probe = 2;
      }
    }
    if (unique_table && k == 0 && dense_probe == 1) {
      if (auto r = or_nothing([&] { return dense_ordered_probe(left, stream, mr, row_base); }); r.has_value()) return r;
    }
    if (unique_table && k == 1 && dense_probe != 0) {
      if (auto r = or_nothing([&] { return dense_left_direct(left, stream, mr, row_base); }); r.has_value()) return r;
    }
    if (_dense_part && k == 0 && dense_probe != 0) {
      if (auto r = or_nothing([&] { return dense_part_probe(left, stream, mr, row_base); }); r.has_value()) return r;
    }
    return std::nullopt;
  }

  // share of the probe rows whose key is valid and inside the dense table's range (a sample of ~64K rows; one small kernel + read-back)
  [[nodiscard]] double dense_inrange_share(table_view const& left, stream_ref stream) const
  {
    auto const& col        = left.column(0);
    bool const probe_nulls = _has_nulls && col.has_nulls();
    hipStream_t const s    = stream.value();
    auto tmp               = cudf::get_current_device_resource_ref();
    join::dense_stage_args a{};
    a.keys        = key_bytes(col, _keyw);
    a.key_width   = _keyw;
    a.key_signed  = _key_signed ? 1 : 0;
    a.mask        = probe_nulls ? col.null_mask() : nullptr;
    a.mask_offset = col.offset();
    a.nrows       = left.num_rows();
    a.dense_lo    = _dense_lo;
    a.dense_range = _dense_range;
    rmm::device_buffer d_args{sizeof(join::dense_stage_args), s, tmp}, d_out{2 * sizeof(unsigned long long), s, tmp};
    join::launch_dense_inrange_sample(a, static_cast<join::dense_stage_args*>(d_args.data()), 1 << 16, static_cast<unsigned long long*>(d_out.data()), s);
    unsigned long long h[2] = {0, 0};
    CUDF_HIP_TRY(hipMemcpyAsync(h, d_out.data(), sizeof(h), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    return h[0] == 0 ? 0.0 : static_cast<double>(h[1]) / static_cast<double>(h[0]);
  }

  // LEFT join against a dense unique table: one pair per probe row, in order - a single pass (engine.hpp launch_dense_left_direct)
  std::optional<join_index_pair> dense_left_direct(table_view const& left, stream_ref stream, rmm::device_async_resource_ref mr, int64_t row_base) const
  {
    auto const& col        = left.column(0);
    bool const probe_nulls = _has_nulls && col.has_nulls();
    if (single_int_width(left) != _keyw || _keyw == 0 || (probe_nulls && _nulls_equal == null_equality::EQUAL) ||
        left.num_rows() < env_flag("CUDF_AMD_JOIN_DENSE_ORDERED_MIN_PROBE", 1 << 20) || left.num_rows() + row_base > (int64_t{1} << 31) - 1)
      return std::nullopt;
    hipStream_t const s = stream.value();
    std::size_t const n = static_cast<std::size_t>(left.num_rows());
    auto out_l          = std::make_unique<rmm::device_uvector<size_type>>(n, s, mr);
    auto out_r          = std::make_unique<rmm::device_uvector<size_type>>(n, s, mr);
    join::dense_stage_args a{};
    a.keys           = key_bytes(col, _keyw);
    a.key_width      = _keyw;
    a.key_signed     = _key_signed ? 1 : 0;
    a.mask           = probe_nulls ? col.null_mask() : nullptr;
    a.mask_offset    = col.offset();
    a.nrows          = left.num_rows();
    a.dense_lo       = _dense_lo;
    a.dense_range    = _dense_range;
    a.head           = static_cast<int32_t const*>(_dense_head.data());
    a.probe_row_base = row_base;
    rmm::device_buffer d_args{sizeof(join::dense_stage_args), s, cudf::get_current_device_resource_ref()};
    join::launch_dense_left_direct(a, static_cast<join::dense_stage_args*>(d_args.data()), out_l->data(), out_r->data(), s);
    // (no synchronisation: the scratch goes back to the pool in stream order, the caller reads the result on this stream)
    return join_index_pair{std::move(out_l), std::move(out_r)};
  }

  // ---- the tables' own passes: count (match cache, per-workgroup pair counts) -> scan -> retrieve, over the open-addressing table
  // in HBM or the dense direct-address table (rounds 1 and 2)
  [[nodiscard]] join_index_pair probe_tables(table_view const& left, join_kind kind, int k, std::optional<std::size_t> output_size, stream_ref stream,
                                             rmm::device_async_resource_ref mr, int64_t row_base) const
  {
    hipStream_t const s = stream.value();
    ensure_classic(stream);
    rmm::device_buffer d_args{sizeof(join_args), s, cudf::get_current_device_resource_ref()};
    // ---- size: given, or counted with one probe pass (reference: compute_join_output_size, size_impl.cuh:26-61)
    // The count pass always runs: its per-workgroup counts place every pair without global atomics (the pass
    // costs ~6% of the retrieve it replaces); a caller-supplied output_size is checked against it.
    std::size_t pairs = 0;
    join_args c = base_args(left, k == 0 ? 0 : 1);
    c.probe_row_base = row_base;
    std::size_t const cache_len = static_cast<std::size_t>(left.num_rows());
    rmm::device_buffer counts{(static_cast<std::size_t>(c.nblocks) + 1) * sizeof(unsigned long long), s,
                              cudf::get_current_device_resource_ref()};
    c.block_counts = static_cast<unsigned long long*>(counts.data());
    rmm::device_buffer cache{cache_len * sizeof(uint32_t), s, cudf::get_current_device_resource_ref()};
    c.match_cache = static_cast<uint32_t*>(cache.data());
    join::launch_count(c, static_cast<join_args*>(d_args.data()), s);
    join::launch_scan(c, s);
    {
      unsigned long long total = 0;
      CUDF_HIP_TRY(hipMemcpyAsync(&total, c.block_counts + c.nblocks, sizeof(total), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));
      pairs = static_cast<std::size_t>(total);
    }
    if (output_size.has_value() && kind != join_kind::FULL_JOIN)
      CUDF_EXPECTS(*output_size == pairs, "hash join: output_size does not match the number of matches", std::invalid_argument);
    std::size_t const complement_room = kind == join_kind::FULL_JOIN ? static_cast<std::size_t>(_right.num_rows()) : 0;
    std::size_t const room            = pairs + complement_room;
    CUDF_EXPECTS(room <= static_cast<std::size_t>(std::numeric_limits<size_type>::max()),
                 "Join result exceeds the maximum column size; use the *_join_size API and chunk the probe side.",
                 std::overflow_error);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(room, s, mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(room, s, mr);
    join_args a = c;
    a.kind      = k;
    dev_scalar<unsigned long long> cursor{static_cast<unsigned long long>(pairs), s};  // complement appends after the pairs
    a.total        = cursor.ptr();
    a.block_counts = c.block_counts;
    a.match_cache  = c.match_cache;
    a.out_probe    = out_l->data();
    a.out_build    = out_r->data();
    a.out_capacity = room;
    rmm::device_buffer matched{};
    if (kind == join_kind::FULL_JOIN) {
      matched = rmm::device_buffer{static_cast<std::size_t>(_right.num_rows()), s, cudf::get_current_device_resource_ref()};
      CUDF_HIP_TRY(hipMemsetAsync(matched.data(), 0, matched.size(), s));
      a.build_matched = static_cast<uint8_t*>(matched.data());
    }
    rmm::device_buffer big{};
    if (_dense && _dense_has_dups) {  // work list for probe rows that hit a hot build key (engine.hpp join_args::big_list)
      big = rmm::device_buffer{(5 * static_cast<std::size_t>(join::BIG_LIST_CAP) + 4) * sizeof(uint32_t), s, cudf::get_current_device_resource_ref()};
      a.big_list  = static_cast<uint32_t*>(big.data());
      a.big_count = a.big_list + 5 * static_cast<std::size_t>(join::BIG_LIST_CAP);
      CUDF_HIP_TRY(hipMemsetAsync(a.big_count, 0, sizeof(uint32_t), s));
    }
    join::launch_retrieve(a, static_cast<join_args*>(d_args.data()), s);
    if (a.big_list != nullptr) join::launch_dense_big_emit(a, static_cast<join_args*>(d_args.data()), s);
    if (kind == join_kind::FULL_JOIN) {
      rmm::device_buffer d_args2{sizeof(join_args), s, cudf::get_current_device_resource_ref()};
      join::launch_complement(a, static_cast<join_args*>(d_args2.data()), s);
      std::size_t const total = static_cast<std::size_t>(cursor.value());
      if (total != room) {  // exact-size outputs
        auto l2 = std::make_unique<rmm::device_uvector<size_type>>(total, s, mr);
        auto r2 = std::make_unique<rmm::device_uvector<size_type>>(total, s, mr);
        CUDF_HIP_TRY(hipMemcpyAsync(l2->data(), out_l->data(), total * sizeof(size_type), hipMemcpyDeviceToDevice, s));
        CUDF_HIP_TRY(hipMemcpyAsync(r2->data(), out_r->data(), total * sizeof(size_type), hipMemcpyDeviceToDevice, s));
        CUDF_HIP_TRY(hipStreamSynchronize(s));
        return {std::move(l2), std::move(r2)};
      }
      return {std::move(out_l), std::move(out_r)};
    }
    CUDF_HIP_TRY(hipStreamSynchronize(s));  // d_args / counts go out of scope
    return {std::move(out_l), std::move(out_r)};
  }

 private:
  void validate_probe(table_view const& left) const
  {
    // reference validate_hash_join_probe, hash_join.cu:47-59
    CUDF_EXPECTS(0 != left.num_columns(), "Hash join left table is empty", std::invalid_argument);
    CUDF_EXPECTS(_right.num_columns() == left.num_columns(), "Mismatch in number of columns to be joined on",
                 std::invalid_argument);
    CUDF_EXPECTS(_has_nulls || !cudf::has_nulls(left),
                 "Left table has nulls while right table was not hashed with null check.", std::invalid_argument);
    CUDF_EXPECTS(same_types(_right, left), "Mismatch in joining column data types", cudf::data_type_error);
  }

  join_args base_args(table_view const& probe, int kind) const
  {
    join_args a{};
    a.build       = _build_dev;
    a.probe       = make_device_table(probe);
    a.table       = const_cast<uint64_t*>(static_cast<uint64_t const*>(_table.data()));
    a.capacity    = _capacity;
    a.nulls_equal = _nulls_equal == null_equality::EQUAL;
    a.check_nulls = _has_nulls && (cudf::has_nulls(_right) || cudf::has_nulls(probe));
    a.kind        = kind;
    if (_dense) {
      a.dense_head     = const_cast<int32_t*>(static_cast<int32_t const*>(_dense_head.data()));
      a.dense_next     = const_cast<int32_t*>(static_cast<int32_t const*>(_dense_next.data()));
      a.dense_lo       = _dense_lo;
      a.dense_range    = _dense_range;
      a.dense_has_dups = _dense_has_dups ? 1 : 0;
    }
    // one 8-byte integer key per side; NULLs are fine as long as they can never match (UNEQUAL): then every key
    // comparison is between valid values and needs no descriptor walk
    a.single64    = is_single64(_right) && is_single64(probe) && (!a.check_nulls || !a.nulls_equal);
    // one workgroup per contiguous chunk of probe rows (>= 2048 rows each, at most 16 workgroups per CU)
    int64_t const n = probe.num_rows();
    a.nblocks       = static_cast<int32_t>(std::clamp<int64_t>((n + 2047) / 2048, 1, 256 * 16));
    a.chunk         = (n + a.nblocks - 1) / a.nblocks;
    a.chunk         = (a.chunk + 1023) / 1024 * 1024;
    return a;
  }

  // right side empty: every left row pairs with JoinNoMatch (reference get_trivial_left_join_indices)
  static join_index_pair trivial_left(table_view const& left, stream_ref stream, rmm::device_async_resource_ref mr,
                                      int64_t row_base = 0)
  {
    auto const n = static_cast<std::size_t>(left.num_rows());
    std::vector<size_type> l(n), r(n, JoinNoMatch);
    for (std::size_t i = 0; i < n; ++i) l[i] = static_cast<size_type>(static_cast<int64_t>(i) + row_base);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(n, stream.value(), mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(n, stream.value(), mr);
    if (n) {
      CUDF_HIP_TRY(hipMemcpyAsync(out_l->data(), l.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      CUDF_HIP_TRY(hipMemcpyAsync(out_r->data(), r.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
    }
    return {std::move(out_l), std::move(out_r)};
  }
  join_index_pair trivial_right(stream_ref stream, rmm::device_async_resource_ref mr) const
  {
    auto const n = static_cast<std::size_t>(_right.num_rows());
    std::vector<size_type> l(n, JoinNoMatch), r(n);
    for (std::size_t i = 0; i < n; ++i) r[i] = static_cast<size_type>(i);
    auto out_l = std::make_unique<rmm::device_uvector<size_type>>(n, stream.value(), mr);
    auto out_r = std::make_unique<rmm::device_uvector<size_type>>(n, stream.value(), mr);
    if (n) {
      CUDF_HIP_TRY(hipMemcpyAsync(out_l->data(), l.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      CUDF_HIP_TRY(hipMemcpyAsync(out_r->data(), r.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
    }
    return {std::move(out_l), std::move(out_r)};
  }

  table_view _right;
  bool _has_nulls;
  null_equality _nulls_equal;
  bool _is_empty;
  device_table _build_dev{};
  rmm::device_async_resource_ref _mr;
  bool _key64{false};
  int _keyw{0};             // 8 / 4: one integer key column of that width whose NULLs never match (the partitioned joins); 0: none
  bool _key_signed{false};
  int32_t _key_class{0};    // CLS_F32 / CLS_F64: the one key column is a float (radix join only: its normalised bits are the key); 0: integer
  bool _kw2{false};         // two key columns of 4 / 8 bytes each as a two-word radix key (is_two_word_key)
  bool _kw2_packed{false};  // ... whose build-side ranges fit one word (decide_packed_words): the single-word radix join serves them
  uint64_t _pk_lo[2]{0, 0}, _pk_range[2]{0, 0};
  int _pk_bits1{0};
  bool _pack2{false};       // two 4-byte integer key columns, packed into the radix join's 8-byte key (_keyw == 4)
  double _classic_load{0.5};
  // the open-addressing table in HBM: built by the constructor, or on first need when the build side took the radix partitions
  mutable std::mutex _classic_mu;
  mutable bool _classic_built{false};
  mutable uint64_t _capacity{0};
  mutable rmm::device_buffer _table{};
  // LDS radix join: the build side in LDS-sized partitions
  bool _radix{false};
  int32_t _rx_nparts{0};
  radix_side _rx_build{};
  bool _dense{false}, _dense_has_dups{false}, _dense_part{false};
  uint64_t _dense_lo{0}, _dense_range{0};
  rmm::device_buffer _dense_head{}, _dense_next{};
};
}  // namespace detail

// ---------------------------------------------------------------- cudf::hash_join
hash_join::~hash_join() = default;

hash_join::hash_join(table_view const& right, null_equality compare_nulls, stream_ref stream,
                     rmm::device_async_resource_ref mr)
  // If we cannot know beforehand about null existence then let's assume that there are nulls (reference hash_join.cu:190-199)
  : hash_join{right, nullable_join::YES, compare_nulls, 0.5, stream, mr}
{
}

hash_join::hash_join(table_view const& right, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
                     stream_ref stream, rmm::device_async_resource_ref mr)
  : _impl{[&] {
      CUDF_FUNC_RANGE();  // (reference hash_join.cu:62: the build is the constructor's work)
      return std::make_unique<detail::hash_join_impl const>(right, has_nulls == nullable_join::YES, compare_nulls, load_factor, stream, mr);
    }()}
{
}

join_index_pair hash_join::inner_join(table_view const& left, std::optional<std::size_t> output_size, stream_ref stream,
                                      rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->probe(left, join_kind::INNER_JOIN, output_size, stream, mr);
}
join_index_pair hash_join::left_join(table_view const& left, std::optional<std::size_t> output_size, stream_ref stream,
                                     rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->probe(left, join_kind::LEFT_JOIN, output_size, stream, mr);
}
join_index_pair hash_join::full_join(table_view const& left, std::optional<std::size_t> output_size, stream_ref stream,
                                     rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->probe(left, join_kind::FULL_JOIN, output_size, stream, mr);
}
std::size_t hash_join::inner_join_size(table_view const& left, stream_ref stream) const
{
  CUDF_FUNC_RANGE();
  return _impl->join_size(left, join_kind::INNER_JOIN, stream);
}
std::size_t hash_join::left_join_size(table_view const& left, stream_ref stream) const
{
  CUDF_FUNC_RANGE();
  return _impl->join_size(left, join_kind::LEFT_JOIN, stream);
}
std::size_t hash_join::full_join_size(table_view const& left, stream_ref stream, rmm::device_async_resource_ref) const
{
  CUDF_FUNC_RANGE();
  return _impl->join_size(left, join_kind::FULL_JOIN, stream);
}

join_match_context hash_join::inner_join_match_context(table_view const& left, stream_ref stream,
                                                       rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return join_match_context{left, _impl->match_counts(left, join_kind::INNER_JOIN, stream, mr)};
}
join_match_context hash_join::left_join_match_context(table_view const& left, stream_ref stream,
                                                      rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return join_match_context{left, _impl->match_counts(left, join_kind::LEFT_JOIN, stream, mr)};
}
join_match_context hash_join::full_join_match_context(table_view const& left, stream_ref stream,
                                                      rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return join_match_context{left, _impl->match_counts(left, join_kind::FULL_JOIN, stream, mr)};
}
join_index_pair hash_join::partitioned_inner_join(join_partition_context const& context, stream_ref stream,
                                                  rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->partitioned(context, join_kind::INNER_JOIN, stream, mr);
}
join_index_pair hash_join::partitioned_left_join(join_partition_context const& context, stream_ref stream,
                                                 rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->partitioned(context, join_kind::LEFT_JOIN, stream, mr);
}
join_index_pair hash_join::partitioned_full_join(join_partition_context const& context, stream_ref stream,
                                                 rmm::device_async_resource_ref mr) const
{
  CUDF_FUNC_RANGE();
  return _impl->partitioned(context, join_kind::FULL_JOIN, stream, mr);
}
// Concatenates the partial probe-side results and appends (JoinNoMatch, r) for every right row r that no partial
// matched (reference hash_join.hpp:414-441, join_utils.cu:45-221).
join_index_pair hash_join::finalize_partitioned_full_join(
  cudf::host_span<cudf::device_span<size_type const> const> left_partials,
  cudf::host_span<cudf::device_span<size_type const> const> right_partials, size_type left_table_num_rows,
  size_type right_table_num_rows, stream_ref stream, rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  CUDF_EXPECTS(left_partials.size() == right_partials.size(), "left and right partial results differ in number",
               std::invalid_argument);
  CUDF_EXPECTS(left_table_num_rows >= 0 && right_table_num_rows >= 0, "negative table size", std::invalid_argument);
  hipStream_t const s = stream.value();
  std::size_t probe_pairs = 0;
  for (std::size_t i = 0; i < left_partials.size(); ++i) {
    CUDF_EXPECTS(left_partials[i].size() == right_partials[i].size(), "partial index vectors differ in size",
                 std::invalid_argument);
    probe_pairs += left_partials[i].size();
  }
  std::size_t const room = probe_pairs + static_cast<std::size_t>(right_table_num_rows);
  CUDF_EXPECTS(room <= static_cast<std::size_t>(std::numeric_limits<size_type>::max()),
               "Join result exceeds the maximum column size", std::overflow_error);
  auto tmp = cudf::get_current_device_resource_ref();
  rmm::device_uvector<size_type> l(room, s, tmp), r(room, s, tmp);
  rmm::device_buffer matched{static_cast<std::size_t>(right_table_num_rows), s, tmp};
  CUDF_HIP_TRY(hipMemsetAsync(matched.data(), 0, matched.size(), s));
  std::size_t pos = 0;
  for (std::size_t i = 0; i < left_partials.size(); ++i) {
    std::size_t const m = left_partials[i].size();
    if (m == 0) continue;
    CUDF_HIP_TRY(hipMemcpyAsync(l.data() + pos, left_partials[i].data(), m * sizeof(size_type), hipMemcpyDeviceToDevice, s));
    CUDF_HIP_TRY(hipMemcpyAsync(r.data() + pos, right_partials[i].data(), m * sizeof(size_type), hipMemcpyDeviceToDevice, s));
    detail::join::launch_mark_matched(right_partials[i].data(), m, static_cast<uint8_t*>(matched.data()), s);
    pos += m;
  }
  std::size_t total = probe_pairs;
  if (right_table_num_rows > 0) {
    detail::join_args a{};
    a.build.nrows   = right_table_num_rows;
    a.build_matched = static_cast<uint8_t*>(matched.data());
    a.out_probe     = l.data();
    a.out_build     = r.data();
    a.out_capacity  = room;
    detail::dev_scalar<unsigned long long> cursor{static_cast<unsigned long long>(probe_pairs), s};
    a.total = cursor.ptr();
    rmm::device_buffer d_args{sizeof(detail::join_args), s, tmp};
    detail::join::launch_complement(a, static_cast<detail::join_args*>(d_args.data()), s);
    total = static_cast<std::size_t>(cursor.value());
  }
  auto out_l = std::make_unique<rmm::device_uvector<size_type>>(total, s, mr);
  auto out_r = std::make_unique<rmm::device_uvector<size_type>>(total, s, mr);
  if (total) {
    CUDF_HIP_TRY(hipMemcpyAsync(out_l->data(), l.data(), total * sizeof(size_type), hipMemcpyDeviceToDevice, s));
    CUDF_HIP_TRY(hipMemcpyAsync(out_r->data(), r.data(), total * sizeof(size_type), hipMemcpyDeviceToDevice, s));
  }
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  return {std::move(out_l), std::move(out_r)};
}

// ---------------------------------------------------------------- free functions
join_index_pair inner_join(table_view const& left, table_view const& right, null_equality compare_nulls, stream_ref stream,
                           rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  auto const has_nulls = (cudf::has_nulls(left) || cudf::has_nulls(right)) ? nullable_join::YES : nullable_join::NO;
  // build on the smaller table; ties build on right (reference join.cu:50-57)
  if (right.num_rows() > left.num_rows()) {
    hash_join hj{left, has_nulls, compare_nulls, 0.5, stream};
    auto [right_result, left_result] = hj.inner_join(right, std::nullopt, stream, mr);
    return {std::move(left_result), std::move(right_result)};
  }
  hash_join hj{right, has_nulls, compare_nulls, 0.5, stream};
  return hj.inner_join(left, std::nullopt, stream, mr);
}

join_index_pair left_join(table_view const& left, table_view const& right, null_equality compare_nulls, stream_ref stream,
                          rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  auto const has_nulls = (cudf::has_nulls(left) || cudf::has_nulls(right)) ? nullable_join::YES : nullable_join::NO;
  hash_join hj{right, has_nulls, compare_nulls, 0.5, stream};
  return hj.left_join(left, std::nullopt, stream, mr);
}

join_index_pair full_join(table_view const& left, table_view const& right, null_equality compare_nulls, stream_ref stream,
                          rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  auto const has_nulls = (cudf::has_nulls(left) || cudf::has_nulls(right)) ? nullable_join::YES : nullable_join::NO;
  hash_join hj{right, has_nulls, compare_nulls, 0.5, stream};
  return hj.full_join(left, std::nullopt, stream, mr);
}
}  // namespace cudf
