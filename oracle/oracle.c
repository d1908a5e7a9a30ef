/* TEST INFRASTRUCTURE ONLY — see oracle.h. Plain C restatement of the reference semantics; every function
 * cites the reference file:line it follows (paths relative to /root/reference/cpp). Sequential, single core. */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- cudf::type_id values (include/cudf/types.hpp:185-217) */
enum {
  T_EMPTY = 0, T_INT8, T_INT16, T_INT32, T_INT64, T_UINT8, T_UINT16, T_UINT32, T_UINT64, T_FLOAT32, T_FLOAT64,
  T_BOOL8, T_TS_DAYS, T_TS_S, T_TS_MS, T_TS_US, T_TS_NS, T_DUR_DAYS, T_DUR_S, T_DUR_MS, T_DUR_US, T_DUR_NS,
  T_DICT32, T_STRING, T_LIST, T_DEC32, T_DEC64, T_DEC128, T_STRUCT
};
/* ---- cudf::aggregation::Kind values (include/cudf/aggregation.hpp:78-121) */
enum { K_SUM = 0, K_SUM_OVERFLOW, K_PRODUCT, K_MIN, K_MAX, K_COUNT_VALID, K_COUNT_ALL, K_ANY, K_ALL,
       K_SUM_OF_SQUARES, K_MEAN, K_M2, K_VARIANCE, K_STD, K_MEDIAN, K_QUANTILE, K_ARGMAX, K_ARGMIN };

static char g_err[512];
const char* orc_last_error(void) { return g_err; }
static int fail(int code, const char* msg)
{
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
void orc_free(void* p) { free(p); }

/* ---- type classes */
enum { C_NONE = 0, C_SINT, C_UINT, C_F32, C_F64, C_BOOL };
static int type_width(int t)
{
  switch (t) {
    case T_INT8: case T_UINT8: case T_BOOL8: return 1;
    case T_INT16: case T_UINT16: return 2;
    case T_INT32: case T_UINT32: case T_FLOAT32: case T_TS_DAYS: case T_DUR_DAYS: case T_DEC32: return 4;
    case T_INT64: case T_UINT64: case T_FLOAT64: case T_TS_S: case T_TS_MS: case T_TS_US: case T_TS_NS:
    case T_DUR_S: case T_DUR_MS: case T_DUR_US: case T_DUR_NS: case T_DEC64: return 8;
    default: return 0;
  }
}
static int type_class(int t)
{
  switch (t) {
    case T_INT8: case T_INT16: case T_INT32: case T_INT64: case T_TS_DAYS: case T_TS_S: case T_TS_MS:
    case T_TS_US: case T_TS_NS: case T_DUR_DAYS: case T_DUR_S: case T_DUR_MS: case T_DUR_US: case T_DUR_NS:
    case T_DEC32: case T_DEC64: return C_SINT;
    case T_UINT8: case T_UINT16: case T_UINT32: case T_UINT64: return C_UINT;
    case T_FLOAT32: return C_F32;
    case T_FLOAT64: return C_F64;
    case T_BOOL8: return C_BOOL;
    default: return C_NONE;
  }
}
static int is_plain_numeric(int t) { return (t >= T_INT8 && t <= T_BOOL8); }
static int is_duration(int t) { return t >= T_DUR_DAYS && t <= T_DUR_NS; }
static int is_decimal(int t) { return t == T_DEC32 || t == T_DEC64; }

/* validity: bit (offset+i), LSB-first, 1 = valid; NULL mask = all valid (include/cudf/utilities/bit.hpp:47-104,
 * column_device_view_base.cuh:163,246). */
static int col_valid(const orc_column* c, int32_t i)
{
  if (!c->mask) return 1;
  int64_t b = (int64_t)c->offset + i;
  return (c->mask[b >> 5] >> (b & 31)) & 1u;
}
static const unsigned char* col_ptr(const orc_column* c, int32_t i)
{
  return (const unsigned char*)c->data + ((int64_t)c->offset + i) * type_width(c->type_id);
}
static int64_t get_sint(const orc_column* c, int32_t i)
{
  const unsigned char* p = col_ptr(c, i);
  switch (type_width(c->type_id)) {
    case 1: return *(const int8_t*)p;
    case 2: { int16_t v; memcpy(&v, p, 2); return v; }
    case 4: { int32_t v; memcpy(&v, p, 4); return v; }
    default: { int64_t v; memcpy(&v, p, 8); return v; }
  }
}
static uint64_t get_uint(const orc_column* c, int32_t i)
{
  const unsigned char* p = col_ptr(c, i);
  switch (type_width(c->type_id)) {
    case 1: return *(const uint8_t*)p;
    case 2: { uint16_t v; memcpy(&v, p, 2); return v; }
    case 4: { uint32_t v; memcpy(&v, p, 4); return v; }
    default: { uint64_t v; memcpy(&v, p, 8); return v; }
  }
}
static double get_f64(const orc_column* c, int32_t i)
{
  const unsigned char* p = col_ptr(c, i);
  if (c->type_id == T_FLOAT32) { float v; memcpy(&v, p, 4); return (double)v; }
  double v; memcpy(&v, p, 8); return v;
}

/* ---- MurmurHash3_x86_32 (public algorithm by Austin Appleby; the reference wraps cuco::murmurhash3_32,
 * include/cudf/hashing/detail/murmurhash3_x86_32.cuh:21-45). */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t orc_murmur3_32(const void* key, uint64_t len, uint32_t seed)
{
  const uint8_t* data = (const uint8_t*)key;
  const uint64_t nblocks = len / 4;
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  for (uint64_t i = 0; i < nblocks; ++i) {
    uint32_t k1; memcpy(&k1, data + 4 * i, 4);
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
    h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64u;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1: k1 ^= tail[0]; k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
  return h1;
}

/* Element hash: null -> UINT32_MAX (detail/row_operator/hashing.cuh:54-73); bool hashed as uint8 0/1
 * (murmurhash3_x86_32.cuh:47-52); floats normalised: x == 0 -> +0, NaN -> quiet NaN
 * (hashing/detail/hash_functions.cuh:19-37; murmurhash3_x86_32.cuh:54-66). */
static uint32_t element_hash(const orc_column* c, int32_t i, uint32_t seed)
{
  if (!col_valid(c, i)) return 0xffffffffu;
  switch (type_class(c->type_id)) {
    case C_BOOL: { uint8_t b = *(const uint8_t*)col_ptr(c, i) != 0; return orc_murmur3_32(&b, 1, seed); }
    case C_F32: {
      float v; memcpy(&v, col_ptr(c, i), 4);
      if (v == 0.0f) v = 0.0f;
      if (isnan(v)) { uint32_t q = 0x7fc00000u; memcpy(&v, &q, 4); }
      return orc_murmur3_32(&v, 4, seed);
    }
    case C_F64: {
      double v; memcpy(&v, col_ptr(c, i), 8);
      if (v == 0.0) v = 0.0;
      if (isnan(v)) { uint64_t q = 0x7ff8000000000000ull; memcpy(&v, &q, 8); }
      return orc_murmur3_32(&v, 8, seed);
    }
    default: return orc_murmur3_32(col_ptr(c, i), (uint64_t)type_width(c->type_id), seed);
  }
}
/* hash_combine (hashing/detail/hashing.hpp:83-86). */
static uint32_t hash_combine(uint32_t lhs, uint32_t rhs) { return lhs ^ (rhs + 0x9e3779b9u + (lhs << 6) + (lhs >> 2)); }
/* Row hash: first column's hash is the init, remaining folded (hashing.cuh:118-134). */
static uint32_t row_hash(const orc_column* cols, int32_t ncols, int32_t i, uint32_t seed)
{
  if (ncols == 0) return seed;
  uint32_t h = element_hash(&cols[0], i, seed);
  for (int32_t c = 1; c < ncols; ++c) h = hash_combine(h, element_hash(&cols[c], i, seed));
  return h;
}
int orc_row_hash(const orc_column* cols, int32_t ncols, uint32_t seed, uint32_t* out)
{
  int32_t n = ncols ? cols[0].size : 0;
  for (int32_t c = 0; c < ncols; ++c)
    if (!type_width(cols[c].type_id)) return fail(ORC_NOT_IMPLEMENTED, "row hash: fixed-width types only");
  for (int32_t i = 0; i < n; ++i) out[i] = row_hash(cols, ncols, i, seed);
  return ORC_OK;
}

/* Element equality: both null -> nulls_equal; one null -> false; floats: NaN == NaN, -0 == +0
 * (detail/row_operator/equality.cuh:59-89,244-262). */
static int element_equal(const orc_column* a, int32_t i, const orc_column* b, int32_t j, int nulls_equal)
{
  int va = col_valid(a, i), vb = col_valid(b, j);
  if (!va || !vb) return (!va && !vb) ? nulls_equal : 0;
  switch (type_class(a->type_id)) {
    case C_F32: case C_F64: {
      double x = get_f64(a, i), y = get_f64(b, j);
      return (isnan(x) && isnan(y)) || x == y;
    }
    case C_BOOL: return (get_uint(a, i) != 0) == (get_uint(b, j) != 0);
    default: return get_uint(a, i) == get_uint(b, j);
  }
}
static int rows_equal(const orc_column* a, int32_t na, int32_t i, const orc_column* b, int32_t j, int nulls_equal)
{
  for (int32_t c = 0; c < na; ++c)
    if (!element_equal(&a[c], i, &b[c], j, nulls_equal)) return 0;
  return 1;
}
static int row_has_null(const orc_column* cols, int32_t ncols, int32_t i)
{
  for (int32_t c = 0; c < ncols; ++c)
    if (!col_valid(&cols[c], i)) return 1;
  return 0;
}

/* ---- aggregation typing (detail/aggregation/aggregation.hpp:878-978). Returns target type_id or -1. */
static int target_type(int src, int kind)
{
  int cls = type_class(src);
  if (cls == C_NONE) return -1;
  switch (kind) {
    case K_MIN: case K_MAX: return src;
    case K_COUNT_VALID: case K_COUNT_ALL: return T_INT32;
    case K_MEAN:
      if (is_plain_numeric(src)) return T_FLOAT64;
      if (is_duration(src) || is_decimal(src)) return src;
      return -1;
    case K_SUM:
      if (cls == C_F32 || cls == C_F64) return src;
      if (is_plain_numeric(src)) return T_INT64; /* integral incl. bool */
      if (is_duration(src) || is_decimal(src)) return src;
      return -1;
    case K_SUM_OVERFLOW:
      /* struct {sum: source type, overflow: bool}; signed integers (not bool) and decimals
       * (detail/aggregation/aggregation.hpp:981-995). Reported here as the sum child's type. */
      if ((cls == C_SINT && is_plain_numeric(src)) || is_decimal(src)) return src;
      return -1;
    case K_PRODUCT: case K_SUM_OF_SQUARES:
      if (cls == C_F32 || cls == C_F64) return src;
      if (is_plain_numeric(src)) return T_INT64;
      return -1;
    case K_M2: case K_VARIANCE: case K_STD:
      return is_plain_numeric(src) ? T_FLOAT64 : -1;
    case K_ARGMAX: case K_ARGMIN: return T_INT32;
    default: return -1;
  }
}
/* Hash-path aggregations this oracle restates (groupby/common/utils.hpp:66-85 lists the hashable kinds). */
static int kind_supported(int kind)
{
  switch (kind) {
    case K_SUM: case K_SUM_OVERFLOW: case K_PRODUCT: case K_MIN: case K_MAX: case K_COUNT_VALID: case K_COUNT_ALL: case K_MEAN:
    case K_SUM_OF_SQUARES: case K_M2: case K_VARIANCE: case K_STD: case K_ARGMAX: case K_ARGMIN: return 1;
    default: return 0;
  }
}

/* ---- group map: open addressing over row indices, like the reference's static_set<size_type>
 * (groupby/hash/compute_groupby.cu:93-102), but sequential. */
typedef struct { int32_t* slot; uint64_t mask; } rowset;
static int rowset_init(rowset* s, int64_t n)
{
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2) cap <<= 1;
  s->slot = (int32_t*)malloc(cap * sizeof(int32_t));
  if (!s->slot) return 0;
  for (uint64_t i = 0; i < cap; ++i) s->slot[i] = -1;
  s->mask = cap - 1;
  return 1;
}

typedef struct {
  double f;    /* float accumulators */
  int64_t i;   /* signed accumulators / counts */
  uint64_t u;  /* unsigned accumulators */
  double mean_, m2_; /* Welford state for M2/VAR/STD */
  int64_t nvalid;
  int64_t nall;
  int32_t arg;
  int overflow; /* SUM_OVERFLOW */
} acc_t;

static void out_col_alloc(orc_out_column* o, int type_id, int32_t n, int nullable)
{
  o->type_id = type_id; o->size = n; o->null_count = 0;
  int w = type_width(type_id);
  o->data = calloc((size_t)(n > 0 ? n : 1), (size_t)(w ? w : 1));
  o->mask = nullable ? (uint32_t*)calloc((size_t)((n + 31) / 32 + 1), 4) : NULL;
}
static void out_set_valid(orc_out_column* o, int32_t i) { if (o->mask) o->mask[i >> 5] |= 1u << (i & 31); }
static void store_bits(orc_out_column* o, int32_t i, const void* src) { int w = type_width(o->type_id); memcpy((char*)o->data + (size_t)i * w, src, (size_t)w); }
static void store_int(orc_out_column* o, int32_t i, int64_t v)
{
  switch (type_width(o->type_id)) {
    case 1: { int8_t x = (int8_t)v; store_bits(o, i, &x); break; }
    case 2: { int16_t x = (int16_t)v; store_bits(o, i, &x); break; }
    case 4: { int32_t x = (int32_t)v; store_bits(o, i, &x); break; }
    default: store_bits(o, i, &v);
  }
}
static void store_float(orc_out_column* o, int32_t i, double v)
{
  if (o->type_id == T_FLOAT32) { float x = (float)v; store_bits(o, i, &x); } else store_bits(o, i, &v);
}

void orc_groupby_free(orc_groupby_result* r)
{
  if (!r) return;
  for (int i = 0; i < r->nkeys; ++i) { free(r->keys[i].data); free(r->keys[i].mask); }
  for (int i = 0; i < r->nresults; ++i) { free(r->results[i].data); free(r->results[i].mask); free(r->results[i].aux); }
  free(r->keys); free(r->results); free(r);
}

int orc_groupby(const orc_column* keys, int32_t nkeys, int32_t include_null_keys, const orc_request* reqs,
                int32_t nreqs, orc_groupby_result** out)
{
  *out = NULL;
  int32_t n = nkeys ? keys[0].size : 0;
  /* groupby.cu:225-229 */
  for (int r = 0; r < nreqs; ++r)
    if (reqs[r].values.size != n) return fail(ORC_LOGIC_ERROR, "Size mismatch between request values and groupby keys.");
  /* groupby.cu:186-201 */
  int ntot = 0;
  for (int r = 0; r < nreqs; ++r)
    for (int k = 0; k < reqs[r].nkinds; ++k, ++ntot)
      if (target_type(reqs[r].values.type_id, reqs[r].kinds[k]) < 0)
        return fail(ORC_LOGIC_ERROR, "Invalid type/aggregation combination.");
  for (int r = 0; r < nreqs; ++r)
    for (int k = 0; k < reqs[r].nkinds; ++k)
      if (!kind_supported(reqs[r].kinds[k])) return fail(ORC_NOT_IMPLEMENTED, "aggregation needs the sort path");
  for (int c = 0; c < nkeys; ++c)
    if (!type_width(keys[c].type_id)) return fail(ORC_NOT_IMPLEMENTED, "fixed-width keys only");

  /* pass 1: group ids in first-appearance order. Rule 1-2: EXCLUDE drops rows with any null key
   * (groupby/hash/groupby.cu:41-43; groupby/common/utils.cpp:14-32); nulls compare EQUAL (:56). */
  rowset set; if (!rowset_init(&set, n)) return fail(ORC_LOGIC_ERROR, "oom");
  int32_t* gid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* rep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* slot_gid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(set.mask + 1));
  int32_t G = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (!include_null_keys && row_has_null(keys, nkeys, i)) { gid[i] = -1; continue; }
    uint64_t s = row_hash(keys, nkeys, i, 0) & set.mask;
    for (;;) {
      int32_t r = set.slot[s];
      if (r < 0) { set.slot[s] = i; slot_gid[s] = G; rep[G] = i; gid[i] = G++; break; }
      if (rows_equal(keys, nkeys, i, keys, r, 1)) { gid[i] = slot_gid[s]; break; }
      s = (s + 1) & set.mask;
    }
  }
  free(set.slot); free(slot_gid);

  orc_groupby_result* res = (orc_groupby_result*)calloc(1, sizeof *res);
  res->nkeys = nkeys; res->keys = (orc_out_column*)calloc((size_t)(nkeys ? nkeys : 1), sizeof(orc_out_column));
  res->nresults = ntot; res->results = (orc_out_column*)calloc((size_t)(ntot ? ntot : 1), sizeof(orc_out_column));
  /* Rule 3: output keys = representative input rows, nullable iff the input key column is nullable
   * (compute_groupby.cu:104-111,154). */
  for (int c = 0; c < nkeys; ++c) {
    out_col_alloc(&res->keys[c], keys[c].type_id, G, keys[c].mask != NULL);
    for (int32_t g = 0; g < G; ++g) {
      store_bits(&res->keys[c], g, col_ptr(&keys[c], rep[g]));
      if (col_valid(&keys[c], rep[g])) out_set_valid(&res->keys[c], g); else res->keys[c].null_count++;
    }
  }
  /* pass 2: one accumulator array per (request, kind) */
  int oc = 0;
  for (int r = 0; r < nreqs; ++r) {
    const orc_column* v = &reqs[r].values;
    int cls = type_class(v->type_id);
    for (int k = 0; k < reqs[r].nkinds; ++k, ++oc) {
      int kind = reqs[r].kinds[k];
      int tgt = target_type(v->type_id, kind);
      acc_t* acc = (acc_t*)calloc((size_t)(G ? G : 1), sizeof(acc_t));
      /* identities: device_operators.cuh:60-76,132-139,190-197 */
      for (int32_t g = 0; g < G; ++g) {
        acc[g].arg = -1;
        if (kind == K_MIN || kind == K_ARGMIN) { acc[g].f = INFINITY; acc[g].i = INT64_MAX; acc[g].u = UINT64_MAX; }
        if (kind == K_MAX || kind == K_ARGMAX) { acc[g].f = -INFINITY; acc[g].i = INT64_MIN; acc[g].u = 0; }
        if (kind == K_PRODUCT) { acc[g].f = 1.0; acc[g].i = 1; acc[g].u = 1; }
      }
      for (int32_t i = 0; i < n; ++i) {
        int32_t g = gid[i];
        if (g < 0) continue;
        acc[g].nall++;
        /* device_aggregators.cuh:428-446: null source skipped for everything except COUNT_ALL */
        if (!col_valid(v, i)) continue;
        acc[g].nvalid++;
        double x = 0; int64_t xi = 0; uint64_t xu = 0;
        if (cls == C_F32 || cls == C_F64) x = get_f64(v, i);
        else if (cls == C_UINT) { xu = get_uint(v, i); xi = (int64_t)xu; x = (double)xu; }
        else if (cls == C_BOOL) { xi = get_uint(v, i) != 0; xu = (uint64_t)xi; x = (double)xi; }
        else { xi = get_sint(v, i); xu = (uint64_t)xi; x = (double)xi; }
        switch (kind) {
          case K_SUM: case K_MEAN:
            /* integral -> int64 wrapping add (unsigned arithmetic = two's complement wrap); float -> fp add */
            acc[g].f += x; acc[g].i = (int64_t)((uint64_t)acc[g].i + (uint64_t)xi); break;
          case K_SUM_OVERFLOW: {
            /* device_aggregators.cuh:136-160, in row order: once a group's flag is up its sum is left alone; the add is
             * done in the source's storage type and flags the group when it overflows that type. (The reference's atomics
             * arrive in any order: flag and sum are reproducible only for inputs where every order agrees - all the
             * reference's own tests, and the ones here, are of that kind.) */
            if (acc[g].overflow) break;
            int w = type_width(v->type_id);
            int64_t lo = w == 1 ? INT8_MIN : w == 2 ? INT16_MIN : w == 4 ? INT32_MIN : INT64_MIN;
            int64_t hi = w == 1 ? INT8_MAX : w == 2 ? INT16_MAX : w == 4 ? INT32_MAX : INT64_MAX;
            __int128 t = (__int128)acc[g].i + (__int128)xi;
            if (t < lo || t > hi) acc[g].overflow = 1;
            /* wrapped to the storage type, as atomic_add leaves it */
            uint64_t wrapped = (uint64_t)acc[g].i + (uint64_t)xi;
            acc[g].i = w == 1 ? (int64_t)(int8_t)wrapped : w == 2 ? (int64_t)(int16_t)wrapped : w == 4 ? (int64_t)(int32_t)wrapped : (int64_t)wrapped;
            break;
          }
          case K_PRODUCT: acc[g].f *= x; acc[g].i = (int64_t)((uint64_t)acc[g].i * (uint64_t)xi); break;
          case K_SUM_OF_SQUARES: acc[g].f += x * x; acc[g].i = (int64_t)((uint64_t)acc[g].i + (uint64_t)xi * (uint64_t)xi); break;
          case K_MIN:
            /* CAS loop around min(old, update) = (update < old) ? update : old — a NaN update never wins
             * (device_atomics.cuh:89-101; device_operators.cuh:27-45) */
            if (x < acc[g].f) acc[g].f = x;
            if (xi < acc[g].i) acc[g].i = xi;
            if (xu < acc[g].u) acc[g].u = xu;
            break;
          case K_MAX:
            if (x > acc[g].f) acc[g].f = x;
            if (xi > acc[g].i) acc[g].i = xi;
            if (xu > acc[g].u) acc[g].u = xu;
            break;
          case K_ARGMIN: {
            int better = (cls == C_F32 || cls == C_F64) ? (x < acc[g].f) : (cls == C_UINT ? (xu < acc[g].u) : (xi < acc[g].i));
            if (acc[g].arg < 0 || better) { acc[g].arg = i; acc[g].f = x; acc[g].i = xi; acc[g].u = xu; }
            break;
          }
          case K_ARGMAX: {
            int better = (cls == C_F32 || cls == C_F64) ? (x > acc[g].f) : (cls == C_UINT ? (xu > acc[g].u) : (xi > acc[g].i));
            if (acc[g].arg < 0 || better) { acc[g].arg = i; acc[g].f = x; acc[g].i = xi; acc[g].u = xu; }
            break;
          }
          case K_M2: case K_VARIANCE: case K_STD:
            /* hash path: M2/VARIANCE/STD are built from SUM_OF_SQUARES, SUM and COUNT_VALID in their target
             * types (extract_single_pass_aggs.cpp:26-177): int64 wrapping for integral sources, fp otherwise */
            acc[g].f += x; acc[g].i = (int64_t)((uint64_t)acc[g].i + (uint64_t)xi);
            acc[g].m2_ += x * x; acc[g].u = acc[g].u + (uint64_t)xi * (uint64_t)xi;
            break;
          default: break;
        }
      }
      /* Rule 5: result nullable iff kind is not COUNT and values.has_nulls() (output_utils.cu:67-86). */
      int is_count = (kind == K_COUNT_VALID || kind == K_COUNT_ALL);
      int nullable = !is_count && v->null_count > 0;
      if (kind == K_M2) nullable = 0;                      /* make_numeric_column without a mask */
      if (kind == K_VARIANCE || kind == K_STD) nullable = 1; /* mask rebuilt from the group counts */
      orc_out_column* o = &res->results[oc];
      out_col_alloc(o, tgt, G, nullable);
      if (kind == K_SUM_OVERFLOW) o->aux = (uint8_t*)calloc((size_t)(G ? G : 1), 1);
      int tcls = type_class(tgt);
      for (int32_t g = 0; g < G; ++g) {
        int valid = 1;
        switch (kind) {
          case K_COUNT_VALID: store_int(o, g, acc[g].nvalid); break;
          case K_COUNT_ALL: store_int(o, g, acc[g].nall); break;
          case K_SUM: case K_PRODUCT: case K_SUM_OF_SQUARES:
            valid = acc[g].nvalid > 0;
            if (tcls == C_F32 || tcls == C_F64) store_float(o, g, acc[g].f); else store_int(o, g, acc[g].i);
            break;
          case K_SUM_OVERFLOW:
            /* children carry no masks; the struct is null when no valid value reached the group (output_utils.cu:83-111;
             * KAT sum_overflow_tests.cpp:190-258: the children of a null struct read 0 / false) */
            valid = acc[g].nvalid > 0;
            store_int(o, g, valid ? acc[g].i : 0);
            o->aux[g] = (uint8_t)(valid && acc[g].overflow);
            break;
          case K_MIN: case K_MAX:
            valid = acc[g].nvalid > 0;
            if (tcls == C_F32 || tcls == C_F64) store_float(o, g, acc[g].f);
            else if (tcls == C_UINT) store_int(o, g, (int64_t)acc[g].u);
            else store_int(o, g, acc[g].i);
            break;
          case K_MEAN: {
            /* MEAN = SUM (in the SUM target type) / COUNT_VALID as FLOAT64; null iff count == 0
             * (extract_single_pass_aggs.cpp:63-74; hash_compound_agg_finalizer.cu:92-133) */
            valid = acc[g].nvalid > 0;
            double s = (cls == C_F32 || cls == C_F64) ? acc[g].f : (double)acc[g].i;
            if (tgt != T_FLOAT64) {
              /* duration / decimal: SUM in the source type (wrapped to its width) DIV count with the source type as output,
               * i.e. C++ integer division of the representation (KATs mean_tests.cpp:152-198: {9,19,17}/{3,4,3} -> {3,4,5}) */
              int w = type_width(tgt);
              int64_t sw = w == 4 ? (int64_t)(int32_t)acc[g].i : acc[g].i;
              store_int(o, g, valid ? sw / acc[g].nvalid : 0);
              break;
            }
            store_float(o, g, valid ? s / (double)acc[g].nvalid : 0.0);
            break;
          }
          case K_M2: case K_VARIANCE: case K_STD: {
            /* m2 = sum_sqr - sum*sum/count, 0 for an empty group (groupby/common/m2_var_std.cu:48-60);
             * variance = m2 / (count - ddof), std = sqrt(variance), null when count == 0 or count - ddof <= 0
             * (:152-187); ddof = 1 (aggregation.hpp default). M2 itself is never null. */
            int is_f = (cls == C_F32 || cls == C_F64);
            double ssq = is_f ? acc[g].m2_ : (double)(int64_t)acc[g].u;
            double sm  = is_f ? acc[g].f : (double)acc[g].i;
            double m2  = acc[g].nvalid > 0 ? ssq - sm * sm / (double)acc[g].nvalid : 0.0;
            if (kind == K_M2) { valid = 1; store_float(o, g, m2); break; }
            valid = acc[g].nvalid > 1;
            double var = valid ? m2 / (double)(acc[g].nvalid - 1) : 0.0;
            store_float(o, g, kind == K_STD ? sqrt(var) : var);
            break;
          }
          case K_ARGMIN: case K_ARGMAX: valid = acc[g].arg >= 0; store_int(o, g, acc[g].arg); break;
          default: break;
        }
        if (valid) out_set_valid(o, g); else if (o->mask) o->null_count++;
      }
      free(acc);
    }
  }
  free(gid); free(rep);
  *out = res;
  return ORC_OK;
}

/* ---- joins (src/join/join.cu:30-118; hash_join/hash_join.cu:32-59; retrieve_impl.cuh:169-222) */
static int join_validate(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright)
{
  if (nright == 0) return fail(ORC_INVALID_ARGUMENT, "Hash join right table is empty");
  if (nleft == 0) return fail(ORC_INVALID_ARGUMENT, "Hash join left table is empty");
  if (nleft != nright) return fail(ORC_INVALID_ARGUMENT, "Mismatch in number of columns to be joined on");
  for (int c = 0; c < nleft; ++c)
    if (left[c].type_id != right[c].type_id) return fail(ORC_DATA_TYPE_ERROR, "Mismatch in joining column data types");
  for (int c = 0; c < nleft; ++c)
    if (!type_width(left[c].type_id)) return fail(ORC_NOT_IMPLEMENTED, "fixed-width keys only");
  return ORC_OK;
}

typedef struct { int32_t* l; int32_t* r; int64_t n, cap; } pairbuf;
static void emit(pairbuf* b, int32_t l, int32_t r)
{
  if (b->n == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 1024;
    b->l = (int32_t*)realloc(b->l, sizeof(int32_t) * (size_t)b->cap);
    b->r = (int32_t*)realloc(b->r, sizeof(int32_t) * (size_t)b->cap);
  }
  b->l[b->n] = l; b->r[b->n] = r;
  b->n++;
}

static int join_core(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright, int32_t nulls_equal,
                     int32_t kind, pairbuf* out)
{
  int rc = join_validate(left, nleft, right, nright);
  if (rc) return rc;
  int32_t nl = left[0].size, nr = right[0].size;
  const int32_t NOMATCH = INT32_MIN; /* JoinNoMatch, join/join.hpp:72 */
  /* chained hash of right rows, heads by bucket; chains kept in ascending right-row order */
  uint64_t cap = 16; while (cap < (uint64_t)nr * 2) cap <<= 1;
  int32_t* head = (int32_t*)malloc(sizeof(int32_t) * cap);
  int32_t* next = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr ? nr : 1));
  unsigned char* rmatched = (unsigned char*)calloc((size_t)(nr ? nr : 1), 1);
  for (uint64_t i = 0; i < cap; ++i) head[i] = -1;
  for (int32_t j = nr - 1; j >= 0; --j) {
    /* UNEQUAL: build side skips rows containing a null (hash_join.cu:77-84; join_common_utils.cuh:36-47) */
    if (!nulls_equal && row_has_null(right, nright, j)) continue;
    uint64_t b = row_hash(right, nright, j, 0) & (cap - 1);
    next[j] = head[b]; head[b] = j;
  }
  for (int32_t i = 0; i < nl; ++i) {
    int found = 0;
    if (nulls_equal || !row_has_null(left, nleft, i)) {
      uint64_t b = row_hash(left, nleft, i, 0) & (cap - 1);
      for (int32_t j = head[b]; j >= 0; j = next[j])
        if (rows_equal(left, nleft, i, right, j, nulls_equal)) { emit(out, i, j); rmatched[j] = 1; found = 1; }
    }
    if (!found && kind != 0) emit(out, i, NOMATCH); /* left/full: unmatched left row (retrieve_impl.cuh:105-121) */
  }
  if (kind == 2) /* full: right complement (join_utils.cu:45-221) */
    for (int32_t j = 0; j < nr; ++j)
      if (!rmatched[j]) emit(out, NOMATCH, j);
  free(head); free(next); free(rmatched);
  return ORC_OK;
}

int orc_join(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright, int32_t nulls_equal,
             int32_t kind, int32_t** out_left, int32_t** out_right, int64_t* out_n)
{
  pairbuf b = {0};
  int rc = join_core(left, nleft, right, nright, nulls_equal, kind, &b);
  if (rc) { free(b.l); free(b.r); return rc; }
  if (!b.l) { b.l = (int32_t*)malloc(4); b.r = (int32_t*)malloc(4); }
  *out_left = b.l; *out_right = b.r; *out_n = b.n;
  return ORC_OK;
}
/* Size only: groups equal right rows first so that the all-duplicates case (65567^2 pairs,
 * join_tests.cpp:2379-2394) costs O(n) instead of O(n^2). Same match rule as join_core. */
int orc_join_size(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright,
                  int32_t nulls_equal, int32_t kind, uint64_t* out_n)
{
  int rc = join_validate(left, nleft, right, nright);
  if (rc) return rc;
  int32_t nl = left[0].size, nr = right[0].size;
  uint64_t cap = 16; while (cap < (uint64_t)nr * 2) cap <<= 1;
  int32_t* head = (int32_t*)malloc(sizeof(int32_t) * cap);
  int32_t* next = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nr ? nr : 1));
  uint64_t* cnt = (uint64_t*)calloc((size_t)(nr ? nr : 1), sizeof(uint64_t));
  unsigned char* matched = (unsigned char*)calloc((size_t)(nr ? nr : 1), 1);
  for (uint64_t i = 0; i < cap; ++i) head[i] = -1;
  uint64_t skipped_right = 0; /* right rows that can never match (UNEQUAL + null) */
  for (int32_t j = 0; j < nr; ++j) {
    if (!nulls_equal && row_has_null(right, nright, j)) { skipped_right++; continue; }
    uint64_t b = row_hash(right, nright, j, 0) & (cap - 1);
    int32_t r = head[b];
    for (; r >= 0; r = next[r])
      if (rows_equal(right, nright, j, right, r, nulls_equal)) break;
    if (r >= 0) cnt[r]++; else { next[j] = head[b]; head[b] = j; cnt[j] = 1; }
  }
  uint64_t n = 0;
  for (int32_t i = 0; i < nl; ++i) {
    uint64_t m = 0;
    if (nulls_equal || !row_has_null(left, nleft, i)) {
      uint64_t b = row_hash(left, nleft, i, 0) & (cap - 1);
      for (int32_t r = head[b]; r >= 0; r = next[r])
        if (rows_equal(left, nleft, i, right, r, nulls_equal)) { m = cnt[r]; matched[r] = 1; break; }
    }
    n += m ? m : (kind != 0 ? 1 : 0);
  }
  if (kind == 2) {
    n += skipped_right;
    for (int32_t j = 0; j < nr; ++j) if (cnt[j] && !matched[j]) n += cnt[j];
  }
  free(head); free(next); free(cnt); free(matched);
  *out_n = n;
  return ORC_OK;
}

/* ---- hash partition (src/partitioning/partitioning.cu:54-92: hash % P, or hash & (P-1) for powers of two —
 * the same value; :569-760 stable within a partition is NOT promised by the reference; the oracle emits the
 * stable order and tests compare partition membership as sets). */
int orc_hash_partition(const orc_column* cols, int32_t ncols, int32_t num_partitions, uint32_t seed,
                       int32_t* out_part, int32_t* out_offsets, int32_t* out_order)
{
  if (num_partitions <= 0) return fail(ORC_LOGIC_ERROR, "num_partitions must be positive");
  int32_t n = ncols ? cols[0].size : 0;
  for (int c = 0; c < ncols; ++c)
    if (!type_width(cols[c].type_id)) return fail(ORC_NOT_IMPLEMENTED, "fixed-width types only");
  int32_t* cnt = (int32_t*)calloc((size_t)num_partitions + 1, sizeof(int32_t));
  for (int32_t i = 0; i < n; ++i) { out_part[i] = (int32_t)(row_hash(cols, ncols, i, seed) % (uint32_t)num_partitions); cnt[out_part[i] + 1]++; }
  for (int p = 0; p < num_partitions; ++p) { out_offsets[p] = cnt[p]; cnt[p + 1] += cnt[p]; }
  for (int32_t i = 0; i < n; ++i) out_order[cnt[out_part[i]]++] = i;
  free(cnt);
  return ORC_OK;
}
