// Prototype of the LDS-ring ("software write-combining buffer") radix scatter of 16-byte records:
// every partition owns a ring of CAP record slots in LDS; a row reserves the next virtual position of its partition with
// ONE returning LDS atomic and writes its record there; after a barrier the owners flush every COMPLETE granule of G
// records to the partition's region (global record index = region base + virtual position, so nothing is ever moved
// inside LDS); a second barrier ends the tile. Two barriers and ~4 LDS operations per row, against seven barriers and ~8
// for the rank / scan / stage / write-out / carry-move scatter of common/wc_scatter.hpp.
//   ./ring_scatter_micro [rows_millions] [P] [dense 0/1] [ring_mib 0 = HBM-sized destination]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T gload(T const* p) { return *(GAS T const*)(p); }
template <typename T> __device__ __forceinline__ void gstore(T* p, T v) { *(GAS T*)(p) = v; }
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct sargs {
  u64 const* keys; u64 const* vals; int64_t n;
  int P, shift, dense; u64 lo; uint32_t mult, bmask; int dshift;
  u64x2* out; int64_t region_cap; int32_t* region_count; int32_t* overflow; int slices;
};

template <bool DENSE>
__device__ __forceinline__ uint32_t digit_of(sargs const& a, u64 key)
{
  if constexpr (DENSE) return ((static_cast<uint32_t>(key - a.lo) * a.mult) & a.bmask) >> a.dshift;
  return static_cast<uint32_t>(mix64(0x9e3779b97f4a7c15ull ^ key) >> a.shift) & static_cast<uint32_t>(a.P - 1);
}

// G: records per granule; CAPL: log2 of the ring capacity per partition; D: tiles prefetched ahead; RPT rows per thread per tile
template <int G, int CAPL, int D, bool DENSE, int RPT = 1, int MODE = 0>
__global__ void __launch_bounds__(1024) k_scatter_ring(sargs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr uint32_t CAP = 1u << CAPL;
  int const P = a.P, B = blockDim.x;
  u64x2* ring     = reinterpret_cast<u64x2*>(lds_raw);
  uint32_t* tail  = reinterpret_cast<uint32_t*>(ring + static_cast<size_t>(P) * CAP);
  uint32_t* limit = tail + P;
  __shared__ int s_pending, s_abort;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  int const PW = P / nwaves;  // partitions owned by a wave (<= 64): owner lane l < PW holds partition wave * PW + l
  int const item = blockIdx.x;
  uint32_t head = 0;
  for (int d = threadIdx.x; d < P; d += B) { tail[d] = 0; limit[d] = CAP; }
  if (threadIdx.x == 0) { s_pending = 0; s_abort = 0; }
  __syncthreads();
  int64_t const T = static_cast<int64_t>(B) * RPT, step = static_cast<int64_t>(a.slices) * T;
  u64 pk[D][RPT], pv[D][RPT];
  int64_t tile = static_cast<int64_t>(item) * T;
  constexpr bool L16 = (MODE & 1) != 0;
  static_assert(!L16 || RPT % 2 == 0);
  // row of slot k of this thread in a tile: 8-byte loads: k * B + t; 16-byte loads: rows (2m, 2m+1) -> (m * B + t) * 2 + {0, 1}
  auto row_in_tile = [&](int k) -> int64_t { return L16 ? (static_cast<int64_t>(k / 2) * B + threadIdx.x) * 2 + (k & 1) : static_cast<int64_t>(k) * B + threadIdx.x; };
  auto load_rows = [&](int64_t base, u64 (&kk)[RPT], u64 (&vv)[RPT]) {
    if constexpr (L16) {
#pragma unroll
      for (int m = 0; m < RPT / 2; ++m) {
        int64_t const r = base + row_in_tile(2 * m);
        if (r + 1 < a.n) {
          u64x2 const k2 = gload(reinterpret_cast<u64x2 const*>(a.keys + r)), v2 = gload(reinterpret_cast<u64x2 const*>(a.vals + r));
          kk[2 * m] = k2.x; kk[2 * m + 1] = k2.y; vv[2 * m] = v2.x; vv[2 * m + 1] = v2.y;
        } else if (r < a.n) { kk[2 * m] = gload(a.keys + r); vv[2 * m] = gload(a.vals + r); }
      }
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        int64_t const r = base + row_in_tile(k);
        if (r < a.n) { kk[k] = gload(a.keys + r); vv[k] = gload(a.vals + r); }
      }
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) load_rows(tile + j * step, pk[j], pv[j]);
  uint32_t const region_cap = static_cast<uint32_t>(a.region_cap);
  // flush every complete granule of this wave's partitions; FINAL: also the partial last granule
  auto flush = [&](bool final) {
    uint32_t ngr = 0, nrec = 0;
    int ab = 0;
    int const dmine = wave * PW + lane;
    if (lane < PW) {
      uint32_t const t = tail[dmine], lim = head + CAP;
      uint32_t const c = static_cast<int32_t>(t - lim) < 0 ? t : lim;
      uint32_t const complete = final ? c : (c & ~static_cast<uint32_t>(G - 1));
      nrec = complete - head;
      ngr  = (nrec + G - 1) / G;
      if (complete > region_cap) { s_abort = 1; ab = 1; }
    }
    constexpr int PB = 64 / G;  // partitions per batch
    for (int b = 0; b * PB < PW; ++b) {
      int const pl        = b * PB + lane / G;
      int const sub       = lane % G;
      uint32_t const mg   = __shfl(ngr, pl), mh = __shfl(head, pl), mr = __shfl(nrec, pl);
      int const mab       = __shfl(ab, pl);
      bool const owner_ok = pl < PW;
      int const d         = wave * PW + pl;
      int64_t const rbase = (static_cast<int64_t>(d) * a.slices + item) * a.region_cap;
      for (uint32_t g = 0;; ++g) {
        bool const act = owner_ok && g < mg;
        if (__ballot(act) == 0) break;
        uint32_t const q = g * G + sub;
        if (act && q < mr && !mab) {
          uint32_t const pos = mh + q;
          u64x2 const rv = ring[static_cast<uint32_t>(d) * CAP + (pos & (CAP - 1))];
          if ((MODE & 2) == 0 || rv.x == 0x123456789ull) gstore(a.out + rbase + pos, rv);
        }
      }
    }
    if (lane < PW) {
      head += final ? nrec : (nrec & ~static_cast<uint32_t>(G - 1));
      limit[dmine] = head + CAP;
    }
  };
  for (; tile < a.n; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      u64 key[RPT], val[RPT];
      bool keep[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        int64_t const r = tile + j * step + row_in_tile(k);
        keep[k] = r < a.n;
        key[k] = pk[j][k]; val[k] = pv[j][k];
      }
      load_rows(tile + (j + D) * step, pk[j], pv[j]);
      if (tile + j * step >= a.n) break;  // (uniform)
      uint32_t d[RPT], pos[RPT], lim[RPT];
      bool pend[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        d[k] = 0; pos[k] = 0; lim[k] = 0;
        if (keep[k]) {
          d[k]   = digit_of<DENSE>(a, key[k]);
          pos[k] = atomicAdd(&tail[d[k]], 1u);
          lim[k] = limit[d[k]];
        }
      }
      bool any_pend = false;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pend[k] = keep[k] && static_cast<int32_t>(pos[k] - lim[k]) >= 0;
        if (keep[k] && !pend[k]) ring[d[k] * CAP + (pos[k] & (CAP - 1))] = u64x2{key[k], val[k]};
        any_pend = any_pend || pend[k];
      }
      if (any_pend) s_pending = 1;
      lds_barrier();
      flush(false);
      lds_barrier();
      while (s_pending) {  // a ring was full: its granules are flushed by now, the waiting rows go in
        lds_barrier();
        if (threadIdx.x == 0) s_pending = 0;
        lds_barrier();
        any_pend = false;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k]) {
            uint32_t const l2 = limit[d[k]];
            if (static_cast<int32_t>(pos[k] - l2) < 0) { ring[d[k] * CAP + (pos[k] & (CAP - 1))] = u64x2{key[k], val[k]}; pend[k] = false; }
            else any_pend = true;
          }
        }
        if (any_pend) s_pending = 1;
        lds_barrier();
        flush(false);
        lds_barrier();
        if (s_abort) break;
      }
      if (s_abort) {
        if (threadIdx.x == 0) *a.overflow = 1;
        return;
      }
    }
  }
  flush(true);
  lds_barrier();
  if (s_abort) { if (threadIdx.x == 0) *a.overflow = 1; return; }
  if (lane < PW) a.region_count[static_cast<int64_t>(wave * PW + lane) * a.slices + item] = static_cast<int32_t>(head);
}


// ---- verification: per-partition record count and checksum, from the input and from the regions
template <bool DENSE>
__global__ void k_check_in(sargs a, u64* cnt, u64* sum)
{
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < a.n; r += stride) {
    u64 const k = a.keys[r], v = a.vals[r];
    uint32_t const d = digit_of<DENSE>(a, k);
    atomicAdd(&cnt[d], 1ull);
    atomicAdd(&sum[d], mix64(k) + 3 * mix64(v ^ k));
  }
}
template <bool DENSE>
__global__ void k_check_out(sargs a, u64* cnt, u64* sum, u64* bad)
{
  int const region = blockIdx.x;  // d * slices + w
  int const d      = region / a.slices;
  int const c      = a.region_count[region];
  u64 lc = 0, ls = 0;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    u64x2 const r = a.out[static_cast<int64_t>(region) * a.region_cap + i];
    if (digit_of<DENSE>(a, r.x) != static_cast<uint32_t>(d)) atomicAdd(bad, 1ull);
    lc += 1;
    ls += mix64(r.x) + 3 * mix64(r.y ^ r.x);
  }
  atomicAdd(&cnt[d], lc);
  atomicAdd(&sum[d], ls);
}
__global__ void k_fill(u64* keys, u64* vals, int64_t n, u64 groups, int dense)
{
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < n; r += stride) {
    u64 const g = mix64(r * 0x9e3779b97f4a7c15ull + 1) % groups;
    keys[r] = dense ? g : mix64(g + 77);
    vals[r] = mix64(r + 12345);
  }
}


// ---- SoA variant for dense keys: a record is (value f64, slot u32) = 12 bytes, written as two streams per region:
// granules of 16 values (128 B) and of 16 slots (64 B). One row per (k, thread); G = 16.
template <int CAPL, int D, int RPT>
__global__ void __launch_bounds__(1024) k_scatter_ring_soa(sargs a, u64* out_val, uint32_t* out_slot)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr uint32_t CAP = 1u << CAPL, G = 16;
  int const P = a.P, B = blockDim.x;
  u64* rval       = reinterpret_cast<u64*>(lds_raw);                                      // [P][CAP]
  uint32_t* rslot = reinterpret_cast<uint32_t*>(rval + static_cast<size_t>(P) * CAP);    // [P][CAP]
  uint32_t* tail  = rslot + static_cast<size_t>(P) * CAP;
  uint32_t* limit = tail + P;
  __shared__ int s_pending, s_abort;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  int const PW = P / nwaves, item = blockIdx.x;
  uint32_t head = 0;
  for (int d = threadIdx.x; d < P; d += B) { tail[d] = 0; limit[d] = CAP; }
  if (threadIdx.x == 0) { s_pending = 0; s_abort = 0; }
  __syncthreads();
  int64_t const T = static_cast<int64_t>(B) * RPT, step = static_cast<int64_t>(a.slices) * T;
  u64 pk[D][RPT], pv[D][RPT];
  int64_t tile = static_cast<int64_t>(item) * T;
  auto load_rows = [&](int64_t base, u64 (&kk)[RPT], u64 (&vv)[RPT]) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      int64_t const r = base + static_cast<int64_t>(k) * B + threadIdx.x;
      if (r < a.n) { kk[k] = gload(a.keys + r); vv[k] = gload(a.vals + r); }
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) load_rows(tile + j * step, pk[j], pv[j]);
  uint32_t const region_cap = static_cast<uint32_t>(a.region_cap);
  uint32_t const smask = (1u << a.dshift) - 1u;
  auto flush = [&](bool final) {
    uint32_t nrec = 0;
    int ab = 0;
    int const dmine = wave * PW + lane;
    if (lane < PW) {
      uint32_t const t = tail[dmine], lim = head + CAP;
      uint32_t const c = static_cast<int32_t>(t - lim) < 0 ? t : lim;
      uint32_t const complete = final ? c : (c & ~(G - 1));
      nrec = complete - head;
      if (complete > region_cap) { s_abort = 1; ab = 1; }
    }
    // values: 8 lanes per granule of 16 (two values per lane); slots: 4 lanes per granule of 16 (four slots per lane)
    for (int b = 0; b * 8 < PW; ++b) {
      int const pl = b * 8 + lane / 8, sub = lane % 8;
      uint32_t const mr = __shfl(nrec, pl), mh = __shfl(head, pl);
      int const mab = __shfl(ab, pl);
      int const d = wave * PW + pl;
      int64_t const rbase = (static_cast<int64_t>(d) * a.slices + item) * a.region_cap;
      for (uint32_t g = 0;; ++g) {
        uint32_t const q = g * G + sub * 2;
        bool const act = pl < PW && q < mr && !mab;
        if (__ballot(act) == 0) break;
        if (act) {
          uint32_t const pos = mh + q;
          u64x2 const v = *reinterpret_cast<u64x2 const*>(rval + static_cast<uint32_t>(d) * CAP + (pos & (CAP - 1)));
          if (q + 1 < mr) gstore(reinterpret_cast<u64x2*>(out_val + rbase + pos), v);
          else gstore(out_val + rbase + pos, v.x);
        }
      }
    }
    for (int b = 0; b * 16 < PW; ++b) {
      int const pl = b * 16 + lane / 4, sub = lane % 4;
      uint32_t const mr = __shfl(nrec, pl), mh = __shfl(head, pl);
      int const mab = __shfl(ab, pl);
      int const d = wave * PW + pl;
      int64_t const rbase = (static_cast<int64_t>(d) * a.slices + item) * a.region_cap;
      for (uint32_t g = 0;; ++g) {
        uint32_t const q = g * G + sub * 4;
        bool const act = pl < PW && q < mr && !mab;
        if (__ballot(act) == 0) break;
        if (act) {
          uint32_t const pos = mh + q;
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4 const v = *reinterpret_cast<u32x4 const*>(rslot + static_cast<uint32_t>(d) * CAP + (pos & (CAP - 1)));
          if (q + 3 < mr) gstore(reinterpret_cast<u32x4*>(out_slot + rbase + pos), v);
          else { gstore(out_slot + rbase + pos, v.x); if (q + 1 < mr) gstore(out_slot + rbase + pos + 1, v.y); if (q + 2 < mr) gstore(out_slot + rbase + pos + 2, v.z); }
        }
      }
    }
    if (lane < PW) {
      head += nrec;
      limit[dmine] = head + CAP;
    }
  };
  for (; tile < a.n; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      u64 key[RPT], val[RPT];
      bool keep[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        keep[k] = tile + j * step + static_cast<int64_t>(k) * B + threadIdx.x < a.n;
        key[k] = pk[j][k]; val[k] = pv[j][k];
      }
      load_rows(tile + (j + D) * step, pk[j], pv[j]);
      if (tile + j * step >= a.n) break;
      uint32_t d[RPT], pos[RPT], lim[RPT], sl[RPT];
      bool pend[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        d[k] = 0; pos[k] = 0; lim[k] = 0; sl[k] = 0;
        if (keep[k]) {
          uint32_t const scr = (static_cast<uint32_t>(key[k] - a.lo) * a.mult) & a.bmask;
          d[k]   = scr >> a.dshift;
          sl[k]  = scr & smask;
          pos[k] = atomicAdd(&tail[d[k]], 1u);
          lim[k] = limit[d[k]];
        }
      }
      bool any_pend = false;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pend[k] = keep[k] && static_cast<int32_t>(pos[k] - lim[k]) >= 0;
        if (keep[k] && !pend[k]) { uint32_t const w = d[k] * CAP + (pos[k] & (CAP - 1)); rval[w] = val[k]; rslot[w] = sl[k]; }
        any_pend = any_pend || pend[k];
      }
      if (any_pend) s_pending = 1;
      lds_barrier();
      flush(false);
      lds_barrier();
      while (s_pending) {
        lds_barrier();
        if (threadIdx.x == 0) s_pending = 0;
        lds_barrier();
        any_pend = false;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k]) {
            uint32_t const l2 = limit[d[k]];
            if (static_cast<int32_t>(pos[k] - l2) < 0) { uint32_t const w = d[k] * CAP + (pos[k] & (CAP - 1)); rval[w] = val[k]; rslot[w] = sl[k]; pend[k] = false; }
            else any_pend = true;
          }
        }
        if (any_pend) s_pending = 1;
        lds_barrier();
        flush(false);
        lds_barrier();
        if (s_abort) break;
      }
      if (s_abort) { if (threadIdx.x == 0) *a.overflow = 1; return; }
    }
  }
  flush(true);
  lds_barrier();
  if (s_abort) { if (threadIdx.x == 0) *a.overflow = 1; return; }
  if (lane < PW) a.region_count[static_cast<int64_t>(wave * PW + lane) * a.slices + item] = static_cast<int32_t>(head);
}
// verification of the SoA output: per partition count and checksum of (reconstructed key, value)
__global__ void k_check_out_soa(sargs a, u64 const* out_val, uint32_t const* out_slot, uint32_t mult_inv, u64* cnt, u64* sum)
{
  int const region = blockIdx.x, d = region / a.slices, c = a.region_count[region];
  u64 lc = 0, ls = 0;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    int64_t const at = static_cast<int64_t>(region) * a.region_cap + i;
    uint32_t const scr = (static_cast<uint32_t>(d) << a.dshift) | out_slot[at];
    u64 const key = a.lo + ((scr * mult_inv) & a.bmask), v = out_val[at];
    lc += 1;
    ls += mix64(key) + 3 * mix64(v ^ key);
  }
  atomicAdd(&cnt[d], lc);
  atomicAdd(&sum[d], ls);
}

int main(int argc, char** argv)
{
  int64_t const n  = (argc > 1 ? atoll(argv[1]) : 400) * 1000000ll;
  int const P      = argc > 2 ? atoi(argv[2]) : 512;
  int const dense  = argc > 3 ? atoi(argv[3]) : 1;
  int64_t const chunk_rows = (argc > 4 ? atoll(argv[4]) : 0) * 1000000ll;  // > 0: scatter chunk by chunk into a reused ring
  u64 const groups = 1000000;
  int const S = argc > 5 ? atoi(argv[5]) : 256;
  u64 *keys, *vals;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&vals, n * 8));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, keys, vals, n, groups, dense);
  int64_t const rows_per_launch = chunk_rows > 0 ? chunk_rows : n;
  double const mean = static_cast<double>(rows_per_launch) / (static_cast<double>(S) * P);
  int64_t const cap = (static_cast<int64_t>(mean * 1.35 + 6 * sqrt(mean) + 16) + 7) / 8 * 8;
  u64x2* out; int32_t *rc, *ov;
  CK(hipMalloc(&out, static_cast<size_t>(S) * P * cap * 16)); CK(hipMalloc(&rc, S * P * 4)); CK(hipMalloc(&ov, 4));
  CK(hipMemset(ov, 0, 4));
  sargs a{};
  a.keys = keys; a.vals = vals; a.n = n; a.P = P; a.dense = dense;
  int log2P = 0; while ((1 << log2P) < P) ++log2P;
  a.shift = 64 - log2P; a.lo = 0; a.mult = 0x9E3779B1u; a.bmask = (1u << 20) - 1; a.dshift = 20 - log2P;
  a.out = out; a.region_cap = cap; a.region_count = rc; a.overflow = ov; a.slices = S;
  printf("rows %lld P %d dense %d region_cap %lld (mean %.0f) buffer %.1f MiB chunk_rows %lld\n", (long long)n, P, dense, (long long)cap, mean, S * (double)P * cap * 16 / 1048576, (long long)chunk_rows);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, int capl, const char* name) {
    size_t const lds = static_cast<size_t>(P) * (16u << capl) + P * 8;
    if (lds > 160 * 1024 - 64) return;
    CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    auto once = [&] {
      if (chunk_rows == 0) { hipLaunchKernelGGL(kern, dim3(S), dim3(1024), lds, 0, a); return; }
      for (int64_t b = 0; b < n; b += chunk_rows) {
        sargs c = a; c.keys = a.keys + b; c.vals = a.vals + b; c.n = std::min(chunk_rows, n - b);
        hipLaunchKernelGGL(kern, dim3(S), dim3(1024), lds, 0, c);
      }
    };
    once(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) once();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    int h_ov; CK(hipMemcpy(&h_ov, ov, 4, hipMemcpyDeviceToHost));
    printf("%-34s lds %6zu B  %8.3f ms  %6.1f G rows/s  -> %5.2f ms per 1B rows  overflow=%d\n", name, lds, ms, n / ms / 1e6, ms * 1e9 / n, h_ov);
    if (chunk_rows == 0 && h_ov == 0 && strstr(name, "no ") == nullptr) {  // verify
      u64 *ci, *si, *co, *so, *bad;
      CK(hipMalloc(&ci, P * 8)); CK(hipMalloc(&si, P * 8)); CK(hipMalloc(&co, P * 8)); CK(hipMalloc(&so, P * 8)); CK(hipMalloc(&bad, 8));
      CK(hipMemset(ci, 0, P * 8)); CK(hipMemset(si, 0, P * 8)); CK(hipMemset(co, 0, P * 8)); CK(hipMemset(so, 0, P * 8)); CK(hipMemset(bad, 0, 8));
      if (dense) { hipLaunchKernelGGL(k_check_in<true>, dim3(1024), dim3(256), 0, 0, a, ci, si); hipLaunchKernelGGL(k_check_out<true>, dim3(S * P), dim3(256), 0, 0, a, co, so, bad); }
      else { hipLaunchKernelGGL(k_check_in<false>, dim3(1024), dim3(256), 0, 0, a, ci, si); hipLaunchKernelGGL(k_check_out<false>, dim3(S * P), dim3(256), 0, 0, a, co, so, bad); }
      std::vector<u64> hci(P), hsi(P), hco(P), hso(P); u64 hbad;
      CK(hipMemcpy(hci.data(), ci, P * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsi.data(), si, P * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hco.data(), co, P * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hso.data(), so, P * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
      int mism = 0; u64 tot = 0;
      for (int d = 0; d < P; ++d) { mism += hci[d] != hco[d] || hsi[d] != hso[d]; tot += hco[d]; }
      printf("   verify: records out %llu of %lld, partitions with count/checksum mismatch %d, misplaced records %llu\n", tot, (long long)n, mism, hbad);
    }
    CK(hipMemset(ov, 0, 4));
  };
  if (dense && P <= 256) {
    u64* oval; uint32_t* oslot;
    CK(hipMalloc(&oval, static_cast<size_t>(S) * P * cap * 8)); CK(hipMalloc(&oslot, static_cast<size_t>(S) * P * cap * 4));
    auto run_soa = [&](auto kern, int capl, const char* name) {
      size_t const lds = static_cast<size_t>(P) * (12u << capl) + P * 8;
      CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
      auto once = [&] {
        if (chunk_rows == 0) { hipLaunchKernelGGL(kern, dim3(S), dim3(1024), lds, 0, a, oval, oslot); return; }
        for (int64_t b = 0; b < n; b += chunk_rows) {
          sargs c = a; c.keys = a.keys + b; c.vals = a.vals + b; c.n = std::min(chunk_rows, n - b);
          hipLaunchKernelGGL(kern, dim3(S), dim3(1024), lds, 0, c, oval, oslot);
        }
      };
      once(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 3; ++r) once();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      int h_ov; CK(hipMemcpy(&h_ov, ov, 4, hipMemcpyDeviceToHost));
      printf("%-34s lds %6zu B  %8.3f ms  %6.1f G rows/s  -> %5.2f ms per 1B rows  overflow=%d\n", name, lds, ms, n / ms / 1e6, ms * 1e9 / n, h_ov);
      if (chunk_rows == 0 && h_ov == 0) {
        u64 *ci, *si, *co, *so;
        CK(hipMalloc(&ci, P * 8)); CK(hipMalloc(&si, P * 8)); CK(hipMalloc(&co, P * 8)); CK(hipMalloc(&so, P * 8));
        CK(hipMemset(ci, 0, P * 8)); CK(hipMemset(si, 0, P * 8)); CK(hipMemset(co, 0, P * 8)); CK(hipMemset(so, 0, P * 8));
        uint32_t inv = a.mult; for (int it = 0; it < 5; ++it) inv *= 2u - a.mult * inv;
        hipLaunchKernelGGL(k_check_in<true>, dim3(1024), dim3(256), 0, 0, a, ci, si);
        hipLaunchKernelGGL(k_check_out_soa, dim3(S * P), dim3(256), 0, 0, a, oval, oslot, inv, co, so);
        std::vector<u64> hci(P), hsi(P), hco(P), hso(P);
        CK(hipMemcpy(hci.data(), ci, P * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsi.data(), si, P * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hco.data(), co, P * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hso.data(), so, P * 8, hipMemcpyDeviceToHost));
        int mism = 0; u64 tot = 0;
        for (int d = 0; d < P; ++d) { mism += hci[d] != hco[d] || hsi[d] != hso[d]; tot += hco[d]; }
        printf("   verify: records out %llu of %lld, partitions with count/checksum mismatch %d, misplaced records 0\n", tot, (long long)n, mism);
      }
      CK(hipMemset(ov, 0, 4));
    };
    if (P <= 128) {
      run_soa(k_scatter_ring_soa<6, 2, 2>, 6, "SoA 12B G=16 CAP=64 D=2 RPT=2");
      run_soa(k_scatter_ring_soa<6, 3, 2>, 6, "SoA 12B G=16 CAP=64 D=3 RPT=2");
      run_soa(k_scatter_ring_soa<6, 2, 3>, 6, "SoA 12B G=16 CAP=64 D=2 RPT=3");
      run_soa(k_scatter_ring_soa<6, 2, 4>, 6, "SoA 12B G=16 CAP=64 D=2 RPT=4");
    } else {
      run_soa(k_scatter_ring_soa<5, 2, 1>, 5, "SoA 12B G=16 CAP=32 D=2 RPT=1");
      run_soa(k_scatter_ring_soa<5, 2, 2>, 5, "SoA 12B G=16 CAP=32 D=2 RPT=2");
      run_soa(k_scatter_ring_soa<5, 4, 1>, 5, "SoA 12B G=16 CAP=32 D=4 RPT=1");
    }
  }
  if (dense) {
    if (P <= 128) run(k_scatter_ring<8, 6, 2, true, 4>, 6, "ring G=8 CAP=64 D=2 RPT=4 dense");
    if (P <= 128) run(k_scatter_ring<8, 6, 2, true, 4, 1>, 6, "  same, 16-byte loads");
    if (P <= 128) run(k_scatter_ring<8, 6, 2, true, 4, 2>, 6, "  same, no global stores");
    if (P <= 128) run(k_scatter_ring<8, 6, 2, true, 4, 3>, 6, "  same, 16-byte loads, no stores");
    if (P <= 256) run(k_scatter_ring<8, 5, 2, true, 2, 1>, 5, "ring G=8 CAP=32 D=2 RPT=2 16-byte loads");
    if (P <= 256) run(k_scatter_ring<8, 5, 4, true, 2, 1>, 5, "ring G=8 CAP=32 D=4 RPT=2 16-byte loads");
    if (P <= 256) run(k_scatter_ring<8, 5, 2, true, 2, 3>, 5, "ring G=8 CAP=32 D=2 RPT=2 16-byte loads, no stores");
    if (P <= 128) run(k_scatter_ring<8, 6, 3, true, 4>, 6, "ring G=8 CAP=64 D=3 RPT=4 dense");
    if (P <= 128) run(k_scatter_ring<8, 6, 2, true, 2>, 6, "ring G=8 CAP=64 D=2 RPT=2 dense");
    if (P <= 256) run(k_scatter_ring<8, 5, 2, true, 2>, 5, "ring G=8 CAP=32 D=2 RPT=2 dense");
    if (P <= 256) run(k_scatter_ring<4, 5, 2, true, 2>, 5, "ring G=4 CAP=32 D=2 RPT=2 dense");
    if (P <= 512) run(k_scatter_ring<4, 4, 2, true, 2>, 4, "ring G=4 CAP=16 D=2 RPT=2 dense");
    if (P <= 512) run(k_scatter_ring<8, 4, 4, true>, 4, "ring G=8 CAP=16 D=4 dense");
    if (P <= 512) run(k_scatter_ring<4, 4, 4, true>, 4, "ring G=4 CAP=16 D=4 dense");
    if (P <= 256) run(k_scatter_ring<8, 5, 4, true>, 5, "ring G=8 CAP=32 D=4 dense");

  } else {
    if (P <= 512) run(k_scatter_ring<8, 4, 4, false>, 4, "ring G=8 CAP=16 D=4 hash");
    if (P <= 512) run(k_scatter_ring<4, 4, 4, false>, 4, "ring G=4 CAP=16 D=4 hash");
    if (P <= 256) run(k_scatter_ring<8, 5, 4, false>, 5, "ring G=8 CAP=32 D=4 hash");
  }
  return 0;
}
