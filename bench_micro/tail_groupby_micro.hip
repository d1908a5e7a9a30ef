// Prototype of the TAILING dense-key groupby (round 4, after fused_groupby_micro.hip showed that weaving the aggregate into the
// scatter's barrier cadence only adds its time): two kernels that run AT THE SAME TIME on every CU,
//   A  k_scatter_pub  (1024 threads, LDS rings): the ring scatter into 256 partitions, lines of 12 records
//      ([12 x f64 value | 12 x u16 slot | 8 spare bytes], written through with sc1), one region per (workgroup, partition) sized for
//      the whole call; every wave publishes the line counts of its 16 partitions once per tile, one tile late, behind a COUNTED
//      s_waitcnt that proves the stores of the previous tile have completed (no drain, no barrier);
//   B  k_tail  (256 threads, one workgroup per partition = per CU): keeps the partition's direct-address table (4096 x 12 bytes) in
//      LDS for the whole call, polls the 256 counters of its partition and accumulates the lines a few microseconds after they were
//      written - while they are still in the Infinity Cache.
// There is no flow control (A never waits for B), so no schedule can deadlock; if B is not co-resident it only reads from HBM.
//   ./tail_groupby_micro [rows_millions] [mode: 0 both concurrently, 1 A then B] [groups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T gload(T const* p) { return *(GAS T const*)(p); }
template <typename T> __device__ __forceinline__ void gstore(T* p, T v) { *(GAS T*)(p) = v; }
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int NP       = 256;   // partitions = workgroups of either kernel = CUs
constexpr int RPL      = 12;    // records per line
constexpr int RLINES   = 3;     // ring lines per partition
constexpr int RCAP     = RPL * RLINES;
constexpr int SLOTS    = 4096;  // table slots per partition
constexpr uint32_t FINAL = 0x80000000u;
constexpr uint32_t PAD   = SLOTS;  // the tag of a padding record: a dummy slot behind the table

struct targs {
  u64 const* keys; u64 const* vals; int64_t n;
  u64 lo, range; uint32_t mult, bmask;
  unsigned char* cells;   // [NP src][NP dst][region_lines][128 bytes]
  uint32_t region_lines;
  uint32_t* produced;     // [src][dst] lines written and complete (| FINAL)
  int32_t* status;        // [0] abort, [1] key out of range, [2] region overflow, [3] consumer timeouts
  double* out_sum; uint32_t* out_cnt;  // [NP][SLOTS]
  long long timeout_ticks;
  u64* stats;             // [0] consumer rounds, [1] empty rounds, [2] lines, [3] max lag lines
};

__device__ __forceinline__ uint32_t div12(uint32_t x) { return __umulhi(x, 0xAAAAAAABu) >> 3; }
__device__ __forceinline__ uint32_t mod3(uint32_t x) { return x - 3u * (__umulhi(x, 0xAAAAAAABu) >> 1); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void const* p, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(bytes), 0x00020000);
}
// 16-byte write-through store the compiler does not track (see fused_groupby_micro.hip)
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store4_sc1(void* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// ------------------------------------------------------------------------------------------------ A: scatter + publish
template <int RPT, int D, bool PUBLISH>
__global__ void __launch_bounds__(1024, 5) k_scatter_pub(targs a)  // (96 registers: a wave of B fits beside four of these on a SIMD)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int B = 1024;
  unsigned char* ring = lds;                                                 // NP * 384
  uint32_t* tail      = reinterpret_cast<uint32_t*>(lds + NP * RLINES * 128);  // NP
  uint32_t* limit     = tail + NP;
  __shared__ int s_pending[2], s_soft[2], s_abort;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 7;
  int const me = blockIdx.x;
  if (threadIdx.x == 0 && a.stats) atomicMin(a.stats + 4, static_cast<u64>(wall_clock64()));
  for (int i = threadIdx.x; i < NP; i += B) { tail[i] = 0; limit[i] = RCAP; }
  if (threadIdx.x < 2) { s_pending[threadIdx.x] = 0; s_soft[threadIdx.x] = 0; }
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int64_t const step = static_cast<int64_t>(NP) * T, end = a.n;
  u64 pk[D][RPT], pv[D][RPT];
  uint32_t const voff = threadIdx.x * 8u;
  auto issue = [&](int64_t tile, u64 (&kk)[RPT], u64 (&vv)[RPT]) {
    int64_t left = end - tile;
    int64_t const base = left > 0 ? tile : 0;
    left = left > 0 ? left : 0;
    uint32_t const bytes = left > 0x1fffffff ? 0xfffffff8u : static_cast<uint32_t>(left) * 8u;
    __amdgpu_buffer_rsrc_t const rk = make_rsrc(a.keys + base, bytes), rv = make_rsrc(a.vals + base, bytes);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      kk[k] = __builtin_bit_cast(u64, __builtin_amdgcn_raw_buffer_load_b64(rk, static_cast<int>(voff), k * B * 8, 0));
      vv[k] = __builtin_bit_cast(u64, __builtin_amdgcn_raw_buffer_load_b64(rv, static_cast<int>(voff), k * B * 8, 0));
    }
  };
  uint32_t head = 0;  // owner lanes (lane < 16): lines of partition wave * 16 + lane flushed
  uint32_t const region_lines = a.region_lines;
  uint32_t* const my_counts = a.produced + me * NP + wave * 16 + (lane & 15);
  bool bad_key = false;

  // flush every complete line of this wave's partitions. COUNTED: the only vector-memory operations this wave has issued since the
  // stores of its previous flush are the 2 * RPT loads of one tile, so `s_waitcnt vmcnt(2 * RPT)` proves those stores have completed
  // and the counts of the previous flush can be published - one tile late, without draining anything.
  auto flush = [&](int ph, bool counted) {
    uint32_t nl = 0;
    int const dmine = wave * 16 + lane;
    if constexpr (PUBLISH) {
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane < 16) store4_sc1(my_counts, head);  // (the lines of every earlier flush)
    }
    if (lane < 16) {
      uint32_t const t = tail[dmine], limv = head * RPL + RCAP;
      uint32_t const c = static_cast<int32_t>(t - limv) < 0 ? t : limv;
      nl = div12(c) - head;
      if (head + nl > region_lines) { s_abort = 1; nl = 0; }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int const pl = b * 8 + (lane >> 3);
      uint32_t const mn = __shfl(nl, pl), mh = __shfl(head, pl);
      uint32_t const d  = static_cast<uint32_t>(wave * 16 + pl);
      unsigned char* const region = a.cells + (static_cast<uint64_t>(me) * NP + d) * region_lines * 128u;
      for (uint32_t g = 0;; ++g) {
        bool const act = g < mn;
        if (__ballot(act) == 0) break;
        if (act) {
          uint32_t const L = mh + g;
          u32x4 const v = *reinterpret_cast<u32x4 const*>(ring + d * (RLINES * 128) + mod3(L) * 128 + sub * 16);
          store16_sc1(region + static_cast<uint64_t>(L) * 128u + sub * 16, v);
        }
      }
    }
    if (lane < 16) {
      head += nl;
      uint32_t const newlim = head * RPL + RCAP;
      limit[dmine] = newlim;
      if (static_cast<int32_t>(tail[dmine] - newlim) > 0) s_pending[ph] = 1;  // rows that still do not fit: another round
    }
  };

  int ph = 0;
#pragma unroll
  for (int j = 0; j < D; ++j) issue(static_cast<int64_t>(me) * T + j * step, pk[j], pv[j]);
  for (int64_t tile = static_cast<int64_t>(me) * T; tile < end; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      int64_t const t0 = tile + j * step;
      int64_t const left64 = end - t0;
      uint32_t const rows_left = left64 <= 0 ? 0u : (left64 > 0x7fffffff ? 0x7fffffffu : static_cast<uint32_t>(left64));
      bool pend[RPT];
      uint32_t d[RPT], tg[RPT], pos[RPT];
      u64 val[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pend[k] = static_cast<uint32_t>(k * B) + threadIdx.x < rows_left;  // (until placed)
        u64 idx = pk[j][k] - a.lo;
        if (idx >= a.range) { bad_key = bad_key || pend[k]; idx = 0; }
        uint32_t const x = (static_cast<uint32_t>(idx) * a.mult) & a.bmask;
        d[k]   = x >> 12;
        tg[k]  = x & 0xFFFu;
        val[k] = pv[j][k];
      }
      issue(t0 + D * step, pk[j], pv[j]);
      auto place = [&](int k) {
        uint32_t const q = div12(pos[k]), r = pos[k] - q * RPL;
        unsigned char* line = ring + d[k] * (RLINES * 128) + mod3(q) * 128;
        *reinterpret_cast<u64*>(line + r * 8) = val[k];
        *reinterpret_cast<uint16_t*>(line + 96 + r * 2) = static_cast<uint16_t>(tg[k]);
      };
      uint32_t lim[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pos[k] = 0; lim[k] = 0;
        if (pend[k]) { pos[k] = atomicAdd(&tail[d[k]], 1u); lim[k] = limit[d[k]]; }
      }
      bool waits = false;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (pend[k] && static_cast<int32_t>(pos[k] - lim[k]) < 0) { place(k); pend[k] = false; }
        waits = waits || pend[k];
      }
      if (waits) s_soft[ph] = 1;
      lds_barrier();
      if (threadIdx.x == 0) { s_pending[ph ^ 1] = 0; s_soft[ph ^ 1] = 0; }
      flush(ph, true);
      lds_barrier();
      // rows that found their ring full: its lines have left by now (the owner raised s_pending[ph] if some row still does not fit)
      if (s_soft[ph]) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) { place(k); pend[k] = false; }
        }
      }
      while (s_pending[ph] && !s_abort) {  // (rare: a partition took more rows of one tile than a ring holds)
        ph ^= 1;
        lds_barrier();
        if (threadIdx.x == 0) { s_pending[ph ^ 1] = 0; s_soft[ph ^ 1] = 0; }
        flush(ph, false);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) { place(k); pend[k] = false; }
        }
      }
      ph ^= 1;
    }
    if (s_abort) break;
  }
  if (bad_key) atomicOr(a.status + 1, 1);
  // pad the partial lines, flush them, publish the final counts
  lds_barrier();
  if (lane < 16) {
    int const dmine = wave * 16 + lane;
    uint32_t const t = tail[dmine], q = div12(t), r = t - q * RPL;
    if (r != 0) {
      unsigned char* line = ring + dmine * (RLINES * 128) + mod3(q) * 128;
      for (uint32_t e = r; e < RPL; ++e) *reinterpret_cast<uint16_t*>(line + 96 + e * 2) = static_cast<uint16_t>(PAD);
      tail[dmine] = t + RPL - r;
    }
  }
  lds_barrier();
  flush(0, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (s_abort) {
    if (threadIdx.x == 0) { atomicOr(a.status, 1); atomicOr(a.status + 2, 1); }
  }
  if (lane < 16) store4_sc1(my_counts, head | FINAL);  // (an aborted call ends too: its consumers see FINAL and stop)
  if (threadIdx.x == 0 && a.stats) atomicMax(a.stats + 5, static_cast<u64>(wall_clock64()));
}

// ------------------------------------------------------------------------------------------------ B: tailing aggregate
// One workgroup per partition, NW waves; wave w polls the counters of the sources [w * 256 / NW, ...) (one per lane), gathers the lines
// they have ready into batches of 8 (8 lanes per line: lanes 0..5 two records each) and keeps U batches of loads in flight (a rolling
// ring of registers: batch i + U is issued when batch i has been accumulated).
template <int NW, int U, int MAXL>
__device__ __forceinline__ void tail_body(targs const& a)
{
  // (dynamic LDS: with 57 KB of STATIC LDS the compiler knows that at most two of these workgroups fit a CU and pads the register
  // allocation of the descriptor to 169 so that no third wave fits a SIMD - and then the wave does not fit beside four waves of A)
  extern __shared__ __attribute__((aligned(16))) unsigned char blds[];
  constexpr int SPW = NP / NW;  // sources per wave
  double* tsum   = reinterpret_cast<double*>(blds);                          // SLOTS + 2
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(tsum + SLOTS + 2);            // SLOTS + 4
  uint32_t (*wl)[SPW * MAXL] = reinterpret_cast<uint32_t (*)[SPW * MAXL]>(tcnt + SLOTS + 4);  // [NW][SPW * MAXL]
  __shared__ int s_flag[2];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 7, grp = lane >> 3;
  int const p = blockIdx.x;
  if (threadIdx.x == 0 && a.stats) { u64 const now = static_cast<u64>(wall_clock64()); atomicMin(a.stats + 6, now); atomicMax(a.stats + 7, now); }
  for (int i = threadIdx.x; i < SLOTS + 1; i += NW * 64) { tsum[i] = 0.0; tcnt[i] = 0; }
  if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
  __syncthreads();
  bool const poller = lane < SPW;
  uint32_t const src = static_cast<uint32_t>(wave * SPW + (poller ? lane : 0));
  uint32_t cons = 0, prod = 0;
  bool fin = false;
  uint32_t const region_lines = a.region_lines;
  unsigned char const* const my_cells = a.cells + static_cast<uint64_t>(p) * region_lines * 128u;  // + src * NP * region_lines * 128
  uint64_t const src_stride = static_cast<uint64_t>(NP) * region_lines * 128u;
  uint32_t const sb = sub < 6 ? sub : 0;
  long long const t_start = wall_clock64();
  u64 n_round = 0, n_empty = 0, n_lines = 0, max_lag = 0;
  auto accumulate = [&](u32x4 v, uint32_t t) {
    uint32_t const t0 = t & 0xFFFFu, t1 = t >> 16;
    double const v0 = __longlong_as_double(static_cast<long long>(static_cast<u64>(v.x) | (static_cast<u64>(v.y) << 32)));
    double const v1 = __longlong_as_double(static_cast<long long>(static_cast<u64>(v.z) | (static_cast<u64>(v.w) << 32)));
    atomicAdd(&tsum[t0], v0); atomicAdd(&tcnt[t0], 1u);
    atomicAdd(&tsum[t1], v1); atomicAdd(&tcnt[t1], 1u);
  };
  for (int round = 0;; ++round) {
    ++n_round;
    uint32_t navail = 0;
    if (poller) {
      uint32_t const pv = __hip_atomic_load(a.produced + src * NP + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prod = pv & ~FINAL;
      fin  = (pv & FINAL) != 0;
      navail = prod - cons;
      if (navail > max_lag) max_lag = navail;
      navail = navail < static_cast<uint32_t>(MAXL) ? navail : static_cast<uint32_t>(MAXL);
    }
    // inclusive scan of navail over the wave
    uint32_t inc = navail;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t const up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    uint32_t const total = __shfl(inc, 63);
    uint32_t const exc = inc - navail;
    for (uint32_t i = 0; i < navail; ++i) wl[wave][exc + i] = (static_cast<uint32_t>(lane) << 16) | (cons + i);
    cons += navail;
    n_lines += navail;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (total > 0) {  // (uniform over the wave)
      uint32_t const nb = (total + 7) >> 3;  // batches of 8 lines
      u32x4 cv[U]; uint32_t ct[U];
      auto issue = [&](uint32_t b, u32x4& v, uint32_t& t) {  // (unconditional: a batch past the end reads the last line again)
        uint32_t q = b * 8 + grp;
        q = q < total ? q : total - 1;
        uint32_t const e = wl[wave][q];
        unsigned char const* line = my_cells + static_cast<uint64_t>(wave * SPW + (e >> 16)) * src_stride + static_cast<uint64_t>(e & 0xFFFFu) * 128u;
        v = gload(reinterpret_cast<u32x4 const*>(line + sb * 16));
        t = gload(reinterpret_cast<uint32_t const*>(line + 96 + sb * 4));
      };
#pragma unroll
      for (int u = 0; u < U; ++u) issue(static_cast<uint32_t>(u), cv[u], ct[u]);
      for (uint32_t b0 = 0; b0 < nb; b0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          uint32_t const b = b0 + u;
          bool const have = b < nb && b * 8 + grp < total && sub < 6;
          u32x4 const v = cv[u]; uint32_t const t = ct[u];
          issue(b + U, cv[u], ct[u]);
          if (have) accumulate(v, t);
        }
      }
    }
    // done when every source is final and consumed
    bool const mine_done = !poller || (fin && cons == prod);
    int const fl = round & 1;
    if (!mine_done) s_flag[fl] = 1;
    if (threadIdx.x == 0 && (round & 15) == 15) {
      if (wall_clock64() - t_start > a.timeout_ticks) { atomicOr(a.status, 1); atomicAdd(a.status + 3, 1); }
      if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_flag[fl] = 2;
    }
    __syncthreads();
    int const f = s_flag[fl];
    if (threadIdx.x == 0) s_flag[fl ^ 1] = 0;
    if (f == 0) break;
    if (f == 2) return;  // aborted
    if (__syncthreads_or(total != 0) == 0) { ++n_empty; __builtin_amdgcn_s_sleep(16); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SLOTS; i += NW * 64) {
    gstore(a.out_sum + static_cast<int64_t>(p) * SLOTS + i, tsum[i]);
    gstore(a.out_cnt + static_cast<int64_t>(p) * SLOTS + i, tcnt[i]);
  }
  if (a.stats) {
    if (lane == 0) { atomicAdd(a.stats + 2, n_lines); atomicMax(a.stats + 3, max_lag); }
    if (threadIdx.x == 0) { atomicAdd(a.stats + 0, n_round); atomicAdd(a.stats + 1, n_empty); }
  }
}

// (8 waves: 64 registers each - two of them beside four 96-register waves of A on a SIMD; 4 waves: 128)
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(64))) k_tail8(targs a) { tail_body<8, 4, 8>(a); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) k_tail4(targs a) { tail_body<4, 8, 8>(a); }

// co-residency probes (which of the two real kernels keeps the other one out of its CU?)
template <bool BAR>
__global__ void __launch_bounds__(1024) k_probe_long(u64* stats, long long ticks)
{
  extern __shared__ char plds[];
  asm volatile("v_mov_b32 v95, 0" ::: "v95");
  asm volatile("s_mov_b32 s95, 0" ::: "s95");
  long long const t0 = wall_clock64();
  if (threadIdx.x == 0) { atomicMin(stats + 4, static_cast<u64>(t0)); plds[0] = 1; }
  if constexpr (BAR) __syncthreads();
  if (threadIdx.x == 0) {
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    atomicMax(stats + 5, static_cast<u64>(wall_clock64()));
  }
  if constexpr (BAR) __syncthreads();
}
template <bool BAR>
__global__ void __launch_bounds__(256) k_probe_short(u64* stats)
{
  extern __shared__ char plds[];
  asm volatile("v_mov_b32 v79, 0" ::: "v79");
  asm volatile("s_mov_b32 s65, 0" ::: "s65");
  if (threadIdx.x == 0) { plds[0] = 1; u64 const now = static_cast<u64>(wall_clock64()); atomicMin(stats + 6, now); atomicMax(stats + 7, now); }
  if constexpr (BAR) __syncthreads();
}
__global__ void k_fill(u64* keys, u64* vals, int64_t n, u64 groups)
{
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < n; r += stride) {
    keys[r] = mix64(r * 0x9e3779b97f4a7c15ull + 1) % groups;
    double const v = static_cast<double>(mix64(r + 12345) >> 11) * (1.0 / 9007199254740992.0);
    vals[r] = static_cast<u64>(__double_as_longlong(v));
  }
}
__global__ void k_reference(targs a, double* sum, uint32_t* cnt)
{
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < a.n; r += stride) {
    uint32_t const x = (static_cast<uint32_t>(a.keys[r] - a.lo) * a.mult) & a.bmask;
    atomicAdd(&sum[x], __longlong_as_double(static_cast<long long>(a.vals[r])));
    atomicAdd(&cnt[x], 1u);
  }
}

int main(int argc, char** argv)
{
  int64_t const n  = static_cast<int64_t>((argc > 1 ? atof(argv[1]) : 400) * 1000000.0);
  int const mode   = argc > 2 ? atoi(argv[2]) : 0;
  u64 const groups = argc > 3 ? atoll(argv[3]) : 1000000;
  int const reps   = getenv("REPS") ? atoi(getenv("REPS")) : 3;
  u64 *keys, *vals;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&vals, n * 8));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, keys, vals, n, groups);
  targs a{};
  a.keys = keys; a.vals = vals; a.n = n; a.lo = 0; a.range = 1u << 20; a.mult = 0x9E3779B1u; a.bmask = (1u << 20) - 1;
  double const mean_lines = static_cast<double>(n) / (NP * NP) / RPL;
  a.region_lines = static_cast<uint32_t>(mean_lines * 1.05 + 8 * sqrt(mean_lines) + 8);
  size_t const cells_bytes = static_cast<size_t>(NP) * NP * a.region_lines * 128;
  CK(hipMalloc(&a.cells, cells_bytes)); CK(hipMalloc(&a.produced, NP * NP * 4));
  CK(hipMalloc(&a.status, 16)); CK(hipMalloc(&a.stats, 64));
  CK(hipMalloc(&a.out_sum, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMalloc(&a.out_cnt, static_cast<size_t>(NP) * SLOTS * 4));
  a.timeout_ticks = 100000000ll / 10;  // 0.1 s at 100 MHz
  double* rs; uint32_t* rc;
  CK(hipMalloc(&rs, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMalloc(&rc, static_cast<size_t>(NP) * SLOTS * 4));
  CK(hipMemset(rs, 0, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMemset(rc, 0, static_cast<size_t>(NP) * SLOTS * 4));
  hipLaunchKernelGGL(k_reference, dim3(4096), dim3(256), 0, 0, a, rs, rc);
  std::vector<double> hrs(static_cast<size_t>(NP) * SLOTS); std::vector<uint32_t> hrc(static_cast<size_t>(NP) * SLOTS);
  CK(hipMemcpy(hrs.data(), rs, hrs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hrc.data(), rc, hrc.size() * 4, hipMemcpyDeviceToHost));
  int dev; CK(hipGetDevice(&dev)); hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("rows %lld groups %llu CUs %d region_lines %u (mean %.0f) cells %.2f GB\n", (long long)n, groups, prop.multiProcessorCount, a.region_lines, mean_lines, cells_bytes / 1e9);
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, ea, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  size_t const lds_a = NP * RLINES * 128 + 2 * NP * 4;
  size_t const lds_b = (SLOTS + 2) * 8 + (SLOTS + 4) * 4 + NP * 8 * 4;

  auto verify = [&](const char* what) {
    int32_t st[4]; u64 stats[8];
    CK(hipMemcpy(st, a.status, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(stats, a.stats, 64, hipMemcpyDeviceToHost));
    std::vector<double> hs(static_cast<size_t>(NP) * SLOTS); std::vector<uint32_t> hc(static_cast<size_t>(NP) * SLOTS);
    CK(hipMemcpy(hs.data(), a.out_sum, hs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), a.out_cnt, hc.size() * 4, hipMemcpyDeviceToHost));
    size_t badc = 0, bads = 0; double maxrel = 0; u64 total = 0;
    for (size_t i = 0; i < hs.size(); ++i) {
      total += hc[i];
      if (hc[i] != hrc[i]) ++badc;
      double const e = fabs(hs[i] - hrs[i]), tol = 1e-9 * (fabs(hrs[i]) + 1);
      if (e > tol) ++bads;
      if (fabs(hrs[i]) > 0) maxrel = fmax(maxrel, e / fabs(hrs[i]));
    }
    printf("   [A start 0, A end %.1f us, B first start %.1f us, B last start %.1f us]\n", (double)(stats[5] - stats[4]) / 100.0, (double)(long long)(stats[6] - stats[4]) / 100.0, (double)(long long)(stats[7] - stats[4]) / 100.0);
    printf("   %s: status abort %d range %d overflow %d timeouts %d | consumer rounds/wg %.0f empty %.0f lines %llu max lag %llu lines | rows counted %llu of %lld, wrong counts %zu, wrong sums %zu, max rel err %.2e\n",
           what, st[0], st[1], st[2], st[3], stats[0] / 256.0, stats[1] / 256.0, stats[2], stats[3], total, (long long)n, badc, bads, maxrel);
  };
  auto reset = [&](hipStream_t s) {
    CK(hipMemsetAsync(a.produced, 0, NP * NP * 4, s)); CK(hipMemsetAsync(a.status, 0, 16, s)); CK(hipMemsetAsync(a.stats, 0, 64, s)); CK(hipMemsetAsync(a.stats + 4, 0xff, 8, s)); CK(hipMemsetAsync(a.stats + 6, 0xff, 8, s));
  };
  auto time_it = [&](const char* name, auto&& body) {
    float best = 1e30f, tot = 0;
    for (int r = 0; r < reps + 1; ++r) {
      reset(0);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sa));
      body();
      CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) { best = ms < best ? ms : best; tot += ms; }
    }
    printf("%-52s %8.3f ms avg %8.3f best  -> %5.2f ms per 1B rows\n", name, tot / reps, best, best * 1e9 / n);
  };
  auto kA  = k_scatter_pub<4, 2, true>;
  auto kA3 = k_scatter_pub<3, 2, true>;
  auto kA0 = k_scatter_pub<4, 2, false>;
  int const BW = getenv("BW4") ? 4 : 8;  // waves of B
  auto kB  = BW == 4 ? k_tail4 : k_tail8;
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kA), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_a)));
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kA3), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_a)));
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kA0), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_a)));
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kB), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
  if (mode == 0 || mode == 2) {
    time_it("A alone (scatter, publishing), RPT4", [&] { hipLaunchKernelGGL(kA, dim3(NP), dim3(1024), lds_a, sa, a); });
    time_it("A alone (scatter, no publishing), RPT4", [&] { hipLaunchKernelGGL(kA0, dim3(NP), dim3(1024), lds_a, sa, a); });
    time_it("A alone (scatter, publishing), RPT3", [&] { hipLaunchKernelGGL(kA3, dim3(NP), dim3(1024), lds_a, sa, a); });
    time_it("A then B on one stream, RPT4", [&] {
      hipLaunchKernelGGL(kA, dim3(NP), dim3(1024), lds_a, sa, a);
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sa, a);
    });
    verify("A then B");
  }
  if (mode == 3) {
    bool const use3 = getenv("RPT3") != nullptr;  // no event edges between the streams: host clock around both
    for (int r = 0; r < 4; ++r) {
      reset(0);
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      if (use3) hipLaunchKernelGGL(kA3, dim3(NP), dim3(1024), lds_a, sa, a); else hipLaunchKernelGGL(kA, dim3(NP), dim3(1024), lds_a, sa, a);
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sb, a);
      CK(hipDeviceSynchronize());
      double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("A on one stream, B on another, no events: host clock %.3f ms -> %.2f ms per 1B rows\n", ms, ms * 1e9 / n);
      verify("   no events");
    }
    for (int r = 0; r < 3; ++r) {
      reset(0);
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(kA, dim3(NP), dim3(1024), lds_a, sa, a);
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sa, a);
      CK(hipDeviceSynchronize());
      double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("A then B on ONE stream: host clock %.3f ms -> %.2f ms per 1B rows\n", ms, ms * 1e9 / n);
    }
  }
  if (mode == 4) {
    auto show = [&](const char* what) {
      u64 stats[8]; CK(hipMemcpy(stats, a.stats, 64, hipMemcpyDeviceToHost));
      printf("%-60s first kernel ends at %.1f us, second kernel's workgroups start at %.1f ... %.1f us\n", what, (double)(stats[5] - stats[4]) / 100.0,
             (double)(long long)(stats[6] - stats[4]) / 100.0, (double)(long long)(stats[7] - stats[4]) / 100.0);
    };
    auto pl0 = k_probe_long<false>; auto pl1 = k_probe_long<true>; auto ps0 = k_probe_short<false>; auto ps1 = k_probe_short<true>;
    for (auto f : {reinterpret_cast<void const*>(pl0), reinterpret_cast<void const*>(pl1), reinterpret_cast<void const*>(ps0), reinterpret_cast<void const*>(ps1)})
      CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int r = 0; r < 2; ++r) {
      reset(0); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(kA3, dim3(NP), dim3(1024), lds_a, sa, a);
      hipLaunchKernelGGL(ps0, dim3(NP), dim3(256), 57640, sb, a.stats);
      CK(hipDeviceSynchronize()); show("real A (RPT3), probe short without barrier");
      reset(0); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(kA3, dim3(NP), dim3(1024), lds_a, sa, a);
      hipLaunchKernelGGL(ps1, dim3(NP), dim3(256), 57640, sb, a.stats);
      CK(hipDeviceSynchronize()); show("real A (RPT3), probe short with barrier");
      reset(0); CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.produced), 0x80000000u, NP * NP)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(pl0, dim3(NP), dim3(1024), lds_a, sa, a.stats, 300000ll);
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sb, a);
      CK(hipDeviceSynchronize()); show("probe long without barrier (3 ms), real B");
      reset(0); CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.produced), 0x80000000u, NP * NP)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(pl1, dim3(NP), dim3(1024), lds_a, sa, a.stats, 300000ll);
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sb, a);
      CK(hipDeviceSynchronize()); show("probe long with barrier (3 ms), real B");
      reset(0); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(pl1, dim3(NP), dim3(1024), lds_a, sa, a.stats, 300000ll);
      hipLaunchKernelGGL(ps1, dim3(NP), dim3(256), 57640, sb, a.stats);
      CK(hipDeviceSynchronize()); show("probe long with barrier (3 ms), probe short with barrier");
    }
  }
  if (mode == 0 || mode == 1) {
    // B first on its own stream (it polls), then A; the timed region ends when both have ended
    time_it("B tailing A (two streams), RPT4", [&] {
      CK(hipEventRecord(ea, sa));
      CK(hipStreamWaitEvent(sb, ea, 0));
      hipLaunchKernelGGL(kA, dim3(NP), dim3(1024), lds_a, sa, a);   // A first: one per CU (its LDS), then B fits exactly once beside it
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sb, a);
      CK(hipEventRecord(eb, sb));
      CK(hipStreamWaitEvent(sa, eb, 0));
    });
    verify("B tailing A");
    time_it("B tailing A (two streams), RPT3", [&] {
      CK(hipEventRecord(ea, sa));
      CK(hipStreamWaitEvent(sb, ea, 0));
      hipLaunchKernelGGL(kA3, dim3(NP), dim3(1024), lds_a, sa, a);   // A first: one per CU (its LDS), then B fits exactly once beside it
      hipLaunchKernelGGL(kB, dim3(NP), dim3(BW * 64), lds_b, sb, a);
      CK(hipEventRecord(eb, sb));
      CK(hipStreamWaitEvent(sa, eb, 0));
    });
    verify("B tailing A (RPT3)");
  }
  return 0;
}
