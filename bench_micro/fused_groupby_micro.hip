// Prototype of the FUSED dense-key groupby: ONE persistent kernel, one 1024-thread workgroup per CU,
// workgroup p owns partition p's direct-address table (4096 slots x 12 bytes) in LDS for the whole call. Every workgroup
//   scatters its row tiles through LDS rings into 128-byte LINES of 12 records ([12 x f64 value | 12 x u16 slot | 8 spare bytes])
//   of the circular cell (me -> d) in a buffer small enough to stay in the Infinity Cache, and
//   consumes the lines other workgroups wrote for it (cell (s -> me), s = 0 ... 255) into its table,
// woven into one tile loop. Hand-off: produced[src][dst] / consumed[src][dst] line counters in global memory, stores and loads of
// the handed-off bytes all sc1 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").
//   ./fused_groupby_micro [rows_millions] [S tiles per sync] [variant] [groups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T gload(T const* p) { return *(GAS T const*)(p); }
template <typename T> __device__ __forceinline__ void gstore(T* p, T v) { *(GAS T*)(p) = v; }
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int NP       = 256;   // partitions = workgroups = CUs
constexpr int RPL      = 12;    // records per line
constexpr int RLINES   = 3;     // ring lines per partition
constexpr int RCAP     = RPL * RLINES;
constexpr int SLOTS    = 4096;  // table slots per partition
constexpr uint32_t FINAL = 0x80000000u;
constexpr uint32_t PAD   = SLOTS;  // the tag of a padding record: a dummy slot behind the table (no compare in the accumulate)
constexpr int AUX_SC1  = 16;    // gfx942+ cache policy bit sc1 of the buffer builtins

struct fargs {
  u64 const* keys; u64 const* vals; int64_t n;
  u64 lo, range; uint32_t mult, bmask;
  unsigned char* cells;   // [NP src][NP dst][CAPL lines][128 bytes]
  uint32_t* produced;     // [src][dst] lines written (| FINAL)
  uint32_t* consumedT;    // [src][dst] lines consumed by dst
  int32_t* status;        // [0] abort, [1] key out of range, [2] timeouts
  double* out_sum; uint32_t* out_cnt;  // [NP][SLOTS]
  int S;                  // tiles between synchronisation steps
  long long timeout_ticks;
  u64* stats;             // [0] sync steps, [1] blocked iterations, [2] pending rounds, [3] drain iterations
};

// 16-byte write-through store the compiler does not track (a store loop of variable length made it wait vmcnt(0) at every use of
// a loaded register); sync() drains with an explicit s_waitcnt before it publishes
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t div12(uint32_t x) { return __umulhi(x, 0xAAAAAAABu) >> 3; }
__device__ __forceinline__ uint32_t mod3(uint32_t x) { return x - 3u * (__umulhi(x, 0xAAAAAAABu) >> 1); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void const* p, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(bytes), 0x00020000);
}

// VAR bit 0: no consume (scatter + hand-off stores only); bit 1: no flush stores; bit 2: plain (not sc1) cell stores / loads
template <int RPT, int D, int CAPL_LOG2, int CSTEPS, int VAR>
__global__ void __launch_bounds__(1024) k_fused(fargs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr uint32_t CAPL = 1u << CAPL_LOG2;
  constexpr int B = 1024;
  unsigned char* ring = lds;                                                   // NP * 384
  double* tsum        = reinterpret_cast<double*>(lds + NP * RLINES * 128);    // (SLOTS + 2) * 8
  uint32_t* tcnt      = reinterpret_cast<uint32_t*>(tsum + SLOTS + 2);         // (SLOTS + 4) * 4
  uint32_t* tail      = tcnt + SLOTS + 4;                                      // NP
  uint32_t* limit     = tail + NP;
  uint32_t* headpub   = limit + NP;
  uint32_t* cons      = headpub + NP;
  __shared__ int s_pending[2], s_blocked[2], s_soft[2], s_flags[4];  // s_flags: [0] abort, [1] not yet flushed, [2] not done (drain)
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int const me = blockIdx.x;
  int const grp = threadIdx.x >> 3, sub = threadIdx.x & 7;  // consumer group: sources 2 grp, 2 grp + 1
  for (int i = threadIdx.x; i < SLOTS + 1; i += B) { tsum[i] = 0.0; tcnt[i] = 0; }
  for (int i = threadIdx.x; i < NP; i += B) { tail[i] = 0; limit[i] = RCAP; headpub[i] = 0; cons[i] = 0; }
  if (threadIdx.x < 2) { s_pending[threadIdx.x] = 0; s_blocked[threadIdx.x] = 0; s_soft[threadIdx.x] = 0; }
  if (threadIdx.x < 4) s_flags[threadIdx.x] = 0;
  __syncthreads();
  __amdgpu_buffer_rsrc_t const cell_rsrc = make_rsrc(a.cells, static_cast<uint32_t>(NP) * NP * CAPL * 128u);
  long long const t_start = wall_clock64();

  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int64_t const step = static_cast<int64_t>(NP) * T, end = a.n;
  u64 pk[D][RPT], pv[D][RPT];
  // the loads of a tile: buffer loads from a scalar base (rows past the end read as 0 - no per-row address arithmetic, no clamp)
  uint32_t const voff = threadIdx.x * 8u;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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
  uint32_t head = 0;                       // owner lanes (lane < 16 of each wave): lines of partition wave * 16 + lane flushed
  uint32_t c_issued[2] = {0, 0}, c_done[2] = {0, 0}, c_prod[2] = {0, 0};  // consumer group state (the same in its 8 lanes)
  bool c_final[2] = {false, false};
  u32x4 cv[CSTEPS]; uint32_t ct[CSTEPS]; int c_src[CSTEPS];   // loads in flight: values, tags, source (-1: none)
#pragma unroll
  for (int s = 0; s < CSTEPS; ++s) { c_src[s] = -1; ct[s] = 0; cv[s] = u32x4{0, 0, 0, 0}; }
  u64 n_sync = 0, n_blocked = 0, n_pendround = 0, n_drain = 0;
  // VAR & 32: wall-clock stamps (10 ns ticks) of the segments of a part, summed per workgroup by wave 0
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int i) {
    if constexpr (VAR & 32) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      long long const now = wall_clock64();
      tk[i] += now - tlast;
      tlast = now;
    }
  };
  bool bad_key = false;

  // ---- flush every complete line of this wave's partitions that its cell has room for
  auto flush = [&](int ph) {
    uint32_t nl = 0;
    int const dmine = wave * 16 + lane;
    if (lane < 16) {
      uint32_t const t = tail[dmine], limv = head * RPL + RCAP;
      uint32_t const c = static_cast<int32_t>(t - limv) < 0 ? t : limv;
      uint32_t const cl = div12(c) - head;
      uint32_t const space = (VAR & 1) ? cl : cons[dmine] + CAPL - head;
      nl = cl < space ? cl : space;
      if (nl < cl) s_blocked[ph] = 1;
    }
    if constexpr ((VAR & 2) == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int const pl = b * 8 + (lane >> 3);
        uint32_t const mn = __shfl(nl, pl), mh = __shfl(head, pl);
        uint32_t const d  = static_cast<uint32_t>(wave * 16 + pl);
        uint32_t const cellbase = (static_cast<uint32_t>(me) * NP + d) << CAPL_LOG2;
        for (uint32_t g = 0;; ++g) {
          bool const act = g < mn;
          if (__ballot(act) == 0) break;
          if (act) {
            uint32_t const L = mh + g;
            u32x4 const v = *reinterpret_cast<u32x4 const*>(ring + d * (RLINES * 128) + mod3(L) * 128 + sub * 16);
            uint32_t const off = (cellbase + (L & (CAPL - 1))) * 128u + static_cast<uint32_t>(sub) * 16u;
            if constexpr (VAR & 4) gstore(reinterpret_cast<u32x4*>(a.cells + off), v);
            else store16_sc1(a.cells + off, v);
          }
        }
      }
    }
    if (lane < 16) {
      head += nl;
      uint32_t const newlim = head * RPL + RCAP;
      limit[dmine]   = newlim;
      headpub[dmine] = head;
      if (static_cast<int32_t>(tail[dmine] - newlim) > 0) s_pending[ph] = 1;  // rows that still do not fit: another round
    }
  };

  // ---- consumer: accumulate what the previous call of this step loaded, then load the next line of one of my two sources
  auto accumulate = [&](u32x4 v, uint32_t t) {
    uint32_t const t0 = t & 0xFFFFu, t1 = t >> 16;
    double const v0 = __longlong_as_double(static_cast<long long>(static_cast<u64>(v.x) | (static_cast<u64>(v.y) << 32)));
    double const v1 = __longlong_as_double(static_cast<long long>(static_cast<u64>(v.z) | (static_cast<u64>(v.w) << 32)));
    atomicAdd(&tsum[t0], v0); atomicAdd(&tcnt[t0], 1u);
    atomicAdd(&tsum[t1], v1); atomicAdd(&tcnt[t1], 1u);
  };
  // accumulate what issue_consume() loaded (the loads were issued most of a tile ago)
  auto accumulate_pending = [&]() {
    if constexpr (VAR & 1) return;
#pragma unroll
    for (int s = 0; s < CSTEPS; ++s) {
      if (c_src[s] >= 0) {
        if constexpr ((VAR & 8) == 0) { if (sub < 6) accumulate(cv[s], ct[s]); } else { if (cv[s].x == 0x12345u && ct[s] == 77u) tcnt[0] = 1; }
        if (c_src[s] == 0) c_done[0] += 1; else c_done[1] += 1;
        c_src[s] = -1;
      }
    }
  };
  // load the next CSTEPS lines of my two sources (the loads go out whether or not a line is waiting - the compiler can then count what
  // is in flight; nothing waiting: a line of the cell that is valid memory, ignored)
  uint32_t const sb = sub < 6 ? sub : 0;
  uint32_t const cbase0 = (((static_cast<uint32_t>(2 * grp) * NP + static_cast<uint32_t>(me)) << CAPL_LOG2) * 128u) + sb * 16u;
  uint32_t const cbase1 = cbase0 + ((static_cast<uint32_t>(NP) << CAPL_LOG2) * 128u);
  uint32_t const ctag = 96u - sb * 12u;  // from a lane's value bytes to its tag dword
  auto load_line = [&](int q, u32x4& v, uint32_t& t) {
    uint32_t const ci  = q > 0 ? c_issued[1] : c_issued[0];
    uint32_t const off = (q > 0 ? cbase1 : cbase0) + ((ci & (CAPL - 1)) << 7);
    if constexpr (VAR & 16) {
      v = u32x4{off, sb, off ^ sb, 0x3ff00000u}; t = ((off >> 7) & 0xFFFu) | (((off >> 5) & 0xFFFu) << 16);
    } else if constexpr (VAR & 4) {
      v = gload(reinterpret_cast<u32x4 const*>(a.cells + off));
      t = gload(reinterpret_cast<uint32_t const*>(a.cells + off + ctag));
    } else {
      v = __builtin_amdgcn_raw_buffer_load_b128(cell_rsrc, static_cast<int>(off), 0, AUX_SC1);
      t = __builtin_amdgcn_raw_buffer_load_b32(cell_rsrc, static_cast<int>(off + ctag), 0, AUX_SC1);
    }
    if (q == 0) c_issued[0] += 1;
    if (q == 1) c_issued[1] += 1;
  };
  auto issue_consume = [&]() {
    if constexpr (VAR & 1) return;
#pragma unroll
    for (int s = 0; s < CSTEPS; ++s) {
      int const q = c_issued[0] < c_prod[0] ? 0 : (c_issued[1] < c_prod[1] ? 1 : -1);
      load_line(q, cv[s], ct[s]);
      c_src[s] = q;
    }
  };
  // rare paths (a full cell, the end of the call): everything that is waiting, loaded and accumulated at once
  auto consume_now = [&]() {
    if constexpr (VAR & 1) return;
    accumulate_pending();
    for (;;) {
      int const q = c_issued[0] < c_prod[0] ? 0 : (c_issued[1] < c_prod[1] ? 1 : -1);
      if (__syncthreads_or(q >= 0) == 0) break;
      u32x4 v; uint32_t t;
      load_line(q, v, t);
      if (q >= 0) {
        if (sub < 6) accumulate(v, t);
        if (q == 0) c_done[0] += 1; else c_done[1] += 1;
      }
    }
  };

  // ---- synchronisation step: publish what I wrote and what I consumed, refresh what the others published
  auto sync = [&](bool final) {
    ++n_sync;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its cell stores have completed (and its loads have landed)
    lds_barrier();
    if (wave == 0) {
      u32x4 h = reinterpret_cast<u32x4 const*>(headpub)[lane];
      if (final) { h.x |= FINAL; h.y |= FINAL; h.z |= FINAL; h.w |= FINAL; }
      __builtin_amdgcn_raw_buffer_store_b128(h, make_rsrc(a.produced + me * NP, NP * 4), lane * 16, 0, AUX_SC1);
    }
    if (sub < 2) __hip_atomic_store(a.consumedT + (2 * grp + sub) * NP + me, sub == 0 ? c_done[0] : c_done[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < NP) cons[threadIdx.x] = __hip_atomic_load(a.consumedT + me * NP + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t const p0 = __hip_atomic_load(a.produced + (2 * grp) * NP + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t const p1 = __hip_atomic_load(a.produced + (2 * grp + 1) * NP + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c_prod[0] = p0 & ~FINAL; c_final[0] = (p0 & FINAL) != 0;
    c_prod[1] = p1 & ~FINAL; c_final[1] = (p1 & FINAL) != 0;
    if (threadIdx.x == 0) {
      int ab = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!ab && wall_clock64() - t_start > a.timeout_ticks) {
        atomicOr(a.status, 1);
        atomicAdd(a.status + 2, 1);
        ab = 1;
      }
      if (ab) s_flags[0] = 1;
    }
    lds_barrier();
  };

  // ---- main loop over my row tiles: D parts per iteration, part j on the register slot j; a part past the end runs empty
  int ph = 0;      // parity of the place attempt (which s_pending word it raises)
  int since = 0;   // tiles since the last synchronisation step
#pragma unroll
  for (int j = 0; j < D; ++j) issue(static_cast<int64_t>(me) * T + j * step, pk[j], pv[j]);
  for (int64_t tile = static_cast<int64_t>(me) * T; tile < end; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      int64_t const t0 = tile + j * step;
      stamp(7);
      int64_t const left64 = end - t0;
      uint32_t const rows_left = left64 <= 0 ? 0u : (left64 > 0x7fffffff ? 0x7fffffffu : static_cast<uint32_t>(left64));
      bool pend[RPT];
      uint32_t d[RPT], tg[RPT], pos[RPT];
      u64 val[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pend[k] = static_cast<uint32_t>(k * B) + threadIdx.x < rows_left;  // (until placed)
        u64 idx = pk[j][k] - a.lo;
        if (idx >= a.range) { bad_key = bad_key || pend[k]; idx = 0; }  // (no atomic here: a vector-memory op that may or may not issue costs every later wait its count)
        uint32_t const x = (static_cast<uint32_t>(idx) * a.mult) & a.bmask;
        d[k]   = x >> 12;
        tg[k]  = x & 0xFFFu;
        val[k] = pv[j][k];
      }
      if constexpr (VAR & 32) { if (d[0] + tg[0] + static_cast<uint32_t>(val[0]) == 0xFFFFFFF3u) tk[7] += 1; }
      stamp(0);
      accumulate_pending();   // the lines loaded during the previous part
      stamp(1);
      issue_consume();        // (before the prefetch: a wait for these leaves the prefetch in flight)
      issue(t0 + D * step, pk[j], pv[j]);
      stamp(2);
      // reserve ring positions; rows whose position lies beyond the ring wait for the flush
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
      stamp(3);
      lds_barrier();
      stamp(4);
      if (threadIdx.x == 0) { s_pending[ph ^ 1] = 0; s_blocked[ph ^ 1] = 0; s_soft[ph ^ 1] = 0; }
      flush(ph);
      stamp(5);
      lds_barrier();
      stamp(6);
      // rows that found their ring full: its lines have left by now (the owner raised s_pending[ph] if some row still does not fit)
      if (s_soft[ph]) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) { place(k); pend[k] = false; }
        }
      }
      while (s_pending[ph]) {  // (rare: a partition took more rows than a ring holds, or its cell is full)
        ++n_pendround;
        bool const blocked = s_blocked[ph] != 0;
        ph ^= 1;
        if (blocked) {  // a cell is full: its consumer has to see my lines, and I have to see its progress
          ++n_blocked;
          since = 0;
          sync(false);
          if (s_flags[0]) break;
          consume_now();
          __builtin_amdgcn_s_sleep(8);
        }
        lds_barrier();
        if (threadIdx.x == 0) { s_pending[ph ^ 1] = 0; s_blocked[ph ^ 1] = 0; s_soft[ph ^ 1] = 0; }
        flush(ph);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) { place(k); pend[k] = false; }
        }
      }
      ph ^= 1;
    }
    since += D;
    if (since >= a.S) { since = 0; sync(false); }
    if (s_flags[0]) return;
  }
  if (bad_key) atomicOr(a.status + 1, 1);
  // ---- end of my rows: pad the partial lines, flush them as the cells allow
  if (lane < 16) {
    int const dmine = wave * 16 + lane;
    uint32_t const t = tail[dmine], q = div12(t), r = t - q * RPL;
    if (r != 0) {
      unsigned char* line = ring + dmine * (RLINES * 128) + mod3(q) * 128;
      for (uint32_t e = r; e < RPL; ++e) *reinterpret_cast<uint16_t*>(line + 96 + e * 2) = static_cast<uint16_t>(PAD);
      tail[dmine] = t + RPL - r;
    }
  }
  for (;;) {
    lds_barrier();
    if (threadIdx.x == 0) s_flags[1] = 0;
    lds_barrier();
    flush(0);
    if (lane < 16 && head * RPL != tail[wave * 16 + lane]) s_flags[1] = 1;
    lds_barrier();
    bool const all_flushed = s_flags[1] == 0;
    sync(all_flushed);
    if (s_flags[0]) return;
    consume_now();
    if (all_flushed) break;
    __builtin_amdgcn_s_sleep(8);
  }
  // ---- drain: consume until every source is final and everything it wrote for me is in my table
  for (;;) {
    ++n_drain;
    consume_now();
    lds_barrier();
    if (threadIdx.x == 0) s_flags[2] = 0;
    lds_barrier();
    bool const mine_done = (VAR & 1) || (c_final[0] && c_final[1] && c_done[0] == c_prod[0] && c_done[1] == c_prod[1]);
    if (!mine_done) s_flags[2] = 1;
    lds_barrier();
    if (s_flags[2] == 0) break;
    sync(true);
    if (s_flags[0]) return;
    __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SLOTS; i += B) {
    gstore(a.out_sum + static_cast<int64_t>(me) * SLOTS + i, tsum[i]);
    gstore(a.out_cnt + static_cast<int64_t>(me) * SLOTS + i, tcnt[i]);
  }
  if constexpr (VAR & 32) { if (threadIdx.x == 0 && a.stats) { for (int i = 0; i < 8; ++i) atomicAdd(a.stats + 8 + i, static_cast<u64>(tk[i])); } }
  if (threadIdx.x == 0 && a.stats) {
    atomicAdd(a.stats + 0, n_sync); atomicAdd(a.stats + 1, n_blocked); atomicAdd(a.stats + 2, n_pendround); atomicAdd(a.stats + 3, n_drain);
  }
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
// reference: global atomics into the same [partition][slot] layout
__global__ void k_reference(fargs a, double* sum, uint32_t* cnt)
{
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < a.n; r += stride) {
    uint32_t const x = (static_cast<uint32_t>(a.keys[r] - a.lo) * a.mult) & a.bmask;
    atomicAdd(&sum[x], __longlong_as_double(static_cast<long long>(a.vals[r])));
    atomicAdd(&cnt[x], 1u);
  }
}

template <typename K>
static void run(K kern, const char* name, fargs a, size_t cells_bytes, double const* ref_sum, uint32_t const* ref_cnt, int reps)
{
  size_t const lds = NP * RLINES * 128 + (SLOTS + 2) * 8 + (SLOTS + 4) * 4 + 4 * NP * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, tot = 0;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemsetAsync(a.produced, 0, NP * NP * 4, 0)); CK(hipMemsetAsync(a.consumedT, 0, NP * NP * 4, 0));
    CK(hipMemsetAsync(a.status, 0, 16, 0)); CK(hipMemsetAsync(a.stats, 0, 128, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(NP), dim3(1024), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) { best = ms < best ? ms : best; tot += ms; }
  }
  int32_t st[4]; u64 stats[16];
  CK(hipMemcpy(st, a.status, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(stats, a.stats, 128, hipMemcpyDeviceToHost));
  printf("%-44s %8.3f ms avg %8.3f best  -> %5.2f ms per 1B rows | abort %d range %d | syncs/wg %.0f blocked %llu pend %llu drain/wg %.0f\n", name, tot / reps, best,
         best * 1e9 / a.n, st[0], st[1], stats[0] / 256.0, stats[1], stats[2], stats[3] / 256.0);
  if (stats[8] + stats[9] + stats[10]) {
    const char* nm[8] = {"decode(wait keys)", "accumulate(wait lines)", "issue loads", "reserve+place", "barrier1", "flush", "barrier2", "soft-place+loop"};
    u64 tot8 = 0; for (int i = 0; i < 8; ++i) tot8 += stats[8 + i];
    printf("   stamps (us per workgroup; whole kernel %.0f us):", best * 1e3);
    for (int i = 0; i < 8; ++i) printf(" %s %.0f |", nm[i], stats[8 + i] / 256.0 / 100.0);
    printf(" sum %.0f\n", tot8 / 256.0 / 100.0);
  }
  if (ref_sum && st[0] == 0) {
    std::vector<double> hs(static_cast<size_t>(NP) * SLOTS); std::vector<uint32_t> hc(static_cast<size_t>(NP) * SLOTS);
    CK(hipMemcpy(hs.data(), a.out_sum, hs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), a.out_cnt, hc.size() * 4, hipMemcpyDeviceToHost));
    size_t badc = 0, bads = 0; double maxrel = 0; u64 total = 0;
    for (size_t i = 0; i < hs.size(); ++i) {
      total += hc[i];
      if (hc[i] != ref_cnt[i]) ++badc;
      double const e = fabs(hs[i] - ref_sum[i]), tol = 1e-9 * (fabs(ref_sum[i]) + 1);
      if (e > tol) ++bads;
      if (fabs(ref_sum[i]) > 0) maxrel = fmax(maxrel, e / fabs(ref_sum[i]));
    }
    printf("   verify: rows counted %llu of %lld, slots with wrong count %zu, wrong sum %zu, max rel err %.2e\n", total, (long long)a.n, badc, bads, maxrel);
  }
}

int main(int argc, char** argv)
{
  int64_t const n  = static_cast<int64_t>((argc > 1 ? atof(argv[1]) : 400) * 1000000.0);
  int const S      = argc > 2 ? atoi(argv[2]) : 4;
  int const variant = argc > 3 ? atoi(argv[3]) : 0;
  u64 const groups = argc > 4 ? atoll(argv[4]) : 1000000;
  int const reps   = 3;
  u64 *keys, *vals;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&vals, n * 8));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, keys, vals, n, groups);
  fargs a{};
  a.keys = keys; a.vals = vals; a.n = n; a.lo = 0; a.range = 1u << 20; a.mult = 0x9E3779B1u; a.bmask = (1u << 20) - 1;
  size_t const max_cells = static_cast<size_t>(NP) * NP * 32 * 128;
  CK(hipMalloc(&a.cells, max_cells)); CK(hipMalloc(&a.produced, NP * NP * 4)); CK(hipMalloc(&a.consumedT, NP * NP * 4));
  CK(hipMalloc(&a.status, 16)); CK(hipMalloc(&a.stats, 128));
  CK(hipMalloc(&a.out_sum, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMalloc(&a.out_cnt, static_cast<size_t>(NP) * SLOTS * 4));
  a.S = S; a.timeout_ticks = 100000000ll * 2;  // 2 s at 100 MHz
  double* rs; uint32_t* rc;
  CK(hipMalloc(&rs, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMalloc(&rc, static_cast<size_t>(NP) * SLOTS * 4));
  CK(hipMemset(rs, 0, static_cast<size_t>(NP) * SLOTS * 8)); CK(hipMemset(rc, 0, static_cast<size_t>(NP) * SLOTS * 4));
  hipLaunchKernelGGL(k_reference, dim3(4096), dim3(256), 0, 0, a, rs, rc);
  std::vector<double> hrs(static_cast<size_t>(NP) * SLOTS); std::vector<uint32_t> hrc(static_cast<size_t>(NP) * SLOTS);
  CK(hipMemcpy(hrs.data(), rs, hrs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hrc.data(), rc, hrc.size() * 4, hipMemcpyDeviceToHost));
  int dev; CK(hipGetDevice(&dev)); hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("rows %lld groups %llu S %d CUs %d\n", (long long)n, groups, S, prop.multiProcessorCount);
  if (prop.multiProcessorCount < NP) { printf("needs %d CUs\n", NP); return 1; }
  double const* R = hrs.data(); uint32_t const* C = hrc.data();
  if (variant == 0 || variant == 1) {
    run(k_fused<4, 2, 4, 3, 0>, "fused RPT4 D2 CAPL16 CS3", a, 0, R, C, reps);
    run(k_fused<4, 2, 3, 3, 0>, "fused RPT4 D2 CAPL8 CS3", a, 0, R, C, reps);
    run(k_fused<4, 2, 5, 3, 0>, "fused RPT4 D2 CAPL32 CS3", a, 0, R, C, reps);
    run(k_fused<3, 2, 4, 3, 0>, "fused RPT3 D2 CAPL16 CS3", a, 0, R, C, reps);
    run(k_fused<2, 2, 4, 2, 0>, "fused RPT2 D2 CAPL16 CS2", a, 0, R, C, reps);
    run(k_fused<4, 2, 4, 4, 0>, "fused RPT4 D2 CAPL16 CS4", a, 0, R, C, reps);
  }
  if (variant == 4) {
    run(k_fused<2, 2, 5, 2, 32>, "fused RPT2 CAPL32 CS2 stamps", a, 0, R, C, 1);
    run(k_fused<2, 2, 5, 2, 33>, "  no consume, stamps", a, 0, nullptr, nullptr, 1);
    run(k_fused<2, 2, 5, 2, 35>, "  no consume, no stores, stamps", a, 0, nullptr, nullptr, 1);
    run(k_fused<4, 2, 5, 3, 35>, "  RPT4 no consume, no stores, stamps", a, 0, nullptr, nullptr, 1);
  }
  if (variant == 3) {
    run(k_fused<2, 2, 5, 2, 0>, "fused RPT2 CAPL32 CS2", a, 0, R, C, reps);
    run(k_fused<2, 2, 5, 2, 8>, "  loads, no LDS accumulate", a, 0, nullptr, nullptr, reps);
    run(k_fused<2, 2, 5, 2, 16>, "  no loads, LDS accumulate of dummies", a, 0, nullptr, nullptr, reps);
    run(k_fused<2, 2, 5, 2, 24>, "  no loads, no accumulate (bookkeeping only)", a, 0, nullptr, nullptr, reps);
    run(k_fused<2, 2, 5, 2, 1>, "  no consume", a, 0, nullptr, nullptr, reps);
    run(k_fused<2, 2, 5, 2, 3>, "  no consume, no stores", a, 0, nullptr, nullptr, reps);
  }
  if (variant == 0 || variant == 2) {
    run(k_fused<4, 2, 4, 3, 1>, "no consume (scatter + stores) RPT4 CAPL16", a, 0, nullptr, nullptr, reps);
    run(k_fused<4, 2, 4, 3, 3>, "no consume, no stores RPT4", a, 0, nullptr, nullptr, reps);
    run(k_fused<4, 2, 4, 3, 4>, "fused, plain stores/loads (INVALID hand-off) RPT4", a, 0, nullptr, nullptr, reps);
  }
  return 0;
}
