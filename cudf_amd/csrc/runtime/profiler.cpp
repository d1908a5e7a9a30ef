// SPDX-License-Identifier: Apache-2.0
#include "../common/profiler.hpp"

#include <dlfcn.h>

#include <atomic>
#include <map>
#include <mutex>

namespace cudf::detail::prof {
namespace {
struct pending {
  char const* name;
  hipEvent_t start, stop;
};
std::mutex g_mu;
bool g_on = false;
std::vector<pending> g_pending;
std::map<std::string, std::pair<int64_t, double>> g_totals;

void drain_locked()
{
  for (auto& p : g_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      auto& t = g_totals[p.name];
      t.first += 1;
      t.second += ms;
    }
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  g_pending.clear();
}
}  // namespace

void enable(bool on)
{
  std::lock_guard<std::mutex> g{g_mu};
  g_on = on;
}
bool enabled() { return g_on; }
void reset()
{
  std::lock_guard<std::mutex> g{g_mu};
  drain_locked();
  g_totals.clear();
}
std::vector<kernel_stat> collect()
{
  std::lock_guard<std::mutex> g{g_mu};
  drain_locked();
  std::vector<kernel_stat> out;
  for (auto const& [k, v] : g_totals) out.push_back({k, v.first, v.second});
  return out;
}

scope::scope(char const* name, hipStream_t stream) : _stream{stream}
{
  if (!g_on) return;
  pending p{name, nullptr, nullptr};
  if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) return;
  (void)hipEventRecord(p.start, stream);
  std::lock_guard<std::mutex> g{g_mu};
  g_pending.push_back(p);
  _slot = static_cast<int>(g_pending.size()) - 1;
}
scope::~scope()
{
  if (_slot < 0) return;
  std::lock_guard<std::mutex> g{g_mu};
  if (_slot < static_cast<int>(g_pending.size())) (void)hipEventRecord(g_pending[_slot].stop, _stream);
}

namespace {
using roctx_push_t = int (*)(char const*);
using roctx_pop_t  = int (*)();
std::atomic<int> g_roctx_state{0};  // 0: not resolved yet, 1: available, 2: absent
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop   = nullptr;
std::once_flag g_roctx_once;

void resolve_roctx()
{
  std::call_once(g_roctx_once, [] {
    for (char const* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
      if (h == nullptr) continue;
      auto push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
      auto pop  = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
      if (push != nullptr && pop != nullptr) {
        g_roctx_push = push;
        g_roctx_pop  = pop;
        g_roctx_state.store(1, std::memory_order_release);
        return;
      }
    }
    g_roctx_state.store(2, std::memory_order_release);
  });
}
}  // namespace

func_range::func_range(char const* name)
{
  int st = g_roctx_state.load(std::memory_order_acquire);
  if (st == 0) {
    resolve_roctx();
    st = g_roctx_state.load(std::memory_order_acquire);
  }
  if (st == 1) {
    (void)g_roctx_push(name);
    _pushed = true;
  }
}
func_range::~func_range()
{
  if (_pushed) (void)g_roctx_pop();
}
}  // namespace cudf::detail::prof
