// heap.cpp -- xmpi_malloc's allocator: user buffers are carved out of a few large HBM arenas.
//
// Why not one hipMalloc per buffer: the zero-copy collectives (zcopy.cpp) need every buffer mapped
// into every peer (hipIpc).  Mapping costs hundreds of microseconds, handles of freed allocations are
// re-issued by the runtime, and attaching while another process detaches failed sporadically on this
// stack ("invalid device pointer", ~0.2 % of the opens with 8 processes allocating and freeing
// per call).  With arenas the peers map a handful of long-lived allocations once; a buffer is an
// offset into one of them, and allocating or freeing it involves no runtime call at all.  An arena
// stays allocated (and mapped) until the last communicator of the process is finalised -- sized for
// 288 GB of HBM, not for a 16 GB card.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "comm.h"

namespace xmpi {
namespace {

constexpr size_t kGranule = 256;        // every block starts on a 256-byte boundary
constexpr size_t kArenaAlign = 2u << 20;

struct Arena {
  char* base = nullptr;
  size_t bytes = 0;
  int device = 0;
  std::map<size_t, size_t> free_blocks;  // offset -> length
  std::map<size_t, size_t> used_blocks;  // offset -> length
};

std::mutex g_heap_mu;
std::vector<Arena*> g_arenas;
size_t g_next_arena = 0;  // size of the next general-purpose arena (doubles up to the cap)
int g_live_comms = 0;

size_t env_size(const char* name, size_t dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const long long x = strtoll(v, &end, 0);
  return (end && end != v && x > 0) ? (size_t)x : dflt;
}

void* take(Arena* a, size_t need) {
  for (auto it = a->free_blocks.begin(); it != a->free_blocks.end(); ++it) {
    if (it->second < need) continue;
    const size_t off = it->first, len = it->second;
    a->free_blocks.erase(it);
    if (len > need) a->free_blocks[off + need] = len - need;
    a->used_blocks[off] = need;
    return a->base + off;
  }
  return nullptr;
}

}  // namespace

void heap_comm_created() {
  std::lock_guard<std::mutex> g(g_heap_mu);
  g_live_comms++;
}

// the last communicator of the process is gone: give empty arenas back
void heap_comm_destroyed(xmpi_comm* c) {
  std::lock_guard<std::mutex> g(g_heap_mu);
  if (--g_live_comms > 0) return;
  for (size_t i = 0; i < g_arenas.size();) {
    Arena* a = g_arenas[i];
    if (a->used_blocks.empty()) {
      registry_remove(c, a->base);
      (void)hipSetDevice(a->device);
      (void)hipFree(a->base);
      delete a;
      g_arenas.erase(g_arenas.begin() + (long)i);
    } else {
      i++;
    }
  }
  g_next_arena = 0;
}

void* heap_alloc(int device, size_t bytes) {
  const size_t need = (std::max<size_t>(bytes, 1) + kGranule - 1) / kGranule * kGranule;
  std::lock_guard<std::mutex> g(g_heap_mu);
  for (Arena* a : g_arenas)
    if (a->device == device)
      if (void* p = take(a, need)) return p;
  const size_t lo = env_size("XMPI_ARENA_MIN_BYTES", 64u << 20), hi = env_size("XMPI_ARENA_MAX_BYTES", 1u << 30);
  if (g_next_arena < lo) g_next_arena = lo;
  size_t want = std::max(need, std::min(g_next_arena, hi));
  want = (want + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  void* base = nullptr;
  if (hipMalloc(&base, want) != hipSuccess) {
    (void)hipGetLastError();
    want = (need + kArenaAlign - 1) / kArenaAlign * kArenaAlign;  // memory is tight: just what was asked for
    if (hipMalloc(&base, want) != hipSuccess) return nullptr;
  }
  g_next_arena = std::min(hi, g_next_arena * 2);
  Arena* a = new Arena;
  a->base = (char*)base;
  a->bytes = want;
  a->device = device;
  a->free_blocks[0] = want;
  g_arenas.push_back(a);
  registry_add(base, want, device);  // the arena is what peers map
  return take(a, need);
}

namespace {
// give a live block back to its arena, merging it with free neighbours
bool release_block(Arena* a, char* p) {
  const size_t off = (size_t)(p - a->base);
  auto u = a->used_blocks.find(off);
  if (u == a->used_blocks.end()) return false;  // not the start of a live block
  size_t lo = off, len = u->second;
  a->used_blocks.erase(u);
  auto next = a->free_blocks.lower_bound(lo);
  if (next != a->free_blocks.end() && next->first == lo + len) {  // merge with the block after
    len += next->second;
    next = a->free_blocks.erase(next);
  }
  if (next != a->free_blocks.begin()) {  // ... and with the block before
    auto prev = std::prev(next);
    if (prev->first + prev->second == lo) {
      lo = prev->first;
      len += prev->second;
      a->free_blocks.erase(prev);
    }
  }
  a->free_blocks[lo] = len;
  return true;
}
}  // namespace

bool heap_free(void* p) {
  std::lock_guard<std::mutex> g(g_heap_mu);
  for (Arena* a : g_arenas)
    if ((char*)p >= a->base && (char*)p < a->base + a->bytes) return release_block(a, (char*)p);
  return false;
}

bool heap_owns(const void* p) {
  std::lock_guard<std::mutex> g(g_heap_mu);
  for (Arena* a : g_arenas)
    if ((const char*)p >= a->base && (const char*)p < a->base + a->bytes) return true;
  return false;
}

// Host-only exercise of the block bookkeeping (no HIP call): random allocate / free traffic on one
// synthetic arena; blocks must never overlap, stay 256-byte aligned, and everything must coalesce back
// into a single free block.  Returns 0, or the number of the check that failed.
int heap_selftest(uint64_t seed, int rounds) {
  Arena a;
  a.base = (char*)(uintptr_t)0x10000000;  // never dereferenced
  a.bytes = 64u << 20;
  a.free_blocks[0] = a.bytes;
  std::vector<std::pair<char*, size_t>> live;
  auto rnd = [&seed]() {
    seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(seed >> 33);
  };
  auto release = [&a](char* p) -> bool { return release_block(&a, p); };
  for (int r = 0; r < rounds; r++) {
    if (live.empty() || rnd() % 3 != 0) {
      const size_t want = (rnd() % 7 == 0) ? (size_t)(rnd() % (4u << 20)) + 1 : (size_t)(rnd() % 5000) + 1;
      const size_t need = (want + kGranule - 1) / kGranule * kGranule;
      char* p = (char*)take(&a, need);
      if (!p) continue;  // full: fine
      if (((uintptr_t)p & (kGranule - 1)) != 0) return 1;
      for (auto& b : live)
        if (p < b.first + b.second && b.first < p + need) return 2;  // overlap
      live.push_back({p, need});
    } else {
      const size_t i = rnd() % live.size();
      if (!release(live[i].first)) return 3;
      if (release(live[i].first)) return 4;  // double free must be refused
      live.erase(live.begin() + (long)i);
    }
    size_t used = 0, freeb = 0;
    for (auto& kv : a.used_blocks) used += kv.second;
    for (auto& kv : a.free_blocks) freeb += kv.second;
    if (used + freeb != a.bytes) return 5;  // bytes leaked or counted twice
    size_t prev_end = (size_t)-1;
    for (auto& kv : a.free_blocks) {  // neighbours must have been merged
      if (kv.first == prev_end) return 6;
      prev_end = kv.first + kv.second;
    }
  }
  for (auto& b : live)
    if (!release(b.first)) return 7;
  if (a.free_blocks.size() != 1 || a.free_blocks.begin()->first != 0 || a.free_blocks.begin()->second != a.bytes) return 8;
  return 0;
}

// diagnostics: arenas / bytes reserved / bytes in use on `device`
void heap_stats(int device, size_t* arenas, size_t* reserved, size_t* in_use) {
  std::lock_guard<std::mutex> g(g_heap_mu);
  *arenas = *reserved = *in_use = 0;
  for (Arena* a : g_arenas) {
    if (a->device != device) continue;
    (*arenas)++;
    *reserved += a->bytes;
    for (auto& kv : a->used_blocks) *in_use += kv.second;
  }
}

}  // namespace xmpi
