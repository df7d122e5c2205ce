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

// Large blocks are COLOURED: block number n of the process starts at an address whose bits 12..15 are bitrev4(n mod 16), i.e.
// in its own 4 KiB slot of a 64 KiB frame, consecutive blocks as far apart as the frame allows.  A fold kernel streams through
// all ranks' send and receive buffers at the same index at the same time; buffers whose addresses are equal modulo a large
// power of two -- which is where an allocator that packs 256 MiB blocks into 1 GiB arenas puts them -- meet in the same
// HBM banks, each alternating between 16 rows.  Measured (scripts/placement_probe.py, placement_patterns.py: reduce_n_multi
// over 8 + 8 buffers of 256 MiB): stride 256 MiB exactly 836-853 us; + k x 4 KiB 754-768; + k x 8 / 16 / 32 / 64 KiB or 1 MiB
// 815-850 (bit 12 the same everywhere); the bit-reversed slots 749-752 whichever 8 of the 16 are the sources.  The skipped
// prefix (< 128 KiB) stays an ordinary free block.  (heap_colour_seed: ranks in different processes would start the count at
// 2 x rank -- a rank's send and receive buffer slots 8 apart, the ranks' pairs the rest -- if colouring is forced on there.)
constexpr size_t kColourMin = (size_t)1 << 20, kColourFrame = (size_t)64 << 10, kColourSlot = 4096;
uint64_t g_colour_next = 0;
bool g_heap_shared = false;  // several ranks of a job are threads of this process

int colour_slot(uint64_t n) {
  const unsigned k = (unsigned)(n & 15);
  return (int)(((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3));
}

// slot < 0: the block starts where the free range starts
void* take(Arena* a, size_t need, int slot) {
  for (auto it = a->free_blocks.begin(); it != a->free_blocks.end(); ++it) {
    const size_t off = it->first, len = it->second;
    size_t start = off;
    if (slot >= 0) {  // by absolute address (an arena's base is 2 MiB aligned in practice, but nothing promises it)
      const uintptr_t abs0 = ((uintptr_t)a->base + off + kColourFrame - 1) / kColourFrame * kColourFrame + (uintptr_t)slot * kColourSlot;
      start = (size_t)(abs0 - (uintptr_t)a->base);
    }
    if (start + need > off + len) continue;
    a->free_blocks.erase(it);
    if (start > off) a->free_blocks[off] = start - off;
    if (off + len > start + need) a->free_blocks[start + need] = off + len - start - need;
    a->used_blocks[start] = need;
    return a->base + start;
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
  g_heap_shared = false;
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

// the first communicator of the process says which rank lives here: ranks in different processes take different colours
void heap_colour_seed(int rank, bool ranks_share_this_process) {
  std::lock_guard<std::mutex> g(g_heap_mu);
  if (g_live_comms == 0) g_colour_next = 2 * (uint64_t)std::max(0, rank);
  if (ranks_share_this_process) g_heap_shared = true;  // known before any of them can allocate (xmpi_init's last barrier)
}

void* heap_alloc(int device, size_t bytes) {
  const size_t need = (std::max<size_t>(bytes, 1) + kGranule - 1) / kGranule * kGranule;
  std::lock_guard<std::mutex> g(g_heap_mu);
  // Where it is on: a heap that serves SEVERAL ranks (threads of one process on one GPU: all buffers of a fold are blocks of
  // this heap, 256 MiB apart -- N = 1 line: kernel 838 / 850 / 721 us without, 703 / 703 / 755 with, same box, interleaved).  With
  // one process per rank the same A/B was inside the noise or against it (8 processes on one GPU: +3 ... +9 %, r03 session 12),
  // so there the blocks stay where they were measured.  XMPI_HEAP_COLOUR=1 / 0 forces it.
  static const int forced = getenv("XMPI_HEAP_COLOUR") ? atoi(getenv("XMPI_HEAP_COLOUR")) : -1;
  const bool coloured = forced >= 0 ? forced != 0 : g_heap_shared;
  const int slot = (coloured && need >= kColourMin) ? colour_slot(g_colour_next++) : -1;
  const size_t colour = slot < 0 ? 0 : kColourFrame + (size_t)slot * kColourSlot;  // room the placement may take
  for (Arena* a : g_arenas)
    if (a->device == device)
      if (void* p = take(a, need, slot)) return p;
  const size_t lo = env_size("XMPI_ARENA_MIN_BYTES", 64u << 20), hi = env_size("XMPI_ARENA_MAX_BYTES", 1u << 30);
  if (g_next_arena < lo) g_next_arena = lo;
  size_t want = std::max(need + colour, std::min(g_next_arena, hi));
  want = (want + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  void* base = nullptr;
  if (hipMalloc(&base, want) != hipSuccess) {
    (void)hipGetLastError();
    want = (need + colour + kArenaAlign - 1) / kArenaAlign * kArenaAlign;  // memory is tight: just what was asked for
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
  return take(a, need, slot);
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
      char* p = (char*)take(&a, need, need >= kColourMin ? colour_slot(rnd()) : -1);
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
