// kmx_api.hip -- C ABI of libkmx (include/kmx.h): context, device memory pool, batch merge driver.
// No CPU fallback anywhere: every entry point needs a live HIP device.
#include "kmx_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <sys/mman.h>
#include <unordered_map>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace kmx;

static thread_local std::string g_create_err;

// ---- memory pools ----------------------------------------------------------------------------------
void* kmx_ctx::dalloc(size_t bytes)
{
  if (bytes == 0) bytes = 256;
  // (round 6) large blocks come in size classes an eighth of a power of two apart: the arenas of a job's batches differ by a few per cent
  // from batch to batch and from pass to pass (their size follows the running estimates), and a request a little above every free
  // block was a hipMalloc of 600 MB -- 18 ms with the GPU idle, in the middle of the whole-job figure (profiles/r06_whole_job_timeline.txt)
  if (bytes >= (8u << 20)) { size_t p2 = (size_t)1 << 23; while (p2 * 2 <= bytes) p2 *= 2; const size_t q = p2 >> 3; bytes = (bytes + q - 1) / q * q; }
  int best = -1;
  for (size_t i = 0; i < pool.size(); i++)
    if (!pool[i].used && pool[i].bytes >= bytes && pool[i].bytes <= 2 * bytes + (1u << 20) &&
        (best < 0 || pool[i].bytes < pool[best].bytes)) best = (int)i;
  if (best >= 0) { pool[best].used = true; return pool[best].p; }
  // a large block that has to be made is made as large as the largest one asked for so far (when that is within 1.5 x): after a few
  // batches every arena block of a job fits every arena request, and the pool stops growing (it had reached 104 GB in 431 blocks over
  // the whole job of configs[2], a quarter of it free, and still missed)
  if (bytes >= (64u << 20)) { big_max = std::max(big_max, bytes); if (big_max <= bytes + bytes / 2) bytes = big_max; }
  void* p = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  struct Tr { size_t b; std::chrono::steady_clock::time_point t; kmx_ctx* c; ~Tr() { static const bool on = getenv("KMX_TRACE") != nullptr || getenv("KMX_TRACE_ALLOC") != nullptr; if (on) { size_t fr = 0, tot = 0; for (auto& x : c->pool) { tot += x.bytes; if (!x.used) fr += x.bytes; }
    fprintf(stderr, "[kmx alloc] device pool +%zu MB: %.2f ms (pool %zu MB in %zu blocks, %zu MB of it free)\n", b >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(), tot >> 20, c->pool.size(), fr >> 20); } } } tr{bytes, t0, this};
  if (hipMalloc(&p, bytes) != hipSuccess) {
    // drop cached blocks and retry once
    for (auto& b : pool) if (!b.used && b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    pool.erase(std::remove_if(pool.begin(), pool.end(), [](const kmx_pool_block& b) { return b.p == nullptr; }), pool.end());
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  }
  pool.push_back({p, bytes, true});
  return p;
}
void kmx_ctx::dfree(void* p)
{
  if (!p) return;
  for (auto& b : pool) if (b.p == p) { b.used = false; return; }
}
void* kmx_ctx::halloc(size_t bytes)
{
  if (bytes == 0) bytes = 256;
  int best = -1;
  for (size_t i = 0; i < hpool.size(); i++)
    if (!hpool[i].used && hpool[i].bytes >= bytes && hpool[i].bytes <= 2 * bytes + (1u << 20) &&
        (best < 0 || hpool[i].bytes < hpool[best].bytes)) best = (int)i;
  if (best >= 0) { hpool[best].used = true; return hpool[best].p; }
  void* p = nullptr;
  p = kmx_pinned_alloc(bytes);
  if (!p) return nullptr;
  hpool.push_back({p, bytes, true});
  return p;
}
void kmx_ctx::hfree(void* p)
{
  if (!p) return;
  for (auto& b : hpool) if (b.p == p) { b.used = false; return; }
}

// ---- context -----------------------------------------------------------------------------------------
extern "C" int kmx_version(void) { return KMX_VERSION; }
extern "C" int kmx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

extern "C" int kmx_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return KMX_E_NODEVICE;
  int cur = -1; (void)hipGetDevice(&cur);
  if (hipSetDevice(device) != hipSuccess) return KMX_E_NODEVICE;
  size_t fr = 0, tot = 0;
  const hipError_t e = hipMemGetInfo(&fr, &tot);
  if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
  if (e != hipSuccess) return KMX_E_HIP;
  if (free_bytes) *free_bytes = fr;
  if (total_bytes) *total_bytes = tot;
  return KMX_OK;
}

extern "C" uint64_t kmx_device_warm(int device, uint64_t bytes, const volatile int* stop)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || bytes == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) return 0;
  const size_t piece = (size_t)1 << 30;
  std::vector<void*> held;
  uint64_t got = 0;
  while (got < bytes) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < tot / 4 + piece) break;      // (a quarter of the device stays free for whoever is working)
    if (stop && *stop) break;
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipMalloc(&p, piece) != hipSuccess) { (void)hipGetLastError(); break; }
    held.push_back(p); got += piece;
    if (held.size() == 1 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 5.0) break;      // (this memory has been handed out before)
  }
  for (void* p : held) (void)hipFree(p);
  return got;
}

extern "C" int kmx_create(int device, kmx_ctx** out)
{
  if (!out) { g_create_err = "kmx_create: out is NULL"; return KMX_E_INVAL; }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_err = std::string("kmx_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                   "); libkmx has no CPU fallback";
    return KMX_E_NODEVICE;
  }
  if (device < 0 || device >= n) { g_create_err = "kmx_create: device index out of range"; return KMX_E_INVAL; }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return KMX_E_NODEVICE; }
  kmx_ctx* c = new kmx_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete c; return KMX_E_NODEVICE; }
  c->n_cu = prop.multiProcessorCount;
  { const char* fo = getenv("KMX_FILE_ORDER"); c->file_order = !(fo && fo[0] == '0'); }
  { const char* v = getenv("KMX_COLS_MIN_LISTS"); if (v && atoi(v) > 0) { c->cols_min_lists = (unsigned)atoi(v); c->cols_min_env = true; }
    v = getenv("KMX_COLS_MIN_LISTS_ORD"); if (v && atoi(v) > 0) { c->cols_min_lists_ord = (unsigned)atoi(v); c->cols_min_env = true; } }
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete c; return KMX_E_NODEVICE; }
  if ((e = hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking)) != hipSuccess) { g_create_err = hipGetErrorString(e); (void)hipStreamDestroy(c->stream); delete c; return KMX_E_NODEVICE; }
  if ((e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking)) != hipSuccess) { g_create_err = hipGetErrorString(e); (void)hipStreamDestroy(c->stream); (void)hipStreamDestroy(c->aux); delete c; return KMX_E_NODEVICE; }
  if ((e = hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking)) != hipSuccess) { g_create_err = hipGetErrorString(e); (void)hipStreamDestroy(c->stream); (void)hipStreamDestroy(c->aux); (void)hipStreamDestroy(c->copy); delete c; return KMX_E_NODEVICE; }
  *out = c;
  return KMX_OK;
}

extern "C" void kmx_destroy(kmx_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->aux);
  (void)hipStreamSynchronize(ctx->copy);
  (void)hipStreamSynchronize(ctx->up);
  for (auto& a : ctx->ahead) if (a.ev) (void)hipEventDestroy(a.ev);
  for (auto& e : ctx->copy_ev) if (e) (void)hipEventDestroy(e);
  for (auto& b : ctx->pool) if (b.p) (void)hipFree(b.p);
  if (ctx->d_hist) (void)hipFree(ctx->d_hist);
  if (ctx->d_rep) (void)hipFree(ctx->d_rep);
  if (ctx->d_stat) (void)hipFree(ctx->d_stat);
  for (auto& b : ctx->hpool) if (b.p) kmx_pinned_free(b.p);
  (void)hipStreamDestroy(ctx->stream);
  kmx_count_chain_forget(ctx);
  if (ctx->ev_split) (void)hipEventDestroy(ctx->ev_split);
  (void)hipStreamDestroy(ctx->aux);
  (void)hipStreamDestroy(ctx->copy);
  (void)hipStreamDestroy(ctx->up);
  delete ctx;
}

extern "C" const char* kmx_last_error(const kmx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" void* kmx_stream(kmx_ctx* ctx) { if (ctx) ctx->stream_shared = true; return ctx ? (void*)ctx->stream : nullptr; }
extern "C" void kmx_free(void* p) { free(p); }
// Page-locked host memory.  hipHostMalloc pins 4 KB pages one by one -- 5.6 ms for 32 MB, all of it under the runtime's lock, so
// that a thread pinning the pipeline's buffers stalls every other thread's launches and copies (the output ring, the read
// buffers).  An anonymous mapping backed by transparent huge pages, touched and then registered, is page-locked in 1.5 ms of which
// 0.1 ms are the runtime's (hipHostRegister over 16 pages of 2 MB instead of 8192 of 4 KB); device copies run at the same 55 GB/s
// (scripts/dev/pin_speed.hip).  Blocks of at least 2 MB take that road; smaller ones, and any failure on it, hipHostMalloc.
namespace {
std::mutex g_pin_mutex;
std::unordered_map<void*, size_t> g_pin_mapped;      // blocks that came from mmap + hipHostRegister: their mapped size
}
void* kmx_pinned_alloc(size_t bytes)
{
  if (bytes == 0) bytes = 256;
  static const bool thp_off = getenv("KMX_PINNED_HIPHOSTMALLOC") != nullptr;
  if (bytes >= (2u << 20) && !thp_off) {
    const size_t n = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    void* q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q != MAP_FAILED) {
      (void)madvise(q, n, MADV_HUGEPAGE);
      static const bool trace = getenv("KMX_TRACE") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t o = 0; o < n; o += 4096) static_cast<volatile char*>(q)[o] = 0;      // fault the pages in here, not under the runtime's lock
      const auto t1 = std::chrono::steady_clock::now();
      const hipError_t re = hipHostRegister(q, n, hipHostRegisterPortable);
      if (trace) fprintf(stderr, "[kmx alloc] pinned %zu MB: touch %.2f ms, register %.2f ms\n", n >> 20, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
      if (re == hipSuccess) {      // (portable: a process that drives several GPUs copies to and from it on all of them, as with hipHostMalloc)
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        g_pin_mapped[q] = n;
        return q;
      }
      (void)hipGetLastError();
      munmap(q, n);
    }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void kmx_pinned_free(void* p)
{
  if (!p) return;
  size_t n = 0;
  { std::lock_guard<std::mutex> lk(g_pin_mutex); auto it = g_pin_mapped.find(p); if (it != g_pin_mapped.end()) { n = it->second; g_pin_mapped.erase(it); } }
  if (n) { (void)hipHostUnregister(p); munmap(p, n); }
  else (void)hipHostFree(p);
}
// [p, p + n) lies in a block kmx_alloc_pinned made by mapping + registering (2 MB and up): a copy from it is a DMA by itself
bool kmx_is_pinned(const void* p, size_t n)
{
  std::lock_guard<std::mutex> lk(g_pin_mutex);
  for (auto& b : g_pin_mapped) if ((const char*)p >= (const char*)b.first && (const char*)p + n <= (const char*)b.first + b.second) return true;
  return false;
}
extern "C" void* kmx_alloc_pinned(size_t bytes) { return kmx_pinned_alloc(bytes); }
extern "C" void kmx_free_pinned(void* p) { kmx_pinned_free(p); }
// ---- kmx_store: count lists resident in HBM between the count and the merge stage ---------------------------------
void kmx_store::start_ahead()
{
  if (ahead_on) return;
  ahead_on = true;
  const char* e = getenv("KMX_STORE_AHEAD");
  if (e && e[0] == '0') return;
  ahead = std::thread([this]() {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [this]() { return stop || spare.size() < 2; });
      if (stop) return;
      size_t held = 0; for (auto& c : chunks) held += c.cap; for (auto& c : spare) held += c.cap;
      if (limit && held + chunk_bytes > limit + chunk_bytes) { cv.wait_for(lk, std::chrono::milliseconds(50)); continue; }      // (at the limit: nothing more ahead)
      lk.unlock();
      void* p = nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      const hipError_t er = hipMalloc(&p, chunk_bytes);
      { static const bool trace = getenv("KMX_TRACE") != nullptr; if (trace) fprintf(stderr, "[kmx alloc] store chunk ahead %zu MB: %.2f ms\n", chunk_bytes >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
      lk.lock();
      if (er != hipSuccess) { (void)hipGetLastError(); cv.wait_for(lk, std::chrono::milliseconds(200)); continue; }
      spare.push_back({(u8*)p, chunk_bytes, 0});
    }
  });
}
bool kmx_store::take_spare(size_t bytes, Chunk& out)
{
  if (bytes > chunk_bytes || spare.empty()) return false;
  out = spare.back(); spare.pop_back();
  cv.notify_one();
  return true;
}
void* kmx_store::alloc(size_t bytes)
{
  bytes = (bytes + 255) / 256 * 256;
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> lk(mu);
  start_ahead();
  if (limit && used + bytes > limit) return nullptr;
  for (size_t i = 0; i < chunks.size(); i++) { auto& c = chunks[i]; if (!chunk_reserved(i) && c.cap - c.fill >= bytes) { void* p = c.p + c.fill; c.fill += bytes; used += bytes; return p; } }
  { Chunk sp; if (take_spare(bytes, sp)) { sp.fill = bytes; chunks.push_back(sp); used += bytes; return sp.p; } }
  int cur = -1; (void)hipGetDevice(&cur);
  if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
  size_t cap = std::max(bytes, chunk_bytes);
  void* p = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess && cap > bytes) { cap = bytes; e = hipMalloc(&p, cap); }      // (the device is nearly full: an exact block)
  { static const bool trace = getenv("KMX_TRACE") != nullptr; if (trace) fprintf(stderr, "[kmx alloc] store chunk %zu MB: %.2f ms\n", cap >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
  if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
  if (e != hipSuccess) return nullptr;
  chunks.push_back({(u8*)p, cap, bytes});
  used += bytes;
  return p;
}
void* kmx_store::try_reserve(size_t bytes)
{
  bytes = (bytes + 255) / 256 * 256;
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> lk(mu);
  start_ahead();
  if (resvs.size() >= 8 || (limit && used + bytes > limit)) return nullptr;
  for (int pass = 0; pass < 2; pass++) {
    for (size_t i = 0; i < chunks.size(); i++) {
      auto& c = chunks[i];
      if (!chunk_reserved(i) && c.cap - c.fill >= bytes) { resvs.push_back({(int)i, c.fill, bytes}); void* p = c.p + c.fill; c.fill += bytes; used += bytes; return p; }
    }
    if (pass) break;
    // no free chunk with that much room: a new one, as alloc() makes it -- the one made ahead when there is one
    { Chunk sp; if (take_spare(bytes, sp)) { chunks.push_back(sp); continue; } }
    int cur = -1; (void)hipGetDevice(&cur);
    if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
    const size_t cap = std::max(bytes, chunk_bytes);
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, cap);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }      // (the device is nearly full: the caller waits for its size and takes exactly what it needs)
    chunks.push_back({(u8*)p, cap, 0});
  }
  return nullptr;
}
void kmx_store::commit(void* p, size_t used_bytes)
{
  used_bytes = (used_bytes + 255) / 256 * 256;
  std::lock_guard<std::mutex> lk(mu);
  for (size_t j = 0; j < resvs.size(); j++) {
    const Resv r = resvs[j];
    if (chunks[r.chunk].p + r.off != (u8*)p) continue;
    if (used_bytes > r.bytes) used_bytes = r.bytes;
    chunks[r.chunk].fill = r.off + used_bytes;      // (nothing was allocated behind the reservation: the chunk was left alone)
    used -= r.bytes - used_bytes;
    resvs.erase(resvs.begin() + j);
    return;
  }
}
extern "C" int kmx_store_create(int device, uint64_t limit_bytes, kmx_store** out)
{
  if (!out) return KMX_E_INVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_create_err = "kmx_store_create: no HIP device; libkmx has no CPU fallback"; return KMX_E_NODEVICE; }
  if (device < 0 || device >= n) { g_create_err = "kmx_store_create: device index out of range"; return KMX_E_INVAL; }
  int cur = -1; (void)hipGetDevice(&cur);
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "kmx_store_create: hipSetDevice failed"; return KMX_E_NODEVICE; }
  size_t fr = 0, tot = 0;
  (void)hipMemGetInfo(&fr, &tot);
  if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
  kmx_store* s = new kmx_store();
  s->device = device;
  s->limit = limit_bytes ? (size_t)limit_bytes : (size_t)(tot / 10 * 6);
  s->chunk_bytes = (size_t)256 << 20;
  { const char* e = getenv("KMX_STORE_CHUNK_MB"); if (e && atol(e) >= 16 && atol(e) <= 65536) s->chunk_bytes = (size_t)atol(e) << 20; }
  *out = s;
  return KMX_OK;
}
extern "C" void kmx_store_destroy(kmx_store* s)
{
  if (!s) return;
  int cur = -1; (void)hipGetDevice(&cur);
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; }
  s->cv.notify_all();
  if (s->ahead.joinable()) s->ahead.join();
  for (auto& c : s->spare) (void)hipFree(c.p);
  for (auto& c : s->chunks) (void)hipFree(c.p);
  if (cur >= 0 && cur != s->device) (void)hipSetDevice(cur);
  delete s;
}
// ---- peer access between the GPUs of a node: a counting context on GPU a fills the store of GPU b with hipMemcpyPeerAsync.  With
//      peer access enabled that copy is one DMA over the xGMI link between the two; without it the runtime stages it through host
//      memory.  Asked for once per ordered pair, the outcome kept (and printed with KMX_TRACE=1): 1 direct, 0 staged. ----
namespace {
std::mutex g_peer_mutex;
signed char g_peer[64][64];      // 0 unknown, 1 enabled, -1 not available
}
int kmx_peer_path(int from, int to)
{
  if (from == to) return 1;
  if (from < 0 || to < 0 || from >= 64 || to >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_peer_mutex);
  if (g_peer[from][to] == 0) {
    int can = 0, cur = -1;
    (void)hipGetDevice(&cur);
    signed char st = -1;
    if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can && hipSetDevice(from) == hipSuccess) {
      const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
      if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) st = 1;
      (void)hipGetLastError();
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    g_peer[from][to] = st;
    if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx] GPU %d -> GPU %d: %s\n", from, to, st == 1 ? "peer access enabled (copies go over xGMI)" : "no peer access (copies are staged through host memory)");
  }
  return g_peer[from][to] == 1 ? 1 : 0;
}
extern "C" int kmx_peer_access(int from_device, int to_device)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || from_device < 0 || to_device < 0 || from_device >= n || to_device >= n) return KMX_E_INVAL;
  return kmx_peer_path(from_device, to_device);
}
extern "C" uint64_t kmx_store_used(const kmx_store* s) { return s ? s->used : 0; }
extern "C" uint64_t kmx_store_limit(const kmx_store* s) { return s ? s->limit : 0; }
extern "C" int kmx_copy_to_host(kmx_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
  if (!ctx) return KMX_E_INVAL;
  if (!bytes) return KMX_OK;
  if (!dst || !src) return ctx->fail(KMX_E_INVAL, "kmx_copy_to_host: null pointer");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  KMX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  return KMX_OK;
}

extern "C" int kmx_copy_to_host_async(kmx_ctx* ctx, void* dst, const void* src, uint64_t bytes, uint32_t* ticket)
{
  if (!ctx) return KMX_E_INVAL;
  if (!ticket || (bytes && (!dst || !src))) return ctx->fail(KMX_E_INVAL, "kmx_copy_to_host_async: null pointer");
  int slot = -1;
  for (int i = 0; i < 8; i++) if (!ctx->copy_out[i]) { slot = i; break; }
  if (slot < 0) return ctx->fail(KMX_E_INVAL, "kmx_copy_to_host_async: KMX_COPIES_AHEAD tickets are out already");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->copy_ev[slot]) KMX_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_ev[slot], hipEventDisableTiming));
  if (bytes) KMX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipEventRecord(ctx->copy_ev[slot], ctx->copy));
  ctx->copy_out[slot] = true;
  *ticket = (uint32_t)slot;
  return KMX_OK;
}
extern "C" int kmx_copy_wait(kmx_ctx* ctx, uint32_t ticket)
{
  if (!ctx) return KMX_E_INVAL;
  if (ticket >= 8 || !ctx->copy_out[ticket]) return ctx->fail(KMX_E_INVAL, "kmx_copy_wait: no such ticket");
  ctx->copy_out[ticket] = false;
  KMX_HIP(ctx, hipEventSynchronize(ctx->copy_ev[ticket]));
  return KMX_OK;
}

extern "C" int kmx_set_profiling(kmx_ctx* ctx, int on) { if (!ctx) return KMX_E_INVAL; ctx->profiling = on != 0; return KMX_OK; }
extern "C" int kmx_set_file_order(kmx_ctx* ctx, int on) { if (!ctx) return KMX_E_INVAL; ctx->file_order = on != 0; return KMX_OK; }

// ---- merge ---------------------------------------------------------------------------------------------
struct TaskHost {
  u32 N = 0, kw = 1, mode = 0, bitw = 0, c = 1, wl = 0, pivot = 0, rt = 0;
  u32 rec_min = 0, share_min = 0, row_bytes = 0, seg_cap = 0;
  u64 lower = 0, upper = 0, total_recs = 0, out_cap_rows = 0, rows_guess = 0;
  u64 arena_rows = 0;            // rows claimed in the arena (kept rows + unused chunk tails)
  bool rows_overflow = false;    // ... and the kernel flagged that they did not fit
  std::vector<u32> len;
  // offsets into the meta blob
  size_t o_recs = 0, o_len = 0, o_smin = 0, o_bounds = 0, o_stats = 0, o_ctrl = 0, o_segs = 0;
  u8* d_out = nullptr; size_t out_bytes = 0;
  u16* d_rowrec = nullptr;                      // BFT: recurrence per hash row (k_bf_rowrec)
  u32 bft_c = 1;                                // BFT: work items (runs of consecutive tiles) of the task
  u8* d_img = nullptr; size_t img_bytes = 0;   // (unused since k_merge_bft writes the transposed matrix directly)
  u64 t_rows = 0, t_cols = 0;                   // BFT: rows / columns of that image rounded up to 8 (merge.hpp:634)
  Seg* d_segs = nullptr;        // directory in use (inside the meta blob, or d_segs_own after a retry)
  Seg* d_segs_own = nullptr;
  // results
  u64 rows = 0, nsegs = 0;
  bool done = false;
  bool handed_back = false;     // the pivot kernel flagged this task: it is (was) re-run with k_merge_rows
  // column-blocked kernel (merge_cols.hip)
  u32 nblk = 0, nb = 0, rt_cols = 0, slots_cap = 0;
  size_t o_skel = 0, o_nskel = 0, o_rbounds = 0;
  u8* d_ov = nullptr;           // the records that are not row keys (key, list, count), the slices' counts, and the directory of their rows
  size_t o_spdir = 0;           // ... offset of (the extension pool's cursor and) that directory in d_ov
  size_t o_ovx = 0; u32 xcap = 0;   // ... the pool the slices' extensions come from
  u64 sparse_rows = 0;          // rows k_cols_sparse added behind the row keys' rows
  u8* d_body = nullptr;         // COUNT/PA: the body assembled in file order on the device (kmx_result_body_dev), when it is not d_out itself
  bool body_ready = false;
  hipEvent_t ev_body = nullptr; // ... queued on the assembly stream by kmx_result_prepare_body: body_dev waits for it
  void* d_body_tmp = nullptr;   // ... its scratch (group offsets / segment copies), freed with the result
  // rows at their final place out of the column-blocked pair (kmx_set_file_order): the row keys' rows wait in d_dense for
  // k_cols_sparse, which writes every slice group's rows in key order at the group's place -- d_out IS the body
  u8* d_dense = nullptr; u32 dpitch = 0, dense_cap = 0; size_t o_gbase = 0;
  u8* d_narrow = nullptr; u32 npitch = 0;
  u64 row_keys = 0;             // row keys of the task as k_cols_prep counted them (ctrl[7])
  bool ordered = false;         // the task's rows lie in file order in d_out
  std::vector<u32> src;         // (row-key merge of a task: which of the task's lists it merges)
  int kernel = 0;               // the kernel that completed (or is to complete) the task: 0 rows, 1 pivot, 2 cols
};

struct kmx_merge_result {
  kmx_ctx* ctx = nullptr;
  std::vector<TaskHost> tasks;
  u8* d_meta = nullptr; size_t meta_bytes = 0;
  u8* h_meta = nullptr;              // pinned staging image of the meta blob
  size_t o_tasks = 0, o_items = 0, o_ticket = 0, o_ctrl0 = 0;
  u32 n_items = 0, grid = 0, max_n = 0, max_c = 0;
  int bf_lds = 0;
  bool is_bf = false, is_bft = false, waited = false;
  bool cols_ext = false, slices_full = false;   // k_merge_cols with slice extensions; a task came back because a slice was full
  bool cols_resc = false;                        // the RESC builds of the column-blocked pair (share-min, recurrence-min 0)
  bool share_fix = false;                        // ... with a task whose share-min is above its recurrence-min: k_share_fix behind them
  bool cols_ord = false;                         // ... and their ORD builds: rows written at their final place
  bool back_other = false;           // tasks were handed back for another reason than ERR_DENSE_CAP
  bool rows_small = false;           // k_merge_rows: the build for cohorts of up to 256 lists (merge_rows_small.hip) -- windows, grid and launch follow it
  bool cols_narrow = false;          // k_merge_cols' NAR build: every task's side store of row keys' rows is the byte-wide one
  bool rerun_rows = false;           // some tasks were re-run with k_merge_rows: any further re-run uses it for all
  bool pivot_auto = false;           // ... and it was libkmx's own choice (feeds the back-off in kmx_ctx)
  bool use_pivot = false;            // COUNT/PA: pivot-tiled kernel first, k_merge_rows as the general fallback
  // column-blocked kernel first (merge_cols.hip): its row keys come from a merge of a few lists of every task
  bool use_cols = false, cols_auto = false, can_pivot = false, auto_sel = false;
  bool divergent = false;            // k_cols_prep found lists that share too few keys: the tasks handed back go straight to k_merge_rows
  u32 given = 0;                     // tasks the kernel in hand was given (all of them, or the ones handed down to it)
  std::vector<TaskHost> subs;        // the row-key merges, one per task
  size_t o_subtasks = 0, o_subitems = 0, o_cols = 0, o_citems = 0;
  u32 n_subitems = 0, n_citems = 0, sub_grid = 0, cols_grid = 0, sub_max_c = 0, sub_max_n = 0, items_grid = 0;
  int status = KMX_OK;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the merge kernel when ctx->profiling
  hipEvent_t ev_mid = nullptr;               // ... between k_merge_cols and k_cols_sparse (kmx_result_kernel_parts_ms)
  hipEvent_t ev2 = nullptr;                  // BFT: behind the transposes (ev1 .. ev2 = their duration)
  u8* d_in = nullptr;                        // kmx_merge_host: the uploaded lists (freed with the result)
  u64* d_rem = nullptr; u32 rem_cap = 0;     // BFT, one-bit recurrences: the single walk's scratch (records put aside until a tile's map is complete)
  bool bft_two_walks = false;                // ... a tile ran out of it: the batch ran again with two walks
  hipEvent_t ev_in = nullptr;                // ... and the end of their upload
  u64* d_hctrl = nullptr;                    // device address of the control words' place in h_meta
  hipEvent_t ev_pre = nullptr;               // cols: preparation (second stream) done
  hipEvent_t ev_up = nullptr;                // cols: meta blob uploaded (second stream)
  hipEvent_t ev_done = nullptr;              // behind the last kernel queued for this result: what wait / read-back / free wait for
                                             // (not the stream: later batches are queued on it already)
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// k_merge_cols takes its row keys from a merge of S of the task's lists: the fewest (8..32) for which a key that 3.5 % of
// the lists lack is in fewer than recurrence-min of the S with probability < 2e-9 (C(S, m) q^m, m = S - r + 1 misses).
// 0: none up to 32 (recurrence-min above 21): the column-blocked kernel is not chosen.
static u32 cols_row_lists(u32 rec_min)
{
  for (u32 S = std::max<u32>(8, rec_min + 2); S <= 32; S++) {
    const u32 m = S - rec_min + 1;
    double p = 1.0;
    for (u32 i = 0; i < m; i++) p *= (double)(S - i) / (double)(i + 1) * 0.035;
    if (p < 2e-9) return S;
  }
  return 0;
}

// The tasks' control words go to the (pinned) host image of the meta blob with the batch itself: a device-to-host copy
// issued later is a blit kernel here, and it would wait for a CU behind the NEXT batch's merge.
__global__ void k_ctrl_mirror(const u64* __restrict__ src, u64* __restrict__ dst, u32 n)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
  __threadfence_system();
}
static int mirror_and_mark(kmx_merge_result* R);
__global__ void k_share_fix(const TaskDev* __restrict__ tasks, u32 kw);

static int launch_batch(kmx_merge_result* R, bool with_bounds)
{
  kmx_ctx* ctx = R->ctx;
  const TaskDev* d_tasks = reinterpret_cast<const TaskDev*>(R->d_meta + R->o_tasks);
  const uint2* d_items = reinterpret_cast<const uint2*>(R->d_meta + R->o_items);
  u32* d_ticket = reinterpret_cast<u32*>(R->d_meta + R->o_ticket);
  const int kw = (int)R->tasks[0].kw, mode = (int)R->tasks[0].mode;
  const ColsOps& CO = cols_ops(kw);
  const u32 nt = (u32)R->tasks.size();
  if (!R->use_cols) KMX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 16, ctx->stream));      // (cols runs once per result: the upload zeroed it)
  if (ctx->profiling && !R->ev0) { KMX_HIP(ctx, hipEventCreate(&R->ev0)); KMX_HIP(ctx, hipEventCreate(&R->ev1)); }
  if (R->is_bf) {
    if (with_bounds && !R->is_bft) KMX_HIP(ctx, launch_range_bounds_bf(d_tasks, nt, R->max_n, R->max_c, ctx->stream));      // (k_merge_bft finds where the lists enter its work items itself)
    if (R->ev0) KMX_HIP(ctx, hipEventRecord(R->ev0, ctx->stream));
    if (R->is_bft) {   // sample-major matrix straight from the merge (merge_bft.hip)
      u32 rt_max = 0; bool rec_bits = true, any_share = false;
      for (auto& H : R->tasks) { rt_max = std::max(rt_max, H.rt); rec_bits = rec_bits && std::max(H.rec_min, H.share_min) <= 1u; any_share = any_share || H.share_min > 0; }
      const u32 bgrid = std::min(R->n_items, (u32)ctx->n_cu * 2u);
      static const bool two_walks_env = getenv("KMX_BFT_TWO_WALKS") != nullptr;
      if (rec_bits && any_share && !R->bft_two_walks && !two_walks_env && !R->d_rem) {
        // the single walk's scratch: a quarter of a tile's records per workgroup (cohort data puts ~1 % aside), at most 1 GB in all
        u64 per_tile = 0;
        for (auto& H : R->tasks) { const u64 tiles = std::max<u64>(1, ((H.upper - H.lower + 1) + H.rt - 1) / H.rt); per_tile = std::max(per_tile, H.total_recs / tiles); }
        u64 cap = std::max<u64>(4096, per_tile / 4);
        cap = std::min<u64>(cap, ((1ull << 30) / 8) / std::max(1u, bgrid));
        R->rem_cap = (u32)cap;
        R->d_rem = (u64*)ctx->dalloc((size_t)bgrid * cap * 8);      // (no room for it: two walks)
      }
      const bool one_walk = R->d_rem && !R->bft_two_walks;
      // (rounds of 8 x 16 records per sample unless a tile is expected to hold more of the average list than that, with margin)
      bool wide = false;
      for (auto& H : R->tasks) { const double m = (double)H.total_recs / H.N * (double)H.rt / (double)(H.upper - H.lower + 1); wide = wide || m + 2.6 * std::sqrt(m) + 8 > (double)bft_round_records(false); }
      KMX_HIP(ctx, launch_merge_bft(d_tasks, d_items, R->n_items, d_ticket, bgrid, R->max_n, rt_max, rec_bits, wide, one_walk ? R->d_rem : nullptr, R->rem_cap, ctx->stream));
    } else
    KMX_HIP(ctx, launch_merge_bf(mode == KMX_MODE_BFC, d_tasks, d_items, R->n_items, d_ticket, R->grid, R->bf_lds, ctx->stream));
  } else if (R->use_cols) {
    // row keys first (bounds + k_merge_rows over a few lists of every task, gathered by k_cols_prep), then the
    // column-blocked merge, then the check that no key outside the rows reaches the recurrence
    const TaskDev* d_subs = reinterpret_cast<const TaskDev*>(R->d_meta + R->o_subtasks);
    const uint2* d_subitems = reinterpret_cast<const uint2*>(R->d_meta + R->o_subitems);
    const ColsDev* d_cols = reinterpret_cast<const ColsDev*>(R->d_meta + R->o_cols);
    const uint2* d_citems = reinterpret_cast<const uint2*>(R->d_meta + R->o_citems);
    // The preparation -- small, latency-bound kernels -- runs on the merge's own stream, behind the previous batch.  Round 1 put it on
    // a second stream, to fill the CUs the previous batch's merge leaves idle towards its end.  But k_merge_cols holds every CU with
    // one persistent workgroup; the row-key kernels then squeeze into what it leaves (a few wave slots, 34 KB of LDS), crawl
    // (k_cols_skel: 0.16 ms alone, up to 4.4 ms there) and slow the merge down with them -- or not, depending on when the host
    // happened to submit: configs[4] on lists from the count stage ran at 8.2 or 5.3 ms per launch with the same binary, and
    // recurrence-min 1 count rows at 0.59-0.66 or 0.70 of the roofline.  In line they cost their own 0.25 ms per batch and nothing
    // else (KMX_COLS_PREP_OVERLAP=1 brings the second stream back).
    static const bool overlap = getenv("KMX_COLS_PREP_OVERLAP") != nullptr;
    hipStream_t ps = overlap ? ctx->aux : ctx->stream;
    if (!R->ev_pre) KMX_HIP(ctx, hipEventCreateWithFlags(&R->ev_pre, hipEventDisableTiming));
    if (overlap && ctx->stream_shared) {   // whatever the caller queued on the stream (producers of the lists) comes first
      KMX_HIP(ctx, hipEventRecord(R->ev_pre, ctx->stream));
      KMX_HIP(ctx, hipStreamWaitEvent(ctx->aux, R->ev_pre, 0));
    }
    if (!overlap) KMX_HIP(ctx, hipStreamWaitEvent(ctx->stream, R->ev_up, 0));      // (the meta blob travels on the second stream)
    KMX_HIP(ctx, launch_range_bounds(kw, d_subs, nt, R->sub_max_n, R->sub_max_c, ps));
    KMX_HIP(ctx, CO.skel(d_subs, d_subitems, R->n_subitems, ps));
    KMX_HIP(ctx, CO.prep(d_tasks, d_subs, d_cols, nt, ps));
    KMX_HIP(ctx, hipEventRecord(R->ev_pre, ps));
    // (the task's own range bounds need the upload only: on the main stream, beside the row-key kernels)
    KMX_HIP(ctx, hipStreamWaitEvent(ctx->stream, R->ev_up, 0));
    KMX_HIP(ctx, launch_range_bounds(kw, d_tasks, nt, R->max_n, R->max_c, ctx->stream));
    KMX_HIP(ctx, hipStreamWaitEvent(ctx->stream, R->ev_pre, 0));
    if (R->ev0) KMX_HIP(ctx, hipEventRecord(R->ev0, ctx->stream));
    // (tuning knobs: KMX_COLS_GRID / KMX_SPARSE_GRID = workgroups of the two persistent kernels -- what each loses on a part of the chip)
    const u32 cols_grid_env = getenv("KMX_COLS_GRID") ? (u32)atoi(getenv("KMX_COLS_GRID")) : 0u;
    const u32 sparse_cus_env = getenv("KMX_SPARSE_CUS") ? (u32)atoi(getenv("KMX_SPARSE_CUS")) : 0u;
    KMX_HIP(ctx, CO.merge(mode, (R->cols_ext ? 1 : 0) | (R->cols_resc ? 2 : 0) | (R->cols_ord ? 4 : 0) | (R->cols_narrow ? 8 : 0), d_tasks, d_cols, d_citems, R->n_citems, d_ticket,
                          cols_grid_env ? std::min(cols_grid_env, R->cols_grid) : R->cols_grid, ctx->stream));
    if (R->ev0) { if (!R->ev_mid) KMX_HIP(ctx, hipEventCreate(&R->ev_mid)); KMX_HIP(ctx, hipEventRecord(R->ev_mid, ctx->stream)); }
    // (the second kernel stays on the merge's stream: on one of its own it takes CUs from the next batch's merge -- step +8 %)
    KMX_HIP(ctx, CO.sparse(mode | (R->cols_resc ? 2 : 0) | (R->cols_ord ? 4 : 0), d_tasks, d_cols, d_items, R->n_items, nt, d_ticket, sparse_cus_env ? std::min(sparse_cus_env, (u32)ctx->n_cu) : (u32)ctx->n_cu, ctx->stream));
    if (R->share_fix) {
      if ((size_t)R->max_n * 4 > 48 * 1024) KMX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_fix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>((size_t)R->max_n * 4, 160 * 1024)));
      hipLaunchKernelGGL(k_share_fix, dim3((unsigned)ctx->n_cu * 2u, nt), dim3(256), (size_t)R->max_n * 4, ctx->stream, (const TaskDev*)d_tasks, kw);
      KMX_HIP(ctx, hipGetLastError());
    }
    if (R->ev0) KMX_HIP(ctx, hipEventRecord(R->ev1, ctx->stream));      // (the launch priced: the rows come out of both kernels)
    return mirror_and_mark(R);
  } else {
    if (with_bounds) KMX_HIP(ctx, launch_range_bounds(kw, d_tasks, nt, R->max_n, R->max_c, ctx->stream));
    if (R->ev0) KMX_HIP(ctx, hipEventRecord(R->ev0, ctx->stream));
    if (R->use_pivot) KMX_HIP(ctx, launch_merge_pivot(kw, mode, d_tasks, d_items, R->n_items, d_ticket, R->grid, ctx->stream));
    else if (R->rows_small) KMX_HIP(ctx, launch_merge_rows_s(kw, mode, d_tasks, d_items, R->n_items, d_ticket, R->grid, R->max_n, ctx->stream));
    else KMX_HIP(ctx, launch_merge_rows(kw, mode, d_tasks, d_items, R->n_items, d_ticket, R->grid, R->max_n, ctx->stream));
  }
  if (R->ev0) KMX_HIP(ctx, hipEventRecord(R->ev1, ctx->stream));
  return mirror_and_mark(R);
}

// Share-min ABOVE the recurrence-min on the column-blocked pair (count rows).  k_cols_sparse sees all records of a key and rescues
// from share-min solid ones, whatever share-min is; k_merge_cols writes a row key's row block by block and rescues every non-solid
// record on the assumption that the row's recurrence-min solid records are share-min of them.  When they need not be
// (share-min > max(1, recurrence-min)) this pass over the finished rows takes those rescues back: a row with fewer than share-min
// solid counts (count >= soft-min of its list) loses its non-solid counts, and the statistics lose them (rows 1 and 5: rescued
// records and their sum; merge.hpp:234-247).  One read of the matrix (+0.5 ms on configs[2]) instead of k_merge_rows for the whole batch.
__global__ __launch_bounds__(256)
void k_share_fix(const TaskDev* __restrict__ tasks, u32 kw)
{
  extern __shared__ u32 sm_s[];      // the task's soft-mins
  const TaskDev& T = tasks[blockIdx.y];
  if (T.mode != KMX_MODE_COUNT || T.share_min <= max(1u, T.rec_min)) return;
  if (T.ctrl[2] & (u64)(ERR_FALLBACK | ERR_ROWS_OVERFLOW)) return;      // (handed back: k_merge_rows does the task over, with its own rescue)
  const u64 rows = T.ctrl[0];
  const u32 lane = threadIdx.x & 63u, N = T.N;
  for (u32 l = threadIdx.x; l < N; l += 256) sm_s[l] = T.soft_min[l];
  __syncthreads();
  const u64 nw = (u64)gridDim.x * 4u;
  for (u64 r = (u64)blockIdx.x * 4u + (threadIdx.x >> 6); r < rows; r += nw) {
    u32* const cnt = reinterpret_cast<u32*>(T.out + r * T.row_bytes + 8u * kw);
    u32 solid = 0, weak = 0;      // (weak: the row holds non-solid counts at all)
    for (u32 l0 = 0; l0 < N; l0 += 64u * 16u) {      // 16 loads of a lane in flight
      u32 c[16];
#pragma unroll
      for (int j = 0; j < 16; j++) { const u32 l = l0 + 64u * j + lane; c[j] = l < N ? cnt[l] : 0u; }
#pragma unroll
      for (int j = 0; j < 16; j++) { const u32 l = l0 + 64u * j + lane; if (c[j]) { if (c[j] >= sm_s[l]) solid++; else weak++; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { solid += (u32)__shfl_xor((int)solid, o); weak += (u32)__shfl_xor((int)weak, o); }
    if (solid >= T.share_min || !weak) continue;
    for (u32 l = lane; l < N; l += 64) {
      const u32 c = cnt[l];
      if (c && c < sm_s[l]) {
        cnt[l] = 0;
        atomicAdd(reinterpret_cast<unsigned long long*>(&T.stats[(u64)N + l]), ~0ULL);                       // - 1
        atomicAdd(reinterpret_cast<unsigned long long*>(&T.stats[5ull * N + l]), 0ULL - (unsigned long long)c);
      }
    }
  }
}

static int mirror_and_mark(kmx_merge_result* R)
{
  kmx_ctx* ctx = R->ctx;
  if (!R->d_hctrl) {
    void* dp = nullptr;
    KMX_HIP(ctx, hipHostGetDevicePointer(&dp, R->h_meta + R->o_ctrl0, 0));
    R->d_hctrl = reinterpret_cast<u64*>(dp);
  }
  const u32 n = (u32)R->tasks.size() * 8;
  hipLaunchKernelGGL(k_ctrl_mirror, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const u64*>(R->d_meta + R->o_ctrl0), R->d_hctrl, n);
  KMX_HIP(ctx, hipGetLastError());
  if (!R->ev_done) KMX_HIP(ctx, hipEventCreateWithFlags(&R->ev_done, hipEventDisableTiming));
  KMX_HIP(ctx, hipEventRecord(R->ev_done, ctx->stream));
  return KMX_OK;
}

extern "C" int kmx_merge_dev(kmx_ctx* ctx, const kmx_merge_task* tasks, uint32_t n_tasks, kmx_merge_result** out)
{
  if (!ctx) return KMX_E_INVAL;
  if (!tasks || !n_tasks || !out) return ctx->fail(KMX_E_INVAL, "kmx_merge_dev: null argument");
  *out = nullptr;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  const u32 kw = tasks[0].key_words, mode = tasks[0].mode;
  if (kw < 1 || kw > 4) return ctx->fail(KMX_E_INVAL, "key_words must be 1 ... 4");
  if (mode > KMX_MODE_BFT) return ctx->fail(KMX_E_INVAL, "unknown mode");
  const bool is_bft = mode == KMX_MODE_BFT;
  const bool is_bf = mode == KMX_MODE_BF || mode == KMX_MODE_BFC || is_bft;
  const ColsOps& CO = cols_ops((int)kw);
  if (is_bf && kw != 1) return ctx->fail(KMX_E_INVAL, "BF/BFC modes take hash keys (key_words = 1)");

  std::unique_ptr<kmx_merge_result> R(new kmx_merge_result());
  R->ctx = ctx; R->is_bf = is_bf; R->is_bft = is_bft;
  R->tasks.resize(n_tasks);
  u64 grand_total = 0;
  {   // cohorts of up to 256 lists (keys of one and two words): k_merge_rows' small build (KMX_ROWS_SMALL=0: never)
    u32 mxn0 = 0; for (u32 t = 0; t < n_tasks; t++) mxn0 = std::max(mxn0, tasks[t].n_lists);
    const char* e = getenv("KMX_ROWS_SMALL");      // (read per batch: the tests switch it)
    R->rows_small = !is_bf && kw <= 2 && mxn0 <= 256 && !(e && e[0] == '0');
  }
  const bool rsm = R->rows_small;
  for (u32 t = 0; t < n_tasks; t++) {
    const kmx_merge_task& K = tasks[t];
    TaskHost& H = R->tasks[t];
    if (K.key_words != kw || K.mode != mode) return ctx->fail(KMX_E_INVAL, "all tasks of a batch must share key_words and mode");
    if (K.n_lists == 0 || !K.lists || !K.soft_min) return ctx->fail(KMX_E_INVAL, "task without lists / soft_min");
    H.N = K.n_lists; H.kw = kw; H.mode = mode; H.bitw = K.bitw;
    H.rec_min = K.rec_min; H.share_min = K.share_min; H.lower = K.lower; H.upper = K.upper;
    if (!is_bf && H.N > (u32)rows_cap((int)kw, ~0u)) return ctx->fail(KMX_E_UNSUPPORTED, "more than " + std::to_string(rows_cap((int)kw, ~0u)) + " lists per merge task (COUNT/PA) not supported yet");
    H.len.resize(H.N);
    u32 pivot = 0;
    for (u32 i = 0; i < H.N; i++) {
      if (K.lists[i].n > 0xFFFFFFF0ULL) return ctx->fail(KMX_E_UNSUPPORTED, "list longer than 2^32 records");
      if (K.lists[i].n && !K.lists[i].recs) return ctx->fail(KMX_E_INVAL, "null record pointer");
      if (((uintptr_t)K.lists[i].recs) & 3u) return ctx->fail(KMX_E_INVAL, "record pointers must be 4-byte aligned");
      H.len[i] = (u32)K.lists[i].n;
      H.total_recs += K.lists[i].n;
      if (H.len[i] > H.len[pivot]) pivot = i;
    }
    const u32 longest = H.len[pivot];
    if (H.N >= 5) {
      // The pivot (range quantiles; row template of k_merge_pivot): in a cohort any list will do, and an
      // outlier (a sample with extra content, an empty one) is the worst template -- take the list of
      // median length among five spread over the task (O(1): this runs per task on the submit path).
      u32 cand[5] = {0, H.N / 4, H.N / 2, (3 * H.N) / 4, H.N - 1};
      std::sort(cand, cand + 5, [&](u32 a, u32 b) { return H.len[a] < H.len[b]; });
      if (H.len[cand[2]] > 0) pivot = cand[2];
    }
    H.pivot = pivot;
    grand_total += H.total_recs;
    if (is_bf) {
      if (K.upper < K.lower) return ctx->fail(KMX_E_INVAL, "BF window upper < lower");
      if (mode == KMX_MODE_BFC && (K.bitw == 0 || K.bitw > 32)) return ctx->fail(KMX_E_INVAL, "bitw must be in 1..32");
      H.row_bytes = mode != KMX_MODE_BFC ? (H.N + 7) / 8 : (u32)(((u64)H.N * K.bitw + 7) / 8);
      u32 rt = (40960u / H.row_bytes) & ~63u; if (rt < 64) rt = 64;
      if (!is_bft && (u64)rt * H.row_bytes > 96 * 1024) return ctx->fail(KMX_E_UNSUPPORTED, "BF row too wide for one LDS tile");
      H.rt = is_bft ? bft_tile_rows(K.rec_min, K.share_min) : rt;
      H.out_bytes = (size_t)((K.upper - K.lower + 1) * H.row_bytes);
      if (is_bft && H.N > bft_max_lists()) return ctx->fail(KMX_E_UNSUPPORTED, "more than " + std::to_string(bft_max_lists()) + " samples per hash:bft task not supported");
      if (is_bft) {   // write_as_bft (merge.hpp:631-644): BitMatrix(ROUND_UP(W, 8), ROUND_UP(N, 8) / 8), transposed, dumped whole
        H.t_rows = (K.upper - K.lower + 1 + 7) & ~7ULL; H.t_cols = (u64)H.row_bytes * 8;
        H.img_bytes = 0;
        H.nblk = (u32)((H.t_cols + bft_block_lists() - 1) / bft_block_lists());
        H.out_bytes = (size_t)(H.t_cols * (H.t_rows >> 3));
      }
    } else {
      H.row_bytes = kw * 8 + (mode == KMX_MODE_COUNT ? 4 * H.N : (H.N + 7) / 8);
      u32 wl = 0; while (wl < 6 && (H.N << (wl + 1)) <= (u32)(rsm ? rows_s_cap((int)kw, H.N) : rows_cap((int)kw, H.N))) wl++;   // window <= one wave
      H.wl = wl;
      if (H.row_bytes > (rsm ? rows_s_image_bytes((int)kw) : rows_image_bytes((int)kw))) return ctx->fail(KMX_E_UNSUPPORTED, "row wider than the LDS row image");
      u64 guess = K.rows_hint ? K.rows_hint : 2ULL * longest + 4096;
      if (ctx->rows_per_longest > 0.0) guess = std::max<u64>(guess, (u64)(ctx->rows_per_longest * 1.125 * (double)longest) + 4096);
      if (guess > H.total_recs) guess = H.total_recs;
      H.rows_guess = std::max<u64>(guess, 1);   // arena = guess + chunk slack, sized once c is known
    }
  }
  if (is_bft) {
    // tiles and work items of a hash:bft batch.  (1) The larger tile needs every task on the one-bit recurrence (the LDS regions are
    // sized per launch).  (2) One tile size for the batch, cut so that the tiles come to a whole number of rounds of one workgroup per
    // CU: configs[3]'s 4 x 3.9 M rows are 764 tiles of 20480 rows = 3 rounds of 256; with the largest tile (24576 rows: 636 tiles)
    // a third of the CUs idle through the last round.  (3) A work item is a RUN of up to `rounds` consecutive tiles of a task, about
    // one item per workgroup: where the lists enter an item is searched once, the cursors carry from tile to tile.
    bool all1 = true; u64 wsum = 0;
    for (auto& H : R->tasks) { all1 = all1 && std::max(H.rec_min, H.share_min) <= 1u; wsum += H.upper - H.lower + 1; }
    u32 mxn = 0; for (auto& H : R->tasks) mxn = std::max(mxn, H.N);
    const u32 cap = std::min(all1 ? bft_tile_rows(1, 1) : bft_tile_rows(2, 2), bft_fit_rows(mxn, all1));      // (thousands of samples: their cursors take the image's room)
    u32 rounds = 1, rt = cap;
    for (;; rounds++) {
      const u64 want = (wsum + (u64)rounds * ctx->n_cu - 1) / ((u64)rounds * ctx->n_cu);
      rt = (u32)std::max<u64>(1024, (want + 255) / 256 * 256);
      if (rt <= cap) break;
    }
    for (auto& H : R->tasks) {
      H.rt = rt;
      const u64 tiles = ((H.upper - H.lower + 1) + rt - 1) / rt;
      H.bft_c = (u32)std::max<u64>(1, (tiles + rounds - 1) / rounds);
      if (getenv("KMX_BFT_TILE_ITEMS")) H.bft_c = (u32)tiles;      // (tuning: an item per tile, handed out by ticket)
    }
  }
  // ---- which COUNT/PA kernel ----
  // k_merge_rows is the general one.  k_merge_pivot (merge_pivot.hip; 64-bit keys, no share-min, <= 1024
  // lists) is faster when MANY lists share most of their keys -- the cohort case the metric is quoted on --
  // and k_merge_cols (merge_cols.hip; a small recurrence-min) faster still there.  Both flag tasks
  // they do not suit, and those are re-run with the next kernel down (cols -> pivot -> rows, see
  // kmx_result_wait; below 513 lists cols -> rows).  Default: from 192 lists per task and recurrence-min <= 21
  // cols; otherwise pivot above 512 lists (where k_merge_rows is down to 4-record windows), else rows.
  // KMX_MERGE_KERNEL=rows|pivot|cols forces one of them (where it is applicable).
  {
    bool rescue = false; u32 min_n = 0xFFFFFFFFu, mx_n = 0, min_rec = 0xFFFFFFFFu, max_rec = 0;
    for (auto& H : R->tasks) {
      rescue |= H.share_min > 0; min_n = std::min(min_n, H.N); mx_n = std::max(mx_n, H.N);
      min_rec = std::min(min_rec, H.rec_min); max_rec = std::max(max_rec, H.rec_min);
    }
    const char* force = getenv("KMX_MERGE_KERNEL");
    // (recurrence-min 0 -- a key only non-solid records hold is a row of zeros -- is beyond k_merge_pivot: it drops such records before
    //  they reach its tables; found by tests/test_merge_gpu.py::test_hip_merge_against_the_second_restatement, round 5)
    const bool can = !is_bf && !rescue && kw == 1 && mx_n <= pivot_max_lists() && min_rec >= 1;
    // (both key widths: merge_cols.hip, merge_cols_k2.hip; share-min up to max(1, recurrence-min): the RESC builds, which also take
    //  recurrence-min 0 -- there a key only non-solid records hold is a row)
    bool resc_ok = true, need_resc = false;
    // (share-min above that on count rows: the RESC builds + k_share_fix behind them)
    R->share_fix = false;
    for (auto& H : R->tasks) {
      const bool above = H.share_min > std::max(1u, H.rec_min);
      resc_ok = resc_ok && (!above || mode == KMX_MODE_COUNT); need_resc = need_resc || H.share_min > 0 || H.rec_min == 0;
      R->share_fix = R->share_fix || (above && mode == KMX_MODE_COUNT);
    }
    bool can_cols = !is_bf && resc_ok && kw <= 2 && mx_n <= (u32)rows_cap((int)kw, mx_n);      // (keys of three and four words: k_merge_rows)
    R->cols_resc = need_resc;
    if (can_cols) {
      // k_merge_cols keeps worst-case room for the records it sets aside (a slice per half tile, block and wave: ~2.3 KB per
      // row at 1000 lists): not for batches where that would take more than KMX_COLS_SCRATCH_GB (default 32) of HBM
      u64 scratch = 0;
      for (auto& H : R->tasks) {
        const u32 nblk = (H.N + CO.block_lists() - 1) / CO.block_lists();
        // (tile slots count ROW KEYS -- the keys a merge of S of the lists keeps, at most the records of those lists -- not rows)
        const u32 S_ = std::max(8u, cols_row_lists(std::max(1u, H.rec_min)));
        const u64 skel_est = std::min<u64>(H.rows_guess, (u64)S_ * (H.total_recs / H.N + 1) * 5 / 4 + 4096);
        const u32 sl = (u32)std::min<u64>(0x7FFFFFF0ULL, skel_est / CO.tile_rows(CO.block_lists()) + 64);
        scratch += CO.scratch_keys(sl, nblk) * 8 + CO.scratch_counts(sl, nblk) * 8 + CO.ext_entries(sl, nblk) * (CO.key_words + 1) * 8 + CO.dir_bytes(sl);
      }
      const char* gb = getenv("KMX_COLS_SCRATCH_GB");
      const char* mb = getenv("KMX_COLS_SCRATCH_MB");
      const u64 budget = mb && atoi(mb) > 0 ? (u64)atoi(mb) << 20 : (u64)(gb && atoi(gb) > 0 ? atoi(gb) : 32) << 30;
      can_cols = scratch <= budget;
    }
    R->can_pivot = can && min_n > 512;      // (as the next kernel down from cols: below 512 lists k_merge_rows is the faster of the two)
    if (force && !strcmp(force, "cols")) { R->use_cols = can_cols; R->use_pivot = !can_cols && can; }
    else if (force && !strcmp(force, "pivot")) R->use_pivot = can;
    else if (force && !strcmp(force, "rows")) R->use_pivot = false;
    else {
      // (cols: from 192 lists -- at 128, k_merge_rows' wide windows and single launch win by 0.15 ms per batch; at 200 cols does --
      //  and 1 M records per batch -- below, its seven launches cost more than they save; recurrence-min up to 21)
      // (rows in file order, count rows with 64-bit keys: from 257 lists -- up to 256 k_merge_rows keeps its wide windows and writes
      //  the rows at their place at no cost, where the pair pays the look-back and the side store: 1.19 / 1.40 ms against 1.72 / 1.93
      //  at 192 / 256 lists, 2.38 against 2.11 at 320, profiles/r04_crossover.txt; PA rows and 128-bit keys: not measured, as before)
      const bool ord_count = ctx->file_order && kw == 1 && R->tasks[0].mode == KMX_MODE_COUNT;
      // (PA rows with 128-bit keys -- k = 32 ... 63, configs[4]'s shape --: k_merge_rows pays for its 20-byte records; the pair is ahead
      //  from 96 lists with the rows left where they fall (1.70 against 2.31 ms) and from 160 in file order (3.09 against 3.91 ms; even
      //  at 128), profiles/r04_crossover_pa63.txt.  The environment's numbers, when given, hold for every kind of row)
      const bool pa_wide = kw == 2 && mode == KMX_MODE_PA && !ctx->cols_min_env;
      const u32 min_lists = pa_wide ? (ctx->file_order ? 160u : 96u) : (ord_count ? ctx->cols_min_lists_ord : ctx->cols_min_lists);
      R->use_cols = can_cols && min_n >= min_lists && grand_total >= (1ull << 20) && cols_row_lists(std::max(1u, max_rec)) != 0;
      if (R->use_cols && ctx->cols_skip) { ctx->cols_skip--; R->use_cols = false; }
      R->cols_auto = R->use_cols;
      R->use_pivot = !R->use_cols && can && min_n > 512;
      if (R->use_pivot && ctx->pivot_skip) { ctx->pivot_skip--; R->use_pivot = false; }   // cohort that did not suit it recently
      R->pivot_auto = R->use_pivot;
      R->auto_sel = true;
    }
    R->given = n_tasks;
    // (an outlier by length is known up front: a list of more than 2.5 times the task's mean length)
    bool long_list = false;
    for (auto& H : R->tasks) { u32 longest = 0; for (u32 l : H.len) longest = std::max(longest, l); long_list |= (u64)longest * H.N * 2 > H.total_recs * 5; }
    R->cols_ext = R->use_cols && (ctx->cols_ext || long_list);
    R->cols_ord = R->use_cols && ctx->file_order;
  }
  // ranges per task: ~3 work items per resident workgroup over the batch, >= 16K records each
  const u32 slots = (u32)ctx->n_cu * 2;
  const char* ipc = getenv("KMX_ITEMS_PER_SLOT");            // tuning knob (default 3): work items per resident workgroup slot
  // (3 for the general kernels; 6 for the column-blocked pair: on lists from the count stage -- uneven key density, k_cols_sparse's
  //  groups -- finer items balance better: 5.2 -> 4.8 ms per step, +2 % on uniform random keys; scripts/r2_tune.sh)
  const u32 per_slot = ipc && atoi(ipc) > 0 ? (u32)atoi(ipc) : (R->use_cols ? 6u : 3u);
  // (cols: one workgroup per CU, and a work item is a column block of a range.  Leaving a few CUs to the small kernels
  //  that prepare the NEXT batch on the second stream was tried: the merge then needs finer work items, net +5 %)
  const u32 cols_cus = (u32)ctx->n_cu * CO.wgs_per_cu();
  const u32 target_items = (R->use_cols ? cols_cus : slots) * per_slot;
  u32 n_items = 0, max_n = 0, max_c = 0;
  // the byte-wide side store of the row keys' rows (count rows in file order; k_merge_cols' NAR build): count rows of 512 to 1022 lists in
  // at most 8 column blocks -- every task of the batch, the build is chosen per launch (KMX_DENSE_NARROW=1: from any size, =0: never)
  R->cols_narrow = false;
  if (R->use_cols && R->cols_ord && mode == KMX_MODE_COUNT) {
    const char* const nenv = getenv("KMX_DENSE_NARROW");
    bool all = true;
    for (auto& H : R->tasks) {
      const u32 nblk0 = (H.N + CO.block_lists() - 1) / CO.block_lists();
      const bool on = nenv ? nenv[0] != '0' : H.N >= 512;
      all = all && on && (H.row_bytes & 7u) == 0 && H.row_bytes / 8 <= 512 && nblk0 <= 8 && CO.block_lists() % 16 == 0;
    }
    R->cols_narrow = all;
  }
  for (auto& H : R->tasks) {
    if (R->use_cols) {
      H.nblk = (H.N + CO.block_lists() - 1) / CO.block_lists();
      // (lists per block: even for count rows -- 8-byte stores --, a multiple of 8 for PA rows -- whole bytes per block)
      H.nb = std::min<u32>(CO.block_lists(), mode == KMX_MODE_COUNT ? ((((H.N + H.nblk - 1) / H.nblk) + 1) & ~1u) : ((((H.N + H.nblk - 1) / H.nblk) + 7) & ~7u));
      if (R->cols_narrow) H.nb = std::min<u32>(CO.block_lists(), (((H.N + H.nblk - 1) / H.nblk) + 15) & ~15u);      // (a row's slice of a block leaves as 16-byte pieces)
      H.nblk = (H.N + H.nb - 1) / H.nb;
      H.rt_cols = CO.tile_rows(H.nb);
    }
    u64 c = grand_total ? (u64)target_items * H.total_recs / grand_total : 1;
    if (R->use_cols) c = (c + H.nblk / 2) / H.nblk;
    const u64 cmax_work = std::max<u64>(1, H.total_recs / 16384);
    c = std::min(c, cmax_work);
    if (is_bf) {
      const u64 tiles = ((H.upper - H.lower + 1) + H.rt - 1) / H.rt;
      c = std::min<u64>(c, tiles);
    } else c = std::min<u64>(c, std::max<u32>(1, H.len[H.pivot]));
    H.c = (u32)std::max<u64>(1, c);
    if (is_bft) H.c = H.bft_c;
    H.seg_cap = is_bf ? 1 : (u32)std::min<u64>(0x7FFFFFFF, H.total_recs / 512 + 8ULL * H.c + 4096);
    if (!is_bf) {   // rows are claimed in chunks: every range may leave one chunk partly unused
      H.out_cap_rows = H.rows_guess + (u64)(H.c + 1) * rows_chunk_rows(H.row_bytes);
      H.out_bytes = (size_t)(H.out_cap_rows * H.row_bytes);
    }
    if (R->use_cols) {
      const u32 S_ = std::max(8u, cols_row_lists(std::max(1u, H.rec_min)));
      const u64 skel_est = std::min<u64>(H.out_cap_rows, (u64)S_ * (H.total_recs / H.N + 1) * 5 / 4 + 4096);
      H.slots_cap = (u32)std::min<u64>(0x7FFFFFF0ULL, skel_est / H.rt_cols + H.c + 2);
    }
    n_items += H.c;
    max_n = std::max(max_n, H.N); max_c = std::max(max_c, H.c);
  }
  R->n_items = n_items; R->max_n = max_n; R->max_c = max_c;
  R->grid = std::min(n_items, is_bf ? slots : (u32)ctx->n_cu * (u32)(R->rows_small ? rows_s_wgs_per_cu((int)kw) : rows_wgs_per_cu((int)kw)));
  if (R->use_pivot) R->grid = std::min(n_items, (u32)ctx->n_cu);   // one 1024-thread workgroup per CU
  if (R->use_cols) {
    // the row-key merges: up to 8 lists spread over each task, same recurrence-min
    R->subs.resize(n_tasks);
    u32 nsub = 0, ncit = 0;
    for (u32 t = 0; t < n_tasks; t++) {
      const TaskHost& H = R->tasks[t];
      TaskHost& Q = R->subs[t];
      { const u32 S = cols_row_lists(std::max(1u, H.rec_min)); Q.N = std::min<u32>(S ? S : 32u, H.N); }
      Q.kw = kw; Q.mode = mode; Q.rec_min = H.rec_min; Q.share_min = 0;
      Q.len.resize(Q.N);
      u32 piv = 0;
      Q.src.resize(Q.N);
      std::vector<u32> win;
      for (u32 i = 0; i < Q.N; i++) {
        // one list from each of Q.N stretches of the task: the one of median length there (an empty or an outlier
        // list is a poor witness of what the cohort shares)
        const u32 w0 = (u32)(((u64)i * H.N) / Q.N), w1 = std::max(w0 + 1, (u32)(((u64)(i + 1) * H.N) / Q.N));
        win.clear();
        for (u32 j = w0; j < w1; j++) win.push_back(j);
        std::nth_element(win.begin(), win.begin() + win.size() / 2, win.end(), [&](u32 a, u32 b) { return H.len[a] != H.len[b] ? H.len[a] < H.len[b] : a < b; });
        const u32 src = win[win.size() / 2];
        Q.src[i] = src;
        Q.len[i] = H.len[src]; Q.total_recs += H.len[src];
        if (Q.len[i] > Q.len[piv]) piv = i;
      }
      Q.pivot = (Q.N >= 2 && Q.len[Q.N / 2] > 0) ? Q.N / 2 : piv;
      Q.row_bytes = 8 * kw;                              // k_cols_skel writes keys only
      u32 wl = 0; while (wl < 6 && (Q.N << (wl + 1)) <= (u32)rows_cap((int)kw, Q.N)) wl++;
      Q.wl = wl;
      Q.rows_guess = std::max<u64>(1, std::min<u64>(H.rows_guess, Q.total_recs));
      // ranges of ~1500 records over the lists (k_cols_skel sorts a range in LDS, <= 2048 records): one segment per range
      u64 c = std::min<u64>(2000, std::max<u64>(1, Q.total_recs / 1500));
      c = std::min<u64>(c, std::max<u32>(1, Q.len[Q.pivot]));
      Q.c = (u32)c;
      Q.seg_cap = Q.c;
      Q.out_cap_rows = (u64)Q.c * CO.skel_cap();
      Q.out_bytes = (size_t)(Q.out_cap_rows * Q.row_bytes);
      nsub += Q.c; ncit += H.c * H.nblk;
      R->sub_max_c = std::max(R->sub_max_c, Q.c); R->sub_max_n = std::max(R->sub_max_n, Q.N);
    }
    R->n_subitems = nsub; R->n_citems = ncit;
    R->sub_grid = std::min(nsub, (u32)ctx->n_cu * (u32)rows_wgs_per_cu((int)kw));
    R->cols_grid = std::min(ncit, cols_cus);
  }
  if (is_bf) {
    int lds = 0;
    for (auto& H : R->tasks) lds = std::max(lds, bf_lds_bytes(H.rt, H.row_bytes, H.N));
    R->bf_lds = lds;
  }

  for (auto& H : R->tasks) H.kernel = R->use_cols ? 2 : R->use_pivot ? 1 : 0;
  // ---- meta blob layout ----
  const bool cols = R->use_cols;
  size_t off = 0;
  R->o_tasks = off; off = align_up(off + sizeof(TaskDev) * n_tasks, 256);
  R->o_items = off; off = align_up(off + sizeof(uint2) * n_items, 256);
  R->o_ticket = off; off += 256;
  R->o_ctrl0 = off;                       // control words of all tasks, contiguous: one D2H copy reads them all
  for (auto& H : R->tasks) { H.o_ctrl = off; off += 64; }
  off = align_up(off, 256);
  auto lay_lists = [&](TaskHost& H) {
    H.o_recs = off; off = align_up(off + 8ull * H.N, 256);
    H.o_len = off; off = align_up(off + 4ull * H.N, 256);
    H.o_smin = off; off = align_up(off + 4ull * H.N, 256);
    H.o_stats = off; off = align_up(off + 8ull * 6 * H.N, 256);
  };
  for (auto& H : R->tasks) lay_lists(H);
  if (is_bft) {
    u32 ncit = 0; for (auto& H : R->tasks) ncit += H.c * H.nblk;
    R->n_citems = ncit;
    R->o_citems = off; off = align_up(off + sizeof(uint2) * ncit, 256);
  }
  if (cols) {
    R->o_subtasks = off; off = align_up(off + sizeof(TaskDev) * n_tasks, 256);
    R->o_subitems = off; off = align_up(off + sizeof(uint2) * R->n_subitems, 256);
    R->o_cols = off; off = align_up(off + sizeof(ColsDev) * n_tasks, 256);
    R->o_citems = off; off = align_up(off + sizeof(uint2) * R->n_citems, 256);
    for (auto& Q : R->subs) { Q.o_ctrl = off; off += 64; }
    off = align_up(off, 256);
    for (auto& Q : R->subs) lay_lists(Q);
  }
  const size_t upload_bytes = off;            // everything above is written by the host
  auto lay_work = [&](TaskHost& H) {
    H.o_bounds = off; off = align_up(off + 4ull * (H.c + 1) * H.N, 256);
    H.o_segs = off; off = align_up(off + sizeof(Seg) * (size_t)H.seg_cap, 256);
  };
  for (auto& H : R->tasks) lay_work(H);
  if (cols) {
    for (auto& Q : R->subs) lay_work(Q);
    for (auto& H : R->tasks) {
      H.o_skel = off; off = align_up(off + 8ull * kw * H.out_cap_rows, 256);
      H.o_nskel = off; off += 256;
      H.o_rbounds = off; off = align_up(off + 4ull * (H.c + 1), 256);
      H.o_gbase = off; off = align_up(off + 4ull * (H.c + 1), 256);
    }
  }
  R->meta_bytes = off;
  R->d_meta = (u8*)ctx->dalloc(off);
  R->h_meta = (u8*)ctx->halloc(upload_bytes);
  if (!R->d_meta || !R->h_meta) { ctx->dfree(R->d_meta); ctx->hfree(R->h_meta); return ctx->fail(KMX_E_NOMEM, "meta allocation failed"); }
  auto drop_blocks = [&]() {
    for (auto& G : R->tasks) { ctx->dfree(G.d_out); ctx->dfree(G.d_ov); ctx->dfree(G.d_img); ctx->dfree(G.d_rowrec); ctx->dfree(G.d_dense); ctx->dfree(G.d_narrow); }
    for (auto& G : R->subs) ctx->dfree(G.d_out);
    ctx->dfree(R->d_meta); ctx->hfree(R->h_meta); ctx->dfree(R->d_rem); R->d_rem = nullptr;
  };
  bool narrow_failed = false;
  for (auto& H : R->tasks) {
    H.d_out = (u8*)ctx->dalloc(H.out_bytes);
    if (H.d_out && cols) {
      // [set-aside slices][their counts][their extensions' places][the extension pool] | [the pool's cursor][the sparse rows' directory]
      H.xcap = (u32)std::min<u64>(0x7FFFFF00ULL, CO.ext_entries(H.slots_cap, H.nblk));
      H.o_ovx = align_up((size_t)(CO.scratch_keys(H.slots_cap, H.nblk) * 8 + CO.scratch_counts(H.slots_cap, H.nblk) * 8), 256);
      H.o_spdir = align_up(H.o_ovx + (size_t)H.xcap * (CO.key_words + 1) * 8, 256);
      H.d_ov = (u8*)ctx->dalloc(H.o_spdir + 256 + (size_t)CO.dir_bytes(H.slots_cap));
      if (H.d_ov && hipMemsetAsync(H.d_ov + H.o_spdir, 0, 256 + (size_t)CO.dir_bytes(H.slots_cap), ctx->aux) != hipSuccess) { ctx->dfree(H.d_ov); H.d_ov = nullptr; }
      if (R->cols_ord) {
        // the side store of the row keys' rows (payload only, 8-byte pitch).  Row keys = the keys a merge of S of the lists keeps:
        // about one list's worth in a cohort (recurrence-min 1: the union of the S lists), or what the context's batches had
        const TaskHost& Q = R->subs[&H - R->tasks.data()];
        u64 longest = 0, sum = 0; for (u32 l : Q.len) { longest = std::max<u64>(longest, l); sum += l; }
        u64 cap = (H.rec_min <= 1 ? std::min<u64>(sum, longest * 5 / 2) : longest * 3 / 2) + 4096;
        if (ctx->keys_per_longest > 0.0) cap = std::max<u64>(cap, (u64)(ctx->keys_per_longest * 1.25 * (double)longest) + 4096);
        H.dense_cap = (u32)std::min<u64>(std::min<u64>(cap, H.out_cap_rows), 0xFFFFFF00ULL);
        H.dpitch = (u32)align_up(H.row_bytes - 8 * kw, 8);
        H.d_dense = (u8*)ctx->dalloc((size_t)H.dense_cap * H.dpitch);
        // count rows of up to 1022 lists in at most 8 column blocks: a byte per count where a block's counts of a row all fit one
        // (+ 8 flag bytes): what k_merge_cols writes and k_cols_sparse reads back is then a quarter of the row (KMX_DENSE_NARROW=0:
        // the 4-byte rows only)
        // (from 512 lists: below, the two ways cost the same within the noise -- grid of N = 200 .. 1000, profiles/r04_dense_narrow_store.txt --;
        //  KMX_DENSE_NARROW=1: wherever it applies, =0: nowhere.  Read per batch: the tests switch it)
        if (R->cols_narrow) {      // [N count bytes, padded to 16][16 bytes: a flag byte per column block]
          H.npitch = (u32)align_up((size_t)H.N, 16) + 16;
          H.d_narrow = (u8*)ctx->dalloc((size_t)H.dense_cap * H.npitch);
          if (!H.d_narrow) narrow_failed = true;      // (no room for the byte-wide rows: the batch runs with the 4-byte rows alone, as before round 5)
        }
      }
    }
    if (!H.d_out || (cols && !H.d_ov) || (R->cols_ord && !H.d_dense)) { drop_blocks(); return ctx->fail(KMX_E_NOMEM, "output arena allocation failed"); }

  }
  if (narrow_failed) { for (auto& H : R->tasks) { ctx->dfree(H.d_narrow); H.d_narrow = nullptr; } R->cols_narrow = false; }      // (decided before the ColsDev structs are filled)
  for (auto& Q : R->subs) {
    Q.d_out = (u8*)ctx->dalloc(Q.out_bytes);
    if (!Q.d_out) { drop_blocks(); return ctx->fail(KMX_E_NOMEM, "row-key arena allocation failed"); }
  }
  memset(R->h_meta, 0, upload_bytes);
  TaskDev* td = reinterpret_cast<TaskDev*>(R->h_meta + R->o_tasks);
  uint2* items = reinterpret_cast<uint2*>(R->h_meta + R->o_items);
  auto fill_dev = [&](TaskHost& H, TaskDev& D) {   // everything but the list tables
    D.recs = reinterpret_cast<const u8* const*>(R->d_meta + H.o_recs);
    D.len = reinterpret_cast<const u32*>(R->d_meta + H.o_len);
    D.soft_min = reinterpret_cast<const u32*>(R->d_meta + H.o_smin);
    D.bounds = reinterpret_cast<u32*>(R->d_meta + H.o_bounds);
    D.stats = reinterpret_cast<u64*>(R->d_meta + H.o_stats);
    D.out = H.d_out;
    D.rowrec = H.d_rowrec;
    D.ctrl = reinterpret_cast<u64*>(R->d_meta + H.o_ctrl);
    H.d_segs = reinterpret_cast<Seg*>(R->d_meta + H.o_segs);
    D.segs = H.d_segs;
    D.out_cap_rows = H.out_cap_rows; D.lower = H.lower; D.upper = H.upper;
    D.seg_cap = H.seg_cap; D.N = H.N; D.c = H.c; D.rec_min = H.rec_min; D.share_min = H.share_min;
    D.mode = H.mode; D.bitw = H.bitw; D.row_bytes = H.row_bytes; D.wl = H.wl; D.pivot = H.pivot; D.rt = H.rt;
  };
  u32 it = 0;
  for (u32 t = 0; t < n_tasks; t++) {
    TaskHost& H = R->tasks[t];
    const kmx_merge_task& K = tasks[t];
    const u8** recs = reinterpret_cast<const u8**>(R->h_meta + H.o_recs);
    u32* len = reinterpret_cast<u32*>(R->h_meta + H.o_len);
    u32* smin = reinterpret_cast<u32*>(R->h_meta + H.o_smin);
    for (u32 i = 0; i < H.N; i++) { recs[i] = (const u8*)K.lists[i].recs; len[i] = H.len[i]; smin[i] = K.soft_min[i]; }
    TaskDev& D = td[t];
    fill_dev(H, D);
    D.item0 = it;
    for (u32 j = 0; j < H.c; j++) items[it++] = make_uint2(t, j);
  }
  if (is_bft) {
    uint2* citems = reinterpret_cast<uint2*>(R->h_meta + R->o_citems);
    u32 ci = 0;
    // (sample blocks of a range next to each other: the blocks of a tile share the rows' recurrences in the L2)
    for (u32 t = 0; t < n_tasks; t++) for (u32 j = 0; j < R->tasks[t].c * R->tasks[t].nblk; j++) citems[ci++] = make_uint2(t, j);
  }
  if (cols) {
    TaskDev* sd = reinterpret_cast<TaskDev*>(R->h_meta + R->o_subtasks);
    uint2* sitems = reinterpret_cast<uint2*>(R->h_meta + R->o_subitems);
    ColsDev* cd = reinterpret_cast<ColsDev*>(R->h_meta + R->o_cols);
    uint2* citems = reinterpret_cast<uint2*>(R->h_meta + R->o_citems);
    u32 si = 0, ci = 0;
    for (u32 t = 0; t < n_tasks; t++) {
      TaskHost& H = R->tasks[t];
      TaskHost& Q = R->subs[t];
      const kmx_merge_task& K = tasks[t];
      const u8** recs = reinterpret_cast<const u8**>(R->h_meta + Q.o_recs);
      u32* len = reinterpret_cast<u32*>(R->h_meta + Q.o_len);
      u32* smin = reinterpret_cast<u32*>(R->h_meta + Q.o_smin);
      for (u32 i = 0; i < Q.N; i++) {
        const u32 src = Q.src[i];
        recs[i] = (const u8*)K.lists[src].recs; len[i] = Q.len[i]; smin[i] = K.soft_min[src];
      }
      fill_dev(Q, sd[t]);
      sd[t].item0 = si;
      for (u32 j = 0; j < Q.c; j++) sitems[si++] = make_uint2(t, j);
      ColsDev& C = cd[t];
      C.skel = reinterpret_cast<u64*>(R->d_meta + H.o_skel);
      C.nskel = reinterpret_cast<u32*>(R->d_meta + H.o_nskel);
      C.rbounds = reinterpret_cast<u32*>(R->d_meta + H.o_rbounds);
      C.ovkeys = reinterpret_cast<u64*>(H.d_ov);
      C.ovcnt = reinterpret_cast<u32*>(H.d_ov + CO.scratch_keys(H.slots_cap, H.nblk) * 8);
      C.ovx = reinterpret_cast<u64*>(H.d_ov + H.o_ovx);
      C.xcur = reinterpret_cast<u32*>(H.d_ov + H.o_spdir);
      C.xcap = H.xcap;
      C.spdir = H.d_ov + H.o_spdir + 256;
      C.slots_cap = H.slots_cap; C.nblk = H.nblk; C.nb = H.nb; C.rt = H.rt_cols;
      if (R->cols_ord) {      // (the sparse rows' directory is not written then: its room holds the look-back chain and the group map)
        const size_t ng = CO.groups(H.slots_cap);
        C.dense = H.d_dense; C.dpitch = H.dpitch; C.dense_cap = H.dense_cap;
        C.dnarrow = H.d_narrow; C.npitch = H.npitch;
        C.gbase = reinterpret_cast<u32*>(R->d_meta + H.o_gbase);
        C.chain = reinterpret_cast<u64*>(H.d_ov + H.o_spdir + 256);
        C.gmap = reinterpret_cast<uint4*>(H.d_ov + H.o_spdir + 256 + ng * 8);
        C.gmax = reinterpret_cast<u32*>(R->d_meta + R->o_ticket) + 2;
        C.ngcap = (u32)ng;
        // (ticket word 3: no task of the batch has room for more groups than this -- k_cols_sparse's tickets end there whatever word 2 says)
        u32* const tk3 = reinterpret_cast<u32*>(R->h_meta + R->o_ticket) + 3;
        *tk3 = std::max<u32>(*tk3, (u32)ng);
      }
      for (u32 j = 0; j < H.c * H.nblk; j++) citems[ci++] = make_uint2(t, j);   // y = range * nblk + block
    }
  }
  auto drop = [&]() {   // hand every block back to the pool on a failed launch
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->aux);
    drop_blocks();
    if (R->ev0) { (void)hipEventDestroy(R->ev0); (void)hipEventDestroy(R->ev1); }
    if (R->ev_mid) (void)hipEventDestroy(R->ev_mid);
    if (R->ev_pre) (void)hipEventDestroy(R->ev_pre);
    if (R->ev_up) (void)hipEventDestroy(R->ev_up);
    if (R->ev_done) (void)hipEventDestroy(R->ev_done);
  };
  hipError_t he = hipMemcpyAsync(R->d_meta, R->h_meta, upload_bytes, hipMemcpyHostToDevice, R->use_cols ? ctx->aux : ctx->stream);
  if (he != hipSuccess) { drop(); return ctx->fail(KMX_E_HIP, std::string("meta upload: ") + hipGetErrorString(he)); }
  if (R->use_cols) {
    he = hipEventCreateWithFlags(&R->ev_up, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventRecord(R->ev_up, ctx->aux);
    if (he != hipSuccess) { drop(); return ctx->fail(KMX_E_HIP, std::string("meta upload event: ") + hipGetErrorString(he)); }
  }
  int rc = launch_batch(R.get(), true);
  if (rc != KMX_OK) { drop(); return rc; }
  *out = R.release();
  return KMX_OK;
}

static int fetch_ctrl(kmx_merge_result* R, bool* overflow, bool* fallback = nullptr)
{
  kmx_ctx* ctx = R->ctx;
  *overflow = false;
  if (fallback) *fallback = false;
  const size_t nt = R->tasks.size();
  const u64* hc = reinterpret_cast<const u64*>(R->h_meta + R->o_ctrl0);      // written by k_ctrl_mirror behind the batch's last kernel
  KMX_HIP(ctx, hipEventSynchronize(R->ev_done));
  for (size_t t = 0; t < nt; t++) {
    TaskHost& H = R->tasks[t];
    const u64* ctrl = hc + t * 8;
    H.arena_rows = ctrl[0]; H.nsegs = ctrl[1]; H.rows = ctrl[3]; H.sparse_rows = ctrl[6]; H.row_keys = ctrl[7];
    if (ctrl[2] & (ERR_ROWS_OVERFLOW | ERR_SEGS_OVERFLOW)) *overflow = true;
    H.rows_overflow = (ctrl[2] & ERR_ROWS_OVERFLOW) != 0;
    H.handed_back = false;
    if ((ctrl[2] & ERR_FALLBACK) && fallback) { *fallback = true; H.handed_back = true; if (ctrl[2] & ERR_DIVERGENT) R->divergent = true; if (ctrl[2] & ERR_SLICES) R->slices_full = true; if (!(ctrl[2] & ERR_DENSE_CAP)) R->back_other = true; }
  }
  return KMX_OK;
}

extern "C" int kmx_result_wait(kmx_merge_result* R)
{
  if (!R) return KMX_E_INVAL;
  if (R->waited) return R->status;
  kmx_ctx* ctx = R->ctx;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  KMX_HIP(ctx, hipEventSynchronize(R->ev_done));
  if (R->is_bf) {
    if (R->is_bft && R->d_rem && !R->bft_two_walks) {
      // the single walk put more records aside in some tile than its scratch holds (lists with hardly a solid record): the batch
      // runs again with two walks -- every tile rewrites all of its bytes, the statistics restart
      const u64* hc = reinterpret_cast<const u64*>(R->h_meta + R->o_ctrl0);
      bool again = false;
      for (size_t t = 0; t < R->tasks.size(); t++) again = again || (hc[t * 8 + 2] & ERR_FALLBACK);
      if (again) {
        if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx merge] k_merge_bft: a tile overflowed the single walk's scratch: the batch runs again with two walks\n");
        R->bft_two_walks = true;
        for (auto& H : R->tasks) {
          KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_stats, 0, 8ull * 6 * H.N, ctx->stream));
          KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_ctrl, 0, 64, ctx->stream));
        }
        int rc2 = launch_batch(R, false);
        if (rc2 != KMX_OK) { R->waited = true; R->status = rc2; return rc2; }
        KMX_HIP(ctx, hipEventSynchronize(R->ev_done));
      }
    }
    for (auto& H : R->tasks) H.rows = R->is_bft ? H.t_cols : H.upper - H.lower + 1;
    R->waited = true; R->status = KMX_OK;
    return KMX_OK;
  }
  const ColsOps& CO = cols_ops((int)R->tasks[0].kw);
  bool overflow = false, fallback = false;
  int rc = fetch_ctrl(R, &overflow, &fallback);
  if (rc != KMX_OK) { R->waited = true; R->status = rc; return rc; }
  if (R->use_cols) {   // what the next batches' side stores of row keys' rows are sized from (k_cols_prep reports the count even when it gives up)
    double ratio = 0.0;
    for (size_t t = 0; t < R->tasks.size(); t++) { u32 longest = 0; for (u32 l : R->subs[t].len) longest = std::max(longest, l); if (longest) ratio = std::max(ratio, (double)R->tasks[t].row_keys / (double)longest); }
    ctx->keys_per_longest = ctx->keys_per_longest > 0.0 ? std::max(ratio, 0.75 * ctx->keys_per_longest + 0.25 * ratio) : ratio;
  }
  while (fallback) {
    // The column-blocked / pivot kernel handed some tasks back (lists that do not resemble each other): those tasks
    // -- only those -- run again with the next kernel down (cols -> pivot -> rows).  Bounds stay valid; their
    // statistics and row space restart.
    const bool from_cols = R->use_cols;
    const bool to_pivot = from_cols && R->can_pivot && !R->divergent;
    const uint2* all_items = reinterpret_cast<const uint2*>(R->h_meta + R->o_items);
    std::vector<uint2> redo;
    u32 n_back = 0;
    for (auto& H : R->tasks) n_back += H.handed_back ? 1u : 0u;
    for (u32 i = 0; i < R->n_items; i++) if (R->tasks[all_items[i].x].handed_back) redo.push_back(all_items[i]);
    if (getenv("KMX_TRACE") && from_cols) CO.dbg_dump();
    if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx merge] %s handed back %u of %zu tasks: re-run with %s\n", from_cols ? "k_merge_cols" : "k_merge_pivot",
                                     n_back, R->tasks.size(), to_pivot ? "k_merge_pivot" : "k_merge_rows");
    if (from_cols) {
      // full set-aside slices (an outlier sample: several times the cohort's k-mers in one list): not a cohort the kernel does not
      // suit -- the context's next batches run the build whose waves claim slice extensions; no pause
      const bool retry_ext = R->slices_full && !R->cols_ext && !R->divergent;
      if (retry_ext) ctx->cols_ext = true;
      // (tasks that came back ONLY because the side store of the row keys' rows was sized too small -- a host estimate the next batch
      //  corrects from this one's row keys, keys_per_longest above -- are no sign of a cohort the kernel does not suit: no back-off)
      if (R->cols_auto && n_back * 4 >= R->given && !retry_ext && R->back_other) {   // a cohort it does not suit: back off for the next batches
        // (doubling; at once to the longest pause when three quarters of the batch came back: a try costs a whole merge)
        ctx->cols_backoff = n_back * 4 >= 3 * R->given ? 64u : std::min(64u, std::max(1u, ctx->cols_backoff * 2));
        ctx->cols_skip = ctx->cols_backoff;
      }
      // (lists that share too few keys for k_merge_cols are beyond k_merge_pivot as well: pause both)
      if (R->divergent && R->auto_sel) { ctx->pivot_backoff = 64u; ctx->pivot_skip = 64u; }
      R->cols_auto = false;
    } else {
      // (also when it ran as the next kernel down from cols in a batch libkmx chose the kernels for)
      if ((R->pivot_auto || R->auto_sel) && n_back * 4 >= R->given) {
        ctx->pivot_backoff = n_back * 4 >= 3 * R->given ? 64u : std::min(64u, std::max(1u, ctx->pivot_backoff * 2));
        ctx->pivot_skip = ctx->pivot_backoff;
      }
      R->pivot_auto = false;
    }
    for (auto& H : R->tasks) {
      if (!H.handed_back) continue;
      KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_stats, 0, 8ull * 6 * H.N, ctx->stream));
      KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_ctrl, 0, 64, ctx->stream));
    }
    {
      uint2* stage = reinterpret_cast<uint2*>(R->h_meta + R->o_items);          // pinned; the full list is rebuilt below
      std::vector<uint2> keep(all_items, all_items + R->n_items);
      memcpy(stage, redo.data(), redo.size() * sizeof(uint2));
      KMX_HIP(ctx, hipMemcpyAsync(R->d_meta + R->o_items, stage, redo.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
      const bool was_pivot = R->use_pivot, was_cols = R->use_cols; const u32 was_items = R->n_items, was_grid = R->grid;
      R->use_cols = false; R->use_pivot = to_pivot; R->n_items = (u32)redo.size();
      R->grid = to_pivot ? std::min(R->n_items, (u32)ctx->n_cu)
                         : std::min(R->n_items, (u32)ctx->n_cu * (u32)(R->rows_small ? rows_s_wgs_per_cu((int)R->tasks[0].kw) : rows_wgs_per_cu((int)R->tasks[0].kw)));
      rc = launch_batch(R, false);
      if (rc == KMX_OK) { hipError_t he = hipEventSynchronize(R->ev_done); if (he != hipSuccess) rc = ctx->fail(KMX_E_HIP, hipGetErrorString(he)); }
      // restore the full item list (an arena-overflow retry below re-runs the whole batch, with k_merge_rows)
      memcpy(stage, keep.data(), keep.size() * sizeof(uint2));
      R->n_items = was_items; R->grid = was_grid;
      if (rc == KMX_OK) { hipError_t he = hipMemcpyAsync(R->d_meta + R->o_items, stage, keep.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream); if (he != hipSuccess) rc = ctx->fail(KMX_E_HIP, hipGetErrorString(he)); }
      if (rc == KMX_OK) { hipError_t he = hipEventRecord(R->ev_done, ctx->stream); if (he != hipSuccess) rc = ctx->fail(KMX_E_HIP, hipGetErrorString(he)); }   // (free must not hand the pinned image back under this copy)
      if (rc != KMX_OK) { R->waited = true; R->status = rc; return rc; }
      (void)was_pivot; (void)was_cols;
      for (auto& H : R->tasks) if (H.handed_back) H.kernel = to_pivot ? 1 : 0;
      R->given = n_back;
    }
    R->rerun_rows = true;
    bool ov2 = false;
    rc = fetch_ctrl(R, &ov2, &fallback);
    if (rc != KMX_OK) { R->waited = true; R->status = rc; return rc; }
    overflow = ov2;
    if (!to_pivot) { fallback = false; R->use_pivot = false; }
  }
  if (R->cols_auto) ctx->cols_backoff = 0;      // the column-blocked kernel completed this batch
  if (R->pivot_auto) ctx->pivot_backoff = 0;   // the pivot kernel completed this batch
  for (int attempt = 0; overflow; attempt++) {
    // the kernel kept counting: re-run with arenas / directories of the size it asked for (a second time when the
    // re-run -- possibly with another kernel -- asks for more still)
    TaskDev* td = reinterpret_cast<TaskDev*>(R->h_meta + R->o_tasks);
    for (size_t t = 0; t < R->tasks.size(); t++) {
      TaskHost& H = R->tasks[t];
      // (the file-order build of k_cols_sparse stops counting a task's rows once a group has overflowed the arena -- the later tickets
      //  of the task skip their groups --, so ctrl[0] may come back BELOW the arena's size: the retry then gets half as much again
      //  instead of the same arena a second time)
      if (H.rows_overflow && H.arena_rows <= H.out_cap_rows) H.arena_rows = H.out_cap_rows + H.out_cap_rows / 2 + 1;
      if (H.arena_rows > H.out_cap_rows) {
        ctx->dfree(H.d_out);
        // (what the kernel claimed, plus a chunk per range: the retry may run with another kernel -- tasks a cohort
        //  kernel handed back are re-run with k_merge_rows -- whose tiles leave other chunk tails unused)
        if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx merge] task %zu: the kernel asked for %llu rows, the arena had %llu (%u bytes a row, %u ranges)\n", t, (unsigned long long)H.arena_rows, (unsigned long long)H.out_cap_rows, H.row_bytes, H.c);
        H.out_cap_rows = H.arena_rows + (u64)(H.c + 1) * rows_chunk_rows(H.row_bytes); H.out_bytes = (size_t)(H.out_cap_rows * H.row_bytes);
        H.d_out = (u8*)ctx->dalloc(H.out_bytes);
        if (!H.d_out) { R->waited = true; R->status = ctx->fail(KMX_E_NOMEM, "output arena allocation failed (retry)"); return R->status; }
        td[t].out = H.d_out; td[t].out_cap_rows = H.out_cap_rows;
      }
      if (H.nsegs > H.seg_cap) {   // a larger directory in its own block
        H.seg_cap = (u32)std::min<u64>(0x7FFFFFFF, H.nsegs + 8ULL * H.c + 4096);
        if (H.d_segs_own) ctx->dfree(H.d_segs_own);
        H.d_segs_own = (Seg*)ctx->dalloc(sizeof(Seg) * (size_t)H.seg_cap);
        if (!H.d_segs_own) { R->waited = true; R->status = ctx->fail(KMX_E_NOMEM, "segment directory allocation failed (retry)"); return R->status; }
        H.d_segs = H.d_segs_own;
        td[t].segs = H.d_segs; td[t].seg_cap = H.seg_cap;
      }
    }
    // reset stats + ctrl, upload patched descriptors, run the merge again (bounds are still valid)
    for (auto& H : R->tasks) {
      KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_stats, 0, 8ull * 6 * H.N, ctx->stream));
      KMX_HIP(ctx, hipMemsetAsync(R->d_meta + H.o_ctrl, 0, 64, ctx->stream));
    }
    KMX_HIP(ctx, hipMemcpyAsync(R->d_meta + R->o_tasks, R->h_meta + R->o_tasks, sizeof(TaskDev) * R->tasks.size(),
                                hipMemcpyHostToDevice, ctx->stream));
    if (R->use_cols || (R->rerun_rows && R->use_pivot)) {   // tasks a kernel handed back must not go through it again
      R->use_pivot = false; R->use_cols = false;
      for (auto& H : R->tasks) H.kernel = 0;
      R->grid = std::min(R->n_items, (u32)ctx->n_cu * (u32)(R->rows_small ? rows_s_wgs_per_cu((int)R->tasks[0].kw) : rows_wgs_per_cu((int)R->tasks[0].kw)));
    }
    rc = launch_batch(R, false);
    if (rc != KMX_OK) { R->waited = true; R->status = rc; return rc; }
    KMX_HIP(ctx, hipEventSynchronize(R->ev_done));
    rc = fetch_ctrl(R, &overflow);
    if (rc == KMX_OK && overflow && attempt >= 2) rc = ctx->fail(KMX_E_HIP, "merge overflowed its exact-size arena (internal error)");
    if (rc != KMX_OK) { R->waited = true; R->status = rc; return rc; }
  }
  {   // what the next batches' arenas are sized from
    double ratio = 0.0;
    for (auto& H : R->tasks) { u32 longest = 0; for (u32 l : H.len) longest = std::max(longest, l); if (longest) ratio = std::max(ratio, (double)H.rows / (double)longest); }
    ctx->rows_per_longest = ctx->rows_per_longest > 0.0 ? std::max(ratio, 0.75 * ctx->rows_per_longest + 0.25 * ratio) : ratio;
  }
  for (auto& H : R->tasks) H.ordered = R->cols_ord && H.kernel == 2;
  R->waited = true; R->status = KMX_OK;
  if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx merge] batch of %zu tasks (%u-bit keys): %s\n", R->tasks.size(), 64u * R->tasks[0].kw, kmx_result_kernel(R));
  return KMX_OK;
}

extern "C" const char* kmx_result_kernel(const kmx_merge_result* R)
{
  if (!R) return "";
  if (R->is_bft) return "k_merge_bft";
  if (R->is_bf) return "k_merge_bf";
  size_t n[3] = {0, 0, 0};            // the kernel that produced most of the result
  for (auto& H : R->tasks) n[H.kernel]++;
  return n[2] >= n[1] && n[2] >= n[0] && n[2] ? "k_merge_cols" : n[1] >= n[0] && n[1] ? "k_merge_pivot" : "k_merge_rows";
}

extern "C" double kmx_result_kernel_ms(kmx_merge_result* R)
{
  if (!R || !R->ev0 || kmx_result_wait(R) != KMX_OK) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, R->ev0, R->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}
extern "C" int kmx_result_kernel_parts_ms(kmx_merge_result* R, double* first_ms, double* second_ms)
{
  if (!R || !first_ms || !second_ms) return KMX_E_INVAL;
  *first_ms = *second_ms = -1.0;
  if (!R->ev0 || !R->ev_mid || kmx_result_wait(R) != KMX_OK || R->rerun_rows) return KMX_OK;      // (not the column-blocked pair, or tasks were run again behind it)
  float a = -1.f, b = -1.f;
  if (hipEventElapsedTime(&a, R->ev0, R->ev_mid) != hipSuccess || hipEventElapsedTime(&b, R->ev_mid, R->ev1) != hipSuccess) { (void)hipGetLastError(); return KMX_OK; }
  *first_ms = a; *second_ms = b;
  return KMX_OK;
}
extern "C" uint64_t kmx_result_rows(const kmx_merge_result* R, uint32_t t) { return (R && t < R->tasks.size()) ? R->tasks[t].rows : 0; }
extern "C" uint64_t kmx_result_sparse_rows(const kmx_merge_result* R, uint32_t t) { return (R && t < R->tasks.size() && R->tasks[t].kernel == 2) ? R->tasks[t].sparse_rows : 0; }
extern "C" uint64_t kmx_result_row_bytes(const kmx_merge_result* R, uint32_t t)
{ return (R && t < R->tasks.size()) ? (R->is_bft ? R->tasks[t].t_rows >> 3 : R->tasks[t].row_bytes) : 0; }
extern "C" uint64_t kmx_result_body_bytes(const kmx_merge_result* R, uint32_t t)
{ return (R && t < R->tasks.size()) ? R->tasks[t].rows * kmx_result_row_bytes(R, t) : 0; }
extern "C" uint64_t kmx_result_algo_bytes(const kmx_merge_result* R, uint32_t t)
{
  if (!R || t >= R->tasks.size()) return 0;
  const TaskHost& H = R->tasks[t];
  return H.total_recs * (H.kw * 8 + 4) + H.rows * kmx_result_row_bytes(R, t);
}
extern "C" double kmx_result_transpose_ms(kmx_merge_result* R)
{
  if (!R || !R->ev2 || kmx_result_wait(R) != KMX_OK) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, R->ev1, R->ev2) != hipSuccess) return -1.0;
  return (double)ms;
}
// COUNT/PA rows of k_merge_rows / k_merge_pivot lie in arena segments (one per chunk a range claimed); the file order is the
// directory sorted by (range, sequence).  One workgroup per segment copies its rows to their place in the body.
struct SegCopy { u64 src, dst, bytes; };
__global__ __launch_bounds__(256) void k_segs_gather(const SegCopy* __restrict__ sc, const u8* __restrict__ arena, u8* __restrict__ body)
{
  const SegCopy c = sc[blockIdx.x];
  const u8* s = arena + c.src; u8* d = body + c.dst;
  // rows are 4-byte aligned at both ends (row_bytes is a multiple of 4 for COUNT rows; PA rows may not be: bytes then)
  if ((((uintptr_t)s | (uintptr_t)d | c.bytes) & 3u) == 0) {
    const u32* s4 = reinterpret_cast<const u32*>(s); u32* d4 = reinterpret_cast<u32*>(d);
    for (u64 i = threadIdx.x; i < c.bytes / 4; i += 256) d4[i] = s4[i];
  } else for (u64 i = threadIdx.x; i < c.bytes; i += 256) d[i] = s[i];
}

// the body of a COUNT/PA task in file order, on the device: d_out itself when the rows already lie that way (one segment from
// row 0: k_merge_cols without rows outside the row keys), else assembled once into d_body (k_cols_gather / k_segs_gather)
static int assemble_body(kmx_merge_result* R, uint32_t t, bool async = false)
{
  kmx_ctx* ctx = R->ctx;
  TaskHost& H = R->tasks[t];
  if (H.ev_body) {      // queued earlier (kmx_result_prepare_body): wait for it
    if (async) return KMX_OK;
    KMX_HIP(ctx, hipEventSynchronize(H.ev_body));
    (void)hipEventDestroy(H.ev_body); H.ev_body = nullptr;
    ctx->dfree(H.d_body_tmp); H.d_body_tmp = nullptr;
    H.body_ready = true;
    return KMX_OK;
  }
  if (H.body_ready) return KMX_OK;
  // (the assembly runs on the context's second stream -- not the copy stream, where the caller's device-to-host pieces of the
  //  PREVIOUS task's body are queued: queued ahead of time it hides behind them)
  hipStream_t as = async ? ctx->aux : ctx->copy;
  auto finish = [&](u8* d_body, void* tmp) -> int {
    if (!async) { hipError_t e = hipStreamSynchronize(as); ctx->dfree(tmp); if (e != hipSuccess) { ctx->dfree(d_body); return ctx->fail(KMX_E_HIP, std::string("body assembly: ") + hipGetErrorString(e)); } H.d_body = d_body; H.body_ready = true; return KMX_OK; }
    hipError_t e = hipEventCreateWithFlags(&H.ev_body, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(H.ev_body, as);
    if (e != hipSuccess) { (void)hipStreamSynchronize(as); ctx->dfree(tmp); ctx->dfree(d_body); if (H.ev_body) { (void)hipEventDestroy(H.ev_body); H.ev_body = nullptr; } return ctx->fail(KMX_E_HIP, std::string("body assembly: ") + hipGetErrorString(e)); }
    H.d_body = d_body; H.d_body_tmp = tmp;
    return KMX_OK;
  };
  const u64 body = H.rows * H.row_bytes;
  if (body == 0) { H.body_ready = true; return KMX_OK; }
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  const ColsOps& CO = cols_ops((int)H.kw);
  if (H.ordered) { H.body_ready = true; return KMX_OK; }      // (the kernels wrote the rows at their final place: d_out is the body)
  if (H.kernel == 2 && H.sparse_rows) {
    const u32 ng = CO.groups(H.slots_cap);
    u8* d_body = (u8*)ctx->dalloc(body);
    u64* d_goff = (u64*)ctx->dalloc((size_t)ng * 8);
    if (!d_body || !d_goff) { ctx->dfree(d_body); ctx->dfree(d_goff); return ctx->fail(KMX_E_NOMEM, "body assembly allocation failed"); }
    const TaskDev* d_tasks = reinterpret_cast<const TaskDev*>(R->d_meta + R->o_tasks);
    const ColsDev* d_cols = reinterpret_cast<const ColsDev*>(R->d_meta + R->o_cols);
    hipError_t e = CO.offsets(d_cols, t, d_goff, as);
    if (e == hipSuccess) e = CO.gather(d_tasks, d_cols, t, ng, d_goff, d_body, as);
    if (e != hipSuccess) { (void)hipStreamSynchronize(as); ctx->dfree(d_goff); ctx->dfree(d_body); return ctx->fail(KMX_E_HIP, std::string("body assembly: ") + hipGetErrorString(e)); }
    return finish(d_body, d_goff);
  }
  std::vector<Seg> segs(H.nsegs);
  KMX_HIP(ctx, hipMemcpyAsync(segs.data(), H.d_segs, sizeof(Seg) * H.nsegs, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  std::sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.range != b.range ? a.range < b.range : a.seq < b.seq; });
  const u64 arena = H.arena_rows * H.row_bytes;
  std::vector<SegCopy> sc; sc.reserve(segs.size());
  u64 done = 0; bool in_place = true;
  for (const Seg& g : segs) {
    const u64 nb = (u64)g.nrows * H.row_bytes;
    if (g.row_off * H.row_bytes + nb > arena || done + nb > body) return ctx->fail(KMX_E_HIP, "corrupt segment directory");
    if (nb) { in_place = in_place && g.row_off * H.row_bytes == done; sc.push_back({g.row_off * H.row_bytes, done, nb}); }
    done += nb;
  }
  if (done != body) return ctx->fail(KMX_E_HIP, "segment directory does not cover the arena");
  if (in_place) { H.body_ready = true; return KMX_OK; }      // (d_out is the body)
  u8* d_body = (u8*)ctx->dalloc(body);
  SegCopy* d_sc = (SegCopy*)ctx->dalloc(sc.size() * sizeof(SegCopy));
  if (!d_body || !d_sc) { ctx->dfree(d_body); ctx->dfree(d_sc); return ctx->fail(KMX_E_NOMEM, "body assembly allocation failed"); }
  hipError_t e = hipMemcpyAsync(d_sc, sc.data(), sc.size() * sizeof(SegCopy), hipMemcpyHostToDevice, ctx->copy);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy);      // (sc is this frame's)
  if (e == hipSuccess) { hipLaunchKernelGGL(k_segs_gather, dim3((unsigned)sc.size()), dim3(256), 0, as, d_sc, H.d_out, d_body); e = hipGetLastError(); }
  if (e != hipSuccess) { (void)hipStreamSynchronize(as); ctx->dfree(d_sc); ctx->dfree(d_body); return ctx->fail(KMX_E_HIP, std::string("body assembly: ") + hipGetErrorString(e)); }
  return finish(d_body, d_sc);
}

// queue the assembly of a COUNT/PA body without waiting for it (kmx_result_body_dev / kmx_result_copy_body then do)
extern "C" int kmx_result_prepare_body(kmx_merge_result* R, uint32_t t)
{
  if (!R || t >= R->tasks.size()) return KMX_E_INVAL;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK || R->is_bf) return rc;
  return assemble_body(R, t, true);
}

// ---- COUNT/PA rows where the kernels left them + their order: what a consumer takes that puts the rows in place itself (a file
//      writer: pwrite of row d at d * row_bytes).  No device-side pass over the rows, no second copy of the matrix in HBM. ----
extern "C" int kmx_result_arena(kmx_merge_result* R, uint32_t t, const void** dev_arena, uint64_t* arena_rows)
{
  if (!R || t >= R->tasks.size() || !dev_arena || !arena_rows) return KMX_E_INVAL;
  kmx_ctx* ctx = R->ctx;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK) return rc;
  if (R->is_bf) return ctx->fail(KMX_E_UNSUPPORTED, "kmx_result_arena: Bloom results are dense (kmx_result_body_dev)");
  *dev_arena = R->tasks[t].d_out; *arena_rows = R->tasks[t].arena_rows;
  return KMX_OK;
}
extern "C" int kmx_result_copy_order(kmx_merge_result* R, uint32_t t, uint32_t* host_order)
{
  if (!R || t >= R->tasks.size() || !host_order) return KMX_E_INVAL;
  kmx_ctx* ctx = R->ctx;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK) return rc;
  if (R->is_bf) return ctx->fail(KMX_E_UNSUPPORTED, "kmx_result_copy_order: Bloom results are dense (kmx_result_body_dev)");
  TaskHost& H = R->tasks[t];
  if (H.rows == 0) return KMX_OK;
  if (H.rows > 0xFFFFFFFFULL || H.arena_rows > 0xFFFFFFFFULL) return ctx->fail(KMX_E_UNSUPPORTED, "more than 2^32 rows in one task");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  const ColsOps& CO = cols_ops((int)H.kw);
  if (H.ordered) { for (u64 r = 0; r < H.rows; r++) host_order[r] = (u32)r; return KMX_OK; }
  if (H.kernel == 2 && H.sparse_rows) {
    const u32 ng = CO.groups(H.slots_cap);
    u32* d_order = (u32*)ctx->dalloc((size_t)H.rows * 4);
    u64* d_goff = (u64*)ctx->dalloc((size_t)ng * 8);
    struct Rel { kmx_ctx* c; void* a; void* b; ~Rel() { c->dfree(a); c->dfree(b); } } rel{ctx, d_order, d_goff};
    if (!d_order || !d_goff) return ctx->fail(KMX_E_NOMEM, "row order allocation failed");
    const TaskDev* d_tasks = reinterpret_cast<const TaskDev*>(R->d_meta + R->o_tasks);
    const ColsDev* d_cols = reinterpret_cast<const ColsDev*>(R->d_meta + R->o_cols);
    KMX_HIP(ctx, CO.offsets(d_cols, t, d_goff, ctx->copy));
    KMX_HIP(ctx, CO.order(d_tasks, d_cols, t, ng, d_goff, d_order, ctx->copy));
    KMX_HIP(ctx, hipMemcpyAsync(host_order, d_order, (size_t)H.rows * 4, hipMemcpyDeviceToHost, ctx->copy));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
    return KMX_OK;
  }
  std::vector<Seg> segs(H.nsegs);
  KMX_HIP(ctx, hipMemcpyAsync(segs.data(), H.d_segs, sizeof(Seg) * H.nsegs, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  std::sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.range != b.range ? a.range < b.range : a.seq < b.seq; });
  u64 done = 0;
  for (const Seg& g : segs) {
    if (g.row_off + g.nrows > H.arena_rows || done + g.nrows > H.rows) return ctx->fail(KMX_E_HIP, "corrupt segment directory");
    for (u32 r = 0; r < g.nrows; r++) host_order[done + r] = (u32)(g.row_off + r);
    done += g.nrows;
  }
  if (done != H.rows) return ctx->fail(KMX_E_HIP, "segment directory does not cover the arena");
  return KMX_OK;
}

extern "C" const void* kmx_result_body_dev(kmx_merge_result* R, uint32_t t)
{
  if (!R || t >= R->tasks.size() || kmx_result_wait(R) != KMX_OK) return nullptr;
  if (R->is_bf) return R->tasks[t].d_out;
  if (assemble_body(R, t) != KMX_OK) return nullptr;
  return R->tasks[t].d_body ? R->tasks[t].d_body : R->tasks[t].d_out;
}

extern "C" int kmx_result_copy_body(kmx_merge_result* R, uint32_t t, void* dst, uint64_t dst_bytes)
{
  if (!R || t >= R->tasks.size()) return KMX_E_INVAL;
  kmx_ctx* ctx = R->ctx;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK) return rc;
  TaskHost& H = R->tasks[t];
  const u64 body = kmx_result_body_bytes(R, t);
  if (dst_bytes < body) return ctx->fail(KMX_E_INVAL, "destination too small");
  if (body == 0) return KMX_OK;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (R->is_bf) {
    KMX_HIP(ctx, hipMemcpyAsync(dst, H.d_out, body, hipMemcpyDeviceToHost, ctx->copy));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
    return KMX_OK;
  }
  // COUNT/PA: the body is put in file order on the device (assemble_body), then comes back in one copy -- straight into the
  // caller's buffer when that is page-locked (kmx_alloc_pinned), through pinned staging otherwise
  if ((rc = assemble_body(R, t)) != KMX_OK) return rc;
  const u8* src = H.d_body ? H.d_body : H.d_out;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();
  if (pinned) {
    KMX_HIP(ctx, hipMemcpyAsync(dst, src, body, hipMemcpyDeviceToHost, ctx->copy));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
    return KMX_OK;
  }
  u8* stage = (u8*)ctx->halloc(body);
  if (!stage) return ctx->fail(KMX_E_NOMEM, "host staging allocation failed");
  struct Rel { kmx_ctx* c; u8* p; ~Rel() { c->hfree(p); } } rel{ctx, stage};
  KMX_HIP(ctx, hipMemcpyAsync(stage, src, body, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  memcpy(dst, stage, body);
  return KMX_OK;
}

extern "C" int kmx_result_copy_body_dev(kmx_merge_result* R, uint32_t t, void* dev_dst, uint64_t dst_bytes)
{
  if (!R || t >= R->tasks.size()) return KMX_E_INVAL;
  kmx_ctx* ctx = R->ctx;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK) return rc;
  const u64 body = kmx_result_body_bytes(R, t);
  if (dst_bytes < body) return ctx->fail(KMX_E_INVAL, "destination too small");
  if (!body) return KMX_OK;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (!R->is_bf && (rc = assemble_body(R, t)) != KMX_OK) return rc;
  KMX_HIP(ctx, hipMemcpyAsync(dev_dst, R->tasks[t].d_body ? R->tasks[t].d_body : R->tasks[t].d_out, body, hipMemcpyDeviceToDevice, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  return KMX_OK;
}

extern "C" int kmx_result_copy_stats(kmx_merge_result* R, uint32_t t, uint64_t* st)
{
  if (!R || t >= R->tasks.size() || !st) return KMX_E_INVAL;
  kmx_ctx* ctx = R->ctx;
  int rc = kmx_result_wait(R);
  if (rc != KMX_OK) return rc;
  TaskHost& H = R->tasks[t];
  const u32 N = H.N;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  KMX_HIP(ctx, hipMemcpyAsync(st, R->d_meta + H.o_stats, 8ull * 6 * N, hipMemcpyDeviceToHost, ctx->copy));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->copy));
  // COUNT/PA kernel fills NON_SOLID (0), RESCUED (1), TOTAL_WO (4) and the rescued total (5); the BF
  // kernel fills UNIQUE_WO (2) instead of NON_SOLID.  Derive the rest exactly as MergeStatistics does
  // (merge.hpp:65-70): every input record is either solid or non-solid.
  for (u32 i = 0; i < N; i++) {
    const u64 rd = st[1 * (u64)N + i], two = st[4 * (u64)N + i], twr = st[5 * (u64)N + i];
    u64 ns, uwo;
    if (R->is_bf) { uwo = st[2 * (u64)N + i]; ns = (u64)H.len[i] - uwo; }
    else { ns = st[0 * (u64)N + i]; uwo = (u64)H.len[i] - ns; }
    st[0 * (u64)N + i] = ns;
    st[2 * (u64)N + i] = uwo;
    st[3 * (u64)N + i] = uwo + rd;
    st[5 * (u64)N + i] = two + twr;
  }
  return KMX_OK;
}

#ifdef KMX_PHASE_PROF
namespace kmx { void rows_phase_prof_dump(); void pivot_phase_prof_dump(); void bft_phase_prof_dump(); }
#endif
extern "C" void kmx_result_free(kmx_merge_result* R)
{
  if (!R) return;
  kmx_ctx* ctx = R->ctx;
  (void)hipSetDevice(ctx->device);
  if (R->ev_done) (void)hipEventSynchronize(R->ev_done);      // (this result's work, not the stream: later batches are queued behind it)
  else (void)hipStreamSynchronize(ctx->stream);
#ifdef KMX_PHASE_PROF
  const ColsOps& CO = cols_ops((int)R->tasks[0].kw);
  if (R->is_bft) kmx::bft_phase_prof_dump();
  if (!R->is_bf) { if (R->use_cols) { if (CO.phase_prof_dump) CO.phase_prof_dump(); } else if (R->use_pivot) kmx::pivot_phase_prof_dump(); else kmx::rows_phase_prof_dump(); }
#endif
  for (auto& H : R->tasks) {
    if (H.ev_body) { (void)hipEventSynchronize(H.ev_body); (void)hipEventDestroy(H.ev_body); }
    ctx->dfree(H.d_out); ctx->dfree(H.d_segs_own); ctx->dfree(H.d_ov); ctx->dfree(H.d_img); ctx->dfree(H.d_rowrec); ctx->dfree(H.d_body); ctx->dfree(H.d_body_tmp); ctx->dfree(H.d_dense); ctx->dfree(H.d_narrow);
  }
  for (auto& Q : R->subs) ctx->dfree(Q.d_out);
  ctx->dfree(R->d_meta);
  ctx->hfree(R->h_meta);
  ctx->dfree(R->d_in);
  ctx->dfree(R->d_rem);
  if (R->ev_in) (void)hipEventDestroy(R->ev_in);
  if (R->ev0) { (void)hipEventDestroy(R->ev0); (void)hipEventDestroy(R->ev1); }
  if (R->ev_mid) (void)hipEventDestroy(R->ev_mid);
  if (R->ev2) (void)hipEventDestroy(R->ev2);
  if (R->ev_pre) (void)hipEventDestroy(R->ev_pre);
  if (R->ev_up) (void)hipEventDestroy(R->ev_up);
  if (R->ev_done) (void)hipEventDestroy(R->ev_done);
  delete R;
}

extern "C" int kmx_merge_host(kmx_ctx* ctx, const kmx_merge_task* tasks, uint32_t n_tasks, kmx_merge_result** out)
{
  if (!ctx) return KMX_E_INVAL;
  if (!tasks || !n_tasks || !out) return ctx->fail(KMX_E_INVAL, "kmx_merge_host: null argument");
  *out = nullptr;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  // the lists of a batch usually lie back to back in one (pinned) buffer: then the whole batch is one copy
  uintptr_t lo = ~(uintptr_t)0, hi = 0; u64 sum = 0, nl = 0;
  for (u32 t = 0; t < n_tasks; t++) {
    const kmx_merge_task& K = tasks[t];
    if (K.key_words < 1 || K.key_words > 4) return ctx->fail(KMX_E_INVAL, "key_words must be 1 ... 4");
    if (!K.lists && K.n_lists) return ctx->fail(KMX_E_INVAL, "task without lists");
    const size_t rb = K.key_words * 8 + 4;
    for (u32 i = 0; i < K.n_lists; i++) {
      const u64 nb = K.lists[i].n * rb;
      if (!nb) continue;
      if (!K.lists[i].recs) return ctx->fail(KMX_E_INVAL, "null record pointer");
      const uintptr_t a = (uintptr_t)K.lists[i].recs;
      if (a & 3u) return ctx->fail(KMX_E_INVAL, "record pointers must be 4-byte aligned");
      if (K.list_on_device && K.list_on_device[i]) continue;      // (resident already: a kmx_store list)
      lo = std::min(lo, a); hi = std::max(hi, (uintptr_t)(a + nb)); sum += nb; nl++;
    }
  }
  const bool one_span = nl && (u64)(hi - lo) <= sum + sum / 4 + 4096;
  u64 total = 0;
  if (one_span) total = hi - lo;
  else for (u32 t = 0; t < n_tasks; t++) for (u32 i = 0; i < tasks[t].n_lists; i++)
    if (!(tasks[t].list_on_device && tasks[t].list_on_device[i])) total += align_up(tasks[t].lists[i].n * (tasks[t].key_words * 8 + 4), 256);
  u8* d_in = (u8*)ctx->dalloc(total ? total : 256);
  if (!d_in) return ctx->fail(KMX_E_NOMEM, "input upload allocation failed");
  std::vector<kmx_merge_task> dt(tasks, tasks + n_tasks);
  std::vector<std::vector<kmx_list>> dl(n_tasks);
  hipError_t e = hipSuccess;
  if (one_span) e = hipMemcpyAsync(d_in, (const void*)lo, total, hipMemcpyHostToDevice, ctx->up);
  u64 off = 0;
  for (u32 t = 0; t < n_tasks && e == hipSuccess; t++) {
    const kmx_merge_task& K = tasks[t];
    const size_t rb = K.key_words * 8 + 4;
    dl[t].resize(K.n_lists);
    for (u32 i = 0; i < K.n_lists && e == hipSuccess; i++) {
      const u64 nb = K.lists[i].n * rb;
      dl[t][i].n = K.lists[i].n;
      if (K.list_on_device && K.list_on_device[i]) dl[t][i].recs = K.lists[i].recs;
      else if (one_span) dl[t][i].recs = nb ? d_in + ((uintptr_t)K.lists[i].recs - lo) : d_in;
      else {
        dl[t][i].recs = d_in + off;
        if (nb) e = hipMemcpyAsync(d_in + off, K.lists[i].recs, nb, hipMemcpyHostToDevice, ctx->up);
        off += align_up(nb, 256);
      }
    }
    dt[t].lists = dl[t].data(); dt[t].list_on_device = nullptr;
  }
  hipEvent_t ev = nullptr;
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(ev, ctx->up);
  // everything this batch queues on the merge streams comes behind its upload (the previous batch's kernels are
  // already queued in front: the copy runs beside them)
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->aux, ev, 0);
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(ctx->up); ctx->dfree(d_in); if (ev) (void)hipEventDestroy(ev);
    return ctx->fail(KMX_E_HIP, std::string("upload: ") + hipGetErrorString(e));
  }
  kmx_merge_result* R = nullptr;
  const int rc = kmx_merge_dev(ctx, dt.data(), n_tasks, &R);
  if (rc != KMX_OK) { (void)hipStreamSynchronize(ctx->up); ctx->dfree(d_in); (void)hipEventDestroy(ev); return rc; }
  R->d_in = d_in; R->ev_in = ev;
  *out = R;
  return KMX_OK;
}

extern "C" int kmx_merge(kmx_ctx* ctx, const kmx_merge_task* task, void** body, uint64_t* body_bytes, uint64_t* rows, uint64_t* stats)
{
  if (!ctx) return KMX_E_INVAL;
  if (!task || !body || !body_bytes || !rows) return ctx->fail(KMX_E_INVAL, "kmx_merge: null argument");
  *body = nullptr; *body_bytes = 0; *rows = 0;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (task->key_words < 1 || task->key_words > 4) return ctx->fail(KMX_E_INVAL, "key_words must be 1 ... 4");
  const size_t rb = task->key_words * 8 + 4;
  std::vector<kmx_list> dl(task->n_lists);
  size_t total = 0;
  for (u32 i = 0; i < task->n_lists; i++) total += align_up(task->lists[i].n * rb, 256);
  u8* d_in = (u8*)ctx->dalloc(total);
  if (!d_in) return ctx->fail(KMX_E_NOMEM, "input upload allocation failed");
  size_t off = 0;
  for (u32 i = 0; i < task->n_lists; i++) {
    dl[i].recs = d_in + off; dl[i].n = task->lists[i].n;
    if (task->lists[i].n) {
      hipError_t e = hipMemcpyAsync(d_in + off, task->lists[i].recs, task->lists[i].n * rb, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) { ctx->dfree(d_in); return ctx->fail(KMX_E_HIP, std::string("upload: ") + hipGetErrorString(e)); }
    }
    off += align_up(task->lists[i].n * rb, 256);
  }
  { hipError_t e = hipStreamSynchronize(ctx->stream);      // (the batch's preparation runs on the second stream)
    if (e != hipSuccess) { ctx->dfree(d_in); return ctx->fail(KMX_E_HIP, std::string("upload: ") + hipGetErrorString(e)); } }
  kmx_merge_task dt = *task;
  dt.lists = dl.data();
  kmx_merge_result* R = nullptr;
  int rc = kmx_merge_dev(ctx, &dt, 1, &R);
  if (rc == KMX_OK) rc = kmx_result_wait(R);
  if (rc == KMX_OK) {
    const u64 nb = kmx_result_body_bytes(R, 0);
    void* b = malloc(nb ? nb : 1);
    if (!b) rc = ctx->fail(KMX_E_NOMEM, "host body allocation failed");
    else {
      rc = kmx_result_copy_body(R, 0, b, nb);
      if (rc == KMX_OK && stats) rc = kmx_result_copy_stats(R, 0, stats);
      if (rc == KMX_OK) { *body = b; *body_bytes = nb; *rows = kmx_result_rows(R, 0); }
      else free(b);
    }
  }
  if (R) kmx_result_free(R);
  else (void)hipStreamSynchronize(ctx->stream);
  ctx->dfree(d_in);
  return rc;
}
