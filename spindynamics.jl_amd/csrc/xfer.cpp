// Host <-> device transfers of the host-pointer entry points (the vectors of src/PublicAPI.jl:50-88 are host arrays).
//
// A plain hipMemcpy on pageable memory is staged by the runtime through its own pinned bounce buffers by ONE thread, and a
// fresh destination array (numpy.empty, Julia's similar) is page-faulted by that same thread while the copy runs: measured
// round 2 at L=32 (9.6 GB per vector) ~0.35 s in, 0.7-1.1 s out.  Here large transfers go through a ring of pinned chunks
// owned by the context: the PCIe DMA of chunk k runs while a team of host threads copies chunk k+1 (in) / k-1 (out) between
// the caller's array and the pinned chunk -- the page faults of a fresh destination are then spread over the team.
//
//   SD_XFER = auto (default) | plain | staged | register      SD_XFER_THREADS (default: min(16, allowed cores))
//   SD_XFER_CHUNK_MB (default 32)                             SD_XFER_MIN_MB (default 16: below it, plain)
// "register" pins the caller's pages for the duration of the call (hipHostRegister) and copies straight from / to them; it is
// kept as a measurable alternative (pinning 9.6 GB costs more than it saves on this host, profiles/xfer_bench.py).
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "sd_internal.hpp"

struct sd_xfer_team {
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  char *dst = nullptr;
  const char *src = nullptr;
  size_t bytes = 0;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false;
  int nthreads = 1;

  explicit sd_xfer_team(int n) : nthreads(n) {
    try {
      for (int t = 1; t < n; ++t) workers.emplace_back([this, t] { loop(t); });
    } catch (...) {            // a destructor is not run for a half-built object: stop and join what did start, then hand the failure on
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv_work.notify_all();
      for (auto &w : workers) w.join();
      throw;
    }
  }
  ~sd_xfer_team() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_work.notify_all();
    for (auto &w : workers) w.join();
  }
  static void slice(char *d, const char *s, size_t bytes, int t, int n) {
    // page-aligned slices so that two threads never fault the same page
    const size_t per = ((bytes / (size_t)n) + 4095) & ~(size_t)4095;
    const size_t lo = std::min(bytes, per * (size_t)t), hi = std::min(bytes, lo + per);
    if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
  }
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv_work.wait(lk, [&] { return stop || generation != seen; });
      if (stop) return;
      seen = generation;
      char *d = dst; const char *s = src; const size_t b = bytes;
      lk.unlock();
      slice(d, s, b, t, nthreads);
      lk.lock();
      if (--pending == 0) cv_done.notify_one();
    }
  }
  // dst[0..bytes) <- src[0..bytes) by the whole team (the caller is member 0); returns when done
  void copy(void *d, const void *s, size_t b) {
    if (nthreads == 1 || b < ((size_t)1 << 20)) { std::memcpy(d, s, b); return; }
    {
      std::lock_guard<std::mutex> lk(mu);
      dst = (char *)d; src = (const char *)s; bytes = b;
      pending = nthreads - 1;
      ++generation;
    }
    cv_work.notify_all();
    slice((char *)d, (const char *)s, b, 0, nthreads);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};

namespace {

constexpr int NBUF = 4;

int allowed_cores() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return n; }
  const unsigned hc = std::thread::hardware_concurrency();
  return hc ? (int)hc : 1;
}

enum Mode { AUTO, PLAIN, STAGED, REGISTER };
Mode mode_from_env() {
  const char *e = getenv("SD_XFER");
  if (!e || !*e || !strcmp(e, "auto")) return AUTO;
  if (!strcmp(e, "plain")) return PLAIN;
  if (!strcmp(e, "staged")) return STAGED;
  if (!strcmp(e, "register")) return REGISTER;
  return AUTO;
}
size_t env_mb(const char *name, size_t dflt) {
  const char *e = getenv(name);
  const long v = e ? atol(e) : 0;
  return (v > 0 ? (size_t)v : dflt) << 20;
}

int ensure_ring(sd_ctx *ctx, size_t chunk) {
  if (ctx->xfer_chunk == chunk && ctx->xfer_buf[0]) return SD_OK;
  sd_xfer_release(ctx);
  for (int k = 0; k < NBUF; ++k) {
    SD_HIP(ctx, hipHostMalloc(&ctx->xfer_buf[k], chunk, hipHostMallocDefault));
    SD_HIP(ctx, hipEventCreateWithFlags(&ctx->xfer_ev[k], hipEventDisableTiming));
  }
  ctx->xfer_chunk = chunk;
  if (!ctx->xfer_team) {
    int nt = getenv("SD_XFER_THREADS") ? atoi(getenv("SD_XFER_THREADS")) : std::min(16, allowed_cores());
    if (nt < 1) nt = 1;
    // thread creation can fail (process / cgroup limits): never let that unwind through the C ABI -- copy with the calling thread alone
    try {
      ctx->xfer_team = new sd_xfer_team(nt);
    } catch (...) {
      try { ctx->xfer_team = new sd_xfer_team(1); } catch (...) { ctx->xfer_team = nullptr; return sd_set_err(ctx, SD_ENOMEM, "host copy team"); }
    }
  }
  return SD_OK;
}

int plain(sd_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  SD_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}

int registered(sd_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  void *host = kind == hipMemcpyHostToDevice ? const_cast<void *>(src) : dst;
  if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return -1;                               // caller falls back to the staged path
  }
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipHostUnregister(host);
  if (e != hipSuccess) return sd_set_err(ctx, SD_EHIP, std::string("registered copy: ") + hipGetErrorString(e));
  return SD_OK;
}

int staged_h2d(sd_ctx *ctx, char *dev, const char *host, size_t bytes, size_t chunk) {
  int rc = ensure_ring(ctx, chunk);
  if (rc) return rc;
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  for (size_t k = 0; k < nchunks; ++k) {
    const int b = (int)(k % NBUF);
    const size_t lo = k * chunk, len = std::min(chunk, bytes - lo);
    if (k >= NBUF) SD_HIP(ctx, hipEventSynchronize(ctx->xfer_ev[b]));          // the DMA that last read this buffer is done
    ctx->xfer_team->copy(ctx->xfer_buf[b], host + lo, len);
    SD_HIP(ctx, hipMemcpyAsync(dev + lo, ctx->xfer_buf[b], len, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipEventRecord(ctx->xfer_ev[b], ctx->stream));
  }
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}

int staged_d2h(sd_ctx *ctx, char *host, const char *dev, size_t bytes, size_t chunk) {
  int rc = ensure_ring(ctx, chunk);
  if (rc) return rc;
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  // DMA runs NBUF-1 chunks ahead of the team's copy out of the pinned ring
  auto issue = [&](size_t k) -> int {
    const int b = (int)(k % NBUF);
    const size_t lo = k * chunk, len = std::min(chunk, bytes - lo);
    SD_HIP(ctx, hipMemcpyAsync(ctx->xfer_buf[b], dev + lo, len, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipEventRecord(ctx->xfer_ev[b], ctx->stream));
    return SD_OK;
  };
  size_t issued = 0;
  for (; issued < nchunks && issued < (size_t)(NBUF - 1); ++issued) if ((rc = issue(issued))) return rc;
  for (size_t k = 0; k < nchunks; ++k) {
    const int b = (int)(k % NBUF);
    const size_t lo = k * chunk, len = std::min(chunk, bytes - lo);
    SD_HIP(ctx, hipEventSynchronize(ctx->xfer_ev[b]));
    if (issued < nchunks) { if ((rc = issue(issued))) return rc; ++issued; }    // its buffer, (k-1) % NBUF, was emptied last turn
    ctx->xfer_team->copy(host + lo, ctx->xfer_buf[b], len);
  }
  return SD_OK;
}

int xfer(sd_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return SD_OK;
  const Mode mode = mode_from_env();            // read per call (three getenv calls beside a >= 16 MB copy): tests switch modes in-process
  const size_t min_bytes = env_mb("SD_XFER_MIN_MB", 16), chunk = env_mb("SD_XFER_CHUNK_MB", 32);
  if (mode == PLAIN || (mode == AUTO && bytes < min_bytes)) return plain(ctx, dst, src, bytes, kind);
  if (mode == REGISTER) {
    const int rc = registered(ctx, dst, src, bytes, kind);
    if (rc >= 0) return rc;
  }
  // no pinned memory to be had (locked-memory limit of the process): one plain copy instead of a failed call
  if (ensure_ring(ctx, chunk) != SD_OK) {
    (void)hipGetLastError();
    sd_xfer_release(ctx);
    ctx->err.clear();
    return plain(ctx, dst, src, bytes, kind);
  }
  // prior work on the stream (the producer of a device source, the consumer of a device destination) is ordered before the
  // DMA because the DMA is queued on the same stream
  return kind == hipMemcpyHostToDevice ? staged_h2d(ctx, (char *)dst, (const char *)src, bytes, chunk)
                                       : staged_d2h(ctx, (char *)dst, (const char *)src, bytes, chunk);
}

}  // namespace

int sd_xfer_h2d(sd_ctx *ctx, void *dev, const void *host, size_t bytes) { return xfer(ctx, dev, host, bytes, hipMemcpyHostToDevice); }
int sd_xfer_d2h(sd_ctx *ctx, void *host, const void *dev, size_t bytes) { return xfer(ctx, host, dev, bytes, hipMemcpyDeviceToHost); }

void sd_xfer_release(sd_ctx *ctx) {
  for (int k = 0; k < NBUF; ++k) {
    if (ctx->xfer_buf[k]) (void)hipHostFree(ctx->xfer_buf[k]);
    if (ctx->xfer_ev[k]) (void)hipEventDestroy(ctx->xfer_ev[k]);
    ctx->xfer_buf[k] = nullptr; ctx->xfer_ev[k] = nullptr;
  }
  ctx->xfer_chunk = 0;
  delete ctx->xfer_team;
  ctx->xfer_team = nullptr;
}
