// Communicators of the sharded recursion-level calls (include/spindyn.h, "sharded recursions").
//
// The reference has no distributed layer; the closest site is the thread loop over the momenta
// in src/KPM_Sqw.jl:218.  A sharded recursion needs exactly two things from its peers: the halo exchange of one vector per
// apply (whole tiles, sd_model_shard_slabs) and the sum of a few doubles per reduction.  Two implementations behind one
// internal interface:
//   * RCCL (sd_comm_rccl_create): grouped ncclSend/ncclRecv of the slabs on a communication stream of its own, fenced
//     against the compute stream with two events, and ncclAllReduce of the device scalars ON the compute stream -- nothing
//     in a recursion step touches the host.  librccl is opened at run time (the copy torch already loaded, if any), so
//     the library has no link-time dependency on it.
//   * callbacks (sd_comm_from_callbacks): the caller moves the bytes.  The Python mirror wraps torch.distributed this way
//     (gloo in the rehearsals on one GPU); scalars take one host round trip per reduction.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "sd_internal.hpp"

namespace {

// the few RCCL entry points used, resolved with dlsym (signatures: /opt/rocm/include/rccl/rccl.h)
struct nccl_uid { char internal[128]; };
typedef void *nccl_comm_t;
struct NcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(nccl_uid *) = nullptr;
  int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;   // ncclFloat64, ncclSum

NcclApi *nccl_api(std::string &err) {
  static NcclApi api;
  static bool tried = false;
  if (tried) { if (!api.lib) err = "librccl could not be loaded"; return api.lib ? &api : nullptr; }
  tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so"};
  void *h = nullptr;
  for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy already in the process (torch's)
  for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { err = std::string("dlopen(librccl): ") + dlerror(); return nullptr; }
  bool ok = true;
  auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) ok = false; return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { err = "librccl lacks an expected symbol"; return nullptr; }
  api.lib = h;
  return &api;
}

#define SD_NCCL(ctx, c, call)                                                                          \
  do {                                                                                                 \
    int r__ = (call);                                                                                  \
    if (r__ != 0) return sd_set_err((ctx), SD_ECOMM, std::string(#call) + ": " + (c)->api->GetErrorString(r__)); \
  } while (0)

}  // namespace

struct sd_comm {
  int kind = 0;                 // 0 callbacks, 1 RCCL
  int rank = 0, nranks = 1;
  sd_comm_callbacks cb{};
  // RCCL
  NcclApi *api = nullptr;
  nccl_comm_t nccl = nullptr;
  int device = 0;
  hipStream_t xstream = nullptr;            // the slab exchange runs here, beside the interior tiles on the compute stream
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  // routed exchange (sd_comm_set_exchange_ops): the halo exchange as an explicit list of sends / receives in batches, with a
  // relay buffer of this rank for pieces that travel owner -> relay -> receiver; empty: the model's slab lists, all at once
  std::vector<sd_xop> xops;
  int64_t relay_elems = 0;
  void *relay = nullptr;
  size_t relay_bytes = 0;
};

int sd_comm_nranks(const sd_comm *c) { return c ? c->nranks : 1; }

int sd_comm_exchange_start(sd_ctx *ctx, sd_comm *c, const sd_model *m, int dtype, const void *src, void *halo) {
  if (!c || m->nranks == 1) return SD_OK;
  if (c->nranks != m->nranks || c->rank != m->rank)
    return sd_set_err(ctx, SD_EARG, "communicator rank/size does not match the model's shard");
  if (c->kind == 0) {
    int rc = c->cb.exchange_start(c->cb.user, dtype, src, halo);
    return rc ? sd_set_err(ctx, SD_ECOMM, "exchange_start callback failed") : SD_OK;
  }
  const size_t per = dtype == SD_C128 ? 2 : 1;      // doubles per element
  // everything queued so far on the compute stream (the pack kernel, the previous apply's readers of the halo) first
  SD_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
  SD_HIP(ctx, hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
  // an error inside the group must not leave the thread's group open (later RCCL calls would be deferred for ever): note the
  // first failure, close the group regardless, then report
  if (!c->xops.empty()) {
    // A routed exchange: one RCCL group per batch, in order, on the exchange stream.  A relay's receive of a piece (batch b) and its
    // forwarding (batch b + 1) are ordered by that stream; consecutive groups overlap on the links, so the second hop of one
    // slice travels while the first hop of the next is under way.
    const size_t need = (size_t)std::max<int64_t>(c->relay_elems, 1) * 16;
    if (c->relay_bytes < need) {
      if (c->relay) (void)hipFree(c->relay);
      c->relay = nullptr; c->relay_bytes = 0;
      SD_HIP(ctx, hipMalloc(&c->relay, need));
      c->relay_bytes = need;
    }
    int nb = 0;
    for (const sd_xop &o : c->xops) nb = std::max(nb, o.batch + 1);
    size_t at = 0;
    for (int b = 0; b < nb; ++b) {
      SD_NCCL(ctx, c, c->api->GroupStart());
      int bad = 0;
      for (; at < c->xops.size() && c->xops[at].batch == b && !bad; ++at) {
        const sd_xop &o = c->xops[at];
        double *base = o.buf == 0 ? (double *)const_cast<void *>(src) : o.buf == 1 ? (double *)halo : (double *)c->relay;
        double *p = base + (size_t)o.offset * per;
        bad = o.kind == 0 ? c->api->Send(p, (size_t)o.count * per, NCCL_DOUBLE, o.peer, c->nccl, c->xstream)
                          : c->api->Recv(p, (size_t)o.count * per, NCCL_DOUBLE, o.peer, c->nccl, c->xstream);
      }
      const int end = c->api->GroupEnd();
      if (bad) return sd_set_err(ctx, SD_ECOMM, std::string("ncclSend/ncclRecv of the routed halo exchange: ") + c->api->GetErrorString(bad));
      if (end) return sd_set_err(ctx, SD_ECOMM, std::string("ncclGroupEnd: ") + c->api->GetErrorString(end));
    }
    SD_HIP(ctx, hipEventRecord(c->ev_done, c->xstream));
    return SD_OK;
  }
  SD_NCCL(ctx, c, c->api->GroupStart());
  int bad = 0;
  for (const sd_slab &s : m->recv_slabs) {          // recv offsets are counted from the start of [owned | halo]
    if (bad) break;
    bad = c->api->Recv((double *)halo + (size_t)(s.local_offset - m->n_local) * per, (size_t)s.count * per, NCCL_DOUBLE, s.peer,
                       c->nccl, c->xstream);
  }
  for (const sd_slab &s : m->send_slabs) {
    if (bad) break;
    bad = c->api->Send((const double *)src + (size_t)s.local_offset * per, (size_t)s.count * per, NCCL_DOUBLE, s.peer, c->nccl,
                       c->xstream);
  }
  const int end = c->api->GroupEnd();
  if (bad) return sd_set_err(ctx, SD_ECOMM, std::string("ncclSend/ncclRecv of the halo exchange: ") + c->api->GetErrorString(bad));
  if (end) return sd_set_err(ctx, SD_ECOMM, std::string("ncclGroupEnd: ") + c->api->GetErrorString(end));
  SD_HIP(ctx, hipEventRecord(c->ev_done, c->xstream));
  return SD_OK;
}

int sd_comm_exchange_wait(sd_ctx *ctx, sd_comm *c, const sd_model *m) {
  if (!c || m->nranks == 1) return SD_OK;
  if (c->kind == 0) {
    int rc = c->cb.exchange_wait(c->cb.user);
    return rc ? sd_set_err(ctx, SD_ECOMM, "exchange_wait callback failed") : SD_OK;
  }
  SD_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));   // the compute stream goes on once the halo has landed
  return SD_OK;
}

int sd_comm_allreduce_dev(sd_ctx *ctx, sd_comm *c, double *vals_dev, int count) {
  if (!c || c->nranks == 1 || count <= 0) return SD_OK;
  if (c->kind == 1) {
    SD_NCCL(ctx, c, c->api->AllReduce(vals_dev, vals_dev, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->nccl, ctx->stream));
    return SD_OK;
  }
  if (count > 16) return sd_set_err(ctx, SD_EINTERNAL, "host-staged reduction of more than 16 scalars");
  double *h = ctx->h_scalars;
  SD_HIP(ctx, hipMemcpyAsync(h, vals_dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (c->cb.allreduce_sum(c->cb.user, h, count)) return sd_set_err(ctx, SD_ECOMM, "allreduce_sum callback failed");
  SD_HIP(ctx, hipMemcpyAsync(vals_dev, h, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));    // h is reused by the next reduction
  return SD_OK;
}

extern "C" {

int sd_comm_from_callbacks(const sd_comm_callbacks *cb, int rank, int nranks, sd_comm **out) {
  if (!out) return SD_EARG;
  *out = nullptr;
  if (!cb || !cb->exchange_start || !cb->exchange_wait || !cb->allreduce_sum || nranks < 1 || rank < 0 || rank >= nranks)
    return SD_EARG;
  sd_comm *c = new (std::nothrow) sd_comm();
  if (!c) return SD_ENOMEM;
  c->kind = 0; c->cb = *cb; c->rank = rank; c->nranks = nranks;
  *out = c;
  return SD_OK;
}

int sd_comm_set_exchange_ops(sd_comm *c, const sd_xop *ops, int64_t n_ops, int64_t relay_elems) {
  if (!c || n_ops < 0 || relay_elems < 0 || (n_ops > 0 && !ops)) return SD_EARG;
  if (c->kind != 1) return SD_EARG;                 // a callback communicator routes its own exchange
  int last = 0;
  for (int64_t i = 0; i < n_ops; ++i) {
    const sd_xop &o = ops[i];
    if (o.batch < last || o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank || (o.kind != 0 && o.kind != 1) || o.buf < 0 ||
        o.buf > 2 || o.offset < 0 || o.count <= 0 || (o.kind == 0 && o.buf == 1) || (o.kind == 1 && o.buf == 0) ||
        (o.buf == 2 && o.offset + o.count > relay_elems))
      return SD_EARG;                               // batches ascending; sends read the vector or the relay buffer, receives fill the halo or the relay buffer
    last = o.batch;
  }
  try { c->xops.assign(ops, ops + n_ops); } catch (const std::bad_alloc &) { return SD_ENOMEM; }
  c->relay_elems = relay_elems;
  return SD_OK;
}

int sd_comm_rccl_unique_id(void *id128) {
  if (!id128) return SD_EARG;
  std::string err;
  NcclApi *api = nccl_api(err);
  if (!api) return SD_ECOMM;
  nccl_uid id;
  if (api->GetUniqueId(&id) != 0) return SD_ECOMM;
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return SD_OK;
}

int sd_comm_rccl_create(sd_ctx *ctx, int rank, int nranks, const void *id128, sd_comm **out) {
  if (!ctx || !out) return SD_EARG;
  *out = nullptr;
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return sd_set_err(ctx, SD_EARG, "bad rank/nranks/id");
  std::string err;
  NcclApi *api = nccl_api(err);
  if (!api) return sd_set_err(ctx, SD_ECOMM, err);
  sd_comm *c = new (std::nothrow) sd_comm();
  if (!c) return SD_ENOMEM;
  c->kind = 1; c->api = api; c->rank = rank; c->nranks = nranks; c->device = ctx->device;
  nccl_uid id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  if (e != hipSuccess) { sd_comm_destroy(c); return sd_set_err(ctx, SD_EHIP, std::string("communicator streams: ") + hipGetErrorString(e)); }
  int r = api->CommInitRank(&c->nccl, nranks, id, rank);
  if (r != 0) {
    std::string msg = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
    c->nccl = nullptr;
    sd_comm_destroy(c);
    return sd_set_err(ctx, SD_ECOMM, msg);
  }
  *out = c;
  return SD_OK;
}

// Diagnostic: exercises every RCCL entry point the communicator uses -- ncclAllReduce of two device doubles (sum over nranks
// copies of (1, 2)) and a grouped ncclSend/ncclRecv of 1024 doubles round the ring of ranks (to itself with one rank) on the
// communication stream, fenced by the two events as in the halo exchange -- and checks the bytes.  With one rank it needs
// no peer: the signature / enum check that a one-GPU box can do; bench.py runs it on every rank before the C RCCL leg.
int sd_comm_selftest(sd_ctx *ctx, sd_comm *c) {
  if (!ctx || !c) return SD_EARG;
  if (c->kind != 1) return sd_set_err(ctx, SD_EARG, "self-test is for the RCCL communicator");
  SD_HIP(ctx, hipSetDevice(ctx->device));
  const int n = 1024;
  double *d = nullptr;
  SD_HIP(ctx, hipMalloc((void **)&d, sizeof(double) * (2 * n + 2)));
  std::vector<double> h(2 * n + 2, 0.0);
  for (int i = 0; i < n; ++i) h[i] = 0.5 * i + 1.0 + 1000.0 * c->rank;
  h[2 * n] = 1.0; h[2 * n + 1] = 2.0;
  int rc = SD_OK;
  auto body = [&]() -> int {
    SD_HIP(ctx, hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    SD_NCCL(ctx, c, c->api->AllReduce(d + 2 * n, d + 2 * n, 2, NCCL_DOUBLE, NCCL_SUM, c->nccl, ctx->stream));
    SD_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
    SD_HIP(ctx, hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
    // with peers: the payload goes round a ring (to rank+1, from rank-1) and carries the sender's rank; alone: to itself
    const int to = (c->rank + 1) % c->nranks, from = (c->rank + c->nranks - 1) % c->nranks;
    SD_NCCL(ctx, c, c->api->GroupStart());
    int bad = c->api->Recv(d + n, n, NCCL_DOUBLE, from, c->nccl, c->xstream);
    if (!bad) bad = c->api->Send(d, n, NCCL_DOUBLE, to, c->nccl, c->xstream);
    const int end = c->api->GroupEnd();             // closed whatever happened inside
    if (bad) return sd_set_err(ctx, SD_ECOMM, std::string("self-test ncclSend/ncclRecv: ") + c->api->GetErrorString(bad));
    if (end) return sd_set_err(ctx, SD_ECOMM, std::string("self-test ncclGroupEnd: ") + c->api->GetErrorString(end));
    SD_HIP(ctx, hipEventRecord(c->ev_done, c->xstream));
    SD_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    SD_HIP(ctx, hipMemcpyAsync(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i)
      if (h[n + i] != 0.5 * i + 1.0 + 1000.0 * from) return sd_set_err(ctx, SD_ECOMM, "ring send/recv returned wrong data");
    if (h[2 * n] != 1.0 * c->nranks || h[2 * n + 1] != 2.0 * c->nranks) return sd_set_err(ctx, SD_ECOMM, "all-reduce returned a wrong sum");
    return SD_OK;
  };
  rc = body();
  (void)hipFree(d);
  return rc;
}

void sd_comm_destroy(sd_comm *c) {
  if (!c) return;
  if (c->kind == 1) {
    (void)hipSetDevice(c->device);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->nccl && c->api) (void)c->api->CommDestroy(c->nccl);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->relay) (void)hipFree(c->relay);
  }
  delete c;
}

}  // extern "C"
